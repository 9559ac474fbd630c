import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sparta_amd as sa
N = 128
m = sa.gen.rmat(16, 1 << 20, seed=3, symmetrize=True, pattern_only=False)
g = sa.BlockingEngine(blocking_algo=7, tau=0.4, col_block_size=64).GetGrouping(m)
v = sa.VBR().fill_from_CSR_inplace(m, g, 64)
B = torch.rand(v.cols * N, device="cuda") - 0.5
C = torch.zeros(v.rows * N, device="cuda")
for K in ("3", "10", "30", "100"):
    for path in ("auto", "stream", "class"):
        os.environ["SPARTA_SPARSE_K"] = K; os.environ["SPARTA_PATH"] = path
        d = v.to_device(0)
        for _ in range(5): d.spmm(B, C, N)
        d.set_class_timing(True)
        ts = []
        for _ in range(20):
            t = d.spmm(B, C, N, timed=True); ts.append((t, d.class_times()))
        t = np.median([x[0] for x in ts]); ct = ts[-1][1]
        i = d.info()
        print("K=%-4s %-6s total %7.1f us | %s | tiles %d/%d/%d steps %d split %d path %d sp_rows %d %s" % (K, path, t * 1e3, {k: round(x * 1e3, 1) for k, x in ct.items()},
              i["tiles16"], i["tiles32"], i["tiles64"], i["stream_steps"], i["split_tiles"], i["last_path"], i["sparse_rows"], d.sparse_info()))
        d.close()
