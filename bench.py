#!/usr/bin/env python3
"""bench.py -- headline benchmark of the block-sparse SpMM path (BASELINE.json: "Block-sparse SpMM GFLOP/s +
%HBM/MFMA roofline").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: C = A_vbs * B with A already reordered
(Jaccard row clustering), built into VBS and resident in HBM, B and C resident in HBM.

  N = 1  : BASELINE.json configs[1] -- SuiteSparse `cant` shape (62 451^2, 3-dof FEM, ~4.3 M nnz; generated, there is no
           network), B = 128 dense columns, fp32, reference layouts (B, C column-major).
  N > 1  : weak scaling of the same workload: the mesh grows N x along z, rank r owns row slab r of A and the matching
           row shard of B; each step = ONE collective on B over RCCL/xGMI + the local SpMM; no collective on C.
           --exchange allgather : B replicated by one all-gather (what a slab that touches all of B needs);
           --exchange blocks    : one all-to-all of only the row-blocks of B each slab touches (plan-time lists), overlapped
                                  with the product against the rank's own shard (sparta_amd/dist.py: RowBlockExchange);
           --exchange auto      : blocks when the slabs need < 50 % of the other shards, else allgather (default).

value = useful GFLOP/s of the whole job = 2 * nnz * n_cols * n_gpus_units / time (dense-block padding is NOT counted).
The JSON line also carries the roofline of the dominant kernel and a CPU baseline (the reference's own
VBR::multiply, compiled from its sources into oracle/_ref, timed on this box's host).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 dense peak
PEAK_CLOCK_MHZ = 2400.0        # the clock that peak is quoted at (256 CUs x 4 SIMD x 64 lanes x 2 flop x 2.4 GHz)


def mixed_roofline_seconds(row_part, nzcount, w, n_cols, cols, accumulate=False):
    """T_lb = sum over block-rows of max(bytes_alg / BW, flops_exec / P) + |B| / BW."""
    h = np.diff(np.asarray(row_part, np.int64)).astype(np.float64)
    nb = np.asarray(nzcount, np.float64)
    flops = 2.0 * nb * h * w * n_cols
    byts = nb * h * w * 4.0 + nb * 4.0 + 16.0 + h * n_cols * 4.0 * (2.0 if accumulate else 1.0)
    t = np.maximum(byts / (PEAK_HBM_GBS * 1e9), flops / (PEAK_MFMA_F32_TFLOPS * 1e12)).sum()
    t += cols * n_cols * 4.0 / (PEAK_HBM_GBS * 1e9)
    return float(t), float(flops.sum()), float(byts.sum() + cols * n_cols * 4.0)


def _stage_collectives_through_host(dist, sa):
    """--backend gloo (debug): gloo has no device collectives for these calls, so the few this file uses go through host copies."""
    import torch

    class _Done:
        def wait(self):
            return True

    real_ag, real_ar, real_bar = dist.all_gather_into_tensor, dist.all_reduce, dist.barrier

    def all_gather_into_tensor(out, inp, group=None):
        o = torch.empty(out.shape, dtype=out.dtype)
        real_ag(o, inp.cpu(), group=group)
        out.copy_(o)

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None):
        c = t.cpu()
        real_ar(c, op=op, group=group)
        t.copy_(c)

    def a2a(self):
        torch.cuda.synchronize()
        o = torch.empty(self._recv_view.shape, dtype=self._recv_view.dtype)
        dist.all_to_all_single(o, self._send_view.cpu(), self.out_splits, self.in_splits, group=self.group)
        self._recv_view.copy_(o)
        return _Done()

    dist.all_gather_into_tensor, dist.all_reduce = all_gather_into_tensor, all_reduce
    dist.barrier = lambda *a, **k: real_bar()
    sa.dist.RowBlockExchange._all_to_all = a2a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--tau", type=float, default=0.6)
    ap.add_argument("--algo", type=int, default=5, help="reference BlockingType: 3 iterative_clocked, 5 iterative_max_size (Keeper)")
    ap.add_argument("--row-block", type=int, default=32, help="max / fixed block-row height (-B)")
    ap.add_argument("--force-fixed", type=int, default=1, help="-F: re-chunk clusters into equal heights")
    ap.add_argument("--col-block", type=int, default=32)
    ap.add_argument("--dtype", choices=["f32", "f16", "bf16"], default="f32",
                    help="storage type of A and B on the device (accumulation and C are always fp32)")
    ap.add_argument("--ncols", type=int, default=128)
    ap.add_argument("--fixed-height", type=int, default=0, help="reorder OFF: fixed block-row height instead of clustering")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the steps from a captured HIP graph (measured: no gain -- the step is one 60 us kernel and eager launches already queue ahead)")
    ap.add_argument("--workload", choices=["cant", "rmat"], default="cant",
                    help="cant: BASELINE configs[1] (the default, what `value` is quoted on); rmat: configs[3] in miniature -- R-MAT 2^scale rows, "
                         "10 edges per row symmetrised, reordered by blocking_algo 7 (single GPU; carried by the sparse-row kernels: roofline.bound = hbm)")
    ap.add_argument("--rmat-scale", type=int, default=20)
    ap.add_argument("--matrix", default=None,
                    help="run the single-GPU pipeline on a matrix file instead of the synthetic workload (.mtx MatrixMarket or .el edge list, read as "
                         "documented = SPARTA_IO_STRICT): e.g. the real SuiteSparse cant.mtx where a copy is at hand (none can be fetched here)")
    ap.add_argument("--exchange", choices=["auto", "allgather", "blocks"], default="auto",
                    help="N > 1: how the ranks' shards of B reach the slabs (see the module docstring)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: DEBUG ONLY -- several ranks on ONE GPU (a one-GPU box), collectives staged through host memory; checks the "
                         "multi-rank logic end to end, its timings mean nothing")
    ap.add_argument("--dist-path", action="store_true", help="run the multi-GPU code path (slab + all-gather + gathered SpMM) even with one rank")
    args = ap.parse_args()

    import torch
    import sparta_amd as sa

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    n_gpus = max(args.gpus, 1)
    if n_gpus > 1 and world == 1:
        raise SystemExit("for --gpus > 1 launch with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the SpMM path has no CPU fallback")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    distributed = world > 1 or args.dist_path
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
            _stage_collectives_through_host(dist, sa)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    w, N = args.col_block, args.ncols
    t0 = time.time()
    # ---- workload ------------------------------------------------------------------------------------------------
    rmat = args.workload == "rmat"
    if rmat and distributed:
        raise SystemExit("--workload rmat is a single-GPU option")
    if args.matrix:
        if distributed:
            raise SystemExit("--matrix is a single-GPU option")
        fmt = sa._lib.FMT_MTX if args.matrix.lower().endswith(".mtx") else sa._lib.FMT_EL
        m = sa.CSR.read_from_edgelist(args.matrix, mat_fmt=fmt, mode=sa._lib.IO_STRICT)
        n_local, shard_rows = m.rows, None
        rmat = False
    elif rmat:
        m = sa.gen.rmat(args.rmat_scale, 10 << args.rmat_scale, seed=3, symmetrize=True, pattern_only=False)
        n_local, shard_rows = m.rows, None
        if args.algo == 5 and args.tau == 0.6 and w == 32:       # untouched defaults: the settings this workload is meant for
            args.algo, args.tau, args.force_fixed, w = 7, 0.4, 0, 64
            args.col_block = 64
    elif not distributed:
        m = sa.gen.cant_like(seed=2)
        n_local, shard_rows = m.rows, None
    else:
        m, n_local, shard_rows = sa.gen.fem3d_slab(9, 9, 257, rank, world, dof=3, pad_to=w, seed=2)
    t_gen = time.time() - t0
    t0 = time.time()
    if args.fixed_height:
        eng = sa.BlockingEngine(blocking_algo="fixed_size", row_block_size=args.fixed_height, col_block_size=w)
    else:
        eng = sa.BlockingEngine(blocking_algo=args.algo, tau=args.tau, col_block_size=w, row_block_size=args.row_block,
                                force_fixed_size=bool(args.force_fixed), sim_measure=1)
    grouping = eng.GetGrouping(m)
    t_reorder = time.time() - t0
    t0 = time.time()
    vb = sa.VBR().fill_from_CSR_inplace(m, grouping, w, args.row_block, bool(args.force_fixed) and not args.fixed_height)
    t_build = time.time() - t0
    h16 = args.dtype != "f32"
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    d = vb.to_device(local_rank, dtype={"f32": sa.F32, "f16": sa.F16, "bf16": sa.BF16}[args.dtype])
    info = d.info()

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    ldb = vb.cols
    if not distributed:
        B = (torch.rand(vb.cols * N, generator=g, dtype=torch.float32) - 0.5).to(dev)      # column-major, ld = cols
        if h16:                                                # 16-bit B, leading dimension padded to a multiple of 8 elements
            ldb = (vb.cols + 7) // 8 * 8
            B32 = B
            B = torch.zeros(ldb * N, dtype=torch.float16 if args.dtype == "f16" else torch.bfloat16, device=dev)
            B.view(N, ldb)[:, :vb.cols] = B32.view(N, vb.cols).to(B.dtype)
        B_shard = B_gath = None
    else:
        B_shard = (torch.rand(shard_rows * N, generator=g, dtype=torch.float32) - 0.5).to(dev).to(tdt)   # column-major, ld = shard_rows
        B_gath = torch.empty(world * shard_rows * N, dtype=tdt, device=dev)
    C = torch.zeros(vb.rows * N, dtype=torch.float32, device=dev)

    # ---- N > 1: which exchange ------------------------------------------------------------------------------------
    ex, exchange, exchange_note, B_tiles = None, None, "", None
    if distributed:
        dist.all_gather_into_tensor(B_gath, B_shard)              # setup: reference result for the self-check, B for the CPU baseline
        exchange = args.exchange
        if exchange != "allgather":
            # the same shard in the row-block-tiled layout (block jb = one contiguous w x N column-major tile)
            B_tiles = B_shard.view(N, shard_rows // w, w).permute(1, 0, 2).contiguous().view(-1)
            ex = sa.dist.RowBlockExchange(vb, rank, world, shard_rows, N, device=local_rank,
                                          dtype={"f32": sa.F32, "f16": sa.F16, "bf16": sa.BF16}[args.dtype])      # collective: need lists
            if exchange == "auto":
                exchange = "blocks" if ex.needed_fraction < 0.5 else "allgather"
        if exchange == "blocks":
            # self-check on the real collective: both exchanges must give the same C (two partial sums vs one: fp32 re-association)
            d.spmm_gathered(B_gath, shard_rows, C, N, accumulate=False)
            C_chk = torch.empty_like(C)
            err = float("inf")
            try:
                ex.step(B_tiles, C_chk)
                torch.cuda.synchronize()
                err = float((C_chk - C).abs().max() / C.abs().max().clamp_min(1e-30))
            except Exception as e:                                 # reported, never silent: the line below names the fallback
                print("rank %d: row-block exchange raised %r" % (rank, e), file=sys.stderr)
            ok = torch.tensor([1.0 if err < 1e-4 else 0.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                print("rank %d: row-block exchange self-check FAILED (max rel err %.3e): using the all-gather" % (rank, err), file=sys.stderr)
                exchange, exchange_note = "allgather", " (row-block exchange failed its self-check)"
            del C_chk
        if exchange == "allgather" and ex is not None:
            ex.close()
            ex = None
    dmain, vbm = (ex.d_own, ex.own) if ex is not None else (d, vb)      # the handle / matrix whose kernel dominates a step
    info = dmain.info()

    def step():
        if not distributed:
            d.spmm(B, C, N, accumulate=False, ldb=ldb)
        elif ex is not None:
            ex.step(B_tiles, C)                                    # pack + ONE all-to-all of the needed row-blocks || own product; + remote product
        else:
            dist.all_gather_into_tensor(B_gath, B_shard)          # the one exchange step (RCCL over xGMI)
            d.spmm_gathered(B_gath, shard_rows, C, N, accumulate=False)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # plan time, once per handle and shape: the first product on a handle measures its two MFMA paths and keeps the faster one
    # (sparta_vbs_spmm, "plan-time autotune") and sizes its scratch buffers.  That is part of building the plan, like the tile lists
    # sparta_vbs_create makes -- it must not land in the timed region when the caller asks for no warm-up.
    step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    # Optional: capture G consecutive steps in a HIP graph (sparta_vbs_spmm is stream-ordered and capture-safe once the
    # plan-time autotune of the first call is over) and replay it K / G times: EXACTLY K steps in the timed region.
    graph, G = None, 1
    if not distributed and args.graph:
        G = max(g_ for g_ in range(1, 11) if args.steps % g_ == 0)
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                step()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    for _ in range(G):
                        step()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            graph.replay()                                   # one untimed replay
            torch.cuda.synchronize()
        except Exception as e:                               # capture not available: eager launches
            print("HIP graph capture failed (%s); launching eagerly" % e, file=sys.stderr)
            graph, G = None, 1
    fence()
    t_start = time.perf_counter()
    if graph is not None:
        for _ in range(args.steps // G):
            graph.replay()
    for _ in range(0 if graph is not None else args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t_start
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3

    # ---- per-kernel device time (HIP events on the launch stream, one pair per kernel launch) ----------------------
    dmain.set_class_timing(True)
    kt, kc = {}, {}
    for _ in range(min(args.steps, 50)):
        step()
        for k, v in dmain.class_times().items():
            kt.setdefault(k, []).append(v)
        for k, v in dmain.clock_mhz().items():    # s_memtime / s_memrealtime over the kernel: the clock the pipes really ran at
            if v > 0:
                kc.setdefault(k, []).append(v)
    dmain.set_class_timing(False)
    kernel_ms = {k: float(np.mean(v)) for k, v in kt.items()}
    kernel_mhz = {k: float(np.mean(v)) for k, v in kc.items()}
    # A step that is ONE kernel launch (aligned stream plan, one tile type): time a run of launches with one event pair on the
    # launch stream instead of one pair per launch -- the per-launch pairs above add ~2 us of event traffic to a 60 us kernel.
    live = [k for k, v in kernel_ms.items() if v > 0]
    ntypes = int(info["tiles64"] > 0) + int(info["tiles16"] + info["tiles32"] > 0)
    if not distributed and len(live) == 1 and (live[0] != "stream" or ntypes == 1):
        reps = min(args.steps, 500)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(10):
            step()
        e0.record()
        for _ in range(reps):
            step()
        e1.record()
        torch.cuda.synchronize()
        kernel_ms[live[0]] = e0.elapsed_time(e1) / reps
    path = {1: "stream", 2: "class", 3: "generic"}.get(dmain.info()["last_path"], "?")

    nnz_local = m.nztot()
    nnz_total = nnz_local
    if distributed:
        tt = torch.tensor([float(nnz_local)], dtype=torch.float64, device=dev)
        dist.all_reduce(tt)
        nnz_total = float(tt.item())
    useful_gflops = 2.0 * nnz_total * N / (ms_per_step * 1e-3) / 1e9

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (rank 0's launch) ------------------------------------------------------
    h = np.diff(vb.row_part)
    flops_exec_full = 2.0 * float(vb.nztot) * N
    # stored area handled by each kernel: the stream kernel takes everything; the per-class kernels take the tiles of
    # their height class (block-rows are cut into <=64-row tiles; a remainder <=32 / <=16 rows goes to the thinner class)
    area = {"stream": float(vbm.nztot), "fixup": 0.0, "class16": 0.0, "class32": 0.0, "class64": 0.0}
    for hh, nb in zip(h, vbm.nzcount):
        r = int(hh)
        while r > 0:
            mt = min(r, 64) if r > 32 else r
            c = "class64" if r > 32 else ("class32" if r > 16 else "class16")
            area[c] += float(mt) * w * float(nb)
            r -= mt
    t_lb, flops_exec, bytes_alg = mixed_roofline_seconds(vbm.row_part, vbm.nzcount, w, N, vbm.cols)
    kernel_ms_total = sum(kernel_ms.values())
    dom = max(kernel_ms, key=lambda k: kernel_ms[k]) if kernel_ms else "stream"
    dom_tflops = 2.0 * area.get(dom, 0.0) * N / (kernel_ms[dom] * 1e-3) / 1e12 if kernel_ms.get(dom, 0) > 0 else 0.0
    kname = {"stream": "vbs_spmm_f32_stream_kernel", "fixup": "vbs_spmm_f32_fixup_kernel", "class16": "vbs_spmm_f32_kernel<16,...>",
             "class32": "vbs_spmm_f32_kernel<32,1,4,1,1,...>", "class64": "vbs_spmm_f32_kernel<32,2,2,1,2,...>"}.get(dom, dom)
    # HBM bytes per launch of that kernel from rocprofv3 PMC passes (profiles/traffic.json, produced by
    # scripts/profile_bench.sh on the same command; FETCH_SIZE doubled per MI355X_MICROARCH.md) -- only reported when the
    # profiled workload is the one running now
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json" if args.dtype == "f32" else "traffic_%s.json" % args.dtype)))
        wk = tj.get("workload", {})
        if (wk.get("vbs_area") == int(vb.nztot) and wk.get("n_cols") == N and wk.get("kernel_path") == path and not distributed
                and wk.get("dtype", "f32") == args.dtype):
            traffic = round(float(tj["hbm_bytes_per_launch"]))
    except Exception:
        traffic = None
    roofline = {
        "bound": "mfma", "achieved": round(dom_tflops, 3), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
        "frac": round(dom_tflops / PEAK_MFMA_F32_TFLOPS, 4), "traffic": traffic,
        "kernel": kname, "kernel_ms": round(kernel_ms.get(dom, 0.0), 5), "path": path,
        "kernels_ms": {k: round(v, 5) for k, v in kernel_ms.items()},
        "all_kernels_tflops_exec": round(flops_exec / (kernel_ms_total * 1e-3) / 1e12, 3) if kernel_ms_total > 0 else 0.0,
        "mixed_roofline_frac": round(t_lb / (kernel_ms_total * 1e-3), 4) if kernel_ms_total > 0 else 0.0,
        "algorithmic_gbs": round(bytes_alg / (kernel_ms_total * 1e-3) / 1e9, 1) if kernel_ms_total > 0 else 0.0,
        "algorithmic_bytes": round(bytes_alg),
    }
    sp = dmain.sparse_info()
    if dom == "sparse":
        try:                                   # PMC-measured HBM-side bytes per step of these kernels (scripts/pmc_rmat.sh), same workload only
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_rmat.json")))
            wk = tj.get("workload", {})
            if wk.get("vbs_area") == int(vb.nztot) and wk.get("n_cols") == N and wk.get("dtype") == args.dtype:
                traffic = round(float(tj["hbm_bytes_per_step"]))
        except Exception:
            pass
        # the matrix is carried by the sparse-row kernels: HBM-bound.  Algorithmic bytes: one N-float row of B per nonzero + its
        # (column, value) + the rows of C once (the transposes of a column-major B / C are implementation traffic, not counted)
        bytes_sp = float(sp["nnz"]) * (N * (2.0 if h16 else 4.0) + 8.0) + float(sp["rows"]) * N * 4.0
        gbs = bytes_sp / (kernel_ms["sparse"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": traffic,
                    "kernel": "sparse_rows_kernel + sparse_segments_kernel (+ b_to_row_major / sparse_c_scatter transposes)",
                    "kernel_ms": round(kernel_ms["sparse"], 5), "path": path, "kernels_ms": {k: round(v_, 5) for k, v_ in kernel_ms.items()},
                    "algorithmic_bytes": round(bytes_sp), "sparse_rows": sp["rows"], "sparse_nnz": sp["nnz"], "hub_rows": sp["hub_rows"],
                    "note": "B rows of hub columns are served by L2 / Infinity Cache: the rate can exceed what HBM alone delivers"}
    if h16 and dom != "sparse":
        # 16-bit storage: the MFMAs take 1/8 (fp16 / bf16 dense peak ~2.5 PFLOP/s) of the fp32 time while the bytes only halve:
        # the kernel is bound by memory traffic.  Algorithmic bytes: packed 16-bit A (read once) + 16-bit B (once) + fp32 C.
        bytes16 = float(info["a_bytes"]) + 2.0 * ldb * N + 4.0 * vb.rows * N
        gbs = bytes16 / (kernel_ms_total * 1e-3) / 1e9 if kernel_ms_total > 0 else 0.0
        roofline.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                         "kernel": ("vbs_spmm_h16_direct_kernel" if (w % 64 != 0 and info["tiles64"] == 0 and os.environ.get("SPARTA_H16_PATH", "a")[0] != "l")
                                    or os.environ.get("SPARTA_H16_PATH", "a")[0] == "d" else "vbs_spmm_h16_stream_kernel"),
                         "algorithmic_bytes": round(bytes16), "algorithmic_gbs": round(gbs, 1),
                         "executed_tflops": round(dom_tflops, 3)})
        roofline.pop("mixed_roofline_frac", None)
    # `peak` is the 2.4 GHz figure of MI355X_MICROARCH.md; under this load the board does not hold 2.4 GHz (power), so the
    # measured shader clock and the fraction of the matrix peak AT THAT CLOCK are reported next to it (informational)
    if kernel_mhz.get(dom, 0) > 0:
        roofline["shader_clock_mhz"] = round(kernel_mhz[dom], 0)
    if kernel_mhz.get(dom, 0) > 0 and not h16:
        roofline["frac_at_measured_clock"] = round(dom_tflops / (PEAK_MFMA_F32_TFLOPS * kernel_mhz[dom] / PEAK_CLOCK_MHZ), 4)

    # ---- CPU baseline: the reference's own VBR::multiply on this host, 1 thread -----------------------------------
    cpu = None
    if not args.no_cpu_baseline:
        try:
            from oracle import ref, oracle as O
            if not distributed:
                Bh = (B.view(N, ldb)[:, :vb.cols].float().contiguous().view(-1) if h16 else B).cpu().numpy()
            else:
                Bh = sa.dist.gathered_to_colmajor(B_gath.float().cpu().numpy(), world, shard_rows, N)
            # bounded sample: a prefix of block-rows worth <= ~2e10 executed flops (about 10-20 s of scalar CPU work)
            per_row = 2.0 * np.diff(vb.row_part) * w * vb.nzcount * N
            cum = np.cumsum(per_row)
            nbr = int(np.searchsorted(cum, 2.0e10, side="right"))
            nbr = max(1, min(nbr, vb.block_rows))
            rows_s = int(vb.row_part[nbr])
            # useful flops of the sample: nnz of the (reordered) rows it covers
            perm = sa.get_permutation(grouping)
            nnz_s = int(np.diff(m.rowptr)[perm[:rows_s]].sum())
            cpu_reps = 1
            if ref.available():
                kind = "reference"
                if nbr == vb.block_rows:
                    rc = ref.RefCSR(m.rows, m.cols, m.rowptr, m.colidx.astype(np.int64), m.vals)
                    rv = ref.RefVBR(rc, grouping, w, args.row_block, bool(args.force_fixed) and not args.fixed_height)
                    t1 = time.perf_counter()
                    rv.multiply(Bh, N)
                    t_cpu = time.perf_counter() - t1
                    while t_cpu * cpu_reps < 10.0 and cpu_reps < 50:       # ~10 s of CPU work in all: repeat the whole multiply
                        t1 = time.perf_counter()
                        rv.multiply(Bh, N)
                        t_cpu = (t_cpu * cpu_reps + time.perf_counter() - t1) / (cpu_reps + 1)
                        cpu_reps += 1
                else:
                    kind = "port"
                    rv = None
            else:
                kind = "port"
                rv = None
            if rv is None:
                t1 = time.perf_counter()
                O.vbr_multiply(vb.rows, vb.cols, w, vb.row_part, vb.nzcount, vb.jab, vb.mab, Bh, N, block_row_range=(0, nbr))
                t_cpu = time.perf_counter() - t1
            cpu = {"value": round(2.0 * nnz_s * N / t_cpu / 1e9, 4), "unit": "GFLOP/s", "cores": 1, "kind": kind,
                   "sample": "block-rows [0,%d) of %d (%d rows, %d nnz), %d repetition%s, %.2f s each; executed dense-block rate %.2f GFLOP/s"
                             % (nbr, vb.block_rows, rows_s, nnz_s, cpu_reps, "" if cpu_reps == 1 else "s", t_cpu, float(cum[nbr - 1]) / t_cpu / 1e9)}
        except Exception as e:  # the baseline is a report, never a reason to lose the measurement
            cpu = {"value": None, "unit": "GFLOP/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}

    out = {
        "metric": "Block-sparse SpMM GFLOP/s", "value": round(useful_gflops, 2), "unit": "GFLOP/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "file" if args.matrix else "synthetic",
        "config": {
            "workload": ("%s (%d x %d, %d nnz), B = %d cols, %s" % (os.path.basename(args.matrix), m.rows, m.cols, nnz_local, N, args.dtype)) if args.matrix else
                        ("R-MAT 2^%d (a,b,c = 0.57,0.19,0.19; 10 edges per row, symmetrised: %d^2, %d nnz), B = %d cols, %s"
                         % (args.rmat_scale, m.rows, nnz_local, N, args.dtype)) if rmat else
                        ("cant-like FEM 9x9x257 mesh x 3 dof (62451^2, %d nnz), B = %d cols, fp32" % (nnz_local, N)) if not distributed else
                        ("row-partitioned FEM 9x9x%d mesh x 3 dof (%d^2 padded, %d nnz), B = %d cols, fp32, %s per step"
                         % (257 * world, world * shard_rows, int(nnz_total), N,
                            "1 all-gather of B" if ex is None else "1 all-to-all of the needed row-blocks of B")),
            "reorder": ("fixed height %d (reorder off)" % args.fixed_height) if args.fixed_height else
                       ("Jaccard %s tau=%.2f row_block=%d force_fixed=%d (reference flags -a %d -t %.2f -B %d -F %d -b %d)"
                        % ({3: "iterative_clocked", 5: "iterative_max_size/Keeper", 7: "LSH-bucketed (extension)"}.get(args.algo, str(args.algo)), args.tau, args.row_block,
                           args.force_fixed, args.algo, args.tau, args.row_block, args.force_fixed, w)),
            "col_block_size": w, "n_cols": N, "block_rows": int(vb.block_rows), "nonzero_blocks": int(len(vb.jab)),
            "vbs_area": int(vb.nztot), "fill": round(nnz_local / max(vb.nztot, 1), 4),
            "mean_block_row_height": round(float(h.mean()), 2),
            "tiles": {"16": info["tiles16"], "32": info["tiles32"], "64": info["tiles64"]}, "kernel_path": path,
            "executed_gflops": round(flops_exec_full * world / (ms_per_step * 1e-3) / 1e9, 1),
            "host_seconds": {"generate": round(t_gen, 2), "reorder": round(t_reorder, 2), "vbs_build": round(t_build, 2)},
            "parallelism": ("single GPU" if not distributed else
                            ("row-partition x%d, B all-gather%s" % (world, exchange_note)) if ex is None else
                            ("row-partition x%d, B row-block all-to-all (%d blocks sent / %d received by rank 0 per step = %.2f %% of the "
                             "all-gather's traffic), own-shard product overlapped" % (world, ex.n_send, ex.n_recv, 100.0 * ex.needed_fraction))),
            "launch": ("HIP graph of %d steps, replayed %d times" % (G, args.steps // G)) if graph is not None else "eager",
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    if distributed:
        dist.destroy_process_group()
    # RCCL writes its version banner to C stdout (fully buffered when redirected: it would surface AFTER our line at exit).
    # Flush C stdio first so that the JSON line is the last thing this process prints.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
