#!/usr/bin/env python3
"""bench.py -- headline benchmark of the block-sparse SpMM path (BASELINE.json: "Block-sparse SpMM GFLOP/s +
%HBM/MFMA roofline").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: C = A_vbs * B with A already reordered
(Jaccard row clustering), built into VBS and resident in HBM, B and C resident in HBM.

Workloads (--workload; BASELINE.json `configs`):
  cant       configs[1] (THE DEFAULT, what `value` is quoted on): SuiteSparse `cant` shape (62 451^2, 3-dof FEM, ~4.3 M nnz;
             generated, there is no network), B = 128 dense columns, fp32, reference layouts (B, C column-major).
             N > 1: weak scaling of the same workload (the mesh grows N x along z; one collective on B per step).
  ogbn-like  configs[2]: ogbn-products-sized power-law graph (2 449 029^2, ~124 M nnz, values 1; the leading corner of an
             R-MAT 2^22 graph, generated on the GPU), B = 256 columns, fp16 storage, fp32 accumulation.
  rmat       configs[3] / configs[4]: R-MAT 2^scale (a, b, c = 0.57, 0.19, 0.19), --rmat-density D (e.g. 0.001 = 0.1 %: D * n^2
             distinct nonzeros, generated on the GPU) or, without it, the round-1 miniature (10 edges per row, symmetrised);
             B = --ncols columns in --dtype.  Reorder ON = blocking_algo 7 (LSH-bucketed Jaccard clustering), reorder OFF =
             --fixed-height H (the reference's `-a 2 -F 1` arm: FixedBlocking, src/general/blocking.cpp:554-562); both arms go
             through the same entry (sparta_vbs_create_from_csr) and the same multiply.
             N > 1: STRONG scaling -- every rank generates the same seeded matrix and runs the same reorder, the block-rows
             are split into N contiguous ranges of equal cost, B is row-sharded, ONE all-gather of B per step over RCCL, no
             collective on C (configs[4]: --rmat-scale 23).

value = useful GFLOP/s of the whole job = 2 * nnz * n_cols / time (dense-block padding is NOT counted).  The JSON line also
carries the roofline of the dominant kernel and a CPU baseline (the reference's own VBR::multiply compiled into oracle/_ref, or
the oracle's restatement of it, timed on this box's host: 1 thread and all cores).

Timing: a short untimed pre-roll (--settle-ms, default 300 ms of steps) brings the GPU out of its idle clock state -- a freshly
leased board runs the first few thousand microseconds 20 % below its steady clock -- then W untimed warm-up steps, then
EXACTLY K timed steps between barrier + synchronize fences.  HIP events recorded on the launch stream bracket the same K steps
(`roofline.kernel_ms` when a step is one kernel launch).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 dense peak
PEAK_MFMA_H16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: fp16 / bf16 dense MFMA peak (~2.5 PF; the 5 PF headline is 2:1 sparse)
PEAK_CLOCK_MHZ = 2400.0        # the clock the MFMA peaks are quoted at
HBM_BYTES = 288e9


def mixed_roofline_seconds(row_part, nzcount, w, n_cols, cols, accumulate=False, s_a=4.0, s_b=4.0, peak_tflops=PEAK_MFMA_F32_TFLOPS):
    """T_lb = sum over block-rows of max(bytes_alg / BW, flops_exec / P) + |B| / BW."""
    h = np.diff(np.asarray(row_part, np.int64)).astype(np.float64)
    nb = np.asarray(nzcount, np.float64)
    flops = 2.0 * nb * h * w * n_cols
    byts = nb * h * w * s_a + nb * 4.0 + 16.0 + h * n_cols * 4.0 * (2.0 if accumulate else 1.0)
    t = np.maximum(byts / (PEAK_HBM_GBS * 1e9), flops / (peak_tflops * 1e12)).sum()
    t += cols * n_cols * s_b / (PEAK_HBM_GBS * 1e9)
    return float(t), float(flops.sum()), float(byts.sum() + cols * n_cols * s_b)


def _stage_collectives_through_host(dist, sa):
    """--backend gloo (debug): gloo has no device collectives for these calls, so the few this file uses go through host copies."""
    import torch

    class _Done:
        def wait(self):
            return True

    real_ag, real_ar, real_bar = dist.all_gather_into_tensor, dist.all_reduce, dist.barrier

    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        o = torch.empty(out.shape, dtype=out.dtype)
        real_ag(o, inp.cpu(), group=group)
        out.copy_(o)
        return _Done() if async_op else None

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


def _emit(out):
    # RCCL writes its version banner to C stdout (fully buffered when redirected: it would surface AFTER our line at exit).
    # Flush C stdio first so that the JSON line is the last thing this process prints.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="untimed pre-roll before the warm-up steps: run steps for about this long so that the GPU has left its idle clock state")
    ap.add_argument("--tau", type=float, default=None)
    ap.add_argument("--algo", type=int, default=None, help="reference BlockingType: 3 iterative_clocked, 5 iterative_max_size (Keeper); 7 = LSH-bucketed extension")
    ap.add_argument("--row-block", type=int, default=32, help="max / fixed block-row height (-B)")
    ap.add_argument("--force-fixed", type=int, default=None, help="-F: re-chunk clusters into equal heights")
    ap.add_argument("--col-block", type=int, default=None)
    ap.add_argument("--dtype", choices=["f32", "f16", "bf16"], default=None,
                    help="storage type of A and B on the device (accumulation and C are always fp32); default per workload")
    ap.add_argument("--ncols", type=int, default=None)
    ap.add_argument("--fixed-height", type=int, default=0, help="reorder OFF: fixed block-row height instead of clustering (-a 2 -F 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the steps from a captured HIP graph (measured: no gain -- the step is one 60 us kernel and eager launches already queue ahead)")
    ap.add_argument("--workload", choices=["cant", "rmat", "ogbn-like", "rmat-part", "cant-weak"], default=None,
                    help="default: cant on one GPU (configs[1]); rmat-part = configs[4] when --gpus N > 1 (strong scaling); cant-weak = round 2's weak-scaled FEM mesh")
    ap.add_argument("--rmat-scale", type=int, default=None)
    ap.add_argument("--slabs", type=int, default=8,
                    help="rmat-part on ONE GPU: the P parts of the P-rank job one after the other (the N = 1 comparator of --gpus P; a power of two)")
    ap.add_argument("--slab-sample", type=int, default=0, help="rmat-part on one GPU: run only this many of the --slabs parts (part 0 + a seeded choice), total extrapolated by cost")
    ap.add_argument("--pad-b", type=int, default=1,
                    help="rmat-part: 1 (default) = the columns of a rank's slab of B lie shard_rows + 64 elements apart when shard_rows is a multiple of 4 KB (columns a large power of two "
                         "apart share cache sets and memory channels: sparta_vbs_spmm_gathered_ld); 0 = the plain all-gather layout")
    ap.add_argument("--ag-chunks", type=int, default=1,
                    help="rmat-part, N > 1: all-gather + product in this many column chunks, the collective of the next chunk overlapped with the kernels of this one "
                         "(1 = north_star's single all-gather per step; the chunked step is checked bit for bit against it)")
    ap.add_argument("--reorder", choices=["off", "on", "auto"], default="off",
                    help="rmat-part: per-part blocking -- off = fixed 64-row grid (-a 2 -F 1), on = blocking_algo 7, auto = clustering kept only where its predicted "
                         "product time beats the fixed grid's")
    ap.add_argument("--rmat-density", type=float, default=0.0, help="distinct nonzeros / n^2 (0.001 = 0.1 %%); 0 = 10 edges per row, symmetrised")
    ap.add_argument("--row-slab", default=None, metavar="K/P",
                    help="rmat with --rmat-density, one GPU: only the rows [K, K+1) * 2^scale / P of the graph (P a power of two), sampled without the rest -- "
                         "one rank's share of a graph too large for one GPU (configs[4]: --rmat-scale 23 --rmat-density 1e-4 --row-slab 0/128)")
    ap.add_argument("--slab-ranks", type=int, default=8,
                    help="--row-slab: B is held the way this rank would see it after the all-gather of that many ranks (slabs of cols / ranks rows, read "
                         "through sparta_vbs_spmm_gathered)")
    ap.add_argument("--force-large", action="store_true", help="run R-MAT densities above 2.5e9 nonzeros anyway (not feasible inside one GPU lease)")
    ap.add_argument("--host-gen", action="store_true", help="generate R-MAT on the host with numpy (small cases / no GPU generator)")
    ap.add_argument("--matrix", default=None,
                    help="run the single-GPU pipeline on a matrix file instead of the synthetic workload (.mtx MatrixMarket or .el edge list, read as "
                         "documented = SPARTA_IO_STRICT): e.g. the real SuiteSparse cant.mtx where a copy is at hand (none can be fetched here)")
    ap.add_argument("--exchange", choices=["auto", "allgather", "blocks"], default="auto",
                    help="N > 1, workload cant: how the ranks' shards of B reach the slabs")
    ap.add_argument("--gather", choices=["auto", "all_gather", "peer_copies"], default="auto",
                    help="N > 1, all-gather exchange: the collective, world-1 point-to-point copies posted together, or (auto, nccl only) whichever a "
                         "plan-time measurement finds faster (logged in config.allgather)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: DEBUG ONLY -- several ranks on ONE GPU (a one-GPU box), collectives staged through host memory; checks the "
                         "multi-rank logic end to end, its timings mean nothing")
    ap.add_argument("--dist-path", action="store_true", help="run the multi-GPU code path even with one rank")
    ap.add_argument("--no-suite", action="store_true", help="N = 1, default workload: do not append the benchmark set (config.suite: synthetic families + the reference's real matrices)")
    ap.add_argument("--check-rows", type=int, default=64, help="rmat / ogbn-like: rows of C checked against a float64 evaluation before timing")
    args = ap.parse_args()
    # ---- --gpus N without a launcher: start the N ranks ourselves, as a CHILD process, before anything here touches the GPU ---------
    # (torch is not even imported yet; a process that has initialised the GPU must never be replaced by exec on these boxes)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if args.workload is None:
        args.workload = "rmat-part" if (args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1) else "cant"
    if args.rmat_scale is None:
        args.rmat_scale = 23 if args.workload == "rmat-part" else 20
    if os.environ.get("SPARTA_BENCH_WATCHDOG"):               # developer aid: dump every thread's Python stack and exit if the run takes longer than this many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SPARTA_BENCH_WATCHDOG"]), exit=True)

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

    # ---- per-workload defaults ----------------------------------------------------------------------------------------
    power_law = args.workload in ("rmat", "ogbn-like", "rmat-part") and not args.matrix
    dflt = {"cant": dict(algo=5, tau=0.6, force_fixed=1, col_block=32, dtype="f32", ncols=128),
            "rmat": dict(algo=7, tau=0.4, force_fixed=0, col_block=64, dtype="f32", ncols=128),
            "ogbn-like": dict(algo=7, tau=0.4, force_fixed=0, col_block=64, dtype="f16", ncols=256),
            "cant-weak": dict(algo=5, tau=0.6, force_fixed=1, col_block=32, dtype="f32", ncols=128),
            # configs[4]: 8.4 M x 8.4 M at 0.01 %, B = 256 columns (fp16: SURVEY.md section 8(d) C5); with --rmat-scale 20: configs[3] (B = 512, bf16)
            "rmat-part": dict(algo=7, tau=0.4, force_fixed=0, col_block=64, dtype="f16" if args.rmat_scale != 20 else "bf16",
                              ncols=256 if args.rmat_scale != 20 else 512)}[args.workload]
    for k, v in dflt.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    if args.workload == "rmat-part":
        if not args.rmat_density > 0.0:
            args.rmat_density = 1e-4 if args.rmat_scale != 20 else 1e-3
        import bench_parts
        bench_parts.run(args, torch, sa, dist, rank, local_rank, world, dev, _emit, cpu_baseline)
        if distributed:
            dist.destroy_process_group()
        return
    w, N = args.col_block, args.ncols
    h16 = args.dtype != "f32"
    esz = 2.0 if h16 else 4.0
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    sdt = {"f32": sa.F32, "f16": sa.F16, "bf16": sa.BF16}[args.dtype]
    if args.matrix and distributed:
        raise SystemExit("--matrix is a single-GPU option")
    if args.workload == "ogbn-like" and distributed:
        raise SystemExit("--workload ogbn-like is a single-GPU option")

    # ---- workload ------------------------------------------------------------------------------------------------
    t0 = time.time()
    gen_stats, n_local, shard_rows = None, None, None
    if args.matrix:
        fmt = sa._lib.FMT_MTX if args.matrix.lower().endswith(".mtx") else sa._lib.FMT_EL
        m = sa.CSR.read_from_edgelist(args.matrix, mat_fmt=fmt, mode=sa._lib.IO_STRICT)
        wl_name = "%s (%d x %d, %d nnz)" % (os.path.basename(args.matrix), m.rows, m.cols, m.nztot())
    elif args.workload == "ogbn-like":
        m, gen_stats = sa.gen.ogbn_products_like(seed=3, device=local_rank, return_stats=True)
        wl_name = "ogbn-products-like power-law graph (leading %d^2 corner of R-MAT 2^22, symmetrised, values 1: %d nnz)" % (m.rows, m.nztot())
    elif args.workload == "rmat":
        n = 1 << args.rmat_scale
        slab, slab_share = None, 1.0
        if args.row_slab:
            if distributed or not args.rmat_density > 0.0:
                raise SystemExit("--row-slab is a single-GPU option of --workload rmat --rmat-density")
            k_, p_ = (int(x) for x in args.row_slab.split("/"))
            slab = (k_, p_)
            for bit in range(p_.bit_length() - 1):               # expected share of the graph's edges: the row bits are independent, P(bit = 1) = c + d = 0.24
                slab_share *= 0.24 if (k_ >> bit) & 1 else 0.76
        if args.rmat_density > 0.0:
            target = int(args.rmat_density * float(n) * float(n) * slab_share)
            # what one GPU must hold for this density: the CSR on the device (column + 16/32-bit value per nonzero) + the generator's sort
            need = target * (4.0 + esz) + 3.0 * 8.0 * target / max(1, 1)
            if target * (4.0 + esz) > 0.8 * HBM_BYTES:
                if rank == 0:
                    _emit({"metric": "Block-sparse SpMM GFLOP/s", "value": None, "unit": "GFLOP/s", "n_gpus": n_gpus, "steps": 0, "warmup": 0,
                           "ms_per_step": None, "higher_is_better": True, "scaling": "strong" if distributed else None, "vs_baseline": None,
                           "dtype": args.dtype, "data": "synthetic",
                           "config": {"workload": "R-MAT 2^%d at density %g (%d nnz), B = %d cols, %s" % (args.rmat_scale, args.rmat_density, target, N, args.dtype),
                                      "infeasible": "the CSR image alone is %.0f GB (column + value per nonzero): does not fit the 288 GB of one MI355X; "
                                                    "not run, not shrunk" % (target * (4.0 + esz) / 1e9)},
                           "roofline": None, "cpu_baseline": None})
                if distributed:
                    dist.destroy_process_group()
                return
            del need
            if target > 2.5e9 and not args.force_large:
                # fits the HBM as a CSR image, but not the tools around it on one box: say so, do not shrink
                if rank == 0:
                    _emit({"metric": "Block-sparse SpMM GFLOP/s", "value": None, "unit": "GFLOP/s", "n_gpus": n_gpus, "steps": 0, "warmup": 0,
                           "ms_per_step": None, "higher_is_better": True, "scaling": "strong" if distributed else None, "vs_baseline": None,
                           "dtype": args.dtype, "data": "synthetic",
                           "config": {"workload": "R-MAT 2^%d at density %g (%d nnz), B = %d cols, %s" % (args.rmat_scale, args.rmat_density, target, N, args.dtype),
                                      "not_run": "the CSR image (%.0f GB) fits one MI355X, but the generator sorts %d 64-bit keys on the GPU (~%.0f GB with the sort's "
                                                 "scratch) and the host-side reorder + hybrid build of this many nonzeros take longer than a 20-minute GPU lease "
                                                 "(measured at 1.1e9 nnz: generate 4.5 s, reorder 74 s, build 28-79 s); --force-large runs it anyway"
                                                 % (target * (4.0 + esz) / 1e9, target, 3.0 * 8.0 * target / 1e9)},
                           "roofline": None, "cpu_baseline": None})
                if distributed:
                    dist.destroy_process_group()
                return
            if args.host_gen:
                raise SystemExit("--host-gen has no density mode")
            m, gen_stats = sa.gen.rmat_device(args.rmat_scale, target_nnz=target, seed=3, values="uniform", device=local_rank, return_stats=True, row_slab=slab)
            if slab is None:
                wl_name = ("R-MAT 2^%d (a,b,c = 0.57,0.19,0.19) at density %.4g %% (%d^2, %d distinct nnz)"
                           % (args.rmat_scale, 100.0 * m.nztot() / float(n) / float(n), m.rows, m.nztot()))
            else:
                wl_name = ("rows [%d, %d) of R-MAT 2^%d (a,b,c = 0.57,0.19,0.19) at graph density %.4g %%: a %d x %d slab with %d distinct nnz = %.3f of the graph's "
                           "%.3g (one rank's share of a row partition by cost; sampled without the rest of the graph)"
                           % (slab[0] * m.rows, (slab[0] + 1) * m.rows, args.rmat_scale, 100.0 * args.rmat_density, m.rows, m.cols, m.nztot(), slab_share,
                              args.rmat_density * float(n) * float(n)))
        else:
            if args.host_gen:
                m = sa.gen.rmat(args.rmat_scale, 10 << args.rmat_scale, seed=3, symmetrize=True, pattern_only=False)
            else:
                m, gen_stats = sa.gen.rmat_device(args.rmat_scale, n_edges=10 << args.rmat_scale, seed=3, symmetrize=True, values="uniform",
                                                  device=local_rank, return_stats=True)
            wl_name = "R-MAT 2^%d (a,b,c = 0.57,0.19,0.19; 10 edges per row, symmetrised: %d^2, %d nnz)" % (args.rmat_scale, m.rows, m.nztot())
    elif not distributed:
        m = sa.gen.cant_like(seed=2)
        wl_name = "cant-like FEM 9x9x257 mesh x 3 dof (62451^2, %d nnz; a synthetic STAND-IN for SuiteSparse cant, which cannot be fetched here)" % m.nztot()
    else:
        m, n_local, shard_rows = sa.gen.fem3d_slab(9, 9, 257, rank, world, dof=3, pad_to=w, seed=2)
        wl_name = None
    t_gen = time.time() - t0
    nnz_global = m.nztot()

    # ---- reorder (host) -------------------------------------------------------------------------------------------------
    t0 = time.time()
    if args.fixed_height:
        eng = sa.BlockingEngine(blocking_algo="fixed_size", row_block_size=args.fixed_height, col_block_size=w)
    else:
        eng = sa.BlockingEngine(blocking_algo=args.algo, tau=args.tau, col_block_size=w, row_block_size=args.row_block,
                                force_fixed_size=bool(args.force_fixed), sim_measure=1)
    grouping = eng.GetGrouping(m)
    t_reorder = time.time() - t0
    ff = bool(args.force_fixed) and not args.fixed_height
    rbs = args.fixed_height if args.fixed_height else args.row_block

    # ---- strong scaling over block-row ranges (power-law workloads, N > 1): this rank's slab --------------------------------
    strong = power_law and distributed
    perm_global = None
    if strong:
        perm_global = sa.get_permutation(grouping)
        part = sa.get_partition(grouping)
        rows_of = np.diff(m.rowptr)[perm_global]
        cost_row = rows_of.astype(np.float64) + 1.0                       # nonzeros (gather traffic) + the row of C
        cost_br = np.add.reduceat(cost_row, part[:-1]) if len(part) > 1 else np.zeros(0)
        ranges = sa.dist.partition_by_cost(cost_br, world)
        # every rank regenerated the matrix and reran the reorder on its own: a silent disagreement would overlap or drop rows -- one MIN / MAX
        # pair over a fingerprint (nonzeros, a hash of rowptr, of the grouping and of the ranges) before anything is built
        import zlib
        fp = np.array([float(m.nztot()), float(zlib.crc32(np.ascontiguousarray(m.rowptr).tobytes())), float(zlib.crc32(np.ascontiguousarray(grouping, np.int64).tobytes())),
                       float(zlib.crc32(np.asarray(ranges, np.int64).tobytes()))], np.float64)
        lo_, hi_ = torch.from_numpy(fp).to(dev), torch.from_numpy(fp.copy()).to(dev)
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        if not torch.equal(lo_, hi_):
            raise SystemExit("rank %d: the ranks built different matrices / groupings / ranges (fingerprints differ): not a partition of one matrix" % rank)
        b0, b1 = ranges[rank]
        my_rows = perm_global[part[b0]:part[b1]]                          # original row ids, in reordered order
        shard_rows = sa.dist.padded_shard_rows(-(-m.cols // world), w)
        m_full_cols = m.cols
        m = sa.dist.row_slab(m, my_rows, world * shard_rows)              # rows of this rank, columns padded to world * shard_rows
        gl = np.repeat(np.arange(b1 - b0, dtype=np.int64), np.diff(part[b0:b1 + 1]))
        grouping = gl
        n_local = m.rows
    nnz_local = m.nztot()

    # ---- VBS image on the device ------------------------------------------------------------------------------------------
    t0 = time.time()
    vb = None
    if power_law or (args.matrix and args.workload != "cant"):
        d = sa.DeviceVBS.from_csr(m, grouping, w, rbs, ff, device=local_rank, dtype=sdt)        # hybrid: nearly empty block-rows stay (column, value)
        eng.grouping_result, eng.col_block_size = grouping, w
        eng.CollectBlockingInfo(m)                                         # block count / stored area of the FULL VBS the reference would build
        vbs_area_ref, vbs_blocks_ref = int(eng.VBR_nzcount), int(eng.VBR_nzblocks_count)
        n_block_rows = len(np.unique(grouping)) if not args.fixed_height else -(-m.rows // args.fixed_height)
    else:
        vb = sa.VBR().fill_from_CSR_inplace(m, grouping, w, rbs, ff)
        d = vb.to_device(local_rank, dtype=sdt)
        vbs_area_ref, vbs_blocks_ref, n_block_rows = int(vb.nztot), int(len(vb.jab)), int(vb.block_rows)
    t_build = time.time() - t0
    info = d.info()
    rows_c, cols_a = info["rows"], info["cols"]

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    ldb = cols_a
    B_shard = B_gath = None
    emul = args.slab_ranks if (args.row_slab and not distributed) else 0      # a rank's slab alone: B in the gathered layout of `emul` ranks
    if emul:
        shard_rows = sa.dist.padded_shard_rows(-(-cols_a // emul), max(w, 64))
        if emul * shard_rows != cols_a:
            raise SystemExit("--row-slab: the columns (%d) must split into --slab-ranks = %d slabs of a multiple of %d rows" % (cols_a, emul, max(w, 64)))
        gd = torch.Generator(device=dev).manual_seed(1234)
        B_gath = torch.empty(emul * shard_rows * N, dtype=tdt, device=dev)
        for s_ in range(emul):                                   # slab by slab: the fp32 draw of the whole B would be 4 x its 16-bit size at once
            B_gath[s_ * shard_rows * N:(s_ + 1) * shard_rows * N] = (torch.rand(shard_rows * N, generator=gd, dtype=torch.float32, device=dev) - 0.5).to(tdt)
        B = None
    elif not distributed:
        B32 = None
        ldb = (cols_a + 7) // 8 * 8 if h16 else cols_a        # 16-bit B: leading dimension padded to a multiple of 8 elements
        if h16 and (ldb * 2) % 4096 == 0 and getattr(args, "pad_b", 1):
            ldb += 64                                           # columns a multiple of 4 KB apart share cache sets / channels (sparta_vbs_spmm_gathered_ld's comment)
        if cols_a * N <= (1 << 28):
            B32 = (torch.rand(cols_a * N, generator=g, dtype=torch.float32) - 0.5).to(dev)      # column-major, ld = cols
        else:                                                   # large B: generated on the device
            gd = torch.Generator(device=dev).manual_seed(1234 + rank)
            B32 = torch.rand(cols_a * N, generator=gd, dtype=torch.float32, device=dev) - 0.5
        if h16:
            B = torch.zeros(ldb * N, dtype=tdt, device=dev)
            B.view(N, ldb)[:, :cols_a] = B32.view(N, cols_a).to(tdt)
            del B32
        else:
            B = B32
    else:
        B_shard = (torch.rand(shard_rows * N, generator=g, dtype=torch.float32) - 0.5).to(dev).to(tdt)   # column-major, ld = shard_rows
        B_gath = torch.empty(world * shard_rows * N, dtype=tdt, device=dev)
        B = None
    C = torch.zeros(rows_c * N, dtype=torch.float32, device=dev)

    # ---- N > 1: which exchange ------------------------------------------------------------------------------------
    ex, exchange, exchange_note, B_tiles = None, None, "", None
    if distributed:
        dist.all_gather_into_tensor(B_gath, B_shard)              # setup: reference result for the self-check, B for the CPU baseline
        exchange = "allgather" if strong else args.exchange
        if exchange != "allgather":
            # the same shard in the row-block-tiled layout (block jb = one contiguous w x N column-major tile)
            B_tiles = B_shard.view(N, shard_rows // w, w).permute(1, 0, 2).contiguous().view(-1)
            ex = sa.dist.RowBlockExchange(vb, rank, world, shard_rows, N, device=local_rank, dtype=sdt)      # collective: need lists
            if exchange == "auto":
                exchange = "blocks" if ex.needed_fraction < 0.5 else "allgather"
        if exchange == "blocks":
            # self-check on the real collective: both exchanges must give the same C (two partial sums vs one: fp32 re-association).
            # The collectives stay OUTSIDE any try block: a rank that failed locally still enters every collective its peers enter.
            d.spmm_gathered(B_gath, shard_rows, C, N, accumulate=False)
            C_chk = torch.empty_like(C)
            ex.step(B_tiles, C_chk)
            torch.cuda.synchronize()
            err = float((C_chk - C).abs().max() / C.abs().max().clamp_min(1e-30))
            ok = torch.tensor([1.0 if err < 1e-4 else 0.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                print("rank %d: row-block exchange self-check FAILED (max rel err %.3e): using the all-gather" % (rank, err), file=sys.stderr)
                exchange, exchange_note = "allgather", " (row-block exchange failed its self-check)"
            del C_chk
        if exchange == "allgather" and ex is not None:
            ex.close()
            ex = None
    # all-gather exchange: the collective or the peer copies, by a plan-time measurement on the real communicator
    gather_pick, gather_mode = None, "all_gather"
    if distributed and ex is None:
        if args.gather == "auto" and args.backend == "nccl" and world > 1:
            gather_pick = sa.dist.pick_allgather(B_shard, B_gath, rank, world, reps=5, sync=torch.cuda.synchronize)     # collective
            gather_mode = gather_pick["mode"]
        elif args.gather == "peer_copies" and args.backend == "nccl" and world > 1:
            gather_mode = "peer_copies"
    dmain, vbm = (ex.d_own, ex.own) if ex is not None else (d, vb)      # the handle / matrix whose kernel dominates a step
    info = dmain.info()

    def step():
        if emul:
            d.spmm_gathered(B_gath, shard_rows, C, N, accumulate=False)       # what the rank runs behind its all-gather
        elif not distributed:
            d.spmm(B, C, N, accumulate=False, ldb=ldb)
        elif ex is not None:
            ex.step(B_tiles, C)                                    # pack + ONE all-to-all of the needed row-blocks || own product; + remote product
        else:
            if gather_mode == "peer_copies":
                sa.dist.allgather_B_peer_copies(B_shard, B_gath, rank, world)    # world - 1 copies, one per xGMI link
            else:
                dist.all_gather_into_tensor(B_gath, B_shard)      # the one exchange step (RCCL over xGMI)
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

    # ---- parity spot check for the generated power-law inputs (full-size parity lives in tests/; this guards the bench line) ----
    check = None
    if power_law and args.check_rows > 0:
        Ch = None
        rng = np.random.Generator(np.random.PCG64(11 + rank))
        perm_l = sa.get_permutation(grouping)
        pick = rng.integers(0, m.rows, size=min(args.check_rows, m.rows))
        if distributed or emul:
            Bh_cols = lambda cols_i: sa.dist.gathered_rows(B_gath, cols_i, emul or world, shard_rows, N)
        else:
            Bv = B.view(N, ldb)
            Bh_cols = lambda cols_i: Bv[:, torch.from_numpy(cols_i.astype(np.int64)).to(dev)].float().cpu().numpy().astype(np.float64)
        Cv = C.view(N, rows_c)
        worst = 0.0
        for r in pick:
            i = perm_l[r]
            cols_i = m.colidx[m.rowptr[i]:m.rowptr[i + 1]]
            if len(cols_i) == 0:
                got = Cv[:, int(r)].cpu().numpy()
                worst = max(worst, float(np.abs(got).max()))
                continue
            a = (m.vals[m.rowptr[i]:m.rowptr[i + 1]] if m.vals is not None else np.ones(len(cols_i), np.float32))
            a = torch.from_numpy(np.ascontiguousarray(a)).to(tdt).float().numpy().astype(np.float64)      # the stored (rounded) values
            bb = Bh_cols(cols_i)                                                                          # N x nnz_i
            want = bb @ a
            scale_ = np.abs(bb) @ np.abs(a) + 1e-30
            got = Cv[:, int(r)].cpu().numpy().astype(np.float64)
            worst = max(worst, float((np.abs(got - want) / scale_).max()))
        check = {"rows": int(len(pick)), "max_err_over_sum_abs": worst, "tolerance": 1e-5}
        worst_all = worst
        if distributed:                                      # every rank learns the verdict: a rank that left alone would strand its peers in the next collective
            tw = torch.tensor([worst if np.isfinite(worst) else 1e30], dtype=torch.float64, device=dev)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            worst_all = float(tw.item())
        if not (worst_all <= 1e-5):
            raise SystemExit("rank %d: parity spot check failed: %.3e of sum|a||b| on this rank, %.3e worst over the ranks (tolerance 1e-5)" % (rank, worst, worst_all))
        del Ch

    # ---- pre-roll: leave the idle clock state (untimed, counted in `config.preroll_steps`) ------------------------------
    preroll = 0
    if args.settle_ms > 0:
        t_pr = time.perf_counter()
        step()
        torch.cuda.synchronize()
        preroll = 1
        if distributed:
            # a step is a collective: every rank must run the SAME number of them -- a clock-driven loop per rank ends one step apart on two ranks
            # and the job deadlocks (one rank in the barrier, its peer in the all-gather).  The count comes from the slowest rank's first step.
            t1 = torch.tensor([time.perf_counter() - t_pr], dtype=torch.float64, device=dev)
            dist.all_reduce(t1, op=dist.ReduceOp.MAX)
            n_more = int(min(20000, max(0, np.ceil(args.settle_ms * 1e-3 / max(float(t1.item()), 1e-6)) - 1)))
            for _ in range(n_more):
                step()
            preroll += n_more
            torch.cuda.synchronize()
        else:
            batch = 8 if (time.perf_counter() - t_pr) < 2e-3 else 1
            while (time.perf_counter() - t_pr) * 1e3 < args.settle_ms and preroll < 20000:
                for _ in range(batch):
                    step()
                preroll += batch
                torch.cuda.synchronize()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    # Optional: capture G consecutive steps in a HIP graph and replay it K / G times: EXACTLY K steps in the timed region.
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
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record()                                             # HIP events on the launch stream around the same K steps
    if graph is not None:
        for _ in range(args.steps // G):
            graph.replay()
    for _ in range(0 if graph is not None else args.steps):
        step()
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t_start
    event_ms_per_step = ev0.elapsed_time(ev1) / args.steps
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3

    # ---- per-kernel device time (HIP events on the launch stream, one pair per kernel launch) ----------------------
    dmain.set_class_timing(True)
    kt, kc = {}, {}
    for _ in range(min(max(args.steps, 10), 50)):
        step()
        for k, v in dmain.class_times().items():
            kt.setdefault(k, []).append(v)
        for k, v in dmain.clock_mhz().items():    # s_memtime / s_memrealtime over the kernel: the clock the pipes really ran at
            if v > 0:
                kc.setdefault(k, []).append(v)
    dmain.set_class_timing(False)
    kernel_ms = {k: float(np.mean(v)) for k, v in kt.items()}
    kernel_mhz = {k: float(np.mean(v)) for k, v in kc.items()}
    # A step that is ONE kernel launch (aligned stream plan, one tile type, no sparse part): its duration is the event time
    # around the timed region / K -- the per-launch pairs above add ~2 us of event traffic to a 60 us kernel.
    live = [k for k, v in kernel_ms.items() if v > 0]
    ntypes = int(info["tiles64"] > 0) + int(info["tiles16"] + info["tiles32"] > 0)
    one_kernel = not distributed and len(live) == 1 and live[0] != "sparse" and (live[0] != "stream" or ntypes == 1)
    if one_kernel:
        kernel_ms[live[0]] = event_ms_per_step
    path = {1: "stream", 2: "class", 3: "generic"}.get(dmain.info()["last_path"], "?")

    nnz_total = float(nnz_global)
    if distributed and not strong:
        tt = torch.tensor([float(nnz_local)], dtype=torch.float64, device=dev)
        dist.all_reduce(tt)
        nnz_total = float(tt.item())
    useful_gflops = 2.0 * nnz_total * N / (ms_per_step * 1e-3) / 1e9
    imbalance = None
    if strong:
        tt = torch.tensor([float(sum(kernel_ms.values()))], dtype=torch.float64, device=dev)
        mx, sm = tt.clone(), tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(sm)
        imbalance = float(mx.item()) / max(float(sm.item()) / world, 1e-30)

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (rank 0's launch) ------------------------------------------------------
    sp = dmain.sparse_info()
    peak_mfma = PEAK_MFMA_H16_TFLOPS if h16 else PEAK_MFMA_F32_TFLOPS
    kernel_ms_total = sum(kernel_ms.values())
    dom = max(kernel_ms, key=lambda k: kernel_ms[k]) if kernel_ms else "stream"
    dense_area = float(info["nztot"])                              # stored area of the block-rows that are MFMA tiles on this handle
    flops_exec_dense = 2.0 * dense_area * N
    if vbm is not None:
        hh_ = np.diff(vbm.row_part)
        area = {"stream": float(vbm.nztot), "fixup": 0.0, "class16": 0.0, "class32": 0.0, "class64": 0.0}
        for hh, nb in zip(hh_, vbm.nzcount):
            r = int(hh)
            while r > 0:
                mt = min(r, 64) if r > 32 else r
                c = "class64" if r > 32 else ("class32" if r > 16 else "class16")
                area[c] += float(mt) * w * float(nb)
                r -= mt
        t_lb, flops_exec, bytes_alg = mixed_roofline_seconds(vbm.row_part, vbm.nzcount, w, N, vbm.cols, s_a=esz, s_b=esz, peak_tflops=peak_mfma)
        mean_h = float(hh_.mean())
    else:
        # hybrid handle (no host image): aggregate section-8(d) bound -- MFMA part max(bytes, flops), sparse rows as (column, value)
        # pairs + their rows of C, B ONCE
        area = {"stream": dense_area}
        dense_rows = rows_c - sp["rows"]
        bytes_dense = dense_area * esz + info["nblocks"] * 4.0 + dense_rows * N * 4.0
        bytes_sparse = float(sp["nnz"]) * (esz + 4.0) + float(sp["rows"]) * (N * 4.0 + 8.0)
        bytes_b = float(cols_a) * N * esz
        t_lb = max(bytes_dense / (PEAK_HBM_GBS * 1e9), flops_exec_dense / (peak_mfma * 1e12)) + (bytes_sparse + bytes_b) / (PEAK_HBM_GBS * 1e9)
        flops_exec, bytes_alg = flops_exec_dense, bytes_dense + bytes_sparse + bytes_b
        mean_h = float(m.rows) / max(n_block_rows, 1)
    dom_tflops = 2.0 * area.get(dom, 0.0) * N / (kernel_ms[dom] * 1e-3) / 1e12 if kernel_ms.get(dom, 0) > 0 else 0.0
    # the <= 32-row tiles of an fp32 handle run the no-barrier kernel (k_f32_direct.hip) for a column-major B unless SPARTA_F32_PLAN says otherwise
    f32_direct = (args.dtype == "f32" and not distributed and os.environ.get("SPARTA_F32_PLAN", "") != "legacy"
                  and info["tiles16"] + info["tiles32"] >= info["tiles64"])
    kname = {"stream": "vbs_spmm_f32_direct_kernel" if f32_direct else "vbs_spmm_f32_stream_kernel", "fixup": "vbs_spmm_f32_fixup_kernel", "class16": "vbs_spmm_f32_kernel<16,...>",
             "class32": "vbs_spmm_f32_kernel<32,1,4,1,1,...>", "class64": "vbs_spmm_f32_kernel<32,2,2,1,2,...>"}.get(dom, dom)
    # HBM bytes per launch of that kernel from rocprofv3 PMC passes (profiles/traffic*.json, produced by scripts/profile_bench.sh on
    # the same command; FETCH_SIZE doubled per MI355X_MICROARCH.md) -- only reported when the profiled workload is the one running now
    # (`traffic` is a committed counter measurement, not one made in this run: `traffic_source` says which file and revision it is from)
    traffic, traffic_source = None, None
    try:
        tname = "traffic.json" if args.dtype == "f32" else "traffic_%s.json" % args.dtype
        tj = json.load(open(os.path.join(ROOT, "profiles", tname)))
        wk = tj.get("workload", {})
        if (vb is not None and wk.get("vbs_area") == int(vb.nztot) and wk.get("n_cols") == N and wk.get("kernel_path") == path and not distributed
                and wk.get("dtype", "f32") == args.dtype and wk.get("kernel_rev", "") == sa.KERNEL_REV):
            traffic = round(float(tj["hbm_bytes_per_launch"]))
            traffic_source = "profiles/%s@%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, scripts/profile_bench.sh)" % (tname, wk.get("kernel_rev", ""))
    except Exception:
        traffic, traffic_source = None, None
    roofline = {
        "bound": "mfma", "achieved": round(dom_tflops, 3), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
        "frac": round(dom_tflops / PEAK_MFMA_F32_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
        "kernel": kname, "kernel_ms": round(kernel_ms.get(dom, 0.0), 5), "path": path,
        "kernels_ms": {k: round(v, 5) for k, v in kernel_ms.items()},
        "all_kernels_tflops_exec": round(flops_exec / (kernel_ms_total * 1e-3) / 1e12, 3) if kernel_ms_total > 0 else 0.0,
        "mixed_roofline_frac": round(t_lb / (kernel_ms_total * 1e-3), 4) if kernel_ms_total > 0 else 0.0,
        "algorithmic_gbs": round(bytes_alg / (kernel_ms_total * 1e-3) / 1e9, 1) if kernel_ms_total > 0 else 0.0,
        "algorithmic_bytes": round(bytes_alg),
        "kernel_ms_source": "HIP events on the launch stream around the K timed steps / K" if one_kernel else "HIP event pair per kernel launch, mean of %d steps" % min(max(args.steps, 10), 50),
    }
    if dom == "sparse":
        try:                                   # PMC-measured HBM-side bytes per step of these kernels (scripts/pmc_rmat.sh), same workload only
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_rmat.json")))
            wk = tj.get("workload", {})
            if wk.get("nnz") == int(nnz_global) and wk.get("n_cols") == N and wk.get("dtype") == args.dtype and wk.get("kernel_rev", "") == sa.KERNEL_REV:
                traffic = round(float(tj["hbm_bytes_per_step"]))
                traffic_source = "profiles/traffic_rmat.json@%s" % wk.get("kernel_rev", "")
        except Exception:
            pass
        # The matrix is carried by the sparse-row kernels: HBM-bound.  `frac` is the section-8(d) figure -- B counted ONCE (the whole
        # step: MFMA part + sparse rows + B + C, all kernels of the step including the layout transposes) -- and `gather_gbs` the
        # bandwidth the gather kernels see: one N-wide row of B per nonzero (re-reads served by L2 / Infinity Cache included).
        bytes_gather = float(sp["nnz"]) * (N * esz + 8.0) + float(sp["rows"]) * N * 4.0
        gbs_gather = bytes_gather / (kernel_ms["sparse"] * 1e-3) / 1e9
        gbs_once = bytes_alg / (kernel_ms_total * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(gbs_once, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs_once / PEAK_HBM_GBS, 4),
                    "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": "sparse_rows_kernel + sparse_segments_kernel (+ b_to_row_major / sparse_c_scatter transposes)",
                    "kernel_ms": round(kernel_ms["sparse"], 5), "path": path, "kernels_ms": {k: round(v_, 5) for k, v_ in kernel_ms.items()},
                    "algorithmic_bytes": round(bytes_alg), "frac_b_once": round(gbs_once / PEAK_HBM_GBS, 4),
                    "gather_bytes": round(bytes_gather), "gather_gbs": round(gbs_gather, 1),
                    "mixed_roofline_frac": round(t_lb / (kernel_ms_total * 1e-3), 4) if kernel_ms_total > 0 else 0.0,
                    "sparse_rows": sp["rows"], "sparse_nnz": sp["nnz"], "hub_rows": sp["hub_rows"],
                    "note": "frac = section-8(d) algorithmic bytes (A once, B ONCE, C once) / all kernels of the step; gather_gbs counts one row of B per "
                            "nonzero (a bandwidth, not a fraction: rows of B served by L2 / Infinity Cache let it exceed what HBM alone delivers)"}
    if h16 and dom != "sparse":
        # 16-bit storage: the MFMAs take 1/8 (fp16 / bf16 dense peak ~2.5 PFLOP/s) of the fp32 time while the bytes only halve:
        # the kernel is bound by memory traffic.  Algorithmic bytes: packed 16-bit A (read once) + 16-bit B (once) + fp32 C.
        bytes16 = 2.0 * float(info["nztot"]) + 2.0 * ldb * N + 4.0 * rows_c * N      # (the stored blocks, 2 bytes per element: padding and the unfetched zero halves of pair tiles are not algorithmic)
        gbs = bytes16 / (kernel_ms_total * 1e-3) / 1e9 if kernel_ms_total > 0 else 0.0
        roofline.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                         "kernel": ("vbs_spmm_h16_stream_kernel" if os.environ.get("SPARTA_H16_PATH", "a")[0] == "l" else "vbs_spmm_h16_direct_kernel"),
                         "algorithmic_bytes": round(bytes16), "algorithmic_gbs": round(gbs, 1),
                         "executed_tflops": round(dom_tflops, 3)})
        roofline.pop("mixed_roofline_frac", None)
    # `peak` is the 2.4 GHz figure of MI355X_MICROARCH.md; under this load the board does not hold 2.4 GHz (power), so the
    # measured shader clock and the fraction of the matrix peak AT THAT CLOCK are reported next to it (informational)
    if kernel_mhz.get(dom, 0) > 0:
        roofline["shader_clock_mhz"] = round(kernel_mhz[dom], 0)
    if kernel_mhz.get(dom, 0) > 0 and not h16 and dom != "sparse":
        roofline["frac_at_measured_clock"] = round(dom_tflops / (PEAK_MFMA_F32_TFLOPS * kernel_mhz[dom] / PEAK_CLOCK_MHZ), 4)

    # ---- CPU baseline: the reference's VBR::multiply on this host -- 1 thread (the reference is single-threaded) and all cores -----
    cpu = None
    if not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(sa, args, m, grouping, vb, w, rbs, ff, N, B, ldb, B_gath, emul or world, shard_rows, distributed or bool(emul), h16, torch)
        except Exception as e:  # the baseline is a report, never a reason to lose the measurement
            cpu = {"value": None, "unit": "GFLOP/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}

    if wl_name is None:
        wl_name = "row-partitioned FEM 9x9x%d mesh x 3 dof (%d^2 padded, %d nnz)" % (257 * world, world * shard_rows, int(nnz_total))
    how = "" if not distributed else (", 1 all-gather of B per step" if ex is None else ", 1 all-to-all of the needed row-blocks of B per step")
    out = {
        "metric": "Block-sparse SpMM GFLOP/s", "value": round(useful_gflops, 2), "unit": "GFLOP/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": ("strong" if strong else "weak") if distributed else None, "vs_baseline": None, "dtype": args.dtype,
        "data": "file" if args.matrix else "synthetic",
        "config": {
            "workload": "%s, B = %d cols, %s%s" % (wl_name, N, {"f32": "fp32", "f16": "fp16", "bf16": "bf16"}[args.dtype], how),
            "reorder": ("fixed height %d (reorder off: reference flags -a 2 -F 1 -B %d -b %d)" % (args.fixed_height, args.fixed_height, w)) if args.fixed_height else
                       ("Jaccard %s tau=%.2f row_block=%d force_fixed=%d (reference flags -a %d -t %.2f -B %d -F %d -b %d)"
                        % ({3: "iterative_clocked", 5: "iterative_max_size/Keeper", 7: "LSH-bucketed (extension)"}.get(args.algo, str(args.algo)), args.tau, args.row_block,
                           args.force_fixed, args.algo, args.tau, args.row_block, args.force_fixed, w)),
            "col_block_size": w, "n_cols": N, "block_rows": int(n_block_rows), "nonzero_blocks": vbs_blocks_ref,
            "vbs_area": vbs_area_ref, "fill": round(nnz_local / max(vbs_area_ref, 1), 6),
            "mean_block_row_height": round(mean_h, 2),
            "device_image": {"mfma_tile_area": int(info["nztot"]), "mfma_blocks": int(info["nblocks"]), "sparse_rows": sp["rows"], "sparse_nnz": sp["nnz"],
                             "a_bytes": int(dmain.info()["a_bytes"]), "a_bytes_at_create": int(info["a_bytes"])},      # (fp32: one image of A once the no-barrier kernel carries the products)
            "tiles": {"16": info["tiles16"], "32": info["tiles32"], "64": info["tiles64"]}, "kernel_path": path,
            "executed_gflops": round(flops_exec * (world if not strong else 1) / (ms_per_step * 1e-3) / 1e9, 1),
            "host_seconds": {"generate": round(t_gen, 2), "reorder": round(t_reorder, 2), "vbs_build": round(t_build, 2)},
            "parallelism": ("single GPU" if not distributed else
                            ("row-range partition x%d by cost (nnz + rows), identical seeded generation + reorder on every rank, B row-sharded, one all-gather "
                             "per step; kernel-time imbalance max/mean %.3f" % (world, imbalance)) if strong else
                            ("row-partition x%d, B all-gather%s" % (world, exchange_note)) if ex is None else
                            ("row-partition x%d, B row-block all-to-all (%d blocks sent / %d received by rank 0 per step = %.2f %% of the "
                             "all-gather's traffic), own-shard product overlapped" % (world, ex.n_send, ex.n_recv, 100.0 * ex.needed_fraction))),
            "launch": ("HIP graph of %d steps, replayed %d times" % (G, args.steps // G)) if graph is not None else "eager",
            "preroll_steps": preroll, "event_ms_per_step": round(event_ms_per_step, 5), "kernel_rev": sa.KERNEL_REV,
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    # ---- the benchmark SET (BASELINE.json: "... SuiteSparse set"): same run, same box, after the headline measurement ------------------------
    if not distributed and args.workload == "cant" and not args.matrix and not args.no_suite and args.dtype == "f32":
        try:
            import bench_suite
            d.close()
            del B, C
            torch.cuda.empty_cache()
            out["config"]["suite"] = bench_suite.run(sa, torch, N=128, device=local_rank)
        except Exception as e:
            out["config"]["suite"] = {"error": repr(e)[:300]}
    if distributed and ex is None:
        out["config"]["allgather"] = dict(gather_pick or {}, mode=gather_mode)
    if gen_stats is not None:
        out["config"]["generator"] = gen_stats
    if check is not None:
        out["config"]["parity_spot_check"] = check
    if distributed:
        dist.destroy_process_group()
    _emit(out)


def cpu_baseline(sa, args, m, grouping, vb, w, rbs, ff, N, B, ldb, B_gath, world, shard_rows, distributed, h16, torch):
    """Reported baseline, not the target: the reference's CPU multiply (VBR::multiply, src/general/vbr.cpp:323-372) on this box's
    host, on a bounded sample of the same workload -- a prefix of block-rows worth ~2e10 executed flops for the single thread (10-20 s)."""
    from oracle import ref, oracle as O
    cols = m.cols
    if not distributed:
        Bh = (B.view(N, ldb)[:, :cols].float().contiguous().view(-1) if h16 else B).cpu().numpy()
    else:
        Bh = sa.dist.gathered_to_colmajor(B_gath.float().cpu().numpy(), world, shard_rows, N)
    perm = sa.get_permutation(grouping)
    kind = "reference"
    if vb is None:
        # hybrid handle: no dense host image of the whole matrix.  Sample = the first block-rows (in reordered order) whose dense VBS
        # executes <= ~2e10 flops; their VBS is built by the product's own host builder (bit-identical to the reference's: tests/).
        part = sa.get_partition(grouping)
        nnz_rows = np.diff(m.rowptr)[perm]
        hts = np.diff(part)
        nnz_br = np.add.reduceat(nnz_rows.astype(np.float64), part[:-1])
        # executed flops of a block-row <= 2 * h * min(nnz, block columns) * w * N.  The sample is a seeded RANDOM set of block-rows (a
        # prefix would be the hub clusters of a power-law matrix), taken in random order while the budget lasts.
        est = 2.0 * hts * np.minimum(nnz_br, float(-(-cols // w))) * w * N
        order = np.random.Generator(np.random.PCG64(5)).permutation(len(hts))
        take, spent = [], 0.0
        for ib in order:
            if spent + est[ib] <= 2.0e10 or not take:
                take.append(int(ib))
                spent += est[ib]
            if spent >= 1.9e10 or len(take) >= 200000:
                break
        take = np.sort(np.asarray(take, np.int64))
        nbr = len(take)
        sel = np.concatenate([np.arange(part[ib], part[ib + 1]) for ib in take]) if nbr else np.zeros(0, np.int64)
        rows_s = int(len(sel))
        sub = sa.dist.row_slab(m, perm[sel], cols)
        gsub = np.repeat(np.arange(nbr, dtype=np.int64), hts[take])
        vb = sa.VBR().fill_from_CSR_inplace(sub, gsub, w, rbs, False)
        kind = "port"
        nbr_total = len(part) - 1
        nnz_s = int(nnz_rows[sel].sum())
        total_rows = m.rows
        sample_what = "a seeded random set of %d of %d block-rows" % (nbr, nbr_total)
    else:
        per_row = 2.0 * np.diff(vb.row_part) * w * vb.nzcount * N
        cum = np.cumsum(per_row)
        nbr = max(1, min(int(np.searchsorted(cum, 2.0e10, side="right")), vb.block_rows))
        rows_s = int(vb.row_part[nbr])
        nnz_s = int(np.diff(m.rowptr)[perm[:min(rows_s, m.rows)]].sum())
        nbr_total, total_rows = vb.block_rows, vb.rows
        sample_what = "block-rows [0,%d) of %d" % (nbr, nbr_total)
    exec_flops = float((2.0 * np.diff(vb.row_part)[:nbr] * w * vb.nzcount[:nbr] * N).sum())
    cpu_reps = 1
    rv = None
    if kind == "reference" and ref.available() and nbr == vb.block_rows:
        rc = ref.RefCSR(m.rows, m.cols, m.rowptr, m.colidx.astype(np.int64), m.vals)
        rv = ref.RefVBR(rc, grouping, w, rbs, ff)
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
        t1 = time.perf_counter()
        O.vbr_multiply(vb.rows, vb.cols, w, vb.row_part, vb.nzcount, vb.jab, vb.mab, Bh, N, block_row_range=(0, nbr))
        t_cpu = time.perf_counter() - t1
    # all host cores: the same loop nest, block-row ranges on separate threads (each writes its own rows of C, vbr.cpp:355); always the
    # oracle's restatement (the compiled reference has one entry point for the whole matrix)
    reps_mt, t_mt, n_thr = 0, 0.0, 1
    Cbuf = np.zeros(vb.rows * N, np.float32)
    while reps_mt < 20 and t_mt < 3.0:
        _, t_once, n_thr = O.vbr_multiply_mt(vb.rows, vb.cols, w, vb.row_part, vb.nzcount, vb.jab, vb.mab, Bh, N, block_row_range=(0, nbr), C_out=Cbuf)
        t_mt += t_once
        reps_mt += 1
    t_mt /= max(reps_mt, 1)
    csr_ref = None
    if kind == "port" and total_rows == m.rows and m.rows >= 100000:
        # large power-law inputs: the dense-block loop above only affords a sliver of the matrix, so the reference's OTHER CPU SpMM,
        # CSR::multiply (src/general/csr.cpp:49-65; oracle restatement), is timed too, on a seeded random 1 % of the rows
        pick = np.sort(np.random.Generator(np.random.PCG64(6)).choice(m.rows, size=max(1, m.rows // 100), replace=False))
        # ... cut back to ~1.5e9 multiply-adds (about 10 s of this loop) where 1 % of the rows holds more
        cum_nnz = np.cumsum(np.diff(m.rowptr)[pick].astype(np.float64)) * N
        keep = max(1, int(np.searchsorted(cum_nnz, 1.5e9, side="right")))
        pick = pick[:keep] if keep < len(pick) else pick
        sub = sa.dist.row_slab(m, pick, cols)
        t1 = time.perf_counter()
        O.csr_multiply(sub.rows, sub.rowptr, sub.colidx.astype(np.int64), sub.vals, Bh, cols, N)
        t_csr = time.perf_counter() - t1
        csr_ref = {"value": round(2.0 * sub.nztot() * N / t_csr / 1e9, 4), "unit": "GFLOP/s", "cores": 1, "kind": "port",
                   "sample": "CSR::multiply on a seeded random %.2f %% of the rows (%d rows, %d nnz), %.2f s" % (100.0 * sub.rows / m.rows, sub.rows, sub.nztot(), t_csr)}
    elif 0 <= total_rows - m.rows < 1024 and m.rows == m.cols and 2.0 * m.nztot() * N <= 4.0e10:      # (the VBS pads the rows to whole blocks: 62 451 -> 62 464)
        # the headline matrix (square, a few million nonzeros): the reference's OTHER CPU SpMM, CSR::multiply (src/general/csr.cpp:49-65), on the WHOLE matrix --
        # the compiled reference where it is present (oracle/_ref), else the oracle's restatement (BASELINE.md section 3 promises both loops)
        t_csr, csr_kind, n_rep = 0.0, "port", 0
        if ref.available():
            rc = ref.RefCSR(m.rows, m.cols, m.rowptr, m.colidx.astype(np.int64), m.vals)
            csr_kind = "reference"
            while t_csr < 3.0 and n_rep < 20:
                t1 = time.perf_counter(); rc.multiply(Bh, N); t_csr += time.perf_counter() - t1; n_rep += 1
        else:
            while t_csr < 3.0 and n_rep < 20:
                t1 = time.perf_counter(); O.csr_multiply(m.rows, m.rowptr, m.colidx.astype(np.int64), m.vals, Bh, cols, N); t_csr += time.perf_counter() - t1; n_rep += 1
        t_csr /= max(n_rep, 1)
        csr_ref = {"value": round(2.0 * m.nztot() * N / t_csr / 1e9, 4), "unit": "GFLOP/s", "cores": 1, "kind": csr_kind,
                   "sample": "CSR::multiply on the whole matrix (%d rows, %d nnz), %d repetitions, %.3f s each" % (m.rows, m.nztot(), n_rep, t_csr)}
    return {"csr_multiply": csr_ref, "value": round(2.0 * nnz_s * N / t_cpu / 1e9, 4), "unit": "GFLOP/s", "cores": 1, "kind": kind,
            "sample": "%s (%d of %d rows = %.2f %%, %d nnz), %d repetition%s, %.2f s each; executed dense-block rate %.2f GFLOP/s"
                      % (sample_what, rows_s, total_rows, 100.0 * rows_s / max(total_rows, 1), nnz_s, cpu_reps, "" if cpu_reps == 1 else "s", t_cpu,
                         exec_flops / t_cpu / 1e9),
            "all_cores": {"value": round(2.0 * nnz_s * N / t_mt / 1e9, 4), "unit": "GFLOP/s", "cores": int(n_thr), "kind": "port",
                          "sample": "same block-rows, %d threads over block-row ranges (CPUs this process may use -- affinity capped by the cgroup quota -- = %d%s), %d repetitions, %.3f s each; executed %.2f GFLOP/s"
                                    % (n_thr, O.usable_cpus(), "" if n_thr >= O.usable_cpus() else ": NOT an all-core figure -- the sample holds fewer block-rows than the host has cores",
                                       reps_mt, t_mt, exec_flops / t_mt / 1e9)}}


if __name__ == "__main__":
    main()
