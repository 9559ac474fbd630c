"""The reference's own real-world inputs (data/minitest/*.el -- bcsstk18 is SuiteSparse HB/bcsstk18 -- committed as DATA under
tests/golden/ref_data/minitest/) through the whole path, against golden vectors the COMPILED reference produced
(tests/golden/make_golden_real.py -> real.npz): reader, reorder (-a 3 and the experiments' -a 5 -F 1), VBS build, and the product.

CPU (-m "not gpu"): product host code and the oracle against the golden vectors.
GPU (-m gpu): SPARTA_SPMM_EXACT bit-identical to the reference's C; MFMA kernels (fp32, fp16, bf16) within 1e-5 * sum|a||b| of the oracle."""
import hashlib
import os

import numpy as np
import pytest

import _util as U

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "ref_data", "minitest")
N_GOLD = 8


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _cases():
    g = U.load("real.npz")
    return [(str(k), str(f), str(mode), str(b)) for k, f, mode, b in g["index"]]


CFG = {"a3": dict(algo=3, tau=0.5, w=64, rbs=1, ff=False), "a5F1": dict(algo=5, tau=0.6, w=32, rbs=32, ff=True)}
_cache = {}


def _read(f, mode):
    import sparta_amd as sa
    if (f, mode) not in _cache:
        _cache[(f, mode)] = sa.CSR.read_from_edgelist(os.path.join(DATA, f), pattern_only=(mode == "pattern"))      # IO_COMPAT: the reference's reader
    return _cache[(f, mode)]


def _built(key, f, mode, b):
    """reader -> reorder -> VBS by the product's host code, each stage checked against the golden vectors"""
    import sparta_amd as sa
    gold = U.load("real.npz")
    cfg = CFG[b]
    m = _read(f, mode)
    k0 = "%s/%s" % (f, mode)
    assert [m.rows, m.cols, m.nztot()] == gold[k0 + "/dims"].tolist()
    assert _sha(m.rowptr) + _sha(m.colidx) + ("" if mode == "pattern" else _sha(m.vals)) == str(gold[k0 + "/csr_sha"])
    e = sa.BlockingEngine(blocking_algo=cfg["algo"], tau=cfg["tau"], col_block_size=cfg["w"], row_block_size=cfg["rbs"], force_fixed_size=cfg["ff"])
    g = e.GetGrouping(m)
    assert np.array_equal(g, gold[key + "/grouping"]), "grouping differs from the compiled reference"
    assert [e.comparison_counter, e.merge_counter] == gold[key + "/counters"].tolist()
    v = sa.VBR().fill_from_CSR_inplace(m, g, cfg["w"], cfg["rbs"], cfg["ff"])
    assert [v.rows, v.cols, v.block_rows, v.block_cols, v.nztot] == gold[key + "/dims"].tolist()
    assert np.array_equal(v.row_part, gold[key + "/row_part"]) and np.array_equal(v.nzcount, gold[key + "/nzcount"])
    assert np.array_equal(v.jab, gold[key + "/jab"]) and _sha(v.mab) == str(gold[key + "/mab_sha"])
    return m, g, v, gold


@pytest.mark.parametrize("key,f,mode,b", _cases())
def test_host_path_matches_the_compiled_reference(key, f, mode, b):
    import sparta_amd as sa
    from oracle import oracle as O
    m, g, v, gold = _built(key, f, mode, b)
    # the oracle's multiply reproduces the reference's C on these inputs (pins the checker the GPU tests use)
    B = sa.gen.dense_rhs(v.cols, N_GOLD, seed=77)
    C = O.vbr_multiply(v.rows, v.cols, v.block_col_size, v.row_part, v.nzcount, v.jab, v.mab, B, N_GOLD)
    assert _sha(C) == str(gold[key + "/C_sha"]) and np.array_equal(C[:4096], gold[key + "/C_head"])
    # CollectBlockingInfo
    e = sa.BlockingEngine(col_block_size=CFG[b]["w"])
    e.grouping_result = g
    e.CollectBlockingInfo(m)
    assert [int(e.VBR_nzcount), int(e.VBR_nzblocks_count), int(e.VBR_longest_row)] == gold[key + "/info"].tolist()


@pytest.mark.parametrize("f", ["bcsstk18_r.el", "wiki-Vote_r.el"])
def test_oracle_reorder_on_real_matrices(f):
    """the oracle's restatement of the two clustering algorithms on real inputs (the two smallest: it is the plain quadratic scan)"""
    from oracle import oracle as O
    gold = U.load("real.npz")
    m = _read(f, "pattern")
    for b, cfg in CFG.items():
        go, co = O.get_grouping(m.rows, m.rowptr, m.colidx, cfg["algo"], 1, cfg["tau"], cfg["w"], cfg["rbs"], False, True, cfg["ff"])
        key = "%s/pattern/%s" % (f, b)
        assert np.array_equal(go, gold[key + "/grouping"])
        assert [co["comparison_counter"], co["merge_counter"]] == gold[key + "/counters"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("key,f,mode,b", _cases())
def test_gpu_product_on_real_matrices(key, f, mode, b):
    import torch
    import sparta_amd as sa
    from oracle import oracle as O
    m, g, v, gold = _built(key, f, mode, b)
    w = v.block_col_size
    # (1) the exact-order kernel: bit-identical to the compiled reference's C
    B = sa.gen.dense_rhs(v.cols, N_GOLD, seed=77)
    d = v.to_device(0)
    Bd = torch.from_numpy(B).cuda()
    Cd = torch.zeros(v.rows * N_GOLD, dtype=torch.float32, device="cuda")
    d.spmm(Bd, Cd, N_GOLD, accumulate=False, algo=sa.SPMM_EXACT)
    torch.cuda.synchronize()
    C = Cd.cpu().numpy()
    assert _sha(C) == str(gold[key + "/C_sha"]), "SPARTA_SPMM_EXACT is not bit-identical to the reference's VBR::multiply"
    # (2) MFMA kernels at N = 8 (generic shapes) and N = 128 (the product kernels), fp32, within 1e-5 * sum|a||b| of the oracle
    for N in (N_GOLD, 128):
        Bn = sa.gen.dense_rhs(v.cols, N, seed=78)
        want = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, Bn, N)
        bound = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, np.abs(v.mab), np.abs(Bn), N)
        Bd = torch.from_numpy(Bn).cuda()
        Cd = torch.full((v.rows * N,), 7.0, dtype=torch.float32, device="cuda")
        d.spmm(Bd, Cd, N, accumulate=False)
        torch.cuda.synchronize()
        assert np.all(np.abs(Cd.cpu().numpy() - want) <= 1e-5 * bound + 1e-30), (N, "fp32 MFMA")
    d.close()
    # (3) fp16 / bf16 storage, N = 128: the oracle on the ROUNDED inputs is the reference (products of two 16-bit values are exact in fp32)
    N = 128
    Bn = sa.gen.dense_rhs(v.cols, N, seed=79)
    for sdt, tdt in ((sa.F16, torch.float16), (sa.BF16, torch.bfloat16)):
        if sdt == sa.F16 and float(np.abs(v.mab).max()) > 6.0e4:
            continue                                                     # bcsstk18's stiffness values (up to 8e6) are outside fp16's range: bf16 only
        mab_r = torch.from_numpy(v.mab).to(tdt).float().numpy()
        Bt = torch.from_numpy(Bn).to(tdt)
        B_r = Bt.float().numpy()
        ldb = (v.cols + 7) // 8 * 8
        Bdev = torch.zeros(ldb * N, dtype=tdt, device="cuda")
        Bdev.view(N, ldb)[:, :v.cols] = Bt.view(N, v.cols).cuda()
        want = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, mab_r, B_r, N)
        bound = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, np.abs(mab_r), np.abs(B_r), N)
        d16 = v.to_device(0, dtype=sdt)
        Cd = torch.full((v.rows * N,), 7.0, dtype=torch.float32, device="cuda")
        d16.spmm(Bdev, Cd, N, accumulate=False, ldb=ldb)
        torch.cuda.synchronize()
        assert np.all(np.abs(Cd.cpu().numpy() - want) <= 1e-5 * bound + 1e-30), (sdt, "16-bit storage")
        d16.close()
    # (4) rows in ascending column order: the handle made straight from the CSR (hybrid: nearly empty block-rows on the sparse-row
    # kernels) gives the same product as the dense-block handle
    dcol = np.diff(m.colidx.astype(np.int64))
    inner = np.ones(len(dcol), bool)
    ends = m.rowptr[1:-1]
    inner[ends[(ends > 0) & (ends < m.nztot())] - 1] = False               # differences across a row boundary do not count
    srt = bool(np.all(dcol[inner] > 0))
    if srt:
        cfg = CFG[b]
        Bn = sa.gen.dense_rhs(v.cols, N, seed=80)
        want = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, v.mab, Bn, N)
        bound = O.vbr_multiply(v.rows, v.cols, w, v.row_part, v.nzcount, v.jab, np.abs(v.mab), np.abs(Bn), N)
        dh = sa.DeviceVBS.from_csr(m, g, w, cfg["rbs"], cfg["ff"], device=0)
        Cd = torch.full((dh.rows * N,), 7.0, dtype=torch.float32, device="cuda")
        dh.spmm(torch.from_numpy(Bn).cuda(), Cd, N, accumulate=False)
        torch.cuda.synchronize()
        assert dh.rows == v.rows
        assert np.all(np.abs(Cd.cpu().numpy() - want) <= 1e-5 * bound + 1e-30), "hybrid handle"
        dh.close()
