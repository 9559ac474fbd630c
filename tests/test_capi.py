"""CPU: the C-ABI shared library loads, exports every symbol include/sparta_amd.h declares, reports errors through
status codes + sparta_last_error, and FAILS LOUDLY (no CPU fallback) when asked to multiply without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import sparta_amd as sa
from sparta_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "sparta_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sparta_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported():
    names = header_functions()
    assert len(names) >= 20
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libsparta_amd.so does not export %s" % n
    assert sorted(_lib.SYMBOLS) == names, "sparta_amd/_lib.py SYMBOLS out of sync with include/sparta_amd.h"


def test_defaults_follow_reference_cli():
    c = _lib.ReorderCfg()
    _lib.lib.sparta_reorder_cfg_default(C.byref(c))
    # include/input.h:15-42
    assert (c.blocking_algo, c.sim_measure, c.use_groups, c.col_block_size, c.row_block_size, c.use_pattern, c.force_fixed_size) == \
        (3, 1, 0, 3, 3, 1, 0)
    assert c.tau == pytest.approx(0.1)


def test_error_reporting():
    # unsorted rows are LEGAL input (the reference never sorts: two of its own data files have them) ...
    m = sa.CSR(3, 3, [0, 2, 2, 3], [2, 1, 0])
    assert sa.BlockingEngine(tau=0.5, col_block_size=2).GetGrouping(m).shape == (3,)
    # ... a broken rowptr is not
    with pytest.raises(sa.SpartaError) as ei:
        sa.BlockingEngine(tau=0.5, col_block_size=2).GetGrouping(sa.CSR(3, 3, [0, 2, 1, 3], [2, 1, 0]))
    assert ei.value.code == _lib.ERR_INVALID and "rowptr" in str(ei.value)
    ok = sa.CSR(3, 3, [0, 2, 2, 3], [1, 2, 0])
    with pytest.raises(sa.SpartaError) as ei:
        sa.BlockingEngine(tau=0.5, col_block_size=2, blocking_algo=1, structured_n=0).GetGrouping(ok)   # m:n structured needs m, n > 0
    assert ei.value.code == _lib.ERR_INVALID
    with pytest.raises(sa.SpartaError) as ei:
        sa.BlockingEngine(tau=0.5, col_block_size=2, blocking_algo=9).GetGrouping(ok)
    assert ei.value.code == _lib.ERR_INVALID
    with pytest.raises(sa.SpartaError):
        sa.BlockingEngine(tau=0.5, col_block_size=0).GetGrouping(ok)
    with pytest.raises(sa.SpartaError):
        sa.CSR(3, 3, [0, 2, 2, 3], [1, 5, 0]) and sa.BlockingEngine(col_block_size=2).GetGrouping(sa.CSR(3, 3, [0, 2, 2, 3], [1, 5, 0]))
    with pytest.raises(ValueError):
        sa.VBR().fill_from_CSR_inplace(ok, [0, 0], 2)
    # null handle / null out pointers come back as status codes, not crashes
    assert _lib.lib.sparta_vbs_info(None, None) == _lib.ERR_INVALID
    assert _lib.lib.sparta_vbs_spmm(None, None, 0, 0, 1, None, 0, 0, 0, 0, None, 0, None) == _lib.ERR_INVALID
    assert b"NULL" in _lib.lib.sparta_last_error()
    assert _lib.lib.sparta_vbs_sparse_info(None, None) == _lib.ERR_INVALID
    # the exchange pack kernel validates before it touches the GPU: nothing to copy is fine anywhere, bad sizes are refused
    assert _lib.lib.sparta_pack_blocks(None, 256, None, 0, None, None) == _lib.OK
    assert _lib.lib.sparta_pack_blocks(None, 100, None, 3, None, None) == _lib.ERR_INVALID
    assert _lib.lib.sparta_pack_blocks(None, 256, None, 3, None, None) == _lib.ERR_INVALID
    # a device handle straight from the CSR: argument errors first, then "no device" here (a GPU box builds it: tests/test_spmm_gpu.py)
    import ctypes as C
    h = C.c_void_p(None)
    assert _lib.lib.sparta_vbs_create_from_csr(None, 3, 3, None, None, None, None, 2, 0, 0, _lib.F32, 0) == _lib.ERR_INVALID
    assert _lib.lib.sparta_vbs_create_from_csr(C.byref(h), 3, 3, None, None, None, None, 2, 0, 0, 77, 0) == _lib.ERR_INVALID
    if sa.device_count() == 0:
        with pytest.raises(sa.SpartaError) as ei:
            sa.DeviceVBS.from_csr(ok, np.array([0, 0, 2]), 2)
        assert ei.value.code == _lib.ERR_NO_DEVICE


def test_empty_rows_and_padding_rules():
    # empty rows cluster together into a block-row with zero blocks; force_fixed pads rows/cols up
    m = sa.CSR(5, 7, [0, 0, 2, 2, 3, 3], [0, 6, 3], [1.0, 2.0, 3.0])
    g = sa.BlockingEngine(tau=0.3, col_block_size=2).GetGrouping(m)
    assert g.tolist() == [0, 1, 0, 3, 0]
    v = sa.VBR().fill_from_CSR_inplace(m, g, 2)
    assert v.row_part.tolist() == [0, 3, 4, 5] and v.nzcount.tolist() == [0, 2, 1] and v.jab.tolist() == [0, 3, 1]
    assert v.block_cols == 4 and v.nztot == 6            # last block column is zero-padded in mab (cols % w != 0)
    vf = sa.VBR().fill_from_CSR_inplace(m, np.arange(5) // 2, 2, 2, True)
    assert (vf.rows, vf.cols) == (6, 8) and vf.row_part.tolist() == [0, 2, 4, 6]


@pytest.mark.skipif(sa.device_count() > 0, reason="this check is for a box WITHOUT a GPU")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    m = sa.gen.uniform_random(64, 64, 300, seed=1)
    v = sa.VBR().fill_from_CSR_inplace_fixed(m, 16, 16)
    with pytest.raises(sa.SpartaError) as ei:
        v.to_device(0)
    assert ei.value.code == _lib.ERR_NO_DEVICE and "no CPU fallback" in str(ei.value)
    with pytest.raises(sa.SpartaError):
        v.multiply(np.zeros(64 * 4, np.float32), 4, np.zeros(64 * 4, np.float32))


def test_product_does_not_import_the_oracle():
    """the oracle is test infrastructure: nothing under sparta_amd/ may import, link or load it"""
    pkg = os.path.join(ROOT, "sparta_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in txt.lower().replace("oracle/", "oracle/") or f in ("__init__.py",) and "oracle" not in txt, \
                    "%s mentions the oracle" % os.path.join(dirpath, f)
    out = os.popen("ldd %s" % _lib.LIB_PATH).read()
    assert "oracle" not in out and "sparta_ref" not in out
