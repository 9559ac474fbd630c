"""The reference-shaped C++ API (include/sparta_compat.hpp): a driver written like the reference's own drivers is
compiled against it with g++, linked to libsparta_amd.so, and must print the README example's numbers."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "compat_driver")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "compat_driver.cpp")
    libdir = os.path.join(ROOT, "sparta_amd")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                               "-L", libdir, "-lsparta_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def _run(mode, *args):
    out = subprocess.check_output([_build(), mode] + list(args), text=True)
    return dict(line.split(":", 1) for line in out.strip().splitlines())


def test_compat_driver_host_side():
    r = _run("host")
    assert r["grouping"].split() == "0 1 1 1 0 5 0 0 8".split()
    assert r["counters"].split() == ["13", "5"]
    assert r["dims"].split() == "9 9 4 3 33".split()
    assert r["jab"].split() == "0 1 2 2 0".split()
    assert r["info"].split() == ["33", "5", "3"]
    # rows 0, 4, 6, 7 are empty; rows 1 and 3 touch the same blocks {0, 1, 2} of 3 columns
    assert r["minhash"].split() == "0 1 2 1 0 5 0 0 8".split()


@pytest.mark.gpu
def test_compat_driver_multiplies_on_gpu():
    r = _run("gpu")
    want = [0, 0, 0, 0, 126, 22, 102, 14, 10, 0, 0, 0, 0, 306, 49, 219, 32, 55]
    assert [float(x) for x in r["C"].split()] == want
    assert [float(x) for x in r["C2"].split()] == [2 * x for x in want]
    # B * A: rows of A in the VBS's order (get_permutation of the printed grouping), small integers -> exact
    import numpy as np
    import sys
    sys.path.insert(0, ROOT)
    import sparta_amd as sa
    rowptr = [0, 0, 3, 6, 10, 10, 11, 11, 11, 12]
    colidx = [2, 5, 8, 5, 6, 8, 1, 3, 7, 8, 6, 1]
    vals = [5, 8, 7, 1, 1, 1, 1, 1, 3, 8, 2, 5]
    A = np.zeros((9, 9))
    for i in range(9):
        for k in range(rowptr[i], rowptr[i + 1]):
            A[i, colidx[k]] = vals[k]
    perm = sa.get_permutation(np.array([int(x) for x in r["grouping"].split()]))
    E = (np.arange(18) % 5 - 2.0).reshape(9, 2).T                    # column-major 2 x 9
    want_ba = (E @ A[perm]).T.reshape(-1)                            # column-major 2 x 9
    assert [float(x) for x in r["BA"].split()] == want_ba.tolist()


def test_compat_driver_reads_blocks_and_writes_the_csv_row_like_the_reference():
    """CSR(ifstream, ...) + save_blocking_data + reorder_by_degree, against the compiled reference's output (tests/golden/io.npz)"""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "io.npz"))
    r = _run("io", os.path.join(ROOT, "tests", "golden", "ref_data", "TEST_matrix_weighted.el"))
    assert r["read"].split() == ["9", "9", "12"]
    assert r["csv"] == str(z["csv/0/csv"]).replace("\n", "|")
    assert r["gfile"].split() == str(z["csv/0/gfile"]).split()
    assert [int(x) for x in r["degrees"].split()] == sorted([0, 3, 3, 4, 0, 1, 0, 0, 1], reverse=True)
    assert r["nnz_after"].split() == ["12"]
