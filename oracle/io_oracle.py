"""oracle/io_oracle.py -- TEST INFRASTRUCTURE ONLY (never imported by the product package `sparta_amd`).

Plain-Python restatement of the reference's on-disk formats, line by line, for SMALL files:

  read_el / read_mtx       src/general/csr.cpp:196-307 / :309-365     (CSR::read_from_edgelist_el / _mtx)
  save_to_edgelist         src/general/csr.cpp:169-179
  read_grouping_file       test/general/Matrix_Analysis.cpp:10-32 (+ the leading-count rule :77-78)
  csv_row                  src/general/utilities.cpp:175-236          (save_blocking_data: header + value line)
  degree_permutation       src/general/csr.cpp:123-155                (through the oracle's std::sort restatement)

Pinned by tests/test_io.py against the compiled reference (oracle/_ref, where it exists) and against the fixtures under
tests/golden/ (the reference's own data/TEST_matrix_weighted.el and data/TEST/TEST.g, and tests/golden/io.npz).
Where the reference throws an uncaught exception or runs into undefined behaviour this raises RefUndefined.
"""
import re

import numpy as np

INT_MIN, INT_MAX = -2 ** 31, 2 ** 31 - 1
NPOS = -1


class RefUndefined(Exception):
    """the reference aborts (uncaught exception) or its behaviour is undefined on this input"""


_INT = re.compile(r"[ \t\n\v\f\r]*[+-]?[0-9]+")
_FLT = re.compile(r"[ \t\n\v\f\r]*[+-]?((([0-9]+\.?[0-9]*|\.[0-9]+)([eE][+-]?[0-9]+)?)|inf(inity)?|nan)", re.I)


def stoi(s):
    """std::stoi: strtol prefix parse, std::invalid_argument without digits, std::out_of_range outside int"""
    m = _INT.match(s)
    if not m:
        raise RefUndefined("stoi('%s'): invalid_argument" % s)
    v = int(m.group(0))
    if not INT_MIN <= v <= INT_MAX:
        raise RefUndefined("stoi: out_of_range")
    return v


def stof(s):
    """std::stof: strtof prefix parse (decimal forms; hex floats are not restated), float32 result"""
    m = _FLT.match(s)
    if not m:
        raise RefUndefined("stof('%s'): invalid_argument" % s)
    with np.errstate(over="ignore"):
        v = np.float32(float(m.group(0)))
    t = m.group(0).strip().lower().lstrip("+-")
    if np.isinf(v) and not t.startswith("inf"):
        raise RefUndefined("stof: out_of_range")
    return v


def _find(s, d):
    return s.find(d)                                      # -1 plays std::string::npos


def _substr0(s, n):
    return s if n == NPOS else s[:n]                      # s.substr(0, npos) is the whole string


def _erase0(s, pos, dlen):
    n = pos + dlen                                        # `int del_pos` = -1 for npos: erases dlen - 1 characters (csr.cpp:226)
    return s[max(n, 0):]


def _lines_after_leading_comments(text):
    """csr.cpp:211: `while (peek == '#' or '%') ignore(2048, '\\n')` then getline-by-getline"""
    pos = 0
    while pos < len(text) and text[pos] in "#%":
        nl = text.find("\n", pos)
        if nl == -1 or nl - pos >= 2048:
            raise RefUndefined("comment line without newline / longer than 2048 characters")
        pos = nl + 1
    rest = text[pos:]
    if rest == "":
        return []
    lines = rest.split("\n")
    if lines[-1] == "":
        lines.pop()                                       # getline does not produce an empty last line after a final '\n'
    return lines


def read_el(text, delimiter=" ", pattern_only=False, symmetrize=False):
    """-> rows, cols, rowptr, colidx, vals (None when pattern_only)"""
    lines = _lines_after_leading_comments(text)
    lines = lines[1:]                                     # csr.cpp:213: the first line is read and never used
    pos, val = [], []
    i, max_col, triangular = -1, 0, True
    dl = len(delimiter)
    for temp in lines:
        dp = _find(temp, delimiter)
        a = stoi(_substr0(temp, dp))
        temp = _erase0(temp, dp, dl)
        dp = _find(temp, delimiter)
        b = stoi(_substr0(temp, dp))
        v = np.float32(1.0)
        if not pattern_only:
            temp = _erase0(temp, dp, dl)
            dp = _find(temp, delimiter)
            v = stof(_substr0(temp, dp))
        if a < 0 or b < 0:
            raise RefUndefined("negative index: out-of-bounds vector access in the reference")
        if b < a:
            triangular = False
        max_col = max(max_col, b)
        if a > i:
            while i < a:
                pos.append([])
                val.append([])
                i += 1
        elif a < i:
            raise RefUndefined("std::invalid_argument: indices must be in ascending order")
        pos[i].append(b)
        val[i].append(v)
    if symmetrize and triangular:
        import bisect
        for ii in range(len(pos)):
            nz = 0
            while nz < len(pos[ii]):
                j = pos[ii][nz]
                if j >= len(pos):
                    raise RefUndefined("symmetrize: pos_holder[j] out of range")
                k = bisect.bisect_left(pos[j], ii)
                if k == len(pos[j]) or pos[j][k] != ii:
                    if not pattern_only:
                        raise RefUndefined("std::invalid_argument: symmetrize only implemented for unweighted graphs")
                    pos[j].insert(k, ii)
                nz += 1
    rows = len(pos)
    rowptr = np.zeros(rows + 1, np.int64)
    for r in range(rows):
        rowptr[r + 1] = rowptr[r] + len(pos[r])
    colidx = np.array([c for r in pos for c in r], np.int64)
    vals = None if pattern_only else np.array([x for r in val for x in r], np.float32)
    return rows, max_col + 1, rowptr, colidx, vals


def read_mtx(text):
    """csr.cpp:309-365: always pattern-only; one line after the size line is skipped; exactly nnz lines are read"""
    lines = _lines_after_leading_comments(text)
    if not lines:
        raise RefUndefined("no size line")
    head = lines[0].split()
    try:
        rows, cols, nnz = int(head[0]), int(head[1]), int(head[2])
    except (ValueError, IndexError):
        raise RefUndefined("bad size line")
    data = lines[2:]                                      # infile.ignore(2048, '\n') skips the line after the size line
    if len(data) < nnz:
        raise RefUndefined("fewer than nnz lines after the skipped one: the reference indexes with an unread value")
    pos = [[] for _ in range(rows)]
    for k in range(nnz):
        f = data[k].split()
        try:
            i, j = int(f[0]) - 1, int(f[1]) - 1
        except (ValueError, IndexError):
            raise RefUndefined("bad entry line")
        if not (0 <= i < rows) or j < 0:
            raise RefUndefined("index out of range")
        pos[i].append(j)
    rowptr = np.zeros(rows + 1, np.int64)
    for r in range(rows):
        rowptr[r + 1] = rowptr[r] + len(pos[r])
    return rows, cols, rowptr, np.array([c for r in pos for c in r], np.int64), None


def save_to_edgelist(rows, rowptr, colidx, delimiter=" ", mtx=False):
    out = []
    for i in range(rows):
        for k in range(int(rowptr[i]), int(rowptr[i + 1])):
            out.append("%d%s%d\n" % ((colidx[k], delimiter, i) if mtx else (i, delimiter, colidx[k])))
    return "".join(out)


def read_grouping_file(text, rows=None):
    g = []
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    for line in lines:
        try:
            g.append(stoi(line))
        except RefUndefined:
            pass                                          # Matrix_Analysis.cpp:23-27: message on stderr, line skipped
    if rows is not None and len(g) == rows + 1:
        g = g[1:]                                         # :78
    return np.array(g, np.int64)


def grouping_file(grouping):
    return "".join("%d\n" % int(x) for x in grouping)     # utilities.cpp:239-243


CSV_COLUMNS = ("matrix", "rows", "cols", "nonzeros", "symmetrize", "blocking_algo", "tau", "row_block_size", "col_block_size",
               "use_pattern", "sim_use_groups", "sim_measure", "reorder", "exp_name", "b_cols", "warmup", "exp_repetitions",
               "multiplication_algo", "n_streams", "time_to_block", "time_to_merge", "time_to_compare", "VBR_nzcount",
               "VBR_nzblocks_count", "VBR_average_height", "VBR_longest_row", "merge_counter", "comparison_counter",
               "average_merge_tau", "average_row_distance", "avg_time_multiply", "std_time_multiply")
_FLOATS = {"tau", "time_to_block", "time_to_merge", "time_to_compare", "VBR_average_height", "average_merge_tau",
           "average_row_distance", "avg_time_multiply", "std_time_multiply"}
_STRINGS = {"matrix", "exp_name"}


def csv_row(**fields):
    """utilities.cpp:178-236: `name,` / `value,` per column; std::to_string(float) is "%f" of the float32 value"""
    header, values = "", ""
    for k in CSV_COLUMNS:
        v = fields.get(k, "" if k in _STRINGS else 0)
        header += k + ","
        if k in _STRINGS:
            values += str(v) + ","
        elif k in _FLOATS:
            values += "%f," % float(np.float32(v))
        else:
            values += "%d," % int(v)
    return header, values


def degree_permutation(rowptr, descending, get_permutation):
    """csr.cpp:123-155 through `get_permutation` = the oracle's restatement of std::sort with `key[i] < key[j]`.
    Descending uses `n[i] >= n[j]`, which equals `-n[i] < -n[j]` only without ties; with ties the reference's sort is only
    defined for <= 16 rows (pure insertion sort), restated here literally."""
    deg = np.diff(np.asarray(rowptr, np.int64))
    if not descending:
        return get_permutation(deg)
    if len(np.unique(deg)) == len(deg):
        return get_permutation(-deg)
    if len(deg) > 16:
        raise RefUndefined("std::sort with a non-strict comparator on ties beyond the insertion-sort threshold")
    v = list(range(len(deg)))
    comp = lambda i, j: deg[i] >= deg[j]                  # noqa: E731
    for i in range(1, len(v)):                            # std::__insertion_sort (bits/stl_algo.h)
        x = v[i]
        if comp(x, v[0]):
            v[1:i + 1] = v[0:i]
            v[0] = x
        else:
            k = i
            while comp(x, v[k - 1]):
                v[k] = v[k - 1]
                k -= 1
            v[k] = x
    return np.array(v, np.int64)


def blocked_ell(rows, cols, bs, nzcount, jab, mab):
    """prepare_cusparse_BLOCKEDELLPACK (src/cuda/cuda_utilities.cpp:1656-1710), loop for loop.  The reference file is CUDA (cuSPARSE
    headers) and cannot be compiled here: PARITY UNPINNED for this function beyond this restatement and the property that the
    Blocked-ELL arrays expand to the same dense matrix as the VBS (tests/test_io.py)."""
    if rows % bs != 0 or cols % bs != 0:
        raise RefUndefined("exit(__LINE__): rows / cols not a multiple of ell_blocksize (:1666-1672)")
    ind_rows = rows // bs                                               # :1674
    ind_cols = 0
    for i in range(ind_rows):                                           # :1675-1679
        if nzcount[i] > ind_cols:
            ind_cols = int(nzcount[i])
    val_cols = ind_cols * bs                                            # :1680
    ind = np.zeros((ind_rows, ind_cols), np.int64)
    val = np.zeros((rows, val_cols), np.float32)
    k_col = 0
    for i in range(ind_rows):                                           # :1688-1695
        for j in range(ind_cols):
            if j < nzcount[i]:
                ind[i, j] = jab[k_col]
                k_col += 1
            else:
                ind[i, j] = -1
    vbr_shift = bel_shift = 0
    flat = val.reshape(-1)
    for k in range(ind_rows):                                           # :1698-1705
        for i in range(bs):
            for j in range(val_cols):
                flat[bel_shift + i * val_cols + j] = mab[vbr_shift + j * bs + i] if ind[k, j // bs] != -1 else 0.0
        vbr_shift += int(nzcount[k]) * bs * bs
        bel_shift += ind_cols * bs * bs
    return bs, ind, val
