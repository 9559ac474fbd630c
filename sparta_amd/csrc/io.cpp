// io.cpp -- the on-disk formats either side of the hot path: edge-list / MatrixMarket readers, the grouping file, the
// reference's 32-column CSV statistics row, and the row permutations its drivers apply before blocking.
//
// Reference: src/general/csr.cpp:67-166 (permute_rows / reorder / reorder_by_degree), :169-179 (save_to_edgelist),
// :183-365 (readers), src/general/utilities.cpp:175-245 (save_blocking_data: CSV row + grouping file),
// test/general/Matrix_Analysis.cpp:10-32,77-78 (grouping-file reader).
//
// Two reader modes.  SPARTA_IO_COMPAT reproduces what the reference's readers DO, quirks included, wherever that is
// defined behaviour; where the reference runs into undefined behaviour or an uncaught exception (malformed line, too few
// lines, out-of-range index) this returns SPARTA_ERR_IO with a message instead.  SPARTA_IO_STRICT reads the formats as
// documented (no dropped line; MatrixMarket banner, values and symmetry honoured).
#include "host_core.hpp"
#include "sparta_amd.h"

#include <algorithm>
#include <cerrno>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

namespace {

using sparta::fail;

// std::stoi as the reference uses it (csr.cpp:224,229): leading whitespace, optional sign, decimal digits; anything
// after the digits is ignored.  false: no digits (std::invalid_argument in the reference) or out of int range.
bool stoi_like(const std::string& s, long* out) {
    const char* p = s.c_str();
    char* end = nullptr;
    errno = 0;
    const long v = std::strtol(p, &end, 10);
    if (end == p) return false;
    if (errno == ERANGE || v < INT_MIN || v > INT_MAX) return false;
    *out = v;
    return true;
}

// std::stof (csr.cpp:241): strtof semantics (decimal, hex, inf/nan accepted), out-of-range is an error
bool stof_like(const std::string& s, float* out) {
    const char* p = s.c_str();
    char* end = nullptr;
    errno = 0;
    const float v = std::strtof(p, &end);
    if (end == p) return false;
    if (errno == ERANGE) return false;                     // std::stof throws std::out_of_range
    *out = v;
    return true;
}

void skip_leading_comments(std::istream& in) {             // csr.cpp:211,312: only at the very top of the file
    while (in.peek() == '#' || in.peek() == '%') in.ignore(2048, '\n');
}

struct Rows {
    std::vector<std::vector<int64_t>> pos;
    std::vector<std::vector<float>> val;
};

int export_rows(Rows& r, int64_t rows, int64_t cols, bool pattern_only, sparta_csr_host* out) {
    int64_t nnz = 0;
    for (int64_t i = 0; i < rows; i++) nnz += (int64_t)r.pos[(size_t)i].size();
    if (cols > INT32_MAX) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_csr_read: more than 2^31-1 columns");
    out->rows = rows; out->cols = cols; out->nnz = nnz; out->pattern_only = pattern_only ? 1 : 0;
    out->rowptr = (int64_t*)std::malloc(sizeof(int64_t) * (size_t)(rows + 1));
    out->colidx = (int32_t*)std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1));
    out->vals = pattern_only ? nullptr : (float*)std::malloc(sizeof(float) * (size_t)std::max<int64_t>(nnz, 1));
    if (!out->rowptr || !out->colidx || (!pattern_only && !out->vals)) {
        sparta_csr_host_free(out);
        return fail(SPARTA_ERR_ALLOC, "sparta_csr_read: out of host memory");
    }
    int64_t p = 0;
    for (int64_t i = 0; i < rows; i++) {
        out->rowptr[i] = p;
        const auto& rp = r.pos[(size_t)i];
        for (size_t k = 0; k < rp.size(); k++) {
            out->colidx[p] = (int32_t)rp[k];
            if (!pattern_only) out->vals[p] = r.val[(size_t)i][k];
            p++;
        }
    }
    out->rowptr[rows] = p;
    return SPARTA_OK;
}

// ---- edge list ---------------------------------------------------------------------------------------------------------
// csr.cpp:196-307.  COMPAT: the first line after the leading comments is read and thrown away (:213), whatever it holds --
// the reference's 13-line test matrix yields 12 nonzeros; row ids must not decrease (:259); rows = last row id + 1, cols =
// largest column id + 1 (:286-287); duplicates and the column order inside a row are kept as they come.
int read_el(std::istream& in, const std::string& delim, bool pattern_only, bool symmetrize, bool compat, sparta_csr_host* out) {
    if (delim.empty()) return fail(SPARTA_ERR_INVALID, "sparta_csr_read: empty delimiter");
    Rows r;
    std::string line;
    int64_t i = -1, max_col = 0, lineno = 0;
    bool triangular = true;
    skip_leading_comments(in);
    if (compat) std::getline(in, line);
    while (std::getline(in, line)) {
        lineno++;
        if (!compat) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty() || line[0] == '#' || line[0] == '%') continue;
        }
        // Field splitting exactly as csr.cpp:219-242: find / substr / erase(0, pos + len) with size_t arithmetic, so a missing
        // delimiter (pos = npos) erases len - 1 characters and the same text is parsed again -- "7" is the entry (7, 7) and
        // "1 5" read with values is (1, 5) = 5.0.  Defined behaviour in the reference, reproduced here.
        std::string t = line;
        auto erase_field = [&](size_t pos) { t.erase(0, std::min(t.size(), pos + delim.size())); };   // pos + len wraps for npos
        size_t dp = t.find(delim);
        long a = 0, b = 0;
        if (!stoi_like(t.substr(0, dp), &a)) return fail(SPARTA_ERR_IO, "edge list: bad row id in data line " + std::to_string(lineno) + ": '" + line + "'");
        erase_field(dp);
        dp = t.find(delim);
        if (!stoi_like(t.substr(0, dp), &b)) return fail(SPARTA_ERR_IO, "edge list: bad column id in data line " + std::to_string(lineno) + ": '" + line + "'");
        float v = 1.0f;
        if (!pattern_only) {
            erase_field(dp);
            dp = t.find(delim);
            if (!stof_like(t.substr(0, dp), &v)) return fail(SPARTA_ERR_IO, "edge list: bad value in data line " + std::to_string(lineno) + ": '" + line + "'");
        }
        if (a < 0 || b < 0) return fail(SPARTA_ERR_IO, "edge list: negative index in data line " + std::to_string(lineno));
        if (b < a) triangular = false;
        max_col = std::max<int64_t>(max_col, b);
        if (a > i) {
            while (i < a) {
                r.pos.emplace_back();
                if (!pattern_only) r.val.emplace_back();
                i++;
            }
        } else if (a < i) {
            return fail(SPARTA_ERR_IO, "edge list: row ids must be in ascending order (data line " + std::to_string(lineno) + ")");
        }
        r.pos[(size_t)i].push_back(b);
        if (!pattern_only) r.val[(size_t)i].push_back(v);
    }
    // csr.cpp:265-284: mirror an upper-triangular pattern.  Same walk as the reference (rows grow while they are walked).
    if (symmetrize && triangular) {
        for (size_t ii = 0; ii < r.pos.size(); ii++) {
            for (size_t nz = 0; nz < r.pos[ii].size(); nz++) {
                const int64_t j = r.pos[ii][nz];
                if ((size_t)j >= r.pos.size())
                    return fail(SPARTA_ERR_IO, "edge list: symmetrize needs a square pattern (column " + std::to_string(j) + " has no row)");
                auto it = std::lower_bound(r.pos[(size_t)j].begin(), r.pos[(size_t)j].end(), (int64_t)ii);
                if (it == r.pos[(size_t)j].end() || *it != (int64_t)ii) {
                    if (!pattern_only)
                        return fail(SPARTA_ERR_INVALID, "symmetrize is only implemented for unweighted (pattern-only) inputs");
                    r.pos[(size_t)j].insert(it, (int64_t)ii);
                }
            }
        }
    }
    return export_rows(r, (int64_t)r.pos.size(), max_col + 1, pattern_only, out);
}

// ---- MatrixMarket --------------------------------------------------------------------------------------------------------
// COMPAT (csr.cpp:309-365): always pattern-only; "rows cols nnz" from the first line after the leading comments, then ONE
// MORE LINE IS SKIPPED (:319), then exactly nnz lines "i j ..." (1-based) are read; symmetry is ignored.  A standard file
// therefore loses its first entry and comes up one line short: the reference then indexes with an unread value (undefined
// behaviour); here that is SPARTA_ERR_IO.
int read_mtx_compat(std::istream& in, sparta_csr_host* out) {
    skip_leading_comments(in);
    std::string line;
    std::getline(in, line);
    std::istringstream hs(line);
    int rows = 0, cols = 0, nnz = 0;
    if (!(hs >> rows >> cols >> nnz) || rows < 0 || cols < 0 || nnz < 0) return fail(SPARTA_ERR_IO, "MatrixMarket: bad size line '" + line + "'");
    in.ignore(2048, '\n');
    Rows r;
    r.pos.resize((size_t)rows);
    for (int k = 0; k < nnz; k++) {
        if (!std::getline(in, line))
            return fail(SPARTA_ERR_IO, "MatrixMarket (reference-compatible mode): " + std::to_string(nnz) + " entries announced, " + std::to_string(k) +
                                           " readable after the skipped line (the reference reads past the end here)");
        std::istringstream ls(line);
        int i = 0, j = 0;
        if (!(ls >> i >> j)) return fail(SPARTA_ERR_IO, "MatrixMarket: bad entry line '" + line + "'");
        i--; j--;
        if (i < 0 || i >= rows || j < 0) return fail(SPARTA_ERR_IO, "MatrixMarket: index out of range in line '" + line + "'");
        r.pos[(size_t)i].push_back(j);
    }
    return export_rows(r, rows, cols, true, out);
}

// STRICT: "%%MatrixMarket matrix coordinate <real|integer|pattern> <general|symmetric|skew-symmetric>"; entries sorted by
// (row, column), stable; symmetric inputs are mirrored.
int read_mtx_strict(std::istream& in, bool pattern_only, sparta_csr_host* out) {
    std::string line;
    bool sym = false, skew = false, file_pattern = false;
    bool have_size = false;
    int64_t rows = 0, cols = 0, nnz = 0, got = 0;
    struct E { int64_t i, j; float v; };
    std::vector<E> es;
    bool first = true;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (first) {
            first = false;
            if (line.rfind("%%MatrixMarket", 0) == 0) {
                std::string low = line;
                std::transform(low.begin(), low.end(), low.begin(), [](unsigned char c) { return (char)std::tolower(c); });
                if (low.find("coordinate") == std::string::npos) return fail(SPARTA_ERR_UNSUPPORTED, "MatrixMarket: only the coordinate format is read");
                if (low.find("complex") != std::string::npos) return fail(SPARTA_ERR_UNSUPPORTED, "MatrixMarket: complex field");
                file_pattern = low.find("pattern") != std::string::npos;
                skew = low.find("skew-symmetric") != std::string::npos;
                sym = !skew && low.find("symmetric") != std::string::npos;
                if (low.find("hermitian") != std::string::npos) sym = true;
                continue;
            }
        }
        if (line.empty() || line[0] == '%' || line[0] == '#') continue;
        std::istringstream ls(line);
        if (!have_size) {
            if (!(ls >> rows >> cols >> nnz) || rows < 0 || cols < 0 || nnz < 0) return fail(SPARTA_ERR_IO, "MatrixMarket: bad size line '" + line + "'");
            have_size = true;
            es.reserve((size_t)nnz * (sym || skew ? 2 : 1));
            continue;
        }
        if (got == nnz) break;
        int64_t i = 0, j = 0;
        double v = 1.0;
        if (!(ls >> i >> j)) return fail(SPARTA_ERR_IO, "MatrixMarket: bad entry line '" + line + "'");
        if (!file_pattern && !(ls >> v)) return fail(SPARTA_ERR_IO, "MatrixMarket: entry without a value: '" + line + "'");
        if (i < 1 || i > rows || j < 1 || j > cols) return fail(SPARTA_ERR_IO, "MatrixMarket: index out of range in '" + line + "'");
        es.push_back(E{i - 1, j - 1, (float)v});
        if ((sym || skew) && i != j) es.push_back(E{j - 1, i - 1, skew ? -(float)v : (float)v});
        got++;
    }
    if (!have_size) return fail(SPARTA_ERR_IO, "MatrixMarket: no size line");
    if (got != nnz) return fail(SPARTA_ERR_IO, "MatrixMarket: " + std::to_string(nnz) + " entries announced, " + std::to_string(got) + " found");
    std::stable_sort(es.begin(), es.end(), [](const E& a, const E& b) { return a.i != b.i ? a.i < b.i : a.j < b.j; });
    Rows r;
    r.pos.resize((size_t)rows);
    const bool po = pattern_only || file_pattern;
    if (!po) r.val.resize((size_t)rows);
    for (const E& e : es) {
        r.pos[(size_t)e.i].push_back(e.j);
        if (!po) r.val[(size_t)e.i].push_back(e.v);
    }
    return export_rows(r, rows, cols, po, out);
}

std::string f2s(float v) {                                  // std::to_string(float): "%f" of the value promoted to double
    char buf[64];
    std::snprintf(buf, sizeof buf, "%f", (double)v);
    return buf;
}

}  // namespace

extern "C" {

static int read_stream(std::istream& in, const char* delimiter, int32_t pattern_only, int32_t mat_fmt, int32_t symmetrize, int32_t mode,
                       sparta_csr_host* out) {
    if (mat_fmt != SPARTA_FMT_EL && mat_fmt != SPARTA_FMT_MTX) return fail(SPARTA_ERR_INVALID, "sparta_csr_read: mat_fmt must be SPARTA_FMT_EL or SPARTA_FMT_MTX");
    if (mode != SPARTA_IO_COMPAT && mode != SPARTA_IO_STRICT) return fail(SPARTA_ERR_INVALID, "sparta_csr_read: mode must be SPARTA_IO_COMPAT or SPARTA_IO_STRICT");
    const std::string delim = delimiter ? delimiter : " ";
    if (mat_fmt == SPARTA_FMT_MTX) {
        if (mode == SPARTA_IO_COMPAT) return read_mtx_compat(in, out);
        return read_mtx_strict(in, pattern_only != 0, out);
    }
    return read_el(in, delim, pattern_only != 0, symmetrize != 0, mode == SPARTA_IO_COMPAT, out);
}

int sparta_csr_read(const char* path, const char* delimiter, int32_t pattern_only, int32_t mat_fmt, int32_t symmetrize, int32_t mode,
                    sparta_csr_host* out) {
    if (!path || !out) return fail(SPARTA_ERR_INVALID, "sparta_csr_read: NULL argument");
    std::memset(out, 0, sizeof *out);
    std::ifstream in(path);
    if (!in.good()) return fail(SPARTA_ERR_IO, std::string("sparta_csr_read: cannot open '") + path + "'");
    return read_stream(in, delimiter, pattern_only, mat_fmt, symmetrize, mode, out);
}

int sparta_csr_read_buffer(const char* text, int64_t len, const char* delimiter, int32_t pattern_only, int32_t mat_fmt, int32_t symmetrize,
                           int32_t mode, sparta_csr_host* out) {
    if ((!text && len > 0) || len < 0 || !out) return fail(SPARTA_ERR_INVALID, "sparta_csr_read_buffer: bad argument");
    std::memset(out, 0, sizeof *out);
    std::istringstream in(std::string(text ? text : "", (size_t)len));
    return read_stream(in, delimiter, pattern_only, mat_fmt, symmetrize, mode, out);
}

void sparta_csr_host_free(sparta_csr_host* m) {
    if (!m) return;
    std::free(m->rowptr);
    std::free(m->colidx);
    std::free(m->vals);
    std::memset(m, 0, sizeof *m);
}

// csr.cpp:169-179: "i<delim>j" per stored entry (.el) or "j<delim>i" (.mtx flavour of the reference: no header, 0-based)
int sparta_csr_write_edgelist(const char* path, int64_t rows, const int64_t* rowptr, const int32_t* colidx, const char* delimiter,
                              int32_t mat_fmt) {
    if (!path || !rowptr || (!colidx && rowptr[rows] > 0) || rows < 0) return fail(SPARTA_ERR_INVALID, "sparta_csr_write_edgelist: bad argument");
    std::ofstream o(path);
    if (!o.good()) return fail(SPARTA_ERR_IO, std::string("sparta_csr_write_edgelist: cannot open '") + path + "'");
    const std::string d = delimiter ? delimiter : " ";
    for (int64_t i = 0; i < rows; i++)
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; k++) {
            if (mat_fmt == SPARTA_FMT_MTX) o << colidx[k] << d << i << "\n";
            else o << i << d << colidx[k] << "\n";
        }
    o.flush();
    return o.good() ? SPARTA_OK : fail(SPARTA_ERR_IO, "sparta_csr_write_edgelist: write failed");
}

// utilities.cpp:239-243: one group id per line
int sparta_grouping_write(const char* path, const int64_t* grouping, int64_t n) {
    if (!path || (!grouping && n > 0) || n < 0) return fail(SPARTA_ERR_INVALID, "sparta_grouping_write: bad argument");
    std::ofstream o(path);
    if (!o.good()) return fail(SPARTA_ERR_IO, std::string("sparta_grouping_write: cannot open '") + path + "'");
    for (int64_t i = 0; i < n; i++) o << grouping[i] << "\n";
    o.flush();
    return o.good() ? SPARTA_OK : fail(SPARTA_ERR_IO, "sparta_grouping_write: write failed");
}

// Matrix_Analysis.cpp:10-32: std::stoi per line, lines that do not start with a number are skipped (the reference prints a
// message and goes on); :77-78: a file with rows + 1 numbers has a count in front, which is dropped.
int sparta_grouping_read(const char* path, int64_t expected_rows, int64_t* out, int64_t capacity, int64_t* n_out) {
    if (!path || !n_out || (!out && capacity > 0)) return fail(SPARTA_ERR_INVALID, "sparta_grouping_read: NULL argument");
    std::ifstream in(path);
    if (!in.good()) return fail(SPARTA_ERR_IO, std::string("sparta_grouping_read: cannot open '") + path + "'");
    std::vector<int64_t> g;
    std::string line;
    while (std::getline(in, line)) {
        long v;
        if (stoi_like(line, &v)) g.push_back(v);
    }
    if (expected_rows >= 0 && (int64_t)g.size() == expected_rows + 1) g.erase(g.begin());
    *n_out = (int64_t)g.size();
    if (expected_rows >= 0 && (int64_t)g.size() != expected_rows)
        return fail(SPARTA_ERR_IO, "sparta_grouping_read: " + std::to_string(g.size()) + " group ids, matrix has " + std::to_string(expected_rows) + " rows");
    if ((int64_t)g.size() > capacity) return fail(SPARTA_ERR_INVALID, "sparta_grouping_read: output buffer too small");
    std::copy(g.begin(), g.end(), out);
    return SPARTA_OK;
}

// utilities.cpp:175-236: header line and value line of the 32-column row, every field followed by a comma
int sparta_blocking_csv_row(const sparta_csv_fields* f, char* header_out, int64_t header_cap, char* values_out, int64_t values_cap) {
    if (!f || !header_out || !values_out) return fail(SPARTA_ERR_INVALID, "sparta_blocking_csv_row: NULL argument");
    std::string header, values;
    auto add = [&](const char* name, const std::string& value) { header += std::string(name) + ","; values += value + ","; };
    using std::to_string;
    add("matrix", f->matrix ? f->matrix : "");
    add("rows", to_string((long)f->rows));
    add("cols", to_string((long)f->cols));
    add("nonzeros", to_string((long)f->nonzeros));
    add("symmetrize", to_string((int)(f->symmetrize != 0)));
    add("blocking_algo", to_string((int)f->blocking_algo));
    add("tau", f2s(f->tau));
    add("row_block_size", to_string((int)f->row_block_size));
    add("col_block_size", to_string((int)f->col_block_size));
    add("use_pattern", to_string((int)(f->use_pattern != 0)));
    add("sim_use_groups", to_string((int)(f->sim_use_groups != 0)));
    add("sim_measure", to_string((int)f->sim_measure));
    add("reorder", to_string((int)f->reorder));
    add("exp_name", f->exp_name ? f->exp_name : "");
    add("b_cols", to_string((int)f->b_cols));
    add("warmup", to_string((int)f->warmup));
    add("exp_repetitions", to_string((int)f->exp_repetitions));
    add("multiplication_algo", to_string((int)f->multiplication_algo));
    add("n_streams", to_string((int)f->n_streams));
    add("time_to_block", f2s(f->time_to_block));
    add("time_to_merge", f2s(f->time_to_merge));
    add("time_to_compare", f2s(f->time_to_compare));
    add("VBR_nzcount", to_string((long)f->vbr_nzcount));
    add("VBR_nzblocks_count", to_string((long)f->vbr_nzblocks_count));
    add("VBR_average_height", f2s(f->vbr_average_height));
    add("VBR_longest_row", to_string((long)f->vbr_longest_row));
    add("merge_counter", to_string((long)f->merge_counter));
    add("comparison_counter", to_string((long)f->comparison_counter));
    add("average_merge_tau", f2s(f->average_merge_tau));
    add("average_row_distance", f2s(f->average_row_distance));
    add("avg_time_multiply", f2s(f->avg_time_multiply));
    add("std_time_multiply", f2s(f->std_time_multiply));
    if ((int64_t)header.size() + 1 > header_cap || (int64_t)values.size() + 1 > values_cap)
        return fail(SPARTA_ERR_INVALID, "sparta_blocking_csv_row: output buffer too small");
    std::memcpy(header_out, header.c_str(), header.size() + 1);
    std::memcpy(values_out, values.c_str(), values.size() + 1);
    return SPARTA_OK;
}

// csr.cpp:123-155: the permutation CSR::reorder_by_degree applies (flag -r: -1 ascending, 1 descending).  The reference
// sorts the row ids with std::sort and the comparators `n[i] < n[j]` (ascending) / `n[i] >= n[j]` (descending).  Same
// algorithm from the same libstdc++ with the same comparator => the same permutation, ties included.
// `>=` is not a strict weak ordering: on ties std::sort's unguarded loops can walk off either end of the reference's array
// (undefined behaviour there).  Here the ids sit between two sentinels that stop such a walk; whenever the reference's sort
// stays inside its array the two results are identical.  perm_out[k] = old index of the row that moves to position k.
int sparta_degree_permutation(int64_t rows, const int64_t* rowptr, int32_t descending, int64_t* perm_out) {
    if (rows < 0 || (rows > 0 && (!rowptr || !perm_out))) return fail(SPARTA_ERR_INVALID, "sparta_degree_permutation: bad argument");
    std::vector<int64_t> deg((size_t)rows);
    for (int64_t i = 0; i < rows; i++) deg[(size_t)i] = rowptr[i + 1] - rowptr[i];
    if (!descending) {
        std::iota(perm_out, perm_out + rows, (int64_t)0);
        std::sort(perm_out, perm_out + rows, [&](int64_t i, int64_t j) { return deg[(size_t)i] < deg[(size_t)j]; });
        return SPARTA_OK;
    }
    std::vector<int64_t> buf((size_t)rows + 2);
    buf[0] = -1;                                            // "before everything"
    std::iota(buf.begin() + 1, buf.end() - 1, (int64_t)0);
    buf[(size_t)rows + 1] = rows;                           // "after everything"
    auto cmp = [&](int64_t i, int64_t j) {
        if (i == -1 || j == rows) return i != j;
        if (j == -1 || i == rows) return false;
        return deg[(size_t)i] >= deg[(size_t)j];
    };
    std::sort(buf.begin() + 1, buf.end() - 1, cmp);
    std::copy(buf.begin() + 1, buf.end() - 1, perm_out);
    return SPARTA_OK;
}

}  // extern "C"

// ---- binary VBS container -----------------------------------------------------------------------------------------------------
// The reorder + build cost is paid once: a VBS (the reference's five arrays, include/matrices.h:93-104) in one little-endian file.
//   bytes  0.. 7  magic "SPARTAVB"          8..11  uint32 version = 1        12..15  uint32 header bytes = 96
//         16..79  int64 rows, cols, block_rows, block_cols, block_col_size, nztot, nblocks, reserved (0)
//         80..87  uint64 FNV-1a 64 of the payload                              88..95  uint64 payload bytes
//   payload: int64 row_part[block_rows + 1] | int64 nzcount[block_rows] | int64 jab[nblocks] | float mab[nztot]
// No reference counterpart ("a binary VBS container so reorder cost is paid once").
namespace {
constexpr char kVbsMagic[8] = {'S', 'P', 'A', 'R', 'T', 'A', 'V', 'B'};
inline uint64_t fnv1a(const void* data, size_t n, uint64_t h) {
    const unsigned char* p = (const unsigned char*)data;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
}  // namespace

extern "C" {

// Blocked-ELL view of a VBS with square blocks -- restates prepare_cusparse_BLOCKEDELLPACK (src/cuda/cuda_utilities.cpp:1656-1710),
// the mapping the reference feeds to cusparseCreateBlockedEll: ell_blocksize = block_col_size; ellColInd[rows/bs][ell_cols] holds
// the block-column ids of a block-row in VBS order, padded with -1 ("the algorithm automatically pads it with a complete zero
// block", :1691); ellValues is row-major rows x (ell_cols * bs), element (k*bs + i, j) = mab[shift_k + j*bs + i] when block j / bs
// of block-row k exists, else 0 (:1700-1701).  The reference requires rows % bs == 0 and cols % bs == 0 (exit otherwise,
// :1666-1672) and silently ASSUMES that every block-row is bs rows tall (it indexes nzcount by i < rows / bs and strides mab by
// bs*bs per block, :1675-1705): here that assumption is checked.
int sparta_vbs_to_blocked_ell(const sparta_vbs_host* v, int64_t* ell_cols_out, int64_t* ell_col_ind, float* ell_values) {
    if (!v || !ell_cols_out) return fail(SPARTA_ERR_INVALID, "sparta_vbs_to_blocked_ell: NULL argument");
    const int64_t bs = v->block_col_size;
    if (bs <= 0 || v->rows % bs != 0 || v->cols % bs != 0)
        return fail(SPARTA_ERR_INVALID, v->rows % std::max<int64_t>(bs, 1) != 0 ? "The number of rows is not multiple of ell_blocksize"
                                                                                : "The number of cols is not multiple of ell_blocksize");
    if (v->block_rows != v->rows / bs) return fail(SPARTA_ERR_INVALID, "sparta_vbs_to_blocked_ell: every block-row must be block_col_size rows tall (fixed square blocking)");
    int64_t ell_cols = 0;
    for (int64_t k = 0; k < v->block_rows; k++) {
        if (v->row_part[k + 1] - v->row_part[k] != bs)
            return fail(SPARTA_ERR_INVALID, "sparta_vbs_to_blocked_ell: every block-row must be block_col_size rows tall (fixed square blocking)");
        ell_cols = std::max(ell_cols, v->nzcount[k]);
    }
    *ell_cols_out = ell_cols;
    if (!ell_col_ind || !ell_values) return SPARTA_OK;                     // size query
    const int64_t vcols = ell_cols * bs;
    int64_t jo = 0, mo = 0;
    for (int64_t k = 0; k < v->block_rows; k++) {
        const int64_t nb = v->nzcount[k];
        for (int64_t j = 0; j < ell_cols; j++) ell_col_ind[k * ell_cols + j] = j < nb ? v->jab[jo + j] : -1;
        for (int64_t i = 0; i < bs; i++) {
            float* row = ell_values + (k * bs + i) * vcols;
            for (int64_t j = 0; j < vcols; j++) row[j] = j / bs < nb ? v->mab[mo + j * bs + i] : 0.0f;
        }
        jo += nb;
        mo += nb * bs * bs;
    }
    return SPARTA_OK;
}

int sparta_vbs_save(const char* path, const sparta_vbs_host* v) {
    if (!path || !v) return fail(SPARTA_ERR_INVALID, "sparta_vbs_save: NULL argument");
    if (v->block_rows < 0 || v->nblocks < 0 || v->nztot < 0 || !v->row_part || (v->block_rows > 0 && !v->nzcount) || (v->nblocks > 0 && !v->jab) ||
        (v->nztot > 0 && !v->mab))
        return fail(SPARTA_ERR_INVALID, "sparta_vbs_save: inconsistent VBS");
    const size_t n_rp = (size_t)(v->block_rows + 1) * 8, n_nz = (size_t)v->block_rows * 8, n_jab = (size_t)v->nblocks * 8, n_mab = (size_t)v->nztot * 4;
    uint64_t h = 14695981039346656037ull;
    h = fnv1a(v->row_part, n_rp, h);
    h = fnv1a(v->nzcount, n_nz, h);
    h = fnv1a(v->jab, n_jab, h);
    h = fnv1a(v->mab, n_mab, h);
    unsigned char head[96];
    std::memset(head, 0, sizeof head);
    std::memcpy(head, kVbsMagic, 8);
    const uint32_t ver = 1, hb = 96;
    std::memcpy(head + 8, &ver, 4);
    std::memcpy(head + 12, &hb, 4);
    const int64_t dims[8] = {v->rows, v->cols, v->block_rows, v->block_cols, v->block_col_size, v->nztot, v->nblocks, 0};
    std::memcpy(head + 16, dims, 64);
    const uint64_t payload = n_rp + n_nz + n_jab + n_mab;
    std::memcpy(head + 80, &h, 8);
    std::memcpy(head + 88, &payload, 8);
    std::ofstream o(path, std::ios::binary);
    if (!o.good()) return fail(SPARTA_ERR_IO, std::string("sparta_vbs_save: cannot open '") + path + "'");
    o.write((const char*)head, 96);
    o.write((const char*)v->row_part, (std::streamsize)n_rp);
    o.write((const char*)v->nzcount, (std::streamsize)n_nz);
    o.write((const char*)v->jab, (std::streamsize)n_jab);
    o.write((const char*)v->mab, (std::streamsize)n_mab);
    o.flush();
    return o.good() ? SPARTA_OK : fail(SPARTA_ERR_IO, "sparta_vbs_save: write failed");
}

int sparta_vbs_load(const char* path, sparta_vbs_host* out) {
    if (!path || !out) return fail(SPARTA_ERR_INVALID, "sparta_vbs_load: NULL argument");
    std::memset(out, 0, sizeof *out);
    std::ifstream in(path, std::ios::binary);
    if (!in.good()) return fail(SPARTA_ERR_IO, std::string("sparta_vbs_load: cannot open '") + path + "'");
    unsigned char head[96];
    in.read((char*)head, 96);
    if (in.gcount() != 96 || std::memcmp(head, kVbsMagic, 8) != 0) return fail(SPARTA_ERR_IO, "sparta_vbs_load: not a SPARTAVB file");
    uint32_t ver, hb;
    std::memcpy(&ver, head + 8, 4);
    std::memcpy(&hb, head + 12, 4);
    if (ver != 1 || hb != 96) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_load: unknown container version");
    int64_t dims[8];
    uint64_t want_hash, payload;
    std::memcpy(dims, head + 16, 64);
    std::memcpy(&want_hash, head + 80, 8);
    std::memcpy(&payload, head + 88, 8);
    const int64_t rows = dims[0], cols = dims[1], br = dims[2], bc = dims[3], w = dims[4], nztot = dims[5], nblocks = dims[6];
    if (rows < 0 || cols < 0 || br < 0 || bc < 0 || w <= 0 || nztot < 0 || nblocks < 0 || br > (int64_t)1 << 40 || nblocks > (int64_t)1 << 40 ||
        nztot > (int64_t)1 << 44)
        return fail(SPARTA_ERR_IO, "sparta_vbs_load: implausible header");
    const size_t n_rp = (size_t)(br + 1) * 8, n_nz = (size_t)br * 8, n_jab = (size_t)nblocks * 8, n_mab = (size_t)nztot * 4;
    if (payload != n_rp + n_nz + n_jab + n_mab) return fail(SPARTA_ERR_IO, "sparta_vbs_load: header and payload size disagree");
    out->rows = rows; out->cols = cols; out->block_rows = br; out->block_cols = bc; out->block_col_size = w; out->nztot = nztot; out->nblocks = nblocks;
    out->row_part = (int64_t*)std::malloc(n_rp);
    out->nzcount = (int64_t*)std::malloc(std::max<size_t>(n_nz, 8));
    out->jab = (int64_t*)std::malloc(std::max<size_t>(n_jab, 8));
    out->mab = (float*)std::malloc(std::max<size_t>(n_mab, 4));
    auto bail = [&](int code, const char* msg) { sparta_vbs_host_free(out); return fail(code, msg); };
    if (!out->row_part || !out->nzcount || !out->jab || !out->mab) return bail(SPARTA_ERR_ALLOC, "sparta_vbs_load: out of host memory");
    in.read((char*)out->row_part, (std::streamsize)n_rp);
    in.read((char*)out->nzcount, (std::streamsize)n_nz);
    in.read((char*)out->jab, (std::streamsize)n_jab);
    in.read((char*)out->mab, (std::streamsize)n_mab);
    if (!in.good() && !(in.eof() && (size_t)in.gcount() == n_mab)) return bail(SPARTA_ERR_IO, "sparta_vbs_load: file is truncated");
    uint64_t h = 14695981039346656037ull;
    h = fnv1a(out->row_part, n_rp, h);
    h = fnv1a(out->nzcount, n_nz, h);
    h = fnv1a(out->jab, n_jab, h);
    h = fnv1a(out->mab, n_mab, h);
    if (h != want_hash) return bail(SPARTA_ERR_IO, "sparta_vbs_load: checksum mismatch (corrupted file)");
    // structural checks: what sparta_vbs_create would reject later is rejected here with the file's name on it
    if (out->row_part[0] != 0 || out->row_part[br] != rows) return bail(SPARTA_ERR_IO, "sparta_vbs_load: row_part does not span [0, rows]");
    int64_t nb = 0, area = 0;
    for (int64_t i = 0; i < br; i++) {
        const int64_t hh = out->row_part[i + 1] - out->row_part[i];
        if (hh < 0 || out->nzcount[i] < 0) return bail(SPARTA_ERR_IO, "sparta_vbs_load: negative height or count");
        nb += out->nzcount[i];
        area += out->nzcount[i] * hh * w;
    }
    if (nb != nblocks || area != nztot) return bail(SPARTA_ERR_IO, "sparta_vbs_load: counts do not add up");
    return SPARTA_OK;
}

}  // extern "C"
