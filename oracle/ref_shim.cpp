// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A C-ABI window onto the *real* reference implementation.  This file is our own
// code; it is compiled TOGETHER WITH the reference's unmodified sources, taken where
// they lie under /root/reference (src/general/{csr,vbr,blocking,utilities}.cpp), by
// oracle/Makefile into oracle/_ref/libsparta_ref.so.  No reference source is copied
// into this repository.  The shared object is used
//   * to pin the plain-C restatement in oracle/sparta_oracle.c (tests/test_oracle_vs_ref.py),
//   * to generate the golden vectors under tests/golden/ (tests/golden/make_golden.py),
//   * as the "reference" CPU baseline timed by bench.py.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
#include <fstream>
#include <vector>
#include <cstring>
#include <algorithm>
#include "matrices.h"     // reference: include/matrices.h  (struct CSR, struct VBR)
#include "blocking.h"     // reference: include/blocking.h  (BlockingEngine, distances)
#include "utilities.h"    // reference: include/utilities.h (merge_rows, get_permutation, ...)

namespace {
// The reference's CSR has no array constructor (include/matrices.h:58-82); build an
// empty one through its stream constructor on an empty stream and install our rows.
CSR* make_empty_csr(bool pattern_only)
{
    std::ifstream nothing("/dev/null");
    CSR* c = new CSR(nothing, " ", pattern_only, el);   // rows = 0, cols = 1
    delete[] c->nzcount;
    delete[] c->ja;
    if (!pattern_only) delete[] c->ma;
    c->nzcount = nullptr; c->ja = nullptr; c->ma = nullptr;
    return c;
}
}

extern "C" {

void* ref_csr_create(long rows, long cols, const long* rowptr, const long* colidx, const float* vals)
{
    const bool pattern_only = (vals == nullptr);
    CSR* c = make_empty_csr(pattern_only);
    c->rows = rows; c->cols = cols; c->pattern_only = pattern_only;
    c->nzcount = new intT[rows];
    c->ja = new intT*[rows];
    if (!pattern_only) c->ma = new DataT*[rows];
    for (long i = 0; i < rows; i++) {
        long n = rowptr[i + 1] - rowptr[i];
        c->nzcount[i] = n;
        c->ja[i] = new intT[n];
        std::copy(colidx + rowptr[i], colidx + rowptr[i + 1], c->ja[i]);
        if (!pattern_only) {
            c->ma[i] = new DataT[n];
            std::copy(vals + rowptr[i], vals + rowptr[i + 1], c->ma[i]);
        }
    }
    return c;
}

// reference readers: src/general/csr.cpp:183-365
void* ref_csr_read(const char* path, const char* delim, int pattern_only, int mat_fmt, int symmetrize)
{
    std::ifstream fin(path);
    if (!fin.good()) return nullptr;
    CSR* c = make_empty_csr(pattern_only != 0);
    c->rows = 0; c->cols = 0;
    try {
        c->read_from_edgelist(fin, delim, pattern_only != 0, (MatrixFormat)mat_fmt, symmetrize != 0);
    } catch (...) {
        c->rows = 0; c->cols = 0;
        delete c;
        return nullptr;
    }
    return c;
}

void ref_csr_destroy(void* h) { delete (CSR*)h; }
long ref_csr_rows(void* h) { return ((CSR*)h)->rows; }
long ref_csr_cols(void* h) { return ((CSR*)h)->cols; }
long ref_csr_nnz(void* h) { return ((CSR*)h)->nztot(); }
int  ref_csr_pattern_only(void* h) { return ((CSR*)h)->pattern_only ? 1 : 0; }

void ref_csr_export(void* h, long* rowptr, long* colidx, float* vals)
{
    CSR* c = (CSR*)h;
    long p = 0;
    for (long i = 0; i < c->rows; i++) {
        rowptr[i] = p;
        for (long k = 0; k < c->nzcount[i]; k++) {
            colidx[p] = c->ja[i][k];
            if (vals) vals[p] = c->pattern_only ? 1.0f : c->ma[i][k];
            p++;
        }
    }
    rowptr[c->rows] = p;
}

// reference: src/general/csr.cpp:49-65
void ref_csr_multiply(void* h, float* B, long B_cols, float* C) { ((CSR*)h)->multiply(B, B_cols, C); }

// reference: src/general/blocking.cpp:633-676 (dispatch), :156-243, :433-549, :554-562
// stats_out: [comparison_counter, merge_counter]; fstats_out: [avg_row_distance, avg_merge_tau]
int ref_get_grouping(void* h, int algo, float tau, long col_block_size, long row_block_size,
                     int use_groups, int use_pattern, int force_fixed_size, int sim_measure,
                     long* grouping_out, long* stats_out, float* fstats_out, long* info_out, float* finfo_out)
{
    CSR* c = (CSR*)h;
    BlockingEngine e;
    e.tau = tau;
    e.col_block_size = col_block_size;
    e.row_block_size = row_block_size;
    e.use_groups = use_groups != 0;
    e.use_pattern = use_pattern != 0;
    e.force_fixed_size = force_fixed_size != 0;
    e.blocking_algo = (BlockingType)algo;
    e.SetComparator(sim_measure);
    std::vector<intT> g = e.GetGrouping(*c);
    std::copy(g.begin(), g.end(), grouping_out);
    if (stats_out) { stats_out[0] = e.comparison_counter; stats_out[1] = e.merge_counter; }
    if (fstats_out) { fstats_out[0] = e.average_row_distance; fstats_out[1] = e.average_merge_tau; }
    if (info_out) {
        e.CollectBlockingInfo(*c);       // reference: src/general/blocking.cpp:576-631
        info_out[0] = e.VBR_nzcount; info_out[1] = e.VBR_nzblocks_count; info_out[2] = e.VBR_longest_row;
        if (finfo_out) finfo_out[0] = e.VBR_average_height;
    }
    return 0;
}

// reference: src/general/utilities.cpp:8-54
void ref_get_permutation(const long* grouping, long n, long* out)
{
    std::vector<intT> g(grouping, grouping + n);
    std::vector<intT> p = get_permutation(g);
    std::copy(p.begin(), p.end(), out);
}
long ref_get_partition(const long* grouping, long n, long* out)
{
    std::vector<intT> g(grouping, grouping + n);
    std::vector<intT> p = get_partition(g);
    std::copy(p.begin(), p.end(), out);
    return (long)p.size();
}
void ref_get_fixed_size_grouping(const long* grouping, long n, long row_block_size, long* out)
{
    std::vector<intT> g(grouping, grouping + n);
    std::vector<intT> p = get_fixed_size_grouping(g, row_block_size);
    std::copy(p.begin(), p.end(), out);
}

// reference: src/general/utilities.cpp:145-173
long ref_merge_rows(const long* A, long nA, long* B, long nB, long* out)
{
    std::vector<intT> a(A, A + nA);
    std::vector<intT> r = merge_rows(a, B, nB);
    std::copy(r.begin(), r.end(), out);
    return (long)r.size();
}

// reference: src/general/blocking.cpp:859-994 (+ the "OPENMP" twins :720-856)
float ref_distance(int which, const long* A, long nA, long gA, long* B, long nB, long gB, long block_size)
{
    std::vector<intT> a(A, A + nA);
    switch (which) {
        case 0: return HammingDistanceGroup(a, gA, B, nB, gB, block_size);
        case 1: return JaccardDistanceGroup(a, gA, B, nB, gB, block_size);
        case 2: return HammingDistanceGroupOPENMP(a, gA, B, nB, gB, block_size);
        default: return JaccardDistanceGroupOPENMP(a, gA, B, nB, gB, block_size);
    }
}

// reference: src/general/vbr.cpp:135-237
void* ref_vbr_create(void* hcsr, const long* grouping, long n, long col_block_size, long row_block_size, int force_fixed_size)
{
    CSR* c = (CSR*)hcsr;
    std::vector<intT> g(grouping, grouping + n);
    VBR* v = new VBR;
    v->rows = 0; v->cols = 0; v->mab = nullptr; v->jab = nullptr; v->nzcount = nullptr; v->row_part = nullptr;
    v->fill_from_CSR_inplace(*c, g, col_block_size, row_block_size, force_fixed_size != 0);
    return v;
}
void ref_vbr_destroy(void* h) { delete (VBR*)h; }
// out: rows, cols, block_rows, block_cols, block_col_size, nztot, total nonzero blocks
void ref_vbr_dims(void* h, long* out)
{
    VBR* v = (VBR*)h;
    out[0] = v->rows; out[1] = v->cols; out[2] = v->block_rows; out[3] = v->block_cols;
    out[4] = v->block_col_size; out[5] = v->nztot;
    long nb = 0; for (long i = 0; i < v->block_rows; i++) nb += v->nzcount[i];
    out[6] = nb;
}
void ref_vbr_export(void* h, long* row_part, long* nzcount, long* jab, float* mab)
{
    VBR* v = (VBR*)h;
    std::copy(v->row_part, v->row_part + v->block_rows + 1, row_part);
    std::copy(v->nzcount, v->nzcount + v->block_rows, nzcount);
    long nb = 0; for (long i = 0; i < v->block_rows; i++) nb += v->nzcount[i];
    std::copy(v->jab, v->jab + nb, jab);
    if (mab) std::copy(v->mab, v->mab + v->nztot, mab);
}
// reference: src/general/vbr.cpp:323-372
void ref_vbr_multiply(void* h, float* B, int B_cols, float* C) { ((VBR*)h)->multiply(B, B_cols, C); }

} // extern "C"
