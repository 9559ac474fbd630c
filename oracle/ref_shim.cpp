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
#include "utilities.h"    // reference: include/utilities.h (merge_rows, get_permutation, save_blocking_data, ...)
#include "input.h"        // reference: include/input.h     (CLineReader: the fields save_blocking_data prints)
#include <sstream>

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

static int g_structured_m = 2, g_structured_n = 4;

extern "C" {

// m:n parameters of IterativeBlockingPatternMN for the following ref_get_grouping calls (blocking_algo 1)
void ref_set_structured(int m, int n) { g_structured_m = m; g_structured_n = n; }

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
    e.structured_m = g_structured_m; e.structured_n = g_structured_n;     // include/blocking.h:20-21 (defaults 2, 4)
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
// reference: src/general/vbr.cpp:239-321 (rows keep their order, block-rows given by a row partition)
void* ref_vbr_create_partition(void* hcsr, const long* row_partition, long n_part, long block_size)
{
    CSR* c = (CSR*)hcsr;
    std::vector<intT> part(row_partition, row_partition + n_part);
    VBR* v = new VBR;
    v->rows = 0; v->cols = 0; v->mab = nullptr; v->jab = nullptr; v->nzcount = nullptr; v->row_part = nullptr;
    v->fill_from_CSR(*c, part, block_size);
    return v;
}
// reference: src/general/vbr.cpp:33-49 (as an element offset into mab) and :108-118
long ref_vbr_block_start(void* h, long row_block_idx) { VBR* v = (VBR*)h; return (long)(v->get_block_start(row_block_idx) - v->mab); }
int ref_vbr_partition_check(void* h, const long* part, long n_part)
{
    std::vector<intT> p(part, part + n_part);
    return ((VBR*)h)->partition_check(p);
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


// reference: src/general/csr.cpp:101-109 / :123-155 / :169-179 (row permutations and the edge-list writer)
void ref_csr_reorder(void* h, const long* grouping, long n)
{
    std::vector<intT> g(grouping, grouping + n);
    ((CSR*)h)->reorder(g);
}
void ref_csr_reorder_by_degree(void* h, int descending) { ((CSR*)h)->reorder_by_degree(descending != 0); }
int ref_csr_save_to_edgelist(void* h, const char* path, const char* delim, int pattern_only, int mat_fmt)
{
    std::ofstream out(path);
    if (!out.good()) return -1;
    ((CSR*)h)->save_to_edgelist(out, delim, pattern_only != 0, (MatrixFormat)mat_fmt);
    return 0;
}

// reference: src/general/utilities.cpp:175-245 (save_blocking_data: the 32-column CSV row + the grouping file).
// ints: [symmetrize, blocking_algo, row_block_size, col_block_size, use_pattern, sim_use_groups, sim_measure, reorder,
//        b_cols, warmup, exp_repetitions, multiplication_algo, n_streams, force_fixed_size]
// Runs GetGrouping with those settings first (save_blocking_data prints the engine's counters and grouping_result).
// timers_in (3 floats, may be NULL): overwrite timer_total / timer_merges / timer_comparisons so that the row is reproducible;
// mult_in (2 floats, may be NULL): multiplication_timer_avg / _std.
int ref_save_blocking_data(void* h, const char* filename, const char* exp_name, const int* ints, float tau, const float* timers_in,
                           const float* mult_in, char* csv_out, long csv_cap, char* grouping_out, long grouping_cap)
{
    CSR* c = (CSR*)h;
    char prog[] = "shim";
    char* argv[] = {prog, nullptr};
    optind = 1;
    CLineReader cl(1, argv);
    cl.filename_ = filename; cl.exp_name_ = exp_name;
    cl.symmetrize_ = ints[0] != 0; cl.blocking_algo_ = ints[1]; cl.row_block_size_ = ints[2]; cl.col_block_size_ = ints[3];
    cl.sim_use_pattern_ = ints[4] != 0; cl.sim_use_groups_ = ints[5] != 0; cl.sim_measure_ = ints[6]; cl.reorder_ = ints[7];
    cl.B_cols_ = ints[8]; cl.warmup_ = ints[9]; cl.exp_repetitions_ = ints[10]; cl.multiplication_algo_ = ints[11];
    cl.n_streams_ = ints[12]; cl.tau_ = tau;
    BlockingEngine e;
    e.tau = tau; e.col_block_size = ints[3]; e.row_block_size = ints[2];
    e.use_groups = ints[5] != 0; e.use_pattern = ints[4] != 0; e.force_fixed_size = ints[13] != 0;
    e.blocking_algo = (BlockingType)ints[1];
    e.SetComparator(ints[6]);
    e.GetGrouping(*c);
    if (timers_in) { e.timer_total = timers_in[0]; e.timer_merges = timers_in[1]; e.timer_comparisons = timers_in[2]; }
    if (mult_in) { e.multiplication_timer_avg = mult_in[0]; e.multiplication_timer_std = mult_in[1]; }
    std::ostringstream csv, g;
    std::streambuf* keep = std::cout.rdbuf();
    std::ostringstream sink;
    std::cout.rdbuf(sink.rdbuf());                       // save_blocking_data chats on stdout
    save_blocking_data(csv, cl, e, *c, true, g);
    std::cout.rdbuf(keep);
    const std::string a = csv.str(), b = g.str();
    if ((long)a.size() + 1 > csv_cap || (long)b.size() + 1 > grouping_cap) return -1;
    std::memcpy(csv_out, a.c_str(), a.size() + 1);
    std::memcpy(grouping_out, b.c_str(), b.size() + 1);
    return 0;
}

} // extern "C"
