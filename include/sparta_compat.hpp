// sparta_compat.hpp -- the reference's C++ call shapes for the reorder-and-multiply path, implemented on
// the C-ABI of include/sparta_amd.h (header-only; link with -lsparta_amd).
//
// A driver written against the reference (e.g. test/cuda/cuda_multiply.cpp:250-269,
// test/general/TEST_blocking_VBR.cpp) keeps its calls:
//
//     BlockingEngine bEngine; bEngine.tau = ...; bEngine.col_block_size = ...;
//     bEngine.GetGrouping(cmat);                                             // include/blocking.h:47
//     VBR vbmat; vbmat.fill_from_CSR_inplace(cmat, bEngine.grouping_result, col_block_size, row_block_size, force_fixed);
//                                                                            // include/matrices.h:118
//     cublas_blockmat_batched(vbmat, mat_B, B_cols, mat_C, dt);              // include/cuda_utilities.h:42
//     vbmat.multiply(mat_B, B_cols, mat_C);                                  // include/matrices.h:121  (runs on the GPU here)
//
// Same type names, field names, argument order and semantics (C is accumulated into; B, C column-major host
// buffers; `dt` in ms covers the device multiply only).  The file readers, the row permutations, the grouping file and
// save_blocking_data's CSV row are here too (same quirks as the reference's, see SPARTA_IO_COMPAT in sparta_amd.h).
// What is NOT here: CLineReader's getopt parsing (a plain struct with the same public fields stands in) and the CPU
// CSR::multiply (the product has no CPU SpMM).
// Errors: the reference prints and continues; these shims throw std::runtime_error carrying sparta_last_error().
#pragma once
#include <cstring>
#include <fstream>
#include <iterator>
#include <ostream>
#include <stdexcept>
#include <string>
#include <iostream>
#include <vector>

#include "sparta_amd.h"

typedef long int intT;   // include/definitions.h:4
typedef float DataT;     // :5
typedef float DataT_C;   // :6

enum MatrixFormat { el, mtx };                                                                                                      // :15
enum BlockingType { iterative, iterative_structured, fixed_size, iterative_clocked, iterative_queue, iterative_max_size, scramble,   // :17
                    minhash /* = 7, extension: SPARTA_BLOCKING_MINHASH, not in the reference */ };

inline std::vector<intT> get_permutation(const std::vector<intT>& grouping);

namespace sparta_compat_detail {
inline void check(int rc, const char* what) {
    if (rc != SPARTA_OK) throw std::runtime_error(std::string(what) + ": " + sparta_last_error());
}
}  // namespace sparta_compat_detail

// include/matrices.h:10-91 -- one heap array per row, exactly as in the reference
struct CSR {
    intT rows = 0, cols = 0;
    intT* nzcount = nullptr;
    intT** ja = nullptr;
    DataT** ma = nullptr;
    bool pattern_only = false;

    CSR() = default;
    // include/matrices.h:58-63: CSR(infile, delimiter, pattern_only, mat_fmt) -- reads what is left in the stream
    CSR(std::ifstream& infile, std::string delimiter = " ", bool pattern_only = true, MatrixFormat mat_fmt = mtx) {
        read_from_edgelist(infile, delimiter, pattern_only, mat_fmt);
    }
    CSR(const CSR&) = delete;
    CSR& operator=(const CSR&) = delete;
    ~CSR() { clean(); }

    void clean() {
        if (ja) for (intT i = 0; i < rows; i++) delete[] ja[i];
        if (ma) for (intT i = 0; i < rows; i++) delete[] ma[i];
        delete[] ja; delete[] ma; delete[] nzcount;
        ja = nullptr; ma = nullptr; nzcount = nullptr; rows = cols = 0;
    }
    intT nztot() const { intT n = 0; for (intT i = 0; i < rows; i++) n += nzcount[i]; return n; }

    // src/general/csr.cpp:183-365, reference behaviour (SPARTA_IO_COMPAT): .el drops its first line, .mtx is pattern-only
    void read_from_edgelist(std::ifstream& infile, std::string delimiter = " ", bool pattern_only = true, MatrixFormat mat_fmt = mtx,
                            bool symmetrize = false) {
        const std::string text((std::istreambuf_iterator<char>(infile)), std::istreambuf_iterator<char>());
        sparta_csr_host h;
        sparta_compat_detail::check(sparta_csr_read_buffer(text.data(), (int64_t)text.size(), delimiter.c_str(), pattern_only, (int32_t)mat_fmt,
                                                           symmetrize, SPARTA_IO_COMPAT, &h), "CSR::read_from_edgelist");
        std::vector<intT> rp(h.rowptr, h.rowptr + h.rows + 1), ci(h.colidx, h.colidx + h.nnz);
        from_flat(*this, h.rows, h.cols, rp.data(), ci.data(), h.pattern_only ? nullptr : h.vals);
        sparta_csr_host_free(&h);
    }
    // src/general/csr.cpp:169-179
    void save_to_edgelist(std::ofstream& outfile, std::string delimiter = " ", bool pattern_only = true, MatrixFormat mat_fmt = el) const {
        (void)pattern_only;
        for (intT i = 0; i < rows; i++)
            for (intT nz = 0; nz < nzcount[i]; nz++) {
                if (mat_fmt == mtx) outfile << ja[i][nz] << delimiter << i << "\n";
                else outfile << i << delimiter << ja[i][nz] << "\n";
            }
    }
    // src/general/csr.cpp:67-76 (utilities.h:95-107): new row k = old row permutation[k]
    void permute_rows(std::vector<intT> permutation) {
        if ((intT)permutation.size() != rows) throw std::invalid_argument("CSR.permute_rows argument bust have same lenght as rows");
        std::vector<intT*> nja((size_t)rows);
        std::vector<DataT*> nma((size_t)rows);
        std::vector<intT> nnz((size_t)rows);
        for (intT k = 0; k < rows; k++) {
            nja[(size_t)k] = ja[permutation[(size_t)k]];
            if (!pattern_only) nma[(size_t)k] = ma[permutation[(size_t)k]];
            nnz[(size_t)k] = nzcount[permutation[(size_t)k]];
        }
        for (intT k = 0; k < rows; k++) {
            ja[k] = nja[(size_t)k];
            if (!pattern_only) ma[k] = nma[(size_t)k];
            nzcount[k] = nnz[(size_t)k];
        }
    }
    // src/general/csr.cpp:101-109
    void reorder(std::vector<intT> grouping) {
        if ((intT)grouping.size() != rows) throw std::invalid_argument("CSR.reorder argument bust have same lenght as rows");
        permute_rows(get_permutation(grouping));
    }
    // src/general/csr.cpp:123-155
    void reorder_by_degree(bool descending = true) {
        std::vector<int64_t> rp((size_t)rows + 1, 0), perm((size_t)rows);
        for (intT i = 0; i < rows; i++) rp[(size_t)i + 1] = rp[(size_t)i] + nzcount[i];
        sparta_compat_detail::check(sparta_degree_permutation(rows, rp.data(), descending, perm.data()), "CSR::reorder_by_degree");
        permute_rows(std::vector<intT>(perm.begin(), perm.end()));
    }

    // build from flat CSR arrays (vals == nullptr -> pattern_only)
    static void from_flat(CSR& out, intT rows, intT cols, const intT* rowptr, const intT* colidx, const DataT* vals) {
        out.clean();
        out.rows = rows; out.cols = cols; out.pattern_only = (vals == nullptr);
        out.nzcount = new intT[rows];
        out.ja = new intT*[rows];
        out.ma = vals ? new DataT*[rows] : nullptr;
        for (intT i = 0; i < rows; i++) {
            intT n = rowptr[i + 1] - rowptr[i];
            out.nzcount[i] = n;
            out.ja[i] = new intT[n];
            std::memcpy(out.ja[i], colidx + rowptr[i], sizeof(intT) * (size_t)n);
            if (vals) { out.ma[i] = new DataT[n]; std::memcpy(out.ma[i], vals + rowptr[i], sizeof(DataT) * (size_t)n); }
        }
    }
    // flatten for the C-ABI (rowptr int64, colidx int32)
    void to_flat(std::vector<int64_t>& rowptr, std::vector<int32_t>& colidx, std::vector<float>& vals) const {
        rowptr.assign((size_t)rows + 1, 0);
        for (intT i = 0; i < rows; i++) rowptr[(size_t)i + 1] = rowptr[(size_t)i] + nzcount[i];
        colidx.resize((size_t)rowptr[(size_t)rows]);
        vals.clear();
        if (!pattern_only) vals.resize(colidx.size());
        for (intT i = 0; i < rows; i++)
            for (intT k = 0; k < nzcount[i]; k++) {
                colidx[(size_t)(rowptr[(size_t)i] + k)] = (int32_t)ja[i][k];
                if (!pattern_only) vals[(size_t)(rowptr[(size_t)i] + k)] = ma[i][k];
            }
    }
};

// src/general/utilities.cpp:8-54,145-173 and src/general/blocking.cpp:859-994 -- same signatures
inline std::vector<intT> get_permutation(const std::vector<intT>& grouping) {
    std::vector<intT> p(grouping.size());
    sparta_compat_detail::check(sparta_get_permutation((const int64_t*)grouping.data(), (int64_t)grouping.size(), (int64_t*)p.data()), "get_permutation");
    return p;
}
inline std::vector<intT> get_partition(const std::vector<intT>& grouping) {
    std::vector<intT> p(grouping.size() + 1);
    int64_t n = 0;
    sparta_compat_detail::check(sparta_get_partition((const int64_t*)grouping.data(), (int64_t)grouping.size(), (int64_t*)p.data(), &n), "get_partition");
    p.resize((size_t)n);
    return p;
}
inline std::vector<intT> get_fixed_size_grouping(const std::vector<intT>& grouping, intT row_block_size) {
    std::vector<intT> g(grouping.size());
    sparta_compat_detail::check(sparta_get_fixed_size_grouping((const int64_t*)grouping.data(), (int64_t)grouping.size(), row_block_size, (int64_t*)g.data()), "get_fixed_size_grouping");
    return g;
}
inline std::vector<intT> merge_rows(std::vector<intT> A, intT* B, intT sizeB) {
    std::vector<intT> out(A.size() + (size_t)sizeB + 1);
    int64_t n = 0;
    sparta_compat_detail::check(sparta_merge_rows((const int64_t*)A.data(), (int64_t)A.size(), (const int64_t*)B, sizeB, (int64_t*)out.data(), &n), "merge_rows");
    out.resize((size_t)n);
    return out;
}
inline float HammingDistanceGroup(std::vector<intT> row_A, intT group_size_A, intT* row_B, intT size_B, intT group_size_B, intT block_size) {
    float d = 0;
    sparta_compat_detail::check(sparta_row_distance(SPARTA_SIM_HAMMING, (const int64_t*)row_A.data(), (int64_t)row_A.size(), group_size_A, (const int64_t*)row_B, size_B, group_size_B, block_size, &d), "HammingDistanceGroup");
    return d;
}
inline float JaccardDistanceGroup(std::vector<intT> row_A, intT group_size_A, intT* row_B, intT size_B, intT group_size_B, intT block_size) {
    float d = 0;
    sparta_compat_detail::check(sparta_row_distance(SPARTA_SIM_JACCARD, (const int64_t*)row_A.data(), (int64_t)row_A.size(), group_size_A, (const int64_t*)row_B, size_B, group_size_B, block_size, &d), "JaccardDistanceGroup");
    return d;
}

// include/blocking.h:9-56
class BlockingEngine {
   public:
    float tau = 0.5;
    intT col_block_size = 1;
    intT row_block_size = 1;
    bool use_groups = false;
    bool use_pattern = true;
    bool force_fixed_size = false;
    int structured_m = 2, structured_n = 4;
    int minhash_bands = 0, minhash_rows = 0, minhash_max_eval = 0, minhash_max_rows = 0;   // blocking_algo == minhash only; 0 = library default
    BlockingType blocking_algo = iterative_clocked;

    intT comparison_counter = 0, merge_counter = 0;
    float timer_total = 0, timer_comparisons = 0, timer_merges = 0;
    float average_row_distance = 0, average_merge_tau = 0;
    float multiplication_timer_avg = 0, multiplication_timer_std = 0;
    intT VBR_nzcount = 0, VBR_nzblocks_count = 0;
    float VBR_average_height = 0;
    intT VBR_longest_row = 0;
    std::vector<intT> grouping_result;

    void SetComparator(int choice) { sim_measure_ = choice; }   // src/general/blocking.cpp:699-717

    std::vector<intT> GetGrouping(const CSR& cmat) {             // src/general/blocking.cpp:633-676
        std::vector<int64_t> rp; std::vector<int32_t> ci; std::vector<float> v;
        cmat.to_flat(rp, ci, v);
        sparta_reorder_cfg cfg;
        sparta_reorder_cfg_default(&cfg);
        cfg.blocking_algo = (int32_t)blocking_algo; cfg.sim_measure = sim_measure_; cfg.tau = tau;
        cfg.use_groups = use_groups; cfg.col_block_size = col_block_size; cfg.row_block_size = row_block_size;
        cfg.use_pattern = use_pattern; cfg.force_fixed_size = force_fixed_size;
        cfg.structured_m = structured_m; cfg.structured_n = structured_n;
        cfg.minhash_bands = minhash_bands; cfg.minhash_rows = minhash_rows; cfg.minhash_max_eval = minhash_max_eval; cfg.minhash_max_rows = minhash_max_rows;
        sparta_reorder_stats st;
        grouping_result.assign((size_t)cmat.rows, 0);
        sparta_compat_detail::check(sparta_reorder(cmat.rows, cmat.cols, rp.data(), ci.data(), &cfg, (int64_t*)grouping_result.data(), &st), "GetGrouping");
        comparison_counter = st.comparison_counter; merge_counter = st.merge_counter;
        timer_total = st.timer_total; timer_comparisons = st.timer_comparisons; timer_merges = st.timer_merges;
        average_row_distance = st.average_row_distance; average_merge_tau = st.average_merge_tau;
        return grouping_result;
    }
    void CollectBlockingInfo(const CSR& cmat) {                  // src/general/blocking.cpp:576-631
        std::vector<int64_t> rp; std::vector<int32_t> ci; std::vector<float> v;
        cmat.to_flat(rp, ci, v);
        int64_t info[3]; float avg = 0;
        sparta_compat_detail::check(sparta_blocking_info(cmat.rows, cmat.cols, rp.data(), ci.data(), (const int64_t*)grouping_result.data(), col_block_size, info, &avg), "CollectBlockingInfo");
        VBR_nzcount = info[0]; VBR_nzblocks_count = info[1];
        if (info[2] > VBR_longest_row) VBR_longest_row = info[2];
        VBR_average_height = avg;
    }

   private:
    int sim_measure_ = 1;
};

// include/input.h:12-45: the fields save_blocking_data prints (no getopt here: fill them from your own argument parser)
struct CLineReader {
    std::string filename_ = "data/TEST_matrix_weighted.el", outfile_ = "results/TEST_results.txt", exp_name_ = "", reader_delimiter_ = " ";
    int mat_fmt_ = 0;
    bool sim_use_groups_ = 0, sim_use_pattern_ = 1, pattern_only_ = 0, force_fixed_size = 0, symmetrize_ = false;
    int blocking_algo_ = 3, seed_ = 0, sim_measure_ = 1, reorder_ = 0, col_block_size_ = 3, row_block_size_ = 3;
    float tau_ = 0.1f;
    int verbose_ = 1, multiplication_algo_ = 0, B_cols_ = 1024, warmup_ = 1, exp_repetitions_ = 5, n_streams_ = 4;
};

// src/general/utilities.cpp:175-245: one header line + one value line (32 columns) to `outfile`, the grouping to `blocking_outfile`
inline void save_blocking_data(std::ostream& outfile, CLineReader& cLine, BlockingEngine& bEngine, CSR& cmat, bool save_blocking,
                               std::ostream& blocking_outfile) {
    bEngine.CollectBlockingInfo(cmat);
    sparta_csv_fields f;
    f.matrix = cLine.filename_.c_str(); f.rows = cmat.rows; f.cols = cmat.cols; f.nonzeros = cmat.nztot();
    f.symmetrize = cLine.symmetrize_; f.blocking_algo = cLine.blocking_algo_; f.tau = cLine.tau_;
    f.row_block_size = cLine.row_block_size_; f.col_block_size = cLine.col_block_size_; f.use_pattern = cLine.sim_use_pattern_;
    f.sim_use_groups = cLine.sim_use_groups_; f.sim_measure = cLine.sim_measure_; f.reorder = cLine.reorder_;
    f.exp_name = cLine.exp_name_.c_str(); f.b_cols = cLine.B_cols_; f.warmup = cLine.warmup_; f.exp_repetitions = cLine.exp_repetitions_;
    f.multiplication_algo = cLine.multiplication_algo_; f.n_streams = cLine.n_streams_;
    f.time_to_block = bEngine.timer_total; f.time_to_merge = bEngine.timer_merges; f.time_to_compare = bEngine.timer_comparisons;
    f.vbr_nzcount = bEngine.VBR_nzcount; f.vbr_nzblocks_count = bEngine.VBR_nzblocks_count;
    f.vbr_average_height = bEngine.VBR_average_height; f.vbr_longest_row = bEngine.VBR_longest_row;
    f.merge_counter = bEngine.merge_counter; f.comparison_counter = bEngine.comparison_counter;
    f.average_merge_tau = bEngine.average_merge_tau; f.average_row_distance = bEngine.average_row_distance;
    f.avg_time_multiply = bEngine.multiplication_timer_avg; f.std_time_multiply = bEngine.multiplication_timer_std;
    char header[2048], values[4096];
    sparta_compat_detail::check(sparta_blocking_csv_row(&f, header, sizeof header, values, sizeof values), "save_blocking_data");
    outfile << header << std::endl;
    outfile << values << std::endl;
    if (save_blocking)
        for (intT i = 0; i < cmat.rows; i++) blocking_outfile << bEngine.grouping_result[(size_t)i] << "\n";
}

// test/general/Matrix_Analysis.cpp:10-32 (the caller drops a leading count: `if (size == rows + 1) erase(begin)`, :78)
inline std::vector<intT> read_grouping_file(const std::string& filename) {
    std::vector<int64_t> buf(1 << 16);
    int64_t n = 0;
    int rc = sparta_grouping_read(filename.c_str(), -1, buf.data(), (int64_t)buf.size(), &n);
    if (rc != SPARTA_OK && n > (int64_t)buf.size()) {
        buf.resize((size_t)n);
        rc = sparta_grouping_read(filename.c_str(), -1, buf.data(), (int64_t)buf.size(), &n);
    }
    if (rc != SPARTA_OK) return std::vector<intT>();             // the reference prints a message and returns an empty vector
    return std::vector<intT>(buf.begin(), buf.begin() + n);
}

// include/matrices.h:93-125
struct VBR {
    intT rows = 0, cols = 0, block_rows = 0, block_cols = 0;
    intT* nzcount = nullptr;
    intT* jab = nullptr;
    intT* row_part = nullptr;
    DataT* mab = nullptr;
    intT block_col_size = 0;
    intT nztot = 0;

    VBR() = default;
    VBR(const VBR&) = delete;
    VBR& operator=(const VBR&) = delete;
    ~VBR() { clean(); }

    void clean() {
        if (dev_) { sparta_vbs_destroy(dev_); dev_ = nullptr; }
        if (dev_t_) { sparta_vbs_destroy(dev_t_); dev_t_ = nullptr; }
        delete[] nzcount; delete[] jab; delete[] row_part; delete[] mab;
        nzcount = jab = row_part = nullptr; mab = nullptr;
        rows = cols = block_rows = block_cols = nztot = block_col_size = 0;
    }

    // src/general/vbr.cpp:135-237
    void fill_from_CSR_inplace(const CSR& cmat, const std::vector<intT>& grouping, intT col_block_size, intT row_block_size = 0,
                               bool force_fixed_size = false) {
        clean();
        std::vector<int64_t> rp; std::vector<int32_t> ci; std::vector<float> v;
        cmat.to_flat(rp, ci, v);
        sparta_vbs_host h;
        sparta_compat_detail::check(sparta_vbs_build(cmat.rows, cmat.cols, rp.data(), ci.data(), cmat.pattern_only ? nullptr : v.data(),
                                                     (const int64_t*)grouping.data(), col_block_size, row_block_size, force_fixed_size, &h),
                                    "VBR::fill_from_CSR_inplace");
        take(h);
        sparta_vbs_host_free(&h);
    }
    // src/general/vbr.cpp:121-132
    void fill_from_CSR_inplace(const CSR& cmat, intT row_block_size, intT col_block_size, bool force_fixed_size = false) {
        std::vector<intT> grouping((size_t)cmat.rows);
        for (intT i = 0; i < cmat.rows; i++) grouping[(size_t)i] = i / row_block_size;
        fill_from_CSR_inplace(cmat, grouping, col_block_size, row_block_size, force_fixed_size);
    }

    // src/general/vbr.cpp:239-321 -- rows keep their order, block-rows from a row partition
    void fill_from_CSR(const CSR& cmat, const std::vector<intT>& row_partition, intT block_size) {
        clean();
        std::vector<int64_t> rp; std::vector<int32_t> ci; std::vector<float> v;
        cmat.to_flat(rp, ci, v);
        sparta_vbs_host h;
        sparta_compat_detail::check(sparta_vbs_build_partition(cmat.rows, cmat.cols, rp.data(), ci.data(), cmat.pattern_only ? nullptr : v.data(),
                                                               (const int64_t*)row_partition.data(), (int64_t)row_partition.size(), block_size, &h),
                                    "VBR::fill_from_CSR");
        take(h);
        sparta_vbs_host_free(&h);
    }
    // src/general/vbr.cpp:33-49 -- where the values of block-row `row_block_idx` start in mab
    DataT* get_block_start(intT row_block_idx) {
        if (row_block_idx >= block_rows) { std::cerr << " [RANGE ERROR] VBR::get_block_start" << std::endl; return mab; }
        DataT* ptr = mab;
        for (intT ib = 0; ib < row_block_idx; ib++) ptr += (row_part[ib + 1] - row_part[ib]) * nzcount[ib] * block_col_size;
        return ptr;
    }
    // src/general/vbr.cpp:108-118 -- 0 = valid partition of [0, rows]
    int partition_check(const std::vector<intT>& candidate_part) {
        return sparta_vbs_partition_check((const int64_t*)candidate_part.data(), (int64_t)candidate_part.size(), rows);
    }

    // include/matrices.h:121 -- C += A*B, host buffers, column-major; runs the MFMA kernels on `device`
    void multiply(DataT* B, int B_cols, DataT_C* C, float* dt = nullptr, int device = 0) const {
        if (!dev_)
            sparta_compat_detail::check(sparta_vbs_create(&dev_, rows, cols, block_rows, block_col_size, (const int64_t*)row_part,
                                                          (const int64_t*)nzcount, (const int64_t*)jab, mab, SPARTA_F32, device), "VBR::multiply (upload)");
        sparta_compat_detail::check(sparta_vbs_spmm(dev_, B, cols, SPARTA_COL_MAJOR, B_cols, C, rows, SPARTA_COL_MAJOR, 1, SPARTA_PTR_HOST, nullptr,
                                                    SPARTA_SPMM_MFMA, dt), "VBR::multiply");
    }

    // C += B * A (dense x VBS): B is B_rows x rows, C is B_rows x cols, column-major (ld = B_rows).  Call shape of the reference's
    // cublas_blockmat_multiplyBA (include/cuda_utilities.h:40) -- whose own arithmetic is not a B * A (src/cuda/cuda_utilities.cpp:649-663).
    void multiply_BA(DataT* B, int B_rows, DataT_C* C, float* dt = nullptr, int device = 0) const {
        if (!dev_t_)
            sparta_compat_detail::check(sparta_vbs_create_transposed(&dev_t_, rows, cols, block_rows, block_col_size, (const int64_t*)row_part,
                                                                     (const int64_t*)nzcount, (const int64_t*)jab, mab, SPARTA_F32, device), "VBR::multiply_BA (upload)");
        sparta_compat_detail::check(sparta_vbs_spmm_ba(dev_t_, B, B_rows, B_rows, C, B_rows, 1, SPARTA_PTR_HOST, nullptr, dt), "VBR::multiply_BA");
    }

   private:
    void take(const sparta_vbs_host& h) {      // the library's arrays -> this object's own (new[] / delete[] as in the reference)
        rows = h.rows; cols = h.cols; block_rows = h.block_rows; block_cols = h.block_cols; block_col_size = h.block_col_size; nztot = h.nztot;
        row_part = new intT[h.block_rows + 1]; std::memcpy(row_part, h.row_part, sizeof(intT) * (size_t)(h.block_rows + 1));
        nzcount = new intT[h.block_rows > 0 ? h.block_rows : 1]; std::memcpy(nzcount, h.nzcount, sizeof(intT) * (size_t)h.block_rows);
        jab = new intT[h.nblocks > 0 ? h.nblocks : 1]; std::memcpy(jab, h.jab, sizeof(intT) * (size_t)h.nblocks);
        mab = new DataT[h.nztot > 0 ? h.nztot : 1]; std::memcpy(mab, h.mab, sizeof(DataT) * (size_t)h.nztot);
    }
    mutable sparta_vbs_t* dev_ = nullptr;   // device image, created on first multiply, released by clean()
    mutable sparta_vbs_t* dev_t_ = nullptr; // device image of the transpose (multiply_BA)
};

// include/cuda_utilities.h:38,42 and include/cutlass_bellpack_lib.h:21,25 -- one fused kernel family behind all of them.
// n_streams is accepted for source compatibility and ignored (there is no per-block launch to spread over streams).
inline void cublas_fixed_blocks_multiply(const VBR& vbmatA, DataT* B, int B_cols, DataT_C* C, float& dt, int n_streams = 4) { (void)n_streams; vbmatA.multiply(B, B_cols, C, &dt); }
inline void cublas_blockmat_batched(const VBR& vbmatA, DataT* B, int B_cols, DataT_C* C, float& dt) { vbmatA.multiply(B, B_cols, C, &dt); }
// include/cuda_utilities.h:40 -- same call shape; computes C (B_rows x A.cols) += B (B_rows x A.rows) * A, which the reference's body does not (see VBR::multiply_BA)
inline void cublas_blockmat_multiplyBA(const VBR& vbmatA, DataT* B, int B_rows, DataT_C* C, float& dt, int n_streams = 4) { (void)n_streams; vbmatA.multiply_BA(B, B_rows, C, &dt); }
inline void cutlas_fixed_blocks_multiply(const VBR& vbmatA, DataT* B, int B_cols, DataT_C* C, float& dt) { vbmatA.multiply(B, B_cols, C, &dt); }
inline void cutlas_blockmat_batched(const VBR& vbmatA, DataT* B, int B_cols, DataT_C* C, float& dt) { vbmatA.multiply(B, B_cols, C, &dt); }
