// tests/cpp/compat_driver.cpp -- a driver written the way the reference's drivers are (test/general/TEST_blocking_VBR.cpp,
// test/cuda/cuda_multiply.cpp:250-269), compiled against include/sparta_compat.hpp instead of the reference's headers.
// argv[1] = "host" (reorder + build only, no GPU needed) or "gpu" (also multiplies and prints a checksum).
// Prints the README example's results so the test can compare text.
#include <cstdio>
#include <cstring>
#include <vector>
#include <sstream>
#include "sparta_compat.hpp"

// written like the reference's Matrix_Blocking driver (test/general/Matrix_Blocking.cpp): read, block, save statistics + grouping
static int io_mode(const char* path) {
    std::ifstream fin(path);
    CSR cmat(fin, " ", false, el);
    std::printf("read: %ld %ld %ld\n", cmat.rows, cmat.cols, cmat.nztot());
    CLineReader cli;
    cli.filename_ = "data/TEST_matrix_weighted.el"; cli.exp_name_ = "exp0"; cli.tau_ = 0.6f; cli.col_block_size_ = 3; cli.row_block_size_ = 3;
    BlockingEngine bEngine;
    bEngine.tau = cli.tau_; bEngine.col_block_size = cli.col_block_size_; bEngine.row_block_size = cli.row_block_size_;
    bEngine.blocking_algo = (BlockingType)cli.blocking_algo_; bEngine.SetComparator(cli.sim_measure_);
    bEngine.GetGrouping(cmat);
    bEngine.timer_total = 1234.5f; bEngine.timer_merges = 77.25f; bEngine.timer_comparisons = 901.0f;    // pinned clocks (golden fixture)
    bEngine.multiplication_timer_avg = 0.125f; bEngine.multiplication_timer_std = 0.001f;
    std::ostringstream csv, g;
    save_blocking_data(csv, cli, bEngine, cmat, true, g);
    std::string c = csv.str(), gs = g.str();
    for (char& ch : c) if (ch == '\n') ch = '|';
    for (char& ch : gs) if (ch == '\n') ch = ' ';
    std::printf("csv:%s\ngfile:%s\n", c.c_str(), gs.c_str());
    cmat.reorder_by_degree(true);
    std::printf("degrees:");
    for (intT i = 0; i < cmat.rows; i++) std::printf(" %ld", cmat.nzcount[i]);
    std::printf("\n");
    cmat.reorder(bEngine.grouping_result);                 // any grouping of the right length permutes the rows
    std::printf("nnz_after: %ld\n", cmat.nztot());
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 2 && std::strcmp(argv[1], "io") == 0) return io_mode(argv[2]);
    const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    // data/TEST_matrix_weighted.el as the reference's reader parses it (first data line dropped)
    intT rowptr[] = {0, 0, 3, 6, 10, 10, 11, 11, 11, 12};
    intT colidx[] = {2, 5, 8, 5, 6, 8, 1, 3, 7, 8, 6, 1};
    DataT vals[] = {5, 8, 7, 1, 1, 1, 1, 1, 3, 8, 2, 5};
    CSR cmat;
    CSR::from_flat(cmat, 9, 9, rowptr, colidx, vals);

    BlockingEngine bEngine;
    bEngine.tau = 0.6f; bEngine.col_block_size = 3; bEngine.row_block_size = 3;
    bEngine.blocking_algo = iterative_clocked; bEngine.SetComparator(1);
    bEngine.GetGrouping(cmat);
    std::printf("grouping:");
    for (intT g : bEngine.grouping_result) std::printf(" %ld", g);
    std::printf("\ncounters: %ld %ld\n", bEngine.comparison_counter, bEngine.merge_counter);

    VBR vbmat;
    vbmat.fill_from_CSR_inplace(cmat, bEngine.grouping_result, 3);
    std::printf("dims: %ld %ld %ld %ld %ld\n", vbmat.rows, vbmat.cols, vbmat.block_rows, vbmat.block_cols, vbmat.nztot);
    std::printf("jab:");
    for (intT i = 0, n = 0; i < vbmat.block_rows; i++) for (intT k = 0; k < vbmat.nzcount[i]; k++) std::printf(" %ld", vbmat.jab[n++]);
    std::printf("\n");
    bEngine.CollectBlockingInfo(cmat);
    std::printf("info: %ld %ld %ld\n", bEngine.VBR_nzcount, bEngine.VBR_nzblocks_count, bEngine.VBR_longest_row);

    // the extension algorithm through the same class (BlockingType minhash = 7): tau = 0 groups exactly the rows with the same block set
    BlockingEngine lsh;
    lsh.tau = 0.0f; lsh.col_block_size = 3; lsh.blocking_algo = minhash; lsh.minhash_bands = 8;
    lsh.GetGrouping(cmat);
    std::printf("minhash:");
    for (intT g : lsh.grouping_result) std::printf(" %ld", g);
    std::printf("\n");

    if (gpu) {
        std::vector<DataT> B(18);
        for (int i = 0; i < 18; i++) B[(size_t)i] = (DataT)(i + 1);
        std::vector<DataT_C> C(18, 0.0f);
        float dt = 0;
        cublas_blockmat_batched(vbmat, B.data(), 2, C.data(), dt);       // include/cuda_utilities.h:42
        std::printf("C:");
        for (float c : C) std::printf(" %g", c);
        std::printf("\n");
        vbmat.multiply(B.data(), 2, C.data());                           // accumulates: C doubles
        std::printf("C2:");
        for (float c : C) std::printf(" %g", c);
        std::printf("\n");
        // dense x VBS with the reference's call shape (include/cuda_utilities.h:40): D (2 x 9) += E (2 x 9) * A
        std::vector<DataT> E(18);
        for (int i = 0; i < 18; i++) E[(size_t)i] = (DataT)(i % 5) - 2.0f;
        std::vector<DataT_C> D(18, 0.0f);
        cublas_blockmat_multiplyBA(vbmat, E.data(), 2, D.data(), dt);
        std::printf("BA:");
        for (float c : D) std::printf(" %g", c);
        std::printf("\n");
    }
    return 0;
}
