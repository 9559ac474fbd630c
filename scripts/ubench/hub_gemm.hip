// hub_gemm.hip -- lab for vbs_spmm_h16_hub_kernel (sparta_amd/csrc/k_hub16.hip): a DENSE hub of T group tiles (128 rows each) x K columns against a
// column-major B of N columns, bf16.  Builds the slices of A, the step records and the worker ranges itself (what vbs_plan.cpp does for a handle), runs the
// product kernel variants and schedules, checks sampled entries against the integers they must be, prints TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I sparta_amd/csrc scripts/ubench/hub_gemm.hip -o scripts/ubench/hub_gemm
//   hub_gemm T K N variant schedule ranges workers [reps]
//     schedule 0: tile-major step list cut into equal contiguous ranges (stream-K);  1: K-range-major (ranges = number of K ranges), same cut
#include "../../sparta_amd/csrc/k_hub16.hip"
#include <cstdio>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__host__ __device__ inline uint32_t hmix(uint32_t a, uint32_t b) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
// values k / 64, |k| <= 64: exact in bf16 and f16, products and moderate sums exact in fp32 -> the check below is an equality
__host__ __device__ inline float a_val(uint32_t row, uint32_t col) { return (float)((int)(hmix(row, col) & 127u) - 64) / 64.0f; }
__host__ __device__ inline float b_val(uint32_t k, uint32_t n) { return (float)((int)(hmix(k + 0x51u, n * 7919u + 3u) & 127u) - 64) / 64.0f; }
__host__ __device__ inline uint16_t to_bf16(float v) { uint32_t u; memcpy(&u, &v, 4); return (uint16_t)(u >> 16); }   // exact for these values

template <int KP>
__global__ void fill_a(uint16_t* A, int64_t n_el, int S, int G) {        // image: [tile][step][sub-tile][row][chunk position][8]
    constexpr int RB = KP * 2, CPR = RB / 16, R256 = 256 / RB;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_el; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t sl = e / (64 * KP);
        const int r = (int)(e % (64 * KP));
        const int row = r / KP, posc = (r % KP) / 8, kk = r % 8;
        const int c = posc ^ ((row / R256) & (CPR - 1));
        const int u = (int)(sl % G);
        const int64_t ts = sl / G;
        const int s = (int)(ts % S), t = (int)(ts / S);
        A[e] = to_bf16(a_val((uint32_t)((G * t + u) * 64 + row), (uint32_t)(s * KP + 8 * c + kk)));
    }
}
__global__ void fill_b(uint16_t* B, int64_t K, int N, int64_t ldb) {
    const int64_t n_el = K * N;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_el; e += (int64_t)gridDim.x * blockDim.x)
        B[(e % K) + (e / K) * ldb] = to_bf16(b_val((uint32_t)(e % K), (uint32_t)(e / K)));
}
// C[rows of sub-tile][128-column slab] = sum of its images (in order), the layout of SK_SLOT_FLOATS images
__global__ void naive_fix(const int32_t* fix, const int32_t* slots, const float* ws, int64_t ws_slab_stride, float* C, int64_t ldc) {
    const int32_t c_row = fix[4 * blockIdx.x], s0 = fix[4 * blockIdx.x + 2], ns = fix[4 * blockIdx.x + 3];
    const int slab = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 31, g = lane >> 5;
    for (int q = 0; q < 32; q++) {
        float v = 0.0f;
        for (int k = 0; k < ns; k++) v += ws[(int64_t)slab * ws_slab_stride + (int64_t)slots[s0 + k] * SK_SLOT_FLOATS + q * 256 + tid];
        const int row = c_row + 32 * (q >> 4) + lm, col = slab * 128 + 32 * wave + ((q & 15) & 3) + 8 * ((q & 15) >> 2) + 4 * g;
        C[row + (int64_t)col * ldc] = v;
    }
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 16;
    const int64_t K = argc > 2 ? atoll(argv[2]) : 16384;
    const int N = argc > 3 ? atoi(argv[3]) : 256;
    const int variant = argc > 4 ? atoi(argv[4]) : 0;
    const int sched = argc > 5 ? atoi(argv[5]) : 0;
    const int R = argc > 6 ? atoi(argv[6]) : 8;
    const int P = argc > 7 ? atoi(argv[7]) : 256;
    const int reps = argc > 8 ? atoi(argv[8]) : 5;
    const int KP = hub_variant_kp(variant), G = hub_variant_g(variant);
    const int S = (int)(K / KP);
    if (K % KP || N % 256 || P % 8 || S % R) { printf("bad shape\n"); return 1; }
    const int64_t rows = (int64_t)T * 64 * G;
    const int64_t n_a = (int64_t)T * S * G * 64 * KP, pad_a = 8 * G * 64 * KP;
    uint16_t *dA, *dB;
    float *dC, *dWs = nullptr;
    CK(hipMalloc(&dA, (n_a + pad_a) * 2));
    CK(hipMemset(dA, 0, (n_a + pad_a) * 2));
    const int64_t ldb = getenv("HUB_LDB") ? atoll(getenv("HUB_LDB")) : K;      // developer: a leading dimension larger than K (the page pattern of a taller B)
    CK(hipMalloc(&dB, ldb * N * 2));
    CK(hipMalloc(&dC, rows * N * 4));
    if (KP == 64) hipLaunchKernelGGL(fill_a<64>, dim3(4096), dim3(256), 0, 0, dA, n_a, S, G);
    else hipLaunchKernelGGL(fill_a<32>, dim3(4096), dim3(256), 0, 0, dA, n_a, S, G);
    hipLaunchKernelGGL(fill_b, dim3(4096), dim3(256), 0, 0, dB, K, N, ldb);
    CK(hipDeviceSynchronize());

    // developer: every step reads one of 64 panels of B / one of 256 slices of A (cache-hot operands: what do the loads cost when nothing misses?); results wrong
    const bool same_a = getenv("HUB_SAME_A") != nullptr, same_b = getenv("HUB_SAME_B") != nullptr;
    // ---- the step list: (tile, step) in schedule order ----
    std::vector<HubStep> steps;
    std::vector<std::pair<int, int>> order;               // (tile, step)
    order.reserve((size_t)T * S);
    if (sched == 0) { for (int t = 0; t < T; t++) for (int s = 0; s < S; s++) order.emplace_back(t, s); }
    else { const int L = S / R; for (int r = 0; r < R; r++) for (int t = 0; t < T; t++) for (int s = r * L; s < (r + 1) * L; s++) order.emplace_back(t, s); }
    const int64_t U = (int64_t)order.size();
    steps.resize((size_t)U + 32);
    for (int64_t q = 0; q < U + 32; q++) {
        const auto ts = order[(size_t)std::min(q, U - 1)];
        HubStep h;
        const int64_t a_off = same_a ? ((int64_t)(ts.first % 4) * S + ts.second % 64) * G * 64 * KP : ((int64_t)ts.first * S + ts.second) * G * 64 * KP;
        h.a_lo = (uint32_t)a_off; h.a_hi = (uint32_t)(a_off >> 32);
        h.b_row = same_b ? (ts.second % 64) * KP : ts.second * KP; h.shard = 0; h.flags = (1 << G) - 1; h.slot = -1; h.tile = ts.first; h.pad = 0;
        steps[(size_t)q] = h;
    }
    // worker ranges in plan order (the kernel maps workgroup ids to it); equal contiguous cut
    std::vector<int32_t> wr((size_t)P * 2);
    std::vector<int64_t> bnd((size_t)P + 1);
    for (int k = 0; k <= P; k++) bnd[(size_t)k] = U * k / P;
    // segments -> LAST / SPLIT flags, slots, fix records per sub-tile
    std::vector<int32_t> fix, fix_slots;
    std::vector<std::vector<int32_t>> tile_slots((size_t)T);
    int32_t n_slots = 0;
    for (int pos = 0; pos < P; pos++) {
        wr[(size_t)pos * 2] = (int32_t)bnd[(size_t)pos]; wr[(size_t)pos * 2 + 1] = (int32_t)bnd[(size_t)pos + 1];
        int64_t a = bnd[(size_t)pos];
        while (a < bnd[(size_t)pos + 1]) {
            // the run of steps of one tile with consecutive step numbers that starts at a
            int64_t b = a;
            while (b + 1 < bnd[(size_t)pos + 1] && order[(size_t)b + 1].first == order[(size_t)a].first && order[(size_t)b + 1].second == order[(size_t)b].second + 1) b++;
            const bool whole = order[(size_t)a].second == 0 && order[(size_t)b].second == S - 1;
            steps[(size_t)b].flags |= STEP_LAST;
            if (!whole) {
                steps[(size_t)b].flags |= STEP_SPLIT;
                steps[(size_t)b].slot = n_slots;
                tile_slots[(size_t)order[(size_t)a].first].push_back(n_slots);
                n_slots += G;
            }
            a = b + 1;
        }
    }
    for (int t = 0; t < T; t++) {
        if (tile_slots[(size_t)t].empty()) continue;
        for (int u = 0; u < G; u++) {
            fix.push_back((G * t + u) * 64); fix.push_back(64); fix.push_back((int32_t)fix_slots.size()); fix.push_back((int32_t)tile_slots[(size_t)t].size());
            for (int32_t sl : tile_slots[(size_t)t]) fix_slots.push_back(sl + u);
        }
    }
    std::vector<HubTile> tiles((size_t)T);
    for (int t = 0; t < T; t++) { HubTile h{}; for (int u = 0; u < G; u++) { h.c_row[u] = (t * G + u) * 64; h.mt[u] = 64; } tiles[(size_t)t] = h; }
    HubStep* dS; int32_t *dW, *dFix = nullptr, *dFs = nullptr; HubTile* dT;
    CK(hipMalloc(&dS, steps.size() * sizeof(HubStep))); CK(hipMemcpy(dS, steps.data(), steps.size() * sizeof(HubStep), hipMemcpyHostToDevice));
    CK(hipMalloc(&dW, wr.size() * 4)); CK(hipMemcpy(dW, wr.data(), wr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dT, tiles.size() * sizeof(HubTile))); CK(hipMemcpy(dT, tiles.data(), tiles.size() * sizeof(HubTile), hipMemcpyHostToDevice));
    const int64_t slab_stride = (int64_t)std::max(n_slots, 1) * SK_SLOT_FLOATS;
    CK(hipMalloc(&dWs, slab_stride * (N / 128) * 4));
    if (!fix.empty()) {
        CK(hipMalloc(&dFix, fix.size() * 4)); CK(hipMemcpy(dFix, fix.data(), fix.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&dFs, fix_slots.size() * 4)); CK(hipMemcpy(dFs, fix_slots.data(), fix_slots.size() * 4, hipMemcpyHostToDevice));
    }
    HubParams p{};
    p.steps = dS; p.worker_range = dW; p.tiles = dT; p.A = dA; p.B = dB; p.B_tail = nullptr; p.C = dC; p.ws = dWs;
    p.ldb = ldb; p.ldc = rows; p.shard_stride = 0; p.ws_slab_stride = slab_stride; p.accumulate = 0; p.c_row_major = 0; p.c_nt = 1; p.w = 64;
    p.n_slabs = N / 256; p.n_workers = P; p.n_cols = N;

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const bool skip = getenv("HUB_SKIP") != nullptr;          // developer: everything but the product kernel
    auto run = [&]() {
        if (!skip) launch_h16_hub(variant, true, false, 0, p);
        if (!fix.empty()) hipLaunchKernelGGL(naive_fix, dim3((unsigned)(fix.size() / 4), (unsigned)(N / 128)), dim3(256), 0, 0, dFix, dFs, dWs, slab_stride, dC, rows);
    };
    run();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    // ---- check: sampled entries are exact ----
    std::vector<float> hC((size_t)rows * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    double worst = 0.0;
    for (int q = 0; q < 400; q++) {
        const int64_t row = (int64_t)(hmix(q, 17) % (uint32_t)rows);
        const int col = (int)(hmix(q, 91) % (uint32_t)N);
        double ref = 0.0;
        for (int64_t k = 0; k < K; k++) ref += (double)a_val((uint32_t)row, (uint32_t)k) * (double)b_val((uint32_t)k, (uint32_t)col);
        const double got = hC[(size_t)(row + (int64_t)col * rows)];
        const double err = std::fabs(got - ref);
        worst = std::max(worst, err);
        if (err > 1e-3 * std::sqrt((double)K)) { if (bad < 5) printf("  MISMATCH row %lld col %d: got %.6f want %.6f\n", (long long)row, col, got, ref); bad++; }
    }
    // ---- timing ----
    float best = 1e30f, sum = 0.0f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0, 0));
        if (!skip) launch_h16_hub(variant, true, false, 0, p);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms); sum += ms;
    }
    const double flop = 2.0 * (double)rows * (double)K * (double)N;
    printf("T %d (x %d rows) K %lld N %d variant %d (KP %d) sched %d R %d P %d: %s (worst err %.3g), splits %d | best %.3f ms = %.1f TFLOP/s, mean %.3f ms = %.1f TFLOP/s\n", T, 64 * G, (long long)K, N,
           variant, KP, sched, R, P, bad ? "WRONG" : "ok", worst, n_slots / G, best, flop / (best * 1e-3) / 1e12, sum / reps, flop / (sum / reps * 1e-3) / 1e12);
    return bad ? 2 : 0;
}
