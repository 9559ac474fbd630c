// vbs_union.cpp -- host: the device form of the column-compacted ("union-pattern") tiles of an fp32 handle (k_union.hip multiplies them).
//
// In: sparta::UnionPlanHost (vbs_build.cpp, mode 3) -- per tile its rows of C, its column list, its dense values [list position][row] and its tail (a few nonzeros per
// row in columns too thinly used for the list).
// Out: per tile type (tiles of <= 32 rows / of 33..64 rows) ONE sequence of 32-deep steps in execution order.  The `max_workers` persistent workgroups of the ONE launch
// are split between the two types by their MFMA work; inside a type tiles are dealt WHOLE, longest first, each to the worker with the fewest steps so far (LPT: the
// makespan is within one tile of the mean); a worker walks its tiles in matrix order.  Per step: a record (row of C, rows, valid list positions, last-step flag, the
// tile's tail), its 32 list entries and its slice of A in MFMA fragment order -- [rt][j][g][row][4] = A[32 rt + row][k = 16 g + 4 j + e], the image the kernel's
// LDS-direct loads copy verbatim.
#include <algorithm>
#include <cstring>
#include <queue>

#include "vbs_device.hpp"

namespace sparta_dev {

int build_union_plan(const sparta::UnionPlanHost& U, int max_workers, UnionDevPlan& P, int dtype) {
    const bool h16 = dtype != SPARTA_F32, bf16 = dtype == SPARTA_BF16;
    auto steps_of = [&](int ty, size_t t) { return std::max<int64_t>(1, ((int64_t)U.tiles[ty][t].nk + 31) / 32); };   // (a tile without a kept column still stores its rows: one step of zeros)
    // workers per type: in proportion to the MFMA work (a step of a 64-row tile is two of a 32-row one), at least one each, never more than tiles
    double work[2] = {0.0, 0.0};
    for (int ty = 0; ty < 2; ty++)
        for (size_t t = 0; t < U.tiles[ty].size(); t++) work[ty] += (double)steps_of(ty, t) * (ty + 1);
    int W[2] = {0, 0};
    max_workers = std::max(max_workers, 2);
    if (work[0] > 0.0 && work[1] > 0.0) {
        W[1] = (int)std::max<int64_t>(1, std::min<int64_t>(max_workers - 1, (int64_t)(max_workers * work[1] / (work[0] + work[1]) + 0.5)));
        W[0] = max_workers - W[1];
    } else if (work[0] > 0.0) W[0] = max_workers;
    else if (work[1] > 0.0) W[1] = max_workers;
    for (int ty = 0; ty < 2; ty++) W[ty] = (int)std::min<int64_t>(W[ty], (int64_t)U.tiles[ty].size());

    for (int ty = 0; ty < 2; ty++) {
        const int mi = ty + 1;
        const std::vector<sparta::UnionPlanHost::Tile>& T = U.tiles[ty];
        if (T.empty()) continue;
        const int Wt = W[ty];
        std::vector<size_t> order(T.size());
        for (size_t t = 0; t < T.size(); t++) order[t] = t;
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return steps_of(ty, x) > steps_of(ty, y); });
        typedef std::pair<int64_t, int> Load;                                  // (steps so far, worker): the lightest worker on top
        std::priority_queue<Load, std::vector<Load>, std::greater<Load>> pq;
        for (int w = 0; w < Wt; w++) pq.push(Load(0, w));
        std::vector<std::vector<size_t>> mine((size_t)Wt);
        int64_t total = 0, tail_total = 0;
        for (size_t t : order) {
            Load l = pq.top(); pq.pop();
            mine[(size_t)l.second].push_back(t);
            l.first += steps_of(ty, t) + 1; total += steps_of(ty, t);           // (+ 1: the epilogue of a tile costs about a step)
            pq.push(l);
            tail_total += (int64_t)T[t].tail_e * 32 * mi;
        }
        if (total + kUnionPadSteps > INT32_MAX || tail_total > INT32_MAX)
            return sparta::fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_create: too many steps of column-compacted tiles for 32-bit step indices");
        P.n_workers[ty] = Wt; P.n_steps[ty] = total;
        P.rec[ty].assign((size_t)(total + kUnionPadSteps), UnionRec{0, 0, 0, 0});
        P.ids[ty].assign((size_t)(total + kUnionPadSteps) * 32, 0);
        if (h16) P.a16[ty].assign((size_t)(total + kUnionPadSteps) * 1024 * mi, (uint16_t)0);
        else P.a[ty].assign((size_t)(total + kUnionPadSteps) * 1024 * mi, 0.0f);
        P.tail[ty].assign((size_t)tail_total * 2 + 2, 0u);
        P.wrange[ty].assign((size_t)Wt * 2, 0);
        std::vector<int64_t> first((size_t)Wt + 1, 0), tfirst((size_t)Wt + 1, 0);
        for (int w = 0; w < Wt; w++) {
            std::sort(mine[(size_t)w].begin(), mine[(size_t)w].end());        // matrix order inside a worker
            int64_t st = 0, tt = 0;
            for (size_t t : mine[(size_t)w]) { st += steps_of(ty, t); tt += (int64_t)T[t].tail_e * 32 * mi; }
            first[(size_t)w + 1] = first[(size_t)w] + st;
            tfirst[(size_t)w + 1] = tfirst[(size_t)w] + tt;
            P.wrange[ty][(size_t)w * 2] = (int32_t)first[(size_t)w];
            P.wrange[ty][(size_t)w * 2 + 1] = (int32_t)first[(size_t)w + 1];
        }
        sparta::parallel_for_dynamic(Wt, 1, [&](int64_t lo, int64_t hi, int) {
            for (int64_t w = lo; w < hi; w++) {
                int64_t s = first[(size_t)w], to = tfirst[(size_t)w];
                for (size_t t : mine[(size_t)w]) {
                    const sparta::UnionPlanHost::Tile& tl = T[t];
                    const float* img = U.a[ty].data() + U.a_off[ty][t];       // [list position][32 mi rows]
                    const int32_t* cl = U.cols[ty].data() + tl.k0;
                    const int64_t ns = steps_of(ty, t), ldt = 32 * mi;
                    for (int64_t x = 0; x < (int64_t)tl.tail_e * ldt; x++) {
                        uint32_t vb;
                        float tv = U.tail_val[ty][(size_t)(tl.tail0 + x)];
                        if (h16) {                                            // the value the 16-bit handle holds: rounded to the storage type, kept as fp32
                            const uint16_t u = to_h16(tv, bf16);
                            if (bf16) { const uint32_t w32 = (uint32_t)u << 16; std::memcpy(&tv, &w32, 4); }
                            else { _Float16 hh; std::memcpy(&hh, &u, 2); tv = (float)hh; }
                        }
                        std::memcpy(&vb, &tv, 4);
                        P.tail[ty][(size_t)(to + x) * 2] = (uint32_t)U.tail_col[ty][(size_t)(tl.tail0 + x)];
                        P.tail[ty][(size_t)(to + x) * 2 + 1] = vb;
                    }
                    for (int64_t q = 0; q < ns; q++, s++) {
                        const int nvalid = (int)std::min<int64_t>(32, std::max<int64_t>(0, (int64_t)tl.nk - 32 * q));
                        P.rec[ty][(size_t)s] = UnionRec{tl.c_row, tl.mt | (nvalid << 8) | (q == ns - 1 ? UREC_LAST : 0) | (tl.tail_e << UREC_TAIL_SHIFT), (int32_t)to, 0};
                        for (int k = 0; k < nvalid; k++) P.ids[ty][(size_t)s * 32 + (size_t)k] = cl[32 * q + k];
                        if (h16) {
                            uint16_t* d16 = P.a16[ty].data() + (size_t)s * 1024 * (size_t)mi;
                            for (int rt = 0; rt < mi; rt++)
                                for (int k = 0; k < nvalid; k++) {
                                    const int m = k >> 4, kg = (k >> 3) & 1, e = k & 7;
                                    const float* src = img + (32 * q + k) * ldt + 32 * rt;
                                    uint16_t* d = d16 + ((size_t)((rt * 2 + m) * 2 + kg) * 32) * 8 + e;
                                    for (int row = 0; row < 32; row++) d[row * 8] = to_h16(src[row], bf16);
                                }
                            continue;
                        }
                        float* dst = P.a[ty].data() + (size_t)s * 1024 * (size_t)mi;
                        for (int rt = 0; rt < mi; rt++)
                            for (int j = 0; j < 4; j++)
                                for (int g = 0; g < 2; g++)
                                    for (int e = 0; e < 4; e++) {
                                        const int k = 16 * g + 4 * j + e;
                                        if (k >= nvalid) continue;
                                        const float* src = img + (32 * q + k) * ldt + 32 * rt;
                                        float* d = dst + (((rt * 4 + j) * 2 + g) * 32) * 4 + e;
                                        for (int row = 0; row < 32; row++) d[row * 4] = src[row];
                                    }
                    }
                    to += (int64_t)tl.tail_e * ldt;
                }
            }
        });
        for (const sparta::UnionPlanHost::Tile& tl : T) { P.area += (int64_t)tl.nk * tl.mt; P.cols += tl.nk; P.rows += tl.mt; }
    }
    return SPARTA_OK;
}

// the tiles' part of A . x from the DEVICE form (records, list entries, fragment-order slices, tails), the way the kernel indexes them; y[row of C] += ...
void union_plan_host_apply(const UnionDevPlan& P, const float* x, double* y) {      // (fp32 plans)
    for (int ty = 0; ty < 2; ty++) {
        const int mi = ty + 1;
        for (int w = 0; w < P.n_workers[ty]; w++)
            for (int64_t s = P.wrange[ty][(size_t)w * 2]; s < P.wrange[ty][(size_t)w * 2 + 1]; s++) {
                const UnionRec r = P.rec[ty][(size_t)s];
                const int mt = r.info & 127, nvalid = (r.info >> 8) & 63, tail_e = (r.info >> UREC_TAIL_SHIFT) & 31;
                const float* sl = P.a[ty].data() + (size_t)s * 1024 * (size_t)mi;
                for (int rt = 0; rt < mi; rt++)
                    for (int row = 0; row < 32 && 32 * rt + row < mt; row++) {
                        double acc = 0.0;
                        for (int k = 0; k < nvalid; k++) {
                            const int g = k >> 4, j = (k & 15) >> 2, e = k & 3;
                            acc += (double)sl[(((rt * 4 + j) * 2 + g) * 32 + row) * 4 + e] * (double)x[P.ids[ty][(size_t)s * 32 + (size_t)k]];
                        }
                        if (r.info & UREC_LAST)
                            for (int e = 0; e < tail_e; e++) {
                                const size_t at = ((size_t)r.tail_off + (size_t)(e * mi + rt) * 32 + (size_t)row) * 2;
                                float v;
                                std::memcpy(&v, &P.tail[ty][at + 1], 4);
                                acc += (double)v * (double)x[P.tail[ty][at]];
                            }
                        y[(size_t)r.c_row + (size_t)(32 * rt + row)] += acc;
                    }
            }
    }
}

}  // namespace sparta_dev
