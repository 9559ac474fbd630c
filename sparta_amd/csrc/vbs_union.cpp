// vbs_union.cpp -- host: the device form of the column-compacted ("union-pattern") tiles of a handle (k_union.hip multiplies them).
//
// In: sparta::UnionPlanHost (vbs_build.cpp, mode 3) -- per tile its rows of C, its column list, its dense values [list position][row] and its tail (a few nonzeros per
// row in columns too thinly used for the list).
// Out: per tile TYPE one sequence of 32-deep steps in execution order.  The type is the tile's height in MFMA row tiles: fp32 handles multiply with the 16 x 16 x 4
// instruction, so a tile of mt rows is of type ceil(mt / 16) - 1 (16, 32, 48, 64 rows: a 48-row cluster costs three row tiles, not four); 16-bit handles with the
// 32 x 32 x 16 one (types 0, 1: <= 32, 33..64 rows).  The `max_workers` persistent workgroups of the ONE launch each walk tiles of every type (the kernel runs one
// body per type, tallest first): tiles are dealt WHOLE, costliest first, to the least loaded CU (its workgroups b, b + CUs, ...) and there to the least loaded worker;
// where whole tiles leave some CUs a tile above the others, a few tiles are cut into pieces of one row tile each (same list) and dealt again.  Per step: a record (row of C, rows, valid list positions, last-step flag, the tile's tail), its 32 list entries and its slice of A as the
// LDS image the kernel's LDS-direct loads copy verbatim (UnionSide::A in vbs_device.hpp).
#include <algorithm>
#include <cstring>
#include <queue>

#include "vbs_device.hpp"

namespace sparta_dev {

int build_union_plan(const sparta::UnionPlanHost& U, int max_workers, UnionDevPlan& P, int dtype, int n_cus) {
    const bool h16 = dtype != SPARTA_F32, bf16 = dtype == SPARTA_BF16;
    const int gran = h16 ? 32 : 16;                                            // rows per MFMA row tile
    // a tile of the device plan: a tile of the host plan (its host type: 0 <= 32 rows, 1 33..64, and index) or a PIECE of one -- rows [row0, row0 + nrows) of it, same list
    struct Ref { int hty; size_t t; int row0, nrows; };
    auto tile_of = [&](const Ref& r) -> const sparta::UnionPlanHost::Tile& { return U.tiles[r.hty][r.t]; };
    auto steps_of = [&](const Ref& r) { return std::max<int64_t>(1, ((int64_t)tile_of(r).nk + 31) / 32); };   // (a tile without a kept column still stores its rows: one step of zeros)
    auto type_of = [&](const Ref& r) { return std::min(kUnionTypes - 1, (std::max(1, r.nrows) + gran - 1) / gran - 1); };
    // A step's cost: its row tiles on the matrix pipe + what every step pays (the gather of the panel, the barrier); + a step for a tile's epilogue.
    auto step_cost = [&](int ty) { return h16 ? 2.0 + (ty + 1) : 0.5 + (ty + 1); };
    auto cost_of = [&](const Ref& r) { return (double)(steps_of(r) + 1) * step_cost(type_of(r)); };
    std::vector<Ref> refs;
    for (int hty = 0; hty < 2; hty++)
        for (size_t t = 0; t < U.tiles[hty].size(); t++) refs.push_back(Ref{hty, t, 0, std::max<int>(1, U.tiles[hty][t].mt)});

    // ONE set of workers for all types: every tile is dealt WHOLE, costliest first, to the least loaded CU and there to its least loaded worker (LPT).  The workgroups
    // b, b + CUs, b + 2 CUs of a launch of 3 x CUs persistent workgroups run on ONE CU (observed on gfx950, scripts/ubench/wg_place.hip: 256 of 256 trios in every launch;
    // for speed only), so what must be even is the work of such a trio, not of a worker.  n_cus = 0 (or a worker count that is not a multiple of it): every worker its own bin.
    const int W = (int)std::min<int64_t>(std::max(max_workers, 1), (int64_t)refs.size());
    const int bins = (n_cus > 0 && W % n_cus == 0) ? n_cus : W, per_bin = W / bins;
    std::vector<int> owner;                                                    // worker of refs[i]
    auto lpt = [&](const std::vector<Ref>& items, std::vector<int>& own, std::vector<double>& bin_load) {
        std::vector<size_t> order(items.size());
        std::vector<double> cost(items.size());
        for (size_t i = 0; i < items.size(); i++) { order[i] = i; cost[i] = cost_of(items[i]); }
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return cost[x] > cost[y]; });
        typedef std::pair<double, int> Load;                                   // (work so far, bin): the lightest bin on top
        std::priority_queue<Load, std::vector<Load>, std::greater<Load>> pq;
        for (int bI = 0; bI < bins; bI++) pq.push(Load(0.0, bI));
        std::vector<double> wl((size_t)W, 0.0);
        own.assign(items.size(), 0);
        bin_load.assign((size_t)bins, 0.0);
        for (size_t i : order) {
            Load l = pq.top(); pq.pop();
            int w = l.second;
            for (int k = 1; k < per_bin; k++) if (wl[(size_t)(l.second + k * bins)] < wl[(size_t)w]) w = l.second + k * bins;
            own[i] = w; wl[(size_t)w] += cost[i];
            l.first += cost[i]; bin_load[(size_t)l.second] = l.first;
            pq.push(l);
        }
        double mk = 0.0;
        for (double x : bin_load) mk = std::max(mk, x);
        return mk;
    };
    std::vector<double> bin_load;
    double makespan = lpt(refs, owner, bin_load);
    // Whole tiles quantise the makespan (2063 equal tiles on 256 CUs: 9 on fifteen of them, 8 on the others -- 12 % above the mean).  Where that costs more than 4 %, the
    // cheapest tile of every overloaded bin is cut into pieces of one MFMA row tile each (same list: the pieces gather the same rows of B again, so only a few tiles are
    // cut) and everything is dealt again; kept if the makespan falls.  SPARTA_UNION_SPLIT=0: never; =2: cut whenever a bin is above the mean (tests).
    const int split_mode = [] { const char* e = std::getenv("SPARTA_UNION_SPLIT"); return e ? atoi(e) : 1; }();
    if (split_mode != 0 && !refs.empty()) {
        double total = 0.0;
        for (double x : bin_load) total += x;
        const double mean = total / bins;
        if (makespan > (split_mode == 2 ? 1.0 : 1.04) * mean) {
            std::vector<uint8_t> cut(refs.size(), 0);
            std::vector<int> cheapest((size_t)bins, -1);
            for (size_t i = 0; i < refs.size(); i++) {
                const int bI = owner[i] % bins;
                if (bin_load[(size_t)bI] <= (split_mode == 2 ? 1.0 : 1.02) * mean || refs[i].nrows <= gran) continue;
                if (cheapest[(size_t)bI] < 0 || cost_of(refs[i]) < cost_of(refs[(size_t)cheapest[(size_t)bI]])) cheapest[(size_t)bI] = (int)i;
            }
            size_t n_cut = 0;
            const size_t cap = std::max<size_t>(1, refs.size() / 8);
            for (int bI = 0; bI < bins && n_cut < cap; bI++)
                if (cheapest[(size_t)bI] >= 0) { cut[(size_t)cheapest[(size_t)bI]] = 1; n_cut++; }
            if (n_cut > 0) {
                std::vector<Ref> pieces;
                for (size_t i = 0; i < refs.size(); i++) {
                    if (!cut[i]) { pieces.push_back(refs[i]); continue; }
                    for (int r0 = 0; r0 < refs[i].nrows; r0 += gran) pieces.push_back(Ref{refs[i].hty, refs[i].t, r0, std::min(gran, refs[i].nrows - r0)});
                }
                std::vector<int> own2;
                std::vector<double> bl2;
                const double mk2 = lpt(pieces, own2, bl2);
                if (mk2 < (split_mode == 2 ? 1.0 : 0.985) * makespan || split_mode == 2) { refs.swap(pieces); owner.swap(own2); makespan = mk2; }
            }
        }
    }
    std::vector<Ref> of_type[kUnionTypes];
    std::vector<std::vector<size_t>> mine_all[kUnionTypes];
    for (int ty = 0; ty < kUnionTypes; ty++) mine_all[ty].resize((size_t)W);
    for (size_t i = 0; i < refs.size(); i++) {
        const int ty = type_of(refs[i]);
        mine_all[ty][(size_t)owner[i]].push_back(of_type[ty].size());
        of_type[ty].push_back(refs[i]);
    }

    for (int ty = 0; ty < kUnionTypes; ty++) {
        const int R = gran * (ty + 1), nrt = ty + 1;                           // rows of the type's slices and tails; row tiles
        P.type_rows[ty] = R;
        const std::vector<Ref>& T = of_type[ty];
        if (T.empty()) continue;
        const int Wt = W;
        std::vector<std::vector<size_t>>& mine = mine_all[ty];
        int64_t total = 0, tail_total = 0;
        for (const Ref& r : T) { total += steps_of(r); tail_total += (int64_t)tile_of(r).tail_e * R; }
        if (total + kUnionPadSteps > INT32_MAX || tail_total > INT32_MAX)
            return sparta::fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_create: too many steps of column-compacted tiles for 32-bit step indices");
        P.n_workers[ty] = Wt; P.n_steps[ty] = total; P.n_tiles[ty] = (int64_t)T.size();
        // elements of a step's slice: its rows of A x 32 k, and -- fp32 plans -- one more KB behind them: the (column, value) pairs of the tile's tail entry the step requests
        // ([rt][row of the row tile] uint2; the kernel gets them with the slice, one LDS-direct load per workgroup, instead of one vector load per wave and row tile)
        const size_t slice = (size_t)R * 32 + (h16 ? 0 : (size_t)kUnionPairFloats);
        P.rec[ty].assign((size_t)(total + kUnionPadSteps), UnionRec{0, 0, 0, 0});
        P.ids[ty].assign((size_t)(total + kUnionPadSteps) * 32, 0);
        if (h16) P.a16[ty].assign((size_t)(total + kUnionPadSteps) * slice, (uint16_t)0);
        else P.a[ty].assign((size_t)(total + kUnionPadSteps) * slice, 0.0f);
        P.tail[ty].assign((size_t)tail_total * 2 + 2, 0u);
        P.wrange[ty].assign((size_t)Wt * 2, 0);
        std::vector<int64_t> first((size_t)Wt + 1, 0), tfirst((size_t)Wt + 1, 0);
        for (int w = 0; w < Wt; w++) {
            std::sort(mine[(size_t)w].begin(), mine[(size_t)w].end(),
                      [&](size_t x, size_t y) { return tile_of(T[x]).c_row + T[x].row0 < tile_of(T[y]).c_row + T[y].row0; });   // matrix order inside a worker
            int64_t st = 0, tt = 0;
            for (size_t t : mine[(size_t)w]) { st += steps_of(T[t]); tt += (int64_t)tile_of(T[t]).tail_e * R; }
            first[(size_t)w + 1] = first[(size_t)w] + st;
            tfirst[(size_t)w + 1] = tfirst[(size_t)w] + tt;
            P.wrange[ty][(size_t)w * 2] = (int32_t)first[(size_t)w];
            P.wrange[ty][(size_t)w * 2 + 1] = (int32_t)first[(size_t)w + 1];
        }
        sparta::parallel_for_dynamic(Wt, 1, [&](int64_t lo, int64_t hi, int) {
            for (int64_t w = lo; w < hi; w++) {
                int64_t s = first[(size_t)w], to = tfirst[(size_t)w];
                for (size_t t : mine[(size_t)w]) {
                    const Ref& ref = T[t];
                    const sparta::UnionPlanHost::Tile& tl = tile_of(ref);
                    const int64_t ldt = 32 * (ref.hty + 1);                   // rows of the host image and of the host tail per entry
                    const float* img = U.a[ref.hty].data() + U.a_off[ref.hty][ref.t];          // [list position][ldt rows]
                    const int32_t* cl = U.cols[ref.hty].data() + tl.k0;
                    const int64_t ns = steps_of(ref);
                    const int rows_here = ref.nrows;                           // rows of the host image this tile (or piece) holds: [row0, row0 + nrows)
                    for (int e = 0; e < tl.tail_e; e++)
                        for (int row = 0; row < R; row++) {
                            uint32_t vb;
                            int32_t tc = 0;
                            float tv = 0.0f;
                            if (row < rows_here) { tc = U.tail_col[ref.hty][(size_t)(tl.tail0 + e * ldt + ref.row0 + row)]; tv = U.tail_val[ref.hty][(size_t)(tl.tail0 + e * ldt + ref.row0 + row)]; }
                            else tc = U.tail_col[ref.hty][(size_t)tl.tail0];  // (a row behind the tile's: any valid row of B, value 0)
                            if (h16) {                                        // the value the 16-bit handle holds: rounded to the storage type, kept as fp32
                                const uint16_t u = to_h16(tv, bf16);
                                if (bf16) { const uint32_t w32 = (uint32_t)u << 16; std::memcpy(&tv, &w32, 4); }
                                else { _Float16 hh; std::memcpy(&hh, &u, 2); tv = (float)hh; }
                            }
                            std::memcpy(&vb, &tv, 4);
                            P.tail[ty][(size_t)(to + (int64_t)e * R + row) * 2] = (uint32_t)tc;
                            P.tail[ty][(size_t)(to + (int64_t)e * R + row) * 2 + 1] = vb;
                        }
                    for (int64_t q = 0; q < ns; q++, s++) {
                        const int nvalid = (int)std::min<int64_t>(32, std::max<int64_t>(0, (int64_t)tl.nk - 32 * q));
                        P.rec[ty][(size_t)s] = UnionRec{tl.c_row + ref.row0, ref.nrows | (nvalid << 8) | (q == ns - 1 ? UREC_LAST : 0) | (tl.tail_e << UREC_TAIL_SHIFT), (int32_t)to, 0};
                        for (int k = 0; k < 32; k++) P.ids[ty][(size_t)s * 32 + (size_t)k] = k < nvalid ? cl[32 * q + k] : (tl.nk > 0 ? cl[0] : 0);   // (behind the list: its first column -- fetched, against zeros of A)
                        if (h16) {
                            uint16_t* d16 = P.a16[ty].data() + (size_t)s * slice;
                            for (int rt = 0; rt < nrt; rt++)
                                for (int k = 0; k < nvalid; k++) {
                                    const int m = k >> 4, kg = (k >> 3) & 1, e = k & 7;
                                    const float* src = img + (32 * q + k) * ldt + ref.row0 + 32 * rt;
                                    uint16_t* d = d16 + ((size_t)((rt * 2 + m) * 2 + kg) * 32) * 8 + e;
                                    for (int row = 0; row < 32 && 32 * rt + row < rows_here; row++) d[row * 8] = to_h16(src[row], bf16);
                                }
                            continue;
                        }
                        // fp32: [rt][h][lane = 16 kq + i][e] = A[16 rt + i][k = 4 (4 h + e) + kq]
                        float* dst = P.a[ty].data() + (size_t)s * slice;
                        if (q < tl.tail_e && q + 1 < ns) {                    // tail entry q rides in the tile's step q (requested there, multiplied in by step q + 1): its pairs
                            uint32_t* pr = reinterpret_cast<uint32_t*>(dst + (size_t)R * 32);
                            for (int row = 0; row < R; row++) {
                                const size_t at = (size_t)(to + q * R + row) * 2;      // (the tile's tail in the plan's layout, written above)
                                pr[2 * row] = P.tail[ty][at]; pr[2 * row + 1] = P.tail[ty][at + 1];
                            }
                        }
                        for (int k = 0; k < nvalid; k++) {
                            const int kq = k & 3, sub = k >> 2, h = sub >> 2, e = sub & 3;
                            const float* src = img + (32 * q + k) * ldt + ref.row0;
                            for (int row = 0; row < rows_here; row++) dst[(((row >> 4) * 2 + h) * 64 + 16 * kq + (row & 15)) * 4 + e] = src[row];
                        }
                    }
                    to += (int64_t)tl.tail_e * R;
                }
            }
        });
        for (const Ref& r : T) {                                               // (a cut tile counts once as a tile, its pieces' steps and list entries each)
            const sparta::UnionPlanHost::Tile& tl = tile_of(r);
            P.area += (int64_t)tl.nk * r.nrows; P.cols += tl.nk; P.rows += r.nrows;
            P.tiles_by_height[tl.mt > 32] += r.row0 == 0; P.steps_by_height[tl.mt > 32] += steps_of(r);
        }
    }
    return SPARTA_OK;
}

// the tiles' part of A . x from the DEVICE form (records, list entries, LDS-image slices, tails), the way the kernel indexes them; y[row of C] += ...
void union_plan_host_apply(const UnionDevPlan& P, const float* x, double* y) {      // (fp32 plans)
    for (int ty = 0; ty < kUnionTypes; ty++) {
        const int R = P.type_rows[ty];
        for (int w = 0; w < P.n_workers[ty]; w++)
            for (int64_t s = P.wrange[ty][(size_t)w * 2]; s < P.wrange[ty][(size_t)w * 2 + 1]; s++) {
                const UnionRec r = P.rec[ty][(size_t)s];
                const int mt = r.info & 127, nvalid = (r.info >> 8) & 63, tail_e = (r.info >> UREC_TAIL_SHIFT) & 31;
                const float* sl = P.a[ty].data() + (size_t)s * ((size_t)R * 32 + (size_t)kUnionPairFloats);
                for (int row = 0; row < mt && row < R; row++) {
                    double acc = 0.0;
                    for (int k = 0; k < nvalid; k++) {
                        const int kq = k & 3, sub = k >> 2, h = sub >> 2, e = sub & 3;
                        acc += (double)sl[(((row >> 4) * 2 + h) * 64 + 16 * kq + (row & 15)) * 4 + e] * (double)x[P.ids[ty][(size_t)s * 32 + (size_t)k]];
                    }
                    if (r.info & UREC_LAST)
                        for (int e = 0; e < tail_e; e++) {
                            const size_t at = ((size_t)r.tail_off + (size_t)e * (size_t)R + (size_t)row) * 2;
                            float v;
                            std::memcpy(&v, &P.tail[ty][at + 1], 4);
                            acc += (double)v * (double)x[P.tail[ty][at]];
                        }
                    y[(size_t)r.c_row + (size_t)row] += acc;
                }
            }
    }
}

}  // namespace sparta_dev
