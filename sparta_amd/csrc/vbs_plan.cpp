// vbs_plan.cpp -- host side of the stream kernels: the plan.  Flattens the row tiles of a VBS (or of a block-row range of it)
// into step lists, cuts them into worker ranges (aligned / split), records the split tiles for the fix-up and, for 16-bit
// handles, packs A into per-step slices.  Plain host C++ (no kernel, no HIP call).  See DESIGN.md section 3.2 (2).
#include <cstdio>
#include <queue>

#include "vbs_device.hpp"

namespace sparta_dev {

// fp32 -> fp16 / bf16 bits, round to nearest even (what the device conversion kernel does too)
uint16_t to_h16(float v, bool bf16) {
    if (bf16) {
        uint32_t u;
        std::memcpy(&u, &v, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
        return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    const _Float16 h = (_Float16)v;
    uint16_t o;
    std::memcpy(&o, &h, 2);
    return o;
}

// (Round 2 also built a "pair plan" + vbs_spmm_f32_pair_kernel -- two vertically adjacent 32-row tiles walked as one 64-row pair that shares
// the B panel of a step -- and a 64-deep variant of the one-tile kernel.  Both measured slower than the kernels that stayed (62.3 and 65.6 us
// against 58.1 on the flagship) and were removed once the no-barrier kernel (k_f32_direct.hip) took over; they are in
// the history of this file and of k_f32_stream.hip.)

int build_stream_plans(const StreamPlanIn& in, StreamPlanHost& P) {
    using sparta::fail;
    const int64_t cols = in.cols, w = in.w, br0 = in.br0, br1 = in.br1, jab_lo = in.jab_lo, mab_lo = in.mab_lo;
    const int64_t* row_part = in.row_part; const int64_t* nzcount = in.nzcount; const int64_t* jab = in.jab; const float* mab = in.mab;
    const int32_t dtype = in.dtype, device = in.device;
    const bool h16 = dtype != SPARTA_F32;
    std::vector<StepRec>(&steps)[2] = P.steps;
    std::vector<int32_t>(&wrange)[2] = P.wrange;
    std::vector<FixRec>& fix = P.fix;
    std::vector<int32_t>& fix_slots = P.fix_slots;
    std::vector<uint16_t>(&a16_steps)[2] = P.a16_steps;  // A as per-step slices in step order, per type (packed once the type's plan is final)
    int& n_workers = P.n_workers; int& n_split = P.n_split;
    int(&plan_aligned)[2] = P.plan_aligned;
    // ---- stream plans (persistent kernels): flatten tiles into 32-deep steps, cut into equal-cost worker ranges ----
    // One plan per tile TYPE: ty = 1 tiles of 33..64 rows (two 32-row MFMA tiles per wave and step), ty = 0 tiles of
    // <= 32 rows (one).  Each type runs in its own launch of a kernel instantiated for that type only.  A single kernel
    // that picks the variant per step looks equivalent but compiles badly: at every join of the two variants the register
    // allocator reconciles the in-flight A/B registers and the accumulators with v_mov behind s_waitcnt vmcnt(0) / the
    // MFMA drain, which collapses the 3-step prefetch (measured: 72 non-MFMA VALU per step, 69 % of the matrix peak).
    // k depth of a step: 32 for fp32; the 16-bit kernels take 64 when the block width allows (their steps are short: fewer, fatter)
    // (Measured and dropped: 16-bit plans of 32-wide blocks walked 64 deep -- two blocks in block columns (2 c, 2 c + 1) as one step, whole
    // 128-byte lines of B, 36 % fewer steps on the flagship -- 23.4 us against 23.1: these kernels are not bound by steps or half lines.)
    const int64_t kp = !h16 ? SK_KP : (w % 64 == 0 ? 64 : 32);
    P.kp = kp;
    if (w % SK_KP == 0) {
        hipDeviceProp_t prop;
        int cus = 256;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        // workers (persistent workgroups) per CU: fp32 two -- the MFMAs of one hide the loads of the other (one: 60.8 us against 54 on the flagship);
        // 16-bit ONE -- these kernels are bound by the L2 / HBM side, a second workgroup per CU only adds cache pressure (flagship 23.4 -> 23.1 us, 64 x 64
        // blocks 23.8 -> 23.1, N = 256 46.7 -> 42.9, banded 32.2 -> 30.6; three: slower still) and 256 workers balance better than 512
        int per_cu = h16 ? 1 : 2;
        if (const char* e = std::getenv("SPARTA_WORKERS_PER_CU")) per_cu = std::max(1, std::min(3, atoi(e)));
        n_workers = ((cus * per_cu + 7) / 8) * 8;
        // modelled cost of a step and of a tile's epilogue, per type
        int c2 = 20, c1 = 13, ct = 6;
        if (h16) { c2 = 12; c1 = 10; }                   // load-bound steps: cost ~ bytes moved, (64 + 128) vs (32 + 128) rows and columns
        if (const char* e = std::getenv("SPARTA_COST_MODEL")) sscanf(e, "%d,%d,%d", &c2, &c1, &ct);
        int64_t split_penalty = 120;                     // cost units (~0.11 us each) the fix-up launch adds to a split plan
        if (const char* e = std::getenv("SPARTA_SPLIT_PENALTY")) split_penalty = atoll(e);
        bool interleave = false;
        // whole-tile plans: 0 = contiguous ranges of tiles, 1 = an XCD's tiles dealt in matrix order to its least loaded worker (measured:
        // +-2 %), 2 (default) = longest tile first: flagship 52.3 -> 49.7 us (3-4 tiles of 4..14 steps per worker: makespan 1.10 -> 1.05 x mean)
        int interleave_mode = 2;
        if (const char* e = std::getenv("SPARTA_STREAM_INTERLEAVE")) interleave_mode = atoi(e);
        interleave = interleave_mode != 0;
        double slot_bias = 0.0;                          // SPARTA_SLOT_BIAS: extra share of the workgroup dispatched first onto a CU
        if (const char* e = std::getenv("SPARTA_SLOT_BIAS")) slot_bias = std::max(-0.9, std::min(0.9, atof(e)));
        int align_mode = -1;                             // SPARTA_STREAM_ALIGN=0 always split, 1 never split, unset: cheaper one
        if (const char* e = std::getenv("SPARTA_STREAM_ALIGN")) align_mode = atoi(e) ? 1 : 0;
        if ((int64_t)cols > INT32_MAX)
            return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_create: matrix too large for 32-bit step indexing");
        // ---- 16-bit handles of 32-wide blocks: PAIR TILES.  Two vertically adjacent block-rows (the upper one exactly 32 rows tall) are walked as ONE 64-row tile
        // over the union of their block columns: the B panel of a step feeds both halves (a third fewer steps and panels on a FEM matrix), and a block one of them
        // lacks is a half of the A slice that is stored as zeros and NOT FETCHED (the kernel gives that half a descriptor of zero records: loads return zeros
        // without touching memory).  Round 2 measured the merge with the zero halves fetched (+29 % bytes of A: 23.0 -> 22.5 us) and dropped it.
        // SPARTA_H16_PAIR=0 switches it off.
        struct PairSrc { int64_t a_off_hi; int32_t h_hi, rows_hi, lo_present, hi_present; };
        std::vector<PairSrc> pair_tab(1);                                 // [0] unused: StepRec::pad carries the index while the plan is built
        std::vector<uint8_t> pair_role((size_t)(br1 - br0), 0);          // 1 = first of a pair, 2 = second
        std::vector<int64_t> jo_of((size_t)(br1 - br0) + 1, 0), mo_of((size_t)(br1 - br0) + 1, 0);
        for (int64_t ib = br0; ib < br1; ib++) {
            jo_of[(size_t)(ib - br0) + 1] = jo_of[(size_t)(ib - br0)] + nzcount[ib];
            mo_of[(size_t)(ib - br0) + 1] = mo_of[(size_t)(ib - br0)] + nzcount[ib] * (row_part[ib + 1] - row_part[ib]) * w;
        }
        const bool pair_on = h16 && [] { const char* e = std::getenv("SPARTA_H16_PAIR"); return !e || atoi(e) != 0; }();
        if (pair_on) {
            for (int64_t ib = br0; ib + 1 < br1; ib++) {
                const int64_t h1 = row_part[ib + 1] - row_part[ib], h2 = row_part[ib + 2] - row_part[ib + 1];
                const bool s1 = in.skip && in.skip[ib - br0], s2 = in.skip && in.skip[ib + 1 - br0];
                if (!s1 && !s2 && h1 == 32 && h2 >= 1 && h2 <= 32 && nzcount[ib] > 0 && nzcount[ib + 1] > 0) {
                    pair_role[(size_t)(ib - br0)] = 1; pair_role[(size_t)(ib + 1 - br0)] = 2;
                    ib++;
                }
            }
        }
        // ---- HUB PLAN, part 1: which block-rows leave the 64-row plan for the GEMM-shaped kernel (k_hub16.hip), and in which groups --------------------------
        // 16-bit handles of 64-wide blocks.  Candidates: block-rows of 33..64 rows with at least hub_min steps (the dense hub of a power-law matrix under the
        // fixed 64 x 64 grid: 10^3..10^4 blocks per block-row).  A workgroup of that kernel multiplies G sub-tiles against ONE panel of B per step, walking the union
        // of their block columns, so the members of a group should own (nearly) the same block columns: candidates are sorted by length and a seed takes, among the
        // next `window` candidates, the ones whose block-column sets are most similar to its own (Jaccard, the reference's own measure -- one level up: blocks
        // instead of nonzeros); they need NOT be neighbours (in an R-MAT graph the block columns of a block-row follow the popcount of its index, not its position).
        // SPARTA_HUB=0: off.  SPARTA_HUB_G = 2 | 4 sub-tiles per group; SPARTA_HUB_MIN_STEPS (per tile), SPARTA_HUB_MIN_TOTAL (all groups); SPARTA_HUB_TAU (least
        // similarity, default 0.5); SPARTA_HUB_RANGES (K ranges of the step order, part 2).
        std::vector<uint8_t> hub_role((size_t)(br1 - br0), 0);
        struct HubGroup { int64_t ib[kHubGMax]; int n; };
        std::vector<HubGroup> hub_groups;
        int hub_G = 4;
        {
            const char* e0 = std::getenv("SPARTA_HUB");
            const bool hub_on = h16 && kp == 64 && !(e0 && atoi(e0) == 0) && [] { const char* e = std::getenv("SPARTA_H16_PATH"); return !(e && e[0] == 'l'); }();
            if (const char* e = std::getenv("SPARTA_HUB_G")) hub_G = atoi(e) == 2 ? 2 : 4;
            int64_t hub_min = 64;
            if (const char* e = std::getenv("SPARTA_HUB_MIN_STEPS")) hub_min = std::max<int64_t>(1, atoll(e));
            double tau = 0.5;
            if (const char* e = std::getenv("SPARTA_HUB_TAU")) tau = atof(e);
            const int window = 48;
            if (hub_on) {
                std::vector<int64_t> cand;
                for (int64_t ib = br0; ib < br1; ib++) {
                    const int64_t h = row_part[ib + 1] - row_part[ib], nb = nzcount[ib];
                    if ((in.skip && in.skip[ib - br0]) || pair_role[(size_t)(ib - br0)] != 0) continue;
                    if (h > 32 && h <= 64 && nb * (w / 64) >= hub_min) cand.push_back(ib);
                }
                std::stable_sort(cand.begin(), cand.end(), [&](int64_t a, int64_t b) { return nzcount[a] > nzcount[b]; });
                auto jaccard = [&](int64_t a, int64_t b) {
                    const int64_t* pa = jab + jab_lo + jo_of[(size_t)(a - br0)]; const int64_t na = nzcount[a];
                    const int64_t* pb = jab + jab_lo + jo_of[(size_t)(b - br0)]; const int64_t nb2 = nzcount[b];
                    int64_t i = 0, j = 0, both = 0;
                    while (i < na && j < nb2) { if (pa[i] == pb[j]) { both++; i++; j++; } else if (pa[i] < pb[j]) i++; else j++; }
                    return (double)both / (double)(na + nb2 - both);
                };
                std::vector<uint8_t> used(cand.size(), 0);
                int64_t hub_steps_total = 0;
                for (size_t q = 0; q < cand.size(); q++) {
                    if (used[q]) continue;
                    used[q] = 1;
                    HubGroup g{{cand[q], 0, 0, 0}, 1};
                    std::vector<std::pair<double, size_t>> sim;
                    int seen = 0;
                    for (size_t r = q + 1; r < cand.size() && seen < window; r++) {
                        if (used[r]) continue;
                        seen++;
                        if ((double)nzcount[cand[r]] < tau * (double)nzcount[cand[q]]) break;          // (sorted by length: nothing similar enough further on)
                        const double js = jaccard(cand[q], cand[r]);
                        if (js >= tau) sim.emplace_back(-js, r);
                    }
                    std::sort(sim.begin(), sim.end());
                    for (size_t k = 0; k < sim.size() && g.n < hub_G; k++) { g.ib[g.n++] = cand[sim[k].second]; used[sim[k].second] = 1; }
                    if (g.n < 2) continue;                              // a tile on its own stays in the 64-row plan
                    std::sort(g.ib, g.ib + g.n);
                    hub_groups.push_back(g);
                    hub_steps_total += nzcount[cand[q]] * (w / 64);
                }
                // not worth a launch of its own below a few steps per worker (SPARTA_HUB_MIN_TOTAL overrides: tests)
                int64_t hub_min_total = 8 * (int64_t)n_workers;
                if (const char* e = std::getenv("SPARTA_HUB_MIN_TOTAL")) hub_min_total = atoll(e);
                if (hub_steps_total < hub_min_total) hub_groups.clear();
                for (const HubGroup& g : hub_groups) for (int u = 0; u < g.n; u++) hub_role[(size_t)(g.ib[u] - br0)] = 1;
            }
        }
        for (int ty = 0; ty < 2; ty++) {
            std::vector<StepRec>& st = steps[ty];
            struct TileSpan { int64_t first, last; int32_t c_row, mt; };     // step range of a tile
            std::vector<TileSpan> spans;
            std::vector<int64_t> cum;                                        // cumulative cost BEFORE step s
            int64_t total_cost = 0;
            {
                int64_t jo2 = 0, mo2 = 0;
                const int64_t row0 = row_part[br0];
                // (Measured and dropped, 16-bit handles of 32-wide blocks: two vertically adjacent block-rows walked as ONE 64-row tile over the union of
                // their block columns, a block one of them lacks being a zero half of the A slice -- the B panel of a step feeds both: -34 % steps and
                // panel traffic, +29 % bytes of A; flagship 23.0 -> 22.5 us.  Not worth the second copy of the tile loop.)
                for (int64_t ib = br0; ib < br1; ib++) {
                    const int64_t h = row_part[ib + 1] - row_part[ib];
                    const int64_t nb = nzcount[ib];
                    const bool skipped = in.skip && in.skip[ib - br0];
#ifdef SPARTA_TIMELINE                              // developer build only (make timeline): a probe makes the products WRONG on purpose
                    const int dbg_probe = [] { const char* e = std::getenv("SPARTA_DBG_PROBE"); return e ? atoi(e) : 0; }();
#else
                    constexpr int dbg_probe = 0;
#endif
                    const bool zero_range = nb == 0 && !skipped && h >= kZeroRangeRows;     // one streamed fill instead of h / 64 fix-up tiles
                    if (zero_range && ty == 0) P.zero_ranges.emplace_back(row_part[ib] - row0, h);
                    const uint8_t role = pair_role[(size_t)(ib - br0)];
                    if (role == 1 && ty == 1) {                              // the pair (ib, ib + 1) as one 64-row tile
                        const int64_t h2 = row_part[ib + 2] - row_part[ib + 1], nb2 = nzcount[ib + 1];
                        const int64_t jo_b = jo_of[(size_t)(ib + 1 - br0)], mo_b = mo_of[(size_t)(ib + 1 - br0)];
                        const int32_t c_row = (int32_t)(row_part[ib] - row0), mt = (int32_t)(32 + h2);
                        TileSpan sp{(int64_t)st.size(), 0, c_row, mt};
                        int64_t a = 0, b2 = 0;
                        while (a < nb || b2 < nb2) {
                            const int64_t ja = a < nb ? jab[jab_lo + jo2 + a] : INT64_MAX, jb2 = b2 < nb2 ? jab[jab_lo + jo_b + b2] : INT64_MAX;
                            const int64_t jb = std::min(ja, jb2);
                            const bool lo = ja == jb, hi = jb2 == jb;
                            for (int64_t ks = 0; ks < w; ks += kp) {
                                StepRec r;
                                r.a_off = lo ? mo2 + (a * w + ks) * h : 0;
                                r.b_row = (int32_t)(jb * w + ks);
                                r.h = (int32_t)h;
                                r.c_row = c_row;
                                r.mt_flags = mt | (lo ? 0 : STEP_LO_ABSENT) | (hi ? 0 : STEP_HI_ABSENT);
                                if ((jb + 1) * w > cols) { r.mt_flags |= STEP_TAIL; r.b_row = (int32_t)ks; }
                                r.slot = -1;
                                r.pad = (int32_t)pair_tab.size();
                                pair_tab.push_back(PairSrc{hi ? mo_b + (b2 * w + ks) * h2 : 0, (int32_t)h2, (int32_t)h2, lo ? 1 : 0, hi ? 1 : 0});
                                cum.push_back(total_cost);
                                total_cost += c2;
                                st.push_back(r);
                            }
                            a += lo; b2 += hi;
                        }
                        total_cost += ct;
                        sp.last = (int64_t)st.size() - 1;
                        st[(size_t)sp.first].mt_flags |= STEP_FIRST;
                        st[(size_t)sp.last].mt_flags |= STEP_LAST;
                        spans.push_back(sp);
                    }
                    if (role != 0 || hub_role[(size_t)(ib - br0)]) { jo2 += nb; mo2 += nb * h * w; continue; }   // block-rows of a pair / of the hub plan: no tiles of their own
                    for (int64_t r0 = 0; r0 < h && !skipped && !zero_range; r0 += SK_TM) {
                        const int32_t mt = (int32_t)std::min<int64_t>(SK_TM, h - r0);
                        if ((mt > 32 ? 1 : 0) != ty) continue;
                        const int32_t c_row = (int32_t)(row_part[ib] - row0 + r0);
                        if (nb == 0) {                                      // nothing to multiply: the fix-up kernel writes the zeros
                            fix.push_back(FixRec{c_row, mt, 0, 0});
                            continue;
                        }
                        TileSpan sp{(int64_t)st.size(), 0, c_row, mt};
                        for (int64_t b = 0; b < nb; b++) {
                            const int64_t jb = jab[jab_lo + jo2 + b];
                            for (int64_t ks = 0; ks < w; ks += kp) {
                                StepRec r;
                                r.a_off = mo2 + r0 + (b * w + ks) * h;
                                r.b_row = (int32_t)(jb * w + ks);
                                r.h = (int32_t)h;
                                r.c_row = c_row;
                                r.mt_flags = mt;
                                if ((jb + 1) * w > cols) { r.mt_flags |= STEP_TAIL; r.b_row = (int32_t)ks; }   // read from the zero-padded B_tail
                                r.slot = -1;
                                r.pad = 0;
                                if (dbg_probe) {                            // developer probe (timing only, results are wrong): which stream bounds a step
                                    if (dbg_probe & 1) r.b_row = (int32_t)ks;                                   // every panel of B is the same (cache-hot) one
                                    if ((dbg_probe & 2) && h16) r.a_off = (ks) * h;                                  // every tile reads the first block of A
                                    if ((dbg_probe & 2) && !h16) r.a_off = r0 + (b * w + ks) * h;
                                    if (dbg_probe & 4) r.c_row = 0;                                             // every tile writes the first rows of C
                                }
                                cum.push_back(total_cost);
                                total_cost += ty ? c2 : c1;
                                st.push_back(r);
                            }
                        }
                        total_cost += ct;
                        sp.last = (int64_t)st.size() - 1;
                        st[(size_t)sp.first].mt_flags |= STEP_FIRST;
                        st[(size_t)sp.last].mt_flags |= STEP_LAST;
                        spans.push_back(sp);
                    }
                    jo2 += nb;
                    mo2 += nb * h * w;
                }
            }
            if ((int64_t)st.size() > INT32_MAX - 64)
                return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_create: matrix too large for 32-bit step indexing");
            const int64_t S = (int64_t)st.size();
            P.n_plan_tiles[ty] = (int64_t)spans.size();
            for (const TileSpan& sp : spans) if (sp.c_row % 32 != 0) P.tiles_row_aligned[ty] = false;
            if (S == 0) continue;
            // 16-bit one-tile plans of 32-wide blocks: TWO sub-workers per workgroup (wave pairs (0, 1) and (2, 3), each wave 64 columns of the slab:
            // vbs_spmm_h16_direct_kernel, WC = 64) unless the plan is a ring plan (short misaligned tiles, see below): 2 x n_workers ranges, sub-worker
            // s of workgroup g at wrange[2 (2 g + s)].  OFF by default (SPARTA_H16_WIDE=1 selects it): N = 128 gains 0.7 % (flagship 21.3 -> 21.1 us), but the
            // 256-column slabs that N % 256 == 0 launches take (vbs_capi.cpp: A read once per 256 columns, N = 256: 42.2 -> 35.7 us) want ONE range per workgroup.
            const bool ring_like = ty == 0 && !P.tiles_row_aligned[ty] && S < 6 * (int64_t)spans.size() &&
                                   [] { const char* e = std::getenv("SPARTA_CSTAGE"); return !e || atoi(e) != 0; }();
            const bool wide = h16 && kp == 32 && ty == 0 && !ring_like && [] { const char* e = std::getenv("SPARTA_H16_WIDE"); return e && atoi(e) != 0; }() &&
                              [] { const char* e = std::getenv("SPARTA_H16_PATH"); return !(e && e[0] == 'l'); }();     // (the LDS-staged kernel walks one range per workgroup)
            if (ty == 0) P.wide16 = wide;
            const int n_workers = wide ? 2 * P.n_workers : P.n_workers;      // (shadows the handle's count inside this plan)
            cum.push_back(total_cost);
            // boundaries: worker (x, j) = the j-th of the P/8 sub-ranges of XCD x's eighth; workgroup id = x + 8 j
            const int per_x = n_workers / 8;
            std::vector<int64_t> bnd((size_t)n_workers + 1, S);
            bnd[0] = 0;
            // ---- WINDOW PLAN (16-bit handles, 64-row tiles; EXPERIMENT, SPARTA_TILE_WINDOW_COLS = columns per window, default off) --------------------------
            // The B panels a worker streams are the kernel's L2 misses (rocprofv3 FETCH_SIZE on a part of the 8 M-row R-MAT: 30 GB per launch = 6.5 x |B|): with
            // contiguous step ranges every worker sits at its own k position of its own tile.  Here a tile is cut into PIECES where its block columns cross into
            // another window of `win` columns (a piece holds at least min_steps steps), the pieces are taken window by window - (window, tile) order - and dealt
            // whole to the least loaded worker, so that at any time all workers walk the same window from its first block column on: the first one to touch a
            // panel takes the miss, the others of its XCD hit.  Every piece of a tile with more than one piece is a partial image for the fix-up, which adds a
            // tile's images in k order whatever worker produced them.
            int64_t win_cols = 0;
            if (const char* e = std::getenv("SPARTA_TILE_WINDOW_COLS")) win_cols = std::max(0, atoi(e));
            const bool win_plan = h16 && ty == 1 && !wide && win_cols >= w && S >= 4 * (int64_t)n_workers;
            std::vector<int32_t> sp_tile, sp_seq, tile_pieces;               // per span (= piece): its tile and place in it; per tile: pieces
            std::vector<TileSpan> tile_spans;                                // the tiles as they were (c_row, mt)
            if (win_plan) {
                int64_t min_steps = 64;
                if (const char* e = std::getenv("SPARTA_TILE_WINDOW_MIN")) min_steps = std::max(1, atoi(e));
                struct Piece { int64_t first, last, win; int32_t tile, seq; };
                std::vector<Piece> pieces;
                auto win_of = [&](int64_t q) { const StepRec& r = st[(size_t)q]; return ((r.mt_flags & STEP_TAIL) ? cols - 1 : (int64_t)r.b_row) / win_cols; };
                tile_spans = spans;
                tile_pieces.assign(spans.size(), 0);
                for (size_t t = 0; t < spans.size(); t++) {
                    int64_t a = spans[t].first;
                    int32_t seq = 0;
                    while (a <= spans[t].last) {
                        int64_t b = a, wcur = win_of(a);
                        while (b < spans[t].last) {
                            const int64_t wn = win_of(b + 1);
                            if (wn != wcur) { if (b - a + 1 >= min_steps) break; wcur = wn; }
                            b++;
                        }
                        if (spans[t].last - b < min_steps) b = spans[t].last;           // no short tail piece
                        pieces.push_back(Piece{a, b, win_of(a), (int32_t)t, seq++});
                        a = b + 1;
                    }
                    tile_pieces[t] = seq;
                }
                std::stable_sort(pieces.begin(), pieces.end(), [](const Piece& x, const Piece& y) { return x.win < y.win; });       // (window, tile)
                // deal: whole pieces, in that order, to the least loaded worker
                std::vector<std::vector<size_t>> mine((size_t)n_workers);
                {
                    typedef std::pair<int64_t, int> LW;                      // (load, worker)
                    std::priority_queue<LW, std::vector<LW>, std::greater<LW>> heap;
                    for (int pos = 0; pos < n_workers; pos++) heap.push(LW(0, pos));
                    for (size_t i = 0; i < pieces.size(); i++) {
                        LW lw = heap.top();
                        heap.pop();
                        mine[(size_t)lw.second].push_back(i);
                        lw.first += (pieces[i].last - pieces[i].first + 1) * (int64_t)c2 + ct;
                        heap.push(lw);
                    }
                }
                std::vector<StepRec> ns;
                std::vector<TileSpan> nspans;
                ns.reserve(st.size());
                for (int pos = 0; pos < n_workers; pos++) {
                    bnd[(size_t)pos] = (int64_t)ns.size();
                    for (size_t i : mine[(size_t)pos]) {
                        const Piece& pc = pieces[i];
                        TileSpan sp = tile_spans[(size_t)pc.tile];
                        sp.first = (int64_t)ns.size();
                        ns.insert(ns.end(), st.begin() + pc.first, st.begin() + pc.last + 1);
                        sp.last = (int64_t)ns.size() - 1;
                        nspans.push_back(sp);
                        sp_tile.push_back(pc.tile);
                        sp_seq.push_back(pc.seq);
                    }
                }
                bnd[(size_t)n_workers] = S;
                for (StepRec& r : ns) r.mt_flags &= ~(STEP_FIRST | STEP_LAST);          // set again per piece below
                st.swap(ns);
                spans.swap(nspans);
                plan_aligned[ty] = 0;
                P.window_plan = true;
            }
            if (!win_plan) {
            // Share of a worker.  The two (three) workgroups that share a CU do not progress at the same rate: the SIMD arbiter
            // serves the OLDER wave first, so the workgroup dispatched first (j < #CU per XCD) runs ahead -- measured on equal
            // ranges: the older one finished 512 steps in ~780 us, the younger one in 1030 us, the last 250 us alone on the CU at
            // single-occupancy speed.  Give the older one `slot_bias` more work and the younger one as much less.
            std::vector<double> wpos((size_t)n_workers, 1.0), wcum((size_t)n_workers + 1, 0.0);
            {
                const int cus_x = std::max(1, per_x / per_cu);
                for (int pos = 0; pos < n_workers; pos++) {
                    const int slot = std::min(per_cu - 1, (pos % per_x) / cus_x);
                    wpos[(size_t)pos] = per_cu == 1 ? 1.0 : 1.0 + slot_bias * (1.0 - 2.0 * slot / (double)(per_cu - 1));
                }
                for (int pos = 0; pos < n_workers; pos++) wcum[(size_t)pos + 1] = wcum[(size_t)pos] + wpos[(size_t)pos];
            }
            for (int k = 1; k < n_workers; k++) {
                const int64_t target = (int64_t)((double)total_cost * wcum[(size_t)k] / wcum[(size_t)n_workers]);
                int64_t pos = std::lower_bound(cum.begin(), cum.end(), target) - cum.begin();
                bnd[(size_t)k] = std::min<int64_t>(std::max(pos, bnd[(size_t)k - 1]), S);
            }
            // Alternative: ranges that end on tile boundaries (no split tile, no fix-up launch).  Splitting balances to one
            // step but pays the fix-up (a second launch that re-reads the partial images: ~10 us measured, `split_penalty`
            // cost units); whole tiles cost at most one tile of imbalance.  Many short tiles -> aligned; few long -> split.
            {
                // cost of a whole tile = its steps + its epilogue (cum[s] = cost before step s; a tile's epilogue is added behind its last step)
                const std::vector<int64_t> cum0 = cum;
                const std::vector<TileSpan> spans0 = spans;
                auto tile_cost = [&](size_t t) { return cum0[(size_t)spans0[t].last + 1] - cum0[(size_t)spans0[t].first]; };
                int64_t lo = 0, hi = total_cost * 2;
                for (size_t t = 0; t < spans.size(); t++) lo = std::max(lo, tile_cost(t));
                auto cap = [&](int64_t L, int64_t bin) { return bin < n_workers ? (int64_t)((double)L * wpos[(size_t)bin]) : L; };
                auto bins_needed = [&](int64_t L) {                 // bin b holds at most L x (its worker's share)
                    int64_t bins = 1, cur = 0;
                    for (size_t t = 0; t < spans.size(); t++) {
                        const int64_t c = tile_cost(t);
                        if (cur > 0 && cur + c > cap(L, bins - 1)) { bins++; cur = 0; }
                        cur += c;
                    }
                    return bins;
                };
                while (lo < hi) {                                   // smallest makespan L that fits n_workers contiguous bins
                    const int64_t mid = lo + (hi - lo) / 2;
                    if (bins_needed(mid) <= n_workers) hi = mid; else lo = mid + 1;
                }
                const int64_t split_makespan = (total_cost + n_workers - 1) / n_workers + split_penalty;
                const bool aligned = align_mode == 1 || (align_mode < 0 && lo <= split_makespan);
                if (aligned) {
                    std::fill(bnd.begin(), bnd.end(), S);
                    bnd[0] = 0;
                    int64_t bin = 0, cur = 0;
                    for (size_t t = 0; t < spans.size(); t++) {
                        const int64_t c = tile_cost(t);
                        if (cur > 0 && cur + c > cap(lo, bin) && bin + 1 < n_workers) { bin++; bnd[(size_t)bin] = spans[t].first; cur = 0; }
                        cur += c;
                    }
                    // one-tile plans whose tiles do not start on multiples of 32 rows keep CONTIGUOUS ranges: the no-barrier kernels then park
                    // finished tiles in its LDS ring and stores whole aligned blocks of C, which only works when a worker's tiles are vertically
                    // adjacent (k_f32_direct.hip, CSTAGE; banded 200k: 66 us dealt longest-first with direct stores)
                    // -- and only SHORT tiles (fewer than 6 steps per tile on average: the regime where a step's loads wait for stores): with long tiles
                    // the stores are rare and the longest-first dealing is worth more (cant-like in tiles of 31.99 rows on average: 63.9 us as a ring
                    // plan, 56.5 dealt longest-first with direct stores)
                    const bool ring_plan = ring_like;
                    if (interleave && !ring_plan) {
                        // Whole tiles can go to any worker.  Keep the 64 workers of an XCD close together in the matrix at every
                        // moment: the XCD takes a contiguous eighth of the tiles (by cost) and deals them, in matrix order, to
                        // its least-loaded worker (uniform tiles: worker j gets tiles j, j+64, ...).  The B rows the XCD
                        // touches at one time are then a narrow moving window that stays in its 4 MB L2, instead of 64
                        // windows spread over the whole eighth.  The step list is rebuilt in worker order.
                        std::vector<std::vector<size_t>> mine((size_t)n_workers);
                        size_t t = 0;
                        int64_t seen = 0;
                        for (int x = 0; x < 8; x++) {
                            const int64_t upto = total_cost * (x + 1) / 8;
                            std::vector<int64_t> load((size_t)per_x, 0);
                            std::vector<size_t> share;                       // the tiles of this XCD's eighth
                            while (t < spans.size() && (x == 7 || seen + tile_cost(t) / 2 <= upto)) {
                                share.push_back(t);
                                seen += tile_cost(t);
                                t++;
                            }
                            // interleave == 2: longest tile first (LPT) -- the makespan of 3-4 tiles of 4..14 steps per worker drops from
                            // 1.10 x the mean (contiguous whole tiles) to 1.05 x; each worker then walks its tiles in matrix order
                            if (interleave_mode == 2)
                                std::stable_sort(share.begin(), share.end(), [&](size_t a, size_t b) { return tile_cost(a) > tile_cost(b); });
                            for (size_t tt : share) {
                                size_t best = 0;
                                for (size_t j = 1; j < load.size(); j++) if (load[j] < load[best]) best = j;
                                mine[(size_t)x * per_x + best].push_back(tt);
                                load[best] += tile_cost(tt);
                            }
                            if (interleave_mode == 2)
                                for (int j = 0; j < per_x; j++) std::sort(mine[(size_t)x * per_x + j].begin(), mine[(size_t)x * per_x + j].end());
                        }
                        std::vector<StepRec> ns;
                        std::vector<TileSpan> nspans;
                        ns.reserve(st.size());
                        nspans.reserve(spans.size());
                        for (int pos = 0; pos < n_workers; pos++) {
                            bnd[(size_t)pos] = (int64_t)ns.size();
                            for (size_t tt : mine[(size_t)pos]) {
                                TileSpan sp = spans[tt];
                                const int64_t len = sp.last - sp.first + 1;
                                ns.insert(ns.end(), st.begin() + sp.first, st.begin() + sp.last + 1);
                                sp.first = (int64_t)ns.size() - len;
                                sp.last = (int64_t)ns.size() - 1;
                                nspans.push_back(sp);
                            }
                        }
                        bnd[(size_t)n_workers] = S;
                        st.swap(ns);
                        spans.swap(nspans);
                    }
                }
                plan_aligned[ty] = aligned ? 1 : 0;
            }
            }   // !win_plan
            wrange[ty].assign((size_t)n_workers * 2, 0);
            std::vector<int32_t> wid_of_pos((size_t)n_workers);
            for (int pos = 0; pos < n_workers; pos++) {
                const int x = pos / per_x, j = pos % per_x;
                const int wid = wide ? 2 * (x + 8 * (j / 2)) + (j % 2) : x + 8 * j;      // wide: neighbours in the matrix share a workgroup
                wid_of_pos[(size_t)pos] = wid;
                wrange[ty][(size_t)wid * 2] = (int32_t)bnd[(size_t)pos];
                wrange[ty][(size_t)wid * 2 + 1] = (int32_t)bnd[(size_t)pos + 1];
            }
            // segments: a tile cut by a boundary is split; every segment writes one workspace image
            // (slots are numbered densely over both types: both launches finish before the fix-up kernel reads them)
            size_t ti = 0;
            const int32_t slot_base = (int32_t)fix_slots.size();
            std::vector<std::vector<std::pair<int32_t, int32_t>>> tile_slots(win_plan ? tile_spans.size() : 0);      // window plan: (place in the tile, slot)
            int32_t n_win_slots = 0;
            for (int pos = 0; pos < n_workers; pos++) {
                const int64_t s0 = bnd[(size_t)pos], s1 = bnd[(size_t)pos + 1];
                if (s0 >= s1) continue;
                while (ti < spans.size() && spans[ti].last < s0) ti++;
                for (size_t t = ti; t < spans.size() && spans[t].first < s1; t++) {
                    const int64_t a = std::max(spans[t].first, s0), b = std::min(spans[t].last, s1 - 1);
                    const bool whole = a == spans[t].first && b == spans[t].last && (!win_plan || tile_pieces[(size_t)sp_tile[t]] == 1);
                    st[(size_t)a].mt_flags |= STEP_FIRST;
                    st[(size_t)b].mt_flags |= STEP_LAST;
                    if (!whole && win_plan) {                                // pieces are dealt whole: one image per piece, grouped per tile below
                        const int32_t slot = slot_base + n_win_slots++;
                        st[(size_t)b].mt_flags |= STEP_SPLIT;
                        st[(size_t)b].slot = slot;
                        tile_slots[(size_t)sp_tile[t]].push_back(std::make_pair(sp_seq[t], slot));
                    } else if (!whole) {
                        const int32_t slot = (int32_t)fix_slots.size();          // dense: one image per segment, both types
                        st[(size_t)b].mt_flags |= STEP_SPLIT;
                        st[(size_t)b].slot = slot;
                        if (a == spans[t].first) {                          // first segment of the tile opens its fix-up record
                            fix.push_back(FixRec{spans[t].c_row, spans[t].mt, (int32_t)fix_slots.size(), 0});
                            n_split++;
                        }
                        fix_slots.push_back(slot);
                        fix.back().n_slots++;
                    }
                }
            }
            if (win_plan)                                                    // a tile's images in k order (its pieces' places), whatever worker made them
                for (size_t t = 0; t < tile_slots.size(); t++) {
                    if (tile_slots[t].empty()) continue;
                    std::sort(tile_slots[t].begin(), tile_slots[t].end());
                    fix.push_back(FixRec{tile_spans[t].c_row, tile_spans[t].mt, (int32_t)fix_slots.size(), (int32_t)tile_slots[t].size()});
                    for (const auto& ps : tile_slots[t]) fix_slots.push_back(ps.second);
                    n_split++;
                }
            // ---- 16-bit handles: the slices of A in STEP order (the dealing above moved whole tiles between workers): slice q of this type sits at
            // base + q x slice, so that the no-barrier kernel advances ONE pointer per step instead of reading an offset from the step record
            // (eight scalar instructions of ~50 per step; the records keep the offsets for the LDS-staged kernel)
            if (h16) {
                // packed on all host threads straight from the fp32 blocks (column-major h x w, element (row, k) of the slice at blk[k * h + row]):
                // slice = [k chunk of 8][row][8], rows past the tile zero
                const size_t slice = (size_t)(ty ? 64 : 32) * (size_t)kp;
                const int64_t tms = ty ? 64 : 32;
                const size_t base = ty ? a16_steps[0].size() : 0;               // offset in the device image: type 0 first
                a16_steps[ty].assign((size_t)S * slice, 0);
                const bool bf = dtype == SPARTA_BF16;
                uint16_t* all = a16_steps[ty].data();
                sparta::parallel_for_dynamic(S, 256, [&](int64_t lo, int64_t hi, int) {
                    for (int64_t q = lo; q < hi; q++) {
                        StepRec& r = st[(size_t)q];
                        uint16_t* dst = all + (size_t)q * slice;
                        if (r.pad != 0) {                                    // pair tile: rows 0..31 from the upper block-row's block, rows 32.. from the lower one's (either may be absent: zeros)
                            const PairSrc& ps = pair_tab[(size_t)r.pad];
                            for (int64_t kk = 0; kk < kp; kk++) {
                                uint16_t* d2 = dst + (kk >> 3) * tms * 8 + (kk & 7);
                                if (ps.lo_present) {
                                    const float* colp = mab + mab_lo + r.a_off + kk * (int64_t)r.h;
                                    for (int64_t rr = 0; rr < 32; rr++) d2[rr * 8] = to_h16(colp[rr], bf);
                                }
                                if (ps.hi_present) {
                                    const float* colp = mab + mab_lo + ps.a_off_hi + kk * (int64_t)ps.h_hi;
                                    for (int64_t rr = 0; rr < ps.rows_hi; rr++) d2[(32 + rr) * 8] = to_h16(colp[rr], bf);
                                }
                            }
                            r.pad = 0;
                            r.a_off = (int64_t)(base + (size_t)q * slice);
                            continue;
                        }
                        const float* blk = mab + mab_lo + r.a_off;
                        const int64_t hh = r.h, mt = r.mt_flags & 0xffff;
                        for (int64_t kk = 0; kk < kp; kk++) {
                            const float* colp = blk + kk * hh;
                            uint16_t* d2 = dst + (kk >> 3) * tms * 8 + (kk & 7);
                            for (int64_t rr = 0; rr < mt; rr++) d2[rr * 8] = to_h16(colp[rr], bf);
                        }
                        r.a_off = (int64_t)(base + (size_t)q * slice);
                    }
                });
            }
            // ---- A in MFMA fragment order for vbs_spmm_f32_direct_kernel (k_f32_direct.hip): one 4 KB slice per step of the one-tile plan,
            // [j = 0..3][g = 0..1][row = 0..31][e = 0..3] = A[row][k = 16 g + 4 j + e], rows past the tile zero.  The legacy image of A stays: row-major / gathered B calls run the LDS-staged kernel on the same plan.
            {
                const char* de = std::getenv("SPARTA_F32_PLAN");
                const bool off = de && std::strcmp(de, "legacy") == 0;         // SPARTA_F32_PLAN=legacy: the LDS-staged kernel for every call
                // (<= 8 GiB of slices, and not more than 4 x the stored elements of the matrix: a plan of many SHORT tiles -- 4 rows pay the 4 KB of 32 --
                // keeps the LDS-staged kernel on the legacy image instead of quadrupling the handle)
                int64_t one_tile_area = 0;
                for (int64_t q = 0; q < S; q++) one_tile_area += (int64_t)(st[(size_t)q].mt_flags & 0xffff) * SK_KP;
                if (!off && ty == 0 && !h16 && S <= ((int64_t)2 << 20) && S * kAFragSlice <= 4 * std::max<int64_t>(one_tile_area, 1) + (1 << 20)) {
                    // k-COMPACTION.  A stored 32 x 32 block of a FEM matrix has 23 non-empty columns on average (the flagship: 27 % of the MFMAs would
                    // multiply columns of zeros).  The sum over k may run in any order as long as both operands agree, so per step the non-empty columns
                    // K[0..nk) of the slice go FIRST: compact index c sits at fragment position k' = (c >> 1) + 16 (c & 1), i.e. MFMA t (t = 4 j + e) takes
                    // c = 2 t from the lanes g = 0 and c = 2 t + 1 from g = 1, and only ceil(nk / 2) MFMAs -- issued in pairs -- are needed.  The kernel
                    // writes the B panel to the SAME positions: every step carries its table pos[k] (32 bytes, a permutation of 0..31: the empty columns
                    // fill the positions behind), chosen so that the eight lanes of a ds_write_b32 that hold the same e of their four k's land in different
                    // LDS banks (residues of pos mod 4 balanced inside each class k mod 4).
                    std::vector<float>& af = P.a_frag;
                    af.assign(((size_t)S + 4) * (size_t)kAFragSlice, 0.0f);                    // + 4: the pipeline requests three steps past a range end
                    sparta::parallel_for_dynamic(S, 256, [&](int64_t lo, int64_t hi, int) {
                        for (int64_t q = lo; q < hi; q++) {
                            StepRec& r = st[(size_t)q];
                            const float* blk = mab + mab_lo + r.a_off;              // element (row, k) of the slice at blk[k * h + row]
                            const int64_t hh = r.h, mt = r.mt_flags & 0xffff;
                            bool nonempty[32];
                            int nk = 0;
                            for (int k = 0; k < 32; k++) {
                                bool any = false;
                                for (int64_t m = 0; m < mt && !any; m++) any = blk[(int64_t)k * hh + m] != 0.0f;
                                nonempty[k] = any; nk += any;
                            }
                            // positions: the first nk compact indices for the non-empty columns, the rest for the empty ones; inside each group the
                            // columns are handed out class by class (k mod 4 = e) so that a class spreads over the four residues of pos mod 4
                            uint8_t pos[32];
                            int order[32], no = 0;
                            for (int pass = 0; pass < 2; pass++)
                                for (int m = 0; m < 8; m++)
                                    for (int e = 0; e < 4; e++) { const int k = 4 * m + ((e + m) & 3); if (nonempty[k] == (pass == 0)) order[no++] = k; }
                            for (int c = 0; c < 32; c++) pos[order[c]] = (uint8_t)((c >> 1) + 16 * (c & 1));
                            float* dst = af.data() + (size_t)q * (size_t)kAFragSlice;
                            std::memcpy(dst, pos, 32);
                            float* frag = dst + 16;
                            for (int k = 0; k < 32; k++) {
                                if (!nonempty[k]) continue;
                                const int kp = pos[k], g = kp >> 4, j = (kp & 15) >> 2, e = kp & 3;     // k' = 16 g + 4 j + e
                                for (int64_t m = 0; m < mt; m++) frag[((j * 2 + g) * 32 + m) * 4 + e] = blk[(int64_t)k * hh + m];
                            }
                            const int n_mfma = std::max(1, (nk + 1) / 2), pairs = (n_mfma + 1) / 2;
                            r.mt_flags = (r.mt_flags & ~(7 << STEP_KPAIRS_SHIFT)) | ((pairs - 1) << STEP_KPAIRS_SHIFT);
                        }
                    });
                    for (size_t q = (size_t)S; q < (size_t)S + 4; q++) {              // the slices behind the end: identity table (never multiplied)
                        uint8_t pos[32];
                        for (int k = 0; k < 32; k++) pos[k] = (uint8_t)k;
                        std::memcpy(af.data() + q * (size_t)kAFragSlice, pos, 32);
                    }
                }
            }
        }
        // ---- HUB PLAN, part 2: step list over the unions, K-range-major order, worker ranges, split segments, the slices of A ---------------------------------
        if (!hub_groups.empty()) {
            const int G = hub_G;
            const int64_t block_cols = (cols - 1) / w + 1, spb = w / 64;                    // steps per block
            const int hub_workers = P.n_workers;                                            // one workgroup per CU and 256-column slab
            int64_t n_ranges = 16;                                                          // K chunks of the step order, chosen below
            struct UStep { int32_t jb; int8_t mask; int32_t bidx[kHubGMax]; };              // one block column of a group's union: who has a block there, and which
            std::vector<std::vector<UStep>> un(hub_groups.size());
            sparta::parallel_for_dynamic((int64_t)hub_groups.size(), 1, [&](int64_t lo, int64_t hi, int) {
                for (int64_t gi = lo; gi < hi; gi++) {
                    const HubGroup& g = hub_groups[(size_t)gi];
                    int64_t pos[kHubGMax] = {0, 0, 0, 0};
                    std::vector<UStep>& u = un[(size_t)gi];
                    for (;;) {
                        int64_t jb = INT64_MAX;
                        for (int k = 0; k < g.n; k++) if (pos[k] < nzcount[g.ib[k]]) jb = std::min(jb, jab[jab_lo + jo_of[(size_t)(g.ib[k] - br0)] + pos[k]]);
                        if (jb == INT64_MAX) break;
                        UStep x{(int32_t)jb, 0, {-1, -1, -1, -1}};
                        for (int k = 0; k < g.n; k++)
                            if (pos[k] < nzcount[g.ib[k]] && jab[jab_lo + jo_of[(size_t)(g.ib[k] - br0)] + pos[k]] == jb) { x.mask |= (int8_t)(1 << k); x.bidx[k] = (int32_t)pos[k]; pos[k]++; }
                        u.push_back(x);
                    }
                }
            });
            // ---- the order of the step list and the cut into worker ranges.  Units = (K chunk, group): the steps of one group tile whose block columns lie in one
            // of C equal ranges of block columns.  The list is chunk-major -- (chunk, group, block column, k slice) -- and the workers take WHOLE units, in that order:
            // the workgroups that run at the same time then walk the same rows of B from the same place on, one group each: the first to touch a panel takes the miss,
            // the others hit in the L2s / the Infinity Cache (a panel is 32 KB per 64 rows of B; the B of the power-law configs is 1-4 GB).  C is chosen so that whole
            // units balance (each worker close to 1 / P of the steps); when no C does (makespan more than 10 % above the mean) the cut is by step count anywhere.
            // SPARTA_HUB_RANGES = C forces the chunk count AND the cut by step count (the first form of this plan, kept for A/B runs).
            int64_t total_union = 0;
            for (const auto& u : un) total_union += (int64_t)u.size();
            const int64_t S_all = total_union * spb;
            auto unit_sizes = [&](int64_t C, std::vector<int64_t>& cost) {          // steps of unit (chunk c, group g) at [c * groups + g]
                cost.assign((size_t)(C * (int64_t)hub_groups.size()), 0);
                for (size_t gi = 0; gi < hub_groups.size(); gi++) {
                    size_t c0 = 0;
                    for (int64_t c = 0; c < C; c++) {
                        const int64_t jb_end = (c + 1) * block_cols / C;
                        size_t c1 = c0;
                        while (c1 < un[gi].size() && un[gi][c1].jb < jb_end) c1++;
                        cost[(size_t)(c * (int64_t)hub_groups.size()) + gi] = (int64_t)(c1 - c0) * spb;
                        c0 = c1;
                    }
                }
            };
            auto aligned_cut = [&](const std::vector<int64_t>& cost, std::vector<int64_t>* bnd_out) -> int64_t {      // greedy: consecutive whole units per worker
                const double T = (double)S_all / (double)hub_workers;
                int64_t worst = 0, load = 0, at = 0;
                int wk = 0;
                if (bnd_out) { bnd_out->assign((size_t)hub_workers + 1, S_all); (*bnd_out)[0] = 0; }
                for (size_t q = 0; q < cost.size(); q++) {
                    if (cost[q] == 0) continue;
                    if (load > 0 && (double)load + 0.5 * (double)cost[q] > T && wk + 1 < hub_workers) {
                        worst = std::max(worst, load); load = 0; wk++;
                        if (bnd_out) (*bnd_out)[(size_t)wk] = at;
                    }
                    load += cost[q]; at += cost[q];
                }
                return std::max(worst, load);
            };
            int64_t forced_ranges = -1;
            if (const char* e = std::getenv("SPARTA_HUB_RANGES")) forced_ranges = std::max<int64_t>(1, atoll(e));
            int64_t best_C = 16;
            bool cut_aligned = false;
            std::vector<int64_t> wbnd;                                       // worker boundaries in steps (aligned cut)
            if (forced_ranges > 0) best_C = std::min(forced_ranges, block_cols);
            else {
                int64_t best_span = INT64_MAX;
                std::vector<int64_t> cost;
                for (int64_t C = 1; C <= std::min<int64_t>(64, block_cols); C++) {
                    unit_sizes(C, cost);
                    // every unit that is not a whole tile leaves partial images for the fix-up (G x N / 128 images of 32 KB, written and read once): measured
                    // 0.11 us per unit at N = 512 against 4.8 us per step of the makespan (part 0 of configs[3] at 1 %: 54 chunks x 89 groups, fix-up 0.53 ms)
                    const int64_t span = aligned_cut(cost, nullptr) + (C > 1 ? (int64_t)(0.025 * (double)C * (double)hub_groups.size()) : 0);
                    if (span < best_span) { best_span = span; best_C = C; }
                }
                if ((double)best_span <= 1.10 * (double)S_all / (double)hub_workers + 1.0) {
                    cut_aligned = true;
                    unit_sizes(best_C, cost);
                    aligned_cut(cost, &wbnd);
                } else best_C = std::min<int64_t>(16, block_cols);
            }
            // ROUND-ROBIN DEALING (round 4, second form).  With consecutive whole units per worker the 16-32 workers that share an XCD at any one time sit in 4-8 different K chunks
            // (worker w starts at unit w * units-per-worker): a panel of B is shared by ~2 workgroups before the L2 drops it (FETCH_SIZE on part 0 of configs[3] at 5 %:
            // 49 GB per launch = A once + 25 x |B|).  Dealt round-robin -- unit t * W + j to worker j, odd rounds backwards so that trends in the group lengths cancel -- the
            // workers of an XCD hold CONSECUTIVE units at every moment: the same K chunk, neighbouring groups (similar lengths: the groups are formed in order of length).
            // The step list is laid out worker by worker, so the kernel and its worker ranges stay as they are.  MEASURED (hub parts of configs[3] at 5 % / 1 %, configs[4]):
            // 12.76 / 12.58 / 6.87 ms against 12.27 / 12.85 / 6.86 with consecutive units, and FETCH_SIZE unchanged (50.5 against 50.3 GB): the workers of an XCD fall
            // out of step inside their first unit (group lengths differ by more than the ~7 steps of panels an L2 holds beside the 512 KB of A slices per step), so the
            // panels are not shared either way.  Off by default (SPARTA_HUB_DEAL=1 switches it on; tests cover both): what it needs is a step the workers of an XCD take
            // TOGETHER at unit boundaries.
            bool dealt = false;
            std::vector<std::vector<int64_t>> units_of;                      // per worker: the units (index into the chunk-major cost table) it runs, in order
            {
                const char* de = std::getenv("SPARTA_HUB_DEAL");
                const bool want = forced_ranges <= 0 && (de ? atoi(de) != 0 : false);
                if (want) {
                    int64_t best_span = INT64_MAX, best = -1;
                    std::vector<int64_t> cost;
                    auto deal = [&](const std::vector<int64_t>& c, std::vector<std::vector<int64_t>>* out) -> int64_t {
                        std::vector<int64_t> load((size_t)hub_workers, 0);
                        if (out) out->assign((size_t)hub_workers, {});
                        int64_t k = 0;
                        for (size_t q = 0; q < c.size(); q++) {
                            if (c[q] == 0) continue;
                            const int64_t t = k / hub_workers, j = k % hub_workers, wk = (t & 1) ? hub_workers - 1 - j : j;
                            load[(size_t)wk] += c[q];
                            if (out) (*out)[(size_t)wk].push_back((int64_t)q);
                            k++;
                        }
                        return *std::max_element(load.begin(), load.end());
                    };
                    for (int64_t C = 1; C <= std::min<int64_t>(64, block_cols); C++) {
                        unit_sizes(C, cost);
                        const int64_t span = deal(cost, nullptr) + (C > 1 ? (int64_t)(0.025 * (double)C * (double)hub_groups.size()) : 0);
                        if (span < best_span) { best_span = span; best = C; }
                    }
                    if (best > 0 && (double)best_span <= 1.10 * (double)S_all / (double)hub_workers + 1.0) {
                        dealt = true; cut_aligned = true; best_C = best;
                        unit_sizes(best_C, cost);
                        deal(cost, &units_of);
                    }
                }
            }
            n_ranges = best_C;
            P.hub_chunks = best_C;
            // order: (K chunk, group, block column, k slice)
            struct ORef { int32_t g, u, ks; };
            std::vector<ORef> order;
            order.reserve((size_t)S_all);
            if (dealt) {
                // the first union entry of every unit (chunk-major table), then the list worker by worker
                const size_t NG = hub_groups.size();
                std::vector<size_t> first((size_t)n_ranges * NG + 1, 0);
                for (size_t gi = 0; gi < NG; gi++) {
                    size_t c = 0;
                    for (int64_t r = 0; r < n_ranges; r++) {
                        first[(size_t)r * NG + gi] = c;
                        const int64_t jb_end = (r + 1) * block_cols / n_ranges;
                        while (c < un[gi].size() && un[gi][c].jb < jb_end) c++;
                    }
                }
                wbnd.assign((size_t)hub_workers + 1, 0);
                for (int wk = 0; wk < hub_workers; wk++) {
                    for (int64_t q : units_of[(size_t)wk]) {
                        const size_t gi = (size_t)q % NG;
                        const int64_t r = q / (int64_t)NG, jb_end = (r + 1) * block_cols / n_ranges;
                        for (size_t c = first[(size_t)q]; c < un[gi].size() && un[gi][c].jb < jb_end; c++)
                            for (int64_t k = 0; k < spb; k++) order.push_back(ORef{(int32_t)gi, (int32_t)c, (int32_t)(k * 64)});
                    }
                    wbnd[(size_t)wk + 1] = (int64_t)order.size();
                }
            } else {
                std::vector<size_t> cur(hub_groups.size(), 0);
                for (int64_t r = 0; r < n_ranges; r++) {
                    const int64_t jb_end = (r + 1) * block_cols / n_ranges;
                    for (size_t gi = 0; gi < hub_groups.size(); gi++) {
                        size_t& c = cur[gi];
                        while (c < un[gi].size() && un[gi][c].jb < jb_end) { for (int64_t k = 0; k < spb; k++) order.push_back(ORef{(int32_t)gi, (int32_t)c, (int32_t)(k * 64)}); c++; }
                    }
                }
            }
            const int64_t S = (int64_t)order.size();
            if (S > INT32_MAX - 64) return fail(SPARTA_ERR_UNSUPPORTED, "sparta_vbs_create: matrix too large for 32-bit step indexing");
            P.hub_g = G; P.hub_workers = hub_workers; P.n_hub_steps = S; P.n_hub_groups = (int64_t)hub_groups.size();
            const int64_t row0 = row_part[br0];
            P.hub_tiles.resize(hub_groups.size());
            for (size_t gi = 0; gi < hub_groups.size(); gi++) {
                HubTile t{{0, 0, 0, 0}, {0, 0, 0, 0}};
                for (int k = 0; k < hub_groups[gi].n; k++) {
                    const int64_t ib = hub_groups[gi].ib[k];
                    t.c_row[k] = (int32_t)(row_part[ib] - row0); t.mt[k] = (int32_t)(row_part[ib + 1] - row_part[ib]);
                    P.hub_area += nzcount[ib] * (row_part[ib + 1] - row_part[ib]) * w;
                    P.n_hub_tiles++;
                }
                P.hub_union_area += (int64_t)un[gi].size() * w * 64 * hub_groups[gi].n;
                P.hub_tiles[gi] = t;
            }
            std::vector<HubStep>& hs = P.hub_steps;
            hs.resize((size_t)S + 32);
            std::vector<int64_t> a_at((size_t)S + 1, 0);                                    // element offset of every step's first slice
            for (int64_t q = 0; q < S; q++) a_at[(size_t)q + 1] = a_at[(size_t)q] + (int64_t)__builtin_popcount((unsigned)(uint8_t)un[(size_t)order[(size_t)q].g][(size_t)order[(size_t)q].u].mask) * 64 * 64;
            for (int64_t q = 0; q < S; q++) {
                const ORef& o = order[(size_t)q];
                const UStep& x = un[(size_t)o.g][(size_t)o.u];
                HubStep h;
                h.a_lo = (uint32_t)a_at[(size_t)q]; h.a_hi = (uint32_t)(a_at[(size_t)q] >> 32);
                h.b_row = (int32_t)((int64_t)x.jb * w + o.ks); h.shard = 0; h.flags = (int32_t)(uint8_t)x.mask; h.slot = -1; h.tile = o.g; h.pad = 0;
                if (((int64_t)x.jb + 1) * w > cols) { h.flags |= STEP_TAIL; h.b_row = o.ks; }
                hs[(size_t)q] = h;
            }
            // worker ranges, in plan order (the kernel maps workgroup ids to this order so that the workers that run at the same time are neighbours here)
            P.hub_wrange.assign((size_t)hub_workers * 2, 0);
            std::vector<std::vector<std::pair<int64_t, int32_t>>> segs_of(hub_groups.size());       // per group: (first step of the segment in the group's own order, slot base)
            int32_t next_id = (int32_t)fix_slots.size();
            auto own_index = [&](int64_t q) { const ORef& o = order[(size_t)q]; return (int64_t)o.u * spb + o.ks / 64; };     // place of step q in its group's k order
            for (int pos = 0; pos < hub_workers; pos++) {
                const int64_t s0 = cut_aligned ? wbnd[(size_t)pos] : S * pos / hub_workers, s1 = cut_aligned ? wbnd[(size_t)pos + 1] : S * (pos + 1) / hub_workers;
                const int wid = pos;
                P.hub_wrange[(size_t)wid * 2] = (int32_t)s0; P.hub_wrange[(size_t)wid * 2 + 1] = (int32_t)s1;
                int64_t a = s0;
                while (a < s1) {
                    int64_t b = a;                                           // the run of consecutive steps of one group that starts at a
                    while (b + 1 < s1 && order[(size_t)b + 1].g == order[(size_t)a].g && own_index(b + 1) == own_index(b) + 1) b++;
                    const size_t gi = (size_t)order[(size_t)a].g;
                    const bool whole = own_index(a) == 0 && own_index(b) == (int64_t)un[gi].size() * spb - 1;
                    hs[(size_t)b].flags |= STEP_LAST;
                    P.hub_segments++;
                    if (!whole) {
                        hs[(size_t)b].flags |= STEP_SPLIT;
                        hs[(size_t)b].slot = next_id;
                        segs_of[gi].emplace_back(own_index(a), next_id);
                        next_id += G;
                    }
                    a = b + 1;
                }
            }
            for (size_t gi = 0; gi < hub_groups.size(); gi++) {
                if (segs_of[gi].empty()) continue;
                std::sort(segs_of[gi].begin(), segs_of[gi].end());           // a tile's images are added in k order, whatever worker made them
                for (int k = 0; k < hub_groups[gi].n; k++) {
                    fix.push_back(FixRec{P.hub_tiles[gi].c_row[k], P.hub_tiles[gi].mt[k], (int32_t)fix_slots.size(), (int32_t)segs_of[gi].size()});
                    for (const auto& sg : segs_of[gi]) fix_slots.push_back(sg.second + k);
                    n_split++;
                }
                for (int k = hub_groups[gi].n; k < G; k++)                   // (the images of the places a short group leaves empty: written by the kernel, read by nobody)
                    for (const auto& sg : segs_of[gi]) fix_slots.push_back(sg.second + k);
            }
            for (int k = 0; k < 32; k++) { HubStep d = hs[(size_t)S - 1]; d.flags &= ~(STEP_LAST | STEP_SPLIT); d.slot = -1; hs[(size_t)S + k] = d; }
            // the slices of A, in execution order: 64 rows x 64 k, element (row, k) at row * 64 + (((k / 8) ^ ((row / 2) & 7)) * 8 + k % 8 -- the LDS image of
            // vbs_spmm_h16_hub_kernel (a ds_read_b128 of 16 rows covers all 64 banks); rows past the sub-tile's height zero
            P.hub_a16_elems = (size_t)a_at[(size_t)S];
            P.hub_a16.reset(new (std::nothrow) uint16_t[std::max<size_t>(P.hub_a16_elems, 1)]);      // NOT zero-filled (22 GB on the hub part of configs[3] at 5 %: 2 s on one thread)
            if (!P.hub_a16) return fail(SPARTA_ERR_ALLOC, "sparta_vbs_create: out of host memory for the hub slices");
            const bool bf = dtype == SPARTA_BF16;
            uint16_t* all = P.hub_a16.get();
            sparta::parallel_for_dynamic(S, 64, [&](int64_t lo, int64_t hi, int) {
                for (int64_t q = lo; q < hi; q++) {
                    const ORef& o = order[(size_t)q];
                    const UStep& x = un[(size_t)o.g][(size_t)o.u];
                    uint16_t* dst = all + a_at[(size_t)q];
                    for (int k = 0; k < hub_groups[(size_t)o.g].n; k++) {
                        if (!((x.mask >> k) & 1)) continue;
                        const int64_t ib = hub_groups[(size_t)o.g].ib[k], hh = row_part[ib + 1] - row_part[ib];
                        const float* blk = mab + mab_lo + mo_of[(size_t)(ib - br0)] + ((int64_t)x.bidx[k] * w + o.ks) * hh;     // column-major hh x w block, k slice o.ks
                        if (hh < 64) std::memset(dst + hh * 64, 0, (size_t)(64 - hh) * 64 * sizeof(uint16_t));              // rows past the sub-tile's height
                        for (int64_t kk = 0; kk < 64; kk++) {
                            const float* colp = blk + kk * hh;
                            for (int64_t rr = 0; rr < hh; rr++) dst[rr * 64 + ((((kk >> 3) ^ ((rr >> 1) & 7)) << 3) | (kk & 7))] = to_h16(colp[rr], bf);
                        }
                        dst += 64 * 64;
                    }
                }
            });
        }
    }

    // (16-bit: P.a16_steps[0] + P.a16_steps[1] is the device image; the caller pads it: the pipelines request a few slices past a range end)
    return SPARTA_OK;
}

}  // namespace sparta_dev
