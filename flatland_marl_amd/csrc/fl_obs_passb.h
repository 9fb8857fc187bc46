// fl_obs_passb.h -- pass B of the trees: the agent-dependent features of every visited cell of every node.  The visited cells
// of ALL trees of a round are split evenly over ALL lanes of the workgroup; a lane only classifies its cells, the cells that
// need work go to two work lists that are processed one entry per lane on packed wavefronts.  Device code.
#pragma once
#include "fl_obs_ctx.h"

// The feature block of ONE visited cell of a branch walk (treeobs.cpp:322-465 / observations.py:296-371) is split in
// two event handlers that merge straight into the node's accumulators (sc = the team's node table) with LDS atomics:
// min / sum / max are associative and tot_dist grows along a walk, so "first hit" = minimum.
//
// occupant of the cell (treeobs.cpp:322-357 / observations.py:296-327)
__device__ __forceinline__ void occ_event(const ObsCtx &X, bool CUTILS, int *sc, int cap, int node, uint32_t sl, uint32_t d, int tot) {
    const int ag = X.slot_agent[sl];
    if (ag < 0) return;
    atomicMin(&nt_w(sc, cap, N_OA, node), tot);
    const int mf = (int)X.a_malf[ag];
    if (mf > 0) {  // cutils keeps a flag (treeobs.cpp:335-337), upstream the largest down counter (observations.py:306-309)
        if (CUTILS) atomicOr(&nt_w(sc, cap, N_RM, node), 1 << 16);
        else atomicMax(&nt_w(sc, cap, N_MALF, node), mf);
    }
    const int rd = X.slot_ready[sl];
    const int radd = rd > 0 ? (CUTILS ? rd - 1 : rd) : 0;  // cutils starts the count at 0 (treeobs.cpp:82-91)
    if (radd) atomicAdd(&nt_w(sc, cap, N_RM, node), radd);
    if (X.a_dir[ag] == d) {
        atomicAdd(&nt_w(sc, cap, N_CNT, node), 1);
        const double sp = CUTILS ? (double)(float)X.a_speed[ag] : X.a_speed[ag];
        // the slowest such agent: speeds are static, so the minimum goes over (rank of the speed, agent)
        if (sp < 1.0) atomicMin(reinterpret_cast<unsigned int *>(&nt_w(sc, cap, N_MS, node)), ((unsigned int)X.a_srank[ag] << 10) | (unsigned int)ag);
    } else {
        atomicAdd(&nt_w(sc, cap, N_CNT, node), 1 << 16);
    }
}

// potential conflict at predicted time pt (treeobs.cpp:378-465 / observations.py:329-367); the caller checked
// Tn > 0, tot < Tn and pt < Tn.  conflict_flags scans the items [v0, v1) of the virtual range R of the cell's key (list_range) and returns six bits:
// bit k (k = 0, 1, 2 for the times pt, pt - 1, pt + 1): some OTHER agent is predicted there then; bit 3 + k: some agent
// predicted there then (self included) satisfies the conflict condition.  Flags of sub-ranges of a list simply OR.
template <int PB, bool ITL, bool TWO>
__device__ __forceinline__ uint32_t conflict_flags(const ObsCtx &X, bool CUTILS, int handle, int cell, uint32_t d, int pt, const ListRange &R, int v0, int v1) {
    const uint32_t bits = nibble(cw_bits(X, cell), d);
    const bool second = PB == 2 && !CUTILS;  // the upstream predictor's index
    const int Tn = second ? X.u_Tn : X.Tn;
    const uint32_t tlast = (uint32_t)(Tn - 1);
    const uint32_t t0 = (uint32_t)pt, t1 = (uint32_t)max(pt - 1, 0), t2 = (uint32_t)min(pt + 1, Tn - 1);
    uint32_t flags = 0;
    auto test_item = [&](uint32_t it) __attribute__((always_inline)) {
        const uint32_t tl = IT_TLO(it), th = IT_THI(it, tlast);
        const uint32_t in = (uint32_t)(tl <= t0 && t0 <= th) | ((uint32_t)(tl <= t1 && t1 <= th) << 1) | ((uint32_t)(tl <= t2 && t2 <= th) << 2);
        const int a = IT_AGENT(it);
        // direction the conflict test uses: upstream takes the one at the matching time step
        // (observations.py:351-363); cutils indexes predicted_dir with predicted_time in all three branches
        // (treeobs.cpp:429-433, 449-453), i.e. the neighbouring waypoint's direction when the agent is not on
        // this waypoint at t0
        uint32_t cd = IT_DIR(it);
        if (CUTILS && !(in & 1u)) cd = t0 > th ? IT_DNEXT(it) : IT_DPREV(it);
        const bool cnd = (d != cd && ((bits >> (3u - ((cd + 2u) & 3u))) & 1u)) || X.a_state[a] == ST_DONE;
        flags |= (a != handle ? in : 0u) | (cnd ? in << 3 : 0u);
#ifdef FL_OBS_COUNTS
        if (a == handle && in) flags |= 64u;  // the walking agent itself is predicted there then
#endif
    };
    // the key's list is short and unsorted: scan it with NB independent loads in flight, most items fall out at the
    // interval test (sorting the lists costs more than it saves; so did a separate pass that first collects the matching
    // items of a chunk and then tests only those -- 1 to 2 % slower on every workload)
    auto scan = [&](const uint32_t *items, auto nb) __attribute__((always_inline)) {
        constexpr int NB = decltype(nb)::value;
        for (int e0 = v0; e0 < v1; e0 += NB) {
            uint32_t itv[NB];
#pragma unroll
            for (int q = 0; q < NB; q++) itv[q] = items[list_index<TWO>(R, min(e0 + q, v1 - 1))];
#pragma unroll
            for (int q = 0; q < NB; q++) {
                const uint32_t tl = IT_TLO(itv[q]), th = IT_THI(itv[q], tlast);
                if (e0 + q < v1 && th >= t1 && tl <= t2) test_item(itv[q]);
            }
        }
    };
    // separate call sites so that each keeps a static address space; whole lists in HBM scratch (large maps without time
    // masks) are fetched in bigger batches: their round trips are what the scan costs
    // (one call site for the two LDS copies: the lanes of a wavefront work on entries of both builders, two sites would run
    // one after the other)
    // Items in HBM scratch: the eight items of a chunk are consecutive words of the index (a list in two pieces: unless the chunk
    // straddles them), so a lane fetches them with TWO 16-byte loads (4-byte aligned: global_load_dwordx4 takes that) instead of
    // eight gathers -- every lane of a wavefront is in another list, i.e. another cache line, and the texture addresser takes a
    // gather of 64 lines at one line a cycle whatever its width.  (Words behind the end of a list are read and masked: the
    // scratch is padded, fl_obs.hip.)
    auto scan_glb = [&](const uint32_t *items) __attribute__((always_inline)) {
        typedef uint32_t u4a __attribute__((ext_vector_type(4), aligned(4)));
        for (int e0 = v0; e0 < v1; e0 += 8) {
            uint32_t itv[8];
            const int p0 = list_index<TWO>(R, e0);
            if (!TWO || e0 >= R.n1 || R.n1 - e0 >= min(8, v1 - e0)) {
                const u4a a = *reinterpret_cast<const u4a *>(items + p0), c = *reinterpret_cast<const u4a *>(items + p0 + 4);
                itv[0] = a.x; itv[1] = a.y; itv[2] = a.z; itv[3] = a.w; itv[4] = c.x; itv[5] = c.y; itv[6] = c.z; itv[7] = c.w;
            } else {
#pragma unroll
                for (int q = 0; q < 8; q++) itv[q] = items[list_index<TWO>(R, min(e0 + q, v1 - 1))];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t tl = IT_TLO(itv[q]), th = IT_THI(itv[q], tlast);
                if (e0 + q < v1 && th >= t1 && tl <= t2) test_item(itv[q]);
            }
        }
    };
    static_assert(CF_CHUNK_HBM == 8 && OBS_GLB_BATCH == 8, "scan_glb fetches eight items a round trip");
    if (ITL) scan(second ? X.u_items : X.items_lds, std::integral_constant<int, 8>());
    else if (second) scan(X.u_items, std::integral_constant<int, 8>());      // (the second index is always LDS-resident)
    else scan_glb(X.items_glb);  // chunked work-list entries: the whole chunk in flight at once
    return flags;
}
// the other-agent test takes the first time (pt, pt - 1, pt + 1) at which somebody else is predicted on the cell
__device__ __forceinline__ bool conflict_hit(uint32_t f) { return (f & 1u) ? (f >> 3) & 1u : ((f & 2u) ? (f >> 4) & 1u : ((f & 4u) ? (f >> 5) & 1u : false)); }

template <int PB, bool ITL, bool TWO>
__device__ __forceinline__ void conflict_event(const ObsCtx &X, bool cu, int *sc, int cap, int node, int handle, int cell, uint32_t d, int tot, int pt) {
    const ListRange R = list_range<PB>(X, cu, cell, pt);
    if (R.n <= 0) return;
    if (conflict_hit(conflict_flags<PB, ITL, TWO>(X, cu, handle, cell, d, pt, R, 0, R.n))) atomicMin(&nt_w(sc, cap, N_PC, node), tot);
}

// Entry i of the occupant / conflict work list.  Lists in HBM scratch may have an LDS HEAD (ObsCtx::wl_head_*): entry i < head
// entries lives in LDS, any other in HBM at the same index (separate accesses: each pointer keeps its static address space).
__device__ __forceinline__ uint2 wl_occ_get(const ObsCtx &X, int i) { return i < X.wl_head_occ_n ? X.wl_head_occ[i] : X.wl_occ[i]; }
__device__ __forceinline__ void wl_occ_put(const ObsCtx &X, int i, uint2 e) { if (i < X.wl_head_occ_n) X.wl_head_occ[i] = e; else X.wl_occ[i] = e; }
__device__ __forceinline__ uint2 wl_cf_get(const ObsCtx &X, int i) { return i < X.wl_head_cf_n ? X.wl_head_cf[i] : X.wl_cf[i]; }
__device__ __forceinline__ void wl_cf_put(const ObsCtx &X, int i, uint2 e) { if (i < X.wl_head_cf_n) X.wl_head_cf[i] = e; else X.wl_cf[i] = e; }
__device__ __forceinline__ void wl_cf_set_y(const ObsCtx &X, int i, uint32_t y) { if (i < X.wl_head_cf_n) X.wl_head_cf[i].y = y; else X.wl_cf[i].y = y; }
// a further chunk of entry i's key is scanned: OR its flag bits into the entry's word and count the chunk down (bits 9 .. 14); returns
// the word as it was before the count-down (the OR and the count-down go to one address in program order)
__device__ __forceinline__ uint32_t wl_cf_done(const ObsCtx &X, int i, uint32_t bits) {
    if (i < X.wl_head_cf_n) {
        if (bits) atomicOr(&X.wl_head_cf[i].y, bits);
        return atomicSub(&X.wl_head_cf[i].y, 1u << 9);
    }
    if (bits) atomicOr(&X.wl_cf[i].y, bits);
    return atomicSub(&X.wl_cf[i].y, 1u << 9);
}

// append e to the conflict work list; one LDS atomic per wavefront.  false = the list is full and the caller handles the event itself
__device__ __forceinline__ bool wl_push_cf(const ObsCtx &X, bool want, uint2 e) {
    const unsigned long long m = __ballot(want);
    if (m == 0) return true;
    const int lane = (int)__lane_id();
    // the first ACTIVE lane reserves the slots for the wavefront; its result is broadcast with v_readfirstlane (a shuffle
    // would be another LDS round trip)
    int base = 0;
    if (lane == __ffsll((long long)__ballot(1)) - 1) base = atomicAdd(&X.wl_cnt[1], __popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    const int idx = base + __popcll(m & ((1ull << lane) - 1ull));
    if (want && idx < X.wl_cf_cap) wl_cf_put(X, idx, e);
    return !want || idx < X.wl_cf_cap;
}

// Slots in BOTH work lists with one LDS atomic per wavefront (the two counters are the halves of one 64-bit word), split in
// two so that the caller can issue the next cell's loads while the atomic is in flight.
__device__ __forceinline__ unsigned long long wl_reserve2_issue(int *cnt, unsigned long long m0, unsigned long long m1) {
    unsigned long long old = 0;
    // one lane adds for the wavefront.  The address goes through a register the compiler cannot see through: for an address it
    // knows to be uniform it rewrites the atomic into its own wave reduction and waits for the result on the spot
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    if ((m0 | m1) != 0ull && (int)__lane_id() == __ffsll((long long)__ballot(1)) - 1)
        old = atomicAdd(reinterpret_cast<unsigned long long *>(cnt) + zero, (unsigned long long)__popcll(m0) | ((unsigned long long)__popcll(m1) << 32));
    return old;
}
__device__ __forceinline__ void wl_reserve2_finish(unsigned long long old, unsigned long long m0, unsigned long long m1, int &i0, int &i1) {
    const int b0 = __builtin_amdgcn_readfirstlane((int)(uint32_t)old), b1 = __builtin_amdgcn_readfirstlane((int)(uint32_t)(old >> 32));
    const unsigned long long lt = (1ull << __lane_id()) - 1ull;
    i0 = b0 + __popcll(m0 & lt);
    i1 = b1 + __popcll(m1 & lt);
}

#ifdef FL_OBS_TIMING
// per-wavefront marks inside a phase (absolute clock): slot k = latest wavefront, slot k2 = 2^40 - earliest wavefront
#define WAVE_MARK(X, k, k2) do { if ((X).dbg && (threadIdx.x & 63) == 0) { const long long now_ = (long long)wall_clock64() & 0xFFFFFFFFFFll; \
    atomicMax((unsigned long long *)&(X).dbg[(X).dbg_base + (k)], (unsigned long long)now_); \
    if ((k2) >= 0) atomicMax((unsigned long long *)&(X).dbg[(X).dbg_base + ((k2) < 0 ? 0 : (k2))], (unsigned long long)((1ll << 40) - now_)); } } while (0)
#else
#define WAVE_MARK(X, k, k2) do {} while (0)
#endif
#ifdef FL_OBS_TIMING
// sums over the wavefronts (and rounds of trees) of the time since the start of the work-list step, slot 51 + k (k = 5: slot 48) of the env
#define WL_ACC(X, k) do { if ((X).dbg && (threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&(X).dbg[(k) == 5 ? 48 : 51 + (k)], (unsigned long long)((long long)wall_clock64() - wl_t0_)); } while (0)
#else
#define WL_ACC(X, k) do {} while (0)
#endif
#ifdef FL_OBS_TIMING
// accumulates the time since the previous stamp of this stage in slot k (summed over the rounds of trees)
#define TREE_STAMP(X, k) do { __syncthreads(); if (threadIdx.x == 0 && (X).dbg) { const long long now_ = (long long)wall_clock64(); (X).dbg[(X).dbg_base + (k)] += now_ - (X).dbg[(X).dbg_base + 15]; (X).dbg[(X).dbg_base + 15] = now_; } } while (0)
#else
#define TREE_STAMP(X, k) do {} while (0)
#endif

__device__ __forceinline__ void team_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Pass B of the trees.  team_prepare: per team (= one agent's tree), inclusive prefix of the nodes' visit counts, a link
// from every node to the next node that has cells, and reset of the node accumulators.  wg_pass_b: the visited cells of ALL
// nodes of ALL trees of the batch are split evenly over ALL lanes of the workgroup; every lane walks its slice (search
// for its first team / node, a skip to the slice start, then ONE lock-step loop over its cells).
//
// N_INCL word of node k: inclusive prefix (24 bits) | index of the next node with cells << 24 (0xFF = none).
// Returns the team's number of cells; first_real = its first node with cells (0xFF = none).
// CAP = slots of the table, STRIDE = words between its fields (CAP, or 32 when two 16-slot tables share one, see team_table)
template <int TEAM, int CAP, bool UPSTREAM, int STRIDE = CAP>
__device__ __forceinline__ int team_prepare(bool have, int tl, int n_nodes, int *scr, int &first_real) {
    constexpr int NCH = (CAP + TEAM - 1) / TEAM;
    const int tbase = ((int)__lane_id() / TEAM) * TEAM;
    const unsigned long long tbits = TEAM == 64 ? ~0ull : ((1ull << (TEAM & 63)) - 1ull);
    int v[NCH];
    unsigned long long real[NCH];  // bit j: node c * TEAM + j has cells
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int k = c * TEAM + tl;
        v[c] = (have && k < n_nodes && k < CAP) ? nt_vis((uint32_t)nt_r(scr, STRIDE, N_TV, k)) : 0;  // (an empty slot has no cells)
        real[c] = (__ballot(v[c] > 0) >> tbase) & tbits;
    }
    int run_base = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int k = c * TEAM + tl;
        int incl = v[c];
#pragma unroll
        for (int off = 1; off < TEAM; off <<= 1) { const int u = __shfl_up(incl, off, TEAM); if (tl >= off) incl += u; }
        incl += run_base;
        int nxt = 0xFF;
#pragma unroll
        for (int c2 = NCH - 1; c2 > c; c2--)
            if (real[c2]) nxt = c2 * TEAM + __ffsll((long long)real[c2]) - 1;
        const unsigned long long above = tl + 1 < TEAM ? real[c] >> ((tl + 1) & 63) : 0ull;
        if (above) nxt = k + __ffsll((long long)above);
        if (k < CAP) {
            nt_w(scr, STRIDE, N_INCL, k) = incl | (nxt << 24);
            nt_w(scr, STRIDE, N_OA, k) = 0x7fffffff; nt_w(scr, STRIDE, N_PC, k) = 0x7fffffff;
            nt_w(scr, STRIDE, N_CNT, k) = 0; nt_w(scr, STRIDE, N_RM, k) = 0;
            nt_w(scr, STRIDE, N_MS, k) = -1;  // nobody slower than 1.0
            if (UPSTREAM) { nt_w(scr, STRIDE, N_OT, k) = 0x7fffffff; nt_w(scr, STRIDE, N_MALF, k) = 0; }
        }
        run_base = __shfl(incl, TEAM - 1, TEAM);
    }
    first_real = 0xFF;
#pragma unroll
    for (int c = NCH - 1; c >= 0; c--)
        if (real[c]) first_real = c * TEAM + __ffsll((long long)real[c]) - 1;
    return run_base;
}

// team_meta: [0,64) cells per team, [64,128) nodes per team, [128,192) agent of the team (or -1), [192,256) BFS levels (cutils),
// [256,320) first node with cells
//
// Step 1: every lane walks its slice of the visited cells and only CLASSIFIES them (three cheap tests per cell: has the
// cell an occupant; does the time-bucket mask of its key say that somebody is predicted there around the queried time;
// is it somebody's target) -- cells that need work go to two LDS work lists.  Step 2: the lists are processed one entry
// per lane, so the expensive handlers run on densely packed wavefronts instead of as rare side branches of a lock-step loop.
// The loop of step 1 is a chain of LDS round trips, so it is software-pipelined by hand: everything the NEXT cell needs
// (rail word, successor, time mask, the next node's descriptor when the walk ends here) is requested while the work-list
// reservation of the current cell is in flight -- about one round trip per cell.
// Tables: PB 0 / 1: team t's table is scr0 + t * team_words, fields CAP words apart.  PB 2 (both builders, trees_merged): the
// teams below X.n_cu (the agents of a round) are flatland_cutils trees (32 slots, N_WORDS_C fields of 32 words); behind them the compact
// upstream trees, TWO to a table of N_WORDS_T fields of 32 words (tree u uses the slots (u & 1) * 16 ... + 15 of every field):
// the field stride is 32 words for every team, a compile-time constant in the classify loop.
// The trees of a round: X.n_cu agents (32 on sixteen wavefronts; 16 on eight -- the 512-thread kernels, two workgroups a CU).
template <int PB, int CAP>
__device__ __forceinline__ int *team_table(const ObsCtx &X, int *scr0, int team_words, int team) {
    if (PB == 3) return scr0 + team * (N_WORDS_C * 32);
    if (PB == 2) {
        const int u = team - X.n_cu;
        return u < 0 ? scr0 + team * (N_WORDS_C * 32) : scr0 + X.n_cu * (N_WORDS_C * 32) + (u >> 1) * (N_WORDS_T * 32) + (u & 1) * 16;
    }
    return scr0 + team * team_words;
}

// late(): work that does not depend on the trees (the rest of phase 1: attribute rows, valid actions), handed out through a queue;
// a wavefront that is done with its share of the work-list step takes some while the others finish theirs.
struct NoLateWork { __device__ __forceinline__ void operator()() const {} };

// TWO: the lists of the index may come in two pieces (see ListRange)
template <int PB, int CAP, bool ITL, typename LATE = NoLateWork, bool TWO = false>
__device__ __forceinline__ void wg_pass_b(const ObsCtx &X, int tid, int nt, int n_teams, int *scr0, int team_words,
                                          const int *team_meta, const LATE &late = LATE()) {
    if (tid == 0) { X.wl_cnt[0] = 0; X.wl_cnt[1] = 0; X.wl_cnt[2] = 0; }
    __syncthreads();
    const int lane = tid & 63;
    // inclusive prefix over the teams' cell counts, one team per lane (n_teams <= 64); every wavefront computes it
    const int tv = lane < n_teams ? team_meta[lane] : 0;
    int tincl = tv;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(tincl, off); if (lane >= off) tincl += u; }
    const int total = __builtin_amdgcn_readlane(tincl, 63);
    const int q = (total + nt - 1) / nt;
    int pos = tid * q;
    const int end = min(pos + q, total);
    // first team whose inclusive prefix exceeds pos: the wavefront's first cell by a scalar binary search (v_readlane, no LDS),
    // then every lane counts the few team boundaries inside the wavefront's range
    const int wpos0 = __builtin_amdgcn_readfirstlane(tid >> 6) * 64 * q;
    int team = 0, t_excl = 0;
    if (wpos0 < total) {
        const int wlast = min(wpos0 + 64 * q, total) - 1;
        int ulo = 0, uhi = n_teams - 1;
        while (ulo < uhi) {
            const int mid = (ulo + uhi) >> 1;
            if (__builtin_amdgcn_readlane(tincl, mid) > wpos0) uhi = mid; else ulo = mid + 1;
        }
        team = ulo;
        t_excl = ulo > 0 ? __builtin_amdgcn_readlane(tincl, ulo - 1) : 0;
        const int ppos = min(pos, total - 1);
        for (int t = ulo; t < n_teams - 1; t++) {
            const int v = __builtin_amdgcn_readlane(tincl, t);
            if (v > wlast) break;
            if (ppos >= v) { team = t + 1; t_excl = v; }
        }
    }
#ifdef FL_OBS_TIMING
    const long long dbg_t1 = (long long)wall_clock64();
    int dbg_skip = 0;
#endif
    if (pos < end) {
        constexpr int cap = PB >= 2 ? 32 : CAP;   // words between the fields of a node table
        const int *vs = team_table<PB, CAP>(X, scr0, team_words, team);
        int nn = team_meta[64 + team];
        int handle = pb_handle<PB>(X, team_meta, team);
        // first node of the team whose inclusive prefix exceeds the team-local position: three pivots per round trip
        const int lpos = pos - t_excl;
        int lo = 0, hi = nn - 1;
        while (lo < hi) {
            const int m2 = (lo + hi) >> 1, m1 = (lo + m2) >> 1, m3 = (m2 + 1 + hi) >> 1;
            const int i1 = nt_r(vs, cap, N_INCL, m1) & 0xFFFFFF, i2 = nt_r(vs, cap, N_INCL, m2) & 0xFFFFFF, i3 = nt_r(vs, cap, N_INCL, m3) & 0xFFFFFF;
            if (i1 > lpos) hi = m1;
            else if (i2 > lpos) { lo = m1 + 1; hi = m2; }
            else if (i3 > lpos) { lo = m2 + 1; hi = m3; }
            else lo = min(m3 + 1, hi);
        }
        int node = lo;
        // what changes with the team: the walking agent's target and time per cell (pt_of) and, with the trees of both builders
        // in one pass, whose rules and whose prediction index apply
        int target;
        double tq;
        bool cu;
        const unsigned long long *tmask_t = X.tmask, *tmask2_t = X.tmask_m2;
        int Tn_t = X.Tn, tshift_t = X.tshift;
        const bool self_filter = PB >= 2 && X.tmask_m2 != nullptr;
        const uint16_t *path_t = X.path;
        int lp_t = 0, tpc_t = 1;
        auto enter_team = [&]() __attribute__((always_inline)) {
            target = X.a_target[handle];
            cu = pb_cu<PB>(X, team);
            tq = (PB == 2 && !cu) ? X.a_tq2[handle] : X.a_tq[handle];
            if (PB == 2) { tmask_t = cu ? X.tmask : X.u_tmask; Tn_t = cu ? X.Tn : X.u_Tn; tshift_t = cu ? X.tshift : X.u_tshift; }
            if (PB >= 2 && self_filter) {
                tmask2_t = cu ? X.tmask_m2 : X.u_tmask_m2;
                path_t = X.path + (size_t)handle * X.pred_cap;
                lp_t = cu ? X.a_lp[handle] : X.a_lp2[handle];
                tpc_t = cu ? X.a_tpc[handle] : X.a_tpc2[handle];
            }
        };
        enter_team();
        // state of the piece being walked
        int left, cell, tot, nxt;
        uint32_t dd;
        {
            const uint32_t inw = (uint32_t)nt_r(vs, cap, N_INCL, node), tvw = (uint32_t)nt_r(vs, cap, N_TV, node);
            const int nvis = nt_vis(tvw), incl = (int)(inw & 0xFFFFFFu);
            const int k = lpos - (incl - nvis);  // offset inside the node's walk
            const uint32_t st = skip_cells(X, (uint32_t)nt_r(vs, cap, N_SE, node) & 0xFFFFu, k);
            cell = (int)(st >> 2); dd = st & 3u;
#ifdef FL_OBS_TIMING
            dbg_skip = k;
#endif
            tot = nt_tot(tvw) + k;
            left = nvis - k;
            nxt = (int)(inw >> 24);
        }
#ifdef FL_OBS_TIMING
        if (X.dbg && lane == 0) atomicMax((unsigned long long *)&X.dbg[24], (unsigned long long)((long long)wall_clock64() - dbg_t1));
#endif
        // The common case -- time masks, successor table, keys = rail indices -- gets its own copy of the loop, without the
        // tests for what is there
        auto walk = [&](auto fast_tag) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const bool has_snext = FAST || X.snext != nullptr, has_tmask = FAST || X.tmask != nullptr;
        // what the loop body needs of the current cell, requested one iteration ahead
        uint32_t cw = 0, sn = 0, ct = 0, n_inw = 0, own_w = 0, n_se = 0, n_tv = 0;
        unsigned long long tm = 0, tm2 = 0;
        int c_hi = 0, c_lo = 0;
        auto request = [&]() __attribute__((always_inline)) {
            if (FAST && self_filter) own_w = path_t[min(tot, lp_t)];  // HBM (L2): the longest latency first (an LDS copy of the paths made no difference)
            cw = cw_load(X, cell);
            if (has_snext) sn = X.snext[((uint32_t)cell << 2) | dd];
            if (FAST || X.Tn > 0) {
                const int key = FAST ? cell : key_of(X, cell);
                if (has_tmask) { tm = tmask_t[key]; if (FAST && self_filter) tm2 = tmask2_t[key]; }
                else { c_hi = X.csr_end[key]; c_lo = key > 0 ? X.csr_end[key - 1] : 0; }
            }
            if (PB != 1 && PB != 3) ct = X.cell_target[cell >> 5];
            if (left == 1 && nxt < nn) {  // the walk ends on this cell: descriptor of the team's next node with cells
                n_se = (uint32_t)nt_r(vs, cap, N_SE, nxt); n_tv = (uint32_t)nt_r(vs, cap, N_TV, nxt);
                n_inw = (uint32_t)nt_r(vs, cap, N_INCL, nxt);
            }
        };
        request();
        // ONE loop over the lane's cells (lanes of a wave run it in lock step)
        while (true) {
            const int e_cell = cell, e_tot = tot, e_node = node, e_handle = handle;
            const uint32_t e_dd = dd;
            const bool e_cu = cu;
            int *sc = const_cast<int *>(vs);
            constexpr int e_cap = cap;
            const uint2 entry = make_uint2(((uint32_t)cell << 2) | dd | ((uint32_t)team << 24), (uint32_t)tot | ((uint32_t)node << 24));
            // occupant?
            const uint32_t sl = cw >> 16;
            const bool occ = sl != 0xFFFFu;
            // somebody predicted on this key around the queried time?
            bool cand = false;
            int pt = 0;
            if ((FAST || X.Tn > 0) && tot < Tn_t) {
                pt = cu ? (int)((float)tot * (float)tq) : (int)((double)tot * tq);
                if (pt < Tn_t) {
                    if (has_tmask) {  // buckets of the times pt - 1 .. pt + 1: at most three consecutive bits from b1 on
                        const int b1 = tb_of(max(pt - 1, 0), tshift_t), b2 = tb_of(min(pt + 1, Tn_t - 1), tshift_t);
                        unsigned long long others = tm;
                        if (FAST && self_filter && tot >= 1 && tot <= lp_t && (int)(own_w >> 2) == cell) {
                            // this cell is waypoint tot of the walking agent's own path: the buckets of that item (same formulas as
                            // the fill of the index) count only where a second item covers them too
                            const int tlast = Tn_t - 1;
                            const int tlo = cu ? (tot - 1) * tpc_t + 1 : tot * tpc_t, te = tlo + tpc_t - 1;
                            const int thi = (tot == lp_t || te >= tlast) ? tlast : te;
                            const int o1 = tb_of(tlo, tshift_t), o2 = tb_of(thi, tshift_t);
                            others = (tm & ~(((2ull << o2) - 1ull) & ~((1ull << o1) - 1ull))) | tm2;
                        }
                        cand = ((uint32_t)(others >> b1) & ((2u << (b2 - b1)) - 1u)) != 0u;
                    } else {
                        cand = c_hi > c_lo;
                    }
                }
            }
            // somebody's target (upstream only: cutils never fills the map, treeobs.cpp:72)
            const bool tgt_hit = !cu && ((ct >> (cell & 31)) & 1u) && cell != target;
            // reserve the work-list slots of the wavefront (without time masks the conflicts are handled in place)
            const bool to_cf = cand && has_tmask;
            const unsigned long long m_occ = __ballot(occ), m_cf = __ballot(to_cf);
            const unsigned long long resv = wl_reserve2_issue(X.wl_cnt, m_occ, m_cf);
            // advance to the next cell and request its data
            pos++;
            left--;
            const bool more = pos < end;
            if (more) {
                if (left > 0) {  // keep walking along the only transition
                    const uint32_t s2 = has_snext ? sn : chain_next(X, ((uint32_t)cell << 2) | dd, cw & 0xFFFFu);
                    cell = (int)(s2 >> 2); dd = s2 & 3u;
                    tot += 1;
                } else {
                    if (nxt < nn) {
                        node = nxt;
                    } else {  // next team with cells (the prefix says cells remain)
                        do { team++; } while (team < n_teams - 1 && team_meta[team] == 0);
                        vs = team_table<PB, CAP>(X, scr0, team_words, team);
                        nn = team_meta[64 + team];
                        handle = pb_handle<PB>(X, team_meta, team);
                        node = team_meta[256 + team];
                        enter_team();
                        n_se = (uint32_t)nt_r(vs, cap, N_SE, node); n_tv = (uint32_t)nt_r(vs, cap, N_TV, node);
                        n_inw = (uint32_t)nt_r(vs, cap, N_INCL, node);
                    }
                    cell = (int)((n_se & 0xFFFFu) >> 2); dd = n_se & 3u;
                    tot = nt_tot(n_tv);
                    left = nt_vis(n_tv);
                    nxt = (int)(n_inw >> 24);
                }
                request();
            }
            // file the current cell
            int i_occ, i_cf;
            wl_reserve2_finish(resv, m_occ, m_cf, i_occ, i_cf);
            if (occ) {
                if (i_occ < X.wl_occ_cap) wl_occ_put(X, i_occ, entry);
                else occ_event(X, e_cu, sc, e_cap, e_node, sl, e_dd, e_tot);  // list full
            }
            if (to_cf) {
                if (i_cf < X.wl_cf_cap) wl_cf_put(X, i_cf, entry);
                else conflict_event<PB, ITL, TWO>(X, e_cu, sc, e_cap, e_node, e_handle, e_cell, e_dd, e_tot, pt);  // list full
            } else if (cand) {
                conflict_event<PB, ITL, TWO>(X, e_cu, sc, e_cap, e_node, e_handle, e_cell, e_dd, e_tot, pt);
            }
            if (tgt_hit) atomicMin(&nt_w(sc, e_cap, N_OT, e_node), e_tot);
            if (!more) break;
        }
        };
        if (PB >= 2 || (X.tmask != nullptr && X.snext != nullptr && X.rkey == nullptr)) walk(std::true_type());  // PB 2 / 3: the launcher saw to it
        else walk(std::false_type());
    }
#ifdef FL_OBS_TIMING
    if (X.dbg && lane == 0) {
        const long long dbg_t2 = (long long)wall_clock64();
        // slowest lane of the env: slice-loop ticks << 40 | cells per lane << 20 | cells skipped
        atomicMax((unsigned long long *)&X.dbg[25], ((unsigned long long)(dbg_t2 - dbg_t1) << 40) | ((unsigned long long)q << 20) | (unsigned long long)dbg_skip);
        atomicMax((unsigned long long *)&X.dbg[26], (unsigned long long)total);
    }
#endif
    __syncthreads();
    TREE_STAMP(X, 11);
    WAVE_MARK(X, 18, -1);
#ifdef FL_OBS_TIMING
    const long long wl_t0_ = (long long)wall_clock64();
#endif
    // step 2: one list entry per lane
    const int n_occ = min(X.wl_cnt[0], X.wl_occ_cap), n_cf = min(X.wl_cnt[1], X.wl_cf_cap);
#ifdef FL_OBS_TIMING
    if (X.dbg && tid == 0) { X.dbg[X.dbg_base + 9] += n_occ; X.dbg[X.dbg_base + 10] += n_cf; }
#endif
    for (int e = tid; e < n_occ; e += nt) {
        const uint2 w = wl_occ_get(X, e);
        const int cell = (int)((w.x & 0xFFFFFFu) >> 2), team = (int)(w.x >> 24);
        constexpr int cap = PB >= 2 ? 32 : CAP;
        int *sc = team_table<PB, CAP>(X, scr0, team_words, team);
        occ_event(X, pb_cu<PB>(X, team), sc, cap, (int)(w.y >> 24), cw_slot(X, cell), w.x & 3u, (int)(w.y & 0xFFFFFFu));
    }
    WAVE_MARK(X, 12, -1);
    WL_ACC(X, 0);
    if (*X.long_lists == 0) {  // every list is short: one pass, every lane scans the list of its entry
        for (int e = tid; e < n_cf; e += nt) {
            const uint2 w = wl_cf_get(X, e);
            const int cell = (int)((w.x & 0xFFFFFFu) >> 2), team = (int)(w.x >> 24);
            const int handle = pb_handle<PB>(X, team_meta, team), tot = (int)(w.y & 511u);
            const bool cu = pb_cu<PB>(X, team);
            const int pt = pt_of<PB>(X, cu, handle, tot);
            const ListRange R = list_range<PB>(X, cu, cell, pt);
#ifdef FL_OBS_COUNTS
            const uint32_t f = R.n > 0 ? conflict_flags<PB, ITL, TWO>(X, cu, handle, cell, w.x & 3u, pt, R, 0, R.n) : 0u;
            if (X.dbg) {
                atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 27], (unsigned long long)R.n);
                if (tb_of(pt, X.tshift) >= 63) atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 28], 1ull);
                if (f & 7u) atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 29], 1ull);
                if (conflict_hit(f)) atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 30], 1ull);
                if (!(f & 7u) && (f & 64u)) atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 31], 1ull);
            }
            if (conflict_hit(f)) {
#else
            if (R.n > 0 && conflict_hit(conflict_flags<PB, ITL, TWO>(X, cu, handle, cell, w.x & 3u, pt, R, 0, R.n))) {
#endif
                constexpr int cap = PB >= 2 ? 32 : CAP;
                int *sc = team_table<PB, CAP>(X, scr0, team_words, team);
                atomicMin(&nt_w(sc, cap, N_PC, (int)(w.y >> 24)), tot);
            }
        }
        WAVE_MARK(X, 13, 17);
        WL_ACC(X, 1);
        late();
        WL_ACC(X, 2);
        __syncthreads();
        WL_ACC(X, 5);
        return;
    }
    constexpr int CF_CHUNK = ITL ? CF_CHUNK_LDS : CF_CHUNK_HBM;
    // ... and the lane of the first entry takes CF_FIRST items itself (more than one chunk's worth measured slower: the wavefront waits
    // for its longest scan)
    constexpr int CF_FIRST = ITL ? CF_CHUNK_LDS : CF_FIRST_HBM;
    // One entry per CF_CHUNK items of a key's list, so that no lane scans a long list alone.  Every lane scans the FIRST chunk of
    // its entry right away (with lists grouped by time bucket that is the whole list of most queries: no second look at the entry,
    // its offsets and its flag word -- which are HBM round trips on large maps); a longer list pushes an entry per further chunk,
    // and those are scanned after a barrier, one per lane on densely packed wavefronts.
    // First entry: tot | further chunks still to come << 9 | flags so far << 15 | node << 24; the others: chunk | index of the first
    // entry << 6 (17 bits) | tot << 23.  Whoever scans a further chunk ORs its flags into the first entry's word and counts it down
    // with ONE returning atomic each (same address, program order); the lane that takes the count to zero has all the flags in the
    // value it got back and files the result -- no pass over all entries afterwards (round 5: that pass and its barrier were 8.7 us
    // of cfg3's 54 us work-list step and 90 of cfg5's 390).
    for (int e0 = 0; e0 < n_cf; e0 += nt) {
        const int e = e0 + tid;
        int nch = 0, cell = 0, handle = 0, tot = 0, pt = 0, team = 0, np = 0;
        ListRange R = {0, 0, 0, 0};
        bool cu = PB == 1;
        uint2 w = make_uint2(0u, 0u);
        uint32_t fl = 0;
        if (e < n_cf) {
            w = wl_cf_get(X, e);
            cell = (int)((w.x & 0xFFFFFFu) >> 2);
            team = (int)(w.x >> 24);
            handle = pb_handle<PB>(X, team_meta, team);
            tot = (int)(w.y & 511u);
            cu = pb_cu<PB>(X, team);
            pt = pt_of<PB>(X, cu, handle, tot);
            R = list_range<PB>(X, cu, cell, pt);
            nch = 1 + min(max(R.n - CF_FIRST + CF_CHUNK - 1, 0) / CF_CHUNK, 62);  // an absurdly long list: the last chunk takes the rest
            const uint32_t f = R.n > 0 ? conflict_flags<PB, ITL, TWO>(X, cu, handle, cell, w.x & 3u, pt, R, 0, min(R.n, CF_FIRST)) : 0u;
#ifdef FL_OBS_COUNTS  // with FL_OBS_TIMING: statistics of the conflict entries (they slow the step down; first chunks only)
            if (X.dbg) {
                atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 27], (unsigned long long)R.n);
                if (tb_of(pt, X.tshift) >= 63) atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 28], 1ull);
                if (f & 7u) atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 29], 1ull);
                if (conflict_hit(f)) atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 30], 1ull);
                if (!(f & 7u) && (f & 64u)) atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 31], 1ull);
            }
#endif
            fl = f & 63u;
        }
        for (int j = 1; __any(j < nch); j++) {
            const bool want = j < nch;
            if (!wl_push_cf(X, want, make_uint2(w.x, (uint32_t)j | ((uint32_t)e << 6) | ((uint32_t)tot << 23)))) {
                // list full: this chunk is scanned here
                fl |= conflict_flags<PB, ITL, TWO>(X, cu, handle, cell, w.x & 3u, pt, R, CF_FIRST + (j - 1) * CF_CHUNK, j == 62 ? R.n : min(R.n, CF_FIRST + j * CF_CHUNK)) & 63u;
            } else if (want) np++;
        }
        if (e < n_cf) {
            if (np == 0) {  // nothing left to others
                if (conflict_hit(fl)) {
                    constexpr int cap = PB >= 2 ? 32 : CAP;
                    int *sc = team_table<PB, CAP>(X, scr0, team_words, team);
                    atomicMin(&nt_w(sc, cap, N_PC, (int)(w.y >> 24)), tot);
                }
            } else {
                wl_cf_set_y(X, e, w.y | ((uint32_t)np << 9) | (fl << 15));
            }
        }
    }
    WAVE_MARK(X, 13, 17);
    WL_ACC(X, 1);
    late();
    WL_ACC(X, 2);
    __syncthreads();
    WAVE_MARK(X, 14, -1);
    WL_ACC(X, 3);
    const int n_cf2 = min(X.wl_cnt[1], X.wl_cf_cap);   // (workgroup-uniform)
    for (int e = n_cf + tid; e < n_cf2; e += nt) {  // the further chunks
        const uint2 w = wl_cf_get(X, e);
        const int cell = (int)((w.x & 0xFFFFFFu) >> 2), team = (int)(w.x >> 24);
        const int first = (int)((w.y >> 6) & 0x1FFFFu), chunk = (int)(w.y & 63u), tot = (int)(w.y >> 23);
        const int handle = pb_handle<PB>(X, team_meta, team);
        const bool cu = pb_cu<PB>(X, team);
        const int pt = pt_of<PB>(X, cu, handle, tot);
        const ListRange R = list_range<PB>(X, cu, cell, pt);
        const uint32_t f = conflict_flags<PB, ITL, TWO>(X, cu, handle, cell, w.x & 3u, pt, R, CF_FIRST + (chunk - 1) * CF_CHUNK, chunk == 62 ? R.n : min(R.n, CF_FIRST + chunk * CF_CHUNK));
        const uint32_t old = wl_cf_done(X, first, (f & 63u) << 15);
        if (((old >> 9) & 63u) == 1u && conflict_hit((old >> 15) & 63u)) {  // the last chunk of its key
            constexpr int cap = PB >= 2 ? 32 : CAP;
            int *sc = team_table<PB, CAP>(X, scr0, team_words, team);
            atomicMin(&nt_w(sc, cap, N_PC, (int)(old >> 24)), (int)(old & 511u));
        }
    }
    WL_ACC(X, 4);
    WL_ACC(X, 5);
    __syncthreads();
}

