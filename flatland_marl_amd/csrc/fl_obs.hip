// fl_obs.hip -- tree-observation builders for B envs, one workgroup per env.
//
// Replaces (paths relative to /root/reference):
//   flatland_cutils/src/loader.cpp:221-327      AgentsLoader::update (snapshot, dist_target, road_type, valid actions)
//   flatland_cutils/src/deadlock_checker.cpp    DeadlockChecker (restated as a least fixpoint, see k_obs phase 1)
//   flatland_cutils/src/predictions.cpp:78-235  shortest-path predictor (greedy strict descent on the distance map)
//   flatland_cutils/src/treeobs.cpp:30-610      get_many / get / _explore_branch / scale_node
//   flatland_cutils/src/tool.h:468-524          calculate_evaluation_orders
//   flatland_cutils/src/feature_parser.cpp:3-98 AgentAttrParser::get_features
//   flatland-rl/flatland/envs/observations.py:60-494 + predictions.py:97-180   upstream TreeObsForRailEnv
//
// Everything is indexed by RAIL CELLS (rail index r, rail state s = r * 4 + orientation; fl_internal.h), not by grid cells:
// the per-cell words, the neighbour / successor tables, the prediction keys and their time masks of a 150x150 map
// (2680 rail cells) fit LDS like those of a 30x30 one; on small maps the static distance / segment / next-hop tables of
// the env are staged in LDS too (TAB_LDS), so that no gather of the kernel leaves the CU.
//
// Layout of one launch (gfx950): one workgroup (up to 16 wavefronts) per env.  The env's rail words, neighbour tables and
// an occupied-cell table are staged in LDS once.  Then, concurrently: eight lanes per agent walk its predicted path (static next-hop /
// eight-hop tables), one wavefront does the per-agent part (deadlock fixpoint, valid actions, 83-float attribute row) and
// the other wavefronts derive the topology of the trees from the static segment table (pass A).  A per-key index of
// prediction items (+ per-key time-bucket masks) is built in LDS.  Pass B splits the visited cells of all trees evenly over
// all lanes, classifies them, and processes the few cells that need work from LDS work lists on packed wavefronts; rows
// are written straight to HBM.  DESIGN.md section 4 describes the phases; tools/obs_phase_clocks.py measures them.
#include "fl_obs.h"

#include <stdio.h>
#include <type_traits>
#include <stdlib.h>
#include <string.h>

#include "../../include/flatland_hip.h"

#define OBS_NT 1024
#ifndef CF_CHUNK
#define CF_CHUNK 16                  // items of a key's list scanned per conflict work-list entry
#endif
#ifndef OBS_GLB_BATCH
#define OBS_GLB_BATCH 8              // items per round trip when the prediction items live in HBM scratch
#endif
#ifndef OBS_WL_OCC_DIV
#define OBS_WL_OCC_DIV 6            // occupant work list = 1 / OBS_WL_OCC_DIV of the work-list entries, conflicts get the rest
#endif
#ifndef OBS_TSHIFT
#define OBS_TSHIFT 1                 // time-bucket width of the per-key masks for long horizons: 1 << OBS_TSHIFT steps
#endif
#define CF_MORE 0x800000u            // work-list entry of a further chunk (see wg_pass_b)
#define CF_DIRECT 32                 // when no list of the env is longer, every conflict entry is scanned by its lane alone, in one pass
#define OBS_ITEMS2_CAP 2048          // items of the second (upstream) index built by stage 1 of the fused launch
#define OBS_ITEMS_LDS_CAP 6144       // prediction items are kept in LDS when an env has at most this many (else HBM scratch)
#define OBS_WL_HBM_ENTRIES 32768     // pass B work-list entries per env when the lists live in HBM scratch (large maps)
// Large maps (items in HBM, hundreds of agents): inside a key's list the items are grouped by bucket of 64 time steps, an
// item sits in every bucket its interval touches, and a conflict query scans only the buckets its three time steps fall in
// (an eighth of a busy cell's list instead of all of it).
#define OBS_BK_NB 8
#define OBS_BK_SHIFT 6

// prediction item: one (agent, waypoint) with the closed time interval during which the agent is predicted there
//   bits 0-1 direction at the waypoint, 2-3 direction at the next waypoint, 4-5 at the previous one,
//   6-9 interval length - 1, 10 "until the end of the horizon", 11-19 t_lo, 20-29 agent
#define IT_DIR(it) ((it)&3u)
#define IT_DNEXT(it) (((it) >> 2) & 3u)
#define IT_DPREV(it) (((it) >> 4) & 3u)
#define IT_THI(it, tlast) ((((it) >> 10) & 1u) ? (uint32_t)(tlast) : IT_TLO(it) + (((it) >> 6) & 15u))
#define IT_TLO(it) (((it) >> 11) & 511u)
#define IT_TOEND(it) (((it) >> 10) & 1u)
#define IT_AGENT(it) ((int)((it) >> 20))

// ---------------------------------------------------------------------------------------------- context
struct ObsCtx {
    int A, R;
    int SS;                       // stride (in rail states) between the per-target slabs of dm / hop8
    const uint32_t *cellw;        // LDS per rail cell: rail bitmap (low 16) | occupied-cell table index (high 16, 0xFFFF = none)
    const uint16_t *nbr;          // LDS [R * 4]: rail index of the neighbour in direction m, FL_R_NONE
    const uint16_t *snext;        // LDS [R * 4] successor of a single-transition state, or nullptr (derived from cellw + nbr)
    const uint16_t *rkey;         // LDS [R] compact prediction key (col * W + row collides when H > W, tool.h:391-398); nullptr: key = r
    const int *slot_agent;        // LDS: highest on-map handle on the cell (last writer of location_has_agent*), -1
    const int *slot_ready;        // LDS: number of off-map agents whose initial position is the cell
    const uint32_t *cell_target;  // LDS bitmap over rail cells: some agent's target (upstream location_has_target)
    const uint16_t *a_vpos;       // LDS per agent: virtual position (rail index)
    const uint8_t *a_dir, *a_state;
    const uint16_t *a_malf;       // real down counter
    const double *a_speed;
    const uint16_t *a_tpc;        // times per cell of the predictor
    const double *a_tq;           // time per cell of the tree walk: float 1.0 / speed of cutils (treeobs.cpp:304, held exactly
                                  // in a double) or np.reciprocal(speed) of the upstream builder (observations.py:277)
    const uint16_t *a_tslot;
    const uint16_t *a_target;     // rail index
    const int *csr_end;           // LDS [K] end offset of key k's item list (start = csr_end[k-1], 0 for k = 0)
    const uint32_t *items_lds;    // IT_* packed items when they fit LDS ...
    const uint32_t *items_glb;    // ... else in HBM scratch (two members so that each keeps a static address space)
    const uint16_t *bk_rel;       // HBM [K * OBS_BK_NB]: end of time bucket b inside key k's list, relative to the list's start;
                                  // nullptr = lists not bucketed
    int Tn;                       // number of predicted time entries (0 = no predictor)
    const uint16_t *dm;           // env base [U][SS] distance map (LDS copy when TAB_LDS, else HBM)
    const uint4 *seg;             // env base [S] static branch-walk table (LDS copy when TAB_LDS, else HBM)
    // pass B work lists (LDS): cells with an occupant / cells whose key has a prediction near the queried time
    uint2 *wl_occ, *wl_cf;
    int wl_occ_cap, wl_cf_cap;
    bool wl_hbm;                  // the lists live in HBM scratch: their flag words are merged with L2 atomics, read them past the L1
    int *wl_cnt;                  // LDS [3] entries pushed to wl_occ / wl_cf, flag: some key needs the second conflict pass
    const int *long_lists;        // LDS flag: some key's list has more than CF_DIRECT items (else no conflict query needs chunks)
    const unsigned long long *tmask;  // LDS per key: time buckets min(t >> tshift, 63) covered by some item; nullptr = none
    int tshift;
    // pass B over the trees of BOTH builders at once (PB = 2): teams below n_cu are flatland_cutils trees and use the members
    // above, the others are upstream trees and use the upstream predictor's index:
    int n_cu;
    const int *u_csr_end;
    const uint32_t *u_items;
    const unsigned long long *u_tmask;
    int u_Tn, u_tshift;
    const double *a_tq2;          // np.reciprocal(speed)
    // Own-path filter of the classify loop (small envs): a cell that is waypoint tot of the walking agent's own predicted
    // path always has that agent's own item around the queried time.  tmask_m2 / u_tmask_m2 = buckets covered by at least
    // TWO items of the key, so the own item's buckets can be taken out of the test exactly; nullptr = no filter.
    const unsigned long long *tmask_m2, *u_tmask_m2;
    const uint16_t *path;         // HBM [A][pred_cap] predicted paths of the env (state per waypoint)
    int pred_cap;
    const uint16_t *a_lp, *a_lp2, *a_tpc2;  // last waypoint in the first / second index, times per cell of the second
    long long *dbg;               // diagnostic builds
    int dbg_base;
};

__device__ __forceinline__ uint32_t cw_bits(const ObsCtx &X, int r) { return X.cellw[r] & 0xFFFFu; }
// occupied-cell table index of the cell, 0xFFFF = nobody on it and nobody waiting to depart from it
__device__ __forceinline__ uint32_t cw_slot(const ObsCtx &X, int r) { return X.cellw[r] >> 16; }
__device__ __forceinline__ uint32_t cw_load(const ObsCtx &X, int r) { return X.cellw[r]; }
__device__ __forceinline__ int key_of(const ObsCtx &X, int r) { return X.rkey ? (int)X.rkey[r] : r; }
// Pass B serves one builder (PB 0 = upstream, 1 = flatland_cutils) or both in one pass (PB 2): cu says which builder's
// rules apply to a team
template <int PB>
__device__ __forceinline__ bool pb_cu(const ObsCtx &X, int team) { return PB == 2 ? team < X.n_cu : PB == 1; }
// the agent of a team: from the team table, or -- both builders in one pass -- from the numbering of trees_merged (no memory access)
template <int PB>
__device__ __forceinline__ int pb_handle(const ObsCtx &X, const int *team_meta, int team) {
    return PB == 2 ? (team < X.n_cu ? team : team - X.n_cu) : team_meta[128 + team];
}
// predicted time at which the walking agent reaches a cell tot steps away (treeobs.cpp:378 / observations.py:329)
template <int PB>
__device__ __forceinline__ int pt_of(const ObsCtx &X, bool cu, int handle, int tot) {
    if (PB == 2 && !cu) return (int)((double)tot * X.a_tq2[handle]);
    return cu ? (int)((float)tot * (float)X.a_tq[handle]) : (int)((double)tot * X.a_tq[handle]);
}
// items of rail cell r's key a conflict query at predicted time pt has to look at: [lo, hi)
template <int PB>
__device__ __forceinline__ void list_range(const ObsCtx &X, bool cu, int r, int pt, int &lo, int &hi) {
    const int key = key_of(X, r);
    if (PB == 2 && !cu) {
        lo = key > 0 ? X.u_csr_end[key - 1] : 0;
        hi = X.u_csr_end[key];
        return;
    }
    const int base = key > 0 ? X.csr_end[key - 1] : 0;
    if (X.bk_rel) {
        const int b1 = min(max(pt - 1, 0) >> OBS_BK_SHIFT, OBS_BK_NB - 1), b2 = min(min(pt + 1, X.Tn - 1) >> OBS_BK_SHIFT, OBS_BK_NB - 1);
        const uint16_t *rel = X.bk_rel + (size_t)key * OBS_BK_NB;
        lo = base + (b1 > 0 ? (int)rel[b1 - 1] : 0);
        hi = base + (int)rel[b2];
    } else {
        lo = base;
        hi = X.csr_end[key];
    }
}
// successor of a state with exactly one transition (chain interior): one LDS load when the table is resident
__device__ __forceinline__ uint32_t chain_next(const ObsCtx &X, uint32_t s, uint32_t bits16) {
    if (X.snext) return X.snext[s];
    const uint32_t nd = first_dir(nibble(bits16, s & 3u));
    return ((uint32_t)X.nbr[(s & ~3u) | nd] << 2) | nd;
}
// state reached by leaving rail cell r in direction m, -1 when there is no rail there
__device__ __forceinline__ int state_towards(const ObsCtx &X, int r, uint32_t m) {
    const uint32_t nr = X.nbr[r * 4 + (int)m];
    return nr == FL_R_NONE ? -1 : (int)((nr << 2) | m);
}

// One node of a tree = one branch walk (_explore_branch: treeobs.cpp:258-610 / observations.py:256-494).
// Where the walk ends, how long it is and its first "unusable switch" are static per start state (segment table,
// fl_dmap.hip); the stop at the agent's own target follows from the distance map: along a chain of single-transition
// cells the distance drops by one per step, so the target is on the chain iff dm[start] <= chain length.
struct NodeDesc {
    int start;      // start state cell << 2 | dir, -1 = null cell
    int tot0;       // tot_dist at the first visited cell
    int nvis;       // number of visited cells (feature block executions)
    int end;        // end state (direction unknown / irrelevant when the walk stops at the target)
    uint32_t flags; // bit 0 target stop, 1 switch, 2 dead end, 3 terminal (zero transition or cycle), 4 zero transition
    int unus;       // tot_dist of the first unusable switch or -1
    uint32_t kids01, kids23;  // start states of the end state's children (u16 each, FL_R_NONE = null), valid for switch / dead end
};
enum { ND_TARGET = 1, ND_SWITCH = 2, ND_DEAD_END = 4, ND_TERMINAL = 8, ND_ZERO = 16 };

// dm_t = the distance slab of the agent's target, target = its rail index (per-agent constants, hoisted by the callers)
__device__ __forceinline__ NodeDesc node_topology(const ObsCtx &X, const uint16_t *dm_t, int target, int start, int tot0) {
    NodeDesc n;
    n.start = start;
    n.tot0 = tot0;
    const uint4 e = X.seg[start];
    const uint32_t dv = dm_t[start];
    const int len = SEG_LEN(e), unus = SEG_UNUS(e);
    n.kids01 = e.z; n.kids23 = e.w;
    if (dv != FL_INF16 && (int)dv <= len) {  // reaches its own target first
        n.nvis = (int)dv + 1;
        n.end = target << 2;
        n.flags = ND_TARGET;
        n.unus = (unus != 0xFFFF && unus < (int)dv) ? tot0 + unus : -1;  // the target cell breaks before that check
    } else {
        n.nvis = len + 1;
        n.end = SEG_END(e);
        const uint32_t k = SEG_KIND(e);
        n.flags = k == SEG_SWITCH ? ND_SWITCH : k == SEG_DEAD_END ? ND_DEAD_END : k == SEG_ZERO ? (ND_TERMINAL | ND_ZERO) : ND_TERMINAL;
        n.unus = unus != 0xFFFF ? tot0 + unus : -1;
    }
    return n;
}

// advance k cells along a chain of single-transition cells (no features)
__device__ __forceinline__ uint32_t skip_cells(const ObsCtx &X, uint32_t s, int k) {
    for (int v = 0; v < k; v++) s = chain_next(X, s, X.snext ? 0u : cw_bits(X, (int)(s >> 2)));
    return s;
}

// per-team node table in LDS: CAP entries per field
enum { F_START = 0, F_TOT, F_VIS, F_END, F_FLAGS, F_UNUS, F_PAR, F_HGT, F_INCL, F_OA, F_PC, F_OT, F_SAME, F_OPP, F_MALF,
       F_READY, F_MS /* u64: two ints per node */, F_WORDS = 18 };

// LDS words of the trees' node tables: one slot per team that can hold an agent plus one dummy slot that the idle teams share
__host__ __device__ inline int obs_scr_words(int nwaves, int A, int tw_c, int tw_t, int tpw_t) {
    const int n_c = 2 * nwaves <= A ? 2 * nwaves : A + 1, n_t = tpw_t * nwaves <= A ? tpw_t * nwaves : A + 1;
    const int w_c = n_c * tw_c, w_t = n_t * tw_t;
    return w_c > w_t ? w_c : w_t;
}

// The feature block of ONE visited cell of a branch walk (treeobs.cpp:322-465 / observations.py:296-371) is split in
// two event handlers that merge straight into the node's accumulators (sc = the team's node table) with LDS atomics:
// min / sum / max are associative and tot_dist grows along a walk, so "first hit" = minimum.
//
// occupant of the cell (treeobs.cpp:322-357 / observations.py:296-327)
template <int PB, int CAP>
__device__ __forceinline__ void occ_event(const ObsCtx &X, bool CUTILS, int *sc, int node, uint32_t sl, uint32_t d, int tot) {
    const int ag = X.slot_agent[sl];
    if (ag < 0) return;
    atomicMin(&sc[F_OA * CAP + node], tot);
    const int mf = CUTILS ? (X.a_malf[ag] != 0) : (int)X.a_malf[ag];
    if (mf > 0) atomicMax(&sc[F_MALF * CAP + node], mf);
    const int rd = X.slot_ready[sl];
    const int radd = rd > 0 ? (CUTILS ? rd - 1 : rd) : 0;  // cutils starts the count at 0 (treeobs.cpp:82-91)
    if (radd) atomicAdd(&sc[F_READY * CAP + node], radd);
    if (X.a_dir[ag] == d) {
        atomicAdd(&sc[F_SAME * CAP + node], 1);
        const double sp = CUTILS ? (double)(float)X.a_speed[ag] : X.a_speed[ag];
        unsigned long long *ms = reinterpret_cast<unsigned long long *>(sc + F_MS * CAP);
        if (sp < 1.0) atomicMin(&ms[node], (unsigned long long)__double_as_longlong(sp));  // positive doubles order like their bits
    } else {
        atomicAdd(&sc[F_OPP * CAP + node], 1);
    }
}

// potential conflict at predicted time pt (treeobs.cpp:378-465 / observations.py:329-367); the caller checked
// Tn > 0, tot < Tn and pt < Tn.  conflict_flags scans items [lo, hi) of the cell's key and returns six bits:
// bit k (k = 0, 1, 2 for the times pt, pt - 1, pt + 1): some OTHER agent is predicted there then; bit 3 + k: some agent
// predicted there then (self included) satisfies the conflict condition.  Flags of sub-ranges of a list simply OR.
template <int PB, bool ITL>
__device__ __forceinline__ uint32_t conflict_flags(const ObsCtx &X, bool CUTILS, int handle, int cell, uint32_t d, int pt, int lo, int hi) {
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
        for (int e0 = lo; e0 < hi; e0 += NB) {
            uint32_t itv[NB];
#pragma unroll
            for (int q = 0; q < NB; q++) itv[q] = items[min(e0 + q, hi - 1)];
#pragma unroll
            for (int q = 0; q < NB; q++) {
                const uint32_t tl = IT_TLO(itv[q]), th = IT_THI(itv[q], tlast);
                if (e0 + q < hi && th >= t1 && tl <= t2) test_item(itv[q]);
            }
        }
    };
    // separate call sites so that each keeps a static address space; whole lists in HBM scratch (large maps without time
    // masks) are fetched in bigger batches: their round trips are what the scan costs
    if (ITL) scan(second ? X.u_items : X.items_lds, std::integral_constant<int, 8>());
    else if (X.tmask) scan(X.items_glb, std::integral_constant<int, CF_CHUNK>());  // chunked work-list entries: the whole chunk in flight at once
    else scan(X.items_glb, std::integral_constant<int, OBS_GLB_BATCH>());
    return flags;
}
// the other-agent test takes the first time (pt, pt - 1, pt + 1) at which somebody else is predicted on the cell
__device__ __forceinline__ bool conflict_hit(uint32_t f) { return (f & 1u) ? (f >> 3) & 1u : ((f & 2u) ? (f >> 4) & 1u : ((f & 4u) ? (f >> 5) & 1u : false)); }

template <int PB, int CAP, bool ITL>
__device__ __forceinline__ void conflict_event(const ObsCtx &X, bool cu, int *sc, int node, int handle, int cell, uint32_t d, int tot, int pt) {
    int lo, hi;
    list_range<PB>(X, cu, cell, pt, lo, hi);
    if (hi <= lo) return;
    if (conflict_hit(conflict_flags<PB, ITL>(X, cu, handle, cell, d, pt, lo, hi))) atomicMin(&sc[F_PC * CAP + node], tot);
}

// flag word of a conflict work-list entry (other lanes OR their bits into it)
__device__ __forceinline__ uint32_t wl_flags(const ObsCtx &X, const uint2 *e) {
    if (X.wl_hbm) return __hip_atomic_load(&e->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return e->y;
}

// append e to a work list; one LDS atomic per wavefront.  false = the list is full and the caller handles the event itself
__device__ __forceinline__ bool wl_push(uint2 *list, int cap, int *count, bool want, uint2 e, int *idx_out = nullptr) {
    const unsigned long long m = __ballot(want);
    if (m == 0) return true;
    const int lane = (int)__lane_id();
    // the first ACTIVE lane reserves the slots for the wavefront; its result is broadcast with v_readfirstlane (a shuffle
    // would be another LDS round trip)
    int base = 0;
    if (lane == __ffsll((long long)__ballot(1)) - 1) base = atomicAdd(count, __popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    const int idx = base + __popcll(m & ((1ull << lane) - 1ull));
    if (want && idx < cap) list[idx] = e;
    if (idx_out) *idx_out = idx;
    return !want || idx < cap;
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
// F_INCL word of node k: inclusive prefix (24 bits) | index of the next node with cells << 24 (0xFF = none).
// Returns the team's number of cells; first_real = its first node with cells (0xFF = none).
template <int TEAM, int CAP>
__device__ __forceinline__ int team_prepare(bool have, int tl, int n_nodes, int *scr, int &first_real) {
    constexpr int NCH = (CAP + TEAM - 1) / TEAM;
    unsigned long long *ms = reinterpret_cast<unsigned long long *>(scr + F_MS * CAP);
    const int tbase = ((int)__lane_id() / TEAM) * TEAM;
    const unsigned long long tbits = TEAM == 64 ? ~0ull : ((1ull << (TEAM & 63)) - 1ull);
    int v[NCH];
    unsigned long long real[NCH];  // bit j: node c * TEAM + j has cells
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int k = c * TEAM + tl;
        v[c] = (have && k < n_nodes && k < CAP && scr[F_START * CAP + k] >= 0) ? scr[F_VIS * CAP + k] : 0;
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
            scr[F_INCL * CAP + k] = incl | (nxt << 24);
            scr[F_OA * CAP + k] = 0x7fffffff; scr[F_PC * CAP + k] = 0x7fffffff; scr[F_OT * CAP + k] = 0x7fffffff;
            scr[F_SAME * CAP + k] = 0; scr[F_OPP * CAP + k] = 0; scr[F_MALF * CAP + k] = 0; scr[F_READY * CAP + k] = 0;
            ms[k] = 0x3FF0000000000000ull;  // 1.0; positive doubles order like their bit patterns
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
template <int PB, int CAP, bool ITL>
__device__ __forceinline__ void wg_pass_b(const ObsCtx &X, int tid, int nt, int n_teams, int *scr0, int team_words,
                                          const int *team_meta) {
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
        const int *vs = scr0 + team * team_words;
        int nn = team_meta[64 + team];
        int handle = pb_handle<PB>(X, team_meta, team);
        // first node of the team whose inclusive prefix exceeds the team-local position: three pivots per round trip
        const int lpos = pos - t_excl;
        int lo = 0, hi = nn - 1;
        while (lo < hi) {
            const int m2 = (lo + hi) >> 1, m1 = (lo + m2) >> 1, m3 = (m2 + 1 + hi) >> 1;
            const int i1 = vs[F_INCL * CAP + m1] & 0xFFFFFF, i2 = vs[F_INCL * CAP + m2] & 0xFFFFFF, i3 = vs[F_INCL * CAP + m3] & 0xFFFFFF;
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
        const bool self_filter = PB == 2 && X.tmask_m2 != nullptr;
        const uint16_t *path_t = X.path;
        int lp_t = 0, tpc_t = 1;
        auto enter_team = [&]() __attribute__((always_inline)) {
            target = X.a_target[handle];
            cu = pb_cu<PB>(X, team);
            tq = (PB == 2 && !cu) ? X.a_tq2[handle] : X.a_tq[handle];
            if (PB == 2) {
                tmask_t = cu ? X.tmask : X.u_tmask; Tn_t = cu ? X.Tn : X.u_Tn; tshift_t = cu ? X.tshift : X.u_tshift;
                if (self_filter) {
                    tmask2_t = cu ? X.tmask_m2 : X.u_tmask_m2;
                    path_t = X.path + (size_t)handle * X.pred_cap;
                    lp_t = cu ? X.a_lp[handle] : X.a_lp2[handle];
                    tpc_t = cu ? X.a_tpc[handle] : X.a_tpc2[handle];
                }
            }
        };
        enter_team();
        // state of the piece being walked
        int left, cell, tot, nxt;
        uint32_t dd;
        {
            const uint32_t inw = (uint32_t)vs[F_INCL * CAP + node];
            const int nvis = vs[F_VIS * CAP + node], incl = (int)(inw & 0xFFFFFFu);
            const int k = lpos - (incl - nvis);  // offset inside the node's walk
            const uint32_t st = skip_cells(X, (uint32_t)vs[F_START * CAP + node], k);
            cell = (int)(st >> 2); dd = st & 3u;
#ifdef FL_OBS_TIMING
            dbg_skip = k;
#endif
            tot = vs[F_TOT * CAP + node] + k;
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
        uint32_t cw = 0, sn = 0, ct = 0, n_inw = 0, own_w = 0;
        unsigned long long tm = 0, tm2 = 0;
        int c_hi = 0, c_lo = 0, n_start = 0, n_tot = 0, n_vis = 0;
        auto request = [&]() __attribute__((always_inline)) {
            if (FAST && self_filter) own_w = path_t[min(tot, lp_t)];  // HBM (L2): the longest latency first (an LDS copy of the paths made no difference)
            cw = cw_load(X, cell);
            if (has_snext) sn = X.snext[((uint32_t)cell << 2) | dd];
            if (FAST || X.Tn > 0) {
                const int key = FAST ? cell : key_of(X, cell);
                if (has_tmask) { tm = tmask_t[key]; if (FAST && self_filter) tm2 = tmask2_t[key]; }
                else { c_hi = X.csr_end[key]; c_lo = key > 0 ? X.csr_end[key - 1] : 0; }
            }
            if (PB != 1) ct = X.cell_target[cell >> 5];
            if (left == 1 && nxt < nn) {  // the walk ends on this cell: descriptor of the team's next node with cells
                n_start = vs[F_START * CAP + nxt]; n_tot = vs[F_TOT * CAP + nxt]; n_vis = vs[F_VIS * CAP + nxt];
                n_inw = (uint32_t)vs[F_INCL * CAP + nxt];
            }
        };
        request();
        // ONE loop over the lane's cells (lanes of a wave run it in lock step)
        while (true) {
            const int e_cell = cell, e_tot = tot, e_node = node, e_handle = handle;
            const uint32_t e_dd = dd;
            const bool e_cu = cu;
            int *sc = scr0 + team * team_words;
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
                        const int b1 = min(max(pt - 1, 0) >> tshift_t, 63), b2 = min(min(pt + 1, Tn_t - 1) >> tshift_t, 63);
                        unsigned long long others = tm;
                        if (FAST && self_filter && tot >= 1 && tot <= lp_t && (int)(own_w >> 2) == cell) {
                            // this cell is waypoint tot of the walking agent's own path: the buckets of that item (same formulas as
                            // the fill of the index) count only where a second item covers them too
                            const int tlast = Tn_t - 1;
                            const int tlo = cu ? (tot - 1) * tpc_t + 1 : tot * tpc_t, te = tlo + tpc_t - 1;
                            const int thi = (tot == lp_t || te >= tlast) ? tlast : te;
                            const int o1 = min(tlo >> tshift_t, 63), o2 = min(thi >> tshift_t, 63);
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
                        vs = scr0 + team * team_words;
                        nn = team_meta[64 + team];
                        handle = pb_handle<PB>(X, team_meta, team);
                        node = team_meta[256 + team];
                        enter_team();
                        n_start = vs[F_START * CAP + node]; n_tot = vs[F_TOT * CAP + node]; n_vis = vs[F_VIS * CAP + node];
                        n_inw = (uint32_t)vs[F_INCL * CAP + node];
                    }
                    cell = n_start >> 2; dd = (uint32_t)n_start & 3u;
                    tot = n_tot;
                    left = n_vis;
                    nxt = (int)(n_inw >> 24);
                }
                request();
            }
            // file the current cell
            int i_occ, i_cf;
            wl_reserve2_finish(resv, m_occ, m_cf, i_occ, i_cf);
            if (occ) {
                if (i_occ < X.wl_occ_cap) X.wl_occ[i_occ] = entry;
                else occ_event<PB, CAP>(X, e_cu, sc, e_node, sl, e_dd, e_tot);  // list full
            }
            if (to_cf) {
                if (i_cf < X.wl_cf_cap) X.wl_cf[i_cf] = entry;
                else conflict_event<PB, CAP, ITL>(X, e_cu, sc, e_node, e_handle, e_cell, e_dd, e_tot, pt);  // list full
            } else if (cand) {
                conflict_event<PB, CAP, ITL>(X, e_cu, sc, e_node, e_handle, e_cell, e_dd, e_tot, pt);
            }
            if (tgt_hit) atomicMin(&sc[F_OT * CAP + e_node], e_tot);
            if (!more) break;
        }
        };
        if (PB == 2 || (X.tmask != nullptr && X.snext != nullptr && X.rkey == nullptr)) walk(std::true_type());  // PB 2: the launcher saw to it
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
    // step 2: one list entry per lane
    const int n_occ = min(X.wl_cnt[0], X.wl_occ_cap), n_cf = min(X.wl_cnt[1], X.wl_cf_cap);
#ifdef FL_OBS_TIMING
    if (X.dbg && tid == 0) { X.dbg[X.dbg_base + 9] += n_occ; X.dbg[X.dbg_base + 10] += n_cf; }
#endif
    for (int e = tid; e < n_occ; e += nt) {
        const uint2 w = X.wl_occ[e];
        const int cell = (int)((w.x & 0xFFFFFFu) >> 2), team = (int)(w.x >> 24);
        occ_event<PB, CAP>(X, pb_cu<PB>(X, team), scr0 + team * team_words, (int)(w.y >> 24), cw_slot(X, cell), w.x & 3u, (int)(w.y & 0xFFFFFFu));
    }
    WAVE_MARK(X, 12, -1);
    if (*X.long_lists == 0) {  // every list is short: one pass, every lane scans the list of its entry
        for (int e = tid; e < n_cf; e += nt) {
            const uint2 w = X.wl_cf[e];
            const int cell = (int)((w.x & 0xFFFFFFu) >> 2), team = (int)(w.x >> 24);
            const int handle = pb_handle<PB>(X, team_meta, team), tot = (int)(w.y & 511u);
            const bool cu = pb_cu<PB>(X, team);
            const int pt = pt_of<PB>(X, cu, handle, tot);
            int lo, hi;
            list_range<PB>(X, cu, cell, pt, lo, hi);
            if (hi > lo && conflict_hit(conflict_flags<PB, ITL>(X, cu, handle, cell, w.x & 3u, pt, lo, hi)))
                atomicMin(&(scr0 + team * team_words)[F_PC * CAP + (int)(w.y >> 24)], tot);
        }
        WAVE_MARK(X, 13, 17);
        __syncthreads();
        return;
    }
    // One entry per CF_CHUNK items of a key's list, so that no lane scans a long list alone: every candidate pushes further
    // entries for the rest of its list, and ALL chunks are scanned after a barrier, one per lane on densely packed wavefronts
    // (scanning the first chunk right away measured 4 % slower on 80 agents, where many lists have several chunks).
    // First entry: tot | chunks << 9 | flags << 15 (OR-ed together below) | node << 24; the others: chunk | index of the
    // first entry << 6 (17 bits) | CF_MORE.
    for (int e0 = 0; e0 < n_cf; e0 += nt) {
        const int e = e0 + tid;
        int nch = 0, lo = 0, hi = 0, cell = 0, handle = 0, tot = 0, pt = 0;
        bool cu = PB == 1;
        uint2 w = make_uint2(0u, 0u);
        if (e < n_cf) {
            w = X.wl_cf[e];
            cell = (int)((w.x & 0xFFFFFFu) >> 2);
            const int team = (int)(w.x >> 24);
            handle = pb_handle<PB>(X, team_meta, team);
            tot = (int)(w.y & 511u);
            cu = pb_cu<PB>(X, team);
            pt = pt_of<PB>(X, cu, handle, tot);
            list_range<PB>(X, cu, cell, pt, lo, hi);
            nch = max(min((hi - lo + CF_CHUNK - 1) / CF_CHUNK, 63), 1);  // an absurdly long list: the last chunk takes the rest
            X.wl_cf[e].y = w.y | ((uint32_t)nch << 9);
        }
        for (int j = 1; __any(j < nch); j++) {
            if (!wl_push(X.wl_cf, X.wl_cf_cap, &X.wl_cnt[1], j < nch, make_uint2(w.x, (uint32_t)j | ((uint32_t)e << 6) | CF_MORE))) {
                // list full: this chunk is scanned here
                const uint32_t f = conflict_flags<PB, ITL>(X, cu, handle, cell, w.x & 3u, pt, lo + j * CF_CHUNK, j == 62 ? hi : min(hi, lo + (j + 1) * CF_CHUNK));
                if (f) atomicOr(&X.wl_cf[e].y, f << 15);
            }
        }
    }
    WAVE_MARK(X, 13, 17);
    __syncthreads();
    WAVE_MARK(X, 14, -1);
    const int n_cf2 = min(X.wl_cnt[1], X.wl_cf_cap);
    bool any_multi = false;
    for (int e = tid; e < n_cf2; e += nt) {
        const uint2 w = X.wl_cf[e];
        const int cell = (int)((w.x & 0xFFFFFFu) >> 2), team = (int)(w.x >> 24);
        const bool more = (w.y & CF_MORE) != 0;
        const int first = more ? (int)((w.y >> 6) & 0x1FFFFu) : e, chunk = more ? (int)(w.y & 63u) : 0;
        const uint32_t fy = more ? wl_flags(X, &X.wl_cf[first]) : w.y;
        const int tot = (int)(fy & 511u), nch = (int)((fy >> 9) & 63u);
        const int handle = pb_handle<PB>(X, team_meta, team);
        const bool cu = pb_cu<PB>(X, team);
        const int pt = pt_of<PB>(X, cu, handle, tot);
        int lo, hi;
        list_range<PB>(X, cu, cell, pt, lo, hi);
        const uint32_t f = conflict_flags<PB, ITL>(X, cu, handle, cell, w.x & 3u, pt, lo + chunk * CF_CHUNK, chunk == 62 ? hi : min(hi, lo + (chunk + 1) * CF_CHUNK));
#ifdef FL_OBS_COUNTS  // with FL_OBS_TIMING: statistics of the conflict entries (they slow the step down)
        if (X.dbg && !more) {
            atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 27], (unsigned long long)(hi - lo));
            if (pt >= (63 << X.tshift)) atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 28], 1ull);
            if (f & 7u) atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 29], 1ull);
            if (conflict_hit(f)) atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 30], 1ull);
            if (!(f & 7u) && (f & 64u)) atomicAdd((unsigned long long *)&X.dbg[X.dbg_base + 31], 1ull);
        }
#endif
        if (nch == 1) {
            if (conflict_hit(f)) atomicMin(&(scr0 + team * team_words)[F_PC * CAP + (int)(fy >> 24)], tot);
        } else {
            if (f) atomicOr(&X.wl_cf[first].y, f << 15);
            any_multi = true;
        }
    }
    if (__any(any_multi) && lane == 0) X.wl_cnt[2] = 1;
    __syncthreads();
    if (X.wl_cnt[2]) {  // keys with more than one chunk: the first entry has collected all flags
        for (int e = tid; e < n_cf2; e += nt) {
            uint2 w = X.wl_cf[e];
            w.y = wl_flags(X, &X.wl_cf[e]);
            if ((w.y & CF_MORE) || ((w.y >> 9) & 63u) == 1u) continue;
            if (conflict_hit((w.y >> 15) & 63u)) atomicMin(&(scr0 + (int)(w.x >> 24) * team_words)[F_PC * CAP + (int)(w.y >> 24)], (int)(w.y & 511u));
        }
        __syncthreads();
    }
}

// the 12 features of node k from its descriptor and accumulators (treeobs.cpp:546-573 / observations.py:433-461)
template <int CAP>
__device__ __forceinline__ void node_row(const ObsCtx &X, int handle, const int *scr, int k, double *f) {
    const int *vs = scr;
    const unsigned long long *ms = reinterpret_cast<const unsigned long long *>(scr + F_MS * CAP);
    const int tot_end = vs[F_TOT * CAP + k] + vs[F_VIS * CAP + k] - 1;
    const uint32_t flags = (uint32_t)vs[F_FLAGS * CAP + k];
    const bool tgt = flags & ND_TARGET;
    double dist_min = 0;
    if (!tgt) {
        const uint16_t dv = X.dm[X.a_tslot[handle] * X.SS + vs[F_END * CAP + k]];
        dist_min = dv == FL_INF16 ? INFINITY : (double)dv;
    }
    const int oa = vs[F_OA * CAP + k], pc = vs[F_PC * CAP + k], ot = vs[F_OT * CAP + k], un = vs[F_UNUS * CAP + k];
    f[0] = tgt ? (double)tot_end : INFINITY;
    f[1] = ot == 0x7fffffff ? INFINITY : (double)ot;
    f[2] = oa == 0x7fffffff ? INFINITY : (double)oa;
    f[3] = pc == 0x7fffffff ? INFINITY : (double)pc;
    f[4] = un < 0 ? INFINITY : (double)un;
    f[5] = (flags & ND_TERMINAL) ? INFINITY : (double)tot_end;
    f[6] = dist_min;
    f[7] = vs[F_SAME * CAP + k]; f[8] = vs[F_OPP * CAP + k]; f[9] = vs[F_MALF * CAP + k];
    f[10] = __longlong_as_double((long long)ms[k]);
    f[11] = vs[F_READY * CAP + k];
}

// children of a node (treeobs.cpp:583-608 / observations.py:464-489): child k (k = 0 left, 1 forward, 2 right, 3 back)
// -> start state or -1 (null cell); tabulated with the segment (fl_dmap.hip k_segments)
__device__ __forceinline__ int child_state(const NodeDesc &nd, int k) {
    if (!(nd.flags & (ND_SWITCH | ND_DEAD_END))) return -1;
    const uint32_t c = ((k < 2 ? nd.kids01 : nd.kids23) >> (16 * (k & 1))) & 0xFFFFu;
    return c == FL_R_NONE ? -1 : (int)c;
}

// scale_node (treeobs.cpp:111-152), float32 arithmetic
__device__ __forceinline__ void scale_and_store(const double *f, float max_dist, int n_agents, float *dst) {
    float v[12];
#pragma unroll
    for (int k = 0; k < 7; k++) v[k] = isinf(f[k]) ? -1.0f : (float)f[k] / max_dist;
    v[7] = f[7] != -1 ? (float)f[7] / (float)n_agents : -1.0f;
    v[8] = f[8] != -1 ? (float)f[8] / (float)n_agents : -1.0f;
    v[9] = f[9] != -1 ? (float)f[9] / (float)n_agents : -1.0f;
    v[10] = f[10] != -1 ? (float)f[10] : -1.0f;
    v[11] = f[11] != -1 ? (float)f[11] / (float)n_agents : -1.0f;
    float4 *d4 = reinterpret_cast<float4 *>(dst);  // rows are 48 B, 16-B aligned
    d4[0] = make_float4(v[0], v[1], v[2], v[3]);
    d4[1] = make_float4(v[4], v[5], v[6], v[7]);
    d4[2] = make_float4(v[8], v[9], v[10], v[11]);
}

__device__ __forceinline__ int kth_set_bit(uint64_t m, int k) {
    for (int i = 0; i < k; i++) m &= m - 1;
    return __ffsll((long long)m) - 1;
}

// LDS arrays of a launch, in carving order (obs_layout on the host decides which exist and where)
enum { L_CELLW = 0, L_NBR, L_SNEXT, L_RKEY, L_SLOT_AGENT, L_SLOT_READY, L_CELL_TARGET, L_A_SPEED, L_A_VPOS, L_A_POS, L_A_TSLOT,
       L_A_TARGET, L_A_MALF, L_A_TPC, L_A_TQ, L_A_TQ2, L_A_RAW, L_RTYPE, L_A_LP, L_A_N, L_A_DIR, L_A_STATE, L_A_FREE, L_A_DEAD, L_MISC, L_TEAM_META, L_WAVE_SCR,
       L_CSR, L_ITEMS, L_WL, L_PARTIAL, L_TMASK, L_TMASK2, L_NH, L_CSR2, L_TMASKB, L_TMASKB2, L_ITEMS2, L_A_LP2, L_A_TPC2, L_SEG, L_DM, L_HOP8, L_COUNT };
#define L_ABSENT 0xFFFFFFFFu
struct ObsLayout {
    unsigned off[L_COUNT];  // byte offset into the dynamic LDS, L_ABSENT = not in this launch
    unsigned total;         // bytes of dynamic LDS
    int nt;                 // threads per workgroup
    int wl_bytes;           // size of the pass B work lists
    int tab_lds;            // the env's dm / seg / nh / hop8 tables are staged in LDS (kernel template TAB_LDS)
    int items_cap, items2_cap;  // entries of the LDS copies of the prediction items (first / second index); an env with more
                                // falls back to the items in HBM scratch / to the two stages
};

// ---------------------------------------------------------------------------------------------- kernel
// outputs of the flatland_cutils builder and of the upstream dense tree builder (k_obs MODE 0 / 1 / 2 = both)
struct ObsArgs {
    int max_nodes, pred_depth, max_depth, tree_pred;  // pred_depth: cutils predictor, tree_pred: upstream predictor
    float *attr, *forest;
    int32_t *adjacency, *node_order, *edge_order;
    uint8_t *valid;
    double *props;
    double *tree_out;
    int n_tree_nodes;
    long long *dbg;  // diagnostic builds only (-DFL_OBS_TIMING): per-env phase clocks
    int tw_c, tw_t, tpw_t;  // node-table words per team of the cutils / upstream builder (0 = builder not in this launch), upstream teams per wavefront
    int use_tmask;     // per-key time-bucket masks in LDS
    int tshift;        // width of their time buckets for horizons beyond 64 steps: 1 << tshift steps (bucket = min(t >> tshift, 63))
    int dual_index;    // fused launch: stage 1 also builds the upstream predictor's index (second set of LDS arrays)
    int bk;            // large maps: the cutils index is grouped by time bucket (OBS_BK_NB); built in the node tables' LDS
    int merged;        // fused launch on small envs: ONE pass B over the trees of both builders (trees_merged)
    ObsLayout L;       // LDS carving of this launch (host-side obs_layout; the kernel only follows it)
};

// upstream dense tree (observations.py:196-254, 464-494): DFS pre-order layout, one TEAM of lanes per agent
// (TEAM = 32: two agents per wavefront, depth <= 2; TEAM = 64: depth 3).  Level L of pass A is handled by 4^L lanes;
// every row that is not a real node is -inf.
//
// pass A of one upstream tree: root row, node topology into the team's table scr (wave-level synchronisation only)
template <int TEAM, int CAP>
__device__ __forceinline__ void upstream_pass_a(const ObsCtx &X, const ObsArgs &P, int b, int i, bool have, int tl, int *scr) {
    const int A = X.A;
    const int D = P.max_depth, NN = P.n_tree_nodes;
    int sz[5];  // sz[l] = nodes of a subtree rooted at depth l
    { int n = 0; for (int l = D; l >= 0; l--) { n = n * 4 + 1; sz[l] = n; } }
    const int ia = have ? i : 0;
    const int g = b * A + ia;
    const int vpos = X.a_vpos[ia];
    const uint32_t dir = X.a_dir[ia];
    const uint32_t rbits = nibble(cw_bits(X, vpos), dir);
    uint32_t orientation = dir;
    if (__popc(rbits) == 1) orientation = first_dir(rbits);
    double *out = P.tree_out + (size_t)g * NN * 12;
    if (have && tl == 0) {
        const uint16_t dv = X.dm[X.a_tslot[i] * X.SS + vpos * 4 + (int)dir];
        double root[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        root[6] = dv == FL_INF16 ? INFINITY : (double)dv;
        root[9] = (double)X.a_malf[i];
        root[10] = X.a_speed[i];
        for (int k = 0; k < 12; k++) out[k] = root[k];
    }
    for (int k = tl; k < CAP; k += TEAM) { scr[F_START * CAP + k] = -1; scr[F_VIS * CAP + k] = 0; }
    team_sync();
    int c_state = -1, c_tot = 1, c_index = -1;
    if (tl < 4) {
        const uint32_t bd = (orientation + (uint32_t)(tl + 3)) & 3u;
        c_index = 1 + tl * sz[1];
        if ((rbits >> (3 - bd)) & 1) c_state = state_towards(X, vpos, bd);
    }
    const uint16_t *dm_t = X.dm + X.a_tslot[ia] * X.SS;  // per-agent constants of the level loop
    const int tgt_r = X.a_target[ia];
    int width = 4;
    for (int level = 1; level <= D; level++) {
        int ch[4] = {-1, -1, -1, -1};
        int ch_tot = 0;
        if (have && tl < width && c_index >= 0 && c_state >= 0) {
            const NodeDesc nd = node_topology(X, dm_t, tgt_r, c_state, c_tot);
            scr[F_START * CAP + c_index] = nd.start; scr[F_TOT * CAP + c_index] = nd.tot0;
            scr[F_VIS * CAP + c_index] = nd.nvis; scr[F_END * CAP + c_index] = nd.end;
            scr[F_FLAGS * CAP + c_index] = (int)nd.flags; scr[F_UNUS * CAP + c_index] = nd.unus;
            ch_tot = nd.tot0 + nd.nvis;
#pragma unroll
            for (int k = 0; k < 4; k++) ch[k] = child_state(nd, k);
        } else if (tl < width) {
            c_index = -1;  // missing node: its whole subtree stays -inf
        }
        if (level == D) break;
        // children of lane p go to lanes 4p .. 4p+3 of the next level (all lanes take part in the shuffles)
        const int src = tl >> 2, which = tl & 3;
        const int p_index = __shfl(c_index, src, TEAM);
        const int s0 = __shfl(ch[0], src, TEAM), s1 = __shfl(ch[1], src, TEAM), s2 = __shfl(ch[2], src, TEAM), s3 = __shfl(ch[3], src, TEAM);
        const int s_tot = __shfl(ch_tot, src, TEAM);
        width *= 4;
        c_index = -1;
        c_state = -1;
        if (tl < width && p_index >= 0) {
            c_state = which == 0 ? s0 : which == 1 ? s1 : which == 2 ? s2 : s3;
            c_tot = s_tot;
            c_index = p_index + 1 + which * sz[level + 1];
        }
    }
    team_sync();
}

// rows 1 .. NN-1 of one upstream tree from its node table (the root row was written by pass A)
template <int TEAM, int CAP>
__device__ __forceinline__ void upstream_rows(const ObsCtx &X, const ObsArgs &P, int b, int i, bool have, int tl, const int *scr) {
    if (!have) return;
    const int NN = P.n_tree_nodes;
    double *out = P.tree_out + (size_t)(b * X.A + i) * NN * 12;
    for (int idx = 1 + tl; idx < NN; idx += TEAM) {
        double *row = out + (size_t)idx * 12;
        if (scr[F_START * CAP + idx] < 0) {
            for (int k = 0; k < 12; k++) row[k] = -INFINITY;
        } else {
            double f[12];
            node_row<CAP>(X, i, scr, idx, f);
            for (int k = 0; k < 12; k++) row[k] = f[k];
        }
    }
}

template <int TEAM, int CAP, bool ITL>
__device__ __forceinline__ void tree_upstream(const ObsCtx &X, const FlDev &d, const ObsArgs &P, int b, int wave, int lane,
                                              int nwaves, int *wave_scr0, int *team_meta) {
    constexpr int TPW = 64 / TEAM;  // teams per wavefront
    const int A = X.A;
    const int team = lane / TEAM, tl = lane % TEAM;
    const int NN = P.n_tree_nodes;
    // team t's node table is slot t; teams that can never hold an agent share the dummy slot behind the real ones
    const int n_slots = min(nwaves * TPW, A);
    int *scr = wave_scr0 + min(wave * TPW + team, n_slots) * (F_WORDS * CAP);
    for (int base = 0; base < A; base += nwaves * TPW) {
        const int i = base + wave * TPW + team;
        const bool have = i < A;
        upstream_pass_a<TEAM, CAP>(X, P, b, i, have, tl, scr);
        TREE_STAMP(X, 6);
        {
            int first;
            const int tot_cells = team_prepare<TEAM, CAP>(have, tl, have ? NN : 1, scr, first);
            const int team_id = wave * TPW + team;
            if (tl == 0) { team_meta[team_id] = have ? tot_cells : 0; team_meta[64 + team_id] = have ? NN : 1; team_meta[128 + team_id] = have ? i : -1; team_meta[256 + team_id] = first; }
        }
        wg_pass_b<false, CAP, ITL>(X, wave * 64 + lane, nwaves * 64, nwaves * TPW, wave_scr0, F_WORDS * CAP, team_meta);
        TREE_STAMP(X, 7);
        upstream_rows<TEAM, CAP>(X, P, b, i, have, tl, scr);
        team_sync();
        TREE_STAMP(X, 8);
    }
}

// Pass A of one flatland_cutils tree (treeobs.cpp:154-256): root row, node topology level by level (BFS), one team of 32
// lanes per agent, two teams per wavefront.  Only wave-level synchronisation, so a wavefront can run it whenever the
// rail bitmap and the agent snapshot are in LDS (the workgroup overlaps it with the path walk of phase 2).
__device__ __forceinline__ void cutils_pass_a(const ObsCtx &X, const FlDev &d, const ObsArgs &P, int b, int i, bool have, int grp,
                                              int gl, int *scr, const uint16_t *a_vpos, const int *a_pos,
                                              const uint8_t *a_dir, const uint8_t *a_state, const double *a_speed,
                                              const uint16_t *a_tslot, float max_dist, uint32_t spk, uint32_t malfw,
                                              int &node_base_out, int &levels_out) {
    constexpr int CAP = 32;
    const int A = X.A, N = P.max_nodes;
    const int ia = have ? i : 0;
    const int g = b * A + ia;
    const int vpos = a_vpos[ia];
    const uint32_t dir = a_dir[ia];
    const uint32_t rbits = nibble(cw_bits(X, vpos), dir);
    uint32_t orientation = dir;
    if (__popc(rbits) == 1) orientation = first_dir(rbits);
    float *F = P.forest + (size_t)g * N * 12;
    scr[F_START * CAP + gl] = -1; scr[F_VIS * CAP + gl] = 0;
    scr[F_PAR * CAP + gl] = -2;
    // level 1: three cells from the root (treeobs.cpp:205-222)
    int c_state = -1, c_parent = 0, c_tot = 1, c_act = 0;
    if (gl < 3) {
        c_act = gl - 1;
        const uint32_t bd = (orientation + (uint32_t)(c_act + 4)) & 3u;
        if ((rbits >> (3 - bd)) & 1) c_state = state_towards(X, vpos, bd);
    }
    const uint16_t *dm_t = X.dm + a_tslot[ia] * X.SS;  // per-agent constants of the level loop
    const int tgt_r = X.a_target[ia];
    int n_cur = 3, node_base = 1, levels = 0;
    if (gl == 0) scr[F_HGT * CAP + 0] = (1 << 2) | 1;  // root: first child = node 1
    while (true) {  // pass A
        levels++;
        const bool active = have && node_base < N && n_cur > 0;
        if (!__any(active)) break;  // wave-uniform: both teams take part in the shuffles below
        const int m = active ? min(n_cur, N - node_base) : 0;
        const bool mine = gl < m;
        const int idx_node = node_base + gl;
        int ch0 = -1, ch1 = -1, ch2 = -1, ch_tot = 0;
        bool explored = false;
        if (mine) {
            if (c_state >= 0) {
                const NodeDesc nd = node_topology(X, dm_t, tgt_r, c_state, c_tot);
                explored = true;
                ch_tot = nd.tot0 + nd.nvis;  // children start one step beyond the end of this walk
                ch0 = child_state(nd, 0); ch1 = child_state(nd, 1); ch2 = child_state(nd, 2);
                scr[F_START * CAP + idx_node] = nd.start; scr[F_TOT * CAP + idx_node] = nd.tot0;
                scr[F_VIS * CAP + idx_node] = nd.nvis; scr[F_END * CAP + idx_node] = nd.end;
                scr[F_FLAGS * CAP + idx_node] = (int)nd.flags; scr[F_UNUS * CAP + idx_node] = nd.unus;
            }
            scr[F_PAR * CAP + idx_node] = c_parent;
        }
        const uint32_t exp_mask = (uint32_t)(__ballot(explored) >> (grp * 32));
        const int n_next = 3 * __popc(exp_mask);
        if (mine) {  // first child's node index (children are numbered consecutively) << 2 | action + 1
            const int fc = explored ? node_base + m + 3 * __popc(exp_mask & ((1u << gl) - 1u)) : 0;
            scr[F_HGT * CAP + idx_node] = (fc << 2) | (c_act + 1);
        }
        // hand the children to the next level's lanes: lane j takes child j % 3 of the (j / 3)-th explored lane
        const int src_rank = gl / 3, which = gl - 3 * src_rank;
        const int src = (gl < n_next) ? kth_set_bit((uint64_t)exp_mask, src_rank) : 0;
        const int s_c0 = __shfl(ch0, src, 32), s_c1 = __shfl(ch1, src, 32), s_c2 = __shfl(ch2, src, 32);
        const int s_tot = __shfl(ch_tot, src, 32);
        if (active) {
            const int parent_base = node_base;
            node_base += m;
            n_cur = n_next;
            if (gl < n_next) {
                c_state = which == 0 ? s_c0 : (which == 1 ? s_c1 : s_c2);
                c_parent = parent_base + src;
                c_tot = s_tot;
                c_act = which - 1;
            }
        }
    }
    // the root row last: its HBM operands (spk, malfunction word) were requested before the level loop
    if (have && gl == 0) {  // root (treeobs.cpp:171-186)
        const uint32_t state = a_state[i];
        double root[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint16_t dv = FL_INF16;
        if (state == ST_DONE) dv = 0;
        else dv = X.dm[a_tslot[i] * X.SS + (is_off_map(state) ? vpos : a_pos[i]) * 4 +  // off the map: vpos = initial position
                       (int)(is_off_map(state) ? SPK_INIT_DIR(spk) : dir)];
        root[6] = dv == FL_INF16 ? INFINITY : (double)dv;
        root[9] = (double)((malfw >> 16) != 0);
        root[10] = (double)(float)a_speed[i];
        scale_and_store(root, max_dist, A, F);
    }
    team_sync();
    node_base_out = node_base;
    levels_out = levels;
}

// rows, adjacency and evaluation orders of one flatland_cutils tree from its node table (after pass B); lane gl of the team
__device__ __forceinline__ void cutils_rows_orders(const ObsCtx &X, const FlDev &d, const ObsArgs &P, int b, int i, bool have, int gl,
                                                   const int *scr, int node_base, int levels, float max_dist) {
    constexpr int CAP = 32;
    const int A = X.A, N = P.max_nodes;
    const int g = b * A + (have ? i : 0);
    float *F = P.forest + (size_t)g * N * 12;
    int32_t *ADJ = P.adjacency + (size_t)g * (N - 1) * 3;
    if (have) {  // rows: lane gl writes node gl + 1
        const int *vs = scr;
        for (int idx = gl + 1; idx < N; idx += 32) {
            int32_t *adj = ADJ + (size_t)(idx - 1) * 3;
            if (idx < node_base) {
                adj[0] = vs[F_PAR * CAP + idx]; adj[1] = idx; adj[2] = (vs[F_HGT * CAP + idx] & 3) - 1;
                if (vs[F_START * CAP + idx] < 0) {
                    const double nn[12] = {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, -1, -1, -1, -1, -1};
                    scale_and_store(nn, max_dist, A, F + (size_t)idx * 12);
                } else {
                    double f[12];
                    node_row<CAP>(X, i, scr, idx, f);
                    if (vs[F_FLAGS * CAP + idx] & ND_ZERO) atomicCAS(&d.err[b], 0, FL_ERR_ZERO_TRANSITION);  // treeobs.cpp:529-535 throws
                    scale_and_store(f, max_dist, A, F + (size_t)idx * 12);
                }
            } else {  // padding rows when the queue ran dry (treeobs.cpp:268-276, 245-249)
                const double nn[12] = {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, -1, -1, -1, -1, -1};
                scale_and_store(nn, max_dist, A, F + (size_t)idx * 12);
                adj[0] = adj[1] = adj[2] = -2;
            }
        }
    }
    // calculate_evaluation_orders (tool.h:468-524): order = height above the leaves.  Lane k holds node k; a node's
    // children are consecutive nodes, so heights settle after as many shuffle rounds as the tree has levels.
    {
        const int packed = gl < node_base ? scr[F_HGT * CAP + gl] : 0;
        const int fc = packed >> 2;          // 0 = no children pushed
        const int parent = gl < node_base ? scr[F_PAR * CAP + gl] : -2;
        const int nchild = fc > 0 ? max(0, min(3, node_base - fc)) : 0;  // children beyond max_nodes were never popped
        // `levels` counted the rounds of pass A including the one that found nothing left: a tree of L levels below the root
        // needs L rounds here (a leaf is 0, every round carries the heights one level up)
        const int max_levels = max(__builtin_amdgcn_readlane(levels, 0), __builtin_amdgcn_readlane(levels, 32));
        int h = 0;
        for (int it = 0; it + 1 < max_levels; it++) {
            const int h0 = __shfl(h, fc, 32), h1 = __shfl(h, fc + 1, 32), h2 = __shfl(h, fc + 2, 32);
            int hn = 0;
            if (nchild > 0) hn = h0 + 1;
            if (nchild > 1) hn = max(hn, h1 + 1);
            if (nchild > 2) hn = max(hn, h2 + 1);
            h = hn;
        }
        const int hp = __shfl(h, parent < 0 ? 0 : parent, 32);
        if (have) {
            int32_t *NO = P.node_order + (size_t)g * N, *EO = P.edge_order + (size_t)g * (N - 1);
            if (gl < N) {
                NO[gl] = gl < node_base ? h : -2;
                if (gl >= 1) EO[gl - 1] = (gl >= node_base || parent < 0) ? -2 : hp;
            }
        }
    }}

// flatland_cutils trees (treeobs.cpp:154-256): two agents per wavefront, a team of 32 lanes each
template <bool ITL>
__device__ __forceinline__ void trees_cutils(const ObsCtx &X, const FlDev &d, const ObsArgs &P, int b, int wave, int lane,
                                             int nwaves, int *wave_scr, int *team_meta,
                                             const uint16_t *a_vpos, const int *a_pos, const uint8_t *a_dir,
                                             const uint8_t *a_state, const double *a_speed, const uint16_t *a_tslot,
                                             float max_dist, bool hoisted) {
    const int A = X.A;
    {
        // two agents per wavefront, a team of 32 lanes each
        constexpr int CAP = 32;
        const int grp = lane >> 5, gl = lane & 31;
        const int N = P.max_nodes;
        // team t's node table is slot t (wg_pass_b); teams that can never hold an agent share the dummy slot behind the real ones
        int *scr = wave_scr + min(wave * 2 + grp, min(nwaves * 2, A)) * (F_WORDS * CAP);
        for (int base = 0; base < A; base += nwaves * 2) {
            const int i = base + wave * 2 + grp;
            const bool have = i < A;
            const int ia = have ? i : 0;
            const int g = b * A + ia;
            const int vpos = a_vpos[ia];
            const uint32_t dir = a_dir[ia];
            const uint32_t rbits = nibble(cw_bits(X, vpos), dir);
            uint32_t orientation = dir;
            if (__popc(rbits) == 1) orientation = first_dir(rbits);
            float *F = P.forest + (size_t)g * N * 12;
            int32_t *ADJ = P.adjacency + (size_t)g * (N - 1) * 3;
            int node_base, levels;
            if (hoisted && base == 0) {  // pass A of the first round already ran beside the path walk
                node_base = team_meta[64 + wave * 2 + grp];
                levels = team_meta[192 + wave * 2 + grp];
            } else {
                cutils_pass_a(X, d, P, b, i, have, grp, gl, scr, a_vpos, a_pos, a_dir, a_state, a_speed, a_tslot, max_dist,
                              d.spk[b * A + (have ? i : 0)], d.malf[b * A + (have ? i : 0)], node_base, levels);
            }
            TREE_STAMP(X, 6);
            {
                int first;
                const int tot_cells = team_prepare<32, CAP>(have, gl, have ? node_base : 1, scr, first);
                const int team_id = wave * 2 + grp;
                if (gl == 0) { team_meta[team_id] = have ? tot_cells : 0; team_meta[64 + team_id] = have ? node_base : 1; team_meta[128 + team_id] = have ? i : -1; team_meta[256 + team_id] = first; }
            }
            wg_pass_b<true, CAP, ITL>(X, wave * 64 + lane, nwaves * 64, nwaves * 2, wave_scr, F_WORDS * CAP, team_meta);
            TREE_STAMP(X, 7);
            cutils_rows_orders(X, d, P, b, i, have, gl, scr, node_base, levels, max_dist);
            team_sync();
            TREE_STAMP(X, 16);
        }
    }
}

// Fused launch on a small env (at most 31 agents, upstream depth <= 2): the trees of BOTH builders go through ONE pass B.
// Node-table slot = pass B team: 0 .. A-1 flatland_cutils trees (team of 32 lanes, wavefront w holds agents 2w, 2w + 1),
// A the shared dummy, A + 1 + u the upstream tree of agent u (team of 16 lanes, four trees a wavefront).
// Pass A of all of them ran beside the path walk (obs_body).
__device__ __forceinline__ int merged_slot_upstream(int A, int u) { return A + 1 + u; }

template <bool ITL>
__device__ __forceinline__ void trees_merged(const ObsCtx &X, const FlDev &d, const ObsArgs &P, int b, int wave, int lane, int nwaves,
                                             int *wave_scr, int *team_meta, float max_dist) {
    constexpr int CAP = 32, TW = F_WORDS * CAP;
    const int A = X.A, NN = P.n_tree_nodes;
    const int grp = lane >> 5, gl = lane & 31, ct = wave * 2 + grp;
    const bool have_c = ct < A;
    int *scr_c = wave_scr + min(ct, A) * TW;
    const int node_base = have_c ? team_meta[64 + ct] : 1, levels = have_c ? team_meta[192 + ct] : 0;
    // (the upstream trees from the last wavefront down, the cutils trees from the first up: their rows are written side by side)
    const int wu = nwaves - 1 - wave;
    const int u = wu * 4 + (lane >> 4), tl = lane & 15;
    const bool have_u = u < A;
    int *scr_u = wave_scr + (have_u ? merged_slot_upstream(A, u) : A) * TW;
    TREE_STAMP(X, 6);
    {
        int first;
        const int cells = team_prepare<32, CAP>(have_c, gl, node_base, scr_c, first);
        if (have_c && gl == 0) { team_meta[ct] = cells; team_meta[64 + ct] = node_base; team_meta[128 + ct] = ct; team_meta[256 + ct] = first; }
        if (wave == 0 && lane == 0) { team_meta[A] = 0; team_meta[64 + A] = 1; team_meta[128 + A] = -1; team_meta[256 + A] = 0xFF; }
    }
    if (wu * 4 < A) {  // wave-uniform
        int first;
        const int cells = team_prepare<16, CAP>(have_u, tl, NN, scr_u, first);
        const int id = merged_slot_upstream(A, u);
        if (have_u && tl == 0) { team_meta[id] = cells; team_meta[64 + id] = NN; team_meta[128 + id] = u; team_meta[256 + id] = first; }
    }
    wg_pass_b<2, CAP, ITL>(X, wave * 64 + lane, nwaves * 64, 2 * A + 1, wave_scr, TW, team_meta);
    TREE_STAMP(X, 7);
    cutils_rows_orders(X, d, P, b, ct, have_c, gl, scr_c, node_base, levels, max_dist);
    if (wu * 4 < A) upstream_rows<16, CAP>(X, P, b, u, have_u, tl, scr_u);
    team_sync();
    TREE_STAMP(X, 16);
}

// One observation build for the workgroup's env.  STAGE 0: stand-alone; the fused launch (both builders) runs STAGE 1
// (cutils; also prepares what the second stage needs) and then STAGE 2 (upstream tree), which reuses the LDS-resident
// rail words / occupancy table / static tables and the predicted paths of stage 1: the upstream predictor's path
// is a prefix of the cutils one (same greedy descent, it only stops at the target and after fewer steps).
// VAR 1 (small maps): the env's distance map, segment, next-hop and eight-hop tables are staged in LDS.  VAR 2 (large maps):
// the pass B work lists live in HBM scratch, which leaves the LDS to the time masks and lifts the cap on their entries.
template <bool CUTILS, int VAR, int STAGE>
__device__ __forceinline__ void obs_body(const FlDev &d, const FlObsScratch &S, const ObsArgs &P) {
    constexpr bool TAB_LDS = (VAR & 1) != 0, WL_HBM = (VAR & 2) != 0;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int A = d.A, R = d.R[b], NS = R * 4, K = d.K[b], U = d.U[b];
    const int Rcap = d.Rcap, Scap = Rcap * 4;
    const int lane = tid & 63, wave = tid >> 6;

    extern __shared__ __align__(16) unsigned char lds[];
    const ObsLayout &L = P.L;
#define LDS_AT(T, which) reinterpret_cast<T *>(lds + L.off[which])
#define LDS_OPT(T, which) (L.off[which] == L_ABSENT ? (T *)nullptr : reinterpret_cast<T *>(lds + L.off[which]))
    uint32_t *cellw = LDS_AT(uint32_t, L_CELLW);  // rail bitmap | occupied-cell table index << 16
    uint16_t *nbr = LDS_AT(uint16_t, L_NBR);
    uint16_t *snext = LDS_OPT(uint16_t, L_SNEXT);
    uint16_t *rkey = LDS_OPT(uint16_t, L_RKEY);
    int *slot_agent = LDS_AT(int, L_SLOT_AGENT);
    int *slot_ready = LDS_AT(int, L_SLOT_READY);
    uint32_t *cell_target = LDS_AT(uint32_t, L_CELL_TARGET);
    double *a_speed = LDS_AT(double, L_A_SPEED);
    uint16_t *a_vpos = LDS_AT(uint16_t, L_A_VPOS);
    int *a_pos = LDS_AT(int, L_A_POS);
    uint16_t *a_tslot = LDS_AT(uint16_t, L_A_TSLOT);
    uint16_t *a_target = LDS_AT(uint16_t, L_A_TARGET);
    uint16_t *a_malf = LDS_AT(uint16_t, L_A_MALF);
    uint16_t *a_tpc = LDS_AT(uint16_t, L_A_TPC);
    double *a_tq = LDS_AT(double, L_A_TQ);
    uint16_t *a_lp = LDS_AT(uint16_t, L_A_LP);
    uint16_t *a_n = LDS_AT(uint16_t, L_A_N);  // waypoints of the agent's predicted path
    uint8_t *a_dir = LDS_AT(uint8_t, L_A_DIR);
    uint8_t *a_state = LDS_AT(uint8_t, L_A_STATE);
    uint8_t *a_free = LDS_AT(uint8_t, L_A_FREE);
    uint8_t *a_dead = LDS_AT(uint8_t, L_A_DEAD);
    int *misc = LDS_AT(int, L_MISC);
    int *team_meta = LDS_AT(int, L_TEAM_META);
    int *wave_scr = LDS_AT(int, L_WAVE_SCR);  // the teams' node tables
    int *csr = LDS_AT(int, L_CSR);
    uint32_t *items_lds = LDS_OPT(uint32_t, L_ITEMS);
    uint32_t *wl_lds = WL_HBM ? nullptr : LDS_AT(uint32_t, L_WL);  // pass B work lists; scratch of the key scan before that
    int *partial = (WL_HBM || L.off[L_PARTIAL] != L_ABSENT) ? LDS_AT(int, L_PARTIAL) : reinterpret_cast<int *>(wl_lds);
    const int wl_entries = WL_HBM ? S.wl_cap : L.wl_bytes / 8;
    unsigned long long *tmask = LDS_OPT(unsigned long long, L_TMASK);
    uint16_t *nh_lds = LDS_OPT(uint16_t, L_NH);
    // second index (fused launch): keys, masks, items and per-agent last waypoint of the upstream predictor
    int *csr2 = LDS_OPT(int, L_CSR2);
    unsigned long long *tmaskb = LDS_OPT(unsigned long long, L_TMASKB);
    unsigned long long *tmask_m2 = LDS_OPT(unsigned long long, L_TMASK2), *tmaskb_m2 = LDS_OPT(unsigned long long, L_TMASKB2);  // own-path filter
    uint32_t *items2 = LDS_OPT(uint32_t, L_ITEMS2);
    uint16_t *a_lp2 = LDS_OPT(uint16_t, L_A_LP2);
    uint16_t *a_tpc2 = LDS_OPT(uint16_t, L_A_TPC2);
    double *a_tq2 = LDS_OPT(double, L_A_TQ2);
    // what phase 1 and the root rows read per agent (pk, spk, malfunction word, latest, earliest, arrival, initial rail cell) and the
    // road types of the rail cells: LDS copies when there is room (small envs), else HBM
    uint32_t *a_raw = LDS_OPT(uint32_t, L_A_RAW);
    uint8_t *rtype_lds = LDS_OPT(uint8_t, L_RTYPE);
    // static tables of the env: LDS copies (TAB_LDS) or HBM
    uint4 *seg_lds = TAB_LDS ? LDS_AT(uint4, L_SEG) : nullptr;
    uint16_t *dm_lds = TAB_LDS ? LDS_AT(uint16_t, L_DM) : nullptr;
    uint16_t *hop8_lds = TAB_LDS ? LDS_AT(uint16_t, L_HOP8) : nullptr;
    const uint4 *gseg = d.seg + (size_t)b * Scap;
    const uint16_t *gdm = d.dm + (size_t)b * d.Ucap * Scap;
    const uint16_t *ghop8 = d.hop8 + (size_t)b * d.Ucap * Scap;
    const uint16_t *gnh = d.nh + (size_t)b * d.Ucap * Rcap;

    const int T = d.T[b], tnow = d.t[b];
#ifdef FL_OBS_TIMING
#define OBS_STAMP(k) do { __syncthreads(); if (tid == 0) { const long long now_ = (long long)wall_clock64(); P.dbg[(size_t)b * 64 + (STAGE == 2 ? 32 : 0) + (k)] = now_; P.dbg[(size_t)b * 64 + (STAGE == 2 ? 32 : 0) + 15] = now_; } } while (0)
#else
#define OBS_STAMP(k) do {} while (0)
#endif
#ifdef FL_OBS_TIMING
    if (tid < 64 && STAGE != 2) P.dbg[(size_t)b * 64 + tid] = 0;
    __syncthreads();
#endif
    if (STAGE == 2 && P.merged && misc[5]) {  // stage 1 built the upstream trees too (trees_merged)
#ifdef FL_OBS_TIMING
        if (tid == 0) { const long long now_ = (long long)wall_clock64(); for (int k = 32; k <= 37; k++) P.dbg[(size_t)b * 64 + k] = now_; }
#endif
        return;
    }
    OBS_STAMP(0);

    const int my_pred_depth = CUTILS ? P.pred_depth : P.tree_pred;
    const bool any_pred = STAGE == 0 ? my_pred_depth >= 0 : true;
    const bool nh_in_lds = nh_lds != nullptr && any_pred;
    // ---- phase 0: stage the rail words and the static tables, clear the per-cell maps, per-agent snapshot into LDS
    if (STAGE != 2) {
        // per-agent snapshot first, on the LAST lanes: its two dependent HBM reads (state, then the rail index of the position)
        // overlap with the staging of the tables by everybody else
        const uint16_t *gridx = d.ridx + (size_t)b * d.H * d.W;
        for (int i = nt - 1 - tid; i < A; i += nt) {
            const int g = b * A + i;
            const uint32_t pk = d.pk[g];
            const uint32_t state = PK_STATE(pk);
            const int pos = d.pos[g];
            const int init_r = d.init_r[g], target_r = d.target_r[g];
            const double speed = d.speed[g];
            const int pos_r = pos < 0 ? -1 : (int)gridx[pos];  // the dynamic state keeps cell ids (C-ABI, step kernel)
            a_pos[i] = pos_r;
            a_vpos[i] = (uint16_t)(is_off_map(state) ? init_r : (is_on_map(state) ? pos_r : target_r));  // loader.cpp:74-82
            a_dir[i] = (uint8_t)PK_DIR(pk);
            a_state[i] = (uint8_t)state;
            a_dead[i] = (uint8_t)PK_DEADLOCK(pk);
            const uint32_t malfw = d.malf[g];
            a_malf[i] = (uint16_t)(malfw & 0xFFFFu);
            if (a_raw) {
                uint32_t *r8 = a_raw + i * 8;
                r8[0] = pk; r8[1] = d.spk[g]; r8[2] = malfw; r8[3] = (uint32_t)d.latest[g]; r8[4] = (uint32_t)d.earliest[g];
                r8[5] = (uint32_t)d.arrival[g]; r8[6] = (uint32_t)init_r;
            }
            a_speed[i] = speed;
            a_tslot[i] = (uint16_t)d.tslot[g];
            a_target[i] = (uint16_t)target_r;
            a_tpc[i] = CUTILS ? (uint16_t)(int)(1.0f / (float)speed) : (uint16_t)(int)(1.0 / speed);
            a_tq[i] = CUTILS ? (double)(float)(1.0 / (double)(float)speed) : 1.0 / speed;
            if (CUTILS && STAGE == 1 && P.dual_index) { a_tpc2[i] = (uint16_t)(int)(1.0 / speed); a_tq2[i] = 1.0 / speed; }  // the upstream predictor's (predictions.py:139)
        }
        {
            const uint16_t *grg = d.rgrid + (size_t)b * Rcap;
            for (int r = tid; r < R; r += nt) cellw[r] = (uint32_t)grg[r] | 0xFFFF0000u;
            // u16 tables: two entries per load (every base is 4-byte aligned: Scap is a multiple of 4, Rcap * U pairs up below)
            const uint32_t *g2 = reinterpret_cast<const uint32_t *>(d.nbr + (size_t)b * Scap);
            uint32_t *l2 = reinterpret_cast<uint32_t *>(nbr);
            for (int c = tid; c < NS / 2; c += nt) l2[c] = g2[c];
            if (snext) {
                g2 = reinterpret_cast<const uint32_t *>(d.snext + (size_t)b * Scap);
                l2 = reinterpret_cast<uint32_t *>(snext);
                for (int c = tid; c < NS / 2; c += nt) l2[c] = g2[c];
            }
            if (rkey) {
                const uint16_t *gk = d.rkey + (size_t)b * Rcap;
                for (int r = tid; r < R; r += nt) rkey[r] = gk[r];
            }
            if (TAB_LDS) {
                const uint2 *gs2 = reinterpret_cast<const uint2 *>(gdm);   // 8-byte pieces: Scap * 2 B is a multiple of 8
                uint2 *ld2 = reinterpret_cast<uint2 *>(dm_lds);
                const int n8 = U * Scap / 4;
                for (int c = tid; c < n8; c += nt) ld2[c] = gs2[c];
                gs2 = reinterpret_cast<const uint2 *>(ghop8);
                ld2 = reinterpret_cast<uint2 *>(hop8_lds);
                for (int c = tid; c < n8; c += nt) ld2[c] = gs2[c];
                for (int c = tid; c < NS; c += nt) seg_lds[c] = gseg[c];
            }
            if (nh_in_lds)
                for (int c = tid; c < U * Rcap; c += nt) nh_lds[c] = gnh[c];
            if (rtype_lds) {
                const uint8_t *grt = d.rtype + (size_t)b * Rcap;
                for (int r = tid; r < R; r += nt) rtype_lds[r] = grt[r];
            }
        }
        for (int i = tid; i < A; i += nt) { slot_agent[i] = -1; slot_ready[i] = 0; }
        for (int c = tid; c < (R + 31) / 32; c += nt) cell_target[c] = 0;
        if (tid < 64) misc[tid] = 0;
        __syncthreads();
        // location_has_agent* (treeobs.cpp:74-81): the last (highest) handle on a cell wins; ready-to-depart counts (:82-91).
        // Occupied cells get an entry in a small table; the per-cell word only holds the entry index.
        for (int i = tid; i < A; i += nt) {
            const uint32_t state = a_state[i];
            const bool on = !is_off_map(state) && a_pos[i] >= 0, off = is_off_map(state);
            if (on || off) {
                const int c = a_vpos[i];  // on the map: the position; off the map: the initial position
                int slot = -1;
                unsigned int cur = *(volatile unsigned int *)&cellw[c];
                while (true) {  // claim (or find) the cell's table entry
                    const unsigned int have = cur >> 16;
                    if (have != 0xFFFFu) { slot = (int)have; break; }
                    if (slot < 0) slot = atomicAdd(&misc[1], 1);
                    const unsigned int old = atomicCAS(&cellw[c], cur, (cur & 0xFFFFu) | ((unsigned int)slot << 16));
                    if (old == cur) break;
                    cur = old;
                }
                if (on) atomicMax(&slot_agent[slot], i);
                else atomicAdd(&slot_ready[slot], 1);
            }
            if (!CUTILS || STAGE == 1) atomicOr(&cell_target[a_target[i] >> 5], 1u << (a_target[i] & 31));
        }
    } else {
        // second stage: only the predictor's times-per-cell differ (int(np.reciprocal(speed)), predictions.py:139)
        for (int i = tid; i < A; i += nt) { a_tpc[i] = (uint16_t)(int)(1.0 / a_speed[i]); a_tq[i] = 1.0 / a_speed[i]; }
    }
    __syncthreads();

    ObsCtx X;
    X.A = A; X.R = R; X.SS = Scap;
    X.cellw = cellw; X.nbr = nbr; X.snext = snext; X.rkey = rkey;
    X.slot_agent = slot_agent; X.slot_ready = slot_ready; X.cell_target = cell_target;
    X.seg = TAB_LDS ? seg_lds : gseg;
    X.dm = TAB_LDS ? dm_lds : gdm;
    X.dbg = P.dbg ? P.dbg + (size_t)b * 64 : nullptr;
    X.dbg_base = STAGE == 2 ? 32 : 0;
    X.a_vpos = a_vpos; X.a_dir = a_dir; X.a_state = a_state; X.a_malf = a_malf; X.a_speed = a_speed;
    X.a_tpc = a_tpc; X.a_tq = a_tq; X.a_tslot = a_tslot; X.a_target = a_target;
    uint32_t *csr_items = S.cell_items + (size_t)b * S.items_cap;
    X.csr_end = csr; X.items_lds = nullptr; X.items_glb = csr_items; X.bk_rel = nullptr;
    X.Tn = my_pred_depth >= 0 ? my_pred_depth + 1 : 0;
    // without the masks nearly every cell on somebody's route would be a conflict candidate: those are handled in place
    X.tmask = (P.use_tmask && X.Tn > 0) ? tmask : nullptr;
    X.wl_hbm = WL_HBM;
    X.wl_occ = WL_HBM ? S.wl + (size_t)b * S.wl_cap : reinterpret_cast<uint2 *>(wl_lds);
    X.wl_occ_cap = X.tmask ? wl_entries / OBS_WL_OCC_DIV : wl_entries;  // a share of the entries
    X.wl_cf = X.wl_occ + X.wl_occ_cap; X.wl_cf_cap = wl_entries - X.wl_occ_cap;
    X.wl_cnt = misc + 8;
    X.long_lists = misc + 11;
    X.tshift = X.Tn <= 64 ? 0 : P.tshift;  // bucket = min(t >> tshift, 63)
    // one pass B over the trees of both builders (stage 1 of the fused launch): the upstream builder's side of the context
    const bool merged = CUTILS && STAGE == 1 && P.merged != 0;
    X.n_cu = A + 1;
    X.u_csr_end = csr2; X.u_items = items2; X.u_tmask = tmaskb; X.a_tq2 = a_tq2;
    X.tmask_m2 = merged ? tmask_m2 : nullptr; X.u_tmask_m2 = tmaskb_m2;
    X.path = S.path + (size_t)b * A * S.pred_cap; X.pred_cap = S.pred_cap;
    X.a_lp = a_lp; X.a_lp2 = a_lp2; X.a_tpc2 = a_tpc2;
    X.u_Tn = P.tree_pred + 1; X.u_tshift = X.u_Tn <= 64 ? 0 : P.tshift;

    OBS_STAMP(1);
    // ---- phase 1 (cutils only): deadlock flags, valid actions, attribute rows.  The deadlock check is the work of ONE
    // wavefront (wave-level synchronisation only) and runs beside the path walkers of phase 2; the per-agent rest is spread
    // over all wavefronts (phase1b).
    auto phase1a = [&]() __attribute__((always_inline)) {
        // DeadlockChecker (deadlock_checker.cpp:11-110) as a least fixpoint: an active agent is "free" when one of
        // its exits leads to an empty cell or to a free, not yet deadlocked agent (or it has no exit at all);
        // every other active agent becomes (and stays) deadlocked.  Equivalent to the reference's DFS + _fix_deps.
        if (A <= 64) {
            // one agent a lane: who blocks whom is looked up once (a bit mask of the agents on the exits), the fixpoint
            // itself runs on ballots, without a memory access
            const int i = lane;
            bool active = false, fr = false;
            unsigned long long blockers = 0;
            if (i < A && is_on_map(a_state[i]) && !a_dead[i]) {
                active = true;
                const uint32_t bits = nibble(cw_bits(X, a_pos[i]), a_dir[i]);
                if (bits == 0) fr = true;
                for (uint32_t m = 0; m < 4; m++) {
                    if (!((bits >> (3 - m)) & 1)) continue;
                    const uint32_t nr = nbr[a_pos[i] * 4 + (int)m];
                    if (nr == FL_R_NONE) { fr = true; continue; }  // leaves the grid / the rail: nobody can be there
                    const uint32_t sl = cw_slot(X, (int)nr);
                    const int opp = sl != 0xFFFFu ? slot_agent[sl] : -1;
                    if (opp < 0) fr = true;
                    else if (!a_dead[opp]) blockers |= 1ull << opp;
                }
            }
            unsigned long long free_set = __ballot(fr);
            while (true) {  // monotone: any evaluation order reaches the same least fixpoint
                if (active && !fr && (free_set & blockers)) fr = true;
                const unsigned long long next = __ballot(fr);
                if (next == free_set) break;
                free_set = next;
            }
            if (i < A) a_free[i] = fr;
            team_sync();
        } else {
            for (int i = lane; i < A; i += 64) {
                bool fr = false;
                if (is_on_map(a_state[i]) && !a_dead[i]) {
                    const uint32_t bits = nibble(cw_bits(X, a_pos[i]), a_dir[i]);
                    if (bits == 0) fr = true;
                    for (uint32_t m = 0; m < 4 && !fr; m++) {
                        if (!((bits >> (3 - m)) & 1)) continue;
                        const uint32_t nr = nbr[a_pos[i] * 4 + (int)m];
                        if (nr == FL_R_NONE) { fr = true; continue; }  // leaves the grid / the rail: nobody can be there
                        const uint32_t sl = cw_slot(X, (int)nr);
                        if (sl == 0xFFFFu || slot_agent[sl] < 0) fr = true;
                    }
                }
                a_free[i] = fr;
            }
            team_sync();
            while (true) {  // monotone: any evaluation order reaches the same least fixpoint
                bool changed = false;
                for (int i = lane; i < A; i += 64) {
                    if (is_on_map(a_state[i]) && !a_dead[i] && !a_free[i]) {
                        const uint32_t bits = nibble(cw_bits(X, a_pos[i]), a_dir[i]);
                        bool fr = false;
                        for (uint32_t m = 0; m < 4 && !fr; m++) {
                            if (!((bits >> (3 - m)) & 1)) continue;
                            const uint32_t nr = nbr[a_pos[i] * 4 + (int)m];
                            const uint32_t sl = nr != FL_R_NONE ? cw_slot(X, (int)nr) : 0xFFFFu;
                            const int opp = sl != 0xFFFFu ? slot_agent[sl] : -1;
                            if (opp >= 0 && !a_dead[opp] && a_free[opp]) fr = true;
                        }
                        if (fr) { a_free[i] = 1; changed = true; }
                    }
                }
                team_sync();
                if (!__any(changed)) break;
            }
        }
        for (int i = lane; i < A; i += 64) {  // commit the new deadlocks; the deadlock flag of the attribute row and of props
            const int g = b * A + i;
            if (is_on_map(a_state[i]) && !a_dead[i] && !a_free[i]) {
                a_dead[i] = 1;
                d.pk[g] |= (1u << 18);
            }
            P.attr[(size_t)g * FL_CUTILS_ATTR + 41] = (float)a_dead[i];
            if (P.props) P.props[(size_t)g * 3 + 1] = (double)a_dead[i];
        }
    };
    // Rest of phase 1, per agent: valid actions, props, attribute row (everything but the deadlock flag).  A team of 32
    // lanes per agent: every lane derives the agent's scalars (broadcast loads) and writes elements gl, gl + 32, gl + 64 of
    // the row, so it can run on any wavefront beside the deadlock check.
    // what phase1b reads from HBM: requested ahead of the hoisted pass A so that the two latencies overlap
    struct AgentRaw { uint32_t pk, spk, malfw; int latest, earliest, arrival, init_r, road_type; };
    auto phase1b_load = [&](int i) __attribute__((always_inline)) {
        const int g = b * A + i, pos = a_pos[i];
        AgentRaw r;
        if (a_raw) {
            const uint32_t *r8 = a_raw + i * 8;
            r.pk = r8[0]; r.spk = r8[1]; r.malfw = r8[2]; r.latest = (int)r8[3]; r.earliest = (int)r8[4]; r.arrival = (int)r8[5]; r.init_r = (int)r8[6];
        } else {
            r.pk = d.pk[g]; r.spk = d.spk[g]; r.malfw = d.malf[g];
            r.latest = d.latest[g]; r.earliest = d.earliest[g]; r.arrival = d.arrival[g];
            r.init_r = d.init_r[g];
        }
        // static per rail cell (fl_host.hip)
        r.road_type = pos < 0 ? 0 : rtype_lds ? (int)rtype_lds[pos] : (int)d.rtype[(size_t)b * Rcap + pos];
        return r;
    };
    auto phase1b = [&](int i, int gl, const AgentRaw &raw) __attribute__((always_inline)) {
        const int g = b * A + i;
        const uint32_t state = a_state[i];
        const uint32_t pk = raw.pk, spk = raw.spk;
        const int pos = a_pos[i];
        const uint32_t dir = a_dir[i];
        const uint32_t scount = PK_SCOUNT(pk), max_count = SPK_MAX_COUNT(spk), init_dir = SPK_INIT_DIR(spk);
        const uint32_t old_dir = PK_OLD_DIR(pk) == 4 ? dir : PK_OLD_DIR(pk);
        // update_dist_target (loader.cpp:163-179)
        const int dmb = a_tslot[i] * X.SS;
        const uint16_t dv_init = X.dm[dmb + raw.init_r * 4 + (int)init_dir];
        const float init_dist = dv_init == FL_INF16 ? INFINITY : (float)dv_init;
        float dist_target;
        if (state == ST_DONE) dist_target = 0;
        else if (is_off_map(state)) dist_target = init_dist;
        else {
            const uint16_t dv = X.dm[dmb + pos * 4 + (int)dir];
            dist_target = dv == FL_INF16 ? INFINITY : (float)dv;
        }
        // valid-action mask (loader.cpp:273-312)
        uint32_t va = 0;
        const uint32_t cell = pos >= 0 ? cw_bits(X, pos) : 0;
        if (state == ST_MOVING || state == ST_STOPPED) {
            if (scount == 0) {
                const uint32_t bits = nibble(cell, dir);
                int cnt = 0;
                bool has_branch = false;
                for (uint32_t a = ACT_LEFT; a <= ACT_RIGHT; a++) {
                    const uint32_t nd = (dir + a + 2u) & 3u;
                    if ((bits >> (3 - nd)) & 1) {
                        va |= 1u << a;
                        cnt++;
                        const uint32_t nr = nbr[pos * 4 + (int)nd];
                        if (nr != FL_R_NONE && __popc(cw_bits(X, (int)nr)) > 2) has_branch = true;
                    }
                }
                if (__popc(cell) > 2 || (cnt == 1 && has_branch)) va |= 1u << ACT_STOP;
            } else va |= 1u << ACT_NOTHING;
        } else if (state == ST_READY) va = (1u << ACT_FORWARD) | (1u << ACT_STOP);
        else va = 1u << ACT_NOTHING;
        if (gl < 5) P.valid[(size_t)g * 5 + gl] = (va >> gl) & 1;
        if (P.props && gl == 5) {
            P.props[(size_t)g * 3 + 0] = (double)dist_target;
            P.props[(size_t)g * 3 + 2] = (double)(state == ST_READY);
        }
        // AgentAttrParser::get_features (feature_parser.cpp:3-98): elements 0 .. 69 are 0 / 1 -- bit j of (m_lo, m_hi)
        const int road_type = raw.road_type;
        const uint32_t malfw = raw.malfw;
        const uint32_t malf01 = (malfw & 0xFFFFu) != 0, nmalf01 = (malfw >> 16) != 0;
        const uint32_t rev = __brev(cell) >> 16;  // element 49 + k = bit 15 - k of the rail word
        unsigned long long m_lo = (state < 7u ? 1ull << state : 0ull) | (road_type < 11 ? 1ull << (7 + road_type) : 0ull) | (1ull << (18 + nmalf01)) |
                                  (1ull << (28 + init_dir)) | (1ull << (32 + dir)) | (1ull << (36 + old_dir)) |
                                  ((unsigned long long)(state == ST_MOVING) << 40) | ((unsigned long long)PK_SIGMALF(pk) << 42) |
                                  ((unsigned long long)(!malf01) << 43) | ((unsigned long long)(scount == 0) << 44) |
                                  ((unsigned long long)(scount == max_count) << 45) | ((unsigned long long)(state == ST_MALF || state == ST_MALF_OFF) << 46) |
                                  ((unsigned long long)is_off_map(state) << 47) | ((unsigned long long)is_on_map(state) << 48) |
                                  ((unsigned long long)(rev & 0x7FFFu) << 49);
        const uint32_t m_hi = (rev >> 15) | (va << 1);
        // elements 70 .. 82
        const float max_t = (float)T, max_dist_target = (float)((d.H + d.W) * 8);
        const float f_step = (float)tnow / max_t;
        const float f_latest = (float)raw.latest / max_t;
        const float f_before = f_latest - f_step;
        const float f_dist = isinf(dist_target) ? 8.0f : dist_target / max_dist_target;
        const float fv[13] = {(float)i / (float)A, f_step, (float)raw.earliest / max_t, f_latest, (float)raw.arrival / max_t, f_before, f_dist,
                              f_before < f_dist ? f_before : f_dist, (float)max_count / 10, (float)a_speed[i] / 1.0f, (float)scount / 10,
                              (float)malf01 / 10, isinf(init_dist) ? 8.0f : init_dist / max_dist_target};
        float *o = P.attr + (size_t)g * FL_CUTILS_ATTR;
        o[gl] = (float)((m_lo >> gl) & 1ull);
        if (gl + 32 != 41) o[gl + 32] = (float)((m_lo >> (gl + 32)) & 1ull);  // element 41: the deadlock flag (phase 1a)
        if (gl + 64 < FL_CUTILS_ATTR) {
            float v = (float)((m_hi >> gl) & 1u);
#pragma unroll
            for (int k = 0; k < 13; k++) v = gl == 6 + k ? fv[k] : v;
            o[gl + 64] = v;
        }
    };
    // all agents of the env, one team of 32 lanes each
    auto phase1b_all = [&]() __attribute__((always_inline)) {
        for (int i = wave * 2 + (lane >> 5); i < A; i += 2 * (nt >> 6)) phase1b(i, lane & 31, phase1b_load(i));
    };

    // eight walker lanes per agent, on at least four wavefronts (consecutive wavefronts of a workgroup land on different
    // SIMDs): a lone wavefront issues at the full rate of its SIMD
    const bool do_p1 = CUTILS && STAGE != 2;
    // (with hundreds of agents every wavefront would walk: one of them is kept back for phase 1, which then runs beside the walk)
    const int nw_walk = (X.Tn > 0 && STAGE != 2) ? min((nt >> 6) - ((do_p1 && (nt >> 6) > 4) ? 1 : 0), max(2, (A + 7) / 8)) : 0;
    const bool p1_beside_walk = nw_walk < (nt >> 6);  // a wavefront is left over
    // job queue of phase 2: pass A of four upstream trees (one pass B for both builders, see trees_merged; longest first), the
    // rest of phase 1 of two agents.  (Ending the phase with the last pass A of a cutils tree and taking the rest of the queue
    // beside the fill of the index measured 0.8 us slower: a job is a chain of HBM reads and takes as long as the fill.)
    const int n_up_jobs = merged ? (A + 3) / 4 : 0, n_p1_jobs = (do_p1 && p1_beside_walk && X.Tn > 0) ? (A + 1) / 2 : 0;
    auto drain_jobs = [&]() __attribute__((always_inline)) {
        while (n_up_jobs + n_p1_jobs > 0) {
            int j = 0;
            if (lane == 0) j = atomicAdd(&misc[6], 1);
            j = __builtin_amdgcn_readfirstlane(j);
            if (j >= n_up_jobs + n_p1_jobs) break;
            if (j < n_up_jobs) {
                const int u = 4 * j + (lane >> 4);
                upstream_pass_a<16, 32>(X, P, b, u, u < A, lane & 15, wave_scr + (u < A ? merged_slot_upstream(A, u) : A) * (F_WORDS * 32));
            } else {
                const int i = 2 * (j - n_up_jobs) + (lane >> 5);
                if (i < A) phase1b(i, lane & 31, phase1b_load(i));
            }
        }
    };
    if (do_p1 && (!p1_beside_walk || X.Tn == 0)) {
        if (wave == 0) phase1a();
        phase1b_all();
        __syncthreads();
    }
    OBS_STAMP(2);
    // ---- phase 2: predicted paths + per-key CSR index of (agent, waypoint, time interval)
    if (X.Tn > 0) {
        // fused launch: stage 1 builds the upstream predictor's index too (same paths, one pass over the waypoints); stage 2
        // then starts at its trees.  misc[4] tells stage 2 that the second index is complete.
        const bool dual = CUTILS && STAGE == 1 && P.dual_index != 0 && P.tree_pred >= 0;
        const bool reuse = STAGE == 2 && P.dual_index != 0 && misc[4] != 0;
        const int Tn2 = P.tree_pred + 1, tshift2 = Tn2 <= 64 ? 0 : P.tshift;  // the same bucket width stage 2 queries with
        // large maps: bucketed lists (OBS_BK_NB).  Their per-(key, bucket) counters -- u16, two per word -- live in the node
        // tables' LDS while the index is built (so no tree work is hoisted beside the walk), the offsets go to HBM afterwards
        const bool bk = CUTILS && STAGE != 2 && P.bk != 0 && X.Tn > 64 && X.tmask != nullptr;
        uint32_t *bkc = reinterpret_cast<uint32_t *>(wave_scr);
        if (!reuse) {
            for (int k = tid; k <= K; k += nt) csr[k] = 0;
            if (tid == 0) misc[11] = 0;
            if (X.tmask) for (int k = tid; k <= K; k += nt) { tmask[k] = 0ull; if (X.tmask_m2) tmask_m2[k] = 0ull; }
            if (bk) for (int k = tid; k < K * OBS_BK_NB / 2; k += nt) bkc[k] = 0u;
        }
        if (dual) for (int k = tid; k <= K; k += nt) { csr2[k] = 0; if (P.use_tmask) tmaskb[k] = 0ull; if (X.tmask_m2) tmaskb_m2[k] = 0ull; }
        __syncthreads();
        const int pred_depth = my_pred_depth;
        if (STAGE != 2) {
        // Roles of the wavefronts while the paths are walked (a chain of dependent loads on one lane per agent): the LAST
        // nw_walk wavefronts walk, the one before them does phase 1, and every wavefront (the walkers afterwards) runs pass A
        // of the first round of cutils trees for its two agents -- none of that needs the prediction index.
        const int w_first = (nt >> 6) - nw_walk, wsel = wave - w_first;
        if (do_p1 && p1_beside_walk && wave == w_first - 1) {
            phase1a();
#ifdef FL_OBS_TIMING
            if (lane == 0) atomicMax((unsigned long long *)&X.dbg[20], (unsigned long long)wall_clock64());
#endif
        }
        if (wsel >= 0) __builtin_amdgcn_s_setprio(3);  // the walk is the critical path: its wavefronts issue first
        // Greedy strict descent on the distance map (predictions.cpp:107-133 / rail_env_shortest_paths.py:245-265): the choice
        // at every (target, cell, orientation) is static (k_nexthop), so the predicted path is the chain of next-hops from
        // the agent's state until nothing is strictly closer (on the target, or at once when it is unreachable), cut after
        // n_max waypoints: cutils walks max_depth iterations and appends the final waypoint (predictions.cpp:131-133),
        // upstream stops after max_depth waypoints (rail_env_shortest_paths.py:245-267) and keeps the current position when
        // there is no path (predictions.py:126,150-156).  EIGHT lanes walk one path: lane j takes j single hops and then
        // eight hops at a time through the static hop8 table, recording the waypoints j, j + 8, j + 16, ...
        const int n_max = CUTILS ? pred_depth + 1 : max(pred_depth, 1);
        for (int base = 0; wsel >= 0 && base < A; base += 8 * nw_walk) {
            const int slot = lane >> 3, j = lane & 7;
            const int i = base + wsel * 8 + slot;
            const bool have = i < A;
            const int ia = have ? i : 0;
            uint16_t *path = S.path + ((size_t)b * A + ia) * S.pred_cap;
            const int u = a_tslot[ia];
            uint32_t st = ((uint32_t)a_vpos[ia] << 2) | a_dir[ia];
            bool alive = have && j < n_max;
            auto lead_in = [&](const uint16_t *nh_u) __attribute__((always_inline)) {
                for (int h = 0; h < 7; h++) {
                    if (alive && h < j) {
                        const uint32_t hop = ((uint32_t)nh_u[st >> 2] >> (3u * (st & 3u))) & 7u;
                        const uint32_t nr = hop == 4u ? (uint32_t)FL_R_NONE : (uint32_t)nbr[(st & ~3u) | hop];
                        if (nr == FL_R_NONE) alive = false;
                        else st = (nr << 2) | hop;
                    }
                }
            };
            // waypoints that can be occupied within the horizon enter the per-key index: they are counted as they are recorded
            // (bucketed lists count per bucket, below)
            const int hz1 = bk ? -1 : max(0, CUTILS ? (X.Tn - 2) / (int)a_tpc[ia] + 1 : (X.Tn - 1) / (int)a_tpc[ia]);
            const int hz2 = dual ? max(0, min(P.tree_pred - 1, (Tn2 - 1) / (int)a_tpc2[ia])) : -1;
            auto walk8 = [&](const uint16_t *h8) __attribute__((always_inline)) {
                int idx = j, last = -1;
                while (__any(alive)) {
                    if (alive) {
                        path[idx] = (uint16_t)st;
                        last = idx;
                        if (idx <= hz1) {
                            const int key = key_of(X, (int)(st >> 2));
                            atomicAdd(&csr[key], 1);
                            if (idx <= hz2) atomicAdd(&csr2[key], 1);
                        }
                        const uint32_t s8 = idx + 8 < n_max ? (uint32_t)h8[st] : (uint32_t)FL_R_NONE;
                        if (s8 == FL_R_NONE) alive = false;
                        else { st = s8; idx += 8; }
                    }
                }
                return last;
            };
            // separate call sites so that each keeps a static address space (LDS copy vs HBM table)
            if (nh_in_lds) lead_in(nh_lds + u * Rcap);
            else lead_in(gnh + (size_t)u * Rcap);
            int m = TAB_LDS ? walk8(hop8_lds + u * Scap) : walk8(ghop8 + (size_t)u * Scap);
            m = max(m, __shfl_xor(m, 1)); m = max(m, __shfl_xor(m, 2)); m = max(m, __shfl_xor(m, 4));
            if (have && j == 0) {
                const int n = m + 1;  // lane 0 always records the current position
                // last waypoint that can be occupied within the horizon; only those enter the per-key index
                const int tpc = a_tpc[i];
                const int horizon = CUTILS ? (X.Tn - 2) / tpc + 1 : (X.Tn - 1) / tpc;
                a_lp[i] = (uint16_t)max(0, min(n - 1, horizon));
                a_n[i] = (uint16_t)n;
                if (dual) {  // the upstream path is a prefix of this one (see stage 2 below)
                    const int tpc2 = a_tpc2[i];
                    const int n_py = (n - 1 < P.tree_pred) ? n : P.tree_pred;
                    a_lp2[i] = (uint16_t)max(0, min(n_py - 1, (Tn2 - 1) / tpc2));
                }
            }
        }
        if (wsel >= 0) __builtin_amdgcn_s_setprio(0);
#ifdef FL_OBS_TIMING
        if (wsel >= 0 && lane == 0) atomicMax((unsigned long long *)&X.dbg[21], (unsigned long long)wall_clock64());
#endif
        {
            const int grp = lane >> 5, gl = lane & 31, team_id = wave * 2 + grp;
#ifdef FL_OBS_TIMING
            const long long t_pa0 = (long long)wall_clock64();
#endif
            if (CUTILS && !bk) {  // pass A of the team's first tree
                int node_base, levels;
                const bool have = team_id < A;
                cutils_pass_a(X, d, P, b, team_id, have, grp, gl, wave_scr + min(team_id, min((nt >> 6) * 2, A)) * (F_WORDS * 32),
                              a_vpos, a_pos, a_dir, a_state, a_speed, a_tslot, (float)T,
                              a_raw ? a_raw[(have ? team_id : 0) * 8 + 1] : d.spk[b * A + (have ? team_id : 0)],
                              a_raw ? a_raw[(have ? team_id : 0) * 8 + 2] : d.malf[b * A + (have ? team_id : 0)], node_base, levels);
                if (gl == 0) { team_meta[64 + team_id] = node_base; team_meta[192 + team_id] = levels; }
            }
#ifdef FL_OBS_TIMING
            if (lane == 0) {
                atomicMax((unsigned long long *)&X.dbg[19], (unsigned long long)wall_clock64());
                atomicMax((unsigned long long *)&X.dbg[23], (unsigned long long)((long long)wall_clock64() - t_pa0));
            }
#endif
            // The wavefronts are done with their roles at very different times (no tree to build, a short walk, a deep
            // tree): what is left of this phase is a queue of jobs that whoever is free takes -- the rest of phase 1, two agents a job
            // (longest first: pass A of four upstream trees when both builders share one pass B, see trees_merged)
            drain_jobs();
        }
#ifdef FL_OBS_TIMING
        if (lane == 0) atomicMax((unsigned long long *)&X.dbg[22], (unsigned long long)wall_clock64());
#endif
        if (bk) {  // bucketed lists: one copy of the item per time bucket its interval touches (cutils: w(t) = 0 for t = 0, min((t-1)/tpc + 1, lp))
            __syncthreads();
            for (int i = wave; i < A; i += (nt >> 6)) {
                const uint16_t *path = S.path + ((size_t)b * A + i) * S.pred_cap;
                const int lp = a_lp[i];
                const int tpc = a_tpc[i], tlast = X.Tn - 1;
                for (int k = lane; k <= lp; k += 64) {
                    const int key = key_of(X, (int)(path[k] >> 2));
                    const int tlo = k == 0 ? 0 : (k - 1) * tpc + 1, span = k == 0 ? 1 : tpc;
                    const int thi = (k == lp || tlo + span - 1 >= tlast) ? tlast : tlo + span - 1;
                    const int b1 = min(tlo >> OBS_BK_SHIFT, OBS_BK_NB - 1), b2 = min(thi >> OBS_BK_SHIFT, OBS_BK_NB - 1);
                    for (int bb = b1; bb <= b2; bb++) {
                        const int kb = key * OBS_BK_NB + bb;
                        atomicAdd(&bkc[kb >> 1], (kb & 1) ? 0x10000u : 1u);
                    }
                    atomicAdd(&csr[key], b2 - b1 + 1);
                }
            }
        }
        } else if (!reuse) {
            // second stage: the upstream path is the prefix of the cutils path kept by stage 1 -- it stops at the target
            // (which ends the cutils path too) and after pred_depth waypoints (rail_env_shortest_paths.py:245-267)
            for (int i = tid; i < A; i += nt) {
                const int n_c = a_n[i];
                const int n_py = (n_c - 1 < pred_depth) ? n_c : pred_depth;
                const int horizon = (X.Tn - 1) / a_tpc[i];
                a_lp[i] = (uint16_t)max(0, min(n_py - 1, horizon));
            }
            __syncthreads();
            for (int i = wave; i < A; i += (nt >> 6)) {
                const uint16_t *path = S.path + ((size_t)b * A + i) * S.pred_cap;
                const int lp = a_lp[i];
                for (int k = lane; k <= lp; k += 64) atomicAdd(&csr[key_of(X, (int)(path[k] >> 2))], 1);
            }
        }
        __syncthreads();
        OBS_STAMP(3);
        // exclusive scan over the keys: per-thread chunk sums, wave-0 scan of the partial sums, rescan.  With the second
        // index both counts share the scan, 16 bits each (the launcher guarantees totals below 65536).
        if (!reuse) {
            const int chunk = (K + 1 + nt - 1) / nt;
            const int lo = min(tid * chunk, K + 1), hi = min(lo + chunk, K + 1);
            int sum = 0;
            int longest = 0;
            for (int k = lo; k < hi; k++) {
                sum += dual ? (csr[k] | (csr2[k] << 16)) : csr[k];
                longest = max(longest, dual ? max(csr[k], csr2[k]) : csr[k]);
            }
            if (longest > CF_DIRECT) misc[11] = 1;  // (lists of the bucketed index count an item once per bucket: they only look longer)
            partial[tid] = sum;
            __syncthreads();
            if (wave == 0) {
                // each lane of wave 0 owns nt / 64 consecutive partials
                constexpr int PER = OBS_NT / 64;
                const int per = nt >> 6;
                int loc[PER], tot = 0;
#pragma unroll
                for (int q = 0; q < PER; q++) { loc[q] = q < per ? partial[lane * per + q] : 0; tot += loc[q]; }
                int incl = tot;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
                int run = incl - tot;
#pragma unroll
                for (int q = 0; q < PER; q++) { if (q < per) partial[lane * per + q] = run; run += loc[q]; }
            }
            __syncthreads();
            int run = partial[tid];
            if (dual) {
                for (int k = lo; k < hi; k++) {
                    const int v = csr[k] | (csr2[k] << 16);
                    csr[k] = run & 0xFFFF; csr2[k] = (int)((unsigned)run >> 16);
                    run += v;
                }
                if (hi == K + 1 && lo < hi) { misc[2] = run & 0xFFFF; misc[3] = (int)((unsigned)run >> 16); }
            } else {
                for (int k = lo; k < hi; k++) { const int v = csr[k]; csr[k] = run; run += v; }  // csr[k] = start of key k
                if (hi == K + 1 && lo < hi) misc[2] = run;                                      // total number of items
            }
        }
        __syncthreads();
        if (reuse) {  // stage 1 built this index
            csr = csr2; X.csr_end = csr2; X.items_lds = items2;
            X.tmask = P.use_tmask ? tmaskb : nullptr;
            if (!X.tmask) { X.wl_occ_cap = wl_entries; X.wl_cf = X.wl_occ + X.wl_occ_cap; X.wl_cf_cap = 0; }
        }
        if (bk) {  // counts of a key's buckets -> their start offsets inside the key's list (bumped to the ends by the fill)
            for (int key = tid; key < K; key += nt) {
                uint32_t *w4 = bkc + key * (OBS_BK_NB / 2);
                uint32_t run = 0;
#pragma unroll
                for (int q = 0; q < OBS_BK_NB / 2; q++) {
                    const uint32_t v = w4[q], c0 = v & 0xFFFFu, c1 = v >> 16;
                    w4[q] = run | ((run + c0) << 16);
                    run += c0 + c1;
                }
            }
            __syncthreads();
        }
        const bool fit = items_lds != nullptr && misc[2] <= L.items_cap;
        const bool dual_fill = dual && misc[3] <= L.items2_cap;
        if (dual && tid == 0) misc[4] = dual_fill ? 1 : 0;
        if (merged && tid == 0) misc[5] = (fit && dual_fill) ? 1 : 0;  // else: the two stages as usual
        if (fit && !reuse) { csr_items = items_lds; X.items_lds = items_lds; }
        // fill: bumping csr[key] turns it from the start into the END offset of key's list (start = csr[key - 1]);
        // one wavefront per agent, one lane per waypoint
        for (int i = wave; !reuse && i < A; i += (nt >> 6)) {
            const uint16_t *path = S.path + ((size_t)b * A + i) * S.pred_cap;
            const int lp = a_lp[i], tpc = a_tpc[i], tlast = X.Tn - 1;
            const int lp2 = dual_fill ? (int)a_lp2[i] : -1, tpc2 = dual_fill ? (int)a_tpc2[i] : 1;
            for (int k = lane; k <= lp; k += 64) {
                const uint32_t w = path[k];
                const uint32_t dnext = k < lp ? (path[k + 1] & 3u) : (w & 3u), dprev = k > 0 ? (path[k - 1] & 3u) : (w & 3u);
                // closed time interval during which the agent is predicted on waypoint k
                int tlo, span;
                if (CUTILS) {  // w(t) = 0 for t = 0, min((t-1)/tpc + 1, lp) afterwards
                    tlo = k == 0 ? 0 : (k - 1) * tpc + 1;
                    span = k == 0 ? 1 : tpc;
                } else {       // w(t) = min(t / tpc, lp)
                    tlo = k * tpc;
                    span = tpc;
                }
                const bool to_end = k == lp || tlo + span - 1 >= tlast;
                const int key = key_of(X, (int)(w >> 2));
                if (X.tmask) {  // time buckets this item covers
                    const int b1 = min(tlo >> X.tshift, 63), b2 = min((to_end ? tlast : tlo + span - 1) >> X.tshift, 63);
                    const unsigned long long bits = ((2ull << b2) - 1ull) & ~((1ull << b1) - 1ull);
                    if (X.tmask_m2) {
                        const unsigned long long seen = atomicOr(&tmask[key], bits);
                        if (seen & bits) atomicOr(&tmask_m2[key], seen & bits);  // covered by a second item
                    } else {
                        atomicOr(&tmask[key], bits);
                    }
                }
                const uint32_t item = ((uint32_t)i << 20) | ((uint32_t)tlo << 11) | ((uint32_t)to_end << 10) |
                                      ((uint32_t)(span - 1) << 6) | (dprev << 4) | (dnext << 2) | (w & 3u);
                if (bk) {  // csr[key] stays the START of the key's list; the bucket's running offset is bumped
                    const int thi = to_end ? tlast : tlo + span - 1;
                    const int b1 = min(tlo >> OBS_BK_SHIFT, OBS_BK_NB - 1), b2 = min(thi >> OBS_BK_SHIFT, OBS_BK_NB - 1);
                    for (int bb = b1; bb <= b2; bb++) {
                        const int kb = key * OBS_BK_NB + bb;
                        const uint32_t old = atomicAdd(&bkc[kb >> 1], (kb & 1) ? 0x10000u : 1u);
                        csr_items[csr[key] + (int)((kb & 1) ? (old >> 16) : (old & 0xFFFFu))] = item;
                    }
                    continue;
                }
                const int slot = atomicAdd(&csr[key], 1);
                csr_items[slot] = item;
                if (k <= lp2) {  // the same waypoint in the upstream predictor's index: w(t) = min(t / tpc, lp)
                    const int tlo2 = k * tpc2, tlast2 = Tn2 - 1;
                    const bool to_end2 = k == lp2 || tlo2 + tpc2 - 1 >= tlast2;
                    const uint32_t dnext2 = k < lp2 ? dnext : (w & 3u);
                    if (P.use_tmask) {
                        const int b1 = min(tlo2 >> tshift2, 63), b2 = min((to_end2 ? tlast2 : tlo2 + tpc2 - 1) >> tshift2, 63);
                        const unsigned long long bits = ((2ull << b2) - 1ull) & ~((1ull << b1) - 1ull);
                        if (X.tmask_m2) {
                            const unsigned long long seen = atomicOr(&tmaskb[key], bits);
                            if (seen & bits) atomicOr(&tmaskb_m2[key], seen & bits);
                        } else {
                            atomicOr(&tmaskb[key], bits);
                        }
                    }
                    const int slot2 = atomicAdd(&csr2[key], 1);
                    items2[slot2] = ((uint32_t)i << 20) | ((uint32_t)tlo2 << 11) | ((uint32_t)to_end2 << 10) |
                                    ((uint32_t)(tpc2 - 1) << 6) | (dprev << 4) | (dnext2 << 2) | (w & 3u);
                }
            }
        }
        __syncthreads();
        if (bk) {  // bucket ends of every key to HBM (the node tables take their LDS back), list of key k = [csr[k], csr[k + 1])
            uint32_t *g = reinterpret_cast<uint32_t *>(S.bk_rel + (size_t)b * d.Rcap * OBS_BK_NB);
            for (int k = tid; k < K * OBS_BK_NB / 2; k += nt) g[k] = bkc[k];
            X.csr_end = csr + 1;
            X.bk_rel = S.bk_rel + (size_t)b * d.Rcap * OBS_BK_NB;
            __syncthreads();
        }
    }

    OBS_STAMP(4);
    // ---- phase 3: trees.  Pass A derives the topology of a tree from the static segment table (O(1) per node, one
    // BFS level per step); pass B evaluates the agent-dependent features with the visited cells of all nodes split
    // evenly over the lanes of the workgroup (wg_pass_b); then one lane per node writes its row.
    const float max_dist = (float)T;
    const int nwaves = nt >> 6;
    const bool items_in_lds = X.items_lds != nullptr;
    if (merged && items_in_lds && misc[5]) {  // (misc[5] was written before the barriers of the index build)
        trees_merged<true>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, max_dist);
    } else if (CUTILS) {
        if (items_in_lds) trees_cutils<true>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, a_vpos, a_pos, a_dir, a_state, a_speed, a_tslot, max_dist, STAGE != 2 && X.Tn > 0 && X.bk_rel == nullptr);
        else trees_cutils<false>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, a_vpos, a_pos, a_dir, a_state, a_speed, a_tslot, max_dist, STAGE != 2 && X.Tn > 0 && X.bk_rel == nullptr);
    } else if (P.max_depth <= 2) {
        if (items_in_lds) tree_upstream<32, 32, true>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta);
        else tree_upstream<32, 32, false>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta);
    } else {
        if (items_in_lds) tree_upstream<64, 88, true>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta);
        else tree_upstream<64, 88, false>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta);
    }
    OBS_STAMP(5);
#undef LDS_AT
#undef LDS_OPT
}

// MODE 0 = flatland_cutils outputs, 1 = upstream dense tree, 2 = both in one launch; VAR: see obs_body
template <int MODE, int VAR>
__global__ __launch_bounds__(OBS_NT) void k_obs(FlDev d, FlObsScratch S, ObsArgs P) {
    if (MODE == 0) obs_body<true, VAR, 0>(d, S, P);
    else if (MODE == 1) obs_body<false, VAR, 0>(d, S, P);
    else {
        obs_body<true, VAR, 1>(d, S, P);
        __syncthreads();
        obs_body<false, VAR, 2>(d, S, P);
    }
}

// ---------------------------------------------------------------------------------------------- host side
int fl_obs_alloc(FlObsScratch &o, const FlDev &d, hipStream_t s, std::vector<void *> &allocs) {
    o.pred_cap = FL_OBS_MAX_PRED + 2;
    o.items_cap = (size_t)d.A * (o.pred_cap + 2 * OBS_BK_NB + 2);  // bucketed lists: an item sits in every time bucket it touches
    const size_t BA = (size_t)d.B * d.A;
    void *p = nullptr;
    if (hipMalloc(&p, BA * o.pred_cap * 2) != hipSuccess) return FL_ERR_HIP;
    o.path = (uint16_t *)p; allocs.push_back(p);
    if (hipMalloc(&p, (size_t)d.B * o.items_cap * 4) != hipSuccess) return FL_ERR_HIP;
    o.cell_items = (uint32_t *)p; allocs.push_back(p);
    if (hipMalloc(&p, (size_t)d.B * 64 * 8) != hipSuccess) return FL_ERR_HIP;
    o.dbg = (long long *)p; allocs.push_back(p);
    if (hipMalloc(&p, (size_t)d.B * d.Rcap * OBS_BK_NB * 2 + 16) != hipSuccess) return FL_ERR_HIP;
    o.bk_rel = (uint16_t *)p; allocs.push_back(p);
    o.wl_cap = OBS_WL_HBM_ENTRIES;
    if (hipMalloc(&p, (size_t)d.B * o.wl_cap * 8) != hipSuccess) return FL_ERR_HIP;
    o.wl = (uint2 *)p; allocs.push_back(p);
    (void)s;
    return FL_OK;
}

void fl_obs_reset(FlObsScratch &o, const FlDev &d, const uint8_t *mask_dev, hipStream_t s) {
    // the sticky deadlock flags live in pk and are cleared by the agent reset kernel (flatland_cutils rebuilds its
    // DeadlockChecker in TreeObsForRailEnv::reset(), treeobs.cpp:22-28 / loader.cpp:207-219)
    (void)o; (void)d; (void)mask_dev; (void)s;
}

// what a launch may keep in LDS besides the arrays every launch needs
struct ObsOptions { int nt, wl_bytes, tab, nh, tmask, dual, items, snext, partial; };

// carve the dynamic LDS of a launch: every array the kernel uses, in one place (the kernel follows ObsLayout::off)
static ObsLayout obs_layout(const FlDev &d, const ObsArgs &P, const ObsOptions &o) {
    ObsLayout L;
    for (int k = 0; k < L_COUNT; k++) L.off[k] = L_ABSENT;
    size_t off = 0;
    auto put = [&](int which, size_t bytes) { L.off[which] = (unsigned)off; off += (bytes + 15) & ~(size_t)15; };
    const size_t R = d.Rcap, NS = R * 4, A = d.A, K1 = R + 1, U = d.Ucap;
    put(L_CELLW, R * 4);
    put(L_NBR, NS * 2);
    if (o.snext) put(L_SNEXT, NS * 2);
    if (d.rkey) put(L_RKEY, R * 2);
    put(L_SLOT_AGENT, A * 4); put(L_SLOT_READY, A * 4);
    put(L_CELL_TARGET, ((R + 31) / 32) * 4);
    put(L_A_SPEED, A * 8); put(L_A_TQ, A * 8);
    if (P.merged) { put(L_A_RAW, A * 32); put(L_RTYPE, R); }
    put(L_A_VPOS, A * 2); put(L_A_POS, A * 4); put(L_A_TSLOT, A * 2); put(L_A_TARGET, A * 2);
    put(L_A_MALF, A * 2); put(L_A_TPC, A * 2); put(L_A_LP, A * 2); put(L_A_N, A * 2);
    put(L_A_DIR, A); put(L_A_STATE, A); put(L_A_FREE, A); put(L_A_DEAD, A);
    put(L_MISC, 64 * 4); put(L_TEAM_META, 320 * 4);
    if (P.merged) put(L_WAVE_SCR, (size_t)(2 * d.A + 1) * (F_WORDS * 32) * 4);  // trees_merged: a slot per tree of either builder + the dummy
    else put(L_WAVE_SCR, (size_t)obs_scr_words(o.nt / 64, d.A, P.tw_c, P.tw_t, P.tpw_t) * 4);
    put(L_CSR, K1 * 4);
    // one pass B for both builders needs the room for twice the node tables: a tighter first-index copy (128 waypoints an agent)
    L.items_cap = P.merged ? (int)std::min<size_t>(OBS_ITEMS_LDS_CAP, std::max<size_t>(1024, A * 128)) : OBS_ITEMS_LDS_CAP;
    L.items2_cap = (int)std::min<size_t>(OBS_ITEMS2_CAP, A * (size_t)(P.tree_pred + 2));  // an agent has at most tree_pred + 1 of them
    if (o.items) put(L_ITEMS, (size_t)L.items_cap * 4);
    if (o.wl_bytes) put(L_WL, (size_t)o.wl_bytes);       // 0: the work lists live in HBM scratch
    if (o.partial || !o.wl_bytes) put(L_PARTIAL, (size_t)o.nt * 4);
    if (o.tmask) put(L_TMASK, K1 * 8);
    if (o.tmask && P.merged) put(L_TMASK2, K1 * 8);
    if (o.nh || o.tab) put(L_NH, U * R * 2);
    if (o.dual) {
        put(L_CSR2, K1 * 4);
        if (o.tmask) put(L_TMASKB, K1 * 8);
        if (o.tmask && P.merged) put(L_TMASKB2, K1 * 8);
        put(L_ITEMS2, (size_t)std::max(L.items2_cap, 4) * 4);
        put(L_A_LP2, A * 2); put(L_A_TPC2, A * 2); put(L_A_TQ2, A * 8);
    }
    if (o.tab) { put(L_SEG, NS * 16); put(L_DM, U * NS * 2); put(L_HOP8, U * NS * 2); }
    L.total = (unsigned)off;
    L.nt = o.nt; L.wl_bytes = o.wl_bytes; L.tab_lds = o.tab;
    return L;
}

// Choose what lives in LDS so that the workgroup fits 160 KiB.
static bool obs_pick_config(const FlDev &d, ObsArgs &P) {
    // diagnostic overrides (experiments on the LDS / occupancy trade-off): FL_OBS_NT, FL_OBS_LDS_LIMIT (bytes), FL_OBS_NO_TAB
    static const int force_nt = getenv("FL_OBS_NT") ? atoi(getenv("FL_OBS_NT")) : 0;
    static const size_t lds_limit = getenv("FL_OBS_LDS_LIMIT") ? (size_t)atol(getenv("FL_OBS_LDS_LIMIT")) : (size_t)160 * 1024;
    static const bool no_tab = getenv("FL_OBS_NO_TAB") != nullptr;
    // FL_OBS_FORCE="nt=512,wl=8192,tab=0,nh=1,tmask=1,dual=0,items=0,snext=1": only configurations with these values
    struct Force { int nt = -1, wl = -1, tab = -1, nh = -1, tmask = -1, dual = -1, items = -1, snext = -1; };
    static const Force force = [] {
        Force f;
        const char *e = getenv("FL_OBS_FORCE");
        if (!e) return f;
        const struct { const char *k; int *v; } keys[] = {{"nt=", &f.nt}, {"wl=", &f.wl}, {"tab=", &f.tab}, {"nh=", &f.nh},
                                                         {"tmask=", &f.tmask}, {"dual=", &f.dual}, {"items=", &f.items}, {"snext=", &f.snext}};
        for (const auto &kv : keys) {
            const char *q = strstr(e, kv.k);
            if (q && (q == e || q[-1] == ',')) *kv.v = atoi(q + strlen(kv.k));
        }
        return f;
    }();
    auto ok = [](int forced, int v) { return forced < 0 || forced == v; };
    const int nts[3] = {OBS_NT, 512, 256};
    const bool nh_fit = (size_t)d.Ucap * d.Rcap * 2 <= 24 * 1024;  // beyond that the next-hop tables stay in HBM / L2
    const bool dual_ok = P.tw_c != 0 && P.tw_t != 0 && P.tree_pred >= 0 &&
                         (long long)d.A * (P.pred_depth + 2) < 65536 && (long long)d.A * (P.tree_pred + 2) < 32768;
    static const int opts[5][3] = {{1, 1, 1}, {1, 0, 1}, {1, 0, 0}, {0, 0, 1}, {0, 0, 0}};  // masks, second index, items
    // Order of preference, from same-box A/B runs (tools/obs_sweep.py): the most wavefronts; the full work-list space; the LDS
    // copy of the items, the time masks and the second index; the successor table; the next-hop tables; own scan scratch.
    // The env's distance / segment / eight-hop tables join them when there is room left (small maps): they make no
    // difference in time (their gathers hit L2 and hide behind the rest) but replace narrow HBM gathers with one coalesced read.
    ObsOptions o;
    o.tab = 0;
    // Small envs, both builders: one pass B over the trees of both (trees_merged).  It needs a wavefront per two agents, the
    // second index, the items and the time masks in LDS, the successor table and keys = rail indices (the fast classify loop).
    static const bool no_merge = getenv("FL_OBS_NO_MERGE") != nullptr;
    P.merged = 0;
    if (!no_merge && dual_ok && P.max_depth <= 2 && d.A <= 31 && d.rkey == nullptr && (!force_nt || force_nt == OBS_NT) && ok(force.nt, OBS_NT) &&
        ok(force.tmask, 1) && ok(force.dual, 1) && ok(force.items, 1) && ok(force.snext, 1)) {
        P.merged = 1;
        o.nt = OBS_NT; o.tmask = 1; o.dual = 1; o.items = 1; o.snext = 1; o.partial = 1;
        // (the static tables in LDS with the work lists in HBM scratch instead measured 1.2 us slower on cfg2: longer staging)
        static const int merged_variants[2][2] = {{24 * 1024, 0}, {16 * 1024, 0}};
        for (int wk = 0; wk < 2 && P.merged; wk++) {
            o.wl_bytes = merged_variants[wk][0]; o.tab = merged_variants[wk][1];
            if (!ok(force.wl, o.wl_bytes) || !ok(force.tab, o.tab) || (o.tab && (no_tab || !nh_fit))) continue;
            // (the eight-hop table or the distance maps in LDS as well: no difference in time, same-box A/B)
            for (o.nh = nh_fit ? 1 : 0; o.nh >= 0; o.nh--) {
                if (!ok(force.nh, o.nh) || (o.tab && !o.nh)) continue;
                const ObsLayout L = obs_layout(d, P, o);
                if (L.total > lds_limit) continue;
                P.L = L; P.use_tmask = 1; P.dual_index = 1; P.bk = 0;
                static const int force_tshift_m = getenv("FL_OBS_TSHIFT") ? atoi(getenv("FL_OBS_TSHIFT")) : -1;
                P.tshift = force_tshift_m >= 0 ? force_tshift_m : 2;  // 4-step buckets: same-box A/B on cfg2, 54.8 us against 55.1 (2-step) and 56.9 (8-step)
                return true;
            }
        }
        P.merged = 0;
        o.tab = 0;
    }
    for (int k = 0; k < 3; k++) {
        o.nt = nts[k];
        if ((force_nt && o.nt != force_nt) || !ok(force.nt, o.nt)) continue;
        // work lists: in LDS when everything else fits beside them (small maps), else in HBM scratch (the LDS goes to the time
        // masks, the items and the second index; no cap on the entries), else LDS lists with whatever still fits
        for (int wk = 0; wk < 4; wk++) {
            o.wl_bytes = wk == 0 || wk == 2 ? 24 * 1024 : wk == 1 ? 0 : 8 * 1024;
            for (int opt = 0; opt < (wk == 0 ? 1 + !dual_ok : 5); opt++) {
                o.tmask = opts[opt][0]; o.dual = opts[opt][1]; o.items = opts[opt][2];
                if (o.dual && !dual_ok) continue;
                if (o.items && d.A * 32 > OBS_ITEMS_LDS_CAP) continue;  // hundreds of agents: their items never fit the LDS copy
                if (!ok(force.wl, o.wl_bytes) || !ok(force.tmask, o.tmask) || !ok(force.dual, o.dual) || !ok(force.items, o.items)) continue;
                for (o.snext = 1; o.snext >= 0; o.snext--)
                    for (o.nh = nh_fit ? 1 : 0; o.nh >= 0; o.nh--)
                        for (o.partial = 1; o.partial >= 0; o.partial--) {  // 4 KB of scan scratch: borrowed when tight
                            if (!ok(force.snext, o.snext) || !ok(force.nh, o.nh)) continue;
                            ObsLayout L = obs_layout(d, P, o);
                            if (L.total > lds_limit) continue;
                            if (!no_tab && force.tab != 0 && nh_fit && o.wl_bytes) {
                                ObsOptions ot = o;
                                ot.tab = 1;
                                const ObsLayout Lt = obs_layout(d, P, ot);
                                if (Lt.total <= lds_limit) L = Lt;
                                else if (force.tab == 1) continue;
                            } else if (force.tab == 1) continue;
                            P.L = L; P.use_tmask = o.tmask; P.dual_index = o.dual;
                            static const bool no_bk = getenv("FL_OBS_NO_BK") != nullptr;
                            P.bk = !no_bk && o.wl_bytes == 0 && !o.items && o.tmask && !o.dual && P.tw_c != 0 && P.pred_depth + 1 > 64 &&
                                   (size_t)obs_scr_words(o.nt / 64, d.A, P.tw_c, P.tw_t, P.tpw_t) * 4 >= (size_t)d.Rcap * OBS_BK_NB * 2;
                            // 2-step buckets where the traffic is and one catch-all bucket for late times (8-step buckets over the
                            // whole horizon measured slower on every map size)
                            static const int force_tshift = getenv("FL_OBS_TSHIFT") ? atoi(getenv("FL_OBS_TSHIFT")) : -1;
                            P.tshift = force_tshift >= 0 ? force_tshift : OBS_TSHIFT;
                            return true;
                        }
            }
        }
    }
    return false;
}

template <typename KernelT>
static int obs_launch(KernelT kern, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    static const bool verbose = getenv("FL_OBS_VERBOSE") != nullptr;  // diagnostic: the configuration obs_pick_config chose
    static int printed = 0;
    const ObsLayout &L = P.L;
    if (verbose && printed < 4) {
        printed++;
        fprintf(stderr, "[fl_obs] %d threads, %u B LDS: static tables in LDS %d, next-hop in LDS %d, successor table %d, work lists %d B, time masks %d, second index %d, items in LDS %d, one pass B for both builders %d\n",
                L.nt, L.total, L.tab_lds, L.off[L_NH] != L_ABSENT, L.off[L_SNEXT] != L_ABSENT, L.wl_bytes, P.use_tmask, P.dual_index, L.off[L_ITEMS] != L_ABSENT, P.merged);
    }
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(L.nt), L.total, s, d, o, P);
    return FL_OK;
}

static void obs_tree_args(ObsArgs &P, int max_depth, int tree_pred, double *tree_out) {
    P.max_depth = max_depth; P.tree_pred = tree_pred; P.tree_out = tree_out;
    int n = 0, p = 1;
    for (int k = 0; k <= max_depth; k++) { n += p; p *= 4; }
    P.n_tree_nodes = n;
    P.tw_t = max_depth <= 2 ? F_WORDS * 32 : F_WORDS * 88;
    P.tpw_t = max_depth <= 2 ? 2 : 1;
}

int fl_launch_obs_cutils(FlObsScratch &o, const FlDev &d, int max_nodes, int pred_depth, float *attr, float *forest,
                         int32_t *adjacency, int32_t *node_order, int32_t *edge_order, uint8_t *valid, double *props,
                         hipStream_t s) {
    if (d.A > 1024 || pred_depth + 2 > o.pred_cap || pred_depth > 510 || max_nodes > FL_OBS_MAX_NODES) return FL_ERR_ARG;
    ObsArgs P = {};
    P.max_nodes = max_nodes; P.pred_depth = pred_depth; P.attr = attr; P.forest = forest; P.adjacency = adjacency;
    P.node_order = node_order; P.edge_order = edge_order; P.valid = valid; P.props = props; P.dbg = o.dbg;
    P.tw_c = F_WORDS * 32;
    if (!obs_pick_config(d, P)) return FL_ERR_ARG;
    return P.L.tab_lds ? obs_launch(k_obs<0, 1>, d, o, P, s) : P.L.wl_bytes == 0 ? obs_launch(k_obs<0, 2>, d, o, P, s) : obs_launch(k_obs<0, 0>, d, o, P, s);
}

int fl_launch_obs_both(FlObsScratch &o, const FlDev &d, int max_nodes, int pred_depth, float *attr, float *forest,
                       int32_t *adjacency, int32_t *node_order, int32_t *edge_order, uint8_t *valid, double *props,
                       int max_depth, int tree_pred, double *tree_out, hipStream_t s) {
    if (d.A > 1024 || pred_depth + 2 > o.pred_cap || pred_depth > 510 || max_nodes > FL_OBS_MAX_NODES) return FL_ERR_ARG;
    if (max_depth > 3 || tree_pred > pred_depth || tree_pred < 0) return FL_ERR_ARG;  // the upstream path must be a prefix
    ObsArgs P = {};
    P.max_nodes = max_nodes; P.pred_depth = pred_depth; P.attr = attr; P.forest = forest; P.adjacency = adjacency;
    P.node_order = node_order; P.edge_order = edge_order; P.valid = valid; P.props = props; P.dbg = o.dbg;
    P.tw_c = F_WORDS * 32;
    obs_tree_args(P, max_depth, tree_pred, tree_out);
    if (!obs_pick_config(d, P)) return FL_ERR_ARG;
    return P.L.tab_lds ? obs_launch(k_obs<2, 1>, d, o, P, s) : P.L.wl_bytes == 0 ? obs_launch(k_obs<2, 2>, d, o, P, s) : obs_launch(k_obs<2, 0>, d, o, P, s);
}

int fl_launch_obs_tree(FlObsScratch &o, const FlDev &d, int max_depth, int pred_depth, double *out, hipStream_t s) {
    if (d.A > 1024 || pred_depth + 2 > o.pred_cap || pred_depth > 510) return FL_ERR_ARG;
    if (max_depth > 3) return FL_ERR_ARG;  // one lane per node of the deepest level: 4^3 = 64
    ObsArgs P = {};
    P.dbg = o.dbg;
    obs_tree_args(P, max_depth, pred_depth, out);
    if (!obs_pick_config(d, P)) return FL_ERR_ARG;
    return P.L.tab_lds ? obs_launch(k_obs<1, 1>, d, o, P, s) : P.L.wl_bytes == 0 ? obs_launch(k_obs<1, 2>, d, o, P, s) : obs_launch(k_obs<1, 0>, d, o, P, s);
}

// diagnostic: the configuration obs_pick_config chooses for the fused launch (cutils + upstream tree of max_depth):
// threads, LDS bytes, static tables in LDS, next-hop in LDS, work-list bytes, time masks, second index, items in LDS
int fl_obs_config_of_fused(const FlDev &d, int pred_depth, int max_depth, int tree_pred, int out[8]) {
    ObsArgs P = {};
    P.pred_depth = pred_depth;
    P.tw_c = F_WORDS * 32;
    obs_tree_args(P, max_depth, tree_pred, nullptr);
    if (!obs_pick_config(d, P)) return FL_ERR_ARG;
    out[0] = P.L.nt; out[1] = (int)P.L.total; out[2] = P.L.tab_lds; out[3] = P.L.off[L_NH] != L_ABSENT; out[4] = P.L.wl_bytes; out[5] = P.use_tmask;
    out[6] = P.dual_index; out[7] = P.L.off[L_ITEMS] != L_ABSENT;
    return FL_OK;
}
