// fl_obs_layout.h -- what the host side (fl_obs.hip: LDS carving, launch configuration) and the device side (fl_obs_*.h) of
// the observation kernels share: tunables, the packing of a prediction item, the node tables of the trees, the list of LDS
// arrays of a launch and the launch arguments.
#pragma once
#include <stdint.h>

#include "fl_obs.h"

#define OBS_NT 1024
// items of a key's list scanned per conflict work-list entry: 16 where the items sit in LDS, 8 where they sit in HBM scratch (there the
// whole chunk is requested at once, and a lane that waits for 16 scattered words waits longer than two lanes for 8 each -- same-box
// A/B round 4: cfg4 0.356 -> 0.341 ms, cfg5 1.352 -> 1.306 ms with 8; cfg3, items in LDS, 0.656 -> 0.670 ms with 8 and 0.700 with 24)
#ifndef CF_CHUNK_LDS
#define CF_CHUNK_LDS 16
#endif
#ifndef CF_CHUNK_HBM
#define CF_CHUNK_HBM 8
#endif
#ifndef CF_FIRST_HBM
#define CF_FIRST_HBM 8              // items the lane of a conflict entry scans itself before the list is split in chunks (HBM items; 16: cfg4 -2.5 %, 24: -5 %)
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
#ifndef CF_DIRECT
#define CF_DIRECT 32                 // when no list of the env is longer, every conflict entry is scanned by its lane alone, in one pass
#endif
#define OBS_ITEMS2_CAP 4096          // items of the second (upstream) index built by stage 1 of the fused launch
#ifndef OBS_ITEMS_LDS_CAP
#define OBS_ITEMS_LDS_CAP 6144       // prediction items are kept in LDS when an env has at most this many (else HBM scratch)
#endif
#define OBS_WL_HBM_ENTRIES 32768     // pass B work-list entries per env when the lists live in HBM scratch (large maps)
#define OBS_PRED_CAP (FL_OBS_MAX_PRED + 2)   // waypoints kept per agent in the path scratch (FlObsScratch::pred_cap: a constant of the build)
// Large maps (items in HBM, hundreds of agents): the items are grouped by bucket of 64 time steps and laid out BUCKET-major (all
// items of bucket 0 by key, then bucket 1, ...: a bucket's share of an env's items stays in the L2 while it is written), an item
// sits in every bucket its interval touches, and a conflict query reads the key's items of the one or two buckets its three time
// steps fall in (an eighth of a busy cell's list instead of all of it).
#define OBS_BK_NB 8
#define OBS_BK_SHIFT 6
// Small maps (index in LDS): sixteen buckets of 32 steps, their offsets in an LDS array of their own -- a query scans the
// handful of items around its time instead of everybody who ever passes the cell (20 items a query at 80 agents).
#ifndef OBS_FB_NB
#define OBS_FB_NB 16
#define OBS_FB_SHIFT 5
#endif
#define OBS_FB_MSHIFT (OBS_FB_NB > 16 ? 1 : 0)   // the 16-bit mask of the buckets in which a stay-to-the-end item starts: bucket >> this

// prediction item: one (agent, waypoint) with the closed time interval during which the agent is predicted there
//   bits 0-1 direction at the waypoint, 2-3 direction at the next waypoint, 4-5 at the previous one,
//   6-11 interval length - 1 (= time steps per cell - 1 <= 63: speeds down to 1/64), 12 "until the end of the horizon", 13-21 t_lo, 22-31 agent
#define IT_DIR(it) ((it)&3u)
#define IT_DNEXT(it) (((it) >> 2) & 3u)
#define IT_DPREV(it) (((it) >> 4) & 3u)
#define IT_THI(it, tlast) ((((it) >> 12) & 1u) ? (uint32_t)(tlast) : IT_TLO(it) + (((it) >> 6) & 63u))
#define IT_TLO(it) (((it) >> 13) & 511u)
#define IT_TOEND(it) (((it) >> 12) & 1u)
#define IT_AGENT(it) ((int)((it) >> 22))
#define IT_MAKE(agent, tlo, to_end, span, dprev, dnext, dir) \
    (((uint32_t)(agent) << 22) | ((uint32_t)(tlo) << 13) | ((uint32_t)(to_end) << 12) | ((uint32_t)((span) - 1) << 6) | ((uint32_t)(dprev) << 4) | ((uint32_t)(dnext) << 2) | (uint32_t)(dir))

// Node table of one tree (= one pass B team) in LDS: field-major, CAP entries per 32-bit word-field.  Descriptor words are
// written by pass A, the accumulators are merged by the event handlers of pass B with LDS atomics.
//   N_SE    start state (lo16, 0xFFFF = no node) | end state (hi16)
//   N_TV    tot_dist at the first visited cell (lo16) | number of visited cells (hi16)
//   N_UF    tot_dist of the first unusable switch (lo16, 0xFFFF = none) | ND_* flags (bits 16-21) | DFS row of the node (bits 22-31, upstream)
//   N_INCL  inclusive prefix of the visit counts (24 bits) | next node with cells << 24 (0xFF = none)   (team_prepare)
//   N_OA, N_PC   min tot_dist of "other agent encountered" / "potential conflict" (0x7fffffff = none)
//   N_CNT   agents in the same direction (lo16) | in the opposite direction (hi16)
//   N_RM    agents ready to depart (lo16) | flatland_cutils: "some occupant is malfunctioning" (bit 16)
//   N_MS    slowest occupant in the same direction: speed rank << 10 | agent (0xFFFFFFFF = none); the speed is static per agent
//           and the ranks are computed on the host, so the minimum of a double is a 32-bit atomicMin
//   N_PH    flatland_cutils: parent + 2 (bits 0-7) | (first child's node index << 2 | action + 1) << 8
//   N_OT, N_MALF   upstream: min tot_dist of "other target encountered", max malfunction down counter of an occupant
enum { N_SE = 0, N_TV, N_UF, N_INCL, N_OA, N_PC, N_CNT, N_RM, N_MS, N_PH, N_WORDS_C = 10, N_OT = 9, N_MALF = 10, N_WORDS_T = 11 };
#define N_NONE 0xFFFFu
enum { ND_TARGET = 1, ND_SWITCH = 2, ND_DEAD_END = 4, ND_TERMINAL = 8, ND_ZERO = 16, ND_PHANTOM = 32 /* the walk left the rail: distance inf */ };

// flatland_cutils trees: 32-lane team, 32 slots (max_nodes <= 32).  Upstream trees: on maps where no (cell, direction) has
// more than two transitions -- every Flatland rail cell type -- level L has at most 2^L nodes, a depth-3 tree 14: a 16-lane
// team with 16 slots (four trees a wavefront).  Any other grid: one slot per DFS row (32 lanes / 32 slots up to depth 2,
// 64 lanes / 88 slots at depth 3).
#define OBS_CAP_C 32
#define OBS_CAP_T_COMPACT 16

// LDS words of the trees' node tables: one slot per team that can hold an agent plus one dummy slot that the idle teams share
__host__ __device__ constexpr int obs_scr_words(int nwaves, int A, int tw_c, int tw_t, int tpw_t) {
    const int tpw_c = tw_c > N_WORDS_C * OBS_CAP_C ? 1 : 2;   // (64-slot tables of max_nodes > 32: one flatland_cutils tree a wavefront)
    const int n_c = tpw_c * nwaves <= A ? tpw_c * nwaves : A + 1, n_t = tpw_t * nwaves <= A ? tpw_t * nwaves : A + 1;
    const int w_c = n_c * tw_c, w_t = n_t * tw_t;
    return w_c > w_t ? w_c : w_t;
}

// LDS arrays of a launch, in carving order (obs_layout on the host decides which exist and where)
enum { L_CELLW = 0, L_NBR, L_SNEXT, L_RKEY, L_SLOT_AGENT, L_SLOT_READY, L_CELL_TARGET, L_A_SPEED, L_A_VPOS, L_A_POS, L_A_TSLOT,
       L_A_TARGET, L_A_MALF, L_A_TPC, L_A_TQ, L_A_TQ2, L_A_RAW, L_RTYPE, L_A_LP, L_A_N, L_A_SRANK, L_A_DIR, L_A_STATE, L_A_FREE, L_A_DEAD, L_MISC, L_TEAM_META, L_WAVE_SCR,
       L_CSR, L_ITEMS, L_WL, L_PARTIAL, L_TMASK, L_TMASK2, L_NH, L_CSR2, L_TMASKB, L_TMASKB2, L_ITEMS2, L_A_LP2, L_A_TPC2, L_BKREL, L_SEG, L_DM, L_HOP8, L_COUNT };
#define L_ABSENT 0xFFFFFFFFu
struct ObsLayout {
    unsigned off[L_COUNT];  // byte offset into the dynamic LDS, L_ABSENT = not in this launch
    unsigned total;         // bytes of dynamic LDS
    int nt;                 // threads per workgroup
    int wl_bytes;           // size of the pass B work lists
    int wl_head;            // work lists in HBM scratch (wl_bytes 0): bytes of their LDS head (L_WL), 0 = none
    int tab_lds;            // the env's dm / seg / nh / hop8 tables are staged in LDS (kernel template TAB_LDS)
    int items_cap, items2_cap;  // entries of the LDS copies of the prediction items (first / second index); an env with more
                                // falls back to the items in HBM scratch / to the two stages
};

// ---- the carving of the dynamic LDS, one function for the host (obs_pick_config: any env size) and for the kernels whose layout
// is a compile-time constant (FIXED launch classes below: every LDS base is an immediate instead of a scalar register)
// what a launch may keep in LDS besides the arrays every launch needs
struct ObsOptions { int nt, wl_bytes, tab, nh, tmask, dual, items, snext, partial, bk_room, own_filter, fb, raw, items_cap, wl_head; };
struct ObsDims { int Rcap, A, Ucap, rkey; };                   // capacities of the batch (rkey: colliding prediction keys, H > W)
struct ObsShape { int merged, tw_c, tw_t, tpw_t, tree_pred; };  // what of ObsArgs decides sizes

__host__ __device__ constexpr unsigned obs_al16(unsigned long long bytes) { return (unsigned)((bytes + 15ull) & ~15ull); }
__host__ __device__ constexpr ObsLayout obs_layout_c(const ObsDims &d, const ObsShape &P, const ObsOptions &o) {
    ObsLayout L = {};
    for (int k = 0; k < L_COUNT; k++) L.off[k] = L_ABSENT;
    unsigned off = 0;
#define OBS_PUT(which, bytes) do { L.off[which] = off; off += obs_al16((unsigned long long)(bytes)); } while (0)
    const unsigned long long R = (unsigned long long)d.Rcap, NS = R * 4, A = (unsigned long long)d.A, K1 = R + 1, U = (unsigned long long)d.Ucap;
    OBS_PUT(L_CELLW, R * 4);
    OBS_PUT(L_NBR, NS * 2);
    if (o.snext) OBS_PUT(L_SNEXT, NS * 2);
    if (d.rkey) OBS_PUT(L_RKEY, R * 2);
    OBS_PUT(L_SLOT_AGENT, A * 4); OBS_PUT(L_SLOT_READY, A * 4);
    OBS_PUT(L_CELL_TARGET, ((R + 31) / 32) * 4);
    OBS_PUT(L_A_SPEED, A * 8); OBS_PUT(L_A_TQ, A * 8);
    if (P.merged && o.raw) { OBS_PUT(L_A_RAW, A * 32); OBS_PUT(L_RTYPE, R); }   // the agents' raw words and the road types (attribute rows)
    OBS_PUT(L_A_VPOS, A * 2); OBS_PUT(L_A_POS, A * 4); OBS_PUT(L_A_TSLOT, A * 2); OBS_PUT(L_A_TARGET, A * 2);
    OBS_PUT(L_A_MALF, A * 2); OBS_PUT(L_A_TPC, A * 2); OBS_PUT(L_A_LP, A * 2); OBS_PUT(L_A_N, A * 2); OBS_PUT(L_A_SRANK, A * 2);
    OBS_PUT(L_A_DIR, A); OBS_PUT(L_A_STATE, A); OBS_PUT(L_A_FREE, A); OBS_PUT(L_A_DEAD, A);
    OBS_PUT(L_MISC, 64 * 4); OBS_PUT(L_TEAM_META, 320 * 4);
    {
        // trees_merged: 32 flatland_cutils + 32 compact upstream tables a round; else a slot per team + the dummy.  Large maps
        // borrow this space for the per-(key, time bucket) counters while the bucketed index is built (P.bk): room for those too
        // (a round = one flatland_cutils tree per 32 lanes: 32 agents on 1024 threads, 16 on 512)
        unsigned long long scr = P.merged ? (unsigned long long)(o.nt / 32) * (N_WORDS_C * OBS_CAP_C + (P.tw_t != 0 ? N_WORDS_T * OBS_CAP_T_COMPACT : 0)) * 4
                                          : (unsigned long long)obs_scr_words(o.nt / 64, d.A, P.tw_c, P.tw_t, P.tpw_t) * 4;
        const unsigned long long bkc = (R + 1) * OBS_BK_NB * 2 + 4;
        if (o.bk_room && bkc > scr) scr = bkc;
        OBS_PUT(L_WAVE_SCR, scr);
    }
    OBS_PUT(L_CSR, K1 * 4);
    // one pass B for both builders needs the room for twice the node tables: a tighter first-index copy (128 waypoints an agent)
    const unsigned long long cap1 = o.items_cap ? (unsigned long long)o.items_cap : (unsigned long long)OBS_ITEMS_LDS_CAP;
    const unsigned long long own_cap = A * 128 > 1024 ? A * 128 : 1024;
    L.items_cap = (P.merged && o.own_filter) ? (int)(cap1 < own_cap ? cap1 : own_cap) : (int)cap1;
    const unsigned long long i2 = A * (unsigned long long)(P.tree_pred + 2);  // an agent has at most tree_pred + 1 of them
    L.items2_cap = (int)(i2 < OBS_ITEMS2_CAP ? i2 : OBS_ITEMS2_CAP);
    if (o.items) OBS_PUT(L_ITEMS, (unsigned long long)L.items_cap * 4);
    if (o.wl_bytes) OBS_PUT(L_WL, o.wl_bytes);       // 0: the work lists live in HBM scratch ...
    else if (o.wl_head) OBS_PUT(L_WL, o.wl_head);    // ... but for their first entries (LDS head)
    if (o.partial || !o.wl_bytes) OBS_PUT(L_PARTIAL, (unsigned long long)o.nt * 4);
    if (o.tmask) OBS_PUT(L_TMASK, K1 * 8);
    if (o.tmask && P.merged && o.own_filter) OBS_PUT(L_TMASK2, K1 * 8);
    if (o.dual) {
        OBS_PUT(L_CSR2, K1 * 4);
        if (o.tmask) OBS_PUT(L_TMASKB, K1 * 8);
        if (o.tmask && P.merged && o.own_filter) OBS_PUT(L_TMASKB2, K1 * 8);
        OBS_PUT(L_ITEMS2, (unsigned long long)(L.items2_cap > 4 ? L.items2_cap : 4) * 4);
        OBS_PUT(L_A_LP2, A * 2); OBS_PUT(L_A_TPC2, A * 2); OBS_PUT(L_A_TQ2, A * 8);
    }
    if (o.fb) OBS_PUT(L_BKREL, K1 * (OBS_FB_NB + 2) * 2);
    // the arrays sized by the number of unique targets come last: a launch class with a compile-time layout fixes everything above
    if (o.nh || o.tab) OBS_PUT(L_NH, U * R * 2);
    if (o.tab) { OBS_PUT(L_SEG, NS * 16); OBS_PUT(L_DM, U * NS * 2); OBS_PUT(L_HOP8, U * NS * 2); }
#undef OBS_PUT
    L.total = off;
    L.nt = o.nt; L.wl_bytes = o.wl_bytes; L.tab_lds = o.tab; L.wl_head = o.wl_bytes ? 0 : o.wl_head;
    return L;
}

// FIXED launch classes: the pinned configurations of the small-env kernels with every capacity rounded up to a class value, so that
// the whole carving is a compile-time constant -- an LDS base is an immediate of the ds_* instruction instead of one of ~50 scalar
// registers the kernel cannot keep (311-373 spilled SGPRs before, every one a v_readlane / v_writelane in the hot loops).  The
// host (obs_pick_config) takes a class when the batch fits its capacities AND its own choice of options is the class's; any
// other batch runs the same kernel with the runtime layout (FIX 0).  The next-hop tables (sized by the unique targets) come last
// in the carving: their base is fixed, their size is the batch's.
// A class is a BASELINE configuration: besides the capacities it fixes the builders' parameters (31 nodes, predictor depths 500 / 30,
// the upstream tree's depth) and, for the larger ones, the exact number of agents -- all constants in its kernel.
//   FIX 1: one round of trees for both builders (MODE 3): at most 32 agents, 256 rail cells, depth 2 -- cfg1, cfg2 (BASELINE configs[0..1])
#define OBS_FIX3_RCAP 680
#define OBS_FIX4_RCAP 2816   // classes 4 / 9 (round 6: bins): cfg5 (2 680 rail cells) and, with the LDS successor table still fitting, Test_12 (2 745) and
#define OBS_FIX4_A 432       // the first level of Test_14 (2 807 cells, 425 agents)
#define OBS_FIX3_WL_HEAD (10 * 1024)
#ifndef OBS_WL_HEAD_MAX
#define OBS_WL_HEAD_MAX (32 * 1024)   // (round 6: 16 -> 32 KB, the builder alone at cfg4 0.243 -> 0.240 ms same box; both builders never have that much left)
#endif
// LDS head of HBM work lists: whatever the carving leaves, in KB steps, at most this, at least OBS_WL_HEAD_MIN
#define OBS_WL_HEAD_MIN (4 * 1024)
#define OBS_ALONE_LDS_LISTS_RCAP 320   // the flatland_cutils builder alone, rounds of 32 agents: LDS work lists up to this many rail cells, HBM lists with an LDS head beyond
template <int FIX> struct ObsFixed;
// EXACT classes (1, 2, 3, 5, 6, 7, 10) are BASELINE configurations down to the number of agents and the upstream depth -- constants in their
// kernels.  BIN classes (round 6) keep the compile-time carving but take the agents as an upper bound (dims.A, `agents` 0) and the depth from
// the call (`max_depth` 0): same-box cost against the exact class 3 % at cfg3, 1.7 % at cfg4, nothing measurable at cfg5 and for the builder
// alone (runtime carving: 7 - 8 %) -- so 4, 8, 9 are bins themselves and 2, 3, 7 have bin twins (12, 13, 17); 11 = class 1 at any depth;
// 14 / 19 = the large-map classes without the LDS successor table: 432 agents / 3 072 rail cells (Test_14; the larger levels of cfg5's row).
// options of the classes of the flatland_cutils builder alone: what obs_pick_config chooses at the classes' capacities (tests/test_obs_config.py)
//                      nt, wl_bytes, tab, nh, tmask, dual, items, snext, partial, bk_room, own_filter, fb, raw, items_cap, wl_head
#define OBS_FIX7_OPT  {OBS_NT, 36 * 1024, 0, 0, 1, 0, 1, 1, 1, 0, 1, 1, 1, OBS_ITEMS_LDS_CAP, 0}
#define OBS_FIX8_OPT  {OBS_NT, 0, 0, 0, 1, 0, 0, 1, 1, 0, 1, 1, 1, OBS_ITEMS_LDS_CAP, OBS_WL_HEAD_MAX}
#define OBS_FIX9_OPT  {OBS_NT, 0, 0, 0, 1, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0}
#define OBS_FIX10_OPT {512, 16 * 1024, 0, 1, 1, 0, 1, 1, 1, 0, 1, 0, 1, OBS_ITEMS_LDS_CAP, 0}
template <> struct ObsFixed<1> {
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 2;   // the builders' parameters of the class (tree_pred: shape)
    static constexpr int agents = 0;   // agents per env, exactly (0 = any number up to dims.A)
    static constexpr ObsDims dims = {256, 32, 0, 0};
    static constexpr ObsShape shape = {1, N_WORDS_C * OBS_CAP_C, N_WORDS_T * OBS_CAP_T_COMPACT, 4, 30};
    //                                  nt, wl_bytes, tab, nh, tmask, dual, items, snext, partial, bk_room, own_filter, fb, raw, items_cap
    static constexpr ObsOptions opt = {OBS_NT, 24 * 1024, 0, 1, 1, 1, 1, 1, 1, 0, 1, 0, 1, OBS_ITEMS_LDS_CAP};
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
//   FIX 2: rounds of 32 agents, work lists in LDS (MODE 4, VAR 0): 80 agents, at most 232 rail cells, depth 3 -- cfg3 (BASELINE configs[2])
//   FIX 3: rounds of 32 agents, work lists in HBM scratch with an LDS head (MODE 4, VAR 2): 80 agents, at most 680 rail cells -- every
//          level of the Round-2 row (Test_8: 603 .. 677) --, depth 2 -- cfg4 (configs[3]).  No LDS copy of the items: every cfg4 env has
//          more items than the 4 096 entries that fitted (round 4: the copy was dead weight); its 16 KB are the larger maps and the head.
//   FIX 4: two stages, hundreds of agents (MODE 2, VAR 2): at most 432 agents / 2816 rail cells, any depth (a bin since round 6) -- cfg5 (configs[4]), Test_12, Test_14
// The LDS of these three is full to the last few hundred bytes (that is how obs_pick_config chose their options), so the classes
// are the BASELINE maps' own sizes rounded up to a multiple of 8 / 16 rail cells; tests/test_obs_config.py checks that each class
// IS what obs_pick_config chooses at the class's capacities.
template <> struct ObsFixed<2> {
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 3;   // the builders' parameters of the class (tree_pred: shape)
    static constexpr int agents = 80;   // agents per env, exactly (0 = any number up to dims.A)
    static constexpr ObsDims dims = {232, 80, 0, 0};
    static constexpr ObsShape shape = {2, N_WORDS_C * OBS_CAP_C, N_WORDS_T * OBS_CAP_T_COMPACT, 4, 30};
    static constexpr ObsOptions opt = {OBS_NT, 36 * 1024, 0, 0, 1, 1, 1, 1, 1, 0, 1, 1, 1, 4096};
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
template <> struct ObsFixed<3> {
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 2;   // the builders' parameters of the class (tree_pred: shape)
    static constexpr int agents = 80;   // agents per env, exactly (0 = any number up to dims.A)
    static constexpr ObsDims dims = {OBS_FIX3_RCAP, 80, 0, 0};
    static constexpr ObsShape shape = {2, N_WORDS_C * OBS_CAP_C, N_WORDS_T * OBS_CAP_T_COMPACT, 4, 30};
    static constexpr ObsOptions opt = {OBS_NT, 0, 0, 0, 1, 1, 0, 1, 1, 0, 1, 1, 1, OBS_ITEMS_LDS_CAP, OBS_FIX3_WL_HEAD};
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
template <> struct ObsFixed<4> {
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 0;   // (a bin: the call's depth, 1 .. 3)
    static constexpr int agents = 0;   // agents per env, exactly (0 = any number up to dims.A)
    static constexpr ObsDims dims = {OBS_FIX4_RCAP, OBS_FIX4_A, 0, 0};
    static constexpr ObsShape shape = {0, N_WORDS_C * OBS_CAP_C, N_WORDS_T * OBS_CAP_T_COMPACT, 4, 30};
    static constexpr ObsOptions opt = {OBS_NT, 0, 0, 0, 1, 0, 0, 1, 1, 1, 0, 0, 0, 0};
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
//   FIX 5: class 1's envs (at most 32 agents, 256 rail cells, depth 2) in rounds of 16 agents on 512 threads and at most 80 KB of LDS
//   (MODE 5): TWO workgroups a CU.  For batches of several envs per CU -- there one env's barriers and round trips are filled by the
//   other's issue (same-box sweep at the cfg2 shape, profiles/r05_cfg2_bsweep.json); a batch of at most one env per CU keeps class 1.
template <> struct ObsFixed<5> {
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 2;
    static constexpr int agents = 0;
    static constexpr ObsDims dims = {256, 32, 0, 0};
    static constexpr ObsShape shape = {3, N_WORDS_C * OBS_CAP_C, N_WORDS_T * OBS_CAP_T_COMPACT, 4, 30};
    static constexpr ObsOptions opt = {512, 16 * 1024, 0, 0, 1, 1, 1, 1, 1, 0, 1, 0, 0, 2048};
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
// ---- classes 6 .. 10: the flatland_cutils builder ALONE (fl_obs_cutils / fl_step_obs without a tree -- the launch the reference's
// solution makes: solution/eval_env.py:15-17, demo.py:39 build TreeCutils(31, 500) only), the counterparts of classes 1 .. 5 on the same
// machinery without the second index and the upstream tables (shape.tw_t = 0; MODE 6 / 7 / 8 = MODE 3 / 4 / 5 with UP = false, class 9: MODE 0)
template <> struct ObsFixed<6> {   // at most 32 agents / 256 rail cells, one round (cfg1, cfg2)
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 0;
    static constexpr int agents = 0;
    static constexpr ObsDims dims = {256, 32, 0, 0};
    static constexpr ObsShape shape = {1, N_WORDS_C * OBS_CAP_C, 0, 0, 0};
    static constexpr ObsOptions opt = {OBS_NT, 24 * 1024, 0, 1, 1, 0, 1, 1, 1, 0, 1, 0, 1, OBS_ITEMS_LDS_CAP};
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
template <> struct ObsFixed<7> {   // 80 agents, at most 232 rail cells, rounds of 32 agents, work lists in LDS (cfg3)
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 0;
    static constexpr int agents = 80;
    static constexpr ObsDims dims = {232, 80, 0, 0};
    static constexpr ObsShape shape = {2, N_WORDS_C * OBS_CAP_C, 0, 0, 0};
    static constexpr ObsOptions opt = OBS_FIX7_OPT;
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
template <> struct ObsFixed<8> {   // 80 agents, at most 680 rail cells, rounds of 32 agents (cfg4: every level of the Round-2 row)
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 0;
    static constexpr int agents = 0;
    static constexpr ObsDims dims = {OBS_FIX3_RCAP, 80, 0, 0};
    static constexpr ObsShape shape = {2, N_WORDS_C * OBS_CAP_C, 0, 0, 0};
    static constexpr ObsOptions opt = OBS_FIX8_OPT;
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
template <> struct ObsFixed<9> {   // at most 432 agents / 2816 rail cells: the stand-alone kernel (MODE 0, VAR 2) with its carving compiled in (cfg5)
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 0;
    static constexpr int agents = 0;
    static constexpr ObsDims dims = {OBS_FIX4_RCAP, OBS_FIX4_A, 0, 0};
    static constexpr ObsShape shape = {0, N_WORDS_C * OBS_CAP_C, 0, 0, 0};
    static constexpr ObsOptions opt = OBS_FIX9_OPT;
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
template <> struct ObsFixed<10> {   // class 6's envs in rounds of 16 agents on 512 threads, two workgroups a CU (batches of several envs per CU)
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 0;
    static constexpr int agents = 0;
    static constexpr ObsDims dims = {256, 32, 0, 0};
    static constexpr ObsShape shape = {3, N_WORDS_C * OBS_CAP_C, 0, 0, 0};
    static constexpr ObsOptions opt = OBS_FIX10_OPT;
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
// ---- bin classes (see above): twins of exact classes with the agents as an upper bound and the call's depth, and the larger large-map classes
template <> struct ObsFixed<11> : ObsFixed<1> { static constexpr int max_depth = 0; };
template <> struct ObsFixed<12> : ObsFixed<2> { static constexpr int max_depth = 0, agents = 0; };
template <> struct ObsFixed<13> : ObsFixed<3> { static constexpr int max_depth = 0, agents = 0; };
template <> struct ObsFixed<17> : ObsFixed<7> { static constexpr int agents = 0; };
#define OBS_FIXB_RCAP 3200
#define OBS_FIXB_A 432
template <> struct ObsFixed<14> {   // both builders, two stages, no LDS successor table: at most 432 agents / 3 072 rail cells, any depth
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 0;
    static constexpr int agents = 0;
    static constexpr ObsDims dims = {OBS_FIXB_RCAP, OBS_FIXB_A, 0, 0};
    static constexpr ObsShape shape = {0, N_WORDS_C * OBS_CAP_C, N_WORDS_T * OBS_CAP_T_COMPACT, 4, 30};
    static constexpr ObsOptions opt = {OBS_NT, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0};
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
// 21: class 7's envs (the builder alone, at most 80 agents / 232 rail cells) in rounds of 16 agents on 512 threads and at most 80 KB of LDS, TWO
// workgroups a CU, for batches of several envs per CU (obs_batch_is_wide) -- the builder alone leaves the LDS that the two-a-CU kernel of both
// builders lacked (round 4: cfg3 0.715 -> 0.81 ms): same box, runtime carving both sides, cfg3 at 1 024 envs 0.467 -> 0.419 ms
template <> struct ObsFixed<21> {
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 0;
    static constexpr int agents = 0;
    static constexpr ObsDims dims = {232, 80, 0, 0};
    static constexpr ObsShape shape = {3, N_WORDS_C * OBS_CAP_C, 0, 0, 0};
    static constexpr ObsOptions opt = {512, 16 * 1024, 0, 0, 1, 0, 1, 1, 1, 0, 1, 1, 1, 4096, 0};
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
// 15 / 18: rounds of 32 agents on maps beyond class 3's: at most 100 agents / 1 344 rail cells (Test_10: 1 265 / 1 319), HBM work lists with an
// LDS head, no LDS copy of the items (a hundred paths across a 100 x 80 map are more items than any copy that fits)
#define OBS_FIX15_RCAP 1344
template <> struct ObsFixed<15> {
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 0;
    static constexpr int agents = 0;
    static constexpr ObsDims dims = {OBS_FIX15_RCAP, 100, 0, 0};
    static constexpr ObsShape shape = {2, N_WORDS_C * OBS_CAP_C, N_WORDS_T * OBS_CAP_T_COMPACT, 4, 30};
    static constexpr ObsOptions opt = {OBS_NT, 0, 0, 0, 1, 1, 0, 1, 1, 0, 0, 0, 0, 0, 8 * 1024};
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
template <> struct ObsFixed<18> {
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 0;
    static constexpr int agents = 0;
    static constexpr ObsDims dims = {OBS_FIX15_RCAP, 100, 0, 0};
    static constexpr ObsShape shape = {2, N_WORDS_C * OBS_CAP_C, 0, 0, 0};
    static constexpr ObsOptions opt = {OBS_NT, 0, 0, 0, 1, 0, 0, 1, 1, 0, 1, 1, 1, OBS_ITEMS_LDS_CAP, 4 * 1024};
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
// 16 / 20: maps TALLER than wide (the reference's prediction key col * W + row collides there, tool.h:391-398: compact keys in LDS, L_RKEY;
// never the one-pass kernels) -- Test_3 (35 x 30), Test_6 (60 x 40), Test_9 (120 x 80): two stages, at most 100 agents / 1 280 rail cells
template <> struct ObsFixed<16> {
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 0;
    static constexpr int agents = 0;
    static constexpr ObsDims dims = {1280, 100, 0, 1};
    static constexpr ObsShape shape = {0, N_WORDS_C * OBS_CAP_C, N_WORDS_T * OBS_CAP_T_COMPACT, 4, 30};
    static constexpr ObsOptions opt = {OBS_NT, 24 * 1024, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0};
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
template <> struct ObsFixed<20> {
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 0;
    static constexpr int agents = 0;
    static constexpr ObsDims dims = {1280, 100, 0, 1};
    static constexpr ObsShape shape = {0, N_WORDS_C * OBS_CAP_C, 0, 0, 0};
    static constexpr ObsOptions opt = {OBS_NT, 24 * 1024, 0, 0, 1, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0};
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
template <> struct ObsFixed<19> {   // the flatland_cutils builder alone, the same capacities
    static constexpr int max_nodes = 31, pred_depth = 500, max_depth = 0;
    static constexpr int agents = 0;
    static constexpr ObsDims dims = {OBS_FIXB_RCAP, OBS_FIXB_A, 0, 0};
    static constexpr ObsShape shape = {0, N_WORDS_C * OBS_CAP_C, 0, 0, 0};
    static constexpr ObsOptions opt = {OBS_NT, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0};
    static constexpr ObsLayout L = obs_layout_c(dims, shape, opt);
};
// what obs_pick_config derives from a class's options for ObsArgs: ONE definition for the kernel (which has them as constants) and
// for the host (obs_fits_fixed only takes the class when its own derivation for the batch gives the same values)
template <int FIX> __host__ __device__ constexpr int obs_fixed_bk() { return ObsFixed<FIX>::shape.merged != 0 ? (ObsFixed<FIX>::opt.fb ? 2 : 0) : ObsFixed<FIX>::opt.bk_room; }
template <int FIX> __host__ __device__ constexpr int obs_fixed_wl_occ_div() { return (ObsFixed<FIX>::shape.merged != 0 && ObsFixed<FIX>::dims.A > 32) ? 3 : OBS_WL_OCC_DIV; }
template <int FIX> __host__ __device__ constexpr int obs_fixed_tshift(int A) { return (ObsFixed<FIX>::shape.merged != 0 && ObsFixed<FIX>::dims.A <= 32) ? (A <= 31 ? 2 : OBS_TSHIFT) : OBS_TSHIFT; }
// kernel of a class: MODE 3 (one round) / 4 (rounds of 32 agents) / 2 (two stages), VAR 1 (static tables in LDS) / 2 (work lists in HBM scratch) / 0
// (the flatland_cutils builder alone, shape.tw_t == 0: MODE 6 / 7 / 8 for the one-pass shapes, MODE 0 else)
template <int FIX> __host__ __device__ constexpr int obs_fixed_mode() {
    return (ObsFixed<FIX>::shape.merged == 1 ? 3 : ObsFixed<FIX>::shape.merged == 2 ? 4 : ObsFixed<FIX>::shape.merged == 3 ? 5 : ObsFixed<FIX>::shape.tw_t == 0 ? -3 : 2) + (ObsFixed<FIX>::shape.tw_t == 0 ? 3 : 0);
}
template <int FIX> __host__ __device__ constexpr int obs_fixed_var() { return ObsFixed<FIX>::opt.tab ? 1 : ObsFixed<FIX>::opt.wl_bytes == 0 ? 2 : 0; }
// The batch's own choice `a` (obs_pick_config's preference walk at the batch's sizes) is the class's kind of configuration: the same
// structure, and of everything that is "whatever LDS the carving leaves" -- the next-hop tables, the agents' raw words, the
// capacity of the items' copy, the head of HBM work lists -- at least what the class `b` has (a smaller batch inside the class's
// capacities affords more of those than the class at its full size; the class's own options then fit it a fortiori).
__host__ __device__ constexpr bool obs_same_options(const ObsOptions &a, const ObsOptions &b) {
    return a.nt == b.nt && a.wl_bytes == b.wl_bytes && a.tab == b.tab && (a.nh == b.nh || !b.nh) && a.tmask == b.tmask && a.dual == b.dual &&
           a.items == b.items && a.snext == b.snext && a.partial == b.partial && a.bk_room == b.bk_room &&
           a.own_filter == b.own_filter && a.fb == b.fb && a.raw >= b.raw && (!b.items || a.items_cap >= b.items_cap) && a.wl_head >= b.wl_head && (a.wl_head == 0) == (b.wl_head == 0);
}

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
    int compact_t;     // upstream trees in compact slots (no direction of a cell of the batch has more than two transitions)
    int use_tmask;     // per-key time-bucket masks in LDS
    int tshift;        // width of their time buckets for horizons beyond 64 steps: 1 << tshift steps (bucket = min(t >> tshift, 63))
    int dual_index;    // fused launch: stage 1 also builds the upstream predictor's index (second set of LDS arrays)
    int bk;            // the lists of the cutils index are grouped by time bucket: 1 = large maps (OBS_BK_NB buckets, counted in the node
                       // tables' LDS, offsets in HBM scratch), 2 = small maps (OBS_FB_NB buckets, offsets in LDS: L_BKREL)
    int bk_nb, bk_shift;
    int merged;        // fused launch: ONE pass B per round over the trees of both builders (trees_merged); 1: one round (at most 32 agents), 2: rounds
                       // of 32 agents, 3: rounds of 16 agents on 512 threads (at most 80 KB of LDS: two workgroups a CU)
    int wl_occ_div;    // the occupant work list gets 1 / wl_occ_div of the work-list entries, the conflict list the rest
    int fix;           // FIXED launch class of this launch (ObsFixed<fix>: the kernel's layout is a compile-time constant), 0 = none
    const int16_t *label;  // flatland_cutils get_many(handles) with a strict subset (MODE 0 only; null: every agent): label[i] = position of agent i in
                       // the list or -1.  Only the listed agents' predictions enter the index, under their list POSITION -- the conflict test
                       // then leaves out position `handle` and reads the state of agent `position` (treeobs.cpp:50-62, 393-465, tool.h:428-434)
    int keep_mode;     // host side: FL_OBS_KEEP_TREE_ROWS is on for this handle (the kernels keep row masks; classes 1 and 5 -- the headline
                       // kernels, which carry no code for it -- are not taken)
    int keep_rows;     // upstream tree: the output buffer still holds the previous launch's rows (FL_OBS_KEEP_TREE_ROWS and the same buffer and
                       // depth as that launch): no -inf pre-fill of the slab, only the rows that were real nodes then and are not now
    int out64;         // launches of the flatland_cutils builder alone: adjacency / node_order / edge_order point to int64 buffers and are written as
                       // the policy network takes them (fl_obs_cutils_policy; cutils_rows_orders)
    int cutils_alone;  // host side: a launch of the flatland_cutils builder alone may take the one-pass kernels (MODE 6 / 7 / 8, classes 6 .. 10)
    int wide;          // the batch has several envs per CU (host side: obs_pick_config then prefers workgroups that fit two a CU for small envs)
    int fix_allowed;   // host side: the configuration was chosen without the diagnostic overrides that rule the fixed launch classes out
    const int *h_R;    // host side: the envs' rail cells (FlObsScratch::h_R; null: unknown) -- an exact class's split launch goes before the bin classes
    int split;         // (2: as 1, but the envs that do not fit run the larger bin class 14 / 19 -- every env on a compile-time carving.)  1: fix != 0 and the batch's capacities exceed the class's rail cells: the class serves the envs that fit it (d.R[b] <=
                       // ObsFixed<fix>::dims.Rcap, decided per workgroup), every other env of the launch runs the same kernel's runtime-carving
                       // body with L below; 0: every env of the launch fits the class
    ObsLayout L;       // LDS carving of this launch (host-side obs_layout; the kernel only follows it -- or, fix != 0, has the same
                       // carving compiled in)
};

// kernel launchers, one translation unit per MODE (0 = flatland_cutils outputs, 1 = upstream dense tree, 2 / 3 = both in one launch);
// var: see obs_body (1 = static tables in LDS, 2 = work lists in HBM scratch)
int fl_obs_launch_m0(int var, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_m1(int var, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_m2(int var, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_m3(int var, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);   // both, one pass B, one round
int fl_obs_launch_m4(int var, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);   // both, one pass B per round of 32 agents
int fl_obs_launch_m5(int var, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);   // the same in rounds of 16 agents on 512 threads (two workgroups a CU)
// the FIXED launch classes (P.fix = k: MODE and VAR are the class's, the LDS carving is compiled in)
int fl_obs_launch_f1(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f2(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f3(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f4(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f5(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
// the flatland_cutils builder alone: MODE 6 / 7 / 8 (runtime carving) and its classes 6 .. 10, class 9's split kernel
int fl_obs_launch_m6(int var, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_m7(int var, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_m8(int var, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f6(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f7(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f8(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f9(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f10(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_s9(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
// bin classes (round 6) and the large-map split kernels whose second body is the larger bin class (P.split 2) instead of the runtime carving
int fl_obs_launch_f11(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f12(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f13(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f14(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f15(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f18(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f16(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f20(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f21(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f17(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_f19(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_s4b(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_s9b(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
// the same classes for a batch with larger maps among its envs (P.split): per env the class's body or the runtime-carving one
int fl_obs_launch_s2(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_s3(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
int fl_obs_launch_s4(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s);
