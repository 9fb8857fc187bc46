// fl_obs_ctx.h -- context of one observation build (the env's LDS-resident tables and per-agent snapshot), the static
// topology of a branch walk, and the node tables of the trees.  Device code; included by fl_obs_body.h.
#pragma once
#include <stdio.h>
#include <type_traits>

#include "../../include/flatland_hip.h"
#include "fl_obs_layout.h"

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
    const uint16_t *a_srank;      // rank of the agent's speed among the env's agents (number of slower agents; static, host-computed)
    const int *csr_end;           // LDS [K] end offset of key k's item list (start = csr_end[k-1], 0 for k = 0)
    const uint32_t *items_lds;    // IT_* packed items when they fit LDS ...
    const uint32_t *items_glb;    // ... else in HBM scratch (two members so that each keeps a static address space)
    const uint16_t *bk_rel_lds;   // the same table in LDS (small maps: finer buckets, see obs_body); at most one of the two is set
    int bk_nb, bk_shift;          // number of time buckets of a list and log2 of their width in steps
    const uint16_t *bk_rel;       // HBM [bk_nb][bk_k1] (large maps: the items are laid out BUCKET-major): row b, entry k + 1 = end of key
                                  // k's items inside time bucket b, relative to the bucket's start bk_base[b]; entry 0 of a row = 0
    const int *bk_base;           // LDS [bk_nb]: start of every time bucket's items
    int bk_k1;                    // entries of a row of bk_rel (keys + 1)
    int Tn;                       // number of predicted time entries (0 = no predictor)
    const uint16_t *dm;           // env base [U][SS] distance map (LDS copy when TAB_LDS, else HBM)
    const uint4 *seg;             // env base [S] static branch-walk table (LDS copy when TAB_LDS, else HBM)
    // pass B work lists (LDS): cells with an occupant / cells whose key has a prediction near the queried time
    uint2 *wl_occ, *wl_cf;
    int wl_occ_cap, wl_cf_cap;
    // LDS HEAD of HBM work lists (round 5): entries [0, wl_head_*_n) of a list live in LDS, the others in HBM scratch at the same
    // index -- the entries a round pushes first never leave the CU (cfg4: the 16 KB an LDS copy of the items held without use)
    uint2 *wl_head_occ, *wl_head_cf;
    int wl_head_occ_n, wl_head_cf_n;   // 0: no head
    int *wl_cnt;                  // LDS [3] entries pushed to wl_occ / wl_cf, flag: some key needs the second conflict pass
    const int *long_lists;        // LDS flag: some key's list has more than CF_DIRECT items (else no conflict query needs chunks)
    const unsigned long long *tmask;  // LDS per key: time buckets tb_of(t, tshift) covered by some item; nullptr = none
    int tshift;
    // pass B over the trees of BOTH builders at once (PB = 2): teams below n_cu are flatland_cutils trees and use the members
    // above, the others are upstream trees and use the upstream predictor's index:
    int n_cu;                     // agents of a round of trees (32, or 16 on 512 threads): the teams below it are flatland_cutils trees
    int round_base;               // first agent of the round of trees being built
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

// The observation tensors are written once and never read back by the kernel: their stores carry the non-temporal hint (round 5;
// -DOBS_NO_NT_STORES: plain stores), so that a gigabyte of rows streaming through the L2 does not push out the lines the launch keeps
// coming back to -- the shared static tables, the prediction items, the work lists' tails, the next step's state.  Same box: cfg3
// 120.0 -> 132.1 M (k_obs 0.671 -> 0.612 ms, and k_step 18.8 -> 16.9 us: its state is still in the L2), cfg5 79.0 -> 81.4 M, cfg4
// 128.0 -> 129.4 M, cfg2 110.9 -> 112.2 M.
// OBS_NT_LEVEL: 0 plain stores; 1 the -inf pre-fill of the upstream slabs only (whole lines, lane after lane); 2 also the 16-byte
// pieces of the rows (forest, tree); 3 also the 4-byte elements (attribute rows, adjacency, orders, masks, properties).
// Measured per level on one box (profiles/r05_nt_store_levels.txt; M agent-steps/s | WRITE_SIZE | FETCH_SIZE MB per launch):
//   cfg3: 120.0 | 907 | 17.3 -> 127.8 | 904 | 15.8 -> 132.1 | 991 | 11.3 -> 132.1 | 1018 | 10.7;   cfg5: 79.0 | 1558 | 401 -> 79.4 -> 81.4 | 1642 | 358 -> 80.8 | 1673 | 346
//   cfg4: 128.0 | 237 | 33.9 -> 129.1 -> 129.4 | 276 | 25.0 -> 129.6 | 287 | 22.2;                   cfg2: 110.9 | 23.8 -> 110.9 -> 112.2 | 27.3 -> 112.7 | 28.2
// A non-temporal store of part of a line leaves the L2 on its own (counted as a 32-byte write) instead of waiting for the rest of the
// line: level 2 is where the time is, level 3 buys 0.5 % at cfg2 for 3 - 4 % more bytes written -- level 2 is the default.
#ifndef OBS_NT_LEVEL
#define OBS_NT_LEVEL 2
#endif
typedef float obs_f4_t __attribute__((ext_vector_type(4)));
typedef double obs_d2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fill_store_d2(double *p, double a, double b) {   // the pre-fill: consecutive lanes, whole lines
#if OBS_NT_LEVEL >= 1
    const obs_d2_t v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<obs_d2_t *>(p));
#else
    *reinterpret_cast<double2 *>(p) = make_double2(a, b);
#endif
}
__device__ __forceinline__ void out_store_f4(float *p, float a, float b, float c, float d) {
#if OBS_NT_LEVEL >= 2
    const obs_f4_t v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<obs_f4_t *>(p));
#else
    *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d);
#endif
}
__device__ __forceinline__ void out_store_d2(double *p, double a, double b) {
#if OBS_NT_LEVEL >= 2
    const obs_d2_t v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<obs_d2_t *>(p));
#else
    *reinterpret_cast<double2 *>(p) = make_double2(a, b);
#endif
}

template <typename T>
__device__ __forceinline__ void out_store(T *p, T v) {   // one element of an output tensor
#if OBS_NT_LEVEL >= 3
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// Time bucket of the per-key masks (64 bits a key): equal buckets of 1 << tshift steps and one catch-all for everything later.
// (Round 4, measured and dropped: a piecewise map -- 2-step buckets to t = 64, 8-step to 192, 16-step to 448 -- instead of the
// catch-all: 1 to 3 % slower on all four workloads, the longer bucket arithmetic of every classified cell costs more than the
// late queries it filters save.)
__device__ __forceinline__ int tb_of(int t, int ts) { return min(t >> ts, 63); }

__device__ __forceinline__ uint32_t cw_bits(const ObsCtx &X, int r) { return X.cellw[r] & 0xFFFFu; }
// occupied-cell table index of the cell, 0xFFFF = nobody on it and nobody waiting to depart from it
__device__ __forceinline__ uint32_t cw_slot(const ObsCtx &X, int r) { return X.cellw[r] >> 16; }
__device__ __forceinline__ uint32_t cw_load(const ObsCtx &X, int r) { return X.cellw[r]; }
__device__ __forceinline__ int key_of(const ObsCtx &X, int r) { return X.rkey ? (int)X.rkey[r] : r; }
// Pass B serves one builder (PB 0 = upstream, 1 = flatland_cutils) or both in one pass (PB 2): cu says which builder's
// rules apply to a team.  PB 3 = the one-pass kernels' machinery (rounds of trees_merged, own-path filter, the fast classify loop)
// with the flatland_cutils trees ALONE -- the launch the reference's solution makes (solution/eval_env.py:15-17 builds TreeCutils only).
template <int PB>
__device__ __forceinline__ bool pb_cu(const ObsCtx &X, int team) { return PB == 3 ? true : PB == 2 ? team < X.n_cu : PB == 1; }
// the agent of a team: from the team table, or -- both builders in one pass -- from the numbering of trees_merged (no memory access)
template <int PB>
__device__ __forceinline__ int pb_handle(const ObsCtx &X, const int *team_meta, int team) {
    return PB == 3 ? X.round_base + team : PB == 2 ? X.round_base + (team < X.n_cu ? team : team - X.n_cu) : team_meta[128 + team];
}
// predicted time at which the walking agent reaches a cell tot steps away (treeobs.cpp:378 / observations.py:329)
template <int PB>
__device__ __forceinline__ int pt_of(const ObsCtx &X, bool cu, int handle, int tot) {
    if (PB == 2 && !cu) return (int)((double)tot * X.a_tq2[handle]);
    return cu ? (int)((float)tot * (float)X.a_tq[handle]) : (int)((double)tot * X.a_tq[handle]);
}
// Items of rail cell r's key a conflict query at predicted time pt has to look at, as a VIRTUAL range [0, n): the first n1 are
// the items lo .. lo + n1 - 1 of the index, the rest lo2 .. (plain and duplicated-bucket lists: n1 = n).  Lists grouped by time
// bucket with LDS-resident offsets (bk_rel_lds) keep the items that stay until the end of the horizon -- the last waypoint of
// a path, usually a target cell -- in a bucket of their own behind the time buckets: the second piece of every query.
// TWO = false (kernels that never build such lists): one piece, plain indexing.
struct ListRange { int lo, n1, lo2, n; };
template <bool TWO>
__device__ __forceinline__ int list_index(const ListRange &R, int v) { return (TWO && v >= R.n1) ? R.lo2 + (v - R.n1) : R.lo + v; }
template <int PB>
__device__ __forceinline__ ListRange list_range(const ObsCtx &X, bool cu, int r, int pt) {
    const int key = key_of(X, r);
    ListRange R;
    R.lo2 = 0;
    if (PB == 2 && !cu) {
        R.lo = key > 0 ? X.u_csr_end[key - 1] : 0;
        R.n1 = R.n = X.u_csr_end[key] - R.lo;
        return R;
    }
    if (X.bk_rel_lds) {  // (two call sites: each table keeps its address space)
        const int base = key > 0 ? X.csr_end[key - 1] : 0;
        const int b1 = min(max(pt - 1, 0) >> X.bk_shift, X.bk_nb - 1), b2 = min(min(pt + 1, X.Tn - 1) >> X.bk_shift, X.bk_nb - 1);
        // per key: bk_nb time buckets, the to-the-end bucket, and a mask of the time buckets in which such an item starts
        const uint16_t *rel = X.bk_rel_lds + key * (X.bk_nb + 2);
        const int o1 = b1 > 0 ? (int)rel[b1 - 1] : 0;
        R.lo = base + o1;
        R.n1 = R.n = (int)rel[b2] - o1;
        if ((uint32_t)rel[X.bk_nb + 1] & ((2u << (b2 >> OBS_FB_MSHIFT)) - 1u)) {  // somebody stays here from a time the query can see
            const int oe = (int)rel[X.bk_nb - 1];
            R.lo2 = base + oe;
            R.n += (int)rel[X.bk_nb] - oe;
        }
    } else if (X.bk_rel) {
        // bucket-major items: the key's items of the one or two time buckets the three queried times fall in -- two pieces
        const int b1 = min(max(pt - 1, 0) >> X.bk_shift, X.bk_nb - 1), b2 = min(min(pt + 1, X.Tn - 1) >> X.bk_shift, X.bk_nb - 1);
        // (start, end) of a key are neighbouring 16-bit words: ONE 2-byte-aligned dword load each (HBM scratch; see conflict_flags)
        struct __attribute__((packed, aligned(2))) Pair16 { uint32_t v; };
        const uint32_t w1 = reinterpret_cast<const Pair16 *>(X.bk_rel + b1 * X.bk_k1 + key)->v;
        const int s1 = (int)(w1 & 0xFFFFu), e1 = (int)(w1 >> 16);
        R.lo = X.bk_base[b1] + s1;
        R.n1 = R.n = e1 - s1;
        if (b2 != b1) {
            const uint32_t w2 = reinterpret_cast<const Pair16 *>(X.bk_rel + b2 * X.bk_k1 + key)->v;
            const int s2 = (int)(w2 & 0xFFFFu), e2 = (int)(w2 >> 16);
            R.lo2 = X.bk_base[b2] + s2;
            R.n += e2 - s2;
        }
    } else {
        const int base = key > 0 ? X.csr_end[key - 1] : 0;
        R.lo = base;
        R.n1 = R.n = X.csr_end[key] - base;
    }
    return R;
}
// successor of a state with exactly one transition (chain interior): one LDS load when the table is resident
__device__ __forceinline__ uint32_t chain_next(const ObsCtx &X, uint32_t s, uint32_t bits16) {
    if (X.snext) return X.snext[s];
    const uint32_t nd = first_dir(nibble(bits16, s & 3u));
    return ((uint32_t)X.nbr[(s & ~3u) | nd] << 2) | nd;
}
// state reached by leaving rail cell r through its transition m; -2 when that transition leaves the rail (see FL_R_PHANTOM)
__device__ __forceinline__ int state_towards(const ObsCtx &X, int r, uint32_t m) {
    const uint32_t nr = X.nbr[r * 4 + (int)m];
    return nr == FL_R_NONE ? -2 : (int)((nr << 2) | m);
}

// One node of a tree = one branch walk (_explore_branch: treeobs.cpp:258-610 / observations.py:256-494).
// Where the walk ends, how long it is and its first "unusable switch" are static per start state (segment table,
// fl_dmap.hip); the stop at the agent's own target follows from the distance map: along a chain of single-transition
// cells the distance drops by one per step, so the target is on the chain iff dm[start] <= chain length.
struct NodeDesc {
    int start;      // start state cell << 2 | dir, -1 = null cell, -2 = an empty cell (the parent's transition leaves the rail)
    int tot0;       // tot_dist at the first visited cell
    int nvis;       // number of visited cells (feature block executions)
    int end;        // end state (direction unknown / irrelevant when the walk stops at the target)
    uint32_t flags; // bit 0 target stop, 1 switch, 2 dead end, 3 terminal (zero transition or cycle), 4 zero transition, 5 off the rail
    int unus;       // tot_dist of the first unusable switch or -1
    uint32_t kids01, kids23;  // start states of the end state's children (u16 each, FL_R_NONE = null), valid for switch / dead end
};

// dm_t = the distance slab of the agent's target, target = its rail index (per-agent constants, hoisted by the callers)
__device__ __forceinline__ NodeDesc node_topology(const ObsCtx &X, const uint16_t *dm_t, int target, int start, int tot0) {
    NodeDesc n;
    n.start = start;
    n.tot0 = tot0;
    if (start < 0) {  // -2: a node without cells on the rail (see FL_R_PHANTOM)
        n.start = 0; n.nvis = 0; n.end = 0; n.flags = ND_TERMINAL | ND_ZERO | ND_PHANTOM; n.unus = -1;
        n.kids01 = n.kids23 = 0xFFFFFFFFu;
        return n;
    }
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
        n.flags = k == SEG_SWITCH ? ND_SWITCH : k == SEG_DEAD_END ? ND_DEAD_END : k == SEG_ZERO ? (ND_TERMINAL | ND_ZERO | (SEG_PHANTOM(e) ? ND_PHANTOM : 0u)) : ND_TERMINAL;
        n.unus = unus != 0xFFFF ? tot0 + unus : -1;
    }
    return n;
}

// advance k cells along a chain of single-transition cells (no features)
__device__ __forceinline__ uint32_t skip_cells(const ObsCtx &X, uint32_t s, int k) {
    for (int v = 0; v < k; v++) s = chain_next(X, s, X.snext ? 0u : cw_bits(X, (int)(s >> 2)));
    return s;
}

// ---- node tables (fl_obs_layout.h): word w of node k of a table with `cap` slots
__device__ __forceinline__ int &nt_w(int *scr, int cap, int w, int k) { return scr[w * cap + k]; }
__device__ __forceinline__ int nt_r(const int *scr, int cap, int w, int k) { return scr[w * cap + k]; }
__device__ __forceinline__ int nt_start(uint32_t se) { const uint32_t s = se & 0xFFFFu; return s == N_NONE ? -1 : (int)s; }
__device__ __forceinline__ int nt_end(uint32_t se) { return (int)(se >> 16); }
__device__ __forceinline__ int nt_tot(uint32_t tv) { return (int)(tv & 0xFFFFu); }
__device__ __forceinline__ int nt_vis(uint32_t tv) { return (int)(tv >> 16); }
__device__ __forceinline__ int nt_unus(uint32_t uf) { const uint32_t u = uf & 0xFFFFu; return u == 0xFFFFu ? -1 : (int)u; }
__device__ __forceinline__ uint32_t nt_flags(uint32_t uf) { return (uf >> 16) & 63u; }
__device__ __forceinline__ int nt_row(uint32_t uf) { return (int)((uf >> 22) & 1023u); }   // (341 rows at depth 4)
// descriptor of a node into its slot; err: latched when a distance does not fit 16 bits (no Flatland map comes close)
__device__ __forceinline__ void nt_store_desc(int *scr, int cap, int k, const NodeDesc &nd, int row, int *err) {
    if ((uint32_t)(nd.tot0 + nd.nvis) > 0xFFFEu && err) atomicCAS(err, 0, FL_ERR_CAPACITY);
    nt_w(scr, cap, N_SE, k) = (int)(((uint32_t)nd.start & 0xFFFFu) | ((uint32_t)nd.end << 16));
    nt_w(scr, cap, N_TV, k) = (int)(((uint32_t)nd.tot0 & 0xFFFFu) | ((uint32_t)nd.nvis << 16));
    nt_w(scr, cap, N_UF, k) = (int)(((uint32_t)(nd.unus < 0 ? 0xFFFF : nd.unus) & 0xFFFFu) | (nd.flags << 16) | ((uint32_t)row << 22));
}
__device__ __forceinline__ void nt_clear_desc(int *scr, int cap, int k) {
    nt_w(scr, cap, N_SE, k) = (int)N_NONE;
    nt_w(scr, cap, N_TV, k) = 0;
}
