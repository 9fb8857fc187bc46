// fl_obs_body.h -- one observation build for the workgroup's env (obs_body: staging, per-agent phase, predicted paths and the
// per-key prediction index, then the trees) and the kernel template.  Device code; the three fl_obs_m<MODE>.hip units
// instantiate it.  DESIGN.md section 4 describes the phases; tools/obs_phase_clocks.py measures them.
//
// Replaces (paths relative to /root/reference):
//   flatland_cutils/src/loader.cpp:221-327      AgentsLoader::update (snapshot, dist_target, road_type, valid actions)
//   flatland_cutils/src/deadlock_checker.cpp    DeadlockChecker (restated as a least fixpoint, see phase 1)
//   flatland_cutils/src/predictions.cpp:78-235  shortest-path predictor (greedy strict descent on the distance map)
//   flatland_cutils/src/treeobs.cpp:30-610      get_many / get / _explore_branch / scale_node
//   flatland_cutils/src/tool.h:468-524          calculate_evaluation_orders
//   flatland_cutils/src/feature_parser.cpp:3-98 AgentAttrParser::get_features
//   flatland-rl/flatland/envs/observations.py:60-494 + predictions.py:97-180   upstream TreeObsForRailEnv
//
// Everything is indexed by RAIL CELLS (rail index r, rail state s = r * 4 + orientation; fl_internal.h), not by grid cells:
// the per-cell words, the neighbour / successor tables, the prediction keys and their time masks of a 150x150 map
// (2680 rail cells) fit LDS like those of a 30x30 one.
//
// Layout of one launch (gfx950): one workgroup (up to 16 wavefronts) per env.  The env's rail words, neighbour tables and
// an occupied-cell table are staged in LDS once.  Then, concurrently: eight lanes per agent walk its predicted path (static
// next-hop / eight-hop tables), one wavefront does the per-agent part (deadlock fixpoint, valid actions, 83-float attribute
// row) and the other wavefronts derive the topology of the trees from the static segment table (pass A).  A per-key index
// of prediction items (+ per-key time-bucket masks) is built in LDS.  Pass B splits the visited cells of all trees evenly
// over all lanes, classifies them, and processes the few cells that need work from work lists on packed wavefronts; rows
// are written straight to HBM.
#pragma once
#include "fl_obs_trees.h"

// One observation build for the workgroup's env.  STAGE 0: stand-alone; the fused launch (both builders) runs STAGE 1
// (cutils; also prepares what the second stage needs) and then STAGE 2 (upstream tree), which reuses the LDS-resident
// rail words / occupancy table / static tables and the predicted paths of stage 1: the upstream predictor's path
// is a prefix of the cutils one (same greedy descent, it only stops at the target and after fewer steps).
// VAR 1 (small maps): the env's distance map, segment, next-hop and eight-hop tables are staged in LDS.  VAR 2 (large maps):
// the pass B work lists live in HBM scratch, which leaves the LDS to the time masks and lifts the cap on their entries.
// MERGED (its own kernel, MODE 3): stage 1 of the fused launch builds the trees of BOTH builders, one pass B per round
// (trees_merged); there is no stage 2.  The launcher sizes the LDS copy of the second index for the exact bound on its items, so
// this mode never has to fall back.
// workgroup k builds env k, or -- more envs than CUs -- the env the host's ordering kernel put k-th (fl_obs_env_order)
// (readfirstlane: the compiler cannot prove the load unclobbered, makes it a vector load, and every address derived from the env
// index would be vector arithmetic -- 29 spilled vector registers in the rounds kernel)
__device__ __forceinline__ int obs_env_of_workgroup(const FlObsScratch &S) { return __builtin_amdgcn_readfirstlane(S.order ? S.order[blockIdx.x] : (int)blockIdx.x); }

// UP = false (one-pass kernels only, MODE 6 / 7 / 8): the flatland_cutils builder ALONE on the one-pass machinery -- no second index, no
// upstream tables / jobs / rows; what the reference's solution launches (solution/eval_env.py:15-17: TreeCutils(31, 500) and nothing else).
template <bool CUTILS, int VAR, int STAGE, int MERGED = 0, int FIX = 0, bool UP = true>
__device__ __forceinline__ void obs_body(const FlDev &d, const FlObsScratch &S, const ObsArgs &P_launch) {
    // A fixed launch class also fixes the builders' parameters (BASELINE's: 31 nodes, predictor depths 500 / 30): the launcher only
    // takes the class for exactly these, the kernel has them as constants (a local copy the compiler takes apart; no memory).
    using FixT = ObsFixed<FIX != 0 ? FIX : 1>;
    ObsArgs P_local = P_launch;
    if (FIX != 0) {
        P_local.max_nodes = FixT::max_nodes; P_local.pred_depth = FixT::pred_depth; P_local.tree_pred = FixT::shape.tree_pred;
        P_local.tshift = obs_fixed_tshift<FIX != 0 ? FIX : 1>(d.A);
        if (FixT::max_depth != 0) {   // (0: the call's depth -- a class for a bin of shapes, or the flatland_cutils builder alone)
            P_local.max_depth = FixT::max_depth;
            P_local.n_tree_nodes = FixT::max_depth == 2 ? 21 : 85;   // (4^(depth + 1) - 1) / 3
        }
    }
    const ObsArgs &P = P_local;
    constexpr bool TAB_LDS = (VAR & 1) != 0, WL_HBM = (VAR & 2) != 0;
    // (the one-round kernel of small envs is never ordered: its env index stays the workgroup id the hardware hands over)
    // (every fixed launch class and every one-pass kernel runs on a known number of threads -- obs_pick_config launches MODE 3 / 4 on
    // OBS_NT, MODE 5 on 512: the strides of the workgroup-wide loops are constants there)
    const int b = MERGED == 1 ? (int)blockIdx.x : obs_env_of_workgroup(S), tid = threadIdx.x,
              nt = MERGED == 3 ? 512 : (FIX != 0 || MERGED == 1 || MERGED == 2) ? OBS_NT : (int)blockDim.x;
    const int A = (FIX != 0 && FixT::agents != 0) ? FixT::agents : d.A, R = d.R[b], NS = R * 4, K = d.K[b], U = d.U[b];
    const int Rcap = d.Rcap, Scap = Rcap * 4;
    const int lane = tid & 63, wave = tid >> 6;

    extern __shared__ __align__(16) unsigned char lds[];
    // FIX != 0: the carving is ObsFixed<FIX>::L, a compile-time constant (every base below folds into an immediate)
#define L_OFF(which) (FIX != 0 ? FixT::L.off[which] : P.L.off[which])
#define L_FIELD(f) (FIX != 0 ? FixT::L.f : P.L.f)
    // ... and so is what obs_pick_config derives from the class's options for ObsArgs (the launcher sets the same values in P)
    // (the same folding for the runtime-carving one-pass kernels, whose launches always have masks / second index / compact tables,
    // measured in the static code only: two spilled VECTOR registers in k_obs<4,2,0> -- not done)
    const int p_bk = FIX != 0 ? obs_fixed_bk<FIX != 0 ? FIX : 1>() : P.bk;
    const int p_bk_nb = FIX != 0 ? (MERGED != 0 ? OBS_FB_NB : OBS_BK_NB) : P.bk_nb, p_bk_shift = FIX != 0 ? (MERGED != 0 ? OBS_FB_SHIFT : OBS_BK_SHIFT) : P.bk_shift;
    const bool p_use_tmask = FIX != 0 ? FixT::opt.tmask != 0 : P.use_tmask != 0, p_dual_index = UP && (FIX != 0 ? FixT::opt.dual != 0 : P.dual_index != 0);
    const bool p_compact_t = FIX != 0 ? true : P.compact_t != 0;   // (every class's shape has the compact upstream tables)
    const int p_wl_occ_div = FIX != 0 ? obs_fixed_wl_occ_div<FIX != 0 ? FIX : 1>() : P.wl_occ_div;
#define LDS_AT(T, which) reinterpret_cast<T *>(lds + L_OFF(which))
#define LDS_OPT(T, which) (L_OFF(which) == L_ABSENT ? (T *)nullptr : reinterpret_cast<T *>(lds + L_OFF(which)))
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
    uint16_t *a_srank = LDS_AT(uint16_t, L_A_SRANK);
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
    int *partial = (WL_HBM || L_OFF(L_PARTIAL) != L_ABSENT) ? LDS_AT(int, L_PARTIAL) : reinterpret_cast<int *>(wl_lds);
    const int wl_entries = WL_HBM ? OBS_WL_HBM_ENTRIES : L_FIELD(wl_bytes) / 8;
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
    // the env whose slabs hold this env's static tables (FlDev::tab: envs of the same map and targets share one set)
    const int tb = __builtin_amdgcn_readfirstlane(d.tab[b]);
    const uint4 *gseg = d.seg + (size_t)tb * Scap;
    const uint16_t *gdm = d.dm + (size_t)tb * d.Ucap * Scap;
    const uint16_t *ghop8 = d.hop8 + (size_t)tb * d.Ucap * Scap;
    const uint16_t *gnh = d.nh + (size_t)tb * d.Ucap * Rcap;

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
    OBS_STAMP(0);

    // get_many(handles) with a strict subset: the stand-alone launches (flatland_cutils: treeobs.cpp:50-62; upstream: observations.py:72-83)
    const int16_t *lab = (STAGE == 0 && MERGED == 0 && FIX == 0) ? P.label : nullptr;
    const int my_pred_depth = CUTILS ? P.pred_depth : P.tree_pred;
    const bool any_pred = STAGE == 0 ? my_pred_depth >= 0 : true;
    const bool nh_in_lds = nh_lds != nullptr && any_pred;
    // ---- phase 0: stage the rail words and the static tables, clear the per-cell maps, per-agent snapshot into LDS
    bool prefilled = false, bg_prefill = false;
    if (STAGE != 2) {
        // per-agent snapshot first, on the LAST lanes: its two dependent HBM reads (state, then the rail index of the position)
        // overlap with the staging of the tables by everybody else
        const uint16_t *gridx = d.ridx + (size_t)tb * d.H * d.W;
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
            a_srank[i] = d.srank[g];
            a_tpc[i] = CUTILS ? (uint16_t)(int)(1.0f / (float)speed) : (uint16_t)(int)(1.0 / speed);
            a_tq[i] = CUTILS ? (double)(float)(1.0 / (double)(float)speed) : 1.0 / speed;
            if (CUTILS && STAGE == 1 && p_dual_index) { a_tpc2[i] = (uint16_t)(int)(1.0 / speed); a_tq2[i] = 1.0 / speed; }  // the upstream predictor's (predictions.py:139)
        }
        {
            const uint16_t *grg = d.rgrid + (size_t)tb * Rcap;
            for (int r = tid; r < R; r += nt) cellw[r] = (uint32_t)grg[r] | 0xFFFF0000u;
            // u16 tables: two entries per load (every base is 4-byte aligned: Scap is a multiple of 4, Rcap * U pairs up below)
            const uint32_t *g2 = reinterpret_cast<const uint32_t *>(d.nbr + (size_t)tb * Scap);
            uint32_t *l2 = reinterpret_cast<uint32_t *>(nbr);
            for (int c = tid; c < NS / 2; c += nt) l2[c] = g2[c];
            if (snext) {
                g2 = reinterpret_cast<const uint32_t *>(d.snext + (size_t)tb * Scap);
                l2 = reinterpret_cast<uint32_t *>(snext);
                for (int c = tid; c < NS / 2; c += nt) l2[c] = g2[c];
            }
            if (rkey) {
                const uint16_t *gk = d.rkey + (size_t)tb * Rcap;
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
                const uint8_t *grt = d.rtype + (size_t)tb * Rcap;
                for (int r = tid; r < R; r += nt) rtype_lds[r] = grt[r];
            }
        }
        // (with a job queue in phase 2 -- cutils builder in the launch, a predictor, a wavefront left beside the walkers -- the rows
        // are pre-filled there by whoever is free, beside the path walk, instead of here by everybody)
        const int nw_walk0 = min((nt >> 6) - ((nt >> 6) > 4 ? 1 : 0), max(2, (A + 7) / 8));
        // (a slice of the pre-fill in every round of the first stage's trees instead measured slower at cfg5, 1.73 against 1.65 ms:
        // the next global load of a wavefront waits for its stores)
        // (FL_OBS_KEEP_TREE_ROWS: the buffer still holds the previous launch's rows -- no pre-fill, upstream_rows sets the stale ones)
        bg_prefill = UP && CUTILS && STAGE == 1 && P.tree_out != nullptr && P.pred_depth >= 0 && nw_walk0 < (nt >> 6) && !P.keep_rows;
        if (UP && (!CUTILS || STAGE == 1) && P.tree_out && !bg_prefill && !P.keep_rows) {
            // the upstream trees of the env: every row that is not a real node is -inf (observations.py:247, 489); the builders
            // only write the real rows later.  Coalesced 16-byte stores beside the staging.  The wait below (before the barrier
            // that ends this phase) lets them reach the L2 ahead of any later store of this workgroup to the same rows -- same CU,
            // same L2 -- without the L2 write-back an agent-scope release fence costs on this multi-XCD part (70 us).
            double2 *o2 = reinterpret_cast<double2 *>(P.tree_out + (size_t)b * A * P.n_tree_nodes * 12);
            const int n2 = A * P.n_tree_nodes * 6;
            const double2 ninf = make_double2(-INFINITY, -INFINITY);
            for (int k = tid; k < n2; k += nt) fill_store_d2(reinterpret_cast<double *>(o2 + k), -INFINITY, -INFINITY);
            prefilled = true;
        }
        for (int i = tid; i < A; i += nt) { slot_agent[i] = -1; slot_ready[i] = 0; }
        for (int c = tid; c < (R + 31) / 32; c += nt) cell_target[c] = 0;
        if (tid < 62) misc[tid] = 0;   // (62, 63: the launch's env and start clock, see k_obs)
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
            if (UP && (!CUTILS || STAGE == 1)) atomicOr(&cell_target[a_target[i] >> 5], 1u << (a_target[i] & 31));
        }
    } else {
        // second stage: only the predictor's times-per-cell differ (int(np.reciprocal(speed)), predictions.py:139)
        for (int i = tid; i < A; i += nt) { a_tpc[i] = (uint16_t)(int)(1.0 / a_speed[i]); a_tq[i] = 1.0 / a_speed[i]; }
    }
    if (prefilled) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    X.a_tpc = a_tpc; X.a_tq = a_tq; X.a_tslot = a_tslot; X.a_target = a_target; X.a_srank = a_srank;
    uint32_t *csr_items = S.cell_items + (size_t)b * S.items_cap;
    X.csr_end = csr; X.items_lds = nullptr; X.items_glb = csr_items; X.bk_rel = nullptr; X.bk_rel_lds = nullptr; X.bk_base = nullptr; X.bk_k1 = 0;
    X.bk_nb = p_bk_nb; X.bk_shift = p_bk_shift;
    X.Tn = my_pred_depth >= 0 ? my_pred_depth + 1 : 0;
    // without the masks nearly every cell on somebody's route would be a conflict candidate: those are handled in place
    X.tmask = (p_use_tmask && X.Tn > 0) ? tmask : nullptr;
    X.wl_occ = WL_HBM ? S.wl + (size_t)b * OBS_WL_HBM_ENTRIES : reinterpret_cast<uint2 *>(wl_lds);
    X.wl_occ_cap = X.tmask ? wl_entries / p_wl_occ_div : wl_entries;  // a share of the entries
    X.wl_cf = X.wl_occ + X.wl_occ_cap; X.wl_cf_cap = wl_entries - X.wl_occ_cap;
    {   // LDS head of HBM lists (one-pass kernels only; the launcher gives it what the carving leaves): split like the lists
        const int head_entries = (WL_HBM && MERGED != 0 && L_OFF(L_WL) != L_ABSENT) ? L_FIELD(wl_head) / 8 : 0;
        uint2 *head = head_entries > 0 ? LDS_AT(uint2, L_WL) : nullptr;
        X.wl_head_occ = head; X.wl_head_occ_n = head_entries / p_wl_occ_div;
        X.wl_head_cf = head + X.wl_head_occ_n; X.wl_head_cf_n = head_entries - X.wl_head_occ_n;
    }
    X.wl_cnt = misc + 8;
    X.long_lists = misc + 11;
    X.tshift = X.Tn <= 64 ? 0 : P.tshift;  // bucket = tb_of(t, tshift)
    // one pass B over the trees of both builders (stage 1 of the fused launch): the upstream builder's side of the context
    constexpr bool merged = MERGED != 0;   // 1: at most 32 agents (one round), 2: rounds of 32 agents, 3: rounds of 16 agents on 512 threads
    constexpr int ROUND = MERGED == 3 ? 16 : 32;
    X.n_cu = ROUND; X.round_base = 0;
    X.u_csr_end = csr2; X.u_items = items2; X.u_tmask = tmaskb; X.a_tq2 = a_tq2;
    X.tmask_m2 = merged ? tmask_m2 : nullptr; X.u_tmask_m2 = tmaskb_m2;
    X.path = S.path + (size_t)b * A * OBS_PRED_CAP; X.pred_cap = OBS_PRED_CAP;
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
                d.pk[g] |= PK_DEADLOCK_BIT;
            }
            out_store(&P.attr[(size_t)g * FL_CUTILS_ATTR + 41], (float)a_dead[i]);
            if (P.props) out_store(&P.props[(size_t)g * 3 + 1], (double)a_dead[i]);
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
        r.road_type = pos < 0 ? 0 : rtype_lds ? (int)rtype_lds[pos] : (int)d.rtype[(size_t)tb * Rcap + pos];
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
        if (gl < 5) out_store(&P.valid[(size_t)g * 5 + gl], (uint8_t)((va >> gl) & 1));
        if (P.props && gl == 5) {
            out_store(&P.props[(size_t)g * 3 + 0], (double)dist_target);
            out_store(&P.props[(size_t)g * 3 + 2], (double)(state == ST_READY));
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
        out_store(&o[gl], (float)((m_lo >> gl) & 1ull));
        if (gl + 32 != 41) out_store(&o[gl + 32], (float)((m_lo >> (gl + 32)) & 1ull));  // element 41: the deadlock flag (phase 1a)
        if (gl + 64 < FL_CUTILS_ATTR) {
            float v = (float)((m_hi >> gl) & 1u);
#pragma unroll
            for (int k = 0; k < 13; k++) v = gl == 6 + k ? fv[k] : v;
            out_store(&o[gl + 64], v);
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
    // One pass B for both builders: the rest of phase 1 waits for the work-list step of the first round of trees, where the
    // wavefronts finish at very different times (late_jobs) -- this phase then ends with the last pass A.
    const int n_p1_all = (do_p1 && p1_beside_walk && X.Tn > 0) ? (A + 1) / 2 : 0;
    const bool late_p1 = merged && n_p1_all > 0;
    const int n_up_jobs = (merged && UP) ? (min(A, ROUND) + 3) / 4 : 0, n_p1_jobs = late_p1 ? 0 : n_p1_all;
    // ... and, last, the -inf pre-fill of the env's upstream rows (see phase 0) in chunks of 16 KB: pure stores that drain beside
    // the latency-bound rest of the phase (no builder writes a row before the trees phase)
    constexpr int PF_CHUNK = 64 * 16;  // double2 per job
    const int pf_n2 = bg_prefill ? A * P.n_tree_nodes * 6 : 0, n_pf_jobs = (pf_n2 + PF_CHUNK - 1) / PF_CHUNK;
    // Order: pass A of the upstream trees first (the trees wait for them), then the rest of phase 1, the pre-fill last.  (The
    // pre-fill first -- streamed by the one free wavefront while hundreds of paths are walked -- doubled the walk at cfg5,
    // 166 -> 323 us: the walk is a chain of L2 / HBM round trips and the stores fill the same queues.)
    auto drain_jobs = [&]() __attribute__((always_inline)) {
        bool stored = false;
        while (n_up_jobs + n_p1_jobs + n_pf_jobs > 0) {
            int j = 0;
            if (lane == 0) j = atomicAdd(&misc[6], 1);
            j = __builtin_amdgcn_readfirstlane(j);
            if (j >= n_up_jobs + n_p1_jobs + n_pf_jobs) break;
            if (j < n_up_jobs) {
                const int u = 4 * j + (lane >> 4);
                upstream_pass_a<16, OBS_CAP_T_COMPACT, true, 32>(X, P, b, u, u < A, lane & 15, merged_table_t<ROUND>(wave_scr, min(u, ROUND - 1)), &d.err[b]);
            } else if (j < n_up_jobs + n_p1_jobs) {
                const int i = 2 * (j - n_up_jobs) + (lane >> 5);
                if (i < A) phase1b(i, lane & 31, phase1b_load(i));
            } else {
                double2 *o2 = reinterpret_cast<double2 *>(P.tree_out + (size_t)b * A * P.n_tree_nodes * 12);
                const double2 ninf = make_double2(-INFINITY, -INFINITY);
                const int k0 = (j - n_up_jobs - n_p1_jobs) * PF_CHUNK + lane;
#pragma unroll
                for (int q = 0; q < 16; q++)
                    if (k0 + q * 64 < pf_n2) fill_store_d2(reinterpret_cast<double *>(o2 + k0 + q * 64), -INFINITY, -INFINITY);
                stored = true;
            }
        }
        if (stored) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // in the L2 before the barrier that ends the phase
    };
    auto late_jobs = [&]() __attribute__((always_inline)) {
        while (late_p1) {
            int j = 0;
            if (lane == 0) j = atomicAdd(&misc[7], 1);
            j = __builtin_amdgcn_readfirstlane(j);
            if (j >= n_p1_all) break;
            const int i = 2 * j + (lane >> 5);
            if (i < A) phase1b(i, lane & 31, phase1b_load(i));
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
        const bool dual = UP && CUTILS && STAGE == 1 && p_dual_index && P.tree_pred >= 0;
        const bool reuse = STAGE == 2 && p_dual_index && misc[4] != 0;
        const int Tn2 = P.tree_pred + 1, tshift2 = Tn2 <= 64 ? 0 : P.tshift;  // the same bucket width stage 2 queries with
        // large maps: bucketed lists (OBS_BK_NB).  Their per-(key, bucket) counters -- u16, two per word -- live in the node
        // tables' LDS while the index is built (so no tree work is hoisted beside the walk), the offsets go to HBM afterwards
        // (P.bk 1).  Small maps (P.bk 2): finer buckets, counters and offsets in an LDS array of their own.
        const bool bk = CUTILS && STAGE != 2 && p_bk != 0 && X.Tn > 64 && X.tmask != nullptr;
        const bool bk_lds = bk && p_bk == 2;
        const int bk_nb = p_bk_nb, bk_shift = p_bk_shift;
        // counters / offsets per key: bk_nb time buckets; with LDS-resident offsets also the bucket of the items that stay until the
        // end of the horizon (a path's last waypoint -- no copies of it in every later time bucket) and a padding entry
        const int bk_w = bk_lds ? bk_nb + 2 : bk_nb;
        uint32_t *bkc = bk_lds ? LDS_AT(uint32_t, L_BKREL) : reinterpret_cast<uint32_t *>(wave_scr);
        if (!reuse) {
            for (int k = tid; k <= K; k += nt) csr[k] = 0;
            if (tid == 0) misc[11] = 0;
            if (X.tmask) for (int k = tid; k <= K; k += nt) { tmask[k] = 0ull; if (X.tmask_m2) tmask_m2[k] = 0ull; }
            if (bk) for (int k = tid; k < (bk_lds ? K * bk_w / 2 : ((K + 1) * bk_nb + 1) / 2); k += nt) bkc[k] = 0u;
        }
        if (dual) for (int k = tid; k <= K; k += nt) { csr2[k] = 0; if (p_use_tmask) tmaskb[k] = 0ull; if (X.tmask_m2) tmaskb_m2[k] = 0ull; }
        if (STAGE != 2 && tid < 64) { team_meta[64 + tid] = 1; team_meta[192 + tid] = 1; }   // (trees the hoisted pass A below does not build: the root alone)
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
            uint16_t *path = S.path + ((size_t)b * A + ia) * OBS_PRED_CAP;
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
            // (bucketed lists with LDS-resident offsets count per bucket right here -- there the walk ends before the pass A beside it
            // anyway; the one item whose interval runs to the end of the horizon, the last one of the path, is only known after the
            // walk and is corrected then.  With hundreds of agents the walk IS the critical path and two LDS atomics per hop doubled
            // it (166 -> 382 us at cfg5): those lists are counted in a pass of their own below, all lanes at once)
            const bool listed = !lab || lab[ia] >= 0;   // (an agent that is not in `handles` has no entry in predicted_pos)
            const int hz1 = ((bk && !bk_lds) || !listed) ? -1 : max(0, CUTILS ? (X.Tn - 2) / (int)a_tpc[ia] + 1 : (X.Tn - 1) / (int)a_tpc[ia]);
            const int tpc_w = a_tpc[ia], tlast_w = X.Tn - 1;
            uint32_t st_hz = 0;
            const int hz2 = dual ? max(0, min(P.tree_pred - 1, (Tn2 - 1) / (int)a_tpc2[ia])) : -1;
            auto walk8 = [&](const uint16_t *h8) __attribute__((always_inline)) {
                int idx = j, last = -1;
                while (__any(alive)) {
                    if (alive) {
                        path[idx] = (uint16_t)st;
                        last = idx;
                        if (idx <= hz1 || idx <= hz2) {
                            const int key = key_of(X, (int)(st >> 2));
                            if (idx <= hz1) {
                                if (bk_lds) {  // cutils: waypoint idx is occupied during [(idx - 1) * tpc + 1, idx * tpc] (0 for idx = 0)
                                    const int tlo = idx == 0 ? 0 : (idx - 1) * tpc_w + 1, thi = min(idx * tpc_w, tlast_w);
                                    const int b1 = min(tlo >> bk_shift, bk_nb - 1), b2 = min(thi >> bk_shift, bk_nb - 1);
                                    for (int bb = b1; bb <= b2; bb++) {
                                        const int kb = key * bk_w + bb;
                                        atomicAdd(&bkc[kb >> 1], (kb & 1) ? 0x10000u : 1u);
                                    }
                                    atomicAdd(&csr[key], b2 - b1 + 1);
                                    if (idx == hz1) st_hz = st;
                                } else {
                                    atomicAdd(&csr[key], 1);
                                }
                            }
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
            const int my_last = TAB_LDS ? walk8(hop8_lds + u * Scap) : walk8(ghop8 + (size_t)u * Scap);
            int m = my_last;
            m = max(m, __shfl_xor(m, 1)); m = max(m, __shfl_xor(m, 2)); m = max(m, __shfl_xor(m, 4));
            if (bk_lds && have && listed) {
                // the last indexed waypoint lp stays occupied until the end of the horizon: from its time bucket(s) to the bucket of
                // such items
                const int lp = max(0, min(m, hz1));
                if (lp == m ? my_last == lp : j == (lp & 7)) {
                    const uint32_t s_lp = lp == m ? st : st_hz;
                    const int key = key_of(X, (int)(s_lp >> 2));
                    const int tlo = lp == 0 ? 0 : (lp - 1) * tpc_w + 1, thi = min(lp * tpc_w, tlast_w);
                    const int b1 = min(tlo >> bk_shift, bk_nb - 1), b2 = min(thi >> bk_shift, bk_nb - 1);
                    for (int bb = b1; bb <= b2; bb++) {
                        const int kb = key * bk_w + bb;
                        atomicSub(&bkc[kb >> 1], (kb & 1) ? 0x10000u : 1u);
                    }
                    // (bk_w is even: the count of that bucket is the low half of the key's last word, the high half collects the
                    // time buckets in which such items start)
                    atomicAdd(&bkc[(key * bk_w + bk_nb) >> 1], 1u);
                    atomicOr(&bkc[(key * bk_w + bk_nb) >> 1], 0x10000u << (b1 >> OBS_FB_MSHIFT));
                    if (b2 > b1) atomicSub(&csr[key], b2 - b1);
                }
            }
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
#ifdef FL_OBS_TIMING
            const long long t_pa0 = (long long)wall_clock64();
#endif
            if (CUTILS && (!bk || bk_lds) && P.max_nodes <= OBS_CAP_C) {  // pass A of the first round's trees (not while the node tables hold the bucket counters)
                // teams of 16 lanes, four trees a wavefront, where no cell has more than two transitions a direction (cutils_pass_a); else 32 lanes
                const int n_first = min(A, merged ? ROUND : (nt >> 6) * 2);   // trees of the first round
                auto hoisted = [&](auto team_tag) __attribute__((always_inline)) {
                    constexpr int TEAM = decltype(team_tag)::value;
                    const int grp = lane / TEAM, gl = lane % TEAM, team_id = wave * (64 / TEAM) + grp;
                    if (__builtin_amdgcn_readfirstlane(wave * (64 / TEAM)) >= n_first) return;   // (no tree on this wavefront)
                    int node_base, levels;
                    const bool have = team_id < n_first;
                    cutils_pass_a<OBS_CAP_C, TEAM>(X, d, P, b, team_id, have, grp, gl,
                                  merged ? merged_table_c(wave_scr, min(team_id, ROUND - 1)) : wave_scr + min(team_id, min((nt >> 6) * 2, A)) * (N_WORDS_C * OBS_CAP_C),
                                  a_vpos, a_pos, a_dir, a_state, a_speed, a_tslot, (float)T,
                                  a_raw ? a_raw[(have ? team_id : 0) * 8 + 1] : d.spk[b * A + (have ? team_id : 0)],
                                  a_raw ? a_raw[(have ? team_id : 0) * 8 + 2] : d.malf[b * A + (have ? team_id : 0)], node_base, levels);
                    if (gl == 0 && have) { team_meta[64 + team_id] = node_base; team_meta[192 + team_id] = levels; }
                };
                if (p_compact_t) hoisted(std::integral_constant<int, 16>());
                else hoisted(std::integral_constant<int, 32>());
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
        if (bk && !bk_lds) {  // the pass that counts the bucketed lists of large maps: one copy of an item per time bucket its interval
                              // touches (cutils: w(t) = 0 for t = 0, min((t-1)/tpc + 1, lp)); four waypoints per lane a round trip
            __syncthreads();
            for (int i = wave; i < A; i += (nt >> 6)) {
                if (lab && lab[i] < 0) continue;
                const uint16_t *path = S.path + ((size_t)b * A + i) * OBS_PRED_CAP;
                const int lp = a_lp[i];
                const int tpc = a_tpc[i], tlast = X.Tn - 1;
                for (int k0 = lane; k0 <= lp; k0 += 256) {
                    uint32_t wv[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) wv[q] = path[min(k0 + 64 * q, lp)];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int k = k0 + 64 * q;
                        if (k > lp) break;
                        const int key = key_of(X, (int)(wv[q] >> 2));
                        const int tlo = k == 0 ? 0 : (k - 1) * tpc + 1, span = k == 0 ? 1 : tpc;
                        const int thi = (k == lp || tlo + span - 1 >= tlast) ? tlast : tlo + span - 1;
                        const int b1 = min(tlo >> bk_shift, bk_nb - 1), b2 = min(thi >> bk_shift, bk_nb - 1);
                        for (int bb = b1; bb <= b2; bb++) {  // bucket-major: row bb, entry key + 1 (entry 0 of a row stays 0)
                            const int kb = bb * (K + 1) + key + 1;
                            atomicAdd(&bkc[kb >> 1], (kb & 1) ? 0x10000u : 1u);
                        }
                    }
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
                const uint16_t *path = S.path + ((size_t)b * A + i) * OBS_PRED_CAP;
                const int lp = a_lp[i];
                for (int k = lane; k <= lp; k += 64) atomicAdd(&csr[key_of(X, (int)(path[k] >> 2))], 1);
            }
        }
        __syncthreads();
        OBS_STAMP(3);
        // exclusive scan over the keys: per-thread chunk sums, wave-0 scan of the partial sums, rescan.  With the second
        // index both counts share the scan, 16 bits each (the launcher guarantees totals below 65536).
        const bool bk_major = bk && !bk_lds;   // large maps: the items laid out bucket-major (see the fill)
        if (!reuse && !bk_major) {
            const int chunk = (K + 1 + nt - 1) / nt;
            const int lo = min(tid * chunk, K + 1), hi = min(lo + chunk, K + 1);
            int sum = 0;
            int longest = 0;
            for (int k = lo; k < hi; k++) {
                sum += dual ? (csr[k] | (csr2[k] << 16)) : csr[k];
                longest = max(longest, dual ? max(csr[k], csr2[k]) : csr[k]);
            }
            // (a query of the bucketed index scans two or three buckets of a list: see below)
            if (longest > CF_DIRECT && !bk) misc[11] = 1;
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
        if (bk_major) {
            // Per (time bucket, key) counts -> the start of the key's items inside the bucket (16 bits, relative to the bucket's start;
            // the starts of the buckets: misc[12 ..]); one exclusive scan over the bk_nb rows of K + 1 entries.
            uint16_t *c16 = reinterpret_cast<uint16_t *>(bkc);
            const int K1 = K + 1, n = K1 * bk_nb;
            {   // what one query can scan: a key's items of two neighbouring buckets
                int most = 0;
                for (int key = tid; key < K; key += nt) {
                    int prev = 0;
                    for (int bb = 0; bb < bk_nb; bb++) { const int c = c16[bb * K1 + key + 1]; most = max(most, prev + c); prev = c; }
                }
                if (most > CF_DIRECT) misc[11] = 1;
            }
            const int chunk = (n + nt - 1) / nt;
            const int lo = min(tid * chunk, n), hi = min(lo + chunk, n);
            int sum = 0;
            for (int k = lo; k < hi; k++) sum += c16[k];
            partial[tid] = sum;
            __syncthreads();
            if (wave == 0) {
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
            {   // starts of the buckets (a lane's range crosses at most a few row starts)
                int run = partial[tid];
                for (int k = lo; k < hi; k++) {
                    if (k % K1 == 0) misc[12 + k / K1] = run;
                    run += c16[k];
                }
                if (hi == n && lo < hi) misc[2] = run;   // total number of items
            }
            __syncthreads();
            {
                int run = partial[tid], row = lo / K1, rbase = lo < n ? misc[12 + row] : 0, next = (row + 1) * K1;
                for (int k = lo; k < hi; k++) {
                    if (k == next) { row++; rbase = misc[12 + row]; next += K1; }
                    const int c = c16[k];
                    c16[k] = (uint16_t)(run - rbase);
                    run += c;
                }
            }
        }
        __syncthreads();
        if (reuse) {  // stage 1 built this index
            csr = csr2; X.csr_end = csr2; X.items_lds = items2;
            X.tmask = p_use_tmask ? tmaskb : nullptr;
            if (!X.tmask) { X.wl_occ_cap = wl_entries; X.wl_cf = X.wl_occ + X.wl_occ_cap; X.wl_cf_cap = 0; }
        }
        if (bk_lds) {  // counts of a key's buckets -> their start offsets inside the key's list (bumped to the ends by the fill)
            for (int key = tid; key < K; key += nt) {
                uint32_t *w4 = bkc + key * (bk_w / 2);
                uint32_t run = 0, prev = 0, most = 0;  // most: the longest run of three consecutive buckets (what one query can scan)
                for (int q = 0; q < bk_nb / 2; q++) {
                    const uint32_t v = w4[q], c0 = v & 0xFFFFu, c1 = v >> 16;
                    most = max(most, max(prev + c0 + c1, c0 + c1));
                    prev = c0 + c1;
                    w4[q] = run | ((run + c0) << 16);
                    run += c0 + c1;
                }
                if (bk_lds) {  // ... plus the items that stay until the end
                    const uint32_t ve = w4[bk_nb / 2];
                    w4[bk_nb / 2] = run | (ve & 0xFFFF0000u);
                    most += ve & 0xFFFFu;
                }
                if (most > CF_DIRECT) misc[11] = 1;
            }
            __syncthreads();
        }
#ifdef FL_OBS_TIMING
        if (STAGE != 2) { __syncthreads(); if (tid == 0) P.dbg[(size_t)b * 64 + 56] = (long long)wall_clock64(); }   // keys scanned, bucket offsets done
#endif
        const bool fit = items_lds != nullptr && misc[2] <= L_FIELD(items_cap);
        const bool dual_fill = dual && misc[3] <= L_FIELD(items2_cap);
        if (dual && tid == 0) misc[4] = dual_fill ? 1 : 0;
        if (merged && UP && !dual_fill) atomicCAS(&d.err[b], 0, FL_ERR_CAPACITY);  // (cannot happen: the LDS copy holds the exact bound)
        if (fit && !reuse) { csr_items = items_lds; X.items_lds = items_lds; }
        // fill: bumping csr[key] turns it from the start into the END offset of key's list (start = csr[key - 1]);
        // one wavefront per agent, one lane per waypoint
        // (the waypoints come from HBM scratch: four of them per lane are requested at once -- 500 waypoints are two round trips
        // instead of eight -- and the first four of the agent the wavefront takes NEXT are requested before the items of this
        // one are emitted)
        constexpr int FU = 4;
        uint32_t pfv[FU] = {0, 0, 0, 0}, pfn[FU] = {0, 0, 0, 0}, pfp[FU] = {0, 0, 0, 0};
        auto prefetch = [&](int ia) __attribute__((always_inline)) {
            if (reuse || ia >= A) return;
            const uint16_t *pth = S.path + ((size_t)b * A + ia) * OBS_PRED_CAP;
            const int lpn = a_lp[ia];
#pragma unroll
            for (int q = 0; q < FU; q++) {
                const int k = min(lane + 64 * q, lpn);
                pfv[q] = pth[k]; pfn[q] = pth[min(k + 1, lpn)]; pfp[q] = pth[max(k - 1, 0)];
            }
        };
        if (bk_major) {
            // Large maps: the items go to HBM scratch, hundreds of thousands of scattered 4-byte stores per env.  Laid out and EMITTED
            // bucket-major -- all items of time bucket 0 (by key), then bucket 1, ... -- the lines a CU has open at a time are one
            // bucket's share of its items (65 KB at cfg5), which stays in the L2 until the lines are full; key-major the 0.5 MB of an
            // env did not, and every 4-byte store left the L2 as a 32-byte write.  The waypoints of an agent whose interval touches a
            // bucket are a contiguous piece of its path (at most 64 / tpc + 2; the last indexed waypoint stays until the end of the
            // horizon): per bucket the pieces of all agents are laid end to end (a prefix over the agents in the scan scratch) and
            // split evenly over the lanes of the workgroup; the waypoints of a lane's NEXT item are requested before this one is emitted.
            const int K1 = K + 1, tlast = X.Tn - 1;
            uint16_t *pre16 = reinterpret_cast<uint16_t *>(partial);       // [A + 1] start of agent i's piece among the bucket's items
            uint16_t *klo16 = pre16 + ((A + 2) & ~1);                        // [A] first waypoint of the piece
            for (int bb = 0; bb < bk_nb; bb++) {
                const int t0 = bb << bk_shift, t1 = bb == bk_nb - 1 ? tlast : min(t0 + (1 << bk_shift) - 1, tlast);
                if (t0 > tlast) break;
                const int gbase = misc[12 + bb];
                for (int ia = tid; ia < A; ia += nt) {
                    const int lp = a_lp[ia], tpc = a_tpc[ia];
                    const int k_lo = min((t0 + tpc - 1) / tpc, lp);              // first waypoint still occupied at t0 (thi = k * tpc)
                    const int k_hi = min(t1 >= 1 ? (t1 - 1) / tpc + 1 : 0, lp);  // last waypoint entered by t1 (tlo = (k - 1) * tpc + 1)
                    klo16[ia] = (uint16_t)k_lo;
                    pre16[ia + 1] = (lab && lab[ia] < 0) ? (uint16_t)0 : (uint16_t)(k_hi - k_lo + 1);
                }
                if (tid == 0) pre16[0] = 0;
                __syncthreads();
                if (wave == 0) {  // inclusive prefix over the agents' piece lengths (A <= 992: at most 16 a lane)
                    const int per = (A + 63) >> 6, a_lo = min(lane * per, A), a_hi = min(a_lo + per, A);
                    int sum = 0;
                    for (int ia = a_lo; ia < a_hi; ia++) sum += pre16[ia + 1];
                    int incl = sum;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
                    int run = incl - sum;
                    for (int ia = a_lo; ia < a_hi; ia++) { run += pre16[ia + 1]; pre16[ia + 1] = (uint16_t)run; }
                }
                __syncthreads();
                const int total = pre16[A];
                // item t of the bucket -> (agent, waypoint): the last agent whose piece starts at or before t
                uint32_t nv = 0, nn = 0, np = 0;
                int n_i = 0, n_k = 0;
                auto request = [&](int t) __attribute__((always_inline)) {
                    if (t >= total) return;
                    int lo_a = 0, hi_a = A - 1;
                    while (lo_a < hi_a) {
                        const int mid = (lo_a + hi_a + 1) >> 1;
                        if ((int)pre16[mid] <= t) lo_a = mid; else hi_a = mid - 1;
                    }
                    n_i = lo_a;
                    n_k = (int)klo16[lo_a] + (t - (int)pre16[lo_a]);
                    const uint16_t *pth = S.path + ((size_t)b * A + lo_a) * OBS_PRED_CAP;
                    const int lp = a_lp[lo_a];
                    nv = pth[n_k]; nn = pth[min(n_k + 1, lp)]; np = pth[max(n_k - 1, 0)];
                };
                request(tid);
                for (int t = tid; t < total; t += nt) {
                    const int i = n_i, k = n_k;
                    const uint32_t w = nv, wnx = nn, wpv = np;
                    request(t + nt);
                    const int lp = a_lp[i], tpc = a_tpc[i];
                    const int tlo = k == 0 ? 0 : (k - 1) * tpc + 1, span = k == 0 ? 1 : tpc;
                    const bool to_end = k == lp || tlo + span - 1 >= tlast;
                    const int thi = to_end ? tlast : tlo + span - 1;
                    const int b1 = min(tlo >> bk_shift, bk_nb - 1), b2 = min(thi >> bk_shift, bk_nb - 1);
                    if (bb < b1 || bb > b2) continue;
                    const int key = key_of(X, (int)(w >> 2));
                    if (X.tmask && bb == b1) {  // time-mask buckets this item covers (once per item)
                        const int m1 = tb_of(tlo, X.tshift), m2 = tb_of(thi, X.tshift);
                        atomicOr(&tmask[key], ((2ull << m2) - 1ull) & ~((1ull << m1) - 1ull));
                    }
                    const uint32_t dnext = k < lp ? (wnx & 3u) : (w & 3u), dprev = k > 0 ? (wpv & 3u) : (w & 3u);
                    const uint32_t item = IT_MAKE(lab ? (int)lab[i] : i, tlo, to_end, span, dprev, dnext, w & 3u);
                    const int kb = bb * K1 + key + 1;
                    const uint32_t old = atomicAdd(&bkc[kb >> 1], (kb & 1) ? 0x10000u : 1u);
                    csr_items[gbase + (int)((kb & 1) ? (old >> 16) : (old & 0xFFFFu))] = item;
                }
                __syncthreads();   // (the next bucket's prefix goes to the same scratch)
            }
        }
        prefetch(bk_major ? A : wave);
        for (int i = wave; !reuse && !bk_major && i < A; i += (nt >> 6)) {
            const uint16_t *path = S.path + ((size_t)b * A + i) * OBS_PRED_CAP;
            const int lp = (lab && lab[i] < 0) ? -1 : (int)a_lp[i], tpc = a_tpc[i], tlast = X.Tn - 1;   // (not listed: no items)
            const int lp2 = dual_fill ? (int)a_lp2[i] : -1, tpc2 = dual_fill ? (int)a_tpc2[i] : 1;
            uint32_t wv[FU], wnx[FU], wpv[FU];
#pragma unroll
            for (int q = 0; q < FU; q++) { wv[q] = pfv[q]; wnx[q] = pfn[q]; wpv[q] = pfp[q]; }
            prefetch(i + (nt >> 6));
            for (int k0 = lane; k0 <= lp; k0 += 64 * FU) {
            if (k0 != lane) {
#pragma unroll
                for (int q = 0; q < FU; q++) {
                    const int k = min(k0 + 64 * q, lp);
                    wv[q] = path[k]; wnx[q] = path[min(k + 1, lp)]; wpv[q] = path[max(k - 1, 0)];
                }
            }
#pragma unroll
            for (int q = 0; q < FU; q++) {
                const int k = k0 + 64 * q;
                if (k > lp) break;
                const uint32_t w = wv[q];
                const uint32_t dnext = k < lp ? (wnx[q] & 3u) : (w & 3u), dprev = k > 0 ? (wpv[q] & 3u) : (w & 3u);
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
                    const int b1 = tb_of(tlo, X.tshift), b2 = tb_of(to_end ? tlast : tlo + span - 1, X.tshift);
                    const unsigned long long bits = ((2ull << b2) - 1ull) & ~((1ull << b1) - 1ull);
                    if (X.tmask_m2) {
                        const unsigned long long seen = atomicOr(&tmask[key], bits);
                        if (seen & bits) atomicOr(&tmask_m2[key], seen & bits);  // covered by a second item
                    } else {
                        atomicOr(&tmask[key], bits);
                    }
                }
                const uint32_t item = IT_MAKE(lab ? (int)lab[i] : i, tlo, to_end, span, dprev, dnext, w & 3u);
                if (bk) {  // csr[key] stays the START of the key's list; the bucket's running offset is bumped
                    const int thi = to_end ? tlast : tlo + span - 1;
                    int b1 = min(tlo >> bk_shift, bk_nb - 1), b2 = min(thi >> bk_shift, bk_nb - 1);
                    if (bk_lds && to_end) b1 = b2 = bk_nb;  // the bucket of the items that stay until the end
                    for (int bb = b1; bb <= b2; bb++) {
                        const int kb = key * bk_w + bb;
                        const uint32_t old = atomicAdd(&bkc[kb >> 1], (kb & 1) ? 0x10000u : 1u);
                        csr_items[csr[key] + (int)((kb & 1) ? (old >> 16) : (old & 0xFFFFu))] = item;
                    }
                } else {
                    const int slot = atomicAdd(&csr[key], 1);
                    csr_items[slot] = item;
                }
                if (k <= lp2) {  // the same waypoint in the upstream predictor's index: w(t) = min(t / tpc, lp)
                    const int tlo2 = k * tpc2, tlast2 = Tn2 - 1;
                    const bool to_end2 = k == lp2 || tlo2 + tpc2 - 1 >= tlast2;
                    const uint32_t dnext2 = k < lp2 ? dnext : (w & 3u);
                    if (p_use_tmask) {
                        const int b1 = tb_of(tlo2, tshift2), b2 = tb_of(to_end2 ? tlast2 : tlo2 + tpc2 - 1, tshift2);
                        const unsigned long long bits = ((2ull << b2) - 1ull) & ~((1ull << b1) - 1ull);
                        if (X.tmask_m2) {
                            const unsigned long long seen = atomicOr(&tmaskb[key], bits);
                            if (seen & bits) atomicOr(&tmaskb_m2[key], seen & bits);
                        } else {
                            atomicOr(&tmaskb[key], bits);
                        }
                    }
                    const int slot2 = atomicAdd(&csr2[key], 1);
                    items2[slot2] = IT_MAKE(i, tlo2, to_end2, tpc2, dprev, dnext2, w & 3u);
                }
            }
            }
        }
        __syncthreads();
#ifdef FL_OBS_TIMING
        if (STAGE != 2 && tid == 0) {
            P.dbg[(size_t)b * 64 + 57] = (long long)wall_clock64();   // items filled
            // what this env keeps in scratch: prediction items of the first index, waypoints of the predicted paths
            int wp = 0;
            for (int i = 0; i < A; i++) wp += a_n[i];
            P.dbg[(size_t)b * 64 + 58] = misc[2];
            P.dbg[(size_t)b * 64 + 59] = wp;
        }
#endif
        if (bk) {  // LDS-resident offsets: the list of key k is [csr[k], csr[k + 1]) now and its bucket ends stay in LDS; bucket-major
                   // (large maps): the ends of the keys inside every bucket go to HBM (the node tables take their LDS back)
            if (bk_lds) {
                X.csr_end = csr + 1;
                X.bk_rel_lds = reinterpret_cast<const uint16_t *>(bkc);
            } else {
                uint32_t *g = reinterpret_cast<uint32_t *>(S.bk_rel + (size_t)b * (d.Rcap + 1) * OBS_BK_NB);
                for (int k = tid; k < ((K + 1) * bk_nb + 1) / 2; k += nt) g[k] = bkc[k];
                X.bk_rel = S.bk_rel + (size_t)b * (d.Rcap + 1) * OBS_BK_NB;
                X.bk_base = misc + 12;
                X.bk_k1 = K + 1;
                __syncthreads();
            }
        }
    }

    OBS_STAMP(4);
    // ---- phase 3: trees.  Pass A derives the topology of a tree from the static segment table (O(1) per node, one
    // BFS level per step); pass B evaluates the agent-dependent features with the visited cells of all nodes split
    // evenly over the lanes of the workgroup (wg_pass_b); then one lane per node writes its row.
    const float max_dist = (float)T;
    const int nwaves = nt >> 6;
    const bool items_in_lds = X.items_lds != nullptr;
    if (merged) {
        if (items_in_lds) trees_merged<true, MERGED >= 2, ROUND, UP>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, a_vpos, a_pos, a_dir, a_state, a_speed, a_tslot, max_dist, late_jobs, (FIX == 1 || FIX == 5 || !UP) ? nullptr : S.rowmask);
        else trees_merged<false, MERGED >= 2, ROUND, UP>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, a_vpos, a_pos, a_dir, a_state, a_speed, a_tslot, max_dist, late_jobs, (FIX == 1 || FIX == 5 || !UP) ? nullptr : S.rowmask);
    } else if (CUTILS && STAGE == 0 && FIX == 0 && P.max_nodes > OBS_CAP_C) {
        // more than 32 nodes a tree (the stand-alone flatland_cutils launch only): a team of 64 lanes, one tree a wavefront
        if (items_in_lds) trees_cutils<true, 64, true>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, a_vpos, a_pos, a_dir, a_state, a_speed, a_tslot, max_dist, false);
        else trees_cutils<false, 64, true>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, a_vpos, a_pos, a_dir, a_state, a_speed, a_tslot, max_dist, false);
    } else if (CUTILS) {
        if (items_in_lds) trees_cutils<true, OBS_CAP_C, STAGE == 0>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, a_vpos, a_pos, a_dir, a_state, a_speed, a_tslot, max_dist, STAGE != 2 && X.Tn > 0 && X.bk_rel == nullptr);
        else trees_cutils<false, OBS_CAP_C, STAGE == 0>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, a_vpos, a_pos, a_dir, a_state, a_speed, a_tslot, max_dist, STAGE != 2 && X.Tn > 0 && X.bk_rel == nullptr);
    } else if (p_compact_t && STAGE == 0 && FIX == 0 && P.max_depth >= 4) {
        // depth 4 (341 rows): 30 compact slots, a team of 32 lanes, two trees a wavefront -- the stand-alone tree launch only
        if (items_in_lds) tree_upstream<32, 32, true, true>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, nullptr);
        else tree_upstream<32, 32, true, false>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, nullptr);
    } else if (p_compact_t) {
        if (items_in_lds) tree_upstream<16, OBS_CAP_T_COMPACT, true, true>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, S.rowmask);
        else tree_upstream<16, OBS_CAP_T_COMPACT, true, false>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, S.rowmask);
    } else if (P.max_depth <= 2) {
        if (items_in_lds) tree_upstream<32, 32, false, true>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, S.rowmask);
        else tree_upstream<32, 32, false, false>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, S.rowmask);
    } else {
        if (items_in_lds) tree_upstream<64, 88, false, true>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, S.rowmask);
        else tree_upstream<64, 88, false, false>(X, d, P, b, wave, lane, nwaves, wave_scr, team_meta, S.rowmask);
    }
    OBS_STAMP(5);
#undef LDS_AT
#undef LDS_OPT
}

// MODE 0 = flatland_cutils outputs, 1 = upstream dense tree, 2 = both in one launch (two stages), 3 = both with one pass B per
// round (MERGED: 3 = envs of at most 32 agents, one round; 4 = rounds of 32 agents; 5 = rounds of 16 agents on 512 threads, two
// workgroups a CU); VAR: see obs_body
template <int MODE, int VAR, int FIX>
__device__ __forceinline__ void obs_kernel_body(const FlDev &d, const FlObsScratch &S, const ObsArgs &P) {
    // what this env takes goes to S.cost: the next launch starts the longest envs first.  Env and start clock wait in two LDS words
    // (not in registers: the kernel sits at its register ceiling, and every scalar that lives through it costs spills)
    if (MODE != 3 && MODE != 6 && S.order && threadIdx.x == 0) {
        extern __shared__ __align__(16) unsigned char lds[];
        int *misc = reinterpret_cast<int *>(lds + (FIX != 0 ? ObsFixed<FIX != 0 ? FIX : 1>::L.off[L_MISC] : P.L.off[L_MISC]));
        misc[62] = obs_env_of_workgroup(S);
        misc[63] = (int)(uint32_t)wall_clock64();
    }
    if (MODE == 0) obs_body<true, VAR, 0, 0, FIX>(d, S, P);
    else if (MODE == 1) obs_body<false, VAR, 0>(d, S, P);
    else if (MODE == 3) obs_body<true, VAR, 1, 1, FIX>(d, S, P);
    else if (MODE == 4) obs_body<true, VAR, 1, 2, FIX>(d, S, P);
    else if (MODE == 5) obs_body<true, VAR, 1, 3, FIX>(d, S, P);
    else if (MODE == 6) obs_body<true, VAR, 1, 1, FIX, false>(d, S, P);   // MODE 3 / 4 / 5 with the flatland_cutils builder alone
    else if (MODE == 7) obs_body<true, VAR, 1, 2, FIX, false>(d, S, P);
    else if (MODE == 8) obs_body<true, VAR, 1, 3, FIX, false>(d, S, P);
    else {
        obs_body<true, VAR, 1, 0, FIX>(d, S, P);
        __syncthreads();
        obs_body<false, VAR, 2, 0, FIX>(d, S, P);
    }
    if (MODE != 3 && MODE != 6 && S.order && threadIdx.x == 0) {
        extern __shared__ __align__(16) unsigned char lds[];
        const int *misc = reinterpret_cast<const int *>(lds + (FIX != 0 ? ObsFixed<FIX != 0 ? FIX : 1>::L.off[L_MISC] : P.L.off[L_MISC]));
        S.cost[misc[62]] = (uint32_t)wall_clock64() - (uint32_t)misc[63];
    }
}
template <int MODE, int VAR, int FIX = 0>
__global__ __launch_bounds__(OBS_NT) void k_obs(FlDev d, FlObsScratch S, ObsArgs P) { obs_kernel_body<MODE, VAR, FIX>(d, S, P); }

// A batch whose largest map exceeds a class's rail cells (P.split): the workgroup looks at ITS env -- the class's body (compile-time
// carving) when the env fits the class, the runtime-carving body (P.L) otherwise; one launch, one order of the workgroups.  The
// class's capacities bound what its carving holds per env (R, K <= dims.Rcap); the HBM strides are the batch's (d.Rcap) either way.
// FIX2 != 0 (P.split 2): the other envs run the larger bin class FIX2 (same MODE and VAR) instead of the runtime carving.
template <int MODE, int VAR, int FIX, int FIX2 = 0>
__global__ __launch_bounds__(OBS_NT) void k_obs_split(FlDev d, FlObsScratch S, ObsArgs P) {
    const int b = (MODE == 3 || MODE == 6) ? (int)blockIdx.x : obs_env_of_workgroup(S);
    if (d.R[b] <= ObsFixed<FIX>::dims.Rcap) obs_kernel_body<MODE, VAR, FIX>(d, S, P);
    else obs_kernel_body<MODE, VAR, FIX2>(d, S, P);
}
