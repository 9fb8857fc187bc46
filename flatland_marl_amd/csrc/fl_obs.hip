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
// Layout of one launch (gfx950): the env's rail bitmap (u16 H*W) and the per-cell occupancy maps are staged
// in LDS once; phase 1 runs one lane per agent (snapshot, deadlock fixpoint, 83-float attribute row);
// phase 2 walks every agent's predicted path (<= 500 dependent distance-map gathers) and builds a
// per-cell CSR index of (agent, waypoint) pairs in HBM scratch; phase 3 builds the trees with one
// wavefront per agent: every BFS level is explored by one lane per queue entry, children are handed to
// the next level's lanes with wave shuffles (no queue in memory), node rows are written straight to HBM.
#include "fl_obs.h"

#include <string.h>

#include "../../include/flatland_hip.h"

#define OBS_NT 512
#define OBS_WAVES (OBS_NT / 64)

// ---------------------------------------------------------------------------------------------- context
struct ObsCtx {
    int A, H, W, HW, K;           // K = number of prediction keys (col * W + row), tool.h:391-398
    const uint16_t *grid;         // LDS
    const int16_t *cell_agent;    // LDS: highest on-map handle on the cell (last writer of location_has_agent*), -1
    const uint16_t *cell_ready;   // LDS: number of off-map agents whose initial position is the cell
    const uint32_t *cell_target;  // LDS bitmap: some agent's target (upstream location_has_target)
    const int *a_vpos;            // LDS per agent: virtual position (cell)
    const uint8_t *a_dir, *a_state;
    const uint16_t *a_malf;       // real down counter
    const double *a_speed;
    const uint16_t *a_tpc;        // times per cell of the predictor
    const uint16_t *a_lp;         // last reachable waypoint index (0 = holds at its virtual position)
    const int *a_tslot;
    const int *a_target;
    const uint32_t *path;         // HBM [A][pcap] cell << 2 | dir
    int pcap;
    const int *csr_head;          // HBM [K + 1]
    const uint32_t *csr_items;    // HBM: agent << 12 | waypoint << 2 | dir
    int Tn;                       // number of predicted time entries (0 = no predictor)
    const uint16_t *dm;           // HBM env base [Umax][HW][4]
};

// waypoint index of agent a at predicted time t
template <bool CUTILS>
__device__ __forceinline__ int waypoint_at(const ObsCtx &X, int a, int t) {
    const int lp = X.a_lp[a], tpc = X.a_tpc[a];
    if (CUTILS) {  // predictions.cpp:207-227: entry t >= 1 is produced by loop index t-1, first advance at index 0
        if (t == 0) return 0;
        return min((t - 1) / tpc + 1, lp);
    }
    return min(t / tpc, lp);  // predictions.py:159-174: advance when index % times_per_cell == 0, index >= 1
}

struct BranchOut {
    double f[12];
    int end_cell;
    uint32_t end_dir;
    int tot_dist;
    bool is_switch, is_dead_end, is_terminal, is_target, zero_transition, cycle_suspect;
};

// deterministic successor of a walk state while the walk keeps going; -1 when the walk stops there
__device__ __forceinline__ int walk_next(const ObsCtx &X, int state, int target) {
    const int cell = state >> 2;
    const uint32_t d = state & 3;
    if (cell == target) return -1;
    const uint32_t g = X.grid[cell];
    const uint32_t bits = nibble(g, d);
    if (__popc(bits) != 1) return -1;
    int total = __popc(g);
    if (g == 0x8421u) total = 2;
    if (total == 1) return -1;
    const uint32_t nd = first_dir(bits);
    return (step_cell(cell, nd, X.W) << 2) | (int)nd;
}

// index of the first revisited state of the (non-terminating) walk from `start`: Brent's cycle detection
__device__ int first_repeat_index(const ObsCtx &X, int start, int target) {
    int power = 1, lam = 1, tort = start, hare = walk_next(X, start, target);
    while (tort != hare) {
        if (power == lam) { tort = hare; power *= 2; lam = 0; }
        hare = walk_next(X, hare, target);
        lam++;
    }
    tort = hare = start;
    for (int i = 0; i < lam; i++) hare = walk_next(X, hare, target);
    int mu = 0;
    while (tort != hare) { tort = walk_next(X, tort, target); hare = walk_next(X, hare, target); mu++; }
    return mu + lam;
}

// _explore_branch: treeobs.cpp:258-610 (CUTILS) / observations.py:256-494 (upstream).
// stop_at_visit >= 0: the visit with that index is a revisited (cell, dir) -> terminal (cycle), see caller.
template <bool CUTILS>
__device__ void explore_branch(const ObsCtx &X, int handle, int cell, uint32_t d, int tot_dist, int stop_at_visit,
                               BranchOut &o) {
    const int W = X.W;
    const int target = X.a_target[handle];
    double own_target = INFINITY, other_agent = INFINITY, other_target = INFINITY, pot_conflict = INFINITY,
           unusable = INFINITY, min_speed = 1.0;
    int same_dir = 0, opp_dir = 0, malfunctioning = 0, ready = 0;
    const float tpc_f = (float)(1.0 / (double)(float)X.a_speed[handle]);  // float time_per_cell = 1.0 / agent.speed
    const double tpc_d = 1.0 / X.a_speed[handle];                          // np.reciprocal(speed)
    o.is_switch = o.is_dead_end = o.is_terminal = o.is_target = o.zero_transition = o.cycle_suspect = false;
    const int max_visits = 4 * X.HW + 4;
    int visit = 0;
    while (true) {
        const int ag = X.cell_agent[cell];
        if (ag >= 0) {  // treeobs.cpp:322-357
            if ((double)tot_dist < other_agent) other_agent = tot_dist;
            const int mf = CUTILS ? (X.a_malf[ag] != 0) : (int)X.a_malf[ag];
            if (mf > malfunctioning) malfunctioning = mf;
            const int rd = X.cell_ready[cell];
            if (rd > 0) ready += CUTILS ? rd - 1 : rd;  // cutils starts the count at 0 (treeobs.cpp:82-91)
            if (X.a_dir[ag] == d) {
                same_dir += 1;
                const double sp = CUTILS ? (double)(float)X.a_speed[ag] : X.a_speed[ag];
                if (sp < min_speed) min_speed = sp;
            } else {
                opp_dir += 1;
            }
        }
        const uint32_t g = X.grid[cell];
        const uint32_t bits = nibble(g, d);
        int total = __popc(g);
        const bool crossing = g == 0x8421u;
        if (X.Tn > 0) {  // potential conflict (treeobs.cpp:378-465 / observations.py:329-367)
            const int pt = CUTILS ? (int)((float)tot_dist * tpc_f) : (int)((double)tot_dist * tpc_d);
            if (pt < X.Tn && tot_dist < X.Tn) {
                const int r = cell / W, c = cell - r * W;
                const int key = c * W + r;
                const int lo = X.csr_head[key], hi = X.csr_head[key + 1];
                if (hi > lo) {
                    const int pre = max(pt - 1, 0), post = min(pt + 1, X.Tn - 1);
                    int sel = -1;
                    for (int k = 0; k < 3 && sel < 0; k++) {  // some OTHER agent predicted on this key at that time
                        const int ts = k == 0 ? pt : (k == 1 ? pre : post);
                        for (int e = lo; e < hi; e++) {
                            const uint32_t it = X.csr_items[e];
                            const int a = (int)(it >> 12);
                            if (a != handle && waypoint_at<CUTILS>(X, a, ts) == (int)((it >> 2) & 1023u)) { sel = ts; break; }
                        }
                    }
                    if (sel >= 0) {
                        for (int e = lo; e < hi; e++) {  // every agent (self included) predicted on this key at `sel`
                            const uint32_t it = X.csr_items[e];
                            const int a = (int)(it >> 12);
                            if (waypoint_at<CUTILS>(X, a, sel) != (int)((it >> 2) & 1023u)) continue;
                            uint32_t cd = it & 3u;
                            if (CUTILS && sel != pt)  // cutils indexes predicted_dir with predicted_time (treeobs.cpp:429-433,449-453)
                                cd = X.path[(size_t)a * X.pcap + waypoint_at<CUTILS>(X, a, pt)] & 3u;
                            if (d != cd && ((bits >> (3u - ((cd + 2u) & 3u))) & 1u) && (double)tot_dist < pot_conflict)
                                pot_conflict = tot_dist;
                            if (X.a_state[a] == ST_DONE && (double)tot_dist < pot_conflict) pot_conflict = tot_dist;
                        }
                    }
                }
            }
        }
        if (!CUTILS && ((X.cell_target[cell >> 5] >> (cell & 31)) & 1u) && cell != target) {
            if ((double)tot_dist < other_target) other_target = tot_dist;  // cutils never fills the map (treeobs.cpp:72)
        }
        if (cell == target && (double)tot_dist < own_target) own_target = tot_dist;
        if (visit == stop_at_visit) { o.is_terminal = true; break; }  // (cell, dir) already visited: cycle
        if (cell == target) { o.is_target = true; break; }
        if (crossing) total = 2;
        const int num = __popc(bits);
        if (total > 2 && 2 > num && (double)tot_dist < unusable) unusable = tot_dist;
        if (num == 1) {
            if (total == 1) { o.is_dead_end = true; break; }
            d = first_dir(bits);
            cell = step_cell(cell, d, W);
            tot_dist += 1;
            if (++visit >= max_visits && stop_at_visit < 0) { o.cycle_suspect = true; return; }
        } else if (num > 0) {
            o.is_switch = true;
            break;
        } else {
            o.zero_transition = true;  // treeobs.cpp:529-535 throws; observations.py:420-425 treats it as terminal
            o.is_terminal = true;
            break;
        }
    }
    const uint16_t dv = X.dm[((size_t)X.a_tslot[handle] * X.HW + cell) * 4 + d];
    const double dmv = dv == FL_INF16 ? INFINITY : (double)dv;
    double dist_next, dist_min;
    if (o.is_target) { dist_next = tot_dist; dist_min = 0; }
    else if (o.is_terminal) { dist_next = INFINITY; dist_min = dmv; }
    else { dist_next = tot_dist; dist_min = dmv; }
    o.f[0] = own_target; o.f[1] = other_target; o.f[2] = other_agent; o.f[3] = pot_conflict; o.f[4] = unusable;
    o.f[5] = dist_next; o.f[6] = dist_min; o.f[7] = same_dir; o.f[8] = opp_dir; o.f[9] = malfunctioning;
    o.f[10] = min_speed; o.f[11] = ready;
    o.end_cell = cell; o.end_dir = d; o.tot_dist = tot_dist;
}

template <bool CUTILS>
__device__ __forceinline__ void explore_branch_exact(const ObsCtx &X, int handle, int cell, uint32_t d, int tot_dist,
                                                     BranchOut &o) {
    explore_branch<CUTILS>(X, handle, cell, d, tot_dist, -1, o);
    if (o.cycle_suspect) {  // a walk longer than the number of (cell, dir) states repeats a state: replay up to the first repeat
        const int k = first_repeat_index(X, (cell << 2) | (int)d, X.a_target[handle]);
        explore_branch<CUTILS>(X, handle, cell, d, tot_dist, k, o);
    }
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

// RailEnvTransitions.transition_list (core/grid/rail_env_grid.py:28-38)
__constant__ uint16_t c_transition_list[11] = {0x0000, 0x8020, 0x9220, 0x8421, 0x9621, 0xCC33,
                                               0x5202, 0x2000, 0x4002, 0x1200, 0xC022};
// rotate_transition (tool.h:300-335): each nibble rotated right by k, then the word by 4k
__device__ __forceinline__ uint32_t rotate_transition(uint32_t cell, int k) {
    uint32_t v = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t nib = (cell >> ((3 - i) * 4)) & 15u;
        nib = ((nib >> k) | (nib << (4 - k))) & 15u;
        v |= nib << ((3 - i) * 4);
    }
    return ((v >> (4 * k)) | (v << (16 - 4 * k))) & 0xFFFFu;
}
__device__ __forceinline__ int road_type_of(uint32_t cell) {  // loader.cpp:122-161
    for (int rot = 0; rot < 4; rot++) {
        const uint32_t t = rot == 0 ? cell : rotate_transition(cell, rot);
        for (int k = 0; k < 11; k++)
            if (c_transition_list[k] == t) return k;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------- kernel
// MODE 0 = flatland_cutils outputs, MODE 1 = upstream dense tree.
struct ObsArgs {
    int max_nodes, pred_depth, max_depth;
    float *attr, *forest;
    int32_t *adjacency, *node_order, *edge_order;
    uint8_t *valid;
    double *props;
    double *tree_out;
    int n_tree_nodes;
};

template <int MODE>
__global__ __launch_bounds__(OBS_NT) void k_obs(FlDev d, FlObsScratch S, ObsArgs P) {
    constexpr bool CUTILS = MODE == 0;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int A = d.A, H = d.H, W = d.W, HW = H * W;
    const int K = (W - 1) * W + H;
    const int lane = tid & 63, wave = tid >> 6;

    extern __shared__ __align__(16) unsigned char lds[];
    size_t off = 0;
    auto carve = [&](size_t bytes) { void *p = lds + off; off += (bytes + 15) & ~(size_t)15; return p; };
    uint16_t *grid = (uint16_t *)carve((size_t)HW * 2);
    int16_t *cell_agent = (int16_t *)carve((size_t)HW * 2);
    uint16_t *cell_ready = (uint16_t *)carve((size_t)HW * 2);
    uint32_t *cell_target = (uint32_t *)carve((size_t)((HW + 31) / 32) * 4);
    double *a_speed = (double *)carve((size_t)A * 8);
    int *a_vpos = (int *)carve((size_t)A * 4);
    int *a_pos = (int *)carve((size_t)A * 4);
    int *a_tslot = (int *)carve((size_t)A * 4);
    int *a_target = (int *)carve((size_t)A * 4);
    uint16_t *a_malf = (uint16_t *)carve((size_t)A * 2);
    uint16_t *a_tpc = (uint16_t *)carve((size_t)A * 2);
    uint16_t *a_lp = (uint16_t *)carve((size_t)A * 2);
    uint8_t *a_dir = (uint8_t *)carve((size_t)A);
    uint8_t *a_state = (uint8_t *)carve((size_t)A);
    uint8_t *a_free = (uint8_t *)carve((size_t)A);
    uint8_t *a_dead = (uint8_t *)carve((size_t)A);
    int *misc = (int *)carve(64 * 4);
    int *wave_par = (int *)carve((size_t)OBS_WAVES * 64 * 4);  // per-wave parent[] scratch for the evaluation orders

    const uint16_t *ggrid = d.grid + (size_t)b * HW;
    const int T = d.T[b], tnow = d.t[b];

    // ---- phase 0: stage the rail bitmap, clear the per-cell maps, per-agent snapshot into LDS
    for (int c = tid; c < HW; c += nt) { grid[c] = ggrid[c]; cell_agent[c] = -1; cell_ready[c] = 0; }
    for (int c = tid; c < (HW + 31) / 32; c += nt) cell_target[c] = 0;
    if (tid < 64) misc[tid] = 0;
    __syncthreads();
    for (int i = tid; i < A; i += nt) {
        const int g = b * A + i;
        const uint32_t pk = d.pk[g], spk = d.spk[g];
        const uint32_t state = PK_STATE(pk);
        const int pos = d.pos[g], init_pos = d.init_pos[g], target = d.target[g];
        const double speed = d.speed[g];
        a_pos[i] = pos;
        a_vpos[i] = is_off_map(state) ? init_pos : (is_on_map(state) ? pos : target);  // loader.cpp:74-82
        a_dir[i] = (uint8_t)PK_DIR(pk);
        a_state[i] = (uint8_t)state;
        a_dead[i] = (uint8_t)PK_DEADLOCK(pk);
        a_malf[i] = (uint16_t)(d.malf[g] & 0xFFFFu);
        a_speed[i] = speed;
        a_tslot[i] = d.tslot[g];
        a_target[i] = target;
        a_tpc[i] = CUTILS ? (uint16_t)(int)(1.0f / (float)speed) : (uint16_t)(int)(1.0 / speed);
        (void)spk;
    }
    __syncthreads();
    // location_has_agent* (treeobs.cpp:74-81): the last (highest) handle on a cell wins; ready-to-depart counts (:82-91)
    for (int i = tid; i < A; i += nt) {
        const uint32_t state = a_state[i];
        if (!is_off_map(state) && a_pos[i] >= 0) {
            // 16-bit atomic max emulated on the containing 32-bit word
            const int c = a_pos[i];
            unsigned int *wptr = (unsigned int *)(cell_agent) + (c >> 1);
            const int sh = (c & 1) * 16;
            unsigned int old = *wptr, assumed;
            do {
                assumed = old;
                const int16_t cur = (int16_t)((assumed >> sh) & 0xFFFFu);
                if (cur >= (int16_t)i) break;
                const unsigned int nw = (assumed & ~(0xFFFFu << sh)) | (((unsigned int)(uint16_t)i) << sh);
                old = atomicCAS(wptr, assumed, nw);
            } while (old != assumed);
        }
        if (is_off_map(state)) {
            const int c = d.init_pos[b * A + i];
            atomicAdd((unsigned int *)cell_ready + (c >> 1), 1u << ((c & 1) * 16));
        }
        if (!CUTILS) atomicOr(&cell_target[a_target[i] >> 5], 1u << (a_target[i] & 31));
    }
    __syncthreads();

    ObsCtx X;
    X.A = A; X.H = H; X.W = W; X.HW = HW; X.K = K;
    X.grid = grid; X.cell_agent = cell_agent; X.cell_ready = cell_ready; X.cell_target = cell_target;
    X.a_vpos = a_vpos; X.a_dir = a_dir; X.a_state = a_state; X.a_malf = a_malf; X.a_speed = a_speed;
    X.a_tpc = a_tpc; X.a_lp = a_lp; X.a_tslot = a_tslot; X.a_target = a_target;
    X.pcap = S.pred_cap;
    X.path = S.path + (size_t)b * A * S.pred_cap;
    int *csr_head = S.cell_head + (size_t)b * (S.keys + 1);
    int *csr_cursor = S.cell_cursor + (size_t)b * (S.keys + 1);
    uint32_t *csr_items = S.cell_items + (size_t)b * A * S.pred_cap;
    X.csr_head = csr_head; X.csr_items = csr_items;
    X.Tn = P.pred_depth >= 0 ? P.pred_depth + 1 : 0;
    X.dm = d.dm + (size_t)b * d.Umax * HW * 4;

    // ---- phase 1 (cutils only): deadlock flags, valid actions, attribute rows
    if (CUTILS) {
        // DeadlockChecker (deadlock_checker.cpp:11-110) as a least fixpoint: an active agent is "free" when one of
        // its exits leads to an empty cell or to a free, not yet deadlocked agent (or it has no exit at all);
        // every other active agent becomes (and stays) deadlocked.  Equivalent to the reference's DFS + _fix_deps.
        for (int i = tid; i < A; i += nt) {
            bool fr = false;
            if (is_on_map(a_state[i]) && !a_dead[i]) {
                const uint32_t bits = nibble(grid[a_pos[i]], a_dir[i]);
                if (bits == 0) fr = true;
                const int r = a_pos[i] / W, c = a_pos[i] - r * W;
                for (uint32_t m = 0; m < 4 && !fr; m++) {
                    if (!((bits >> (3 - m)) & 1)) continue;
                    const int nr = r + (m == 0 ? -1 : m == 2 ? 1 : 0), nc = c + (m == 1 ? 1 : m == 3 ? -1 : 0);
                    if (nr < 0 || nc < 0 || nr >= H || nc >= W || cell_agent[nr * W + nc] < 0) fr = true;
                }
            }
            a_free[i] = fr;
        }
        __syncthreads();
        while (true) {
            for (int i = tid; i < A; i += nt) {
                if (is_on_map(a_state[i]) && !a_dead[i] && !a_free[i]) {
                    const uint32_t bits = nibble(grid[a_pos[i]], a_dir[i]);
                    bool fr = false;
                    for (uint32_t m = 0; m < 4 && !fr; m++) {
                        if (!((bits >> (3 - m)) & 1)) continue;
                        const int opp = cell_agent[step_cell(a_pos[i], m, W)];
                        if (opp >= 0 && !a_dead[opp] && a_free[opp]) fr = true;
                    }
                    if (fr) { a_free[i] = 1; misc[0] = 1; }
                }
            }
            __syncthreads();
            const int ch = misc[0];
            __syncthreads();
            if (!ch) break;
            if (tid == 0) misc[0] = 0;
            __syncthreads();
        }
        for (int i = tid; i < A; i += nt) {
            const int g = b * A + i;
            const uint32_t state = a_state[i];
            if (is_on_map(state) && !a_dead[i] && !a_free[i]) {
                a_dead[i] = 1;
                d.pk[g] |= (1u << 18);
            }
            const uint32_t pk = d.pk[g], spk = d.spk[g];
            const int pos = a_pos[i];
            const uint32_t dir = a_dir[i];
            const uint32_t scount = PK_SCOUNT(pk), max_count = SPK_MAX_COUNT(spk), init_dir = SPK_INIT_DIR(spk);
            const uint32_t old_dir = PK_OLD_DIR(pk) == 4 ? dir : PK_OLD_DIR(pk);
            // update_dist_target (loader.cpp:163-179)
            const size_t dmb = (size_t)a_tslot[i] * HW;
            const uint16_t dv_init = X.dm[(dmb + d.init_pos[g]) * 4 + init_dir];
            const float init_dist = dv_init == FL_INF16 ? INFINITY : (float)dv_init;
            float dist_target;
            if (state == ST_DONE) dist_target = 0;
            else if (is_off_map(state)) dist_target = init_dist;
            else {
                const uint16_t dv = X.dm[(dmb + pos) * 4 + dir];
                dist_target = dv == FL_INF16 ? INFINITY : (float)dv;
            }
            // valid-action mask (loader.cpp:273-312)
            uint32_t va = 0;
            const uint32_t cell = pos >= 0 ? grid[pos] : 0;
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
                            if (__popc((uint32_t)grid[step_cell(pos, nd, W)]) > 2) has_branch = true;
                        }
                    }
                    if (__popc(cell) > 2 || (cnt == 1 && has_branch)) va |= 1u << ACT_STOP;
                } else va |= 1u << ACT_NOTHING;
            } else if (state == ST_READY) va = (1u << ACT_FORWARD) | (1u << ACT_STOP);
            else va = 1u << ACT_NOTHING;
            uint8_t *vout = P.valid + (size_t)g * 5;
            for (int a = 0; a < 5; a++) vout[a] = (va >> a) & 1;
            if (P.props) {
                P.props[(size_t)g * 3 + 0] = (double)dist_target;
                P.props[(size_t)g * 3 + 1] = (double)a_dead[i];
                P.props[(size_t)g * 3 + 2] = (double)(state == ST_READY);
            }
            // AgentAttrParser::get_features (feature_parser.cpp:3-98)
            float *o = P.attr + (size_t)g * FL_CUTILS_ATTR;
            int n = 0;
            const int road_type = pos >= 0 ? road_type_of(cell) : 0;
            const uint32_t malfw = d.malf[g];
            const int malf01 = (malfw & 0xFFFFu) != 0, nmalf01 = (malfw >> 16) != 0;
            for (int k = 0; k < 7; k++) o[n++] = (k == (int)state) ? 1.0f : 0.0f;
            for (int k = 0; k < 11; k++) o[n++] = (k == road_type) ? 1.0f : 0.0f;
            for (int k = 0; k < 10; k++) o[n++] = (k == nmalf01) ? 1.0f : 0.0f;
            for (int k = 0; k < 4; k++) o[n++] = (k == (int)init_dir) ? 1.0f : 0.0f;
            for (int k = 0; k < 4; k++) o[n++] = (k == (int)dir) ? 1.0f : 0.0f;
            for (int k = 0; k < 4; k++) o[n++] = (k == (int)old_dir) ? 1.0f : 0.0f;
            o[n++] = (float)(state == ST_MOVING);
            o[n++] = (float)a_dead[i];
            o[n++] = (float)PK_SIGMALF(pk);
            o[n++] = (float)(!malf01);
            o[n++] = (float)(scount == 0);
            o[n++] = (float)(scount == max_count);
            o[n++] = (float)(state == ST_MALF || state == ST_MALF_OFF);
            o[n++] = (float)is_off_map(state);
            o[n++] = (float)is_on_map(state);
            for (int k = 15; k >= 0; k--) o[n++] = (float)((cell >> k) & 1u);
            for (int a = 0; a < 5; a++) o[n++] = (float)((va >> a) & 1u);
            const float max_t = (float)T, max_dist_target = (float)((H + W) * 8);
            const float f_step = (float)tnow / max_t;
            const float f_latest = (float)d.latest[g] / max_t;
            const float f_before = f_latest - f_step;
            const float f_dist = isinf(dist_target) ? 8.0f : dist_target / max_dist_target;
            o[n++] = (float)i / (float)A;
            o[n++] = f_step;
            o[n++] = (float)d.earliest[g] / max_t;
            o[n++] = f_latest;
            o[n++] = (float)d.arrival[g] / max_t;
            o[n++] = f_before;
            o[n++] = f_dist;
            o[n++] = f_before < f_dist ? f_before : f_dist;
            o[n++] = (float)max_count / 10;
            o[n++] = (float)a_speed[i] / 1.0f;
            o[n++] = (float)scount / 10;
            o[n++] = (float)malf01 / 10;
            o[n++] = isinf(init_dist) ? 8.0f : init_dist / max_dist_target;
        }
    }

    // ---- phase 2: predicted paths + per-key CSR index of (agent, waypoint)
    if (X.Tn > 0) {
        for (int k = tid; k <= K; k += nt) csr_head[k] = 0;
        __syncthreads();
        const int pred_depth = P.pred_depth;
        for (int i = tid; i < A; i += nt) {
            uint32_t *path = S.path + ((size_t)b * A + i) * S.pred_cap;
            int cell = a_vpos[i];
            uint32_t dd = a_dir[i];
            const int target = a_target[i];
            const size_t dmb = (size_t)a_tslot[i] * HW;
            int n = 0;
            bool none = false;
            if (cell == target) {  // holds at its position (DONE agents): predictions.cpp:208-214
                path[n++] = ((uint32_t)cell << 2) | dd;
            } else {
                uint32_t distance = 0x10000u;  // +inf
                int depth = 0;
                // cutils walks max_depth iterations and stops where nothing is strictly closer (predictions.cpp:107-133);
                // upstream stops at the target (rail_env_shortest_paths.py:245-265)
                while (depth < pred_depth && (CUTILS || cell != target)) {
                    const uint32_t g = grid[cell];
                    const uint32_t bits = nibble(g, dd);
                    int best = -1;
                    if (__popc(g) == 1) {  // is_dead_end: only the reverse exit
                        const uint32_t ex = (dd + 2u) & 3u;
                        if ((bits >> (3 - ex)) & 1) {
                            const uint32_t v = X.dm[(dmb + step_cell(cell, ex, W)) * 4 + ex];
                            if (v != FL_INF16 && v < distance) { best = (int)ex; distance = v; }
                        }
                    } else {
                        for (int j = -1; j <= 1; j++) {  // L, F, R: the reference's iteration order decides ties
                            const uint32_t nd = (dd + (uint32_t)(j + 4)) & 3u;
                            if ((bits >> (3 - nd)) & 1) {
                                const uint32_t v = X.dm[(dmb + step_cell(cell, nd, W)) * 4 + nd];
                                if (v != FL_INF16 && v < distance) { best = (int)nd; distance = v; }
                            }
                        }
                    }
                    path[n++] = ((uint32_t)cell << 2) | dd;
                    depth++;
                    if (best < 0) { none = true; break; }
                    cell = step_cell(cell, (uint32_t)best, W);
                    dd = (uint32_t)best;
                }
                if (CUTILS) { if (!none) path[n++] = ((uint32_t)cell << 2) | dd; }
                else {
                    if (none) { n = 1; }  // path None: the agent stands still (predictions.py:150-156)
                    else if (depth < pred_depth) path[n++] = ((uint32_t)cell << 2) | dd;
                }
            }
            // last waypoint that can be occupied within the horizon
            int lp = n - 1;
            const int tpc = a_tpc[i];
            const int horizon = CUTILS ? (X.Tn - 2) / tpc + 1 : (X.Tn - 1) / tpc;
            if (lp > horizon) lp = horizon;
            if (lp < 0) lp = 0;
            a_lp[i] = (uint16_t)lp;
            for (int k = 0; k <= lp; k++) {
                const int c = (int)(path[k] >> 2);
                const int r = c / W, col = c - r * W;
                atomicAdd(&csr_head[col * W + r], 1);
            }
        }
        __syncthreads();
        // exclusive scan over K + 1 keys: per-thread chunk sums, serial scan of the nt partial sums, rescan
        {
            int *partial = (int *)wave_par;  // reuse (>= OBS_NT ints)
            const int chunk = (K + 1 + nt - 1) / nt;
            const int lo = min(tid * chunk, K + 1), hi = min(lo + chunk, K + 1);
            int s = 0;
            for (int k = lo; k < hi; k++) s += csr_head[k];
            partial[tid] = s;
            __syncthreads();
            if (tid == 0) {
                int run = 0;
                for (int k = 0; k < nt; k++) { const int v = partial[k]; partial[k] = run; run += v; }
            }
            __syncthreads();
            int run = partial[tid];
            for (int k = lo; k < hi; k++) { const int v = csr_head[k]; csr_head[k] = run; csr_cursor[k] = run; run += v; }
        }
        __syncthreads();
        for (int i = tid; i < A; i += nt) {
            const uint32_t *path = S.path + ((size_t)b * A + i) * S.pred_cap;
            const int lp = a_lp[i];
            for (int k = 0; k <= lp; k++) {
                const uint32_t w = path[k];
                const int c = (int)(w >> 2);
                const int r = c / W, col = c - r * W;
                const int slot = atomicAdd(&csr_cursor[col * W + r], 1);
                csr_items[slot] = ((uint32_t)i << 12) | ((uint32_t)k << 2) | (w & 3u);
            }
        }
        __syncthreads();
    }

    // ---- phase 3: trees, one wavefront per agent
    const float max_dist = (float)T;
    for (int i = wave; i < A; i += OBS_WAVES) {
        const int g = b * A + i;
        const int vpos = a_vpos[i];
        const uint32_t dir = a_dir[i];
        const uint32_t rbits = nibble(grid[vpos], dir);
        uint32_t orientation = dir;
        if (__popc(rbits) == 1) orientation = first_dir(rbits);
        if (CUTILS) {
            const int N = P.max_nodes;
            float *F = P.forest + (size_t)g * N * 12;
            int32_t *ADJ = P.adjacency + (size_t)g * (N - 1) * 3;
            volatile int *par = wave_par + wave * 64;
            if (lane == 0) {  // root (treeobs.cpp:171-186)
                const uint32_t state = a_state[i];
                double root[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                uint16_t dv = FL_INF16;
                if (state == ST_DONE) dv = 0;
                else dv = X.dm[((size_t)a_tslot[i] * HW + (is_off_map(state) ? d.init_pos[g] : a_pos[i])) * 4 +
                               (is_off_map(state) ? SPK_INIT_DIR(d.spk[g]) : dir)];
                root[6] = dv == FL_INF16 ? INFINITY : (double)dv;
                root[9] = (double)((d.malf[g] >> 16) != 0);
                root[10] = (double)(float)a_speed[i];
                scale_and_store(root, max_dist, A, F);
            }
            if (lane < 64) par[lane] = -2;
            // level 1: three cells from the root (treeobs.cpp:205-222)
            int c_cell = -1, c_parent = 0, c_tot = 1, c_act = 0;
            uint32_t c_dir = 0;
            bool c_null = true;
            if (lane < 3) {
                c_act = lane - 1;
                c_dir = (orientation + (uint32_t)(c_act + 4)) & 3u;
                if ((rbits >> (3 - c_dir)) & 1) { c_cell = step_cell(vpos, c_dir, W); c_null = false; }
            }
            int n_cur = 3, node_base = 1;
            while (node_base < N && n_cur > 0) {
                const int m = min(n_cur, N - node_base);
                const bool mine = lane < m;
                const int idx_node = node_base + lane;
                // children descriptors this lane would push (treeobs.cpp:583-608)
                int ch_cell[3] = {-1, -1, -1};
                uint32_t ch_dir[3] = {0, 0, 0};
                int ch_tot = 0;
                bool explored = false;
                if (mine) {
                    if (c_null) {
                        const double nn[12] = {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, -1, -1, -1, -1, -1};
                        scale_and_store(nn, max_dist, A, F + (size_t)idx_node * 12);
                    } else {
                        BranchOut br;
                        explore_branch_exact<true>(X, i, c_cell, c_dir, c_tot, br);
                        if (br.zero_transition) atomicCAS(&d.err[b], 0, FL_ERR_ZERO_TRANSITION);
                        scale_and_store(br.f, max_dist, A, F + (size_t)idx_node * 12);
                        explored = true;
                        ch_tot = br.tot_dist + 1;
                        const uint32_t pbits = nibble(grid[br.end_cell], br.end_dir);
                        for (int k = 0; k < 3; k++) {
                            const uint32_t bd = (br.end_dir + (uint32_t)(k + 3)) & 3u, rev = (bd + 2u) & 3u;
                            ch_dir[k] = bd;
                            if (br.is_dead_end && ((pbits >> (3 - rev)) & 1)) { ch_cell[k] = step_cell(br.end_cell, rev, W); ch_dir[k] = rev; }
                            else if (br.is_switch && ((pbits >> (3 - bd)) & 1)) ch_cell[k] = step_cell(br.end_cell, bd, W);
                        }
                    }
                    int32_t *adj = ADJ + (size_t)(idx_node - 1) * 3;
                    adj[0] = c_parent; adj[1] = idx_node; adj[2] = c_act;
                    par[idx_node] = c_parent;
                }
                const uint64_t exp_mask = __ballot(explored);
                const int n_next = 3 * __popcll(exp_mask);
                // hand the children to the next level's lanes: lane j takes child j % 3 of the (j / 3)-th explored lane
                const int src_rank = lane / 3, which = lane - 3 * src_rank;
                const int src = (lane < n_next) ? kth_set_bit(exp_mask, src_rank) : 0;
                const int s_c0 = __shfl(ch_cell[0], src), s_c1 = __shfl(ch_cell[1], src), s_c2 = __shfl(ch_cell[2], src);
                const uint32_t s_d0 = __shfl(ch_dir[0], src), s_d1 = __shfl(ch_dir[1], src), s_d2 = __shfl(ch_dir[2], src);
                const int s_tot = __shfl(ch_tot, src);
                node_base += m;
                n_cur = n_next;
                if (lane < n_next) {
                    c_cell = which == 0 ? s_c0 : (which == 1 ? s_c1 : s_c2);
                    c_dir = which == 0 ? s_d0 : (which == 1 ? s_d1 : s_d2);
                    c_null = c_cell < 0;
                    c_parent = (node_base - m) + src;
                    c_tot = s_tot;
                    c_act = which - 1;
                }
            }
            // padding rows when the queue ran dry (treeobs.cpp:268-276, 245-249)
            for (int idx = node_base + lane; idx < N; idx += 64) {
                const double nn[12] = {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, -1, -1, -1, -1, -1};
                scale_and_store(nn, max_dist, A, F + (size_t)idx * 12);
                int32_t *adj = ADJ + (size_t)(idx - 1) * 3;
                adj[0] = adj[1] = adj[2] = -2;
            }
            // calculate_evaluation_orders (tool.h:468-524): order = height above the leaves; parents precede children
            // in BFS numbering, so one reverse sweep settles it
            int32_t *NO = P.node_order + (size_t)g * N, *EO = P.edge_order + (size_t)g * (N - 1);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                int h[FL_OBS_MAX_NODES];
                for (int k = 0; k < N; k++) h[k] = 0;
                for (int k = N - 1; k >= 1; k--) {
                    const int p = par[k];
                    if (p >= 0 && h[p] < h[k] + 1) h[p] = h[k] + 1;
                }
                const int n_real = node_base;  // real (non padding) nodes
                for (int k = 0; k < N; k++) NO[k] = k < n_real ? h[k] : -2;
                for (int k = 1; k < N; k++) EO[k - 1] = par[k] < 0 ? -2 : h[par[k]];
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            // upstream dense tree, DFS pre-order layout; level L is explored by 4^L lanes (observations.py:196-254, 464-494)
            const int D = P.max_depth, NN = P.n_tree_nodes;
            double *out = P.tree_out + (size_t)g * NN * 12;
            if (lane == 0) {
                const uint16_t dv = X.dm[((size_t)a_tslot[i] * HW + vpos) * 4 + dir];
                double root[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                root[6] = dv == FL_INF16 ? INFINITY : (double)dv;
                root[9] = (double)a_malf[i];
                root[10] = a_speed[i];
                for (int k = 0; k < 12; k++) out[k] = root[k];
            }
            // subtree sizes: sz[l] = nodes of a subtree rooted at depth l
            int sz[6];
            { int n = 0; for (int l = D; l >= 0; l--) { n = n * 4 + 1; sz[l] = n; } }
            // lane state for the level being explored
            int c_cell = -1, c_tot = 1, c_index = 0;
            uint32_t c_dir = 0;
            bool c_real = false;
            if (lane < 4) {
                const uint32_t bd = (orientation + (uint32_t)(lane + 3)) & 3u;
                c_index = 1 + lane * sz[1];
                if ((rbits >> (3 - bd)) & 1) { c_cell = step_cell(vpos, bd, W); c_dir = bd; c_real = true; }
            }
            int width = 4;
            for (int level = 1; level <= D; level++) {
                int ch_cell[4] = {-1, -1, -1, -1};
                uint32_t ch_dir[4] = {0, 0, 0, 0};
                int ch_tot = 0;
                if (lane < width && c_index >= 0) {
                    double *row = out + (size_t)c_index * 12;
                    if (c_real) {
                        BranchOut br;
                        explore_branch_exact<false>(X, i, c_cell, c_dir, c_tot, br);
                        for (int k = 0; k < 12; k++) row[k] = br.f[k];
                        ch_tot = br.tot_dist + 1;
                        const uint32_t pbits = nibble(grid[br.end_cell], br.end_dir);
                        for (int k = 0; k < 4; k++) {
                            const uint32_t bd = (br.end_dir + (uint32_t)(k + 3)) & 3u, rev = (bd + 2u) & 3u;
                            if (br.is_dead_end && ((pbits >> (3 - rev)) & 1)) { ch_cell[k] = step_cell(br.end_cell, rev, W); ch_dir[k] = rev; }
                            else if (br.is_switch && ((pbits >> (3 - bd)) & 1)) { ch_cell[k] = step_cell(br.end_cell, bd, W); ch_dir[k] = bd; }
                        }
                    } else {
                        // missing child: the whole subtree is -inf
                        const int n = sz[level] * 12;
                        for (int k = 0; k < n; k++) row[k] = -INFINITY;
                    }
                }
                if (level == D) break;
                // children of lane p go to lanes 4p .. 4p+3 of the next level (all lanes take part in the shuffles)
                const int src = lane >> 2, which = lane & 3;
                const int p_index = __shfl(c_index, src);
                const bool p_real = __shfl((int)c_real, src) != 0;
                const int s0 = __shfl(ch_cell[0], src), s1 = __shfl(ch_cell[1], src), s2 = __shfl(ch_cell[2], src), s3 = __shfl(ch_cell[3], src);
                const uint32_t e0 = __shfl(ch_dir[0], src), e1 = __shfl(ch_dir[1], src), e2 = __shfl(ch_dir[2], src), e3 = __shfl(ch_dir[3], src);
                const int s_tot = __shfl(ch_tot, src);
                width *= 4;
                c_index = -1;  // -1: nothing to write (covered by an ancestor's -inf fill, or lane unused)
                c_real = false;
                if (lane < width && p_index >= 0 && p_real) {
                    c_cell = which == 0 ? s0 : which == 1 ? s1 : which == 2 ? s2 : s3;
                    c_dir = which == 0 ? e0 : which == 1 ? e1 : which == 2 ? e2 : e3;
                    c_tot = s_tot;
                    c_index = p_index + 1 + which * sz[level + 1];
                    c_real = c_cell >= 0;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- host side
int fl_obs_alloc(FlObsScratch &o, const FlDev &d, hipStream_t s, std::vector<void *> &allocs) {
    o.pred_cap = FL_OBS_MAX_PRED + 2;
    o.keys = (d.W - 1) * d.W + d.H;
    const size_t BA = (size_t)d.B * d.A;
    void *p = nullptr;
    if (hipMalloc(&p, BA * o.pred_cap * 4) != hipSuccess) return FL_ERR_HIP;
    o.path = (uint32_t *)p; allocs.push_back(p);
    if (hipMalloc(&p, BA * o.pred_cap * 4) != hipSuccess) return FL_ERR_HIP;
    o.cell_items = (uint32_t *)p; allocs.push_back(p);
    if (hipMalloc(&p, (size_t)d.B * (o.keys + 1) * 4) != hipSuccess) return FL_ERR_HIP;
    o.cell_head = (int *)p; allocs.push_back(p);
    if (hipMalloc(&p, (size_t)d.B * (o.keys + 1) * 4) != hipSuccess) return FL_ERR_HIP;
    o.cell_cursor = (int *)p; allocs.push_back(p);
    (void)s;
    return FL_OK;
}

void fl_obs_reset(FlObsScratch &o, const FlDev &d, const uint8_t *mask_dev, hipStream_t s) {
    // the sticky deadlock flags live in pk and are cleared by the agent reset kernel (flatland_cutils rebuilds its
    // DeadlockChecker in TreeObsForRailEnv::reset(), treeobs.cpp:22-28 / loader.cpp:207-219)
    (void)o; (void)d; (void)mask_dev; (void)s;
}

static size_t obs_lds_bytes(const FlDev &d) {
    const size_t HW = (size_t)d.H * d.W, A = d.A;
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    return al(HW * 2) * 3 + al(((HW + 31) / 32) * 4) + al(A * 8) + al(A * 4) * 4 + al(A * 2) * 3 + al(A) * 4 + al(64 * 4) +
           al((size_t)OBS_WAVES * 64 * 4) + 64;
}

int fl_launch_obs_cutils(FlObsScratch &o, const FlDev &d, int max_nodes, int pred_depth, float *attr, float *forest,
                         int32_t *adjacency, int32_t *node_order, int32_t *edge_order, uint8_t *valid, double *props,
                         hipStream_t s) {
    const size_t lds = obs_lds_bytes(d);
    if (lds > 160 * 1024) return FL_ERR_ARG;
    if (d.A >= (1 << 20) || pred_depth + 2 > o.pred_cap) return FL_ERR_ARG;
    ObsArgs P;
    ObsArgs Z = {};
    P = Z;
    P.max_nodes = max_nodes; P.pred_depth = pred_depth; P.attr = attr; P.forest = forest; P.adjacency = adjacency;
    P.node_order = node_order; P.edge_order = edge_order; P.valid = valid; P.props = props;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)k_obs<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
        if (hipFuncSetAttribute((const void *)k_obs<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_obs<0>, dim3(d.B), dim3(OBS_NT), lds, s, d, o, P);
    return FL_OK;
}

int fl_launch_obs_tree(FlObsScratch &o, const FlDev &d, int max_depth, int pred_depth, double *out, hipStream_t s) {
    const size_t lds = obs_lds_bytes(d);
    if (lds > 160 * 1024) return FL_ERR_ARG;
    if (pred_depth + 2 > o.pred_cap) return FL_ERR_ARG;
    ObsArgs P;
    ObsArgs Z = {};
    P = Z;
    P.max_depth = max_depth; P.pred_depth = pred_depth; P.tree_out = out;
    int n = 0, p = 1;
    for (int k = 0; k <= max_depth; k++) { n += p; p *= 4; }
    P.n_tree_nodes = n;
    if (max_depth > 3) return FL_ERR_ARG;  // one lane per node of the deepest level: 4^3 = 64
    hipLaunchKernelGGL(k_obs<1>, dim3(d.B), dim3(OBS_NT), lds, s, d, o, P);
    return FL_OK;
}
