/*
 * fl_oracle.c -- CPU ORACLE, part 1: env container, numpy-legacy MT19937, DistanceMap BFS,
 * MotionCheck, RailEnv.step().  TEST INFRASTRUCTURE ONLY (see fl_oracle.h).
 * All file:line citations are relative to /root/reference/flatland-rl/flatland/.
 */
#include <stdarg.h>

#include "fl_oracle_internal.h"

static char g_err[256];
void orc_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *orc_last_error(void) { return g_err; }

/* ------------------------------------------------------------------ env container */
#define ALLOC_I(n) ((int *)calloc((size_t)(n), sizeof(int)))

OrcEnv *orc_create(int H, int W, int A) {
    OrcEnv *e = (OrcEnv *)calloc(1, sizeof(OrcEnv));
    e->H = H; e->W = W; e->A = A;
    e->grid = (uint16_t *)calloc((size_t)H * W, 2);
    e->init_r = ALLOC_I(A); e->init_c = ALLOC_I(A); e->init_dir = ALLOC_I(A);
    e->tgt_r = ALLOC_I(A); e->tgt_c = ALLOC_I(A); e->earliest = ALLOC_I(A); e->latest = ALLOC_I(A);
    e->max_count = ALLOC_I(A);
    e->speed = (double *)calloc((size_t)A, sizeof(double));
    e->r = ALLOC_I(A); e->c = ALLOC_I(A); e->dir = ALLOC_I(A); e->state = ALLOC_I(A); e->prev_state = ALLOC_I(A);
    e->saved = ALLOC_I(A); e->scount = ALLOC_I(A); e->malf = ALLOC_I(A); e->nmalf = ALLOC_I(A);
    e->old_r = ALLOC_I(A); e->old_c = ALLOC_I(A); e->old_dir = ALLOC_I(A); e->arrival = ALLOC_I(A);
    e->done = (uint8_t *)calloc((size_t)A, 1);
    e->sig_in_malf = (uint8_t *)calloc((size_t)A, 1);
    e->deadlocked = (uint8_t *)calloc((size_t)A, 1);
    e->tslot = ALLOC_I(A); e->ut_r = ALLOC_I(A); e->ut_c = ALLOC_I(A);
    e->dm = NULL;
    return e;
}

void orc_destroy(OrcEnv *e) {
    if (!e) return;
    free(e->grid); free(e->init_r); free(e->init_c); free(e->init_dir); free(e->tgt_r); free(e->tgt_c);
    free(e->earliest); free(e->latest); free(e->max_count); free(e->speed);
    free(e->r); free(e->c); free(e->dir); free(e->state); free(e->prev_state); free(e->saved); free(e->scount);
    free(e->malf); free(e->nmalf); free(e->old_r); free(e->old_c); free(e->old_dir); free(e->arrival);
    free(e->done); free(e->sig_in_malf); free(e->deadlocked); free(e->tslot); free(e->ut_r); free(e->ut_c);
    free(e->dm);
    free(e);
}

/* ------------------------------------------------------------------ numpy legacy RandomState (MT19937)
 * third-party: numpy (randomkit / _legacy); restated from the published algorithm, pinned by the
 * malfunction trajectories in tests/golden (num_malfunctions, malfunction_down_counter per step). */
static void mt_generate(uint32_t *mt) {
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
    int kk;
    uint32_t y;
    for (kk = 0; kk < 624 - 397; kk++) {
        y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1) ? MATRIX_A : 0);
    }
    for (; kk < 623; kk++) {
        y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1) ? MATRIX_A : 0);
    }
    y = (mt[623] & UPPER) | (mt[0] & LOWER);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1) ? MATRIX_A : 0);
}

static uint32_t mt_next32(uint32_t *mt, int *pos) {
    uint32_t y;
    if (*pos >= 624) {
        mt_generate(mt);
        *pos = 0;
    }
    y = mt[(*pos)++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* RandomState.rand(): 53-bit double from two words */
static uint64_t mt_rand53(uint32_t *mt, int *pos) {
    uint64_t a = mt_next32(mt, pos) >> 5, b = mt_next32(mt, pos) >> 6;
    return (a << 26) | b;
}

/* RandomState.randint(lo, hi) legacy masked rejection on 32-bit words (hi exclusive) */
static int64_t mt_randint(uint32_t *mt, int *pos, int64_t lo, int64_t hi) {
    uint64_t rng = (uint64_t)(hi - 1 - lo), mask, v;
    if (rng == 0) return lo;
    mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    do {
        v = mt_next32(mt, pos) & mask;
    } while (v > rng);
    return lo + (int64_t)v;
}

void orc_mt_seed_by_array(uint32_t *mt, int32_t *mt_pos, const uint32_t *init_key, int key_length) {
    int i, j, k;
    mt[0] = 19650218u;
    for (i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    i = 1; j = 0;
    k = (624 > key_length ? 624 : key_length);
    for (; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + init_key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
        if (j >= key_length) j = 0;
    }
    for (k = 623; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
    *mt_pos = 624;
}

double orc_mt_rand(uint32_t *mt_key, int32_t *mt_pos) {
    int p = *mt_pos;
    uint64_t u = mt_rand53(mt_key, &p);
    *mt_pos = p;
    return (double)u / 9007199254740992.0;
}

void orc_set_rng(OrcEnv *e, const uint32_t *mt_key, int mt_pos) {
    memcpy(e->mt, mt_key, sizeof e->mt);
    e->mti = mt_pos;
}
void orc_get_rng(OrcEnv *e, uint32_t *mt_key, int32_t *mt_pos) {
    memcpy(mt_key, e->mt, sizeof e->mt);
    *mt_pos = e->mti;
}

/* ------------------------------------------------------------------ DistanceMap (envs/distance_map.py) */
typedef struct { int r, c, o, d; } BfsNode;

/* _get_and_update_neighbors (distance_map.py:121-160) */
static int dm_neighbors(const uint16_t *grid, int H, int W, uint16_t *out, int pr, int pc, int cur_dist, int enforce,
                        BfsNode *q, int qt) {
    int lo = 0, hi = 4, nd;
    if (enforce >= 0) { lo = (enforce + 2) % 4; hi = lo + 1; }
    for (nd = lo; nd < hi; nd++) {
        int nr = pr + ORC_DR[nd], nc = pc + ORC_DC[nd], o;
        if (nr >= 0 && nr < H && nc >= 0 && nc < W) {
            int desired = (nd + 2) % 4;
            for (o = 0; o < 4; o++) {
                if (orc_tbit(grid[nr * W + nc], o, desired)) {
                    uint16_t *slot = &out[(nr * W + nc) * 4 + o];
                    int nd_ = cur_dist + 1;
                    int newd = (*slot == 0xFFFF) ? nd_ : (*slot < nd_ ? *slot : nd_);
                    q[qt].r = nr; q[qt].c = nc; q[qt].o = o; q[qt].d = newd;
                    qt++;
                    *slot = (uint16_t)newd;
                }
            }
        }
    }
    return qt;
}

/* _distance_map_walker (distance_map.py:81-119) */
void orc_distance_map_bfs(const uint16_t *grid, int H, int W, int tr, int tc, uint16_t *out) {
    size_t n = (size_t)H * W * 4;
    uint8_t *visited = (uint8_t *)calloc(n, 1);
    /* every state is expanded at most once and each expansion appends <= 4 nodes; +16 for the seed call */
    BfsNode *q = (BfsNode *)malloc(sizeof(BfsNode) * (n * 4 + 32));
    int qh = 0, qt = 0, o;
    memset(out, 0xFF, n * 2);
    for (o = 0; o < 4; o++) { out[(tr * W + tc) * 4 + o] = 0; visited[(tr * W + tc) * 4 + o] = 1; }
    qt = dm_neighbors(grid, H, W, out, tr, tc, 0, -1, q, qt);
    while (qh < qt) {
        BfsNode nd = q[qh++];
        size_t id = ((size_t)nd.r * W + nd.c) * 4 + nd.o;
        if (!visited[id]) {
            visited[id] = 1;
            qt = dm_neighbors(grid, H, W, out, nd.r, nd.c, nd.d, nd.o, q, qt);
        }
    }
    free(q);
    free(visited);
}

/* DistanceMap._compute (distance_map.py:57-79): one BFS per unique target, agents with the same target share it */
static void compute_distance_maps(OrcEnv *e) {
    int i, u;
    e->U = 0;
    for (i = 0; i < e->A; i++) {
        for (u = 0; u < e->U; u++)
            if (e->ut_r[u] == e->tgt_r[i] && e->ut_c[u] == e->tgt_c[i]) break;
        if (u == e->U) { e->ut_r[u] = e->tgt_r[i]; e->ut_c[u] = e->tgt_c[i]; e->U++; }
        e->tslot[i] = u;
    }
    free(e->dm);
    e->dm = (uint16_t *)malloc((size_t)e->U * e->H * e->W * 4 * 2);
    for (u = 0; u < e->U; u++)
        orc_distance_map_bfs(e->grid, e->H, e->W, e->ut_r[u], e->ut_c[u], e->dm + (size_t)u * e->H * e->W * 4);
}

int orc_num_targets(const OrcEnv *e) { return e->U; }
void orc_get_distance_map(const OrcEnv *e, uint16_t *dm, int32_t *target_slot) {
    memcpy(dm, e->dm, (size_t)e->U * e->H * e->W * 4 * 2);
    memcpy(target_slot, e->tslot, sizeof(int32_t) * e->A);
}

/* ------------------------------------------------------------------ load / reset */
int orc_load(OrcEnv *e, const uint16_t *grid, const int32_t *init_pos, const int32_t *init_dir,
             const int32_t *target, const double *speed, const int32_t *earliest, const int32_t *latest,
             int T, uint64_t malf_threshold, int malf_min, int malf_max, const uint32_t *mt_key, int mt_pos) {
    int i;
    memcpy(e->grid, grid, (size_t)e->H * e->W * 2);
    for (i = 0; i < e->A; i++) {
        e->init_r[i] = init_pos[2 * i]; e->init_c[i] = init_pos[2 * i + 1]; e->init_dir[i] = init_dir[i];
        e->tgt_r[i] = target[2 * i]; e->tgt_c[i] = target[2 * i + 1];
        e->speed[i] = speed[i];
        e->earliest[i] = earliest[i]; e->latest[i] = latest[i];
        /* SpeedCounter.max_count (step_utils/speed_counter.py:39-41): int(1/speed) - 1 */
        e->max_count[i] = (int)(1.0 / speed[i]) - 1;
        e->arrival[i] = -1; /* fresh EnvAgent from_line (agent_utils.py:150-171) */
    }
    e->T = T;
    e->malf_threshold = malf_threshold; e->malf_min = malf_min; e->malf_max = malf_max;
    orc_set_rng(e, mt_key, mt_pos);
    compute_distance_maps(e);
    orc_reset(e);
    return ORC_OK;
}

/* EnvAgent.reset (agent_utils.py:90-105) for all agents; rail_env.py:335-344 */
void orc_reset(OrcEnv *e) {
    int i;
    for (i = 0; i < e->A; i++) {
        e->r[i] = e->c[i] = -1;
        e->dir[i] = e->init_dir[i];
        e->old_r[i] = e->old_c[i] = -1; e->old_dir[i] = -1;
        e->malf[i] = 0; e->nmalf[i] = 0;
        e->saved[i] = 0; e->scount[i] = 0;
        e->state[i] = ST_WAITING; e->prev_state[i] = -1;
        e->done[i] = 0; e->sig_in_malf[i] = 0;
        /* note: arrival_time is NOT cleared by EnvAgent.reset(); orc_load() sets it to None for fresh agents */
    }
    e->t = 0;
    e->done_all = 0;
    orc_obs_cutils_reset(e);
}

int orc_elapsed(const OrcEnv *e) { return e->t; }

void orc_get_state(const OrcEnv *e, int32_t *out) {
    int i;
    for (i = 0; i < e->A; i++) {
        int32_t *o = out + (size_t)i * ORC_STATE_COLS;
        o[0] = e->r[i]; o[1] = e->c[i]; o[2] = e->dir[i]; o[3] = e->state[i]; o[4] = e->malf[i]; o[5] = e->nmalf[i];
        o[6] = e->scount[i]; o[7] = e->saved[i]; o[8] = e->arrival[i]; o[9] = e->old_r[i]; o[10] = e->old_c[i];
        o[11] = e->old_dir[i];
    }
}

/* ------------------------------------------------------------------ MotionCheck (envs/agent_chains.py)
 * The reference builds a networkx DiGraph whose NODES are cells (off-map agents get a private virtual node,
 * :28-32) and whose edges are cur->next.  Restated on node ids without a graph library:
 *   stops  = nodes with a self loop                                  (find_stops2 :59-63)
 *   swaps  = nodes on a 2-cycle u->v, v->u, u != v                    (find_swaps :107-117)
 *   losers = for every node v with >= 2 distinct predecessor nodes, all predecessors except the one whose
 *            "agent" attribute (last agent added on that node, :34) is minimal   (find_conflicts :176-202)
 *   blocked = every node from which a stop / swap / loser node is reachable     (block_preds :125-149,
 *             find_stop_preds :65-105)
 *   check_motion(i) = node(cur[i]) not blocked (a node with a self loop is a stop, hence blocked)  (:204-236)
 * Two agents may share a node (rail_env.py:599-602 places an agent without an occupancy check); the
 * node-level statement above covers that case too.
 */
void orc_motion_check(int n, const int32_t *cur, const int32_t *nxt, uint8_t *can_move) {
    int i, j, changed;
    uint8_t *blocked = (uint8_t *)calloc((size_t)n, 1); /* per agent = its node's flag, kept consistent below */
    int *node_agent = (int *)malloc(sizeof(int) * n);   /* "agent" attr of node cur[i]: last (max) index on it */
    for (i = 0; i < n; i++) {
        node_agent[i] = i;
        for (j = 0; j < n; j++)
            if (cur[j] == cur[i] && j > node_agent[i]) node_agent[i] = j;
    }
    for (i = 0; i < n; i++) {
        if (nxt[i] == cur[i]) blocked[i] = 1; /* stop */
        else {
            int winner = -1;
            for (j = 0; j < n; j++) {
                /* swap: some agent on my next node heads for my node */
                if (cur[j] == nxt[i] && nxt[j] == cur[i]) blocked[i] = 1;
                /* contention on nxt[i]: predecessor nodes are keyed by their agent attribute */
                if (nxt[j] == nxt[i] && (winner < 0 || node_agent[j] < winner)) winner = node_agent[j];
            }
            if (winner != node_agent[i]) blocked[i] = 1;
        }
    }
    do { /* propagate to predecessors (and between agents sharing a node) until fixpoint */
        changed = 0;
        for (i = 0; i < n; i++) {
            if (blocked[i]) continue;
            for (j = 0; j < n; j++) {
                if (blocked[j] && (cur[j] == nxt[i] || cur[j] == cur[i])) { blocked[i] = 1; changed = 1; break; }
            }
        }
    } while (changed);
    for (i = 0; i < n; i++) can_move[i] = !blocked[i];
    free(blocked);
    free(node_agent);
}

/* ------------------------------------------------------------------ action preprocessing */
/* transition_utils.check_action (step_utils/transition_utils.py:6-44): returns new_direction, *tv in {-1 None,0,1} */
static int check_action(const OrcEnv *e, int action, int r, int c, int dir, int *tv) {
    int bits = orc_nibble(orc_cell(e, r, c), dir);
    int k = orc_popcount((unsigned)bits);
    int nd = dir;
    *tv = -1;
    if (action == ACT_LEFT) { nd = dir - 1; if (k <= 1) *tv = 0; }
    else if (action == ACT_RIGHT) { nd = dir + 1; if (k <= 1) *tv = 0; }
    nd = ((nd % 4) + 4) % 4;
    if (action == ACT_FORWARD && k == 1) {
        /* fast_argmax (:84-91): first set transition in N,E,S,W order */
        int m;
        for (m = 0; m < 3; m++) if ((bits >> (3 - m)) & 1) break;
        nd = m;
        *tv = 1;
    }
    return nd;
}

/* check_valid_action / check_action_on_agent (transition_utils.py:47-82) */
static int check_valid_action(const OrcEnv *e, int action, int r, int c, int dir) {
    int tv, nd = check_action(e, action, r, c, dir, &tv);
    int nr = r + ORC_DR[nd], nc = c + ORC_DC[nd];
    int cell_ok = orc_in_bounds(e, nr, nc) && orc_cell(e, nr, nc) > 0;
    if (tv < 0) tv = orc_tbit(orc_cell(e, r, c), dir, nd);
    return cell_ok && tv;
}

/* RailEnv.preprocess_action (envs/rail_env.py:425-446) + step_utils/action_preprocessing.py:7-59 */
static int preprocess_action(const OrcEnv *e, int i, int raw) {
    int a = raw, r, c, d;
    if (a < 0 || a > 4) a = ACT_NOTHING;                 /* process_illegal_action :7-11 */
    if (a == ACT_NOTHING) {                              /* process_do_nothing :14-21 */
        if (e->state[i] == ST_MOVING) a = ACT_FORWARD;
        else if (e->saved[i]) a = e->saved[i];
        else a = ACT_NOTHING;
    }
    if (e->state[i] == ST_WAITING) a = ACT_NOTHING;      /* preprocess_action_when_waiting :30-36 */
    if (e->r[i] < 0) { r = e->init_r[i]; c = e->init_c[i]; d = e->init_dir[i]; }   /* rail_env.py:436-438 */
    else { r = e->r[i]; c = e->c[i]; d = e->dir[i]; }
    if ((a == ACT_LEFT || a == ACT_RIGHT) && !check_valid_action(e, a, r, c, d)) a = ACT_FORWARD; /* :24-27,51-59 */
    if (a >= ACT_LEFT && a <= ACT_RIGHT && !check_valid_action(e, a, r, c, d)) a = ACT_STOP;     /* rail_env.py:443-444 */
    return a;
}

/* ------------------------------------------------------------------ end-of-episode reward */
/* get_shortest_paths (envs/rail_env_shortest_paths.py:203-274) for one agent with max_depth=None:
 * returns len(path) or 0 when the path is None. get_valid_move_actions_ (:17-71) enumerates the candidates. */
static int shortest_path_len(const OrcEnv *e, int i) {
    int r, c, d = e->dir[i], len = 0;
    double distance = INFINITY;
    int s = e->state[i];
    if (orc_is_off_map(s)) { r = e->init_r[i]; c = e->init_c[i]; }
    else if (orc_is_on_map(s)) { r = e->r[i]; c = e->c[i]; }
    else { r = e->tgt_r[i]; c = e->tgt_c[i]; }
    while (!(r == e->tgt_r[i] && c == e->tgt_c[i])) {
        uint16_t cell = orc_cell(e, r, c);
        int bits = orc_nibble(cell, d), k = orc_popcount((unsigned)bits);
        int cand[3], nc_ = 0, j, best = -1;
        if (orc_popcount(cell) == 1) {                       /* is_dead_end (core/transition_map.py:311-329) */
            int ex = (d + 2) % 4;
            if ((bits >> (3 - ex)) & 1) cand[nc_++] = ex;
        } else {
            (void)k;
            for (j = -1; j <= 1; j++) {
                int nd = (d + j + 4) % 4;
                if ((bits >> (3 - nd)) & 1) cand[nc_++] = nd;
            }
        }
        for (j = 0; j < nc_; j++) {
            int nd = cand[j];
            uint16_t v = orc_dm_at(e, i, r + ORC_DR[nd], c + ORC_DC[nd], nd);
            double dv = (v == 0xFFFF) ? INFINITY : (double)v;
            if (dv < distance) { best = nd; distance = dv; }
        }
        len++;
        if (best < 0) return 0; /* path None */
        r += ORC_DR[best]; c += ORC_DC[best]; d = best;
    }
    return len + 1;
}

/* EnvAgent.get_travel_time_on_shortest_path (envs/agent_utils.py:129-136) */
static int travel_time(const OrcEnv *e, int i) {
    int distance = shortest_path_len(e, i);
    return (int)ceil((double)distance / e->speed[i]);
}

/* RailEnv._handle_end_reward (rail_env.py:397-423) */
static int end_reward(const OrcEnv *e, int i) {
    if (e->state[i] == ST_DONE) {
        int v = e->latest[i] - e->arrival[i];
        return v < 0 ? v : 0;
    }
    if (orc_is_off_map(e->state[i])) return -1 * (travel_time(e, i) + 0);
    return (e->latest[i] - e->t) - travel_time(e, i);   /* get_current_delay (agent_utils.py:141-147) */
}

/* ------------------------------------------------------------------ RailEnv.step (rail_env.py:501-634) */
int orc_step(OrcEnv *e, const uint8_t *actions, int32_t *rewards, uint8_t *dones, uint8_t *done_all) {
    int A = e->A, i;
    int *np_r, *np_c, *np_d, *pa;
    int32_t *cur = NULL, *nxt = NULL;
    uint8_t *can_move;
    int all_done = 1, rc = ORC_OK;
    e->t += 1;                                                         /* :505 */
    if (e->done_all) { orc_set_error("Episode is done, cannot call step()"); return ORC_ERR_DONE; } /* :508-509 */
    np_r = ALLOC_I(A); np_c = ALLOC_I(A); np_d = ALLOC_I(A); pa = ALLOC_I(A);
    cur = (int32_t *)malloc(sizeof(int32_t) * A); nxt = (int32_t *)malloc(sizeof(int32_t) * A);
    can_move = (uint8_t *)malloc((size_t)A);
    for (i = 0; i < A; i++) rewards[i] = 0;                            /* :511 */

    for (i = 0; i < A; i++) {                                          /* loop 1 :519-569 */
        int a, upd, n;
        e->old_r[i] = e->r[i]; e->old_c[i] = e->c[i]; e->old_dir[i] = e->dir[i];   /* :521-522 */
        /* malfunction_generators.py:46-53 + malfunction_handler.py:35-46 */
        if (mt_rand53(e->mt, &e->mti) < e->malf_threshold)
            n = (int)mt_randint(e->mt, &e->mti, e->malf_min, e->malf_max + 1) + 1;
        else
            n = 0;
        if (e->malf[i] == 0) {
            e->malf[i] = n;
            if (n > 0) e->nmalf[i] += 1;
        }
        a = preprocess_action(e, i, actions[i] == 255 ? ACT_NOTHING : actions[i]);      /* :527-529 */
        /* ActionSaver.save_action_if_allowed (step_utils/action_saver.py:16-24) */
        if (a >= ACT_LEFT && a <= ACT_RIGHT && !e->saved[i] && e->state[i] != ST_DONE) e->saved[i] = a;
        upd = (e->scount[i] == e->max_count[i]) && !(e->malf[i] > 0) && a != ACT_STOP;   /* :535-537 */
        if (e->r[i] < 0 && e->state[i] != ST_DONE && a == ACT_STOP) e->saved[i] = 0;     /* :540-542 */
        if (e->state[i] == ST_DONE) {                                                    /* :546-547 */
            np_r[i] = e->r[i]; np_c[i] = e->c[i]; np_d[i] = e->dir[i];
        } else if (e->r[i] < 0 && e->saved[i]) {                                         /* :549-551 */
            np_r[i] = e->init_r[i]; np_c[i] = e->init_c[i]; np_d[i] = e->init_dir[i];
        } else if (e->saved[i] && upd) {                                                 /* :553-560 */
            int tv, nd = check_action(e, e->saved[i], e->r[i], e->c[i], e->dir[i], &tv); /* env_utils.py:26-43 */
            np_r[i] = e->r[i] + ORC_DR[nd]; np_c[i] = e->c[i] + ORC_DC[nd]; np_d[i] = nd;
            a = e->saved[i];
        } else {                                                                         /* :561-562 */
            np_r[i] = e->r[i]; np_c[i] = e->c[i]; np_d[i] = e->dir[i];
        }
        pa[i] = a;
        /* motionCheck.addAgent (:569, agent_chains.py:19-37): None -> virtual node (-1, i).
         * node ids: on-map (r,c) -> (r+1)*(W+2)+(c+1) (tolerates a one-cell overshoot), virtual -> base + i */
        {
            int base = (e->H + 2) * (e->W + 2);
            cur[i] = e->r[i] < 0 ? base + i : (e->r[i] + 1) * (e->W + 2) + e->c[i] + 1;
            nxt[i] = np_r[i] < 0 ? base + i : (np_r[i] + 1) * (e->W + 2) + np_c[i] + 1;
        }
    }

    orc_motion_check(A, cur, nxt, can_move);                           /* :572 */

    for (i = 0; i < A; i++) {                                          /* loop 2 :574-627 */
        int mv, in_malf, malf_done, dep, stop, vmove, at_target, conflict, s, ns;
        int a = pa[i];
        in_malf = e->malf[i] > 0;
        mv = in_malf ? 0 : can_move[i];                                /* :578-581 */
        mv = mv || (e->state[i] == ST_STOPPED && e->scount[i] != e->max_count[i]);   /* :583-584 */
        /* generate_state_transition_signals :369-395 */
        malf_done = e->malf[i] == 0;
        dep = e->t >= e->earliest[i];
        stop = a == ACT_STOP;
        vmove = (a >= ACT_LEFT && a <= ACT_RIGHT) && mv;
        at_target = e->r[i] >= 0 && e->r[i] == e->tgt_r[i] && e->c[i] == e->tgt_c[i];
        conflict = (!mv) && e->scount[i] == e->max_count[i];
        e->sig_in_malf[i] = (uint8_t)in_malf;
        /* TrainStateMachine.step (step_utils/state_machine.py:12-121) */
        s = e->state[i];
        switch (s) {
        case ST_WAITING: ns = in_malf ? ST_MALF_OFF : (dep ? ST_READY : ST_WAITING); break;
        case ST_READY: ns = in_malf ? ST_MALF_OFF : (vmove ? ST_MOVING : ST_READY); break;
        case ST_MALF_OFF:
            if (malf_done) {
                if (dep) ns = vmove ? ST_MOVING : (stop ? ST_STOPPED : ST_READY);
                else ns = ST_WAITING;
            } else ns = ST_MALF_OFF;
            break;
        case ST_MOVING: ns = in_malf ? ST_MALF : (at_target ? ST_DONE : ((stop || conflict) ? ST_STOPPED : ST_MOVING)); break;
        case ST_STOPPED: ns = in_malf ? ST_MALF : (vmove ? ST_MOVING : ST_STOPPED); break;
        case ST_MALF: ns = malf_done ? (vmove ? ST_MOVING : ST_STOPPED) : ST_MALF; break;
        default: ns = ST_DONE; break;
        }
        e->prev_state[i] = s;
        e->state[i] = ns;
        mv = mv && e->state[i] != ST_DONE;                             /* :596 */
        if (orc_is_on_map(e->state[i])) {                              /* :599-607 */
            if (orc_is_off_map(e->prev_state[i])) {
                e->r[i] = e->init_r[i]; e->c[i] = e->init_c[i]; e->dir[i] = e->init_dir[i];
            } else if (mv && e->scount[i] == e->max_count[i]) {
                e->r[i] = np_r[i]; e->c[i] = np_c[i]; e->dir[i] = np_d[i];
                /* update_if_reached (state_machine.py:139-144) */
                if (e->r[i] == e->tgt_r[i] && e->c[i] == e->tgt_c[i]) {
                    e->prev_state[i] = e->state[i];
                    e->state[i] = ST_DONE;
                }
            }
        }
        /* state_position_sync_check (step_utils/env_utils.py:45-52) */
        if ((orc_is_on_map(e->state[i]) && e->r[i] < 0) || (orc_is_off_map(e->state[i]) && e->r[i] >= 0)) {
            orc_set_error("Agent ID %d state %d / position desync", i, e->state[i]);
            rc = ORC_ERR_SYNC;
        }
        /* handle_done_state :493-499 */
        if (e->state[i] == ST_DONE && e->arrival[i] < 0) {
            e->arrival[i] = e->t;
            e->done[i] = 1;
            e->r[i] = e->c[i] = -1;
        }
        all_done &= (e->state[i] == ST_DONE);                          /* :615 */
        /* SpeedCounter.update_counter (speed_counter.py:10-14) */
        if (e->state[i] == ST_MOVING && e->old_r[i] >= 0) e->scount[i] = (e->scount[i] + 1) % (e->max_count[i] + 1);
        if (e->malf[i] > 0) e->malf[i] -= 1;                           /* malfunction_handler.py:48-50 */
        if (e->scount[i] == 0 && e->r[i] >= 0) e->saved[i] = 0;       /* :626-627 */
    }

    /* end_of_episode_update :476-491 */
    if (all_done || e->t >= e->T) {
        for (i = 0; i < A; i++) {
            rewards[i] += end_reward(e, i);
            e->done[i] = 1;
        }
        e->done_all = 1;
    }
    for (i = 0; i < A; i++) dones[i] = e->done[i];
    *done_all = e->done_all;
    free(np_r); free(np_c); free(np_d); free(pa); free(cur); free(nxt); free(can_move);
    return rc;
}
