/*
 * fl_oracle_obs.c -- CPU ORACLE, part 2: observation builders.  TEST INFRASTRUCTURE ONLY (see fl_oracle.h).
 *
 *  (1) flatland_cutils (the reference's C++/pybind11 module):
 *        AgentsLoader::update / Agent::Agent   flatland_cutils/src/loader.cpp:8-120, 221-327
 *        DeadlockChecker                       flatland_cutils/src/deadlock_checker.cpp:3-114
 *        ShortestPathPredictorForRailEnv::get  flatland_cutils/src/predictions.cpp:13-235
 *        TreeObsForRailEnv::get_many/get/_explore_branch/scale_node  flatland_cutils/src/treeobs.cpp:30-610
 *        calculate_evaluation_orders           flatland_cutils/src/tool.h:468-524
 *        AgentAttrParser::get_features         flatland_cutils/src/feature_parser.cpp:3-98
 *  (2) upstream Python TreeObsForRailEnv       flatland-rl/flatland/envs/observations.py:60-494
 *        + ShortestPathPredictorForRailEnv     flatland-rl/flatland/envs/predictions.py:97-180
 *
 * All float32 arithmetic of (1) is kept in float (compile with -ffp-contract=off).
 */
#include "fl_oracle_internal.h"

/* ------------------------------------------------------------------ shared helpers */
typedef struct { int r, c, d; } Way;

/* get_valid_move_actions_ (predictions.cpp:13-76 == rail_env_shortest_paths.py:17-71): candidate next
 * directions in the order the reference iterates them (std::set ordered by action L<F<R == OrderedSet
 * insertion order d-1, d, d+1). */
static int valid_moves(const OrcEnv *e, int r, int c, int d, int *cand) {
    uint16_t cell = orc_cell(e, r, c);
    int bits = orc_nibble(cell, d), n = 0, j;
    if (orc_popcount(cell) == 1) { /* is_dead_end */
        int ex = (d + 2) % 4;
        if ((bits >> (3 - ex)) & 1) cand[n++] = ex;
    } else {
        for (j = -1; j <= 1; j++) {
            int nd = (d + j + 4) % 4;
            if ((bits >> (3 - nd)) & 1) cand[n++] = nd;
        }
    }
    return n;
}

static float dmf(const OrcEnv *e, int agent, int r, int c, int d) {
    uint16_t v = orc_dm_at(e, agent, r, c, d);
    return v == 0xFFFF ? INFINITY : (float)v;
}

static void virtual_position(const OrcEnv *e, int i, int *r, int *c) {
    int s = e->state[i];
    if (orc_is_off_map(s)) { *r = e->init_r[i]; *c = e->init_c[i]; }
    else if (orc_is_on_map(s)) { *r = e->r[i]; *c = e->c[i]; }
    else { *r = e->tgt_r[i]; *c = e->tgt_c[i]; }
}

/* ------------------------------------------------------------------ predictors
 * Both produce, per agent, the waypoint list `path` (path[0] = current virtual position) and a function
 * t -> waypoint index.  pred_pos[t][a] = col*width + row (tool.h:391-398 / grid_utils.py:262-292). */
typedef struct {
    int depth;      /* max_pred_depth */
    int T;          /* number of time entries = depth + 1 */
    int *pos;       /* [T][A] */
    int *dir;       /* [T][A] */
    /* get_many(handles) with a strict subset (treeobs.cpp:50-62): predicted_pos / predicted_dir hold one entry per LISTED handle, in
     * the order of the list; entry j belongs to agent list[j].  The conflict test then works on list positions: it leaves out
     * position `agent.handle` (tool.h:428-434) and reads agents[position].state (treeobs.cpp:413, 435, 455) -- reproduced as it is.
     * list == NULL: every agent, position = handle. */
    int n;
    const int *list;
} Pred;

/* cutils: get_shortest_paths (predictions.cpp:78-144) -- strict greedy descent, <= max_depth iterations,
 * does NOT stop at the target by itself (it stops there because nothing is < 0). */
static int cutils_path(const OrcEnv *e, int i, int max_depth, Way *path) {
    int r, c, d = e->dir[i], n = 0, depth = 0;
    float distance = INFINITY;
    virtual_position(e, i, &r, &c);
    while (depth < max_depth) {
        int cand[3], nc = valid_moves(e, r, c, d, cand), j, best = -1;
        for (j = 0; j < nc; j++) {
            int nd = cand[j];
            float v = dmf(e, i, r + ORC_DR[nd], c + ORC_DC[nd], nd);
            if (v < distance) { best = nd; distance = v; }
        }
        path[n].r = r; path[n].c = c; path[n].d = d; n++;
        depth++;
        if (best < 0) return n;
        r += ORC_DR[best]; c += ORC_DC[best]; d = best;
    }
    path[n].r = r; path[n].c = c; path[n].d = d; n++; /* predictions.cpp:131-133 (max_depth != -1) */
    return n;
}

/* ShortestPathPredictorForRailEnv::get (predictions.cpp:146-235) + treeobs.cpp:52-65 */
static void cutils_predict(const OrcEnv *e, Pred *p) {
    int A = e->A, i, t;
    Way *path = (Way *)malloc(sizeof(Way) * (size_t)(p->depth + 2));
    for (i = 0; i < A; i++) {
        int n = cutils_path(e, i, p->depth, path);
        int vr, vc, vd = e->dir[i];
        int tpc = (int)(1 / (float)e->speed[i]); /* int(1 / agent_speed), agent_speed is a C++ float */
        int idx, k = 1; /* next unread waypoint (path[0] popped, :195-197) */
        int nr, nc_, nd;
        virtual_position(e, i, &vr, &vc);
        nr = vr; nc_ = vc; nd = vd;
        p->pos[0 * A + i] = vc * e->W + vr; p->dir[0 * A + i] = vd;
        for (idx = 0; idx < p->depth + 1; idx++) {
            if (!((nr == e->tgt_r[i] && nc_ == e->tgt_c[i]) || k >= n)) {
                if (idx % tpc == 0) { nr = path[k].r; nc_ = path[k].c; nd = path[k].d; k++; }
            }
            t = idx + 1;
            if (t < p->T) { p->pos[t * A + i] = nc_ * e->W + nr; p->dir[t * A + i] = nd; }
        }
    }
    free(path);
}

/* python: get_shortest_paths (rail_env_shortest_paths.py:203-274) with max_depth; returns n (0 = None) */
static int py_path(const OrcEnv *e, int i, int max_depth, Way *path) {
    int r, c, d = e->dir[i], n = 0, depth = 0;
    double distance = INFINITY;
    virtual_position(e, i, &r, &c);
    while (!(r == e->tgt_r[i] && c == e->tgt_c[i]) && depth < max_depth) {
        int cand[3], nc = valid_moves(e, r, c, d, cand), j, best = -1;
        for (j = 0; j < nc; j++) {
            int nd = cand[j];
            uint16_t v16 = orc_dm_at(e, i, r + ORC_DR[nd], c + ORC_DC[nd], nd);
            double v = v16 == 0xFFFF ? INFINITY : (double)v16;
            if (v < distance) { best = nd; distance = v; }
        }
        path[n].r = r; path[n].c = c; path[n].d = d; n++;
        depth++;
        if (best < 0) return 0;
        r += ORC_DR[best]; c += ORC_DC[best]; d = best;
    }
    if (depth < max_depth) { path[n].r = r; path[n].c = c; path[n].d = d; n++; }
    return n;
}

/* ShortestPathPredictorForRailEnv.get (predictions.py:97-180) + observations.py:68-85 */
static void py_predict(const OrcEnv *e, Pred *p) {
    int A = e->A, i;
    Way *path = (Way *)malloc(sizeof(Way) * (size_t)(p->depth + 2));
    for (i = 0; i < A; i++) {
        int n = py_path(e, i, p->depth, path);
        int vr, vc, vd = e->dir[i];
        int tpc = (int)(1.0 / e->speed[i]); /* int(np.reciprocal(speed)) */
        int index, k = 1, nr, nc_, nd;
        virtual_position(e, i, &vr, &vc);
        nr = vr; nc_ = vc; nd = vd;
        p->pos[0 * A + i] = vc * e->W + vr; p->dir[0 * A + i] = vd;
        for (index = 1; index < p->depth + 1; index++) {
            if (!((nr == e->tgt_r[i] && nc_ == e->tgt_c[i]) || k >= n)) {
                if (index % tpc == 0) { nr = path[k].r; nc_ = path[k].c; nd = path[k].d; k++; }
            }
            p->pos[index * A + i] = nc_ * e->W + nr; p->dir[index * A + i] = nd;
        }
    }
    free(path);
}

/* ------------------------------------------------------------------ per-cell lookups (location_has_*) */
typedef struct {
    int8_t *has_agent;  /* [H*W] */
    int8_t *adir;
    int *amalf;         /* cutils: 0/1; python: down counter */
    double *aspeed;     /* python: f64; cutils reads it back as float */
    int *ready;         /* cutils: count-1 (treeobs.cpp:82-91); python: count (observations.py:106-109) */
    int8_t *has_ready;
    int8_t *has_target; /* python only (observations.py:57-58) */
} CellMaps;

static void build_maps(const OrcEnv *e, CellMaps *m, int cutils) {
    int HW = e->H * e->W, i;
    m->has_agent = (int8_t *)calloc((size_t)HW, 1); m->adir = (int8_t *)calloc((size_t)HW, 1);
    m->amalf = (int *)calloc((size_t)HW, sizeof(int)); m->aspeed = (double *)calloc((size_t)HW, sizeof(double));
    m->ready = (int *)calloc((size_t)HW, sizeof(int)); m->has_ready = (int8_t *)calloc((size_t)HW, 1);
    m->has_target = (int8_t *)calloc((size_t)HW, 1);
    for (i = 0; i < e->A; i++) {
        if (!orc_is_off_map(e->state[i]) && e->r[i] >= 0) { /* treeobs.cpp:74-81 / observations.py:96-103 */
            int p = e->r[i] * e->W + e->c[i];
            m->has_agent[p] = 1; m->adir[p] = (int8_t)e->dir[i];
            m->aspeed[p] = e->speed[i];
            m->amalf[p] = cutils ? (e->malf[i] != 0) : e->malf[i]; /* loader.cpp:38-39 reads it through py::bool_ */
        }
        if (orc_is_off_map(e->state[i])) {
            int p = e->init_r[i] * e->W + e->init_c[i];
            if (cutils) {
                if (m->has_ready[p]) m->ready[p] += 1; else { m->has_ready[p] = 1; m->ready[p] = 0; }
            } else {
                m->has_ready[p] = 1; m->ready[p] += 1;
            }
        }
        m->has_target[e->tgt_r[i] * e->W + e->tgt_c[i]] = 1;
    }
}
static void free_maps(CellMaps *m) {
    free(m->has_agent); free(m->adir); free(m->amalf); free(m->aspeed); free(m->ready); free(m->has_ready);
    free(m->has_target);
}

/* ------------------------------------------------------------------ the branch walk
 * treeobs.cpp:258-610 (cutils = 1) and observations.py:256-494 (cutils = 0).  Features are produced in
 * double; every value that the cutils version holds in float32 is exactly representable (small integers,
 * float32 speeds widened), so the cast at the end is exact. */
typedef struct {
    double f[12];
    int end_r, end_c, end_d; /* final (position, direction) */
    int last_is_switch, last_is_dead_end, last_is_terminal, last_is_target;
    int tot_dist;
    int err;
} Branch;

static void explore_branch(const OrcEnv *e, const CellMaps *m, const Pred *p, int handle, int r, int c, int d,
                           int tot_dist, int cutils, Branch *out) {
    int A = e->A, W = e->W;
    int exploring = 1;
    double own_target = INFINITY, other_agent = INFINITY, other_target = INFINITY, pot_conflict = INFINITY,
           unusable = INFINITY;
    int same_dir = 0, opp_dir = 0, malfunctioning = 0, ready = 0;
    double min_speed = 1.0;
    float tpc_f = (float)(1.0 / (double)(float)e->speed[handle]); /* float time_per_cell = 1.0 / agent.speed (treeobs.cpp:304) */
    double tpc_d = 1.0 / e->speed[handle];                        /* np.reciprocal(speed) (observations.py:277) */
    /* visited (cell,dir) set of this branch only */
    uint8_t *visited = (uint8_t *)calloc((size_t)e->H * e->W * 4, 1);
    memset(out, 0, sizeof *out);
    while (exploring) {
        int pcell = r * W + c;
        uint16_t cell;
        int bits, total, num, crossing, predicted_time;
        if (m->has_agent[pcell]) {
            if ((double)tot_dist < other_agent) other_agent = tot_dist;
            if (m->amalf[pcell] > malfunctioning) malfunctioning = m->amalf[pcell];
            ready += m->has_ready[pcell] ? m->ready[pcell] : 0;
            if (m->adir[pcell] == d) {
                double sp = cutils ? (double)(float)m->aspeed[pcell] : m->aspeed[pcell];
                same_dir += 1;
                if (sp < min_speed) min_speed = sp;
            } else {
                opp_dir += 1;
            }
        }
        cell = orc_cell(e, r, c);
        bits = orc_nibble(cell, d);
        total = orc_popcount(cell);
        crossing = (cell == 0x8421);
        /* potential conflicts with the other agents' predicted paths */
        if (p) {
            if (cutils) predicted_time = (int)((float)(int)tot_dist * tpc_f);   /* treeobs.cpp:378 */
            else predicted_time = (int)((double)tot_dist * tpc_d);               /* observations.py:329 */
            if (predicted_time < p->T && tot_dist < p->T) {
                int int_position = c * W + r;
                int pre = predicted_time - 1 < 0 ? 0 : predicted_time - 1;
                int post = predicted_time + 1 > p->T - 1 ? p->T - 1 : predicted_time + 1;
                int times[3], k, a, j, sel = -1;
                const int n_list = p->list ? p->n : A;
                times[0] = predicted_time; times[1] = pre; times[2] = post;
                for (k = 0; k < 3 && sel < 0; k++) /* "in np.delete(predicted_pos[t], handle)" */
                    for (j = 0; j < n_list; j++) {
                        a = p->list ? p->list[j] : j;
                        if (j != handle && p->pos[times[k] * A + a] == int_position) { sel = k; break; }
                    }
                if (sel >= 0) {
                    int ts = times[sel];
                    /* cutils indexes predicted_dir with predicted_time in all three branches (treeobs.cpp:429-433,
                     * 449-453); python uses the matching step (observations.py:351-363) */
                    int td = cutils ? predicted_time : ts;
                    for (j = 0; j < n_list; j++) {
                        a = p->list ? p->list[j] : j;
                        if (p->pos[ts * A + a] != int_position) continue;
                        {
                            int cd = p->dir[td * A + a];
                            if (d != cd && ((bits >> (3 - ((cd + 2) % 4))) & 1) && (double)tot_dist < pot_conflict)
                                pot_conflict = tot_dist;
                            if (e->state[j] == ST_DONE && (double)tot_dist < pot_conflict) pot_conflict = tot_dist;   /* agents[list position] */
                        }
                    }
                }
            }
        }
        if (!cutils && m->has_target[pcell] && !(r == e->tgt_r[handle] && c == e->tgt_c[handle])) {
            if ((double)tot_dist < other_target) other_target = tot_dist; /* cutils: map never filled (treeobs.cpp:72) */
        }
        if (r == e->tgt_r[handle] && c == e->tgt_c[handle] && (double)tot_dist < own_target) own_target = tot_dist;
        if (visited[pcell * 4 + d]) { out->last_is_terminal = 1; break; }
        visited[pcell * 4 + d] = 1;
        if (r == e->tgt_r[handle] && c == e->tgt_c[handle]) { out->last_is_target = 1; break; }
        if (crossing) total = 2;
        num = orc_popcount((unsigned)bits);
        exploring = 0;
        if (total > 2 && 2 > num && (double)tot_dist < unusable) unusable = tot_dist;
        if (num == 1) {
            if (total == 1) out->last_is_dead_end = 1;
            if (!out->last_is_dead_end) {
                int m_;
                exploring = 1;
                for (m_ = 0; m_ < 3; m_++) if ((bits >> (3 - m_)) & 1) break;
                d = m_;
                r += ORC_DR[d]; c += ORC_DC[d];
                tot_dist += 1;
            }
        } else if (num > 0) {
            out->last_is_switch = 1;
            break;
        } else {
            if (cutils) { out->err = ORC_ERR_ZERO_TRANSITION; }   /* treeobs.cpp:529-535 throws */
            out->last_is_terminal = 1;                             /* observations.py:420-425 */
            break;
        }
    }
    free(visited);
    {
        double dist_next, dist_min;
        uint16_t v = orc_dm_at(e, handle, r, c, d);
        double dmv = v == 0xFFFF ? INFINITY : (double)v;
        if (out->last_is_target) { dist_next = tot_dist; dist_min = 0; }
        else if (out->last_is_terminal) { dist_next = INFINITY; dist_min = dmv; }
        else { dist_next = tot_dist; dist_min = dmv; }
        out->f[0] = own_target; out->f[1] = other_target; out->f[2] = other_agent; out->f[3] = pot_conflict;
        out->f[4] = unusable; out->f[5] = dist_next; out->f[6] = dist_min; out->f[7] = same_dir; out->f[8] = opp_dir;
        out->f[9] = malfunctioning; out->f[10] = min_speed; out->f[11] = ready;
    }
    out->end_r = r; out->end_c = c; out->end_d = d; out->tot_dist = tot_dist;
}

/* ------------------------------------------------------------------ cutils: loader / deadlock / attr */
static const uint16_t TRANSITION_LIST[11] = { /* RailEnvTransitions.transition_list (core/grid/rail_env_grid.py:28-38) */
    0x0000, 0x8020, 0x9220, 0x8421, 0x9621, 0xCC33, 0x5202, 0x2000, 0x4002, 0x1200, 0xC022};

/* rotate_transition (tool.h:300-335): every 4-bit block rotated right by k, then the 16-bit word by 4k */
static uint16_t rotate_transition(uint16_t cell, int degrees) {
    int k = degrees / 90, i;
    uint16_t v = 0;
    for (i = 0; i < 4; i++) {
        unsigned nib = (cell >> ((3 - i) * 4)) & 15u;
        nib = ((nib >> k) | (nib << (4 - k))) & 15u;
        v |= (uint16_t)(nib << ((3 - i) * 4));
    }
    return (uint16_t)(((v >> (4 * k)) | (v << (16 - 4 * k))) & 0xFFFF);
}

static int road_type_of(uint16_t cell) { /* Agent::update_transitions (loader.cpp:122-161) */
    int rot, k;
    for (rot = 0; rot < 4; rot++) {
        uint16_t t = rot == 0 ? cell : rotate_transition(cell, rot * 90);
        for (k = 0; k < 11; k++)
            if (TRANSITION_LIST[k] == t) return k;
    }
    return 0;
}

typedef struct {
    int n;
    const OrcEnv *e;
    int *agent_at; /* [H*W] handle or -1: DeadlockChecker::agent_positions */
    int *checked;
    int *dep;      /* [A][4] */
    int *ndep;
} Dlk;

static int dl_active(const OrcEnv *e, int i) { return orc_is_on_map(e->state[i]); }

/* DeadlockChecker::_check_blocked (deadlock_checker.cpp:30-75) */
static int dl_check_blocked(Dlk *k, OrcEnv *e, int h) {
    int bits = orc_nibble(orc_cell(e, e->r[h], e->c[h]), e->dir[h]); /* agent.cell_transitions */
    int dir;
    k->checked[h] = 1;
    for (dir = 0; dir < 4; dir++) {
        int nr, nc, opp;
        if (!((bits >> (3 - dir)) & 1)) continue;
        nr = e->r[h] + ORC_DR[dir]; nc = e->c[h] + ORC_DC[dir];
        opp = orc_in_bounds(e, nr, nc) ? k->agent_at[nr * e->W + nc] : -1;
        if (opp == -1) { k->checked[h] = 2; return 0; }
        if (e->deadlocked[opp]) continue;
        if (k->checked[opp] == 0) dl_check_blocked(k, e, opp);
        if (k->checked[opp] == 2 && !e->deadlocked[opp]) { k->checked[h] = 2; return 0; }
        k->dep[h * 4 + k->ndep[h]++] = opp;
    }
    if (k->ndep[h] == 0) {
        k->checked[h] = 2;
        if (bits == 0) return 0;
        e->deadlocked[h] = 1;
        return 1;
    }
    return 0;
}

/* DeadlockChecker::update_deadlocks + _fix_deps (deadlock_checker.cpp:11-28, 77-110) */
static void update_deadlocks(OrcEnv *e) {
    int A = e->A, HW = e->H * e->W, i, j, any;
    Dlk k;
    k.e = e; k.n = A;
    k.agent_at = (int *)malloc(sizeof(int) * (size_t)HW);
    k.checked = (int *)calloc((size_t)A, sizeof(int));
    k.dep = (int *)calloc((size_t)A * 4, sizeof(int));
    k.ndep = (int *)calloc((size_t)A, sizeof(int));
    for (i = 0; i < HW; i++) k.agent_at[i] = -1;
    for (i = 0; i < A; i++)
        if (dl_active(e, i)) k.agent_at[e->r[i] * e->W + e->c[i]] = i;
    for (i = 0; i < A; i++)
        if (dl_active(e, i) && !e->deadlocked[i] && !k.checked[i]) dl_check_blocked(&k, e, i);
    any = 1;
    while (any) {
        any = 0;
        for (i = 0; i < A; i++) {
            if (k.checked[i] == 1) {
                int cnt = 0;
                for (j = 0; j < k.ndep[i]; j++) {
                    int opp = k.dep[i * 4 + j];
                    if (k.checked[opp] == 2) {
                        if (e->deadlocked[opp]) cnt += 1;
                        else { k.checked[i] = 2; any = 1; }
                    }
                }
                if (cnt == k.ndep[i]) { k.checked[i] = 2; e->deadlocked[i] = 1; any = 1; }
            }
        }
    }
    for (i = 0; i < A; i++)
        if (k.checked[i] == 1) { e->deadlocked[i] = 1; k.checked[i] = 2; }
    free(k.agent_at); free(k.checked); free(k.dep); free(k.ndep);
}

void orc_obs_cutils_reset(OrcEnv *e) { memset(e->deadlocked, 0, (size_t)e->A); }

/* valid-action mask (loader.cpp:273-312) */
static void valid_actions_of(const OrcEnv *e, int i, uint8_t *va) {
    int s = e->state[i], a;
    for (a = 0; a < 5; a++) va[a] = 0;
    if (s == ST_MOVING || s == ST_STOPPED) {
        if (e->scount[i] == 0) {
            int bits = orc_nibble(orc_cell(e, e->r[i], e->c[i]), e->dir[i]);
            int has_branch = 0, cnt = 0;
            for (a = ACT_LEFT; a <= ACT_RIGHT; a++) {
                int nd = (e->dir[i] + a - 2 + 4) % 4;
                va[a] = (uint8_t)((bits >> (3 - nd)) & 1);
                if (va[a]) {
                    int nr = e->r[i] + ORC_DR[nd], nc = e->c[i] + ORC_DC[nd];
                    cnt += 1;
                    if (orc_in_bounds(e, nr, nc) && orc_popcount(orc_cell(e, nr, nc)) > 2) has_branch = 1; /* is_branch_cell */
                }
            }
            if (orc_popcount(orc_cell(e, e->r[i], e->c[i])) > 2 || (cnt == 1 && has_branch)) va[ACT_STOP] = 1;
        } else {
            va[ACT_NOTHING] = 1;
        }
    } else if (s == ST_READY) {
        va[ACT_FORWARD] = 1; va[ACT_STOP] = 1;
    } else {
        va[ACT_NOTHING] = 1;
    }
}

/* scale_node (treeobs.cpp:111-152); in float32 */
static void scale_node(const double *f, float max_dist, int n_agents, float *o) {
    int k;
    for (k = 0; k < 7; k++) o[k] = isinf(f[k]) ? -1.0f : (float)f[k] / max_dist;
    o[7] = f[7] != -1 ? (float)f[7] / (float)n_agents : -1.0f;
    o[8] = f[8] != -1 ? (float)f[8] / (float)n_agents : -1.0f;
    o[9] = f[9] != -1 ? (float)f[9] / (float)n_agents : -1.0f;
    o[10] = f[10] != -1 ? (float)f[10] : -1.0f;
    o[11] = f[11] != -1 ? (float)f[11] / (float)n_agents : -1.0f;
}

typedef struct { int r, c, d, action_dir, parent, tot_dist, is_null; } QCell;

int orc_obs_cutils(OrcEnv *e, int max_nodes, int pred_depth, float *attr, float *forest, int32_t *adjacency,
                   int32_t *node_order, int32_t *edge_order, uint8_t *valid, double *props) {
    return orc_obs_cutils_handles(e, max_nodes, pred_depth, NULL, 0, attr, forest, adjacency, node_order, edge_order, valid, props);
}

/* get_many(handles) (treeobs.cpp:30-108): handles == NULL: every agent.  A strict subset has to be a permutation of 0 .. n-1 (every
 * handle below the length of the list: get_possible_conflicting erases position `handle`, tool.h:428-434 -- undefined behaviour
 * otherwise); the trees of ALL agents are written (row i = agent i), the caller picks the listed ones. */
int orc_obs_cutils_handles(OrcEnv *e, int max_nodes, int pred_depth, const int32_t *handles, int n_handles, float *attr, float *forest,
                           int32_t *adjacency, int32_t *node_order, int32_t *edge_order, uint8_t *valid, double *props) {
    int A = e->A, i, rc = ORC_OK;
    Pred p;
    CellMaps m;
    float *dist_target = (float *)malloc(sizeof(float) * (size_t)A);
    float *init_dist = (float *)malloc(sizeof(float) * (size_t)A);
    QCell *queue = (QCell *)malloc(sizeof(QCell) * (size_t)(3 * max_nodes + 8));
    static const double NULL_NODE[12] = {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY,
                                         -1, -1, -1, -1, -1};
    /* AgentsLoader::update (loader.cpp:221-327) */
    for (i = 0; i < A; i++) {
        init_dist[i] = dmf(e, i, e->init_r[i], e->init_c[i], e->init_dir[i]);        /* update_dist_target :163-179 */
        if (e->state[i] == ST_DONE) dist_target[i] = 0;
        else if (orc_is_off_map(e->state[i])) dist_target[i] = init_dist[i];
        else dist_target[i] = dmf(e, i, e->r[i], e->c[i], e->dir[i]);
        valid_actions_of(e, i, valid + (size_t)i * 5);
    }
    update_deadlocks(e);
    /* predictor + per-cell maps (treeobs.cpp:47-92) */
    p.depth = pred_depth; p.T = pred_depth + 1;
    p.pos = (int *)malloc(sizeof(int) * (size_t)p.T * A);
    p.dir = (int *)malloc(sizeof(int) * (size_t)p.T * A);
    p.n = n_handles; p.list = handles;
    cutils_predict(e, &p);
    build_maps(e, &m, 1);

    for (i = 0; i < A; i++) {
        /* TreeObsForRailEnv::get (treeobs.cpp:154-256) */
        float *F = forest + (size_t)i * max_nodes * 12;
        int32_t *ADJ = adjacency + (size_t)i * (max_nodes - 1) * 3;
        int32_t *NO = node_order + (size_t)i * max_nodes, *EO = edge_order + (size_t)i * (max_nodes - 1);
        int vr, vc, bits, num, orientation, qh = 0, qt = 0, n_nodes = 1, k, a;
        double root[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        virtual_position(e, i, &vr, &vc);
        bits = orc_nibble(orc_cell(e, vr, vc), e->dir[i]);
        num = orc_popcount((unsigned)bits);
        root[6] = dist_target[i];
        root[9] = (e->nmalf[i] != 0);             /* num_malfunctions read through py::bool_ (loader.cpp:40-41) */
        root[10] = (double)(float)e->speed[i];
        scale_node(root, (float)e->T, A, F);
        orientation = e->dir[i];
        if (num == 1) for (orientation = 0; orientation < 3; orientation++) if ((bits >> (3 - orientation)) & 1) break;
        for (a = -1; a <= 1; a++) {
            int bd = (orientation + a + 4) % 4;
            QCell q;
            q.d = bd; q.action_dir = a; q.parent = 0; q.tot_dist = 1;
            if ((bits >> (3 - bd)) & 1) { q.r = vr + ORC_DR[bd]; q.c = vc + ORC_DC[bd]; q.is_null = 0; }
            else { q.r = q.c = -1; q.is_null = 1; }
            queue[qt++] = q;
        }
        while (n_nodes < max_nodes) {
            int idx_node = n_nodes;
            int32_t *adj = ADJ + (size_t)(idx_node - 1) * 3;
            if (qh == qt) { /* empty queue: padding (treeobs.cpp:268-276, 245-249) */
                scale_node(NULL_NODE, (float)e->T, A, F + (size_t)idx_node * 12);
                adj[0] = adj[1] = adj[2] = -2;
            } else {
                QCell q = queue[qh++];
                if (q.is_null) {
                    scale_node(NULL_NODE, (float)e->T, A, F + (size_t)idx_node * 12);
                } else {
                    Branch br;
                    int pbits;
                    explore_branch(e, &m, &p, i, q.r, q.c, q.d, q.tot_dist, 1, &br);
                    if (br.err) { rc = br.err; orc_set_error("WRONG CELL TYPE detected in tree-search (0 transitions possible)"); }
                    scale_node(br.f, (float)e->T, A, F + (size_t)idx_node * 12);
                    pbits = orc_nibble(orc_cell(e, br.end_r, br.end_c), br.end_d);
                    for (a = -1; a <= 1; a++) { /* children (treeobs.cpp:583-608) */
                        int bd = (br.end_d + 4 + a) % 4, rev = (bd + 2) % 4;
                        QCell ch;
                        ch.action_dir = a; ch.parent = idx_node; ch.tot_dist = br.tot_dist + 1; ch.is_null = 0;
                        if (br.last_is_dead_end && ((pbits >> (3 - rev)) & 1)) {
                            ch.r = br.end_r + ORC_DR[rev]; ch.c = br.end_c + ORC_DC[rev]; ch.d = rev;
                        } else if (br.last_is_switch && ((pbits >> (3 - bd)) & 1)) {
                            ch.r = br.end_r + ORC_DR[bd]; ch.c = br.end_c + ORC_DC[bd]; ch.d = bd;
                        } else {
                            ch.r = ch.c = -1; ch.d = bd; ch.is_null = 1;
                        }
                        queue[qt++] = ch;
                    }
                }
                adj[0] = q.parent; adj[1] = idx_node; adj[2] = q.action_dir;
            }
            n_nodes++;
        }
        /* calculate_evaluation_orders (tool.h:468-524): order = height above the leaves */
        {
            int n_real = 1, changed;
            for (k = 0; k < max_nodes - 1; k++) if (ADJ[k * 3] != -2) n_real++;
            for (k = 0; k < max_nodes; k++) NO[k] = k < n_real ? 0 : -2;
            do {
                changed = 0;
                for (k = 0; k < max_nodes - 1; k++) {
                    int par = ADJ[k * 3], ch = ADJ[k * 3 + 1];
                    if (par >= 0 && NO[par] < NO[ch] + 1) { NO[par] = NO[ch] + 1; changed = 1; }
                }
            } while (changed);
            for (k = 0; k < max_nodes - 1; k++) EO[k] = ADJ[k * 3] < 0 ? -2 : NO[ADJ[k * 3]];
        }
    }

    /* AgentAttrParser::get_features (feature_parser.cpp:3-98) */
    for (i = 0; i < A; i++) {
        float *o = attr + (size_t)i * 83;
        int n = 0, k, s = e->state[i];
        uint16_t cell = e->r[i] >= 0 ? orc_cell(e, e->r[i], e->c[i]) : 0;
        int road_type = e->r[i] >= 0 ? road_type_of(cell) : 0;
        int nmalf01 = e->nmalf[i] != 0, malf01 = e->malf[i] != 0;
        int old_dir = e->old_dir[i] < 0 ? e->dir[i] : e->old_dir[i];
        float max_t = (float)e->T, max_dist_target = (float)((e->H + e->W) * 8);
        float f_handle, f_step, f_earliest, f_latest, f_arrival, f_before_late, f_dist, f_antic, f_init_dist;
        for (k = 0; k < 7; k++) o[n++] = (k == s) ? 1.0f : 0.0f;
        for (k = 0; k < 11; k++) o[n++] = (k == road_type) ? 1.0f : 0.0f;
        for (k = 0; k < 10; k++) o[n++] = (k == nmalf01) ? 1.0f : 0.0f;
        for (k = 0; k < 4; k++) o[n++] = (k == e->init_dir[i]) ? 1.0f : 0.0f;
        for (k = 0; k < 4; k++) o[n++] = (k == e->dir[i]) ? 1.0f : 0.0f;
        for (k = 0; k < 4; k++) o[n++] = (k == old_dir) ? 1.0f : 0.0f;
        o[n++] = (float)(s == ST_MOVING);
        o[n++] = (float)e->deadlocked[i];
        o[n++] = (float)e->sig_in_malf[i];          /* state_machine.st_signals.in_malfunction (loader.cpp:16-18) */
        o[n++] = (float)(e->malf[i] == 0);           /* malfunction_counter_complete (loader.cpp:35-37) */
        o[n++] = (float)(e->scount[i] == 0);
        o[n++] = (float)(e->scount[i] == e->max_count[i]);
        o[n++] = (float)(s == ST_MALF || s == ST_MALF_OFF);
        o[n++] = (float)orc_is_off_map(s);
        o[n++] = (float)orc_is_on_map(s);
        for (k = 15; k >= 0; k--) o[n++] = (float)((cell >> k) & 1);
        for (k = 0; k < 5; k++) o[n++] = (float)valid[(size_t)i * 5 + k];
        f_handle = (float)i / (float)A;
        f_step = (float)e->t / max_t;
        f_earliest = (float)e->earliest[i] / max_t;
        f_latest = (float)e->latest[i] / max_t;
        f_arrival = (float)e->arrival[i] / max_t;
        f_before_late = f_latest - f_step;
        f_dist = isinf(dist_target[i]) ? 8.0f : dist_target[i] / max_dist_target;
        f_antic = f_before_late < f_dist ? f_before_late : f_dist;
        f_init_dist = isinf(init_dist[i]) ? 8.0f : init_dist[i] / max_dist_target;
        o[n++] = f_handle; o[n++] = f_step; o[n++] = f_earliest; o[n++] = f_latest; o[n++] = f_arrival;
        o[n++] = f_before_late; o[n++] = f_dist; o[n++] = f_antic;
        o[n++] = (float)e->max_count[i] / 10;        /* speed_max_count / fp::speed_max_count */
        o[n++] = (float)e->speed[i] / 1.0f;
        o[n++] = (float)e->scount[i] / 10;
        o[n++] = (float)malf01 / 10;
        o[n++] = f_init_dist;
        if (props) {
            props[i * 3 + 0] = (double)dist_target[i];
            props[i * 3 + 1] = (double)e->deadlocked[i];
            props[i * 3 + 2] = (double)(s == ST_READY);
        }
    }
    free_maps(&m);
    free(p.pos); free(p.dir); free(dist_target); free(init_dist); free(queue);
    return rc;
}

/* ------------------------------------------------------------------ upstream python tree (observations.py) */
static int py_subtree_size(int depth, int max_depth) {
    int n = 0, p = 1, k;
    for (k = depth; k <= max_depth; k++) { n += p; p *= 4; }
    return n;
}

static double *py_fill_missing(double *o, int depth, int max_depth) {
    int n = py_subtree_size(depth, max_depth) * 12, k;
    for (k = 0; k < n; k++) o[k] = -INFINITY;
    return o + n;
}

/* _explore_branch recursion (observations.py:256-494); writes the subtree in DFS pre-order */
static double *py_branch(const OrcEnv *e, const CellMaps *m, const Pred *p, int handle, int r, int c, int d,
                         int tot_dist, int depth, int max_depth, double *o) {
    Branch br;
    int i, pbits;
    explore_branch(e, m, p, handle, r, c, d, tot_dist, 0, &br);
    memcpy(o, br.f, sizeof br.f);
    o += 12;
    if (depth >= max_depth) return o; /* :491-492 childs cleared at depth == max_depth */
    pbits = orc_nibble(orc_cell(e, br.end_r, br.end_c), br.end_d);
    for (i = -1; i <= 2; i++) { /* L, F, R, B */
        int bd = (br.end_d + 4 + i) % 4, rev = (bd + 2) % 4;
        if (br.last_is_dead_end && ((pbits >> (3 - rev)) & 1))
            o = py_branch(e, m, p, handle, br.end_r + ORC_DR[rev], br.end_c + ORC_DC[rev], rev, br.tot_dist + 1, depth + 1,
                          max_depth, o);
        else if (br.last_is_switch && ((pbits >> (3 - bd)) & 1))
            o = py_branch(e, m, p, handle, br.end_r + ORC_DR[bd], br.end_c + ORC_DC[bd], bd, br.tot_dist + 1, depth + 1,
                          max_depth, o);
        else
            o = py_fill_missing(o, depth + 1, max_depth);
    }
    return o;
}

int orc_obs_pytree(OrcEnv *e, int max_depth, int pred_depth, double *out) {
    return orc_obs_pytree_handles(e, max_depth, pred_depth, NULL, 0, out);
}

/* TreeObsForRailEnv.get_many(handles) (observations.py:60-115): handles == NULL: every agent.  With a list, predicted_pos[t] /
 * predicted_dir[t] hold the LISTED handles' predictions in list order (:75-83); the conflict test deletes list position `handle`
 * (np.delete(..., handle, 0), :337) and reads env.agents[list position].state (:344) -- defined for permutations of 0 .. n-1 (any
 * other list: IndexError in the reference).  Rows of ALL agents are returned (row i = agent i; the reference returns the listed ones). */
int orc_obs_pytree_handles(OrcEnv *e, int max_depth, int pred_depth, const int32_t *handles, int n_handles, double *out) {
    int A = e->A, i, N = py_subtree_size(0, max_depth);
    Pred p, *pp = NULL;
    CellMaps m;
    if (handles) {
        int j;
        if (n_handles < 1 || n_handles > A) return ORC_ERR_ARG;
        for (j = 0; j < n_handles; j++) if (handles[j] < 0 || handles[j] >= n_handles) return ORC_ERR_ARG;
    }
    if (pred_depth >= 0) {
        p.depth = pred_depth; p.T = pred_depth + 1;
        p.n = handles ? n_handles : 0; p.list = handles;
        p.pos = (int *)malloc(sizeof(int) * (size_t)p.T * A);
        p.dir = (int *)malloc(sizeof(int) * (size_t)p.T * A);
        py_predict(e, &p);
        pp = &p;
    }
    build_maps(e, &m, 0);
    for (i = 0; i < A; i++) { /* get() (observations.py:196-254) */
        double *o = out + (size_t)i * N * 12;
        int vr, vc, bits, num, orientation, k;
        uint16_t v;
        virtual_position(e, i, &vr, &vc);
        bits = orc_nibble(orc_cell(e, vr, vc), e->dir[i]);
        num = orc_popcount((unsigned)bits);
        v = orc_dm_at(e, i, vr, vc, e->dir[i]);
        for (k = 0; k < 12; k++) o[k] = 0;
        o[6] = v == 0xFFFF ? INFINITY : (double)v;
        o[9] = e->malf[i];
        o[10] = e->speed[i];
        o += 12;
        orientation = e->dir[i];
        if (num == 1) for (orientation = 0; orientation < 3; orientation++) if ((bits >> (3 - orientation)) & 1) break;
        for (k = -1; k <= 2; k++) {
            int bd = (orientation + k + 4) % 4;
            if ((bits >> (3 - bd)) & 1)
                o = py_branch(e, &m, pp, i, vr + ORC_DR[bd], vc + ORC_DC[bd], bd, 1, 1, max_depth, o);
            else
                o = py_fill_missing(o, 1, max_depth);
        }
    }
    free_maps(&m);
    if (pp) { free(p.pos); free(p.dir); }
    return ORC_OK;
}
