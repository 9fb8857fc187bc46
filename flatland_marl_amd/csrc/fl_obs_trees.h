// fl_obs_trees.h -- the trees of both builders: pass A (topology from the static segment table, one BFS level per step), rows
// (the 12 features of a node from its descriptor and the accumulators of pass B), evaluation orders, and the round loops that
// tie pass A / pass B / rows together.  Device code.
#pragma once
#include "fl_obs_passb.h"

// the 12 features of node k from its descriptor and accumulators (treeobs.cpp:546-573 / observations.py:433-461)
template <bool CUTILS>
__device__ __forceinline__ void node_row(const ObsCtx &X, int handle, const int *scr, int cap, int k, double *f) {
    const uint32_t tv = (uint32_t)nt_r(scr, cap, N_TV, k), uf = (uint32_t)nt_r(scr, cap, N_UF, k);
    const int tot_end = nt_tot(tv) + nt_vis(tv) - 1;
    const uint32_t flags = nt_flags(uf);
    const bool tgt = flags & ND_TARGET;
    double dist_min = 0;
    if (flags & ND_PHANTOM) dist_min = INFINITY;  // the distance map of a cell without rail
    else if (!tgt) {
        const uint16_t dv = X.dm[X.a_tslot[handle] * X.SS + nt_end((uint32_t)nt_r(scr, cap, N_SE, k))];
        dist_min = dv == FL_INF16 ? INFINITY : (double)dv;
    }
    const int oa = nt_r(scr, cap, N_OA, k), pc = nt_r(scr, cap, N_PC, k), ot = CUTILS ? 0x7fffffff : nt_r(scr, cap, N_OT, k), un = nt_unus(uf);
    const uint32_t cnt = (uint32_t)nt_r(scr, cap, N_CNT, k), rm = (uint32_t)nt_r(scr, cap, N_RM, k), ms = (uint32_t)nt_r(scr, cap, N_MS, k);
    f[0] = tgt ? (double)tot_end : INFINITY;
    f[1] = ot == 0x7fffffff ? INFINITY : (double)ot;
    f[2] = oa == 0x7fffffff ? INFINITY : (double)oa;
    f[3] = pc == 0x7fffffff ? INFINITY : (double)pc;
    f[4] = un < 0 ? INFINITY : (double)un;
    f[5] = (flags & ND_TERMINAL) ? INFINITY : (double)tot_end;
    f[6] = dist_min;
    f[7] = (double)(cnt & 0xFFFFu); f[8] = (double)(cnt >> 16);
    f[9] = CUTILS ? (double)((rm >> 16) & 1u) : (double)nt_r(scr, cap, N_MALF, k);
    // the slowest same-direction occupant (its speed as the builder reads it: float in flatland_cutils), 1.0 when there is none
    f[10] = ms == 0xFFFFFFFFu ? 1.0 : (CUTILS ? (double)(float)X.a_speed[ms & 1023u] : X.a_speed[ms & 1023u]);
    f[11] = (double)(rm & 0xFFFFu);
}

// children of a node (treeobs.cpp:583-608 / observations.py:464-489): child k (k = 0 left, 1 forward, 2 right, 3 back) of a walk
// that ends at a switch or a dead end starts at the state the segment table has for it (fl_dmap.hip k_segments: kids01 / kids23,
// u16 each; FL_R_NONE = null cell, FL_R_PHANTOM = a cell without rail, see there); any other walk has no children.  Pass A hands
// the packed words to the children's lanes, which decode their own.

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
    // rows are 48 B, 16-B aligned
    out_store_f4(dst, v[0], v[1], v[2], v[3]);
    out_store_f4(dst + 4, v[4], v[5], v[6], v[7]);
    out_store_f4(dst + 8, v[8], v[9], v[10], v[11]);
}


// ---- upstream dense tree (observations.py:196-254, 464-494).  Output: DFS pre-order rows (node, L, F, R, B); a row that is
// not a real node is -inf.  The env's whole slab is pre-filled with -inf in phase 0 (obs_body), the builder writes the real rows.
// Node-table slots: COMPACT (no direction of a cell of the batch has more than two transitions -- every Flatland rail cell
// type; the host checks the grids): level L has at most 2^L nodes, slot = 2^L - 2 + q with lane q of the team; a depth-3 tree
// has 14 slots, a team is 16 lanes.  Otherwise level L is handled by 4^L lanes and the slot is the node's DFS row.
//
// pass A of one upstream tree: node topology into the team's table scr (wave-level synchronisation only; no output yet)
template <int TEAM, int CAP, bool COMPACT, int STRIDE = CAP>
__device__ __forceinline__ void upstream_pass_a(const ObsCtx &X, const ObsArgs &P, int b, int i, bool have, int tl, int *scr, int *err) {
    const int D = P.max_depth;
    int sz[5];  // sz[l] = nodes of a subtree rooted at depth l
    { int n = 0; for (int l = D; l >= 0; l--) { n = n * 4 + 1; sz[l] = n; } }
    const int ia = have ? i : 0;
    const int vpos = X.a_vpos[ia];
    const uint32_t dir = X.a_dir[ia];
    const uint32_t rbits = nibble(cw_bits(X, vpos), dir);
    uint32_t orientation = dir;
    if (__popc(rbits) == 1) orientation = first_dir(rbits);
    for (int k = tl; k < CAP; k += TEAM) nt_clear_desc(scr, STRIDE, k);
    team_sync();
    constexpr int FAN = COMPACT ? 2 : 4;    // lanes per parent at the next level
    int c_state = -1, c_tot = 1, c_index = -1;
    if (tl < FAN) {  // level 1: the root's branches left, forward, right, back of `orientation`
        int j = tl;
        if (COMPACT) {  // the tl-th branch that exists
            uint32_t m4 = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) m4 |= ((rbits >> (3u - ((orientation + (uint32_t)(q + 3)) & 3u))) & 1u) << q;
            if (__popc(m4) > 2 && err) atomicCAS(err, 0, FL_ERR_CAPACITY);
            if (tl == 1) m4 &= m4 - 1;
            j = m4 ? __ffs((int)m4) - 1 : -1;
        }
        if (j >= 0) {
            const uint32_t bd = (orientation + (uint32_t)(j + 3)) & 3u;
            c_index = 1 + j * sz[1];
            if ((rbits >> (3 - bd)) & 1) c_state = state_towards(X, vpos, bd);
        }
    }
    const uint16_t *dm_t = X.dm + X.a_tslot[ia] * X.SS;  // per-agent constants of the level loop
    const int tgt_r = X.a_target[ia];
    int width = FAN;
    for (int level = 1; level <= D; level++) {
        // what a node hands to its children: their start states as the segment table has them (u16 each, FL_R_NONE = null cell), the
        // tot_dist they start at and its own DFS row -- three words
        uint32_t k01 = 0xFFFFFFFFu, k23 = 0xFFFFFFFFu, ch_tot = 0;
        if (have && tl < width && c_index >= 0 && c_state != -1) {
            const NodeDesc nd = node_topology(X, dm_t, tgt_r, c_state, c_tot);
            nt_store_desc(scr, STRIDE, COMPACT ? width - 2 + tl : c_index, nd, c_index, err);
            ch_tot = (uint32_t)(nd.tot0 + nd.nvis);
            if (nd.flags & (ND_SWITCH | ND_DEAD_END)) { k01 = nd.kids01; k23 = nd.kids23; }
        } else if (tl < width) {
            c_index = -1;  // missing node: its whole subtree stays -inf
        }
        if (level == D) break;
        // children of lane p go to lanes FAN * p .. FAN * p + FAN - 1 of the next level (all lanes take part in the shuffles)
        const int src = tl / FAN, which = tl % FAN;
        const uint32_t s01 = (uint32_t)__shfl((int)k01, src, TEAM), s23 = (uint32_t)__shfl((int)k23, src, TEAM);
        const uint32_t s_ti = (uint32_t)__shfl((int)((ch_tot & 0xFFFFu) | ((uint32_t)(c_index + 1) << 16)), src, TEAM);
        const int p_index = (int)(s_ti >> 16) - 1, s_tot = (int)(s_ti & 0xFFFFu);
        width *= FAN;
        c_index = -1;
        c_state = -1;
        if (tl < width && p_index >= 0) {
            int kk = which;
            if (COMPACT) {  // the which-th child that exists
                const uint32_t m4 = (uint32_t)((s01 & 0xFFFFu) != FL_R_NONE) | ((uint32_t)((s01 >> 16) != FL_R_NONE) << 1) |
                                    ((uint32_t)((s23 & 0xFFFFu) != FL_R_NONE) << 2) | ((uint32_t)((s23 >> 16) != FL_R_NONE) << 3);
                if (which == 0 && __popc(m4) > 2 && err) atomicCAS(err, 0, FL_ERR_CAPACITY);
                const uint32_t m = which == 1 ? (m4 & (m4 - 1)) : m4;
                kk = m ? __ffs((int)m) - 1 : -1;
            }
            if (kk >= 0) {
                const uint32_t c = ((kk < 2 ? s01 : s23) >> (16 * (kk & 1))) & 0xFFFFu;
                c_state = c == FL_R_NONE ? -1 : c == FL_R_PHANTOM ? -2 : (int)c;
                c_tot = s_tot;
                c_index = p_index + 1 + kk * sz[level + 1];
            }
        }
    }
    team_sync();
}

// number of slots pass B / the rows look at
template <bool COMPACT>
__device__ __forceinline__ int upstream_slots(const ObsArgs &P) { return COMPACT ? (2 << P.max_depth) - 2 : P.n_tree_nodes; }

// the real rows of one upstream tree: the root (observations.py:217-229) and the nodes of its table; the rest of the slab is -inf already
// Which rows are real nodes is kept per agent (rowmask: 96 bits, depth <= 3) from launch to launch: a caller that hands over the SAME
// output buffer again, untouched (FL_OBS_KEEP_TREE_ROWS), gets no pre-fill of the slab -- 63 % of cfg5's output bytes are that
// constant -- only the rows that were real then and are not now are set to -inf.  All lanes of the team take part (shuffles).
template <int TEAM, int CAP, bool COMPACT, int STRIDE = CAP>
__device__ __forceinline__ void upstream_rows(const ObsCtx &X, const ObsArgs &P, int b, int i, bool have, int tl, const int *scr, uint4 *rowmask) {
    const int NN = P.n_tree_nodes;
    const int ns = upstream_slots<COMPACT>(P);
    if (rowmask) {
        uint32_t m0 = 0, m1 = 0, m2 = 0;
        if (have) {
            if (tl == 0) m0 = 1u;   // the root
            for (int k = tl; k < ns; k += TEAM) {
                if (nt_start((uint32_t)nt_r(scr, STRIDE, N_SE, k)) < 0) continue;
                const int row = nt_row((uint32_t)nt_r(scr, STRIDE, N_UF, k));
                if (row < 32) m0 |= 1u << row; else if (row < 64) m1 |= 1u << (row - 32); else m2 |= 1u << (row - 64);
            }
        }
#pragma unroll
        for (int off = 1; off < TEAM; off <<= 1) {
            m0 |= (uint32_t)__shfl_xor((int)m0, off, TEAM); m1 |= (uint32_t)__shfl_xor((int)m1, off, TEAM); m2 |= (uint32_t)__shfl_xor((int)m2, off, TEAM);
        }
        if (have) {
            uint4 *mk = rowmask + (size_t)b * X.A + i;
            if (P.keep_rows) {
                const uint4 old = *mk;      // (every lane of the team: one broadcast load)
                uint32_t st[3] = {old.x & ~m0, old.y & ~m1, old.z & ~m2};
                double2 *o2 = reinterpret_cast<double2 *>(P.tree_out + (size_t)(b * X.A + i) * NN * 12);
                const double2 ninf = make_double2(-INFINITY, -INFINITY);
                int n = 0;
#pragma unroll
                for (int w = 0; w < 3; w++)
                    for (uint32_t m = st[w]; m; m &= m - 1, n++)
                        if (n % TEAM == tl) {
                            const int row = w * 32 + __ffs((int)m) - 1;
#pragma unroll
                            for (int q = 0; q < 6; q++) out_store_d2(reinterpret_cast<double *>(o2 + row * 6 + q), -INFINITY, -INFINITY);
                        }
            }
            if (tl == 0) *mk = make_uint4(m0, m1, m2, 1u);
        }
    }
    if (!have) return;
    double *out = P.tree_out + (size_t)(b * X.A + i) * NN * 12;
    if (tl == TEAM - 1) {  // (a lane without a slot in the compact tables)
        const int vpos = X.a_vpos[i];
        const uint16_t dv = X.dm[X.a_tslot[i] * X.SS + vpos * 4 + (int)X.a_dir[i]];
        double root[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        root[6] = dv == FL_INF16 ? INFINITY : (double)dv;
        root[9] = (double)X.a_malf[i];
        root[10] = X.a_speed[i];
        double2 *row = reinterpret_cast<double2 *>(out);
#pragma unroll
        for (int q = 0; q < 6; q++) out_store_d2(reinterpret_cast<double *>(row + q), root[2 * q], root[2 * q + 1]);
    }
    for (int k = tl; k < ns; k += TEAM) {
        if (nt_start((uint32_t)nt_r(scr, STRIDE, N_SE, k)) < 0) continue;
        double f[12];
        node_row<false>(X, i, scr, STRIDE, k, f);
        double2 *row = reinterpret_cast<double2 *>(out + (size_t)nt_row((uint32_t)nt_r(scr, STRIDE, N_UF, k)) * 12);  // rows are 96 B, 16-B aligned
#pragma unroll
        for (int q = 0; q < 6; q++) out_store_d2(reinterpret_cast<double *>(row + q), f[2 * q], f[2 * q + 1]);
    }
}

template <int TEAM, int CAP, bool COMPACT, bool ITL>
__device__ __forceinline__ void tree_upstream(const ObsCtx &X, const FlDev &d, const ObsArgs &P, int b, int wave, int lane,
                                              int nwaves, int *wave_scr0, int *team_meta, uint4 *rowmask) {
    constexpr int TPW = 64 / TEAM;  // teams per wavefront
    constexpr int TW = N_WORDS_T * CAP;
    const int A = X.A;
    const int team = lane / TEAM, tl = lane % TEAM;
    // team t's node table is slot t; teams that can never hold an agent share the dummy slot behind the real ones
    const int n_slots = min(nwaves * TPW, A);
    int *scr = wave_scr0 + min(wave * TPW + team, n_slots) * TW;
    const int ns = upstream_slots<COMPACT>(P);
    for (int base = 0; base < A; base += nwaves * TPW) {
        const int i = base + wave * TPW + team;
        const bool have = i < A;
        upstream_pass_a<TEAM, CAP, COMPACT>(X, P, b, i, have, tl, scr, &d.err[b]);
        TREE_STAMP(X, 6);
        {
            int first;
            const int tot_cells = team_prepare<TEAM, CAP, true>(have, tl, have ? ns : 1, scr, first);
            const int team_id = wave * TPW + team;
            if (tl == 0) { team_meta[team_id] = have ? tot_cells : 0; team_meta[64 + team_id] = have ? ns : 1; team_meta[128 + team_id] = have ? i : -1; team_meta[256 + team_id] = first; }
        }
        wg_pass_b<0, CAP, ITL>(X, wave * 64 + lane, nwaves * 64, nwaves * TPW, wave_scr0, TW, team_meta);
        TREE_STAMP(X, 7);
        upstream_rows<TEAM, CAP, COMPACT>(X, P, b, i, have, tl, scr, rowmask);
        team_sync();
        TREE_STAMP(X, 8);
    }
}

// Pass A of one flatland_cutils tree (treeobs.cpp:154-256): root row, node topology level by level (BFS), one team of 32
// lanes per agent, two teams per wavefront.  Only wave-level synchronisation, so a wavefront can run it whenever the
// rail bitmap and the agent snapshot are in LDS (the workgroup overlaps it with the path walk of phase 2).
// TC = slots of the node table: 32 (max_nodes <= 32, the solution's 31) or 64 (max_nodes up to 64, the stand-alone flatland_cutils launch
// only).  TEAM = lanes of the team: TC (two trees / one tree a wavefront), or 16 for 32-slot tables on maps whose cells have at most two
// transitions a direction (FlDev::max_branch, every Flatland rail cell type) -- FOUR trees a wavefront: an explored node has then at most
// two real children of three, so a level of size s needs s - 2 nodes before it and the 31 nodes of a tree leave no level more than 12
// (3, 6, 12, then at most 9); pass A is bound by instruction issue, half the wavefronts issue half the instructions.  grp / gl: the team
// inside the wavefront and the lane inside the team.
template <int TC = OBS_CAP_C, int TEAM = TC>
__device__ __forceinline__ void cutils_pass_a(const ObsCtx &X, const FlDev &d, const ObsArgs &P, int b, int i, bool have, int grp,
                                              int gl, int *scr, const uint16_t *a_vpos, const int *a_pos,
                                              const uint8_t *a_dir, const uint8_t *a_state, const double *a_speed,
                                              const uint16_t *a_tslot, float max_dist, uint32_t spk, uint32_t malfw,
                                              int &node_base_out, int &levels_out) {
    constexpr int CAP = TC;
    const int A = X.A, N = P.max_nodes;
    const int ia = have ? i : 0;
    const int g = b * A + ia;
    const int vpos = a_vpos[ia];
    const uint32_t dir = a_dir[ia];
    const uint32_t rbits = nibble(cw_bits(X, vpos), dir);
    uint32_t orientation = dir;
    if (__popc(rbits) == 1) orientation = first_dir(rbits);
    float *F = P.forest + (size_t)g * N * 12;
#pragma unroll
    for (int k = gl; k < CAP; k += TEAM) {
        nt_clear_desc(scr, CAP, k);
        // N_PH: parent + 2 | (first child's node index << 2 | action + 1) << 8; the root: no parent, first child = node 1
        nt_w(scr, CAP, N_PH, k) = k == 0 ? (((1 << 2) | 1) << 8) : 0;
    }
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
    while (true) {  // pass A
        levels++;
        const bool active = have && node_base < N && n_cur > 0;
        if (!__any(active)) break;  // wave-uniform: both teams take part in the shuffles below
        const int m = active ? min(n_cur, N - node_base) : 0;
        if (TEAM < TC && m > TEAM) atomicCAS(&d.err[b], 0, FL_ERR_CAPACITY);   // (cannot happen with at most two transitions a direction)
        const bool mine = gl < m;
        const int idx_node = node_base + gl;
        // what an explored node hands to its three children: their start states as the segment table has them (u16 each, FL_R_NONE =
        // null cell) and the tot_dist they start at -- two words
        uint32_t k01 = 0xFFFFFFFFu, k2t = 0xFFFFu;
        bool explored = false;
        if (mine && c_state != -1) {
            const NodeDesc nd = node_topology(X, dm_t, tgt_r, c_state, c_tot);
            explored = true;
            const bool kids = (nd.flags & (ND_SWITCH | ND_DEAD_END)) != 0;
            k01 = kids ? nd.kids01 : 0xFFFFFFFFu;
            k2t = (kids ? (nd.kids23 & 0xFFFFu) : 0xFFFFu) | ((uint32_t)(nd.tot0 + nd.nvis) << 16);  // children start one step beyond the end of this walk
            nt_store_desc(scr, CAP, idx_node, nd, 0, &d.err[b]);
        }
        // explored lanes of the team, and how many of them below this lane (teams of 32: the mask is the team's half of the ballot)
        int n_next, my_rank;
        if (TEAM == 64) {
            const unsigned long long em = __ballot(explored);
            n_next = 3 * __popcll(em);
            my_rank = __popcll(em & ((1ull << gl) - 1ull));
        } else {
            const unsigned long long bal = __ballot(explored);
            const uint32_t em = TEAM == 32 ? (grp ? (uint32_t)(bal >> 32) : (uint32_t)bal) : ((uint32_t)(bal >> (grp * TEAM)) & ((1u << (TEAM & 31)) - 1u));
            n_next = 3 * __popc(em);
            my_rank = __popc(em & ((1u << gl) - 1u));
        }
        if (mine) {  // first child's node index (children are numbered consecutively) << 2 | action + 1
            const int fc = explored ? node_base + m + 3 * my_rank : 0;
            nt_w(scr, CAP, N_PH, idx_node) = (c_parent + 2) | (((fc << 2) | (c_act + 1)) << 8);
        }
        // hand the children to the next level's lanes: lane j takes child j % 3 of the (j / 3)-th explored lane.  Which lane that is:
        // every explored lane PUSHES its number to the lane of its rank (ds_permute; the others push to the team's last lane, whose
        // rank no explored lane has: fewer nodes than lanes are explored), lane j reads it from lane j / 3 -- two cross-lane moves
        // instead of a loop over the set bits of the mask (pass A is bound by instruction issue: a few nodes a level on 32 lanes).
        const int src_rank = gl / 3, which = gl - 3 * src_rank;
        const int tbase = (int)__lane_id() - gl;
        const int pushed = __builtin_amdgcn_ds_permute((tbase + (explored ? my_rank : TEAM - 1)) << 2, gl);
        const int src_of_rank = __shfl(pushed, src_rank, TEAM);
        const int src = (gl < n_next) ? src_of_rank : 0;
        const uint32_t s_k01 = (uint32_t)__shfl((int)k01, src, TEAM), s_k2t = (uint32_t)__shfl((int)k2t, src, TEAM);
        if (active) {
            const int parent_base = node_base;
            node_base += m;
            n_cur = n_next;
            if (gl < n_next) {
                const uint32_t c = which == 0 ? (s_k01 & 0xFFFFu) : (which == 1 ? (s_k01 >> 16) : (s_k2t & 0xFFFFu));
                c_state = c == FL_R_NONE ? -1 : c == FL_R_PHANTOM ? -2 : (int)c;
                c_parent = parent_base + src;
                c_tot = (int)(s_k2t >> 16);
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
// I64 (the launches of the builder alone): when P.out64 is set the three index tensors are written as the POLICY takes them -- int64, the
// adjacency's parent / child columns offset by tree * max_nodes with tree = b * A + agent, every negative entry (padding, action -1) -2 (plfActor.get_feature's casts +
// Network.modify_adjacency, solution/plfActor.py:48-74, nn/net_tree.py:105-116) -- instead of int32 + a second kernel (fl_policy_pack).
template <int TC = OBS_CAP_C, bool I64 = false>
__device__ __forceinline__ void cutils_rows_orders(const ObsCtx &X, const FlDev &d, const ObsArgs &P, int b, int i, bool have, int gl,
                                                   const int *scr, int node_base, int levels, float max_dist) {
    constexpr int CAP = TC;
    const int A = X.A, N = P.max_nodes;
    const int g = b * A + (have ? i : 0);
    float *F = P.forest + (size_t)g * N * 12;
    int32_t *ADJ = P.adjacency + (size_t)g * (N - 1) * 3;
    const bool w64 = I64 && P.out64 != 0;
    long long *ADJ64 = reinterpret_cast<long long *>(P.adjacency) + (size_t)g * (N - 1) * 3;
    const long long tree_off = (long long)g * N;
    if (have) {  // rows: lane gl writes node gl + 1
        for (int idx = gl + 1; idx < N; idx += TC) {
            int32_t *adj = ADJ + (size_t)(idx - 1) * 3;
            long long *adj64 = ADJ64 + (size_t)(idx - 1) * 3;
            if (idx < node_base) {
                const uint32_t ph = (uint32_t)nt_r(scr, CAP, N_PH, idx);
                if (w64) { out_store(&adj64[0], tree_off + (long long)((int)(ph & 0xFFu) - 2)); out_store(&adj64[1], tree_off + (long long)idx); const int act = (int)((ph >> 8) & 3u) - 1; out_store(&adj64[2], act < 0 ? -2ll : (long long)act); }   // (adjacency[adjacency < 0] = -2 takes the action column too)
                else {
                out_store(&adj[0], (int32_t)((int)(ph & 0xFFu) - 2)); out_store(&adj[1], (int32_t)idx); out_store(&adj[2], (int32_t)((int)((ph >> 8) & 3u) - 1));
                }
                if (nt_start((uint32_t)nt_r(scr, CAP, N_SE, idx)) < 0) {
                    const double nn[12] = {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, -1, -1, -1, -1, -1};
                    scale_and_store(nn, max_dist, A, F + (size_t)idx * 12);
                } else {
                    double f[12];
                    node_row<true>(X, i, scr, CAP, idx, f);
                    if (nt_flags((uint32_t)nt_r(scr, CAP, N_UF, idx)) & ND_ZERO) atomicCAS(&d.err[b], 0, FL_ERR_ZERO_TRANSITION);  // treeobs.cpp:529-535 throws
                    scale_and_store(f, max_dist, A, F + (size_t)idx * 12);
                }
            } else {  // padding rows when the queue ran dry (treeobs.cpp:268-276, 245-249)
                const double nn[12] = {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, INFINITY, -1, -1, -1, -1, -1};
                scale_and_store(nn, max_dist, A, F + (size_t)idx * 12);
                if (w64) { out_store(&adj64[0], -2ll); out_store(&adj64[1], -2ll); out_store(&adj64[2], -2ll); }
                else { out_store(&adj[0], (int32_t)-2); out_store(&adj[1], (int32_t)-2); out_store(&adj[2], (int32_t)-2); }
            }
        }
    }
    // calculate_evaluation_orders (tool.h:468-524): order = height above the leaves.  Lane k holds node k; a node's
    // children are consecutive nodes, so heights settle after as many shuffle rounds as the tree has levels.
    {
        const uint32_t ph = gl < node_base ? (uint32_t)nt_r(scr, CAP, N_PH, gl) : 0u;
        const int fc = (int)(ph >> 10);      // 0 = no children pushed
        const int parent = gl < node_base ? (int)(ph & 0xFFu) - 2 : -2;
        const int nchild = fc > 0 ? max(0, min(3, node_base - fc)) : 0;  // children beyond max_nodes were never popped
        // `levels` counted the rounds of pass A including the one that found nothing left: a tree of L levels below the root
        // needs L rounds here (a leaf is 0, every round carries the heights one level up)
        const int max_levels = TC == 64 ? __builtin_amdgcn_readlane(levels, 0) : max(__builtin_amdgcn_readlane(levels, 0), __builtin_amdgcn_readlane(levels, 32));
        int h = 0;
        for (int it = 0; it + 1 < max_levels; it++) {
            const int h0 = __shfl(h, fc, TC), h1 = __shfl(h, fc + 1, TC), h2 = __shfl(h, fc + 2, TC);
            int hn = 0;
            if (nchild > 0) hn = h0 + 1;
            if (nchild > 1) hn = max(hn, h1 + 1);
            if (nchild > 2) hn = max(hn, h2 + 1);
            h = hn;
        }
        const int hp = __shfl(h, parent < 0 ? 0 : parent, TC);
        if (have) {
            int32_t *NO = P.node_order + (size_t)g * N, *EO = P.edge_order + (size_t)g * (N - 1);
            if (gl < N && w64) {
                out_store(reinterpret_cast<long long *>(P.node_order) + (size_t)g * N + gl, (long long)(gl < node_base ? h : -2));
                if (gl >= 1) out_store(reinterpret_cast<long long *>(P.edge_order) + (size_t)g * (N - 1) + gl - 1, (long long)((gl >= node_base || parent < 0) ? -2 : hp));
            } else if (gl < N) {
                out_store(&NO[gl], (int32_t)(gl < node_base ? h : -2));
                if (gl >= 1) out_store(&EO[gl - 1], (int32_t)((gl >= node_base || parent < 0) ? -2 : hp));
            }
        }
    }
}

// flatland_cutils trees (treeobs.cpp:154-256): two agents per wavefront, a team of 32 lanes each
template <bool ITL, int TC = OBS_CAP_C, bool I64 = false>
__device__ __forceinline__ void trees_cutils(const ObsCtx &X, const FlDev &d, const ObsArgs &P, int b, int wave, int lane,
                                             int nwaves, int *wave_scr, int *team_meta,
                                             const uint16_t *a_vpos, const int *a_pos, const uint8_t *a_dir,
                                             const uint8_t *a_state, const double *a_speed, const uint16_t *a_tslot,
                                             float max_dist, bool hoisted) {
    constexpr int CAP = TC, TW = N_WORDS_C * TC, TPW = 64 / TC;   // teams (trees) per wavefront
    const int A = X.A;
    const int grp = TC == 64 ? 0 : lane >> 5, gl = lane & (TC - 1);
    // team t's node table is slot t (wg_pass_b); teams that can never hold an agent share the dummy slot behind the real ones
    int *scr = wave_scr + min(wave * TPW + grp, min(nwaves * TPW, A)) * TW;
    for (int base = 0; base < A; base += nwaves * TPW) {
        const int i = base + wave * TPW + grp;
        const bool have = i < A;
        int node_base, levels;
        if (hoisted && base == 0) {  // pass A of the first round already ran beside the path walk (32-lane teams only)
            node_base = team_meta[64 + wave * TPW + grp];
            levels = team_meta[192 + wave * TPW + grp];
        } else {
            cutils_pass_a<TC>(X, d, P, b, i, have, grp, gl, scr, a_vpos, a_pos, a_dir, a_state, a_speed, a_tslot, max_dist,
                              d.spk[b * A + (have ? i : 0)], d.malf[b * A + (have ? i : 0)], node_base, levels);
        }
        TREE_STAMP(X, 6);
        {
            int first;
            const int tot_cells = team_prepare<TC, CAP, false>(have, gl, have ? node_base : 1, scr, first);
            const int team_id = wave * TPW + grp;
            if (gl == 0) { team_meta[team_id] = have ? tot_cells : 0; team_meta[64 + team_id] = have ? node_base : 1; team_meta[128 + team_id] = have ? i : -1; team_meta[256 + team_id] = first; }
        }
        wg_pass_b<1, CAP, ITL, NoLateWork, !ITL>(X, wave * 64 + lane, nwaves * 64, nwaves * TPW, wave_scr, TW, team_meta);   // (items in HBM: a query may be two pieces)
        TREE_STAMP(X, 7);
        cutils_rows_orders<TC, I64>(X, d, P, b, i, have, gl, scr, node_base, levels, max_dist);
        team_sync();
        TREE_STAMP(X, 16);
    }
}

// Fused launch, both builders in ONE pass B per round (compact upstream trees; stage 1 built the upstream predictor's index
// too).  Round r covers the agents [32 r, 32 r + 32).  Pass B teams of a round: t in [0, 32) = the flatland_cutils tree of agent
// 32 r + t (a team of 32 lanes, wavefront t / 2), 32 + u = the upstream tree of agent 32 r + u (a team of 16 lanes, four trees
// a wavefront, from the last wavefront down so that on a small env they land on wavefronts without a cutils tree).
// Pass A of the first round ran beside the path walk (obs_body).
__device__ __forceinline__ int *merged_table_c(int *wave_scr, int t) { return wave_scr + t * (N_WORDS_C * 32); }
// (two compact upstream trees share a table whose fields are 32 words apart, see team_table)
template <int ROUND>
__device__ __forceinline__ int *merged_table_t(int *wave_scr, int u) {
    return wave_scr + ROUND * (N_WORDS_C * 32) + (u >> 1) * (N_WORDS_T * 32) + (u & 1) * 16;
}
// the upstream tree a lane works on in a merged round: u in [0, ROUND) or -1
template <int ROUND>
__device__ __forceinline__ int merged_upstream_of(int wave, int lane, int nwaves) {
    const int u = (nwaves - 1 - wave) * 4 + (lane >> 4);
    return u < ROUND ? u : -1;
}

// MULTI = false: an env of at most 32 agents -- one round, whose pass A ran beside the path walk: no pass A code here.
// ROUND = agents of a round = 2 * wavefronts of the workgroup: 32 (1024 threads) or 16 (512 threads, two workgroups a CU).
// UP = false: the flatland_cutils trees alone (pass B PB 3): no upstream tables, teams or rows.
template <bool ITL, bool MULTI, int ROUND, bool UP, typename LATE>
__device__ __forceinline__ void trees_merged(ObsCtx &X, const FlDev &d, const ObsArgs &P, int b, int wave, int lane, int nwaves,
                                             int *wave_scr, int *team_meta, const uint16_t *a_vpos, const int *a_pos, const uint8_t *a_dir,
                                             const uint8_t *a_state, const double *a_speed, const uint16_t *a_tslot, float max_dist,
                                             const LATE &late, uint4 *rowmask) {
    constexpr int CT = OBS_CAP_T_COMPACT;
    const int A = X.A;
    const int grp = lane >> 5, gl = lane & 31, ct = wave * 2 + grp;   // (ct covers 0 .. ROUND - 1)
    int *scr_c = merged_table_c(wave_scr, min(ct, ROUND - 1));
    const int u = UP ? merged_upstream_of<ROUND>(wave, lane, nwaves) : -1, tl = lane & 15;
    const bool wave_has_u = UP && (nwaves - 1 - wave) * 4 < ROUND;  // wave-uniform
    int *scr_u = merged_table_t<ROUND>(wave_scr, u < 0 ? 0 : u);
    const int ns = upstream_slots<true>(P);
    for (int base = 0; base < (MULTI ? A : 1); base += ROUND) {
        X.round_base = base;
        const int i_c = base + ct, i_u = base + u;
        const bool have_c = ct < ROUND && i_c < A, have_u = u >= 0 && i_u < A;
        int node_base = 1, levels = 0;
        if (!MULTI || base == 0) {  // pass A of the first round already ran beside the path walk
            if (have_c) { node_base = team_meta[64 + ct]; levels = team_meta[192 + ct]; }
        } else {
            if (ct < ROUND)
                cutils_pass_a(X, d, P, b, i_c, have_c, grp, gl, scr_c, a_vpos, a_pos, a_dir, a_state, a_speed, a_tslot, max_dist,
                              d.spk[b * A + (have_c ? i_c : 0)], d.malf[b * A + (have_c ? i_c : 0)], node_base, levels);
            if (wave_has_u) upstream_pass_a<16, CT, true, 32>(X, P, b, i_u, have_u, tl, scr_u, &d.err[b]);
        }
        TREE_STAMP(X, 6);
        if (ct < ROUND) {
            int first;
            const int cells = team_prepare<32, OBS_CAP_C, false>(have_c, gl, have_c ? node_base : 1, scr_c, first);
            if (gl == 0) { team_meta[ct] = have_c ? cells : 0; team_meta[64 + ct] = have_c ? node_base : 1; team_meta[192 + ct] = levels; team_meta[256 + ct] = first; }
        }
        if (wave_has_u) {
            int first;
            const int cells = team_prepare<16, CT, true, 32>(have_u, tl, have_u ? ns : 1, scr_u, first);
            const int id = ROUND + (u < 0 ? 0 : u);
            if (u >= 0 && tl == 0) { team_meta[id] = have_u ? cells : 0; team_meta[64 + id] = have_u ? ns : 1; team_meta[256 + id] = first; }
        }
        wg_pass_b<UP ? 2 : 3, OBS_CAP_C, ITL, LATE, MULTI>(X, wave * 64 + lane, nwaves * 64, UP ? 2 * ROUND : ROUND, wave_scr, 0, team_meta, late);
        if (base == 0) late();  // (whatever the queue still holds)
        TREE_STAMP(X, 7);
        if (ct < ROUND) cutils_rows_orders<OBS_CAP_C, !UP>(X, d, P, b, i_c, have_c, gl, scr_c, node_base, levels, max_dist);
        if (wave_has_u) upstream_rows<16, CT, true, 32>(X, P, b, i_u, have_u, tl, scr_u, rowmask);
        team_sync();
        TREE_STAMP(X, 16);
    }
}
