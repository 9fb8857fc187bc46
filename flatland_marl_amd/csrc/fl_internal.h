// fl_internal.h -- device-side data layout and helpers shared by the HIP kernels (gfx950 only).
//
// HBM layout: struct-of-arrays over (env b, agent i), g = b * A + i.
//   dynamic (20 B/agent): pos i32 (cell id r * W + c, -1 = None), old_pos i32, arrival i32,
//                         malf u32 (lo16 down counter, hi16 num_malfunctions), pk u32 (packed small fields, see PK_* below)
//   static  (40 B/agent): init_pos i32, target i32 (cell ids, read by the step), init_r u16, target_r u16 (rail indices, read by
//                         the observation kernels), earliest i32, latest i32, spk u32, tslot i32, speed f64
//   per env: t, T, done_all, mt_pos, mt[624], malf_thr u64, malf_min/max, U, R, K, grid u32[H*W] (step only)
//
// RAIL-CELL INDEX SPACE.  Only 12-20 % of the cells of a Flatland map carry rail, so every static table of the observation
// path is indexed by the env's rail cells in row-major order (rail index r in [0, R)) and by rail states s = r * 4 + o
// (o = orientation), not by grid cells: ridx maps a cell to its rail index, nbr gives the rail index of the neighbour
// in each direction.  At 150x150 (R = 2680 of 22500 cells) the tables are 8.4x smaller than their dense form:
//   rgrid u16[Rcap] transitions, nbr u16[Rcap*4], snext u16[Scap] (successor of a single-transition state),
//   dm u16[Ucap][Scap] (distance per unique target and rail state, 0xFFFF = unreachable), seg uint4[Scap], nh u16[Ucap][Rcap],
//   hop8 u16[Ucap][Scap].  Rcap / Ucap = capacity of the batch (largest env, or fl_reserve), Scap = 4 * Rcap <= 65532.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FL_INF16 0xFFFFu

#define FL_R_NONE 0xFFFFu  /* no rail cell / no such state (u16 tables) */

// static branch-walk table entry (uint4) of a start state: the walk of _explore_branch ignoring the agent's own target.
// x = end rail state | kind << 20;  y = steps to the end | first "unusable switch" offset << 16 (0xFFFF = none);
// z, w = the start states of the end state's children left, forward | right, back (u16 each, FL_R_NONE = null cell):
// treeobs.cpp:583-608 / observations.py:464-489 are a function of the end state alone
// A transition that leaves the rail (a malformed map): the reference walks onto the empty cell, finds no transition there and
// ends the walk as on a zero-transition cell (treeobs.cpp:528-535 throws, observations.py:420-425 makes the node terminal), with
// the distance of that cell: inf.  Chain: kind SEG_ZERO + SEG_PHANTOM (the walk's last rail cell is the end state); child of a
// switch / dead end: start state FL_R_PHANTOM (a node without cells).
enum { SEG_SWITCH = 0, SEG_DEAD_END = 1, SEG_ZERO = 2, SEG_CYCLE = 3 };
#define SEG_PHANTOM(e) (((e).x >> 22) & 1u)
#define FL_R_PHANTOM 0xFFFEu
#define SEG_END(e) ((int)((e).x & 0xFFFFFu))
#define SEG_KIND(e) (((e).x >> 20) & 3u)
#define SEG_LEN(e) ((int)((e).y & 0xFFFFu))
#define SEG_UNUS(e) ((int)((e).y >> 16))

enum { ST_WAITING = 0, ST_READY = 1, ST_MALF_OFF = 2, ST_MOVING = 3, ST_STOPPED = 4, ST_MALF = 5, ST_DONE = 6 };
enum { ACT_NOTHING = 0, ACT_LEFT = 1, ACT_FORWARD = 2, ACT_RIGHT = 3, ACT_STOP = 4 };

// pk bit fields
#define PK_DIR(pk) ((pk)&3u)
#define PK_OLD_DIR(pk) (((pk) >> 2) & 7u) /* 4 = None */
#define PK_STATE(pk) (((pk) >> 5) & 7u)
#define PK_PREV(pk) (((pk) >> 8) & 7u) /* 7 = None */
#define PK_SAVED(pk) (((pk) >> 11) & 3u)
#define PK_SCOUNT(pk) (((pk) >> 13) & 63u) /* SpeedCounter.counter: 0 .. max_count <= 63 (speeds down to 1/64) */
#define PK_SIGMALF(pk) (((pk) >> 19) & 1u)
#define PK_DEADLOCK(pk) (((pk) >> 20) & 1u)
#define PK_DEADLOCK_BIT (1u << 20)
#define PK_DONE(pk) (((pk) >> 21) & 1u)
__host__ __device__ inline uint32_t pk_make(uint32_t dir, uint32_t old_dir, uint32_t state, uint32_t prev, uint32_t saved,
                                            uint32_t scount, uint32_t sig, uint32_t dead, uint32_t done) {
    return dir | (old_dir << 2) | (state << 5) | (prev << 8) | (saved << 11) | (scount << 13) | (sig << 19) | (dead << 20) |
           (done << 21);
}
// spk: init_dir bits 0-1, max_count bits 2-7
#define SPK_INIT_DIR(s) ((s)&3u)
#define SPK_MAX_COUNT(s) (((s) >> 2) & 63u)

struct FlDev {
    int B, A, H, W;
    int Ucap, Rcap;  // capacity per env: unique targets, rail cells (rail states: 4 * Rcap)
    int max_branch;  // most transitions any (rail cell, direction) of the batch has: 2 on every Flatland rail cell type (the upstream
                     // trees then use compact node tables, fl_obs_layout.h)
    // per env
    int *t;
    int *T;
    uint8_t *done_all;
    int *mt_pos;
    uint32_t *mt;  // [B][624]
    uint64_t *malf_thr;
    int *malf_min, *malf_max;
    int *U;    // [B] unique targets
    int *R;    // [B] rail cells
    int *K;    // [B] prediction keys (R, or the distinct col * W + row values of the rail cells when H > W)
    int *tab;  // [B] the env whose slabs hold THIS env's static tables (grid, ridx, rgrid, rtype, nbr, snext, rkey, ut_r, dm, seg, nh, hop8):
               // envs with the same rail grid and the same unique targets share one set (round 5) -- a batch is usually many
               // replicas over a pool of a few maps, and one set per map stays in the L2 where one per env streams from HBM.
               // tab[b] <= b; tab[b] == b: the env holds its own.  Every env has its slabs; only the owners' are built and read.
    int *err;  // [B] first error code raised by a kernel for env b (0 = none)
    int *env_list;  // [B + 1] scratch of the table (re)builds: the envs being rebuilt, their number at [B] (k_env_list)
    long long *metrics;  // [B][4] running sums: terminal rewards, arrived agents, agent-steps, finished episodes
    int *last_episode;   // [B][2] sum of rewards and arrived agents of the env's last finished episode
    double *score_sums;  // [B][3] running sums over the env's finished episodes: normalized reward 1 + R / (T * A), arrived / A, their number
                         // (flatland/evaluators/service.py:875-879, 900-913)
    uint32_t *grid;   // [B][H*W] per cell: transition bitmap (lo16) | bit 16 + m: the neighbour towards m is on the map and has rail (step kernel)
    uint16_t *ridx;   // [B][H*W] rail index of a cell, FL_R_NONE = no rail
    uint16_t *rgrid;  // [B][Rcap] transition bitmap per rail cell
    uint8_t *rtype;   // [B][Rcap] flatland_cutils road_type 0..10 of the cell (loader.cpp:122-161: index of the matching basic transition)
    uint16_t *nbr;    // [B][Rcap*4] rail index of the neighbour of rail cell r in direction m (N,E,S,W), FL_R_NONE = none
    uint16_t *snext;  // [B][Scap] successor state of a state with exactly one transition (k_segments), FL_R_NONE otherwise
    uint16_t *rkey;   // [B][Rcap] compact prediction key of a rail cell; nullptr when H <= W (key = rail index)
    uint16_t *ut_r;   // [B][Ucap] rail index of unique target u
    uint16_t *dm;     // [B][Ucap][Scap] distance map
    uint4 *seg;       // [B][Scap] static branch-walk table, see fl_dmap.hip k_segments
    uint16_t *nh;     // [B][Ucap][Rcap] next hop of the greedy distance-map descent: 3 bits per orientation (4 = none)
    uint16_t *hop8;   // [B][Ucap][Scap] state after eight greedy hops, FL_R_NONE if the path ends earlier (k_hop8)
    // static per agent
    int *init_pos, *target, *earliest, *latest, *tslot;
    uint16_t *init_r, *target_r;
    uint16_t *srank;  // number of agents of the env with a smaller speed (the observation kernels take minima over speeds as ranks)
    uint32_t *spk;
    double *speed;
    // dynamic per agent
    int *pos, *old_pos, *arrival;
    uint32_t *malf, *pk;
};

__device__ __forceinline__ bool is_off_map(uint32_t s) { return s <= ST_MALF_OFF; }
__device__ __forceinline__ bool is_on_map(uint32_t s) { return s >= ST_MOVING && s <= ST_MALF; }
// 4-bit transition nibble of a 16-bit cell for an agent facing dir: bit 3 = N, 2 = E, 1 = S, 0 = W
__device__ __forceinline__ uint32_t nibble(uint32_t cell, uint32_t dir) { return (cell >> ((3u - dir) * 4u)) & 15u; }
__device__ __forceinline__ uint32_t tbit(uint32_t cell, uint32_t dir, uint32_t m) { return (nibble(cell, dir) >> (3u - m)) & 1u; }
// first set transition in N,E,S,W order of a non-zero nibble
__device__ __forceinline__ uint32_t first_dir(uint32_t bits) { return (uint32_t)__clz((int)bits) - 28u; }
__device__ __forceinline__ int step_cell(int cell, uint32_t dir, int W) {
    // N(-1,0) E(0,1) S(1,0) W(0,-1), branch-free: sign = +1 for N/E, -1 for S/W; magnitude -W for N/S, 1 for E/W
    const int sgn = 1 - (int)(dir & 2u), ew = (int)(dir & 1u);
    return cell + __mul24(sgn, __mul24(ew, W + 1) - W);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// murmur3 finaliser; synthetic action stream shared with flatland_marl_amd/synth.py
__host__ __device__ inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint32_t action_hash(uint32_t seed, uint32_t b, uint32_t t, uint32_t a) {
    uint32_t h = seed * 0x9E3779B1u;
    h = mix32(h ^ (b * 0x85EBCA77u));
    h = mix32(h ^ (t * 0xC2B2AE3Du));
    h = mix32(h ^ (a * 0x27D4EB2Fu));
    return h;
}
__host__ __device__ inline uint32_t synth_action(uint32_t seed, uint32_t b, uint32_t t, uint32_t a, int kind) {
    uint32_t h = action_hash(seed, b, t, a);
    if (kind == 0) return h % 5u;
    if (kind == 2) return h;  // shortest-path following: the step kernel derives the action from the agent's state (fl_step_body.h)
    uint32_t r = h % 100u;
    return r >= 95 ? 0u : r >= 90 ? 4u : r >= 85 ? 3u : r >= 80 ? 1u : 2u;
}

// kernel launchers (defined in the .hip files).  mask_dev: u8[B] or nullptr (= every env): only the envs with a non-zero
// entry are rebuilt.  fl_launch_env_list turns the mask into d.env_list and goes first.
// (through_tab: list the OWNERS of the masked envs' tables, FlDev::tab -- duplicates allowed, a table is then rebuilt with the same values by several workgroups)
void fl_launch_env_list(const FlDev &d, const uint8_t *mask_dev, hipStream_t s, bool through_tab = false);
void fl_launch_distance_maps(const FlDev &d, const uint8_t *mask_dev, hipStream_t s);
void fl_launch_segments(const FlDev &d, const uint8_t *mask_dev, hipStream_t s);   // seg + snext
void fl_launch_nexthop(const FlDev &d, const uint8_t *mask_dev, hipStream_t s);
void fl_launch_hop8(const FlDev &d, const uint8_t *mask_dev, hipStream_t s);       // after fl_launch_nexthop
int fl_dmap_prepare(const FlDev &d);                                               // LDS attribute / size check
int fl_dmap_fits(const FlDev &d);                                                  // the size check alone
void fl_launch_metrics(const FlDev &d, long long *out4, int reset, hipStream_t s);
void fl_launch_scores(const FlDev &d, double *out3, int reset, hipStream_t s);
void fl_launch_policy_pack(int B, int A, int E, const int32_t *adj, const int32_t *no, const int32_t *eo, long long *adj_out,
                           long long *no_out, long long *eo_out, hipStream_t s);
void fl_launch_info(const FlDev &d, uint8_t *action_required, int32_t *malfunction, uint8_t *state, double *scores, hipStream_t s);
void fl_launch_reset(const FlDev &d, const uint8_t *mask_dev, int fresh, hipStream_t s);
void fl_launch_motion_check(int n_cases, int max_agents, const int *offsets, const int *cur, const int *nxt, uint8_t *can_move,
                            hipStream_t s);
size_t fl_step_lds_bytes(int A);
int fl_step_prepare();
void fl_launch_step(const FlDev &d, const uint8_t *actions, uint32_t seed, uint32_t stream_base, int synth_kind,
                    int32_t *rewards, uint8_t *dones, uint8_t *done_all, int auto_reset, hipStream_t s);
