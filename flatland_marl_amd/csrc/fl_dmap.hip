// fl_dmap.hip -- static per-env tables in rail-state space (s = rail index * 4 + orientation), built on the GPU from the
// resident rail bitmap at commit, at a live map replacement and at fl_distance_map_rebuild (optionally for a masked
// subset of the envs, e.g. the envs that just reset):
//   k_distance_map  reverse BFS from every unique target over rail states.  Replaces DistanceMap._compute /
//                   _distance_map_walker / _get_and_update_neighbors (flatland-rl/flatland/envs/distance_map.py:57-160).
//   k_segments      branch-walk table + successor table (treeobs.cpp:322-539 / observations.py:296-425 are deterministic
//                   chains between switches)
//   k_nexthop       greedy strict descent choice per (target, rail cell, orientation)
//                   (predictions.cpp:78-144, rail_env_shortest_paths.py:203-274)
//   k_hop8          eight next-hops at once
//
// Only rail states are stored and written: the distance-map slab of one (env, target) is 8 * R bytes (21 KB at 150x150
// with R = 2680) instead of 8 * H * W (180 KB), every address is written exactly once.
#include "fl_internal.h"
#include "../../include/flatland_hip.h"

#define DM_QCAP 4096       /* BFS ring buffer per target: states discovered and not yet expanded */
#define DM_WAVES 4         /* targets (wavefronts) per workgroup; they share the LDS copy of the env's nbr / rgrid */

// Which envs a (re)build covers: list[0 .. *count) = the envs with mask[b] != 0 (all envs without a mask), in ascending
// order.  The table kernels loop over (listed env, piece of work) pairs with a grid that does not depend on the count, so
// a masked rebuild in the step loop -- usually no env or one -- costs four near-empty launches, not four launches of
// B * Ucap * ... workgroups that exit at once.
__global__ __launch_bounds__(1024) void k_env_list(int B, const uint8_t *__restrict__ mask, const int *__restrict__ tab, int *__restrict__ list, int *__restrict__ count) {
    __shared__ int wave_tot[16];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += 1024) {
        const int b = b0 + (int)threadIdx.x;
        const bool on = b < B && (!mask || mask[b] != 0);
        const unsigned long long m = __ballot(on);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; w++) off += wave_tot[w];
        if (on) list[off + __popcll(m & ((1ull << lane) - 1ull))] = tab ? tab[b] : b;   // (tab: the owner of env b's tables)
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; w++) t += wave_tot[w]; base += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base;
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// d[s] = 1 + min over the transitions (o -> m) of s = (r, o) of d[(nbr(r, m), m)], d[target, *] = 0.  One WAVEFRONT per
// (env, unique target): the BFS is a chain of short levels (a handful of states each), so it is latency-bound and a
// wavefront has no barriers to pay; its queue is a ring in LDS, the visited set a bitmap in LDS, the env's neighbour
// table and rail bitmap are staged in LDS once per workgroup.  Distances go straight to HBM, unreached states get
// 0xFFFF at the end (each address written once).
__global__ __launch_bounds__(64 * DM_WAVES) void k_distance_map(FlDev d, const int *__restrict__ env_list, const int *__restrict__ env_count) {
    const int per = (d.Ucap + DM_WAVES - 1) / DM_WAVES;
    const int n_work = *env_count * per;
    extern __shared__ uint32_t lds[];
    for (int work = blockIdx.x; work < n_work; work += gridDim.x) {
    const int b = env_list[work / per], u0 = (work % per) * DM_WAVES;
    const int U = d.U[b];
    if (u0 >= U) continue;   // (block-uniform)
    const int R = d.R[b], S = R * 4, Scap = d.Rcap * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint16_t *nbr = reinterpret_cast<uint16_t *>(lds);                  // [Scap]
    uint16_t *rg = nbr + Scap;                                          // [Rcap] (+ pad to a word)
    const int bw = (Scap + 31) / 32;
    uint32_t *bitmaps = reinterpret_cast<uint32_t *>(rg + ((d.Rcap + 1) & ~1));
    uint16_t *queues = reinterpret_cast<uint16_t *>(bitmaps + DM_WAVES * bw);
    int *ovf = reinterpret_cast<int *>(queues + DM_WAVES * DM_QCAP);
    {
        const uint16_t *gn = d.nbr + (size_t)b * Scap, *gr = d.rgrid + (size_t)b * d.Rcap;
        for (int k = tid; k < S; k += blockDim.x) nbr[k] = gn[k];
        for (int k = tid; k < R; k += blockDim.x) rg[k] = gr[k];
        if (tid == 0) *ovf = 0;
    }
    __syncthreads();
    const int u = u0 + wave;
    if (u < U) {  // no workgroup barrier inside this block
        uint32_t *bm = bitmaps + wave * bw;
        uint16_t *q = queues + wave * DM_QCAP;
        uint16_t *out = d.dm + ((size_t)b * d.Ucap + u) * Scap;
        for (int k = lane; k < bw; k += 64) bm[k] = 0;
        wave_sync();
        const int tr = d.ut_r[(size_t)b * d.Ucap + u];
        int head = 0, tail = 0;  // wave-uniform ring positions (monotonic; slot = position & (DM_QCAP - 1))
        bool overflow = false;
        // a lane that found an unvisited state: distance to HBM, state to the ring (wave-aggregated append)
        auto visit = [&](bool want, uint32_t s, uint32_t dist) {
            bool fresh = false;
            if (want) {
                const uint32_t bit = 1u << (s & 31u);
                fresh = !(atomicOr(&bm[s >> 5], bit) & bit);
            }
            const unsigned long long m = __ballot(fresh);
            if (m == 0) return;
            if (fresh) {
                out[s] = (uint16_t)dist;
                q[(tail + __popcll(m & ((1ull << lane) - 1ull))) & (DM_QCAP - 1)] = (uint16_t)s;
            }
            tail += __popcll(m);
        };
        // distance_map.py:88-99: the target cell is 0 for all four orientations and pre-visited; the walk is seeded with
        // _get_and_update_neighbors(target, 0, enforce_target_direction=-1)
        if (lane < 4) {
            const uint32_t s = (uint32_t)tr * 4u + lane;
            atomicOr(&bm[s >> 5], 1u << (s & 31u));
            out[s] = 0;
        }
        wave_sync();
        {
            const uint32_t nd = lane & 3u, a = (lane >> 2) & 3u;  // lanes 0..15: (direction to the neighbour, its orientation)
            const uint32_t nr = lane < 16 ? nbr[tr * 4 + nd] : FL_R_NONE;
            const bool want = nr != FL_R_NONE && tbit(rg[nr], a, (nd + 2u) & 3u);
            visit(want, nr * 4u + a, 1u);
        }
        wave_sync();
        uint32_t dist = 1;
        while (head < tail && !overflow) {
            const int lvl_end = tail;
            for (int k0 = head; k0 < lvl_end; k0 += 64) {
                const int k = k0 + lane;
                const bool on = k < lvl_end;
                uint32_t o = 0, nr = FL_R_NONE, g = 0;
                if (on) {
                    const uint32_t s = q[k & (DM_QCAP - 1)];
                    o = s & 3u;
                    // the agent landed here with orientation o, so it came from the cell in direction (o + 2) % 4 (:133-136)
                    nr = nbr[(s & ~3u) | ((o + 2u) & 3u)];
                    if (nr != FL_R_NONE) g = rg[nr];
                }
#pragma unroll
                for (uint32_t a = 0; a < 4; a++) visit(nr != FL_R_NONE && tbit(g, a, o), nr * 4u + a, dist + 1);
                if (tail - k0 > DM_QCAP) overflow = true;  // the ring wrapped onto entries of this level not yet read
                wave_sync();
            }
            head = lvl_end;
            dist++;
            if (dist >= 0xFFFEu) overflow = true;
        }
        for (int s = lane; s < S; s += 64)
            if (!((bm[s >> 5] >> (s & 31)) & 1u)) out[s] = FL_INF16;
        if (overflow && lane == 0) atomicCAS(&d.err[b], 0, FL_ERR_CAPACITY);
    }
    __syncthreads();  // the next piece of work reuses the LDS
    }
}

// ---------------------------------------------------------------------------------------------- segment table
// The branch walk of the tree observations (treeobs.cpp:322-539 / observations.py:296-425) is a deterministic
// chain over rail states until a switch, a dead end, a zero-transition cell or a revisited state; only the stop at the
// agent's own target depends on the agent.  The chain's end, its length, and the first "unusable switch" offset are
// therefore static per start state and are tabulated once per env, together with the successor of every state that
// has exactly one transition (what the walks of the observation kernels follow cell by cell).
__device__ __forceinline__ int seg_next(const uint16_t *rg, const uint16_t *nbr, int state) {
    const uint32_t g = rg[state >> 2];
    const uint32_t bits = nibble(g, state & 3);
    int total = __popc(g);
    if (g == 0x8421u) total = 2;
    if (__popc(bits) != 1 || total == 1) return -1;
    const uint32_t nd = first_dir(bits);
    const uint32_t nr = nbr[(state & ~3) | (int)nd];
    if (nr == FL_R_NONE) return -1;  // malformed map (the transition leaves the rail): treat as the end of the chain
    return (int)((nr << 2) | nd);
}

__global__ __launch_bounds__(256) void k_segments(FlDev d, const int *__restrict__ env_list, const int *__restrict__ env_count) {
    const int Scap = d.Rcap * 4;
    const int per_env = (Scap + 255) / 256;
    const int n_work = *env_count * per_env;
    for (int work = blockIdx.x; work < n_work; work += gridDim.x) {
    const int b = env_list[work / per_env];
    const int s0 = (work % per_env) * 256 + threadIdx.x;
    const int S = d.R[b] * 4;
    if (s0 >= Scap) continue;
    uint4 out = make_uint4((uint32_t)s0 | (SEG_ZERO << 20), 0xFFFF0000u, 0xFFFFFFFFu, 0xFFFFFFFFu);
    uint16_t sn = FL_R_NONE;
    if (s0 < S) {
        const uint16_t *rg = d.rgrid + (size_t)b * d.Rcap, *nbr = d.nbr + (size_t)b * Scap;
        {   // successor table: any state with exactly one transition whose exit stays on the rail (dead ends included)
            const uint32_t bits = nibble(rg[s0 >> 2], s0 & 3);
            if (__popc(bits) == 1) {
                const uint32_t nd = first_dir(bits);
                const uint32_t nr = nbr[(s0 & ~3) | (int)nd];
                if (nr != FL_R_NONE) sn = (uint16_t)((nr << 2) | nd);
            }
        }
        const int cap = S;  // a longer walk has revisited a state
        int cur = s0, k = 0, unus = 0xFFFF;
        uint32_t kind = SEG_ZERO, phantom = 0;
        while (true) {
            const uint32_t g = rg[cur >> 2];
            const uint32_t bits = nibble(g, cur & 3);
            int total = __popc(g);
            if (g == 0x8421u) total = 2;
            const int num = __popc(bits);
            if (total > 2 && 2 > num && unus == 0xFFFF) unus = k;
            if (num == 1) {
                if (total == 1) { kind = SEG_DEAD_END; break; }
                const int nx = seg_next(rg, nbr, cur);
                if (nx < 0) { kind = SEG_ZERO; phantom = 1; break; }  // (one transition, not a dead end: it leaves the rail)
                cur = nx;
                k++;
                if (k > cap) {  // cycle: find the first revisited state (Brent), the walk is terminal there
                    int power = 1, lam = 1, tort = s0, hare = seg_next(rg, nbr, s0);
                    while (tort != hare) {
                        if (power == lam) { tort = hare; power *= 2; lam = 0; }
                        hare = seg_next(rg, nbr, hare);
                        lam++;
                    }
                    tort = hare = s0;
                    for (int i = 0; i < lam; i++) hare = seg_next(rg, nbr, hare);
                    int mu = 0;
                    while (tort != hare) { tort = seg_next(rg, nbr, tort); hare = seg_next(rg, nbr, hare); mu++; }
                    k = mu + lam;
                    cur = hare;  // state at index mu == state at index mu + lam
                    kind = SEG_CYCLE;
                    break;
                }
            } else if (num > 1) { kind = SEG_SWITCH; break; }
            else { kind = SEG_ZERO; break; }
        }
        // children of the end state (switch: the transitions left / forward / right / back of the arrival direction; dead end:
        // the reversed directions, treeobs.cpp:583-608)
        uint32_t kid[4] = {FL_R_NONE, FL_R_NONE, FL_R_NONE, FL_R_NONE};
        if (kind == SEG_SWITCH || kind == SEG_DEAD_END) {
            const uint32_t ecell = (uint32_t)cur >> 2, edir = (uint32_t)cur & 3u;
            const uint32_t pbits = nibble(rg[ecell], edir);
            for (uint32_t c = 0; c < 4; c++) {
                const uint32_t bd = (edir + c + 3u) & 3u;
                const uint32_t use = kind == SEG_DEAD_END ? (bd + 2u) & 3u : bd;
                if ((pbits >> (3u - use)) & 1u) {
                    const uint32_t nr = nbr[ecell * 4u + use];
                    kid[c] = nr != FL_R_NONE ? ((nr << 2) | use) : FL_R_PHANTOM;
                }
            }
        }
        out = make_uint4((uint32_t)cur | (kind << 20) | (phantom << 22), (uint32_t)k | ((uint32_t)unus << 16), kid[0] | (kid[1] << 16), kid[2] | (kid[3] << 16));
    }
    d.seg[(size_t)b * Scap + s0] = out;
    d.snext[(size_t)b * Scap + s0] = sn;
    }
}

// ---------------------------------------------------------------------------------------------- next-hop table
// The shortest-path predictors (predictions.cpp:78-144, rail_env_shortest_paths.py:203-274) descend the distance map
// greedily: among the valid move actions in the order left, forward, right (a dead end only offers its reverse exit)
// they take the first one with the smallest distance, if it is finite.  On a BFS map that choice is static per
// (target, rail cell, orientation); it is tabulated here, 3 bits per orientation (4 = nothing closer), 12 bits per cell.
__global__ __launch_bounds__(256) void k_nexthop(FlDev d, const int *__restrict__ env_list, const int *__restrict__ env_count) {
    const int per_bu = (d.Rcap + 255) / 256, per_env = d.Ucap * per_bu;
    const int n_work = *env_count * per_env;
    for (int work = blockIdx.x; work < n_work; work += gridDim.x) {
    const int b = env_list[work / per_env], u = (work % per_env) / per_bu;
    const int bu = b * d.Ucap + u;
    const int r = (work % per_bu) * 256 + threadIdx.x;
    if (r >= d.Rcap) continue;
    const int Scap = d.Rcap * 4;
    uint32_t out = 0x924;  // 4 | 4<<3 | 4<<6 | 4<<9
    if (u < d.U[b] && r < d.R[b]) {
        const uint32_t g = d.rgrid[(size_t)b * d.Rcap + r];
        const uint16_t *nbr = d.nbr + (size_t)b * Scap + r * 4;
        const uint16_t *dm = d.dm + (size_t)bu * Scap;
        out = 0;
        for (uint32_t dd = 0; dd < 4; dd++) {
            const uint32_t bits = nibble(g, dd);
            uint32_t best = 4, bestv = FL_INF16;
            if (__popc(g) == 1) {  // is_dead_end: only the reverse exit
                const uint32_t ex = (dd + 2u) & 3u;
                if (((bits >> (3 - ex)) & 1) && nbr[ex] != FL_R_NONE) {
                    const uint32_t v = dm[(size_t)nbr[ex] * 4 + ex];
                    if (v != FL_INF16) { best = ex; bestv = v; }
                }
            } else {
                for (int j = 0; j < 3; j++) {
                    const uint32_t nd = (dd + (uint32_t)(j + 3)) & 3u;
                    if (((bits >> (3 - nd)) & 1) && nbr[nd] != FL_R_NONE) {
                        const uint32_t v = dm[(size_t)nbr[nd] * 4 + nd];
                        if (v != FL_INF16 && v < bestv) { best = nd; bestv = v; }
                    }
                }
            }
            // strict descent: the hop must be closer than the state itself (always true on a consistent map)
            const uint32_t here = dm[(size_t)r * 4 + dd];
            if (best != 4 && !(bestv < here)) best = 4;
            out |= best << (3 * dd);
        }
    }
    d.nh[(size_t)bu * d.Rcap + r] = (uint16_t)out;
    }
}

// Eight greedy hops at once: hop8[b][u][s] = the state reached from s after eight next-hops towards target u, or
// FL_R_NONE when the greedy path ends earlier.  The observation kernels walk a predicted path with eight lanes, lane j
// covering the waypoints j, j + 8, j + 16, ... (a chain of gathers an eighth as long as the single-step chain).
__global__ __launch_bounds__(256) void k_hop8(FlDev d, const int *__restrict__ env_list, const int *__restrict__ env_count) {
    const int Scap = d.Rcap * 4;
    const int per_bu = (Scap + 255) / 256, per_env = d.Ucap * per_bu;
    const int n_work = *env_count * per_env;
    for (int work = blockIdx.x; work < n_work; work += gridDim.x) {
    const int b = env_list[work / per_env], u = (work % per_env) / per_bu;
    const int bu = b * d.Ucap + u;
    const int s0 = (work % per_bu) * 256 + threadIdx.x;
    if (s0 >= Scap) continue;
    const uint16_t *nh = d.nh + (size_t)bu * d.Rcap;
    const uint16_t *nbr = d.nbr + (size_t)b * Scap;
    uint32_t st = (uint32_t)s0;
    bool ok = u < d.U[b] && s0 < d.R[b] * 4;
    for (int k = 0; k < 8 && ok; k++) {
        const uint32_t hop = ((uint32_t)nh[st >> 2] >> (3u * (st & 3u))) & 7u;
        if (hop == 4u) ok = false;
        else {
            const uint32_t nr = nbr[(st & ~3u) | hop];
            if (nr == FL_R_NONE) ok = false;
            else st = (nr << 2) | hop;
        }
    }
    d.hop8[(size_t)bu * Scap + s0] = ok ? (uint16_t)st : (uint16_t)FL_R_NONE;
    }
}

static size_t dm_lds_bytes(const FlDev &d) {
    const size_t Scap = (size_t)d.Rcap * 4;
    return Scap * 2 + (size_t)((d.Rcap + 1) & ~1) * 2 + (size_t)DM_WAVES * ((Scap + 31) / 32) * 4 + (size_t)DM_WAVES * DM_QCAP * 2 + 16;
}

int fl_dmap_fits(const FlDev &d) { return dm_lds_bytes(d) > 160 * 1024 ? FL_ERR_ARG : FL_OK; }  // no HIP call: usable before a device is touched

int fl_dmap_prepare(const FlDev &d) {
    if (dm_lds_bytes(d) > 160 * 1024) return FL_ERR_ARG;
    if (hipFuncSetAttribute((const void *)k_distance_map, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    return FL_OK;
}

// grid of a table kernel: every piece of work of a full build gets its own workgroup; a masked rebuild (count unknown on
// the host, usually tiny) gets a fixed modest grid whose workgroups loop
static unsigned table_grid(const FlDev &d, bool masked, size_t per_env) {
    const size_t full = (size_t)d.B * per_env;
    const size_t g = masked ? (full < 2048 ? full : 2048) : full;
    return (unsigned)(g > 0x7fffffffull ? 0x7fffffffull : g);
}

void fl_launch_env_list(const FlDev &d, const uint8_t *mask_dev, hipStream_t s, bool through_tab) {
    hipLaunchKernelGGL(k_env_list, dim3(1), dim3(1024), 0, s, d.B, mask_dev, through_tab ? d.tab : (const int *)nullptr, d.env_list, d.env_list + d.B);
}

void fl_launch_distance_maps(const FlDev &d, const uint8_t *mask_dev, hipStream_t s) {
    const size_t per = (d.Ucap + DM_WAVES - 1) / DM_WAVES;
    hipLaunchKernelGGL(k_distance_map, dim3(table_grid(d, mask_dev != nullptr, per)), dim3(64 * DM_WAVES), dm_lds_bytes(d), s, d, d.env_list, d.env_list + d.B);
}

void fl_launch_segments(const FlDev &d, const uint8_t *mask_dev, hipStream_t s) {
    const size_t per = ((size_t)d.Rcap * 4 + 255) / 256;
    hipLaunchKernelGGL(k_segments, dim3(table_grid(d, mask_dev != nullptr, per)), dim3(256), 0, s, d, d.env_list, d.env_list + d.B);
}

void fl_launch_nexthop(const FlDev &d, const uint8_t *mask_dev, hipStream_t s) {
    const size_t per = (size_t)d.Ucap * ((d.Rcap + 255) / 256);
    hipLaunchKernelGGL(k_nexthop, dim3(table_grid(d, mask_dev != nullptr, per)), dim3(256), 0, s, d, d.env_list, d.env_list + d.B);
}

void fl_launch_hop8(const FlDev &d, const uint8_t *mask_dev, hipStream_t s) {
    const size_t per = (size_t)d.Ucap * (((size_t)d.Rcap * 4 + 255) / 256);
    hipLaunchKernelGGL(k_hop8, dim3(table_grid(d, mask_dev != nullptr, per)), dim3(256), 0, s, d, d.env_list, d.env_list + d.B);
}
