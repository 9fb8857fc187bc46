// fl_dmap.hip -- distance-map build: reverse BFS from every unique target over (cell, orientation) states.
// Replaces DistanceMap._compute/_distance_map_walker/_get_and_update_neighbors
// (flatland-rl/flatland/envs/distance_map.py:57-160).
//
// One 256-thread workgroup per (env, unique target).  The visited set is a bitmap in LDS (H*W*4 bits),
// the two BFS frontiers live in LDS, the distances are written once (by the discovering lane) to the
// env's u16 slab in HBM.  d[r,c,o] = 1 + min_m { d[(r,c)+delta_m, m] : bit(o->m) } with d[target,*] = 0.
#include "fl_internal.h"
#include "../../include/flatland_hip.h"

#define DM_FRONTIER_CAP 4096

__device__ __forceinline__ void dm_visit(uint32_t s, uint32_t dist, uint32_t *bitmap, uint16_t *out, uint32_t *fr_next,
                                         uint32_t *cnt_next, int *overflow) {
    uint32_t bit = 1u << (s & 31u);
    uint32_t old = atomicOr(&bitmap[s >> 5], bit);
    if (!(old & bit)) {
        out[s] = (uint16_t)dist;
        uint32_t idx = atomicAdd(cnt_next, 1u);
        if (idx < DM_FRONTIER_CAP) fr_next[idx] = s;
        else *overflow = 1;
    }
}

__global__ __launch_bounds__(256) void k_distance_map(FlDev d) {
    const int b = blockIdx.x / d.Umax, u = blockIdx.x % d.Umax;
    if (u >= d.U[b]) return;
    const int H = d.H, W = d.W, HW = H * W;
    const int tid = threadIdx.x, nt = blockDim.x;
    uint16_t *out = d.dm + ((size_t)(b * d.Umax + u) * HW) * 4;
    const uint16_t *grid = d.grid + (size_t)b * HW;
    extern __shared__ uint32_t lds[];
    const int nwords = (HW * 4 + 31) / 32;
    uint32_t *bitmap = lds;
    uint32_t *fr0 = bitmap + nwords;
    uint32_t *fr1 = fr0 + DM_FRONTIER_CAP;
    uint32_t *cnt = fr1 + DM_FRONTIER_CAP;  // [2]
    int *overflow = (int *)(cnt + 2);

    // all states unreachable; 8 bytes (4 orientations) per cell
    uint2 inf2 = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    for (int c = tid; c < HW; c += nt) reinterpret_cast<uint2 *>(out)[c] = inf2;
    for (int i = tid; i < nwords; i += nt) bitmap[i] = 0;
    if (tid < 2) cnt[tid] = 0;
    if (tid == 0) *overflow = 0;
    __syncthreads();

    const int target = d.ut[b * d.Umax + u];
    const int tr = target / W, tc = target % W;
    // distance_map.py:88-99: target cell = 0 for all four orientations, and those states are pre-visited
    if (tid < 4) {
        uint32_t s = (uint32_t)target * 4u + tid;
        atomicOr(&bitmap[s >> 5], 1u << (s & 31u));
        out[s] = 0;
    }
    __syncthreads();
    // seed: _get_and_update_neighbors(position, 0, enforce_target_direction=-1) (:92)
    if (tid < 4) {
        const int nd = tid;
        const int nr = tr + (nd == 0 ? -1 : nd == 2 ? 1 : 0), nc = tc + (nd == 1 ? 1 : nd == 3 ? -1 : 0);
        if (nr >= 0 && nr < H && nc >= 0 && nc < W) {
            const uint32_t cell = grid[nr * W + nc];
            const uint32_t desired = (nd + 2) & 3;
            for (uint32_t a = 0; a < 4; a++)
                if (tbit(cell, a, desired)) dm_visit((uint32_t)(nr * W + nc) * 4u + a, 1u, bitmap, out, fr0, &cnt[0], overflow);
        }
    }
    __syncthreads();

    uint32_t dist = 1;
    int cur = 0;
    while (true) {
        const uint32_t n = min(cnt[cur], (uint32_t)DM_FRONTIER_CAP);
        if (n == 0) break;
        uint32_t *fc = cur ? fr1 : fr0, *fn = cur ? fr0 : fr1;
        for (uint32_t k = tid; k < n; k += nt) {
            const uint32_t s = fc[k];
            const int cell = (int)(s >> 2);
            const uint32_t o = s & 3u;
            const int r = cell / W, c = cell % W;
            // the agent landed here with orientation o, so it came from the cell in direction (o+2)%4 (:133-136)
            const uint32_t back = (o + 2) & 3;
            const int nr = r + (back == 0 ? -1 : back == 2 ? 1 : 0), nc = c + (back == 1 ? 1 : back == 3 ? -1 : 0);
            if (nr >= 0 && nr < H && nc >= 0 && nc < W) {
                const uint32_t g = grid[nr * W + nc];
                if (g) {
                    for (uint32_t a = 0; a < 4; a++)
                        if (tbit(g, a, o)) dm_visit((uint32_t)(nr * W + nc) * 4u + a, dist + 1, bitmap, out, fn, &cnt[cur ^ 1], overflow);
                }
            }
        }
        __syncthreads();
        if (tid == 0) cnt[cur] = 0;
        cur ^= 1;
        dist++;
        __syncthreads();
        if (dist >= 0xFFFEu) {
            if (tid == 0) *overflow = 1;
            break;
        }
    }
    __syncthreads();
    if (tid == 0 && *overflow) atomicCAS(&d.err[b], 0, FL_ERR_CAPACITY);
}

// ---------------------------------------------------------------------------------------------- segment table
// The branch walk of the tree observations (treeobs.cpp:322-539 / observations.py:296-425) is a deterministic
// chain over (cell, orientation) states until a switch, a dead end, a zero-transition cell or a revisited state;
// only the stop at the agent's own target depends on the agent.  The chain's end, its length, and the first
// "unusable switch" offset are therefore static per start state and are tabulated once per env.
__device__ __forceinline__ int seg_next(const uint16_t *grid, int H, int W, int state) {
    const int cell = state >> 2;
    const uint32_t d = state & 3;
    const uint32_t g = grid[cell];
    const uint32_t bits = nibble(g, d);
    int total = __popc(g);
    if (g == 0x8421u) total = 2;
    if (__popc(bits) != 1 || total == 1) return -1;
    const uint32_t nd = first_dir(bits);
    const int r = cell / W, c = cell - r * W;
    const int nr = r + (nd == 0 ? -1 : nd == 2 ? 1 : 0), nc = c + (nd == 1 ? 1 : nd == 3 ? -1 : 0);
    if (nr < 0 || nc < 0 || nr >= H || nc >= W) return -1;  // malformed map: treat as the end of the chain
    return ((nr * W + nc) << 2) | (int)nd;
}

__global__ __launch_bounds__(256) void k_segments(FlDev d) {
    const int HW = d.H * d.W, NS = HW * 4;
    const int per_env = (NS + 255) / 256;
    const int b = blockIdx.x / per_env;
    const int s0 = (blockIdx.x % per_env) * 256 + threadIdx.x;
    if (s0 >= NS) return;
    const uint16_t *grid = d.grid + (size_t)b * HW;
    uint2 out = make_uint2((uint32_t)s0 | (SEG_ZERO << 20), 0xFFFF0000u);
    if (grid[s0 >> 2] != 0) {
        const int cap = 65534;  // rail states < 65534 is enforced at fl_load_env, so a longer walk has revisited a state
        int cur = s0, k = 0, unus = 0xFFFF;
        uint32_t kind = SEG_ZERO;
        while (true) {
            const uint32_t g = grid[cur >> 2];
            const uint32_t bits = nibble(g, cur & 3);
            int total = __popc(g);
            if (g == 0x8421u) total = 2;
            const int num = __popc(bits);
            if (total > 2 && 2 > num && unus == 0xFFFF) unus = k;
            if (num == 1) {
                if (total == 1) { kind = SEG_DEAD_END; break; }
                const int nx = seg_next(grid, d.H, d.W, cur);
                if (nx < 0) { kind = SEG_ZERO; break; }
                cur = nx;
                k++;
                if (k >= cap) {  // cycle: find the first revisited state (Brent), the walk is terminal there
                    int power = 1, lam = 1, tort = s0, hare = seg_next(grid, d.H, d.W, s0);
                    while (tort != hare) {
                        if (power == lam) { tort = hare; power *= 2; lam = 0; }
                        hare = seg_next(grid, d.H, d.W, hare);
                        lam++;
                    }
                    tort = hare = s0;
                    for (int i = 0; i < lam; i++) hare = seg_next(grid, d.H, d.W, hare);
                    int mu = 0;
                    while (tort != hare) { tort = seg_next(grid, d.H, d.W, tort); hare = seg_next(grid, d.H, d.W, hare); mu++; }
                    k = mu + lam;
                    cur = hare;  // state at index mu == state at index mu + lam
                    kind = SEG_CYCLE;
                    break;
                }
            } else if (num > 1) { kind = SEG_SWITCH; break; }
            else { kind = SEG_ZERO; break; }
        }
        out = make_uint2((uint32_t)cur | (kind << 20), (uint32_t)k | ((uint32_t)unus << 16));
    }
    d.seg[(size_t)b * NS + s0] = out;
}

// ---------------------------------------------------------------------------------------------- next-hop table
// The shortest-path predictors (predictions.cpp:78-144, rail_env_shortest_paths.py:203-274) descend the distance map
// greedily: among the valid move actions in the order left, forward, right (a dead end only offers its reverse exit)
// they take the first one with the smallest distance, if it is finite.  On a BFS map that choice is static per
// (target, cell, orientation); it is tabulated here, 3 bits per orientation (4 = nothing closer), 12 bits per cell.
__global__ __launch_bounds__(256) void k_nexthop(FlDev d) {
    const int HW = d.H * d.W, W = d.W;
    const long long n = (long long)d.B * d.Umax * HW;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int cell = (int)(idx % HW);
    const int bu = (int)(idx / HW);
    const int b = bu / d.Umax, u = bu % d.Umax;
    uint32_t out = 0x924;  // 4 | 4<<3 | 4<<6 | 4<<9
    const uint32_t g = d.grid[(size_t)b * HW + cell];
    if (u < d.U[b] && g != 0) {
        const uint16_t *dm = d.dm + ((size_t)bu * HW) * 4;
        out = 0;
        for (uint32_t dd = 0; dd < 4; dd++) {
            const uint32_t bits = nibble(g, dd);
            uint32_t best = 4, bestv = FL_INF16;
            if (__popc(g) == 1) {  // is_dead_end: only the reverse exit
                const uint32_t ex = (dd + 2u) & 3u;
                if ((bits >> (3 - ex)) & 1) {
                    const uint32_t v = dm[(size_t)step_cell(cell, ex, W) * 4 + ex];
                    if (v != FL_INF16) { best = ex; bestv = v; }
                }
            } else {
                for (int j = 0; j < 3; j++) {
                    const uint32_t nd = (dd + (uint32_t)(j + 3)) & 3u;
                    if ((bits >> (3 - nd)) & 1) {
                        const uint32_t v = dm[(size_t)step_cell(cell, nd, W) * 4 + nd];
                        if (v != FL_INF16 && v < bestv) { best = nd; bestv = v; }
                    }
                }
            }
            // strict descent: the hop must be closer than the state itself (always true on a consistent map)
            const uint32_t here = dm[(size_t)cell * 4 + dd];
            if (best != 4 && !(bestv < here)) best = 4;
            out |= best << (3 * dd);
        }
    }
    d.nh[idx] = (uint16_t)out;
}

// Eight greedy hops at once: hop8[b][u][state] = the state reached from `state` after eight next-hops towards target u, or
// FL_HOP_NONE when the greedy path ends earlier.  The observation kernels walk a predicted path with eight lanes, lane j
// covering the waypoints j, j + 8, j + 16, ... (a chain of L2 gathers an eighth as long as the single-step chain).
__global__ __launch_bounds__(256) void k_hop8(FlDev d) {
    const int HW = d.H * d.W, W = d.W;
    const long long n = (long long)d.B * d.Umax * HW * 4;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int st0 = (int)(idx % (HW * 4));
    const long long bu = idx / (HW * 4);
    const uint16_t *nh = d.nh + (size_t)bu * HW;
    int cell = st0 >> 2;
    uint32_t dd = st0 & 3;
    bool ok = (int)(bu % d.Umax) < d.U[bu / d.Umax];
    for (int k = 0; k < 8 && ok; k++) {
        const uint32_t hop = ((uint32_t)nh[cell] >> (3u * dd)) & 7u;
        if (hop == 4u) ok = false;
        else { cell = step_cell(cell, hop, W); dd = hop; }
    }
    d.hop8[idx] = ok ? (((uint32_t)cell << 2) | dd) : FL_HOP_NONE;
}

// hop8 restricted to rail states and expressed in rail-state space (see FlDev::chop8)
__global__ __launch_bounds__(256) void k_chop8(FlDev d) {
    const int HW = d.H * d.W;
    const long long per = (long long)d.Rmax * 4;
    const long long n = (long long)d.B * d.Umax * per;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const long long bu = idx / per;
    const int b = (int)(bu / d.Umax), rs = (int)(idx % per);
    uint16_t out = 0xFFFF;
    if ((rs >> 2) < d.R[b]) {
        const uint32_t cell = d.rcell[(size_t)b * d.Rmax + (rs >> 2)];
        const uint32_t s8 = d.hop8[bu * ((long long)HW * 4) + ((long long)cell << 2 | (rs & 3))];
        if (s8 != FL_HOP_NONE) out = (uint16_t)(((uint32_t)d.ridx[(size_t)b * HW + (s8 >> 2)] << 2) | (s8 & 3u));
    }
    d.chop8[idx] = out;
}

void fl_launch_hop8(const FlDev &d, hipStream_t s) {
    const long long n = (long long)d.B * d.Umax * d.H * d.W * 4;
    hipLaunchKernelGGL(k_hop8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d);
    if (d.chop8) {
        const long long nc = (long long)d.B * d.Umax * d.Rmax * 4;
        hipLaunchKernelGGL(k_chop8, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, d);
    }
}

void fl_launch_nexthop(const FlDev &d, hipStream_t s) {
    const long long n = (long long)d.B * d.Umax * d.H * d.W;
    hipLaunchKernelGGL(k_nexthop, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d);
}

void fl_launch_segments(const FlDev &d, hipStream_t s) {
    const int NS = d.H * d.W * 4;
    hipLaunchKernelGGL(k_segments, dim3(d.B * ((NS + 255) / 256)), dim3(256), 0, s, d);
}

void fl_launch_distance_maps(const FlDev &d, hipStream_t s) {
    const int HW = d.H * d.W;
    size_t lds = ((size_t)(HW * 4 + 31) / 32 + 2 * DM_FRONTIER_CAP + 4) * sizeof(uint32_t);
    hipLaunchKernelGGL(k_distance_map, dim3(d.B * d.Umax), dim3(256), lds, s, d);
}
