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

void fl_launch_distance_maps(const FlDev &d, hipStream_t s) {
    const int HW = d.H * d.W;
    size_t lds = ((size_t)(HW * 4 + 31) / 32 + 2 * DM_FRONTIER_CAP + 4) * sizeof(uint32_t);
    hipLaunchKernelGGL(k_distance_map, dim3(d.B * d.Umax), dim3(256), lds, s, d);
}
