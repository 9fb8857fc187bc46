// fl_step.hip -- RailEnv.step() for B envs in lock-step, one workgroup per env, one lane per agent.
// Replaces flatland-rl/flatland/envs/rail_env.py:501-634 and everything it calls:
//   malfunction RNG       envs/malfunction_generators.py:46-53, step_utils/malfunction_handler.py:35-50
//                         (numpy legacy RandomState / MT19937, one stream per env consumed in handle order)
//   action preprocessing  rail_env.py:425-446, step_utils/action_preprocessing.py:7-59, transition_utils.py:6-82
//   tentative move        step_utils/env_utils.py:26-43
//   MotionCheck           envs/agent_chains.py:19-236 (restated on cells; see resolve_conflicts below)
//   state machine         rail_env.py:369-395, step_utils/state_machine.py:12-144
//   commit / done / counters rail_env.py:596-627, :493-499, speed_counter.py:10-14
//   end-of-episode reward rail_env.py:397-423,476-491, agent_utils.py:129-147, rail_env_shortest_paths.py:203-274
//
// Design (gfx950): the env's MT19937 block lives in LDS for the step (twist = 3 data-parallel passes);
// every agent evaluates its malfunction draw speculatively at stream offset 2*i, and only the (rare)
// firing agents are replayed serially because their rejection-sampled duration shifts the offsets of
// all later agents.  Cell conflicts are resolved through an LDS hash table keyed by cell id
// (occupant / lowest-handle winner / blocked flag) with a barrier-terminated backward propagation.
#include "fl_internal.h"
#include "../../include/flatland_hip.h"

#include "fl_step_body.h"

template <bool SYNTH>
__global__ __launch_bounds__(1024) void k_step(FlDev d, const uint8_t *__restrict__ actions, uint32_t seed,
                                               uint32_t stream_base, int synth_kind, int32_t *__restrict__ rewards,
                                               uint8_t *__restrict__ dones, uint8_t *__restrict__ done_all_out,
                                               int auto_reset, int wcap, int S, int sshift) {
    extern __shared__ uint32_t lds_raw[];
    step_body<SYNTH>(d, actions, seed, stream_base, synth_kind, rewards, dones, done_all_out, auto_reset, wcap, S, sshift, lds_raw);
}

__global__ void k_reset(FlDev d, const uint8_t *mask, int fresh) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.B * d.A) return;
    const int b = g / d.A;
    if (mask && !mask[b]) return;
    const uint32_t init_dir = SPK_INIT_DIR(d.spk[g]);
    d.pos[g] = -1;
    d.old_pos[g] = -1;
    if (fresh) d.arrival[g] = -1;
    d.malf[g] = 0;
    d.pk[g] = pk_make(init_dir, 4, ST_WAITING, 7, 0, 0, 0, 0, 0);
    if (g == b * d.A) { d.t[b] = 0; d.done_all[b] = 0; }
}

__global__ void k_metrics(FlDev d, long long *out4, int reset) {
    // Σ over envs of (terminal rewards, arrived agents, agent-steps, episodes); one workgroup, wave reduction
    long long acc[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < d.B; b += blockDim.x)
        for (int k = 0; k < 4; k++) {
            acc[k] += d.metrics[(size_t)b * 4 + k];
            if (reset) d.metrics[(size_t)b * 4 + k] = 0;
        }
    for (int k = 0; k < 4; k++) {
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off);
        if ((threadIdx.x & 63) == 0 && acc[k] != 0) atomicAdd((unsigned long long *)&out4[k], (unsigned long long)acc[k]);
    }
}

void fl_launch_metrics(const FlDev &d, long long *out4, int reset, hipStream_t s) {
    (void)hipMemsetAsync(out4, 0, 4 * sizeof(long long), s);
    hipLaunchKernelGGL(k_metrics, dim3(1), dim3(256), 0, s, d, out4, reset);
}

// sums of the evaluator's per-episode scores over the envs, in env order (one thread: a fixed order of additions)
__global__ void k_scores(FlDev d, double *out3, int reset) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // (the episode count is the scores' OWN counter, incremented where the sums are and reset with them: a caller that resets
    // fl_metrics and fl_scores at different times still gets sums and a count of the same window)
    double s0 = 0.0, s1 = 0.0, ep = 0.0;
    for (int b = 0; b < d.B; b++) {
        s0 += d.score_sums[(size_t)b * 3 + 0];
        s1 += d.score_sums[(size_t)b * 3 + 1];
        ep += d.score_sums[(size_t)b * 3 + 2];
        if (reset) { d.score_sums[(size_t)b * 3 + 0] = 0.0; d.score_sums[(size_t)b * 3 + 1] = 0.0; d.score_sums[(size_t)b * 3 + 2] = 0.0; }
    }
    out3[0] = s0; out3[1] = s1; out3[2] = ep;
}

void fl_launch_scores(const FlDev &d, double *out3, int reset, hipStream_t s) {
    hipLaunchKernelGGL(k_scores, dim3(1), dim3(64), 0, s, d, out3, reset);
}

// plfActor.get_feature casts + Network.modify_adjacency (solution/plfActor.py:48-74, nn/net_tree.py:105-116)
__global__ void k_policy_pack(int B, int A, int E, const int32_t *adj, const int32_t *no, const int32_t *eo,
                              long long *adj_out, long long *no_out, long long *eo_out) {
    const long long n_adj = (long long)B * A * E * 3, n_no = (long long)B * A * (E + 1), n_eo = (long long)B * A * E;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n_adj; k += stride) {
        const long long tree = k / (3ll * E);  // b * A + a
        const int col = (int)(k % 3);
        long long v = adj[k];
        // -2 -> -(B*A*N) so that the offset cannot make it non-negative; offsets on parent / child; negatives -> -2
        if (v == -2) v = -(long long)B * A * (E + 1);
        if (col < 2) v += tree * (E + 1);
        adj_out[k] = v < 0 ? -2 : v;
    }
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n_no; k += stride) no_out[k] = no[k];
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n_eo; k += stride) eo_out[k] = eo[k];
}

void fl_launch_policy_pack(int B, int A, int E, const int32_t *adj, const int32_t *no, const int32_t *eo, long long *adj_out,
                           long long *no_out, long long *eo_out, hipStream_t s) {
    const long long n = (long long)B * A * E * 3;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_policy_pack, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, B, A, E, adj, no, eo, adj_out, no_out, eo_out);
}

// RailEnv.get_info_dict (rail_env.py:452-468) as tensors + the evaluator's per-episode scores
__global__ void k_info(FlDev d, uint8_t *action_required, int32_t *malfunction, uint8_t *state, double *scores) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < d.B * d.A) {
        const uint32_t pk = d.pk[g];
        const uint32_t st = PK_STATE(pk);
        if (action_required) action_required[g] = (st == ST_READY || (is_on_map(st) && PK_SCOUNT(pk) == 0)) ? 1 : 0;
        if (malfunction) malfunction[g] = (int32_t)(d.malf[g] & 0xFFFFu);
        if (state) state[g] = (uint8_t)st;
    }
    if (scores && g < d.B) {
        // normalized reward = sum(rewards) / (max_episode_steps * n_agents) + 1, completion = arrived / n_agents
        scores[(size_t)g * 2 + 0] = 1.0 + (double)d.last_episode[(size_t)g * 2 + 0] / ((double)d.T[g] * (double)d.A);
        scores[(size_t)g * 2 + 1] = (double)d.last_episode[(size_t)g * 2 + 1] / (double)d.A;
    }
}

void fl_launch_info(const FlDev &d, uint8_t *action_required, int32_t *malfunction, uint8_t *state, double *scores, hipStream_t s) {
    const int n = d.B * d.A;
    hipLaunchKernelGGL(k_info, dim3((n + 255) / 256), dim3(256), 0, s, d, action_required, malfunction, state, scores);
}

void fl_launch_reset(const FlDev &d, const uint8_t *mask_dev, int fresh, hipStream_t s) {
    const int n = d.B * d.A;
    hipLaunchKernelGGL(k_reset, dim3((n + 255) / 256), dim3(256), 0, s, d, mask_dev, fresh);
}

// MotionCheck alone on caller-supplied (current cell, wanted cell) lists: one workgroup per case, the same device function
// the step uses (fl_step_body.h motion_check_cells).
__global__ __launch_bounds__(1024) void k_motion_check(const int *__restrict__ offsets, const int *__restrict__ cur,
                                                       const int *__restrict__ nxt, uint8_t *__restrict__ can_move, int S,
                                                       int sshift) {
    extern __shared__ uint32_t lds_raw[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lo = offsets[blockIdx.x], A = offsets[blockIdx.x + 1] - lo;
    StepLds L;
    L.mtl = nullptr; L.words = nullptr;
    L.hkey = (int *)lds_raw; L.hocc = L.hkey + S; L.hwin = L.hocc + S; L.hcnt = L.hwin + S; L.hblk = L.hcnt + S;
    L.a_cur = L.hblk + S; L.a_nxt = L.a_cur + nt; L.misc = L.a_nxt + nt;
    for (int k = tid; k < S; k += nt) { L.hkey[k] = -1; L.hocc[k] = -1; L.hwin[k] = 0x7fffffff; L.hcnt[k] = 0; L.hblk[k] = 0; }
    if (tid < 16) L.misc[tid] = 0;
    __syncthreads();
    const bool act = tid < A;
    const int pos = act ? cur[lo + tid] : -1, np_pos = act ? nxt[lo + tid] : -1;
    const bool blocked = motion_check_cells(L, act, tid, A, pos, np_pos, 1 << 28, S - 1, sshift, tid);
    if (act) can_move[lo + tid] = blocked ? 0 : 1;
}

void fl_launch_motion_check(int n_cases, int max_agents, const int *offsets, const int *cur, const int *nxt, uint8_t *can_move,
                            hipStream_t s) {
    int nt = ((max_agents + 63) / 64) * 64;
    if (nt < 64) nt = 64;
    const StepGeom q = step_geom(max_agents);
    const size_t lds = ((size_t)5 * q.S + 2 * nt + 16) * 4;
    (void)hipFuncSetAttribute((const void *)k_motion_check, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_motion_check, dim3(n_cases), dim3(nt), lds, s, offsets, cur, nxt, can_move, q.S, q.sshift);
}

// more than 64 KiB of dynamic LDS (A > 512) needs the attribute; set once per handle at commit, on the handle's device
int fl_step_prepare() {
    if (hipFuncSetAttribute((const void *)k_step<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    if (hipFuncSetAttribute((const void *)k_step<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    return FL_OK;
}

#ifndef STEP_NT_MIN
#define STEP_NT_MIN 256
#endif
size_t fl_step_lds_bytes(int A) {
    int nt = ((A + 63) / 64) * 64;
    if (nt < STEP_NT_MIN) nt = STEP_NT_MIN;
    return step_lds_words(A, nt) * 4;
}

void fl_launch_step(const FlDev &d, const uint8_t *actions, uint32_t seed, uint32_t stream_base, int synth_kind,
                    int32_t *rewards, uint8_t *dones, uint8_t *done_all, int auto_reset, hipStream_t s) {
    int nt = ((d.A + 63) / 64) * 64;
    if (nt < STEP_NT_MIN) nt = STEP_NT_MIN;
    const StepGeom q = step_geom(d.A);
    const int wcap = q.wcap, S = q.S, sshift = q.sshift;
    const size_t lds = step_lds_words(d.A, nt) * 4;
    if (actions)
        hipLaunchKernelGGL(k_step<false>, dim3(d.B), dim3(nt), lds, s, d, actions, seed, stream_base, synth_kind, rewards,
                           dones, done_all, auto_reset, wcap, S, sshift);
    else
        hipLaunchKernelGGL(k_step<true>, dim3(d.B), dim3(nt), lds, s, d, actions, seed, stream_base, synth_kind, rewards,
                           dones, done_all, auto_reset, wcap, S, sshift);
}
