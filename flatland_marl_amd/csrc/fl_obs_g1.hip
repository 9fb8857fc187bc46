// fl_obs_g1.hip -- RailEnv.step() and the observations it returns (rail_env.py:501-634 -> :660-666) as ONE launch for the envs of fixed
// launch class 1 (ObsFixed<1>: at most 32 agents / 256 rail cells, both builders in one round of trees -- cfg1, cfg2): the workgroup
// first steps its env (fl_step_body.h, the same device function k_step runs; its MT19937 block, hash table and node lists sit in the
// LDS the trees' node tables take later), then builds the observations of the new state (fl_obs_body.h, class 1's body).  Two kernels'
// worth of launch ramp and drain become one, and the agents' words the builders read were stored a moment ago by the same CU.
#include "fl_obs_body.h"
#include "fl_step_body.h"

template <bool SYNTH>
__global__ __launch_bounds__(OBS_NT) void k_obs_step(FlDev d, FlObsScratch S, ObsArgs P, FlStepArgs Q, int wcap, int hs, int sshift) {
    extern __shared__ __align__(16) unsigned char lds[];
    step_body<SYNTH>(d, Q.actions, Q.seed, Q.stream_base, Q.synth_kind, Q.rewards, Q.dones, Q.done_all, Q.flags, wcap, hs, sshift,
                     reinterpret_cast<uint32_t *>(lds + ObsFixed<1>::L.off[L_WAVE_SCR]));
    __syncthreads();   // (workgroup-scope fence + barrier: the agents' words the step stored are what the builders' snapshot loads)
    obs_kernel_body<3, 0, 1>(d, S, P);
}

int fl_obs_launch_g1(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, const FlStepArgs &step, hipStream_t s) {
    const StepGeom q = step_geom(d.A);
    // the step's LDS (for OBS_NT lanes) inside the node tables' space of the class's carving
    static_assert(ObsFixed<1>::L.off[L_CSR] > ObsFixed<1>::L.off[L_WAVE_SCR], "the node tables are followed by the key offsets");
    if (step_lds_words(d.A, OBS_NT) * 4 > (size_t)(ObsFixed<1>::L.off[L_CSR] - ObsFixed<1>::L.off[L_WAVE_SCR])) return FL_ERR_ARG;
    if (step.actions) {
        auto kern = k_obs_step<false>;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
        hipLaunchKernelGGL(kern, dim3(d.B), dim3(OBS_NT), P.L.total, s, d, o, P, step, q.wcap, q.S, q.sshift);
    } else {
        auto kern = k_obs_step<true>;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
        hipLaunchKernelGGL(kern, dim3(d.B), dim3(OBS_NT), P.L.total, s, d, o, P, step, q.wcap, q.S, q.sshift);
    }
    return FL_OK;
}
