// fl_obs_m3.hip -- the observation kernel for MODE 3 (both builders in one launch, one pass B, envs of at most 32 agents: trees_merged), VAR 0 / 1 / 2 (fl_obs_body.h).  One
// translation unit per MODE so that the three compile in parallel and a phase-level header can be A/B-built on its own.
#include "fl_obs_body.h"

template <typename KernelT>
static int obs_launch(KernelT kern, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}

int fl_obs_launch_m3(int var, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    return var == 1 ? obs_launch(k_obs<3, 1>, d, o, P, s) : var == 2 ? obs_launch(k_obs<3, 2>, d, o, P, s) : obs_launch(k_obs<3, 0>, d, o, P, s);
}
