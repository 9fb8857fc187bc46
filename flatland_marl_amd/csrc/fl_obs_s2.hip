// fl_obs_s2.hip -- FIXED launch class 2 for a batch with larger maps among its envs (ObsArgs::split): rounds of 32 agents, work lists in LDS (cfg3).  Per workgroup
// the class's body (ObsFixed<2>: compile-time LDS carving) for an env that fits the class, the runtime-carving body for any other
// (k_obs_split, fl_obs_body.h).  One translation unit per class (they compile in parallel with the MODE units).
#include "fl_obs_body.h"

int fl_obs_launch_s2(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    static_assert(obs_fixed_mode<2>() == 4 && obs_fixed_var<2>() == 0, "class 2 is MODE 4, VAR 0");
    auto kern = k_obs_split<4, 0, 2>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}
