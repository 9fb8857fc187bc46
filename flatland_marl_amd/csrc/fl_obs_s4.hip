// fl_obs_s4.hip -- FIXED launch class 4 for a batch with larger maps among its envs (ObsArgs::split): two stages, hundreds of agents (cfg5).  Per workgroup
// the class's body (ObsFixed<4>: compile-time LDS carving) for an env that fits the class, the runtime-carving body for any other
// (k_obs_split, fl_obs_body.h).  One translation unit per class (they compile in parallel with the MODE units).
#include "fl_obs_body.h"

int fl_obs_launch_s4(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    static_assert(obs_fixed_mode<4>() == 2 && obs_fixed_var<4>() == 2, "class 4 is MODE 2, VAR 2");
    auto kern = k_obs_split<2, 2, 4>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}
