// fl_obs_s9.hip -- FIXED launch class 9 (the flatland_cutils builder alone on large maps, cfg5) for a batch with larger maps among its envs
// (ObsArgs::split): per workgroup the class's body for an env that fits the class, the runtime-carving body for any other (k_obs_split).
#include "fl_obs_body.h"

int fl_obs_launch_s9(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    static_assert(obs_fixed_mode<9>() == 0 && obs_fixed_var<9>() == 2, "class 9 is MODE 0, VAR 2");
    auto kern = k_obs_split<0, 2, 9>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}
