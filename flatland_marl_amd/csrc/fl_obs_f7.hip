// fl_obs_f7.hip -- the observation kernel of FIXED launch class 7 (ObsFixed<7>, fl_obs_layout.h): the flatland_cutils builder alone with the
// LDS carving compiled in (the counterpart of class 2).  One translation unit per class (they compile in parallel with the MODE units).
#include "fl_obs_body.h"
static_assert(ObsFixed<7>::L.total <= 160 * 1024 || ObsFixed<7>::opt.nh, "the class's carving fits the LDS of a CU");

int fl_obs_launch_f7(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    auto kern = k_obs<obs_fixed_mode<7>(), obs_fixed_var<7>(), 7>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}
