// fl_obs_f21.hip -- the observation kernel of BIN launch class 21 (ObsFixed<21>, fl_obs_layout.h): compile-time LDS carving, the agents an
// upper bound and the upstream depth the call's.  One translation unit per class (they compile in parallel with the MODE units).
#include "fl_obs_body.h"
static_assert(ObsFixed<21>::L.total <= 160 * 1024 || ObsFixed<21>::opt.nh, "the class's carving fits the LDS of a CU");

int fl_obs_launch_f21(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    auto kern = k_obs<obs_fixed_mode<21>(), obs_fixed_var<21>(), 21>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}
