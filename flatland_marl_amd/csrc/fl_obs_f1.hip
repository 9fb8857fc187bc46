// fl_obs_f1.hip -- the observation kernel of FIXED launch class 1 (ObsFixed<1>, fl_obs_layout.h): MODE 3, VAR 0 with the LDS carving
// compiled in -- one round of trees for both builders, at most 32 agents / 256 rail cells (cfg1, cfg2).  One translation unit per class (they compile in parallel with the MODE units).
#include "fl_obs_body.h"
static_assert(ObsFixed<1>::L.total <= 160 * 1024 || ObsFixed<1>::opt.nh, "the class's carving fits the LDS of a CU");

int fl_obs_launch_f1(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    auto kern = k_obs<3, 0, 1>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}
