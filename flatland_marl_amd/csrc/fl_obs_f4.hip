// fl_obs_f4.hip -- the observation kernel of FIXED launch class 4 (ObsFixed<4>, fl_obs_layout.h): MODE 2, VAR 2 with the LDS carving
// compiled in -- two stages, at most 400 agents / 2688 rail cells (cfg5).  One translation unit per class (they compile in parallel with the MODE units).
#include "fl_obs_body.h"
static_assert(ObsFixed<4>::L.total <= 160 * 1024 || ObsFixed<4>::opt.nh, "the class's carving fits the LDS of a CU");

int fl_obs_launch_f4(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    auto kern = k_obs<2, 2, 4>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}
