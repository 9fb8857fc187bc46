// fl_obs_f2.hip -- the observation kernel of FIXED launch class 2 (ObsFixed<2>, fl_obs_layout.h): MODE 4, VAR 0 with the LDS carving
// compiled in -- rounds of 32 agents, LDS work lists, at most 80 agents / 232 rail cells (cfg3).  One translation unit per class (they compile in parallel with the MODE units).
#include "fl_obs_body.h"
static_assert(ObsFixed<2>::L.total <= 160 * 1024 || ObsFixed<2>::opt.nh, "the class's carving fits the LDS of a CU");

int fl_obs_launch_f2(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    auto kern = k_obs<4, 0, 2>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}
