// fl_obs_f3.hip -- the observation kernel of FIXED launch class 3 (ObsFixed<3>, fl_obs_layout.h): MODE 4, VAR 2 with the LDS carving
// compiled in -- rounds of 32 agents, work lists in HBM scratch, at most 80 agents / 656 rail cells (cfg4).  One translation unit per class (they compile in parallel with the MODE units).
#include "fl_obs_body.h"
static_assert(ObsFixed<3>::L.total <= 160 * 1024 || ObsFixed<3>::opt.nh, "the class's carving fits the LDS of a CU");

int fl_obs_launch_f3(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    auto kern = k_obs<4, 2, 3>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}
