// fl_obs_f5.hip -- the observation kernel of FIXED launch class 5 (ObsFixed<5>, fl_obs_layout.h): MODE 5, VAR 0 with the LDS carving
// compiled in -- class 1's envs (at most 32 agents / 256 rail cells) in rounds of 16 agents on 512 threads, two workgroups a CU, for
// batches of several envs per CU.  One translation unit per class (they compile in parallel with the MODE units).
#include "fl_obs_body.h"
static_assert(ObsFixed<5>::L.total <= 160 * 1024 || ObsFixed<5>::opt.nh, "the class's carving fits the LDS of a CU");

int fl_obs_launch_f5(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    static_assert(obs_fixed_mode<5>() == 5 && obs_fixed_var<5>() == 0 && ObsFixed<5>::L.total <= 80 * 1024, "class 5 is MODE 5, VAR 0, two workgroups a CU");
    auto kern = k_obs<5, 0, 5>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}
