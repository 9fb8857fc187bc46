// fl_obs_s9b.hip -- large maps, the flatland_cutils builder alone: class 9's body for the envs that fit it, bin class 19's (no LDS successor
// table, at most 3 072 rail cells / 432 agents) for the others -- ObsArgs::split 2.
#include "fl_obs_body.h"

int fl_obs_launch_s9b(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    static_assert(obs_fixed_mode<9>() == 0 && obs_fixed_var<9>() == 2 && obs_fixed_mode<19>() == 0 && obs_fixed_var<19>() == 2, "classes 9 and 19 are MODE 0, VAR 2");
    auto kern = k_obs_split<0, 2, 9, 19>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}
