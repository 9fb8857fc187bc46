// fl_obs_s4b.hip -- large maps, both builders (two stages): class 4's body (LDS successor table, at most 2 688 rail cells) for the envs that
// fit it, bin class 14's (no successor table, at most 3 072 rail cells / 432 agents) for the others -- ObsArgs::split 2, every env of the
// launch on a compile-time carving (the levels of cfg5's Round-2 row: 2 680 .. 3 025 rail cells).
#include "fl_obs_body.h"

int fl_obs_launch_s4b(const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    static_assert(obs_fixed_mode<4>() == 2 && obs_fixed_var<4>() == 2 && obs_fixed_mode<14>() == 2 && obs_fixed_var<14>() == 2, "classes 4 and 14 are MODE 2, VAR 2");
    auto kern = k_obs_split<2, 2, 4, 14>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}
