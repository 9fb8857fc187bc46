// fl_obs_m6.hip -- the observation kernel for MODE 6: MODE 3 (one pass B per round, fl_obs_body.h) with the flatland_cutils builder ALONE --
// no second index, no upstream tables or rows -- the launch the reference's solution makes (solution/eval_env.py:15-17); VAR 0 / 1 / 2.
#include "fl_obs_body.h"

template <typename KernelT>
static int obs_launch(KernelT kern, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}

int fl_obs_launch_m6(int var, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    return var == 1 ? obs_launch(k_obs<6, 1>, d, o, P, s) : var == 2 ? obs_launch(k_obs<6, 2>, d, o, P, s) : obs_launch(k_obs<6, 0>, d, o, P, s);
}
