// fl_obs_m5.hip -- the observation kernel for MODE 5: MODE 4 (both builders in one launch, one pass B per round of trees) in rounds of
// 16 agents on 512 threads and at most 80 KB of LDS, so that a CU holds TWO workgroups: one env's barriers and L2 round trips are
// filled by the other env's issue.  VAR 0 / 2 (fl_obs_body.h).
#include "fl_obs_body.h"

template <typename KernelT>
static int obs_launch(KernelT kern, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return FL_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(d.B), dim3(P.L.nt), P.L.total, s, d, o, P);
    return FL_OK;
}

int fl_obs_launch_m5(int var, const FlDev &d, const FlObsScratch &o, const ObsArgs &P, hipStream_t s) {
    if (var == 1) return FL_ERR_ARG;   // (the static tables never join a launch this tight on LDS)
    return var == 2 ? obs_launch(k_obs<5, 2>, d, o, P, s) : obs_launch(k_obs<5, 0>, d, o, P, s);
}
