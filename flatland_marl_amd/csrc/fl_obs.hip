// placeholder until the observation kernels land
#include "fl_obs.h"
#include "../../include/flatland_hip.h"
int fl_obs_alloc(FlObsScratch &o, const FlDev &d, hipStream_t s, std::vector<void *> &allocs) { return FL_OK; }
void fl_obs_reset(FlObsScratch &o, const FlDev &d, const uint8_t *mask_dev, hipStream_t s) {}
int fl_launch_obs_cutils(FlObsScratch &o, const FlDev &d, int max_nodes, int pred_depth, float *attr, float *forest,
                         int32_t *adjacency, int32_t *node_order, int32_t *edge_order, uint8_t *valid, double *props,
                         hipStream_t s) { return FL_ERR_ARG; }
int fl_launch_obs_tree(FlObsScratch &o, const FlDev &d, int max_depth, int pred_depth, double *out, hipStream_t s) { return FL_ERR_ARG; }
