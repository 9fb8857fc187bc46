/* placeholder: replaced below */
#include "fl_oracle_internal.h"
void orc_obs_cutils_reset(OrcEnv *e) { memset(e->deadlocked, 0, (size_t)e->A); }
int orc_obs_cutils(OrcEnv *e, int max_nodes, int pred_depth, float *attr, float *forest, int32_t *adjacency,
                   int32_t *node_order, int32_t *edge_order, uint8_t *valid, double *props) { return ORC_ERR_ARG; }
int orc_obs_pytree(OrcEnv *e, int max_depth, int pred_depth, double *out) { return ORC_ERR_ARG; }
