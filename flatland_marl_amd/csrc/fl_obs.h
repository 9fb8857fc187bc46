// fl_obs.h -- observation kernels' host interface (scratch owned by the batch handle).
#pragma once
#include <vector>

#include "fl_internal.h"

#define FL_OBS_MAX_NODES 64  /* a tree's nodes are the lanes of one team: 32 lanes (two trees a wavefront) up to 32 nodes, a whole wavefront up to 64 */
#define FL_OBS_MAX_PRED 500

// A batch is "wide" when small envs (at most 32 agents) are better off as TWO 512-thread workgroups a CU (class 5) than as one 1 024-thread
// workgroup (class 1): a co-resident pair takes 1.6 x what one 1 024-thread workgroup takes alone (same-box sweep at the cfg2 shape,
// profiles/r05_cfg2_bsweep.json: 65 against 40 us), so the launch is ceil(B / 2 CUs) pairs against ceil(B / CUs) single workgroups --
// two a CU from 257 envs on except at three and five envs a CU (round 5, first: from four envs a CU on; 384 envs +4 %, 512 +2.6 %)
static inline bool obs_batch_is_wide(int B, int n_cu) {
    if (n_cu <= 0) return false;
    const long long pairs = (B + 2LL * n_cu - 1) / (2LL * n_cu), singles = (B + (long long)n_cu - 1) / n_cu;
    return 8 * pairs < 5 * singles;
}

struct FlObsScratch {
    int pred_cap;      // waypoints kept per agent (pred_depth + 2)
    uint16_t *path;    // [B][A][pred_cap] predicted waypoints: rail state (rail index << 2 | direction)
    long long *dbg;    // [B][32] phase clocks of diagnostic builds (-DFL_OBS_TIMING)
    uint32_t *cell_items;  // [B][items_cap] prediction items (IT_* packing, fl_obs.hip) when they do not fit LDS
    size_t items_cap;
    uint16_t *bk_rel;  // [B][8][Rcap + 1] ends of the keys' items inside every time bucket (large maps: bucket-major items, see OBS_BK_NB)
    uint2 *wl;         // [B][wl_cap] pass B work lists of the large-map kernels
    int wl_cap;
    uint32_t *cost;    // [B] clock ticks of the env's previous observation launch (its workgroup writes them)
    int *order;        // [B] env of workgroup k, longest first (k_env_order), or null: env k (batches of at most one env per CU)
    int n_cu;          // CUs of the device
    unsigned order_age;  // ordered launches so far: the order is recomputed every OBS_ORDER_EVERY-th (host side)
    uint4 *rowmask;    // [B][A] the rows of the agent's upstream tree that were real nodes in the previous launch (x, y, z: bits 0 .. 95 of the
                       // DFS row index; w: 1 = valid) -- see FL_OBS_KEEP_TREE_ROWS
    const double *rows_out; int rows_depth;   // host side: buffer and depth the masks describe (null: none yet)
    int keep_rows;     // host side: fl_obs_set_mode(FL_OBS_KEEP_TREE_ROWS)
    int16_t *label;    // [A] get_many(handles) with a strict subset: position of agent i in the list or -1 (fl_obs_cutils_handles uploads it per call)
    const int *h_R;    // HOST [B] rail cells of every env (the handle's copy; null: unknown) -- which envs of a batch fit a fixed launch class
    int last_fix, last_split, last_fit;  // host side, diagnostic: class of the last fused launch (0 = runtime carving), whether the class
                                         // served only the envs that fit it, and how many envs took the class's body
};

int fl_obs_alloc(FlObsScratch &o, const FlDev &d, hipStream_t s, std::vector<void *> &allocs);
int fl_launch_obs_cutils(FlObsScratch &o, const FlDev &d, int max_nodes, int pred_depth, float *attr, float *forest,
                         int32_t *adjacency, int32_t *node_order, int32_t *edge_order, uint8_t *valid, double *props,
                         hipStream_t s, const int16_t *label_dev = nullptr, int out64 = 0);   // out64: adjacency / node_order / edge_order are int64 policy tensors
int fl_launch_obs_both(FlObsScratch &o, const FlDev &d, int max_nodes, int pred_depth, float *attr, float *forest,
                       int32_t *adjacency, int32_t *node_order, int32_t *edge_order, uint8_t *valid, double *props,
                       int max_depth, int tree_pred, double *tree_out, hipStream_t s);
int fl_launch_obs_tree(FlObsScratch &o, const FlDev &d, int max_depth, int pred_depth, double *out, hipStream_t s, const int16_t *label_dev = nullptr);
// more envs than CUs: the order in which the workgroups take the envs (longest previous launch first); returns the scratch the launch uses
FlObsScratch fl_obs_env_order(FlObsScratch &o, const FlDev &d, hipStream_t s);
int fl_obs_config_of_fused(const FlDev &d, int pred_depth, int max_depth, int tree_pred, int out[11], int wide = 0);  // diagnostic (wide: several envs per CU)
