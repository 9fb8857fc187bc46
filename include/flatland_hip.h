/*
 * flatland_hip.h -- C-ABI of the MI355X-native batched Flatland3 stepper + tree-observation builder.
 *
 * Drop-in boundary for the reference's hot path.  Each entry point names the reference interface it
 * replaces (paths relative to /root/reference):
 *
 *   fl_create / fl_reserve / fl_load_env / fl_commit
 *                                         RailEnv.__init__ + reset() state hand-over, also into a live batch
 *                                         (flatland-rl/flatland/envs/rail_env.py:100-207, 260-357) and
 *                                         flatland_cutils TreeObsForRailEnv.set_env()/reset()
 *                                         (flatland_cutils/src/treeobs.cpp:17-28, loader.cpp:207-219,329-333)
 *   fl_distance_map / _rebuild[_masked]   DistanceMap.get() / reset() + _compute() (flatland/envs/distance_map.py:27-79)
 *   fl_reset                              RailEnv.reset_agents() (rail_env.py:236-241, agent_utils.py:90-105)
 *   fl_step                               RailEnv.step(action_dict) (rail_env.py:501-634)
 *   fl_obs_cutils                         flatland_cutils.TreeObsForRailEnv.get_many() + get_properties()
 *                                         (flatland_cutils/src/main.cpp:17-22, treeobs.cpp:30-108, 612-640)
 *   fl_obs_tree                           flatland.envs.observations.TreeObsForRailEnv.get_many()
 *                                         (flatland/envs/observations.py:60-115)
 *   fl_obs_cutils_policy                  ... the same with adjacency / node_order / edge_order written as the POLICY takes them: int64, the
 *                                         adjacency modified (solution/plfActor.py:48-74 casts, solution/nn/net_tree.py:105-116) -- the one
 *                                         call of the reference's solution (solution/eval_env.py:15-17 builds this builder only)
 *   fl_obs_cutils_handles / fl_obs_tree_handles
 *                                         get_many(handles) with a handle list: the listed agents' predictions only, by list position
 *                                         (flatland_cutils/src/treeobs.cpp:50-62, 393-465; flatland/envs/observations.py:72-83, 337-366)
 *   fl_step_obs                           RailEnv.step() incl. the observations it returns (rail_env.py:634 -> :660-666)
 *   fl_obs_cutils_tree                    both observation builders above in one launch
 *   fl_info                               RailEnv.get_info_dict / action_required (rail_env.py:243-258, 452-468),
 *                                         evaluator scores (flatland/evaluators/service.py:875-879, 900-913)
 *   fl_policy_pack                        plfActor.get_feature + Network.modify_adjacency
 *                                         (solution/plfActor.py:48-74, solution/nn/net_tree.py:105-116)
 *   fl_get_state / fl_get_rng             EnvAgent attribute reads (agent_utils.py:57-88), np_random.get_state()
 *   fl_set_state / fl_get_state_aux       AgentsLoader's per-call read of a caller-owned env (flatland_cutils/src/loader.cpp:8-120,
 *                                         221-327), RailEnvPersister.set_full_state (flatland/envs/persistence.py:182-222)
 *   fl_motion_check                       MotionCheck.addAgent / find_conflicts / check_motion (flatland/envs/agent_chains.py:19-236)
 *
 * Conventions: plain pointers and sizes only.  Pointers named *_dev are DEVICE pointers (hipMalloc /
 * torch CUDA tensors), everything else is host memory.  All buffers are caller-owned; the library keeps
 * its own device-resident state inside the opaque handle.  Calls on one handle are not thread-safe.
 * Every function returns FL_OK (0) or an FL_ERR_* code; fl_last_error() gives the message.
 * Work is enqueued on the handle's HIP stream (fl_set_stream); fl_sync() waits for it.
 */
#ifndef FLATLAND_HIP_H
#define FLATLAND_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fl_batch fl_batch;

enum {
    FL_OK = 0,
    FL_ERR_ARG = 1,          /* bad argument / unsupported size */
    FL_ERR_HIP = 2,          /* HIP runtime error (no device, OOM, launch failure) */
    FL_ERR_EPISODE_DONE = 3, /* RailEnv.step: Exception("Episode is done, cannot call step()") rail_env.py:508-509 */
    FL_ERR_STATE_SYNC = 4,   /* env_utils.state_position_sync_check ValueError, step_utils/env_utils.py:45-52 */
    FL_ERR_ZERO_TRANSITION = 5, /* treeobs.cpp:531-535 std::invalid_argument (0 transitions in tree search) */
    FL_ERR_CAPACITY = 6      /* an internal fixed-capacity buffer overflowed (BFS frontier, prediction index) */
};

/* Limits of this build (FL_ERR_ARG / FL_ERR_CAPACITY beyond them):
 *   FL_MAX_AGENTS          agents per env (one lane per agent in the step kernel)
 *   FL_MAX_SPEED_COUNT     SpeedCounter.max_count = int(1 / speed) - 1 (6 bits of the packed agent word and of a prediction item's
 *                          interval), i.e. speed >= 1/64 (Flatland's own speed maps stop at 1/4)
 *   FL_MAX_RAIL_CELLS      rail cells per env: rail states r * 4 + orientation are u16 (0xFFFF, 0xFFFE reserved).  In practice the
 *                          LDS sets the limit: the distance-map kernel holds an env's neighbour table, bitmaps and queues
 *                          (about 10 900 cells; fl_reserve / fl_commit say so), the observation kernels its rail-cell index
 *                          (about 6 000 cells with 400 agents; the largest Round-2 map, 158 x 158 / 41 cities, has 2 710)
 *   FL_MAX_CUTILS_NODES    flatland_cutils max_nodes (a tree's nodes are the lanes of its team: 32 lanes up to 32 nodes -- the solution
 *                          uses 31 --, a whole wavefront up to 64; beyond 32 the fused entry points run the two builders as two launches)
 *   FL_MAX_PRED_DEPTH      predictor depth of either builder (the solution uses 500 / 30)
 *   FL_MAX_TREE_DEPTH      max_depth of the upstream TreeObsForRailEnv (85 rows at depth 3, 341 at depth 4; depth 4 on grids whose cells
 *                          have at most two transitions per direction -- every Flatland rail cell type --, the builders then run as
 *                          two launches)
 *   every env of one batch shares (A, H, W); an env loaded into a live batch has to fit the capacities of the first
 *   commit (fl_reserve). */
#define FL_MAX_AGENTS 1024
#define FL_MAX_SPEED_COUNT 63
#define FL_MAX_RAIL_CELLS 16383
#define FL_MAX_CUTILS_NODES 64
#define FL_MAX_PRED_DEPTH 500
#define FL_MAX_TREE_DEPTH 4

#define FL_STEP_AUTO_RESET 1
#define FL_STEP_FILTER_REQUIRED 2
#define FL_ACTION_ABSENT 255 /* agent missing from the action dict (rail_env.py:527 -> DO_NOTHING) */
#define FL_STATE_COLS 12     /* row,col,dir,state,malf,nmalf,speed_counter,saved_action,arrival,old_row,old_col,old_dir */
#define FL_AUX_COLS 4        /* prev_state (-1 = None), st_signals.in_malfunction of the last step, deadlocked, done */
#define FL_CUTILS_ATTR 83
#define FL_NODE_FEATURES 12

const char *fl_last_error(void);
int fl_version(void);
/* number of visible HIP devices (0 on a CPU-only host); never initialises a device context */
int fl_device_count(void);

/* B envs, each A agents on an H x W grid, resident on HIP device `device`. */
int fl_create(int B, int A, int H, int W, int device, fl_batch **out);
void fl_destroy(fl_batch *h);
/* run on this hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = HIP's default (null) stream.
 * Until this is called the handle uses a private non-blocking stream. */
int fl_set_stream(fl_batch *h, void *hip_stream);
int fl_sync(fl_batch *h);

/* Stage env b on the host side of the handle.  init_pos/target: int32[A][2] (row, col), both on rail cells; speed: float64[A];
 * malf_threshold = ceil((1 - exp(-rate)) * 2^53), 0 for no malfunctions (malfunction_generators.py:24-53);
 * mt_key[624], mt_pos: numpy RandomState (MT19937) state AFTER reset().
 * After the first fl_commit the same call REPLACES env b of the live batch (RailEnv.reset(regenerate_rail=True,
 * regenerate_schedule=True), rail_env.py:288-320): a new map, new agents, a new RNG state; it takes effect at the next
 * fl_commit.  The new env has to fit the batch's capacities (FL_ERR_CAPACITY otherwise; see fl_reserve).  A refused call
 * leaves the staged env unchanged. */
int fl_load_env(fl_batch *h, int b, const uint16_t *grid, const int32_t *init_pos, const int32_t *init_dir,
                const int32_t *target, const double *speed, const int32_t *earliest, const int32_t *latest,
                int max_episode_steps, uint64_t malf_threshold, int malf_min, int malf_max,
                const uint32_t *mt_key, int mt_pos);
/* Before the first fl_commit: capacity per env for maps loaded later -- unique targets and rail cells.  Default: the
 * largest values among the envs of the first commit. */
int fl_reserve(fl_batch *h, int max_targets, int max_rail_cells);
/* First call: upload all staged envs, build the distance maps and static tables on the GPU, reset all agents.
 * Later calls: the same for exactly the envs loaded since the last commit (the rest of the batch keeps running state);
 * no-op if there are none. */
int fl_commit(fl_batch *h);
/* Per-env RNG replacement after commit (host arrays: mt_key uint32[B][624], mt_pos int32[B]). */
int fl_set_rng(fl_batch *h, const uint32_t *mt_key, const int32_t *mt_pos);
int fl_get_rng(fl_batch *h, uint32_t *mt_key, int32_t *mt_pos);

/* Reset agents (not the RNG, not the maps).  mask: host u8[B] or NULL (= all).  fresh != 0 also clears
 * arrival_time (a freshly loaded env); fresh == 0 follows EnvAgent.reset() literally (agent_utils.py:90-105). */
int fl_reset(fl_batch *h, const uint8_t *mask, int fresh);
/* Same with the mask on the device (u8[B] or NULL), e.g. the done_all tensor of the last step: no host round trip. */
int fl_reset_dev(fl_batch *h, const uint8_t *mask_dev, int fresh);

/* One lock-step tick of all B envs.  actions_dev u8[B][A] (FL_ACTION_ABSENT allowed);
 * rewards_dev int32[B][A], dones_dev u8[B][A], done_all_dev u8[B].
 * auto_reset is a flag word: bit 0 (FL_STEP_AUTO_RESET): an env whose episode ended is reset (fresh) at the start of
 * this call instead of failing with FL_ERR_EPISODE_DONE; bit 1 (FL_STEP_FILTER_REQUIRED): the action of an agent
 * without action_required (rail_env.py:243-258) is ignored, i.e. eval_env.parse_actions (solution/eval_env.py:33-39)
 * fused into the step.  Errors raised inside the kernel surface at fl_check(). */
int fl_step(fl_batch *h, const uint8_t *actions_dev, int32_t *rewards_dev, uint8_t *dones_dev,
            uint8_t *done_all_dev, int auto_reset);
/* Same, with the counter-hash synthetic action stream generated on device (flatland_marl_amd/synth.py):
 * kind 0 = uniform 0..4, 1 = forward-biased, 2 = shortest-path following on the distance map with counter-hash stops (dense
 * traffic, agents arrive); env b uses stream id stream_base + b and its own step counter. */
int fl_step_synth(fl_batch *h, uint32_t seed, uint32_t stream_base, int kind, int32_t *rewards_dev,
                  uint8_t *dones_dev, uint8_t *done_all_dev, int auto_reset);
/* RailEnv.step() returns the observations of the new state (rail_env.py:634 -> _get_observations, :660-666): fl_step /
 * fl_step_synth (actions_dev == NULL: the synthetic stream seed / stream_base / kind) followed by fl_obs_cutils and, when
 * tree_max_depth > 0, fl_obs_tree, as one call (two launches back to back on the handle's stream; a single fused launch
 * measured slower).  flags as fl_step's auto_reset word.  An env whose episode is over (and no auto-reset) gets
 * FL_ERR_EPISODE_DONE at fl_check(); its observation is then that of the unchanged state. */
int fl_step_obs(fl_batch *h, const uint8_t *actions_dev, uint32_t seed, uint32_t stream_base, int kind,
                int32_t *rewards_dev, uint8_t *dones_dev, uint8_t *done_all_dev, int flags, int max_nodes,
                int pred_depth, float *attr_dev, float *forest_dev, int32_t *adjacency_dev, int32_t *node_order_dev,
                int32_t *edge_order_dev, uint8_t *valid_actions_dev, double *props_dev, int tree_max_depth,
                int tree_pred_depth, double *tree_out_dev);
/* fl_obs_cutils + fl_obs_tree in ONE launch (same outputs, bit for bit): the second builder reuses the first one's
 * LDS-resident maps and predicted paths (the upstream predictor's path is a prefix of the cutils one).
 * Requires 0 <= tree_pred_depth <= pred_depth. */
int fl_obs_cutils_tree(fl_batch *h, int max_nodes, int pred_depth, float *attr_dev, float *forest_dev,
                       int32_t *adjacency_dev, int32_t *node_order_dev, int32_t *edge_order_dev,
                       uint8_t *valid_actions_dev, double *props_dev, int tree_max_depth, int tree_pred_depth,
                       double *tree_out_dev);

/* Running sums over all envs since the last reset of the counters, written to out4_dev int64[4] (device):
 * (sum of terminal rewards, arrived agents, agent-steps, finished episodes) -- the scalars the multi-GPU
 * harness all-reduces; mirrors eval_env.final_metric's inputs (solution/eval_env.py:81-94). */
int fl_metrics(fl_batch *h, int64_t *out4_dev, int reset);
/* The evaluator's aggregate scores over every episode finished since the last reset of the counters
 * (flatland/evaluators/service.py:875-879, 900-913: mean_normalized_reward = mean over episodes of 1 + sum(rewards) / (T * A),
 * mean_percentage_complete = mean over episodes of arrived / A), as sums a multi-GPU harness can all-reduce:
 * out3_dev float64[3] (device) = (sum of normalized rewards, sum of completion ratios, episodes).  The per-episode terms
 * are accumulated on the device in double precision in episode order per env and summed over the envs in env order; the episode
 * count is the scores' OWN counter (incremented where the sums are, reset with them): resetting fl_metrics and fl_scores at
 * different times never mixes windows. */
int fl_scores(fl_batch *h, double *out3_dev, int reset);
/* Synchronise and return the first error any kernel recorded (FL_OK if none); clears it. */
int fl_check(fl_batch *h);

/* flatland_cutils observation for all agents of all envs (device outputs):
 * attr f32[B][A][83], forest f32[B][A][max_nodes][12], adjacency i32[B][A][max_nodes-1][3],
 * node_order i32[B][A][max_nodes], edge_order i32[B][A][max_nodes-1], valid_actions u8[B][A][5],
 * props f64[B][A][3] = (dist_target, deadlocked, ready_not_depart) (may be NULL). */
int fl_obs_cutils(fl_batch *h, int max_nodes, int pred_depth, float *attr_dev, float *forest_dev,
                  int32_t *adjacency_dev, int32_t *node_order_dev, int32_t *edge_order_dev,
                  uint8_t *valid_actions_dev, double *props_dev);
/* flatland_cutils get_many(handles) with a STRICT SUBSET of the handles (flatland_cutils/src/treeobs.cpp:50-62): the conflict test of
 * every tree then sees the predictions of the listed agents only, indexed by their POSITION in the list -- it leaves out position
 * `handle` (tool.h:428-434) and reads agents[position].state (treeobs.cpp:413, 435, 455); reproduced as it is.  handles: host
 * int32[n_handles], the same list for every env of the batch; it has to be a permutation of 0 .. n_handles-1 (any other subset is
 * undefined behaviour in the reference: FL_ERR_ARG).  The outputs are those of fl_obs_cutils for ALL agents (row i = agent i;
 * the reference returns the attribute rows of all agents and the trees of the listed ones in list order: the caller gathers). */
int fl_obs_cutils_handles(fl_batch *h, int max_nodes, int pred_depth, const int32_t *handles, int n_handles, float *attr_dev,
                          float *forest_dev, int32_t *adjacency_dev, int32_t *node_order_dev, int32_t *edge_order_dev,
                          uint8_t *valid_actions_dev, double *props_dev);
/* fl_obs_cutils with the three index tensors written as the POLICY NETWORK takes them, in the same launch (no fl_policy_pack pass):
 * int64, the adjacency's parent / child columns offset by tree * max_nodes with tree = b * A + agent, every negative entry -2 (padding rows
 * stay (-2, -2, -2); the action column's -1 becomes -2 too, as `adjacency[adjacency < 0] = -2` does) --
 * plfActor.get_feature's casts (solution/plfActor.py:48-74) + Network.modify_adjacency (solution/nn/net_tree.py:105-116) for a batch
 * flattened over (env, agent).  adjacency i64[B][A][max_nodes-1][3], node_order i64[B][A][max_nodes], edge_order i64[B][A][max_nodes-1];
 * the other outputs as fl_obs_cutils'.  Equal to fl_obs_cutils + fl_policy_pack element for element (tests/test_gpu_fullsize.py). */
int fl_obs_cutils_policy(fl_batch *h, int max_nodes, int pred_depth, float *attr_dev, float *forest_dev, int64_t *adjacency_dev,
                         int64_t *node_order_dev, int64_t *edge_order_dev, uint8_t *valid_actions_dev, double *props_dev);
/* upstream TreeObsForRailEnv(max_depth, ShortestPathPredictorForRailEnv(pred_depth)); pred_depth < 0: no predictor.
 * out f64[B][A][(4^(max_depth+1)-1)/3][12], DFS pre-order (node, L, F, R, B); missing subtree = -inf. */
int fl_obs_tree(fl_batch *h, int max_depth, int pred_depth, double *out_dev);
/* TreeObsForRailEnv.get_many(handles) of the upstream builder with a handle list (observations.py:60-115): predicted_pos / predicted_dir hold
 * the LISTED handles' predictions in list order (:72-83), the conflict test deletes list position `handle` and reads
 * env.agents[position].state (:337-366); reproduced as it is.  handles: host int32[n_handles], the same list for every env; a permutation
 * of 0 .. n_handles-1 (a listed handle >= len(handles) is an IndexError in the reference: FL_ERR_ARG).  The output holds the rows of ALL
 * agents (row i = agent i; the reference returns the listed handles' nodes: the caller picks them). */
int fl_obs_tree_handles(fl_batch *h, int max_depth, int pred_depth, const int32_t *handles, int n_handles, double *out_dev);
/* Opt-in modes of the observation launches of this handle (flags: OR of the values below, 0 = defaults).
 * FL_OBS_KEEP_TREE_ROWS: the caller promises that the upstream-tree output buffer handed to fl_obs_tree / fl_obs_cutils_tree /
 *   fl_step_obs is the buffer of the previous such call with the same max_depth, NOT modified in between.  The builder then stops
 *   re-writing the constant part of its output -- the -inf rows of missing subtrees (observations.py:247, 489: 63 % of the bytes of
 *   a depth-3 tree at 400 agents) -- and only sets the rows that were real nodes in the previous call and are not now.  Same
 *   tensors, bit for bit.  A call with another buffer or depth (or the first one) fills the whole slab as without the flag. */
#define FL_OBS_KEEP_TREE_ROWS 1
int fl_obs_set_mode(fl_batch *h, int flags);

/* RailEnv.get_info_dict (rail_env.py:452-468) as device tensors: action_required u8[B][A], malfunction i32[B][A],
 * state u8[B][A] (any may be NULL), and the evaluator's scores of each env's last finished episode
 * (flatland/evaluators/service.py:875-879,900-913): scores f64[B][2] = (1 + sum(rewards) / (T * A), arrived / A). */
int fl_info(fl_batch *h, uint8_t *action_required_dev, int32_t *malfunction_dev, uint8_t *state_dev, double *scores_dev);

/* Policy-input boundary, stateless (device pointers, enqueued on hip_stream): the int64 tensors
 * plfActor.get_feature builds (solution/plfActor.py:48-74) with Network.modify_adjacency already applied
 * (solution/nn/net_tree.py:105-116: parent/child get the offset (b*A + a) * num_nodes, every negative entry -- the
 * -2 padding AND the action code -1 -- becomes -2).  adjacency i32[B][A][E][3] -> i64, node_order i32[B][A][E+1] -> i64,
 * edge_order i32[B][A][E] -> i64. */
int fl_policy_pack(int B, int A, int E, const int32_t *adjacency_dev, const int32_t *node_order_dev,
                   const int32_t *edge_order_dev, int64_t *adjacency_out_dev, int64_t *node_order_out_dev,
                   int64_t *edge_order_out_dev, void *hip_stream);

/* Host read-backs (synchronising). state int32[B][A][FL_STATE_COLS]; elapsed int32[B]. */
int fl_get_state(fl_batch *h, int32_t *state, int32_t *elapsed);
/* The rest of the dynamic agent state, int32[B][A][FL_AUX_COLS]: state_machine.previous_state (-1 = None),
 * state_machine.st_signals.in_malfunction as of the last step (read by flatland_cutils, loader.cpp:16-18), the sticky
 * DeadlockChecker flag (deadlock_checker.cpp:3-114), dones[handle]. */
int fl_get_state_aux(fl_batch *h, int32_t *aux);
/* Inject the dynamic state of every agent: what AgentsLoader reads from a caller-owned env each call
 * (flatland_cutils/src/loader.cpp:8-120, 221-327) and what RailEnvPersister.set_full_state restores
 * (flatland/envs/persistence.py:182-222).  state int32[B][A][FL_STATE_COLS] as fl_get_state returns it; aux
 * int32[B][A][FL_AUX_COLS] or NULL (= previous_state None, in_malfunction = malf > 0, not deadlocked, done = state DONE);
 * elapsed int32[B] (_elapsed_steps) and done_all u8[B] (dones["__all__"]) or NULL (= unchanged).  Host arrays.
 * A state / position mismatch is refused with FL_ERR_STATE_SYNC (step_utils/env_utils.py:45-52). */
int fl_set_state(fl_batch *h, const int32_t *state, const int32_t *aux, const int32_t *elapsed, const uint8_t *done_all);
/* MotionCheck alone (flatland/envs/agent_chains.py:19-236: addAgent for every agent in handle order, find_conflicts,
 * check_motion) on n_cases independent agent lists: case c = agents offsets[c] .. offsets[c+1]-1 (offsets[0] = 0, at most
 * 1024 agents per case); cur / nxt = current and wanted cell id (>= 0) or -1 for "off the map" (the reference's private
 * virtual node (-1, i)); can_move u8 per agent.  Host arrays; runs the conflict resolution of the step kernel. */
int fl_motion_check(int device, int n_cases, const int32_t *offsets, const int32_t *cur, const int32_t *nxt, uint8_t *can_move);
/* distance map of env b: returns number of unique targets in *n_targets; dm u16[n][H][W][4] (0xFFFF = inf),
 * target_slot int32[A].  dm may be NULL to query n only. */
int fl_distance_map(fl_batch *h, int b, int *n_targets, uint16_t *dm, int32_t *target_slot);
/* Rebuild the distance maps (and the static branch-walk / next-hop tables) of all envs on the GPU from the resident
 * grids: DistanceMap.reset() + _compute() (distance_map.py:47-79). */
int fl_distance_map_rebuild(fl_batch *h);
/* Same for the envs with mask_dev[b] != 0 only (device u8[B], e.g. the done_all tensor of the step that just ran): the
 * rebuild the reference does inside reset() (rail_env.py:288-295, 320), without a host round trip. */
int fl_distance_map_rebuild_masked(fl_batch *h, const uint8_t *mask_dev);
/* RailEnv.agent_positions (rail_env.py:360-367) of env b: int32[H][W], -1 = free */
int fl_positions_map(fl_batch *h, int b, int32_t *out);
/* ALGORITHMIC bytes per agent-step the bench prices the roofline with, for the given obs mix */
double fl_algorithmic_bytes_per_agent_step(fl_batch *h, int with_cutils_obs, int tree_depth);

#ifdef __cplusplus
}
#endif
#endif
