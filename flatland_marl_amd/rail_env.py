"""Dict-shaped facade over the batched HIP env for B = 1: keeps the API surface of the reference's
`flatland.envs.rail_env.RailEnv` (reset()/step(), rail_env.py:260-357, 501-634), of the
`ObservationBuilder` plugin interface (core/env_observation_builder.py:18-73), of the native module
`flatland_cutils.TreeObsForRailEnv` (flatland_cutils/src/main.cpp:17-22) and of the caller
`LocalTestEnvWrapper` (solution/eval_env.py:97-114), so solution/plfActor.py consumes it unchanged.

Everything of step() / the observations is computed by the HIP kernels through the C-ABI; this file only reshapes tensors
into the reference's dicts / lists.  reset(regenerate_rail=True, regenerate_schedule=True) runs the native host generators
(flatland_marl_amd/generators.py -> csrc/gen) on the env's MT19937 stream; an env can also be created from the static
description a reference env has after reset() (RailEnv.from_static, flatland_marl_amd.from_reference_env).
"""
import collections
from enum import IntEnum

import numpy as np

from . import generators
from .hip_backend import BatchedRailEnv, EpisodeDoneError, FlatlandHipError

# flatland.envs.malfunction_generators.MalfunctionParameters / ParamMalfunctionGen (malfunction_generators.py:19-53)
MalfunctionParameters = collections.namedtuple("MalfunctionParameters", ["malfunction_rate", "min_duration", "max_duration"])


class ParamMalfunctionGen:
    def __init__(self, parameters):
        self.MFP = parameters


class NoMalfunctionGen(ParamMalfunctionGen):
    def __init__(self):
        super().__init__(MalfunctionParameters(0.0, 0, 0))


class TrainState(IntEnum):  # flatland/envs/step_utils/states.py:5-25
    WAITING = 0
    READY_TO_DEPART = 1
    MALFUNCTION_OFF_MAP = 2
    MOVING = 3
    STOPPED = 4
    MALFUNCTION = 5
    DONE = 6

    def is_malfunction_state(self):
        return self in (TrainState.MALFUNCTION, TrainState.MALFUNCTION_OFF_MAP)

    def is_off_map_state(self):
        return self in (TrainState.WAITING, TrainState.READY_TO_DEPART, TrainState.MALFUNCTION_OFF_MAP)

    def is_on_map_state(self):
        return self in (TrainState.MOVING, TrainState.STOPPED, TrainState.MALFUNCTION)


class RailEnvActions(IntEnum):  # flatland/envs/rail_env_action.py:5-27
    DO_NOTHING = 0
    MOVE_LEFT = 1
    MOVE_FORWARD = 2
    MOVE_RIGHT = 3
    STOP_MOVING = 4


class _SpeedCounter:  # step_utils/speed_counter.py:4-53 (read-only view)
    def __init__(self, speed):
        self.speed = float(speed)
        self.max_count = int(1 / self.speed) - 1
        self.counter = 0

    @property
    def is_cell_entry(self):
        return self.counter == 0

    @property
    def is_cell_exit(self):
        return self.counter == self.max_count


class _MalfunctionHandler:  # step_utils/malfunction_handler.py:10-67 (read-only view)
    def __init__(self):
        self.malfunction_down_counter = 0
        self.num_malfunctions = 0

    @property
    def in_malfunction(self):
        return self.malfunction_down_counter > 0

    @property
    def malfunction_counter_complete(self):
        return self.malfunction_down_counter == 0


class _ActionSaver:  # step_utils/action_saver.py:4-36 (read-only view)
    def __init__(self):
        self.saved_action = None

    @property
    def is_action_saved(self):
        return self.saved_action is not None


class EnvAgent:
    """read-only mirror of flatland.envs.agent_utils.EnvAgent (agent_utils.py:57-88), refreshed after every step."""

    def __init__(self, handle, st):
        self.handle = handle
        self.initial_position = tuple(int(v) for v in st["init_pos"][handle])
        self.initial_direction = int(st["init_dir"][handle])
        self.target = tuple(int(v) for v in st["target"][handle])
        self.earliest_departure = int(st["earliest"][handle])
        self.latest_arrival = int(st["latest"][handle])
        self.speed_counter = _SpeedCounter(st["speed"][handle])
        self.malfunction_handler = _MalfunctionHandler()
        self.action_saver = _ActionSaver()
        self.position = None
        self.direction = self.initial_direction
        self.old_position = None
        self.old_direction = None
        self.arrival_time = None
        self.state = TrainState.WAITING
        self.moving = False

    def _refresh(self, row):
        r, c, d, s, malf, nmalf, scount, saved, arrival, orow, ocol, odir = (int(v) for v in row)
        self.position = None if r < 0 else (r, c)
        self.direction = d
        self.state = TrainState(s)
        self.malfunction_handler.malfunction_down_counter = malf
        self.malfunction_handler.num_malfunctions = nmalf
        self.speed_counter.counter = scount
        self.action_saver.saved_action = None if saved == 0 else RailEnvActions(saved)
        self.arrival_time = None if arrival < 0 else arrival
        self.old_position = None if orow < 0 else (orow, ocol)
        self.old_direction = None if odir < 0 else odir


class _Rail:
    def __init__(self, grid):
        self.grid = np.array(grid, dtype=np.uint16)
        self.height, self.width = self.grid.shape


class _DistanceMap:
    """DistanceMap.get() (flatland/envs/distance_map.py:27-45): float64 [A, H, W, 4], inf = unreachable."""

    def __init__(self, env):
        self._env = env
        self._cache = None

    def get(self):
        if self._cache is None:
            dm, slot = self._env._batch.distance_map(0)
            full = dm[slot].astype(np.float64)
            full[dm[slot] == 0xFFFF] = np.inf
            self._cache = full
        return self._cache


class ObservationBuilder:
    """core/env_observation_builder.py:18-73"""

    def __init__(self):
        self.env = None

    def set_env(self, env):
        self.env = env

    def reset(self):
        pass

    def get_many(self, handles=None):
        raise NotImplementedError

    def get(self, handle=0):
        """ONE agent's observation (core/env_observation_builder.py:56-73); the builders compute whole batches, so this is
        get_many([handle]) unpacked"""
        return self.get_many([handle])[handle]


def cutils_handle_list(handles, n_agents):
    """get_many(handles) of flatland_cutils: None for "every agent in order" (the usual call, rail_env.py:665), else the list for
    fl_obs_cutils_handles -- a strict subset makes the reference's conflict test work on list POSITIONS (treeobs.cpp:50-62, 393-465),
    which is only defined when the list is a permutation of 0 .. n-1 (tool.h:428-434 erases position `handle`)."""
    h = [int(x) for x in handles]
    if h == list(range(n_agents)) or not h:     # ([]: the reference's `assert((!handles.empty(), "Input Error"))` is a comma expression that never
        return None                             # fires, treeobs.cpp:33 -- it returns every agent's attribute rows and an empty forest)
    if sorted(h) != list(range(len(h))):
        raise ValueError("get_many(handles=%r): the reference's behaviour is undefined for this list -- its conflict test erases list position "
                         "`handle` (flatland_cutils tool.h:428-434), so a strict subset has to be a permutation of 0 .. len(handles)-1" % (h,))
    return h


def upstream_handle_list(handles, n_agents, has_predictor=True):
    """get_many(handles) of the upstream builder: None for "every agent in order" (or no predictor: no conflict test, the list does not
    matter), else the list for fl_obs_tree_handles -- predicted_pos / predicted_dir hold the listed handles' predictions in list order
    (observations.py:72-83) and the conflict test deletes list position `handle` (np.delete, :337): a handle >= len(handles) is an IndexError
    in the reference, raised here too; lists with repeated handles are not reproduced (ValueError)."""
    h = [int(x) for x in handles]
    if not has_predictor or h == list(range(n_agents)):
        return None
    if h and max(h) >= len(h):
        raise IndexError("index %d is out of bounds for axis 0 with size %d (get_many(handles=%r): the reference's conflict test deletes list "
                         "position `handle`, observations.py:337)" % (max(h), len(h), h))
    if not h or sorted(h) != list(range(len(h))):
        raise ValueError("get_many(handles=%r): lists with repeated handles are not reproduced" % (h,))
    return h


class TreeObsForRailEnv(ObservationBuilder):
    """Drop-in for flatland_cutils.TreeObsForRailEnv(max_nodes, max_pred_depth) (treeobs.h:133-169)."""
    checks_errors = True

    def __init__(self, max_nodes=31, max_pred_depth=500):
        super().__init__()
        self.max_nodes, self.max_pred_depth = max_nodes, max_pred_depth
        self._last = None

    def set_env(self, env):
        self.env = env
        if env._batch is not None:            # (before the env's first reset() there is nothing to configure yet: reset() binds again)
            env._batch.max_nodes, env._batch.pred_depth = self.max_nodes, self.max_pred_depth
            env._batch._obs = None

    def get_many(self, handles):
        """-> (agent_attr [A][83], (nodes [A][N][12], adjacency [A][N-1][3], node_order [A][N], edge_order [A][N-1]))
        as nested lists, like the pybind11 STL casters return them (treeobs.h:160-161)."""
        h = list(handles)
        o = self.env._batch.obs_cutils(cutils_handle_list(h, self.env.get_num_agents()))
        self.env._batch.check()       # the one synchronising error check of a step (RailEnv.step leaves it to the builder)
        self._last = {k: v[0].cpu().numpy() for k, v in o.items()}
        L = self._last
        # (the attribute rows of ALL agents, the trees of the listed ones in list order: feature_parser.cpp:100-118, treeobs.cpp:93-101)
        return (L["agent_attr"].tolist(),
                (L["forest"][h].tolist(), L["adjacency"][h].tolist(), L["node_order"][h].tolist(),
                 L["edge_order"][h].tolist()))

    def get(self, handle=0):
        """ONE agent's observation as RailEnv would see it: its row of get_many(every handle) (the pybind11 class has no get();
        get_many([handle]) alone would be the reference's strict-subset semantics, see cutils_handle_list)"""
        attr, (nodes, adj, node_order, edge_order) = self.get_many(range(self.env.get_num_agents()))
        return attr[handle], (nodes[handle], adj[handle], node_order[handle], edge_order[handle])

    def get_properties(self):
        """treeobs.cpp:612-640"""
        e, L = self.env, self._last
        cfg = {"curr_step": e._elapsed_steps, "n_agents": e.get_num_agents(), "max_timesteps": e._max_episode_steps,
               "height": e.height, "width": e.width}
        props = {"dist_target": L["props"][:, 0].tolist(), "deadlocked": L["props"][:, 1].tolist(),
                 "ready_not_depart": L["props"][:, 2].tolist(),
                 "earliest_departure": [float(a.earliest_departure) for a in e.agents],
                 "latest_arrival": [float(a.latest_arrival) for a in e.agents],
                 "speed": [float(np.float32(a.speed_counter.speed)) for a in e.agents]}
        return cfg, props, L["valid_actions"].astype(bool).tolist()


NODE_FIELDS = ("dist_own_target_encountered", "dist_other_target_encountered", "dist_other_agent_encountered",
               "dist_potential_conflict", "dist_unusable_switch", "dist_to_next_branch", "dist_min_to_target",
               "num_agents_same_direction", "num_agents_opposite_direction", "num_agents_malfunctioning",
               "speed_min_fractional", "num_agents_ready_to_depart")
# flatland.envs.observations.Node (observations.py:20-32): the 12 features + the `childs` dict
Node = collections.namedtuple("Node", NODE_FIELDS + ("childs",))


def nodes_from_dense(arr, max_depth):
    """the nested Node namedtuples TreeObsForRailEnv.get() returns (observations.py:117-254, 464-494) from one agent's dense
    [N(max_depth), 12] array in DFS pre-order (node, L, F, R, B): a missing child is -inf (:247, :489), the nodes of the last
    level have an empty `childs` dict (:491-492)."""
    sz = [(4 ** (max_depth - d + 1) - 1) // 3 for d in range(max_depth + 2)]   # rows of a subtree rooted at depth d

    def build(idx, depth):
        childs = {}
        if depth < max_depth:
            for k, ch in enumerate("LFRB"):
                c = idx + 1 + k * sz[depth + 1]
                childs[ch] = build(c, depth + 1) if arr[c, 0] != -np.inf else -np.inf
        return Node(*(float(v) for v in arr[idx]), childs)
    return build(0, 0)


def dense_from_nodes(node, max_depth):
    """inverse of nodes_from_dense (what the golden fixtures store): [N(max_depth), 12], -inf rows for missing subtrees"""
    rows = []

    def walk(n, depth):
        if not isinstance(n, tuple):
            rows.extend([[-np.inf] * 12] * ((4 ** (max_depth - depth + 1) - 1) // 3))
            return
        rows.append([getattr(n, f) for f in NODE_FIELDS])
        if depth < max_depth:
            for ch in "LFRB":
                walk(n.childs.get(ch, -np.inf), depth + 1)
    walk(node, 0)
    return np.array(rows, dtype=np.float64)


class TreeObsUpstream(ObservationBuilder):
    """flatland.envs.observations.TreeObsForRailEnv(max_depth, ShortestPathPredictorForRailEnv(pred_depth))
    (observations.py:34-532).  get_many() returns what the reference returns: {handle: Node}, nested namedtuples with a
    `childs` dict ('L', 'F', 'R', 'B' -> Node or -inf).  get_many_dense() / the batched tensor API keep the dense form the
    kernel writes: float64 [N(max_depth), 12] per agent in DFS pre-order (node, L, F, R, B), missing subtree = -inf."""

    checks_errors = True
    FIELDS = NODE_FIELDS
    tree_explored_actions_char = ["L", "F", "R", "B"]     # observations.py:41

    def __init__(self, max_depth=2, pred_depth=30, predictor=None):
        super().__init__()
        self.max_depth = max_depth
        self.pred_depth = pred_depth if predictor is None else getattr(predictor, "max_depth", pred_depth)
        self.observation_dim = 11                         # observations.py:46

    def get_many_dense(self, handles=None):
        handles = list(range(self.env.get_num_agents())) if handles is None else list(handles)
        t = self.env._batch.obs_tree(self.max_depth, self.pred_depth, upstream_handle_list(handles, self.env.get_num_agents(), self.pred_depth >= 0) if handles else None)
        self.env._batch.check()
        arr = t[0].cpu().numpy()
        return {h: arr[h] for h in handles}

    def get_many(self, handles=None):
        if handles is None:
            return {}                 # observations.py:66-67: None -> no handles, no observations (get_many_dense(None): every agent)
        return {h: nodes_from_dense(a, self.max_depth) for h, a in self.get_many_dense(handles).items()}

    def get(self, handle=0):
        """observations.py:117-254 computes ONE agent's tree against the predictions the last get_many() prepared -- RailEnv's call with
        every handle (rail_env.py:665): the agent's node of get_many(every handle), not get_many([handle]) (a one-entry prediction list)"""
        return self.get_many(list(range(len(self.env.agents))))[handle]


class _FromDescription:
    """what `env.rail_generator` / `env.line_generator` are on an env made from a description or a file (the reference:
    rail_from_file / line_from_file closures, rail_generators.py:116-145, line_generators.py:168-206): hands the env's own rail /
    line back"""

    def __init__(self, what):
        self.what = what

    def __repr__(self):
        return "<%s of the loaded description>" % self.what


class RailEnv:
    """B = 1 view with the reference's constructor, attribute and method surface (rail_env.py:35-777)."""

    def __init__(self, width, height, rail_generator=None, line_generator=None, number_of_agents=2, obs_builder_object=None,
                 malfunction_generator_and_process_data=None, malfunction_generator=None, remove_agents_at_target=True,
                 random_seed=None, record_steps=False, *, device=0):
        """Same arguments as flatland.envs.rail_env.RailEnv (rail_env.py:100-112).  rail_generator / line_generator:
        generators.sparse_rail_generator(...) / sparse_line_generator(...) (the Round-2 generators; others: build the env
        with the reference and use RailEnv.from_static).  Nothing is generated before reset(), like the reference."""
        if not remove_agents_at_target:
            raise NotImplementedError("remove_agents_at_target=False is unused by the solution and not supported")
        if malfunction_generator_and_process_data is not None:
            raise NotImplementedError("the deprecated malfunction closures are not supported; pass malfunction_generator")
        self.width, self.height = int(width), int(height)
        self.rail_generator = rail_generator if rail_generator is not None else generators.sparse_rail_generator()
        self.line_generator = line_generator if line_generator is not None else generators.sparse_line_generator()
        self.number_of_agents = int(number_of_agents)
        self.malfunction_generator = malfunction_generator if malfunction_generator is not None else NoMalfunctionGen()
        self.remove_agents_at_target = True
        self.record_steps = record_steps
        self._device = device
        self.obs_builder = obs_builder_object if obs_builder_object is not None else TreeObsForRailEnv()
        self.np_random = None
        self.random_seed = None
        if random_seed:
            self._seed(random_seed)
        self.num_resets = 0
        self._static, self._hints, self._batch = None, None, None
        self._from_static = self._from_file = False
        self.rail, self.agents, self.distance_map = None, [], None
        self._max_episode_steps = None
        self._elapsed_steps = 0
        self.dones = dict.fromkeys(list(range(self.number_of_agents)) + ["__all__"], False)
        self.rewards_dict = {}
        self.obs_dict = None

    @classmethod
    def from_static(cls, static, obs_builder_object=None, device=0, from_file=False):
        """an env from the static description a (reference) env has after reset(): no generators involved.
        from_file: the description comes from an env FILE (RailEnvPersister.load_new, persistence.py:105-129): reset() with
        regenerate_rail or regenerate_schedule then behaves like the reference's rail_from_file / line_from_file env -- the same
        rail and line, the timetable drawn again from the env's MT19937 stream (see generators.redraw_timetable)."""
        H, W = np.asarray(static["grid"]).shape
        mfp = MalfunctionParameters(float(static["malf_rate"]), int(static["malf_min"]), int(static["malf_max"]))
        env = cls(W, H, number_of_agents=len(static["init_dir"]), obs_builder_object=obs_builder_object,
                  malfunction_generator=ParamMalfunctionGen(mfp), device=device)
        # no generators to run: reset() re-adopts this same rail and line (solution/eval_env.py:102 calls env.reset() on such an env)
        env.rail_generator, env.line_generator = _FromDescription("rail"), _FromDescription("line")
        env._from_static = True
        env._from_file = bool(from_file)
        env._adopt(static)
        return env

    def _seed(self, seed):  # rail_env.py:210-222
        self.np_random = generators.np_random(seed)
        self.random_seed = seed
        return [seed]

    def _rng_state(self):
        """the env's MT19937 stream: on the device while a batch exists (the step's malfunction draws advance it there)"""
        if self._batch is not None:
            key, pos = self._batch.rng_state()
            return key[0], int(pos[0])
        if self.np_random is None:
            self.np_random = np.random.RandomState()     # unseeded env, like gym's seeding.np_random(None)
        st = self.np_random.get_state()
        return np.asarray(st[1], dtype=np.uint32), int(st[2])

    def _adopt(self, static):
        """(re)create the B = 1 batch for a new static description"""
        if self._batch is not None:
            self._batch.close()
        self._static = static
        self._batch = BatchedRailEnv([static], device=self._device)
        self.rail = _Rail(static["grid"])
        self.height, self.width = self.rail.height, self.rail.width
        self.number_of_agents = self._batch.A
        self._max_episode_steps = int(static["T"])
        self.agents = [EnvAgent(i, static) for i in range(self.number_of_agents)]
        self.distance_map = _DistanceMap(self)
        self.obs_builder.set_env(self)

    def get_num_agents(self):
        return len(self.agents)

    @property
    def agent_positions(self):
        """RailEnv.agent_positions (rail_env.py:341-342, 360-367): int [H, W], the handle of the agent on a cell, -1 = free"""
        return self._batch.positions_map(0)

    def get_agent_handles(self):
        return range(self.get_num_agents())

    def action_required(self, agent):  # rail_env.py:243-258
        return agent.state == TrainState.READY_TO_DEPART or \
            (agent.state.is_on_map_state() and agent.speed_counter.is_cell_entry)

    def _refresh(self):
        st, el = self._batch.state()
        for a, row in zip(self.agents, st[0]):
            a._refresh(row)
        self._elapsed_steps = int(el[0])

    def get_info_dict(self):  # rail_env.py:452-468
        return {"action_required": {i: self.action_required(a) for i, a in enumerate(self.agents)},
                "malfunction": {i: a.malfunction_handler.malfunction_down_counter for i, a in enumerate(self.agents)},
                "speed": {i: a.speed_counter.speed for i, a in enumerate(self.agents)},
                "state": {i: a.state for i, a in enumerate(self.agents)}}

    def _get_observations(self):  # rail_env.py:660-666
        self.obs_dict = self.obs_builder.get_many(list(range(self.get_num_agents())))
        return self.obs_dict

    def reset(self, regenerate_rail=True, regenerate_schedule=True, *, random_seed=None):
        """rail_env.py:260-357.  regenerate_rail: a new map, lines and timetable on the env's own MT19937 stream; neither flag:
        EnvAgent.reset() for every agent; regenerate_schedule alone fails like the reference does with the sparse generators."""
        if random_seed:
            self._seed(random_seed)
            if self._batch is not None:       # the device copy of the stream follows the re-seed
                st = self.np_random.get_state()
                self._batch.set_rng_state(np.asarray(st[1], dtype=np.uint32)[None], np.array([st[2]], dtype=np.int32))
        mfp = self.malfunction_generator.MFP
        if self._from_static and self._from_file and (regenerate_rail or regenerate_schedule):
            # an env loaded from a FILE (rail_from_file + line_from_file): the same rail and line come back, the agents are fresh
            # (EnvAgent.from_line, rail_env.py:315-317) and timetable_generator draws earliest_departure / latest_arrival /
            # max_episode_steps AGAIN from the env's stream, without agents_hints (rail_env.py:310-331)
            key, pos = self._rng_state()
            self._adopt(generators.redraw_timetable(self._static, key, pos, num_cities=2))
        elif self._from_static:
            # an env made from a description: the same map, lines and timetable again; regenerate_schedule makes fresh agents,
            # without it EnvAgent.reset() keeps arrival_time (agent_utils.py:90-105).  The MT19937 stream runs on.
            self._batch.reset(fresh=bool(regenerate_rail or regenerate_schedule))
        elif regenerate_rail or self._static is None:
            key, pos = self._rng_state()
            hints = {}
            static = generators.generate_env(self.width, self.height, self.number_of_agents, self.rail_generator, self.line_generator,
                                             key, pos, mfp.malfunction_rate, mfp.min_duration, mfp.max_duration, hints=hints)
            self._hints = hints
            self._adopt(static)
        elif regenerate_schedule:
            # the reference passes no hints to the line generator on this path and SparseLineGen fails on them
            # (rail_env.py:311-316, line_generators.py:96): same error here
            raise TypeError("'NoneType' object is not subscriptable")
        else:
            self._batch.reset(fresh=False)      # EnvAgent.reset() literally: arrival_time survives (agent_utils.py:90-105)
        self.num_resets += 1
        self.obs_builder.set_env(self)      # rail_env.py:305 (a builder assigned to env.obs_builder after construction is bound here)
        self.obs_builder.reset()
        self.dones = dict.fromkeys(list(range(self.number_of_agents)) + ["__all__"], False)
        self.rewards_dict = {i: 0 for i in range(self.number_of_agents)}
        self._refresh()
        return self._get_observations(), self.get_info_dict()

    def step(self, action_dict_):  # rail_env.py:501-634
        action_dict = action_dict_
        if self.dones["__all__"]:
            self._elapsed_steps += 1
            raise Exception("Episode is done, cannot call step()")
        acts = np.full((1, self.number_of_agents), 255, dtype=np.uint8)
        for i, a in action_dict.items():
            if 0 <= int(i) < self.number_of_agents:
                v = int(a)
                acts[0, int(i)] = v if 0 <= v <= 4 else 7   # illegal values become DO_NOTHING inside the kernel
        rew, done, done_all = self._batch.step(acts)
        try:
            if not getattr(self.obs_builder, "checks_errors", False):
                self._batch.check()
            obs = self._get_observations()      # the builders call fl_check themselves: one sync + read-back per step
        except EpisodeDoneError as e:
            raise Exception("Episode is done, cannot call step()") from e
        except FlatlandHipError as e:
            if e.code == 4:
                raise ValueError(str(e)) from e
            raise
        rew, done = rew[0].cpu().numpy(), done[0].cpu().numpy()
        self.rewards_dict = {i: int(rew[i]) for i in range(self.number_of_agents)}
        for i in range(self.number_of_agents):
            self.dones[i] = bool(done[i])
        self.dones["__all__"] = bool(done_all[0].item())
        self._refresh()
        self.obs_dict = obs
        return obs, self.rewards_dict, self.dones, self.get_info_dict()


class LocalTestEnvWrapper:
    """Counterpart of solution/eval_env.py:9-114 (TestEnvWrapper / LocalTestEnvWrapper)."""

    def __init__(self, env):
        self.env = env
        self.obs_properties = {}

    def action_required(self):
        return {i: self.env.action_required(a) for i, a in enumerate(self.env.agents)}

    def parse_actions(self, actions):  # eval_env.py:33-39
        req = self.action_required()
        return {idx: act for idx, act in actions.items() if req[idx]}

    def update_obs_properties(self):  # eval_env.py:56-62
        cfg, props, valid = self.env.obs_builder.get_properties()
        self.obs_properties = {}
        self.obs_properties.update(cfg)
        self.obs_properties.update(props)
        self.obs_properties["valid_actions"] = valid

    @staticmethod
    def parse_features(feature, obs_properties):  # eval_env.py:64-79
        fl = {"agent_attr": np.array(feature[0]), "forest": np.array(feature[1][0])}
        fl["forest"][fl["forest"] == np.inf] = -1
        fl["adjacency"] = np.array(feature[1][1])
        fl["node_order"] = np.array(feature[1][2])
        fl["edge_order"] = np.array(feature[1][3])
        fl.update(obs_properties)
        return fl

    def get_valid_actions(self):
        return self.obs_properties["valid_actions"]

    def reset(self):
        feature, _ = self.env.reset()
        self.update_obs_properties()
        return [self.parse_features(feature, self.obs_properties)]

    def step(self, actions):
        actions = self.parse_actions(actions)
        feature, reward, done, info = self.env.step(actions)
        self.update_obs_properties()
        return [self.parse_features(feature, self.obs_properties)], reward, done

    def final_metric(self):  # eval_env.py:81-94 (counts every off-map, non-READY agent as "arrived")
        assert self.env.dones["__all__"]
        env = self.env
        n_arrival = sum(1 for a in env.agents if a.position is None and a.state != TrainState.READY_TO_DEPART)
        total_reward = sum(env.rewards_dict.values())
        norm_reward = 1 + total_reward / env._max_episode_steps / env.get_num_agents()
        return n_arrival / env.get_num_agents(), total_reward, norm_reward
