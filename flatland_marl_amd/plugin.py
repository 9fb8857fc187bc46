"""Observation builders that plug into ANY env object through the reference's own plugin API
(`ObservationBuilder.set_env / reset / get_many / get`, flatland/core/env_observation_builder.py:18-73):

    env = flatland.envs.rail_env.RailEnv(..., obs_builder_object=flatland_marl_amd.plugin.TreeObsForRailEnv(31, 500))

`TreeObsForRailEnv(max_nodes, max_pred_depth)` replaces the pybind11 class `flatland_cutils.TreeObsForRailEnv`
(flatland_cutils/src/main.cpp:17-22), `TreeObsUpstream(max_depth, predictor)` replaces
`flatland.envs.observations.TreeObsForRailEnv` (observations.py:34-532).  Like the reference module they keep whatever object
`set_env` hands them (treeobs.cpp:17-21 -- at that time a reference RailEnv has no rail yet, rail_env.py:178-179) and read it
again on every call, duck-typed:

  reset()      static read, as AgentsLoader::reset / RailLoader::reset do (loader.cpp:207-219, 329-333): `env.rail.grid`,
               `env._max_episode_steps`, the agents' line (initial position / direction, target, speed) and timetable ->
               fl_create / fl_load_env / fl_commit (distance maps and static tables are built on the GPU from the grid); a new
               DeadlockChecker, i.e. all sticky flags cleared (loader.cpp:186-199), then one AgentsLoader::update
               (treeobs.cpp:22-28) when some agent is on the map;
  get_many()   dynamic read, as Agent::Agent does per call (loader.cpp:8-120): position, direction, state, the malfunction
               handler, the speed counter, arrival time, old position / direction, `state_machine.st_signals.in_malfunction`,
               `env._elapsed_steps` -> fl_set_state (the builder's own sticky deadlock flags of the previous call carried
               forward, deadlock_checker.cpp:11-29) -> fl_obs_cutils / fl_obs_tree -> the reference's return shapes.

Everything is computed by the HIP kernels through the C-ABI; there is no CPU path (BatchedRailEnv raises without a GPU).
The env's own `distance_map` is not read: the reference's is the BFS of `env.rail.grid` towards the agents' targets
(distance_map.py:57-160), which is what the GPU builds; `verify_distance_map=True` compares the two at reset().
`handles`: `RailEnv` always passes every handle (rail_env.py:665).  With a strict subset the flatland_cutils builder of the reference
keeps only the listed agents' predictions, indexed by list position (treeobs.cpp:50-62): reproduced (fl_obs_cutils_handles;
goldens from the reference in tests/golden/subset_cfg2.npz); lists for which the reference's behaviour is undefined -- a handle >=
len(handles), tool.h:428-434 -- raise ValueError.  The upstream builder does the same since round 6 (fl_obs_tree_handles: observations.py:72-83,
337-366; a handle >= len(handles) is the reference's IndexError).

Cost of a call (tools/plugin_latency.py, profiles/r05_plugin_latency.json): the per-call host work is the reference's own -- one
pass over the agents' attributes -- and nothing that grows with the map: the grid is read and hashed at reset() only, the per-call
check of the static side compares the agents' line / timetable arrays.  `get_many(handles, as_arrays=True)` returns the numpy
arrays themselves instead of the nested lists the pybind11 casters of the reference build.
"""
import time

import numpy as np

from .hip_backend import BatchedRailEnv, FlatlandHipError
from .reference_bridge import static_of_env, agents_static_of_env, dynamic_state_of_env, AGENT_STATIC_KEYS
from . import rail_env as _re


def _agents_signature(st):
    return tuple(np.asarray(st[k]).tobytes() for k in AGENT_STATIC_KEYS)


def _static_signature(st):
    g = np.asarray(st["grid"])
    return (g.shape, g.tobytes()) + _agents_signature(st)


class _EnvBinding:
    """a caller-owned env mirrored into a B = 1 batch on the device"""
    make_batch = BatchedRailEnv          # (tests of the host-side extraction substitute a recorder: there is no CPU compute path)

    def __init__(self, device, verify_distance_map):
        self.device = device
        self.verify = verify_distance_map
        self.batch = None
        self.sig = None
        self.static = None
        self.dead = None          # the DeadlockChecker's sticky flags (deadlock_checker.cpp:3-9)
        self.elapsed = 0
        self.agents_sig = None    # the agents' line / timetable as of the last static read (compared on every call)
        self.profile = None       # a dict: seconds per stage of the calls, accumulated (tools/plugin_latency.py)

    def lap(self, key, t0, sync=False):
        """profiling only: seconds since t0 into stage `key` (sync: after the handle's stream has drained)"""
        if self.profile is None:
            return t0
        if sync:
            self.batch.sync()
        t1 = time.perf_counter()
        self.profile[key] = self.profile.get(key, 0.0) + (t1 - t0)
        return t1

    def load_static(self, env, max_nodes=31, pred_depth=500):
        st = static_of_env(env)
        sig = _static_signature(st)
        if sig != self.sig:
            H, W = st["grid"].shape
            b = self.batch
            if b is not None and (b.H, b.W, b.A) == (H, W, len(st["init_dir"])):
                try:
                    b.replace_env(0, st)          # same shape: into the live handle (fl_load_env + fl_commit)
                except FlatlandHipError as e:
                    if e.code != 6:               # FL_ERR_CAPACITY: more rail cells / targets than the first map -> new handle
                        raise
                    b = None
            else:
                b = None
            if b is None:
                if self.batch is not None:
                    self.batch.close()
                b = self.make_batch([st], device=self.device, max_nodes=max_nodes, pred_depth=pred_depth)
            self.batch, self.sig, self.static = b, sig, st
            if self.verify:
                self._verify_distance_map(env)
        self.agents_sig = _agents_signature(st)
        if self.batch.max_nodes != max_nodes:
            self.batch._obs = None
        self.batch.max_nodes, self.batch.pred_depth = max_nodes, pred_depth
        self.dead = np.zeros(self.batch.A, dtype=np.int32)
        return st

    def _verify_distance_map(self, env):
        dm, slot = self.batch.distance_map(0)
        ours = dm[slot].astype(np.float64)
        ours[dm[slot] == 0xFFFF] = np.inf
        theirs = np.asarray(env.distance_map.get(), dtype=np.float64)
        if theirs.shape != ours.shape or not np.array_equal(ours, theirs):
            raise ValueError("env.distance_map is not the distance map of env.rail.grid towards the agents' targets "
                             "(distance_map.py:57-160): a caller-supplied distance map is not supported")

    def push_dynamic(self, env):
        """Agent::Agent for every agent (loader.cpp:8-120); the line and timetable are re-read too, like the reference does on
        every call (loader.cpp:19-73), and a change of them reloads the static side.  The rail is the reference's RailLoader: read
        at reset() only (loader.cpp:329-333) -- nothing here is proportional to the map."""
        t0 = time.perf_counter() if self.profile is not None else 0.0
        if _agents_signature(agents_static_of_env(env)) != self.agents_sig:
            dead = self.dead
            self.load_static(env, self.batch.max_nodes, self.batch.pred_depth)
            if dead is not None and len(dead) == len(self.dead):
                self.dead = dead                 # the checker object lives until reset() (loader.cpp:186-199)
        state, aux, elapsed = dynamic_state_of_env(env)
        aux[:, 2] = self.dead
        self.elapsed = elapsed
        t0 = self.lap("extract_python", t0)
        self.batch.set_state(state[None], aux[None], np.array([elapsed], dtype=np.int32))
        self.lap("fl_set_state", t0)


class TreeObsForRailEnv(_re.TreeObsForRailEnv):
    """`flatland_cutils.TreeObsForRailEnv(max_nodes, max_pred_depth)` for any env object (treeobs.h:133-169).  Given this
    library's own `RailEnv` it reads the env's device-resident state directly (the base class); any other env is mirrored."""

    def __init__(self, max_nodes=31, max_pred_depth=500, *, device=0, verify_distance_map=False):
        super().__init__(int(max_nodes), int(max_pred_depth))
        self._bind = _EnvBinding(device, verify_distance_map)
        self._cfg = None
        self._native = False

    def set_env(self, env):                      # treeobs.cpp:17-21: keeps the object, reads nothing
        self._native = isinstance(env, _re.RailEnv)
        if self._native:
            return super().set_env(env)
        self.env = env

    def reset(self):                             # treeobs.cpp:22-28
        if self._native:
            return super().reset()
        env = self.env
        st = self._bind.load_static(env, self.max_nodes, self.max_pred_depth)
        H, W = st["grid"].shape
        self._cfg = {"n_agents": len(st["init_dir"]), "max_timesteps": int(st["T"]), "height": int(getattr(env, "height", H)),
                     "width": int(getattr(env, "width", W))}
        self._compute()                          # AgentsLoader::update inside reset() (treeobs.cpp:22-28): the checker sees the state
                                                 # at reset, and get_properties() is valid straight after it

    def _compute(self, handles=None):
        b = self._bind
        b.push_dynamic(self.env)
        t0 = time.perf_counter() if b.profile is not None else 0.0
        o = b.batch.obs_cutils(handles)
        b.batch.check()                          # (synchronises: the kernel has run)
        t0 = b.lap("kernel", t0)
        self._last = {k: v[0].cpu().numpy() for k, v in o.items()}
        b.dead = self._last["props"][:, 1].astype(np.int32)
        b.lap("read_back", t0)
        return self._last

    def get_many(self, handles, as_arrays=False):
        """-> (agent_attr [n][83], (nodes [n][N][12], adjacency [n][N-1][3], node_order [n][N], edge_order [n][N-1])) as nested
        lists, what the pybind11 STL casters return (treeobs.h:160-161, treeobs.cpp:30-108); as_arrays=True: the same five as
        numpy arrays (float32 / int32), without the conversion to Python lists."""
        if self._native:
            return super().get_many(handles)
        if self._bind.batch is None:
            raise RuntimeError("TreeObsForRailEnv.get_many() before reset()")
        h = list(handles)
        # a strict subset: the reference's conflict test then sees the listed agents' predictions only, by list position
        # (treeobs.cpp:50-62) -- fl_obs_cutils_handles; lists the reference has no defined behaviour for raise ValueError
        L = self._compute(_re.cutils_handle_list(h, self._bind.batch.A))
        # the attribute rows of ALL agents (feature_parser.cpp:100-118), the trees of the listed ones in list order (treeobs.cpp:93-101)
        if as_arrays:
            return (L["agent_attr"], (L["forest"][h], L["adjacency"][h], L["node_order"][h], L["edge_order"][h]))
        t0 = time.perf_counter() if self._bind.profile is not None else 0.0
        out = (L["agent_attr"].tolist(),
               (L["forest"][h].tolist(), L["adjacency"][h].tolist(), L["node_order"][h].tolist(), L["edge_order"][h].tolist()))
        self._bind.lap("tolist", t0)
        return out

    def get_properties(self):
        """treeobs.cpp:612-640: the values of the last get_many() / reset()"""
        if self._native:
            return super().get_properties()
        st, L = self._bind.static, self._last
        cfg = dict(self._cfg, curr_step=int(self._bind.elapsed))
        if L is None:
            raise RuntimeError("TreeObsForRailEnv.get_properties() before get_many()")
        props = {"dist_target": L["props"][:, 0].tolist(), "deadlocked": L["props"][:, 1].tolist(),
                 "ready_not_depart": L["props"][:, 2].tolist(),
                 "earliest_departure": [float(v) for v in st["earliest"]],
                 "latest_arrival": [float(v) for v in st["latest"]],
                 "speed": [float(np.float32(v)) for v in st["speed"]]}
        return cfg, props, L["valid_actions"].astype(bool).tolist()

    # the tensors of the last call, for callers that want arrays instead of nested lists
    def last_arrays(self):
        return self._last


class TreeObsUpstream(_re.TreeObsUpstream):
    """`flatland.envs.observations.TreeObsForRailEnv(max_depth, predictor)` for any env object: {handle: Node} with nested
    `childs` dicts (observations.py:117-254, 464-494).  predictor: anything with `max_depth` (the shortest-path predictor,
    predictions.py:91-180, is part of the kernel) or None (no conflict prediction, observations.py:72)."""

    tree_explored_actions_char = ["L", "F", "R", "B"]     # observations.py:45

    def __init__(self, max_depth, predictor=None, *, device=0, verify_distance_map=False):
        super().__init__(max_depth=int(max_depth), pred_depth=-1 if predictor is None else int(predictor.max_depth))
        self.predictor = predictor
        self._bind = _EnvBinding(device, verify_distance_map)
        self._native = False

    def set_env(self, env):                      # observations.py:524-527
        self._native = isinstance(env, _re.RailEnv)
        self.env = env
        if self.predictor is not None and hasattr(self.predictor, "set_env"):
            self.predictor.set_env(env)

    def reset(self):                             # observations.py:57-58: the targets' lookup = the static side
        if not self._native:
            self._bind.load_static(self.env)

    def get_many_dense(self, handles=None):
        if self._native:
            return super().get_many_dense(handles)
        if self._bind.batch is None:
            raise RuntimeError("TreeObsForRailEnv.get_many() before reset()")
        b = self._bind
        b.push_dynamic(self.env)
        pred = -1 if self.predictor is None else int(self.predictor.max_depth)
        hs = None if handles is None else list(handles)
        n = len(self.env.agents)
        # a handle list: the reference's semantics (the listed agents' predictions only, by list position: observations.py:72-83, 337-366)
        t = b.batch.obs_tree(self.max_depth, pred, _re.upstream_handle_list(hs, n, pred >= 0) if hs else None)
        b.batch.check()
        arr = t[0].cpu().numpy()
        return {h: arr[h] for h in (range(arr.shape[0]) if hs is None else hs)}   # (the dense form is this library's own: None = every agent)

    def get_many(self, handles=None):
        if handles is None:
            return {}                            # observations.py:66-67: None -> no handles, no observations
        return {h: _re.nodes_from_dense(a, self.max_depth) for h, a in self.get_many_dense(handles).items()}

    def get(self, handle=0):                     # (see rail_env.TreeObsUpstream.get: the agent's node of get_many(every handle))
        return self.get_many(list(range(len(self.env.agents))))[handle]
