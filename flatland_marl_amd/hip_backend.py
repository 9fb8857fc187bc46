"""ctypes binding of the C-ABI in include/flatland_hip.h (csrc/libflatland_hip.so) and the batched
tensor-level env on top of it.  torch is used for device buffers and streams only.

There is NO CPU fallback: every compute entry point fails loudly when the HIP library is missing or
no GPU is visible.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libflatland_hip.so")

FL_OK = 0
ERR_NAMES = {1: "FL_ERR_ARG", 2: "FL_ERR_HIP", 3: "FL_ERR_EPISODE_DONE", 4: "FL_ERR_STATE_SYNC",
             5: "FL_ERR_ZERO_TRANSITION", 6: "FL_ERR_CAPACITY"}
ACTION_ABSENT = 255
STATE_COLS = 12
AUX_COLS = 4
STATE_NAMES = ("row", "col", "dir", "state", "malf", "nmalf", "scount", "saved", "arrival",
               "old_row", "old_col", "old_dir")

# every symbol include/flatland_hip.h declares
SYMBOLS = ("fl_last_error", "fl_version", "fl_device_count", "fl_create", "fl_destroy", "fl_set_stream", "fl_sync",
           "fl_load_env", "fl_reserve", "fl_commit", "fl_set_rng", "fl_get_rng", "fl_reset", "fl_reset_dev", "fl_step", "fl_step_synth", "fl_step_obs", "fl_check",
           "fl_metrics", "fl_scores", "fl_info", "fl_obs_cutils", "fl_obs_cutils_policy", "fl_obs_cutils_handles", "fl_obs_cutils_tree", "fl_obs_tree", "fl_obs_tree_handles", "fl_obs_set_mode", "fl_policy_pack", "fl_get_state", "fl_get_state_aux", "fl_set_state", "fl_motion_check", "fl_distance_map", "fl_distance_map_rebuild", "fl_distance_map_rebuild_masked", "fl_positions_map",
           "fl_algorithmic_bytes_per_agent_step")

_lib = None


class FlatlandHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, code), msg))
        self.code = code


class EpisodeDoneError(FlatlandHipError):
    """RailEnv.step raises Exception("Episode is done, cannot call step()") (rail_env.py:508-509)."""


def build(force=False):
    env = dict(os.environ)
    if force:
        env["FORCE"] = "1"
    subprocess.check_call([os.path.join(HERE, "csrc", "build.sh")], env=env, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                "%s is missing: build it with flatland_marl_amd/csrc/build.sh (hipcc --offload-arch=gfx950); "
                "there is no CPU fallback" % LIB_PATH)
        # torch ships its own HIP runtime (same soname as /opt/rocm's): it has to be the one this process loads first,
        # otherwise torch.cuda finds no device once the library below has pulled in the system runtime
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        vp, i32, u32, u64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64
        L.fl_last_error.restype = C.c_char_p
        L.fl_create.argtypes = [i32, i32, i32, i32, i32, C.POINTER(vp)]
        L.fl_destroy.argtypes = [vp]
        L.fl_destroy.restype = None
        L.fl_set_stream.argtypes = [vp, vp]
        L.fl_sync.argtypes = [vp]
        L.fl_load_env.argtypes = [vp, i32] + [vp] * 7 + [i32, u64, i32, i32, vp, i32]
        L.fl_commit.argtypes = [vp]
        L.fl_reserve.argtypes = [vp, i32, i32]
        L.fl_distance_map_rebuild_masked.argtypes = [vp, vp]
        L.fl_set_rng.argtypes = [vp, vp, vp]
        L.fl_get_rng.argtypes = [vp, vp, vp]
        L.fl_reset.argtypes = [vp, vp, i32]
        L.fl_reset_dev.argtypes = [vp, vp, i32]
        L.fl_get_state_aux.argtypes = [vp, vp]
        L.fl_set_state.argtypes = [vp, vp, vp, vp, vp]
        L.fl_motion_check.argtypes = [i32, i32, vp, vp, vp, vp]
        L.fl_step.argtypes = [vp, vp, vp, vp, vp, i32]
        L.fl_step_synth.argtypes = [vp, u32, u32, i32, vp, vp, vp, i32]
        L.fl_check.argtypes = [vp]
        L.fl_metrics.argtypes = [vp, vp, i32]
        L.fl_scores.argtypes = [vp, vp, i32]
        L.fl_obs_cutils.argtypes = [vp, i32, i32] + [vp] * 7
        L.fl_obs_tree.argtypes = [vp, i32, i32, vp]
        if hasattr(L, "fl_obs_tree_handles"):
            L.fl_obs_tree_handles.argtypes = [vp, i32, i32, vp, i32, vp]
        if hasattr(L, "fl_obs_cutils_policy"):
            L.fl_obs_cutils_policy.argtypes = [vp, i32, i32] + [vp] * 7
        if hasattr(L, "fl_obs_cutils_handles"):
            L.fl_obs_cutils_handles.argtypes = [vp, i32, i32, vp, i32] + [vp] * 7
        if hasattr(L, "fl_obs_set_mode"):             # (an older build loaded through bench.py --lib for a same-box A/B run has none)
            L.fl_obs_set_mode.argtypes = [vp, i32]
        L.fl_step_obs.argtypes = [vp, vp, u32, u32, i32, vp, vp, vp, i32, i32, i32] + [vp] * 7 + [i32, i32, vp]
        L.fl_obs_cutils_tree.argtypes = [vp, i32, i32] + [vp] * 7 + [i32, i32, vp]
        L.fl_info.argtypes = [vp, vp, vp, vp, vp]
        L.fl_policy_pack.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
        L.fl_get_state.argtypes = [vp, vp, vp]
        L.fl_distance_map.argtypes = [vp, i32, C.POINTER(i32), vp, vp]
        L.fl_distance_map_rebuild.argtypes = [vp]
        L.fl_positions_map.argtypes = [vp, i32, vp]
        if hasattr(L, "fl_debug_last_obs_class"):
            L.fl_debug_last_obs_class.argtypes = [vp, vp]       # diagnostic, not part of the public header
        L.fl_algorithmic_bytes_per_agent_step.argtypes = [vp, i32, i32]
        L.fl_algorithmic_bytes_per_agent_step.restype = C.c_double
        _lib = L
    return _lib


def _chk(rc):
    if rc != FL_OK:
        msg = lib().fl_last_error().decode()
        if rc == 3:
            raise EpisodeDoneError(rc, msg)
        raise FlatlandHipError(rc, msg)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def malf_threshold(rate):
    """ceil((1 - exp(-rate)) * 2**53): `np_random.rand() < _malfunction_prob(rate)` as a 53-bit integer
    compare (malfunction_generators.py:24-33,46-53)."""
    if rate <= 0:
        return 0
    p = float(1 - np.exp(-rate))
    return int(math.ceil(p * 2.0 ** 53))


def motion_check(offsets, cur, nxt, device=0):
    """MotionCheck (agent_chains.py:19-236) on independent agent lists through the step kernel's conflict resolution
    (fl_motion_check): cur / nxt are cell ids or -1 (off the map); returns can_move as a bool array."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    cur = np.ascontiguousarray(cur, dtype=np.int32)
    nxt = np.ascontiguousarray(nxt, dtype=np.int32)
    out = np.zeros(max(len(cur), 1), dtype=np.uint8)
    _chk(lib().fl_motion_check(int(device), len(offsets) - 1, _p(offsets), _p(cur), _p(nxt), _p(out)))
    return out[:len(cur)].astype(bool)


def policy_pack(adjacency, node_order, edge_order, adj_out, no_out, eo_out):
    """int32 device tensors [B,A,E,3] / [B,A,E+1] / [B,A,E] -> int64 outputs (fl_policy_pack) on torch's current stream."""
    import torch
    B, A, E = adjacency.shape[:3]
    s = torch.cuda.current_stream(adjacency.device).cuda_stream
    _chk(lib().fl_policy_pack(B, A, E, adjacency.data_ptr(), node_order.data_ptr(), edge_order.data_ptr(),
                              adj_out.data_ptr(), no_out.data_ptr(), eo_out.data_ptr(), C.c_void_p(s)))


class BatchedRailEnv:
    """B independent Flatland envs stepped in lock-step on one MI355X.

    `envs` is a list of B mappings with the static description of each env, as produced by the
    reference after reset(): grid u16[H,W], init_pos i32[A,2], init_dir i32[A], target i32[A,2],
    speed f64[A], earliest i32[A], latest i32[A], T, malf_rate, malf_min, malf_max, mt_key u32[624], mt_pos.
    All envs of one batch share (A, H, W).  reserve = (max unique targets, max rail cells) leaves room for maps loaded
    into the live batch later (replace_env); default: the largest env of `envs`.
    """

    def __init__(self, envs, device=0, max_nodes=31, pred_depth=500, reserve=None):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedRailEnv needs a HIP device (torch.cuda.is_available() is False); "
                               "the HIP path has no CPU fallback")
        L = lib()
        e0 = envs[0]
        self.B = len(envs)
        self.H, self.W = np.asarray(e0["grid"]).shape
        self.A = int(len(e0["init_dir"]))
        self.device = torch.device("cuda", device)
        self.max_nodes, self.pred_depth = max_nodes, pred_depth
        h = C.c_void_p()
        _chk(L.fl_create(self.B, self.A, self.H, self.W, device, C.byref(h)))
        self.h = h
        self.T = np.zeros(self.B, dtype=np.int32)
        if reserve is not None:
            _chk(L.fl_reserve(h, int(reserve[0]), int(reserve[1])))
        for b, e in enumerate(envs):
            self._load(b, e)
        with torch.cuda.device(self.device):
            _chk(L.fl_commit(h))
        B, A = self.B, self.A
        self.rewards = torch.zeros((B, A), dtype=torch.int32, device=self.device)
        self.dones = torch.zeros((B, A), dtype=torch.uint8, device=self.device)
        self.done_all = torch.zeros((B,), dtype=torch.uint8, device=self.device)
        self._obs = None
        self._tree = {}
        self.use_torch_stream()

    def _load(self, b, e):
        grid = np.ascontiguousarray(e["grid"], dtype=np.uint16)
        assert grid.shape == (self.H, self.W) and len(e["init_dir"]) == self.A
        a32 = lambda k: np.ascontiguousarray(e[k], dtype=np.int32)  # noqa: E731
        ip, idr, tg, ea, la = a32("init_pos"), a32("init_dir"), a32("target"), a32("earliest"), a32("latest")
        sp = np.ascontiguousarray(e["speed"], dtype=np.float64)
        key = np.ascontiguousarray(e["mt_key"], dtype=np.uint32)
        _chk(lib().fl_load_env(self.h, b, _p(grid), _p(ip), _p(idr), _p(tg), _p(sp), _p(ea), _p(la), int(e["T"]),
                               malf_threshold(float(e["malf_rate"])), int(e["malf_min"]), int(e["malf_max"]),
                               _p(key), int(e["mt_pos"])))
        self.T[b] = int(e["T"])

    def replace_env(self, b, env, commit=True):
        """RailEnv.reset(regenerate_rail=True, regenerate_schedule=True) for env b of the live batch (rail_env.py:288-320):
        a new map, new agents and a new RNG state; its distance maps and static tables are rebuilt on the GPU, its agents
        reset, the other envs keep running.  commit=False stages several replacements for one commit()."""
        self._load(b, env)
        if commit:
            self.commit()

    def commit(self):
        with self.torch.cuda.device(self.device):
            _chk(lib().fl_commit(self.h))

    def close(self):
        if getattr(self, "h", None):
            lib().fl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_torch_stream(self):
        """enqueue on torch's current stream so torch ops and the kernels order naturally."""
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        _chk(lib().fl_set_stream(self.h, C.c_void_p(s)))

    # ---- dynamics
    def reset(self, mask=None, fresh=True):
        """mask: host array-like [B], a uint8 device tensor [B] (no host round trip), or None (= all envs)."""
        t = self.torch
        if isinstance(mask, t.Tensor) and mask.is_cuda:
            assert mask.dtype == t.uint8 and mask.numel() == self.B and mask.is_contiguous()
            _chk(lib().fl_reset_dev(self.h, mask.data_ptr(), int(fresh)))
            return
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        _chk(lib().fl_reset(self.h, None if m is None else _p(m), int(fresh)))

    def step(self, actions, auto_reset=False, filter_required=False):
        """actions: uint8 tensor [B, A] on the device (255 = agent not in the action dict).
        filter_required: ignore the actions of agents without action_required (eval_env.parse_actions)."""
        t = self.torch
        if not (isinstance(actions, t.Tensor) and actions.is_cuda):
            actions = t.as_tensor(np.ascontiguousarray(actions, dtype=np.uint8)).to(self.device)
        actions = actions.contiguous()
        assert actions.dtype == t.uint8 and actions.shape == (self.B, self.A)
        _chk(lib().fl_step(self.h, actions.data_ptr(), self.rewards.data_ptr(), self.dones.data_ptr(),
                           self.done_all.data_ptr(), int(bool(auto_reset)) | (2 if filter_required else 0)))
        return self.rewards, self.dones, self.done_all

    def info(self):
        """get_info_dict as device tensors + evaluator scores of each env's last finished episode."""
        t = self.torch
        if not hasattr(self, "_info"):
            B, A = self.B, self.A
            self._info = dict(action_required=t.zeros((B, A), dtype=t.uint8, device=self.device),
                              malfunction=t.zeros((B, A), dtype=t.int32, device=self.device),
                              state=t.zeros((B, A), dtype=t.uint8, device=self.device),
                              scores=t.zeros((B, 2), dtype=t.float64, device=self.device))
        i = self._info
        _chk(lib().fl_info(self.h, i["action_required"].data_ptr(), i["malfunction"].data_ptr(), i["state"].data_ptr(),
                           i["scores"].data_ptr()))
        return i

    def step_obs(self, actions=None, seed=0, stream_base=0, kind=0, auto_reset=False, filter_required=False, tree_depth=0,
                 tree_pred=30):
        """RailEnv.step() as the reference defines it: the tick AND the observations of the new state, one launch.
        actions None: the on-device synthetic stream (seed, stream_base, kind).  Returns (rewards, dones, done_all,
        cutils observation dict, upstream tree tensor or None)."""
        t = self.torch
        ap = None
        if actions is not None:
            if not (isinstance(actions, t.Tensor) and actions.is_cuda):
                actions = t.as_tensor(np.ascontiguousarray(actions, dtype=np.uint8)).to(self.device)
            actions = actions.contiguous()
            assert actions.dtype == t.uint8 and actions.shape == (self.B, self.A)
            ap = actions.data_ptr()
        o = self._obs_buffers()
        tree = None
        if tree_depth > 0:
            key = (tree_depth,)
            if key not in self._tree:
                n = (4 ** (tree_depth + 1) - 1) // 3
                self._tree[key] = t.zeros((self.B, self.A, n, 12), dtype=t.float64, device=self.device)
            tree = self._tree[key]
        _chk(lib().fl_step_obs(self.h, ap, int(seed), int(stream_base), int(kind), self.rewards.data_ptr(),
                               self.dones.data_ptr(), self.done_all.data_ptr(),
                               int(bool(auto_reset)) | (2 if filter_required else 0), self.max_nodes, self.pred_depth,
                               o["agent_attr"].data_ptr(), o["forest"].data_ptr(), o["adjacency"].data_ptr(),
                               o["node_order"].data_ptr(), o["edge_order"].data_ptr(), o["valid_actions"].data_ptr(),
                               o["props"].data_ptr(), int(tree_depth), int(tree_pred),
                               tree.data_ptr() if tree is not None else None))
        return self.rewards, self.dones, self.done_all, o, tree

    def step_synth(self, seed, stream_base=0, kind=0, auto_reset=True):
        _chk(lib().fl_step_synth(self.h, int(seed), int(stream_base), int(kind), self.rewards.data_ptr(),
                                 self.dones.data_ptr(), self.done_all.data_ptr(), int(auto_reset)))
        return self.rewards, self.dones, self.done_all

    def metrics(self, reset=False):
        """int64[4] device tensor: (sum terminal rewards, arrived agents, agent-steps, finished episodes)."""
        if not hasattr(self, "_metrics"):
            self._metrics = self.torch.zeros(4, dtype=self.torch.int64, device=self.device)
        _chk(lib().fl_metrics(self.h, self._metrics.data_ptr(), int(reset)))
        return self._metrics

    def scores(self, reset=False):
        """float64[3] device tensor: (sum of normalized rewards, sum of completion ratios, episodes) over the episodes finished
        since the counters were reset -- the evaluator's mean_normalized_reward / mean_percentage_complete as sums
        (flatland/evaluators/service.py:875-879, 900-913).  The episode count is the scores' own counter, reset with the sums
        (independent of metrics(reset=True))."""
        if not hasattr(self, "_scores"):
            self._scores = self.torch.zeros(3, dtype=self.torch.float64, device=self.device)
        _chk(lib().fl_scores(self.h, self._scores.data_ptr(), int(reset)))
        return self._scores

    def check(self):
        _chk(lib().fl_check(self.h))

    def sync(self):
        _chk(lib().fl_sync(self.h))

    # ---- observations
    def _obs_buffers(self):
        if self._obs is None:
            t, B, A, N = self.torch, self.B, self.A, self.max_nodes
            dev = self.device
            self._obs = dict(
                agent_attr=t.zeros((B, A, 83), dtype=t.float32, device=dev),
                forest=t.zeros((B, A, N, 12), dtype=t.float32, device=dev),
                adjacency=t.zeros((B, A, N - 1, 3), dtype=t.int32, device=dev),
                node_order=t.zeros((B, A, N), dtype=t.int32, device=dev),
                edge_order=t.zeros((B, A, N - 1), dtype=t.int32, device=dev),
                valid_actions=t.zeros((B, A, 5), dtype=t.uint8, device=dev),
                props=t.zeros((B, A, 3), dtype=t.float64, device=dev))
        return self._obs

    def obs_cutils(self, handles=None):
        """flatland_cutils.TreeObsForRailEnv.get_many + get_properties for every agent of every env.  handles: get_many(handles)
        with a strict subset (a permutation of 0 .. n-1, the same list for every env; fl_obs_cutils_handles): the tensors still
        hold every agent's rows, the trees computed against the predictions of the listed agents only."""
        o = self._obs_buffers()
        if handles is not None:
            if not hasattr(lib(), "fl_obs_cutils_handles"):
                raise FlatlandHipError(1, "the loaded library has no fl_obs_cutils_handles (an older build loaded through --lib?)")
            hs = np.ascontiguousarray(handles, dtype=np.int32)
            _chk(lib().fl_obs_cutils_handles(self.h, self.max_nodes, self.pred_depth, _p(hs), len(hs), o["agent_attr"].data_ptr(),
                                             o["forest"].data_ptr(), o["adjacency"].data_ptr(), o["node_order"].data_ptr(),
                                             o["edge_order"].data_ptr(), o["valid_actions"].data_ptr(), o["props"].data_ptr()))
            return o
        _chk(lib().fl_obs_cutils(self.h, self.max_nodes, self.pred_depth, o["agent_attr"].data_ptr(),
                                 o["forest"].data_ptr(), o["adjacency"].data_ptr(), o["node_order"].data_ptr(),
                                 o["edge_order"].data_ptr(), o["valid_actions"].data_ptr(), o["props"].data_ptr()))
        return o

    def obs_both(self, max_depth=2, pred_depth=30):
        """obs_cutils() and obs_tree(max_depth, pred_depth) in one launch; returns (cutils dict, tree tensor)."""
        o = self._obs_buffers()
        n = (4 ** (max_depth + 1) - 1) // 3
        key = (max_depth,)
        if key not in self._tree:
            self._tree[key] = self.torch.zeros((self.B, self.A, n, 12), dtype=self.torch.float64, device=self.device)
        out = self._tree[key]
        _chk(lib().fl_obs_cutils_tree(self.h, self.max_nodes, self.pred_depth, o["agent_attr"].data_ptr(),
                                      o["forest"].data_ptr(), o["adjacency"].data_ptr(), o["node_order"].data_ptr(),
                                      o["edge_order"].data_ptr(), o["valid_actions"].data_ptr(), o["props"].data_ptr(),
                                      max_depth, pred_depth, out.data_ptr()))
        return o, out

    def keep_tree_rows(self, on=True):
        """FL_OBS_KEEP_TREE_ROWS: the upstream-tree tensor this object hands out is its own buffer, the same from call to call -- as long
        as the caller does not write into it, the builder only updates the rows that change (no -inf pre-fill of the slab per call).
        HAZARD: obs_tree / obs_both / step_obs hand out that very tensor; an in-place op on it (replacing -inf before a network, say) breaks
        the promise silently -- clone it first.  FL_OBS_KEEP_VERIFY=1 (environment, diagnostic) checks the promise before every such launch
        and latches an error for check() when a constant row is no longer -inf or a real row is."""
        if not hasattr(lib(), "fl_obs_set_mode"):
            raise FlatlandHipError(1, "the loaded library has no fl_obs_set_mode (an older build loaded through --lib?)")
        _chk(lib().fl_obs_set_mode(self.h, 1 if on else 0))

    def policy_inputs(self, obs=None):
        """(agents_attr f32[B,A,83], forest f32[B,A,N,12], adjacency i64[B,A,N-1,3], node_order i64[B,A,N],
        edge_order i64[B,A,N-1]) on the device, exactly what Network.forward consumes after its own
        modify_adjacency (solution/nn/net_tree.py:72-116); adjacency is already modified."""
        t = self.torch
        o = obs if obs is not None else self.obs_cutils()
        B, A, E = o["adjacency"].shape[:3]
        if not hasattr(self, "_pol") or self._pol[2].shape[-1] != E:
            self._pol = (t.empty((B, A, E, 3), dtype=t.int64, device=self.device),
                         t.empty((B, A, E + 1), dtype=t.int64, device=self.device),
                         t.empty((B, A, E), dtype=t.int64, device=self.device))
        adj, no, eo = self._pol
        policy_pack(o["adjacency"], o["node_order"], o["edge_order"], adj, no, eo)
        return o["agent_attr"], o["forest"], adj, no, eo

    def obs_policy(self):
        """The consumer's call: the flatland_cutils observation with the index tensors as Network.forward takes them, ONE launch
        (fl_obs_cutils_policy) -- (agents_attr f32[B,A,83], forest f32[B,A,N,12], adjacency i64[B,A,N-1,3] already modified,
        node_order i64[B,A,N], edge_order i64[B,A,N-1]); valid_actions / props land in the obs_cutils() buffers.  Equal to
        policy_inputs(obs_cutils()) element for element."""
        t = self.torch
        o = self._obs_buffers()
        B, A, N = self.B, self.A, self.max_nodes
        if not hasattr(self, "_pol64") or self._pol64[1].shape[-1] != N:       # (max_nodes may be set anew by a builder's set_env)
            self._pol64 = (t.empty((B, A, N - 1, 3), dtype=t.int64, device=self.device), t.empty((B, A, N), dtype=t.int64, device=self.device),
                           t.empty((B, A, N - 1), dtype=t.int64, device=self.device))
        adj, no, eo = self._pol64
        L = lib()
        if not hasattr(L, "fl_obs_cutils_policy"):
            raise FlatlandHipError(1, "the loaded library has no fl_obs_cutils_policy (an older build loaded through --lib?)")
        _chk(L.fl_obs_cutils_policy(self.h, self.max_nodes, self.pred_depth, o["agent_attr"].data_ptr(), o["forest"].data_ptr(), adj.data_ptr(),
                                    no.data_ptr(), eo.data_ptr(), o["valid_actions"].data_ptr(), o["props"].data_ptr()))
        return o["agent_attr"], o["forest"], adj, no, eo

    def obs_tree(self, max_depth=2, pred_depth=30, handles=None):
        """upstream TreeObsForRailEnv(max_depth, ShortestPathPredictorForRailEnv(pred_depth)) as a dense tensor.  handles: get_many(handles)
        with a list (a permutation of 0 .. n-1, the same for every env; fl_obs_tree_handles): every agent's rows, the trees computed against
        the predictions of the listed agents only, by list position (observations.py:72-83, 337-366)."""
        n = (4 ** (max_depth + 1) - 1) // 3
        key = (max_depth,)
        if key not in self._tree:
            self._tree[key] = self.torch.zeros((self.B, self.A, n, 12), dtype=self.torch.float64, device=self.device)
        out = self._tree[key]
        if handles is not None:
            L = lib()
            if not hasattr(L, "fl_obs_tree_handles"):
                raise FlatlandHipError(1, "the loaded library has no fl_obs_tree_handles (an older build loaded through --lib?)")
            hs = np.ascontiguousarray(handles, dtype=np.int32)
            _chk(L.fl_obs_tree_handles(self.h, max_depth, pred_depth, _p(hs), len(hs), out.data_ptr()))
            return out
        _chk(lib().fl_obs_tree(self.h, max_depth, pred_depth, out.data_ptr()))
        return out

    # ---- read-backs
    def state(self):
        st = np.zeros((self.B, self.A, STATE_COLS), dtype=np.int32)
        el = np.zeros(self.B, dtype=np.int32)
        _chk(lib().fl_get_state(self.h, _p(st), _p(el)))
        return st, el

    def state_aux(self):
        """int32[B, A, 4]: previous_state (-1 = None), in_malfunction signal of the last step, deadlocked, done."""
        aux = np.zeros((self.B, self.A, AUX_COLS), dtype=np.int32)
        _chk(lib().fl_get_state_aux(self.h, _p(aux)))
        return aux

    def set_state(self, state, aux=None, elapsed=None, done_all=None):
        """inject the dynamic agent state (fl_set_state): state int32[B, A, 12] as state() returns it."""
        state = np.ascontiguousarray(state, dtype=np.int32)
        assert state.shape == (self.B, self.A, STATE_COLS)
        keep = [state]
        if aux is not None:
            aux = np.ascontiguousarray(aux, dtype=np.int32)
            assert aux.shape == (self.B, self.A, AUX_COLS)
        if elapsed is not None:
            elapsed = np.ascontiguousarray(elapsed, dtype=np.int32)
            assert elapsed.shape == (self.B,)
        if done_all is not None:
            done_all = np.ascontiguousarray(done_all, dtype=np.uint8)
            assert done_all.shape == (self.B,)
        keep += [aux, elapsed, done_all]
        _chk(lib().fl_set_state(self.h, _p(state), None if aux is None else _p(aux), None if elapsed is None else _p(elapsed),
                                None if done_all is None else _p(done_all)))

    def rng_state(self):
        key = np.zeros((self.B, 624), dtype=np.uint32)
        pos = np.zeros(self.B, dtype=np.int32)
        _chk(lib().fl_get_rng(self.h, _p(key), _p(pos)))
        return key, pos

    def set_rng_state(self, key, pos):
        key = np.ascontiguousarray(key, dtype=np.uint32)
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        assert key.shape == (self.B, 624) and pos.shape == (self.B,)
        _chk(lib().fl_set_rng(self.h, _p(key), _p(pos)))

    def distance_map(self, b):
        n = C.c_int(0)
        slot = np.zeros(self.A, dtype=np.int32)
        _chk(lib().fl_distance_map(self.h, b, C.byref(n), None, _p(slot)))
        dm = np.zeros((n.value, self.H, self.W, 4), dtype=np.uint16)
        _chk(lib().fl_distance_map(self.h, b, C.byref(n), _p(dm), _p(slot)))
        return dm, slot

    def rebuild_distance_maps(self, mask=None):
        """DistanceMap.reset() + _compute() on the GPU (asynchronous on the handle's stream) for every env, or for the
        envs with a non-zero entry in `mask` (uint8 device tensor [B], e.g. the done_all tensor of the last step)."""
        if mask is None:
            _chk(lib().fl_distance_map_rebuild(self.h))
        else:
            assert mask.is_cuda and mask.dtype == self.torch.uint8 and mask.numel() == self.B and mask.is_contiguous()
            _chk(lib().fl_distance_map_rebuild_masked(self.h, mask.data_ptr()))

    def positions_map(self, b):
        out = np.zeros((self.H, self.W), dtype=np.int32)
        _chk(lib().fl_positions_map(self.h, b, _p(out)))
        return out

    def last_obs_class(self):
        """diagnostic: (fixed launch class, split, envs on the class's body) of the last obs_both / step_obs launch -- class 0 = the
        runtime-carving kernel; split 1 = the class served only the envs that fit it, the others ran the runtime-carving body"""
        out = (C.c_int * 3)()
        _chk(lib().fl_debug_last_obs_class(self.h, out))
        return tuple(out)

    def algorithmic_bytes_per_agent_step(self, with_cutils_obs=True, tree_depth=0):
        return float(lib().fl_algorithmic_bytes_per_agent_step(self.h, int(with_cutils_obs), int(tree_depth)))


class MixedBatch:
    """Envs of DIFFERENT shapes stepped together: the reference's evaluator runs tests of different map sizes and agent counts
    back to back (solution/debug-environments/parameters_flatland_round_2_new.csv: 30x30 / 7 agents ... 158x158 / 425 agents).
    A C-ABI handle holds envs of one (A, H, W); this groups any list of env descriptions by shape into one handle per shape,
    each on a HIP stream of its own (kernels of different shapes overlap on the GPU), and keeps the caller's env order.

    env i lives in group `self.where[i][0]` at batch index `self.where[i][1]`; the per-group tensors are what BatchedRailEnv
    returns, `pick(i, tensors)` gives env i's slice of a per-group result list."""

    def __init__(self, envs, device=0, max_nodes=31, pred_depth=500):
        import torch
        self.torch = torch
        shapes, self.where = {}, []
        for e in envs:
            H, W = np.asarray(e["grid"]).shape
            key = (int(len(e["init_dir"])), int(H), int(W))
            g = shapes.setdefault(key, [])
            self.where.append((key, len(g)))
            g.append(e)
        self.keys = list(shapes)
        self.where = [(self.keys.index(k), b) for k, b in self.where]
        self.streams, self.groups = [], []
        for k in self.keys:
            s = torch.cuda.Stream(device=torch.device("cuda", device))
            with torch.cuda.stream(s):
                self.groups.append(BatchedRailEnv(shapes[k], device=device, max_nodes=max_nodes, pred_depth=pred_depth))
            self.streams.append(s)
        self.n = len(envs)

    def _on_streams(self, fn):
        """fn(k, group) enqueued on every group's own stream (the groups' kernels overlap); the CALLER's current stream then waits for
        all of them, so whatever the caller does next with the returned tensors -- a .cpu(), a torch op, its own kernels -- is ordered
        after the groups' work (the groups' streams are non-blocking: nothing orders them with the caller's stream otherwise)"""
        t = self.torch
        cur = t.cuda.current_stream(self.groups[0].device)
        out = []
        for k, (g, s) in enumerate(zip(self.groups, self.streams)):
            s.wait_stream(cur)                   # ... and the group's work after what the caller enqueued before (e.g. the actions' upload)
            with t.cuda.stream(s):
                out.append(fn(k, g))
        for s in self.streams:
            cur.wait_stream(s)
        return out

    def _each(self, fn):
        return self._on_streams(lambda k, g: fn(g))

    def pick(self, i, per_group):
        g, b = self.where[i]
        r = per_group[g]
        if isinstance(r, dict):
            return {k: v[b] for k, v in r.items()}
        if isinstance(r, tuple):
            return tuple(self.pick_one(x, b) for x in r)
        return r[b]

    @staticmethod
    def pick_one(x, b):
        return {k: v[b] for k, v in x.items()} if isinstance(x, dict) else (None if x is None else x[b])

    def step(self, actions, auto_reset=False, filter_required=False):
        """actions: one uint8 array [A_i] per env (255 = agent not in the dict), in the caller's env order"""
        per = [np.full((g.B, g.A), ACTION_ABSENT, dtype=np.uint8) for g in self.groups]
        for i, a in enumerate(actions):
            g, b = self.where[i]
            per[g][b] = np.asarray(a, dtype=np.uint8)
        return self._on_streams(lambda k, g: g.step(per[k], auto_reset=auto_reset, filter_required=filter_required))

    def stream_of(self, i):
        """stream id of env i in the on-device action stream of step_synth (groups in order, envs of a group consecutive)"""
        g, b = self.where[i]
        return sum(x.B for x in self.groups[:g]) + b

    def step_synth(self, seed, kind=0, auto_reset=True):
        """the on-device action stream; env i uses stream id stream_of(i)"""
        base = [sum(x.B for x in self.groups[:k]) for k in range(len(self.groups))]
        return self._on_streams(lambda k, g: g.step_synth(seed, base[k], kind, auto_reset=auto_reset))

    def obs_cutils(self):
        return self._each(lambda g: g.obs_cutils())

    def obs_both(self, max_depth=2, pred_depth=30):
        return self._each(lambda g: g.obs_both(max_depth, pred_depth))

    def obs_policy(self):
        """the consumer's call per group (fl_obs_cutils_policy): the adjacency of group k is offset over ITS (env, agent) flattening"""
        return self._each(lambda g: g.obs_policy())

    def state(self, i):
        g, b = self.where[i]
        st, el = self.groups[g].state()
        return st[b], int(el[b])

    def metrics(self):
        """int64[4] on the host: the sums over all groups (what a multi-GPU harness all-reduces)"""
        return sum(m.cpu().numpy() for m in self._each(lambda g: g.metrics()))

    def check(self):
        for g in self.groups:
            g.check()

    def sync(self):
        for g in self.groups:
            g.sync()

    def close(self):
        for g in self.groups:
            g.close()
