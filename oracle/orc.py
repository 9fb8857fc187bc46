"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libfl_oracle.so")
STATE_COLS = 12
STATE_NAMES = ("row", "col", "dir", "state", "malf", "nmalf", "scount", "saved", "arrival",
               "old_row", "old_col", "old_dir")
_lib = None


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in ("fl_oracle.c", "fl_oracle_obs.c", "fl_oracle.h", "fl_oracle_internal.h")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "libfl_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_create.argtypes = [C.c_int] * 3
        _lib.orc_destroy.argtypes = [C.c_void_p]
        _lib.orc_load.argtypes = [C.c_void_p] + [C.c_void_p] * 7 + [C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_int]
        _lib.orc_set_rng.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib.orc_get_rng.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_reset.argtypes = [C.c_void_p]
        _lib.orc_step.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        _lib.orc_get_state.argtypes = [C.c_void_p, C.c_void_p]
        _lib.orc_elapsed.argtypes = [C.c_void_p]
        _lib.orc_last_error.restype = C.c_char_p
        _lib.orc_num_targets.argtypes = [C.c_void_p]
        _lib.orc_get_distance_map.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_distance_map_bfs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.orc_motion_check.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_obs_cutils_reset.argtypes = [C.c_void_p]
        _lib.orc_obs_cutils.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7
        _lib.orc_obs_pytree_handles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _lib.orc_obs_cutils_handles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 7
        _lib.orc_obs_pytree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.orc_mt_seed_by_array.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def malf_threshold(rate):
    """ceil((1 - exp(-rate)) * 2**53) as an integer; 0 when rate <= 0 (malfunction_generators.py:24-33)."""
    if rate <= 0:
        return 0
    p = 1 - np.exp(-rate)
    import math
    return int(math.ceil(float(p) * 2.0 ** 53))


class OracleEnv:
    """One scalar reference-semantics env (mirrors RailEnv.reset/step at array level)."""

    def __init__(self, fx):
        """fx: mapping with the static arrays of a golden fixture (grid, init_pos, ...)."""
        L = lib()
        self.grid = np.ascontiguousarray(fx["grid"], dtype=np.uint16)
        self.H, self.W = self.grid.shape
        self.A = int(len(fx["init_dir"]))
        self.T = int(fx["T"])
        self.h = L.orc_create(self.H, self.W, self.A)
        a32 = lambda k: np.ascontiguousarray(fx[k], dtype=np.int32)  # noqa: E731
        self._keep = [a32("init_pos"), a32("init_dir"), a32("target"),
                      np.ascontiguousarray(fx["speed"], dtype=np.float64), a32("earliest"), a32("latest"),
                      np.ascontiguousarray(fx["mt_key"], dtype=np.uint32)]
        k = self._keep
        thr = malf_threshold(float(fx["malf_rate"]))
        rc = L.orc_load(self.h, _p(self.grid), _p(k[0]), _p(k[1]), _p(k[2]), _p(k[3]), _p(k[4]), _p(k[5]),
                        self.T, thr, int(fx["malf_min"]), int(fx["malf_max"]), _p(k[6]), int(fx["mt_pos"]))
        assert rc == 0

    def __del__(self):
        try:
            lib().orc_destroy(self.h)
        except Exception:
            pass

    def reset(self):
        lib().orc_reset(self.h)

    def set_rng(self, key, pos):
        key = np.ascontiguousarray(key, dtype=np.uint32)
        lib().orc_set_rng(self.h, _p(key), int(pos))

    def get_rng(self):
        key = np.zeros(624, dtype=np.uint32)
        pos = np.zeros(1, dtype=np.int32)
        lib().orc_get_rng(self.h, _p(key), _p(pos))
        return key, int(pos[0])

    def step(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.uint8)
        rew = np.zeros(self.A, dtype=np.int32)
        done = np.zeros(self.A, dtype=np.uint8)
        da = np.zeros(1, dtype=np.uint8)
        rc = lib().orc_step(self.h, _p(actions), _p(rew), _p(done), _p(da))
        if rc != 0:
            raise RuntimeError(lib().orc_last_error().decode())
        return rew, done, bool(da[0])

    def state(self):
        out = np.zeros((self.A, STATE_COLS), dtype=np.int32)
        lib().orc_get_state(self.h, _p(out))
        return out

    def distance_map(self):
        U = lib().orc_num_targets(self.h)
        dm = np.zeros((U, self.H, self.W, 4), dtype=np.uint16)
        slot = np.zeros(self.A, dtype=np.int32)
        lib().orc_get_distance_map(self.h, _p(dm), _p(slot))
        return dm, slot

    def obs_cutils(self, max_nodes=31, pred_depth=500, handles=None):
        """handles: get_many(handles) with a strict subset (a permutation of 0 .. n-1, treeobs.cpp:50-62): the trees of ALL agents are
        returned (row i = agent i), computed against the predictions of the listed handles only"""
        A, N = self.A, max_nodes
        out = dict(attr=np.zeros((A, 83), np.float32), forest=np.zeros((A, N, 12), np.float32),
                   adjacency=np.zeros((A, N - 1, 3), np.int32), node_order=np.zeros((A, N), np.int32),
                   edge_order=np.zeros((A, N - 1), np.int32), valid=np.zeros((A, 5), np.uint8),
                   props=np.zeros((A, 3), np.float64))
        if handles is not None:
            hs = np.ascontiguousarray(handles, dtype=np.int32)
            assert sorted(hs.tolist()) == list(range(len(hs))), "a strict subset has to be a permutation of 0 .. n-1"
            rc = lib().orc_obs_cutils_handles(self.h, N, pred_depth, _p(hs), len(hs), *[_p(out[k]) for k in
                                              ("attr", "forest", "adjacency", "node_order", "edge_order", "valid", "props")])
        else:
            rc = lib().orc_obs_cutils(self.h, N, pred_depth, *[_p(out[k]) for k in
                                      ("attr", "forest", "adjacency", "node_order", "edge_order", "valid", "props")])
        if rc != 0:
            raise RuntimeError("orc_obs_cutils rc=%d %s" % (rc, lib().orc_last_error().decode()))
        return out

    def obs_pytree(self, max_depth, pred_depth, handles=None):
        """handles: get_many(handles) of the upstream builder with a list (a permutation of 0 .. n-1, observations.py:60-115): the trees of
        ALL agents (row i = agent i) against the predictions of the listed handles only"""
        n = (4 ** (max_depth + 1) - 1) // 3
        out = np.zeros((self.A, n, 12), dtype=np.float64)
        if handles is not None:
            hs = np.ascontiguousarray(handles, dtype=np.int32)
            rc = lib().orc_obs_pytree_handles(self.h, max_depth, pred_depth, _p(hs), len(hs), _p(out))
        else:
            rc = lib().orc_obs_pytree(self.h, max_depth, pred_depth, _p(out))
        if rc != 0:
            raise RuntimeError("orc_obs_pytree rc=%d" % rc)
        return out


def motion_check(cur, nxt):
    cur = np.ascontiguousarray(cur, dtype=np.int32)
    nxt = np.ascontiguousarray(nxt, dtype=np.int32)
    out = np.zeros(len(cur), dtype=np.uint8)
    lib().orc_motion_check(len(cur), _p(cur), _p(nxt), _p(out))
    return out.astype(bool)


def distance_map_bfs(grid, target):
    grid = np.ascontiguousarray(grid, dtype=np.uint16)
    H, W = grid.shape
    out = np.zeros((H, W, 4), dtype=np.uint16)
    lib().orc_distance_map_bfs(_p(grid), H, W, int(target[0]), int(target[1]), _p(out))
    return out
