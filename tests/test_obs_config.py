"""CPU: the launch configuration obs_pick_config (csrc/fl_obs.hip) chooses for the fused observation launch of every bench
workload -- threads, what lives in LDS, one pass B for both builders, compact upstream trees.  The LDS budget (160 KiB a
workgroup) is tight: a few hundred bytes more silently cost a configuration (and 20 % throughput) once, so the choices are
pinned here.  Host code only (fl_debug_obs_config_of needs no GPU)."""
import ctypes

import numpy as np
import pytest

from flatland_marl_amd import hip_backend as hb
from flatland_marl_amd import workload as wl

KEYS = ("nt", "lds", "tab", "nh", "wl", "tmask", "dual", "items", "merged", "compact", "fix")


def _config(A, R, U, depth, tall=0, max_branch=2, pred=500, tree_pred=30, wide=False):
    """wide: a batch of several envs per CU (the launcher then prefers workgroups that fit two a CU for small envs)"""
    L = ctypes.CDLL(hb.LIB_PATH)       # plain dlopen: no torch, no GPU
    out = (ctypes.c_int * 11)()
    fn = L.fl_debug_obs_config_of_wide if wide else L.fl_debug_obs_config_of
    assert fn(A, R, U, tall, max_branch, pred, depth, tree_pred, out) == 0
    return dict(zip(KEYS, out))


def _sizes(workload):
    envs, _ = wl.make_envs(workload, B=len(wl.WORKLOADS[workload]["bases"]))
    A = len(envs[0]["init_dir"])
    R = max(int((np.asarray(e["grid"]) != 0).sum()) for e in envs)
    U = max(len({tuple(t) for t in np.asarray(e["target"]).tolist()}) for e in envs)
    return A, R, U


@pytest.mark.parametrize("workload,depth,expect", [
    # depth 2 is the bench default, depth 3 is BASELINE's definition of configs[2] / configs[4].
    # merged: one pass B for the trees of both builders (1 = one round, 2 = rounds of 32 agents); wl = 0: work lists in HBM scratch;
    # tmask 3 = time masks + the second set of the classify loop's own-path filter; items = entries of the LDS copy of the items
    # fix = 1: the launch class with a compile-time LDS carving (ObsFixed<1>: at most 32 agents / 256 rail cells; items = its capacity)
    ("cfg1", 2, dict(nt=1024, wl=24576, tmask=3, dual=1, items=4096, merged=1, compact=1, fix=1)),
    ("cfg2", 2, dict(nt=1024, wl=24576, tmask=3, dual=1, items=4096, merged=1, compact=1, fix=1)),
    ("cfg3", 3, dict(nt=1024, wl=36864, tmask=3, dual=1, items=4096, merged=2, compact=1, fix=2)),
    ("cfg4", 2, dict(nt=1024, wl=0, tmask=3, dual=1, items=0, merged=2, compact=1, fix=3)),      # (round 5: no LDS copy of the items, an LDS head of the HBM lists)
    ("cfg3", 2, dict(nt=1024, wl=36864, tmask=3, dual=1, items=4096, merged=2, compact=1, fix=12)),   # not BASELINE's depth: class 2's bin twin (round 6)
    ("cfg5", 2, dict(nt=1024, wl=0, tmask=1, dual=0, items=0, merged=0, compact=1, fix=4)),            # (class 4 is a bin: any depth up to 3)
    # the flatland_cutils builder alone (depth 0; round 6): the one-pass kernels without the upstream builder, classes 6 .. 9
    ("cfg1", 0, dict(nt=1024, wl=24576, tmask=3, dual=0, items=4096, merged=1, compact=1, fix=6)),
    ("cfg2", 0, dict(nt=1024, wl=24576, tmask=3, dual=0, items=4096, merged=1, compact=1, fix=6)),
    ("cfg3", 0, dict(nt=1024, wl=36864, tmask=3, dual=0, items=6144, merged=2, compact=1, fix=7)),
    ("cfg4", 0, dict(nt=1024, wl=0, tmask=3, dual=0, items=0, merged=2, compact=1, fix=8)),
    ("cfg5", 0, dict(nt=1024, wl=0, tmask=1, dual=0, items=0, merged=0, compact=1, fix=9)),
    ("cfg5", 3, dict(nt=1024, wl=0, tmask=1, dual=0, items=0, merged=0, compact=1, fix=4)),   # round 2: 512 threads (85-slot tables)
])
def test_observation_launch_configuration_of_the_bench_workloads(workload, depth, expect):
    got = _config(*_sizes(workload), depth)
    assert {k: got[k] for k in expect} == expect, got
    assert got["lds"] <= 160 * 1024


def test_largest_round2_map_keeps_sixteen_wavefronts():
    got = _config(425, 2710, 60, 3)        # Test_14: 158x158, 425 agents, 41 cities (tests/golden/gen_Test_14_L0.npz)
    assert got["nt"] == 1024 and got["compact"] == 1 and got["lds"] <= 160 * 1024, got


def test_grids_with_three_way_cells_take_the_dfs_slot_tables():
    got = _config(20, 213, 5, 3, max_branch=3)
    assert got["compact"] == 0 and got["merged"] == 0, got
    got = _config(20, 213, 5, 2, tall=1)   # colliding prediction keys: two stages
    assert got["merged"] == 0 and got["compact"] == 1, got


@pytest.mark.parametrize("fix,A,R,U,depth", [(1, 32, 256, 8, 2), (2, 80, 232, 10, 3), (3, 80, 680, 24, 2), (4, 432, 2816, 53, 3),
                                             (6, 32, 256, 8, 0), (7, 80, 232, 10, 0), (8, 80, 680, 24, 0), (9, 432, 2816, 53, 0)])
def test_every_fixed_launch_class_is_the_choice_at_its_own_capacities(fix, A, R, U, depth):
    """ObsFixed<k>::opt is hand-written; this keeps it honest: at the class's capacities obs_pick_config's own preference walk
    lands on exactly those options (otherwise the class is never taken and `fix` stays 0), the carving fits 160 KiB, and one
    more agent or rail cell goes to another class (a bin class, round 6) or to the runtime carving with the same kernel code.  depth 0 = the
    flatland_cutils builder alone (classes 6 .. 9)."""
    got = _config(A, R, U, depth)
    assert got["fix"] == fix and got["lds"] <= 160 * 1024, got
    assert _config(A + 1, R, U, depth)["fix"] != fix and _config(A, R + 1, U, depth)["fix"] != fix


def test_fixed_launch_class_boundaries():
    """ObsFixed<1> (compile-time LDS carving of the one-round kernel) is taken exactly when the batch fits its capacities --
    at most 32 agents, 256 rail cells, keys = rail indices -- and obs_pick_config's own choice of options is the class's; any
    other batch runs the runtime-layout kernel (same code, fix = 0)."""
    assert _config(32, 256, 8, 2)["fix"] == 1
    assert _config(32, 257, 8, 2)["fix"] == 0 and _config(32, 257, 8, 2)["merged"] == 1
    assert _config(33, 200, 8, 2)["fix"] == 12                # (rounds of 32 agents: class 2's bin twin)
    assert _config(20, 213, 5, 2, tall=1)["fix"] == 16        # (colliding prediction keys: two stages -- the tall-map bin class)
    assert _config(20, 213, 5, 3)["fix"] == 11                # depth 3 of the upstream tree: not class 1's (BASELINE: depth 2) -- its bin twin
    assert _config(79, 200, 7, 3)["fix"] == 12                # classes 2 and 3 are for exactly 80 agents: 79 take the bin twin
    assert _config(20, 213, 5, 2, tree_pred=60)["fix"] == 0   # 20 * 62 items of the second index exceed the class's 1024
    got = _config(20, 213, 20, 2)
    assert got["fix"] == 1 and got["lds"] <= 160 * 1024      # the next-hop tables (last in the carving) at the batch's size


def test_wide_batches_of_small_envs_run_two_workgroups_a_cu():
    """ObsFixed<5>: class 1's envs (at most 32 agents / 256 rail cells, depth 2) on 512 threads in at most 80 KB of LDS, so that a CU
    holds two workgroups -- taken for a batch of several envs per CU (wide), whatever the batch's own sizes inside the class's
    capacities are; a batch of at most one env per CU keeps class 1; beyond the capacities the runtime carving, still two a CU."""
    for A, R, U in ((32, 256, 8), (20, 213, 5), (7, 138, 2), (17, 100, 3)):
        got = _config(A, R, U, 2, wide=True)
        assert got["fix"] == 5 and got["nt"] == 512 and got["merged"] == 3 and got["lds"] <= 80 * 1024, got
        assert _config(A, R, U, 2)["fix"] == 1
    got = _config(32, 257, 8, 2, wide=True)
    assert got["fix"] == 0 and got["nt"] == 512 and got["lds"] <= 80 * 1024, got
    assert _config(20, 213, 5, 3, wide=True)["fix"] == 0          # depth 3: not the class's
    assert _config(80, 193, 10, 3, wide=True)["fix"] == 2         # envs of more than 32 agents: their own classes, one a CU


def test_small_envs_go_two_workgroups_a_cu_when_that_ends_the_launch_sooner():
    """obs_batch_is_wide (csrc/fl_obs.h): a co-resident pair of 512-thread workgroups takes 1.6 x one 1 024-thread workgroup, so class 5
    is the choice when ceil(B / 2 CUs) pairs end before ceil(B / CUs) single workgroups (profiles/r05_cfg2_bsweep.json)"""
    L = ctypes.CDLL(hb.LIB_PATH)
    wide = {B: bool(L.fl_debug_batch_is_wide(B, 256)) for B in (1, 256, 257, 384, 512, 513, 640, 768, 769, 1024, 1280, 1536, 2048, 8192)}
    assert wide == {1: False, 256: False, 257: True, 384: True, 512: True, 513: False, 640: False, 768: False, 769: True, 1024: True,
                    1280: True, 1536: True, 2048: True, 8192: True}, wide
    assert not L.fl_debug_batch_is_wide(1024, 0)


# (agents, rail cells of level 1, unique targets -- an upper estimate) of the fifteen tests of the Round-2 table
# (solution/debug-environments/parameters_flatland_round_2_new.csv; profiles/r05_soak_round2.txt)
ROUND2_SHAPES = {"Test_0": (7, 120, 4), "Test_1": (10, 99, 4), "Test_2": (20, 123, 6), "Test_3": (50, 183, 6), "Test_4": (80, 177, 10), "Test_5": (80, 239, 14),
                 "Test_6": (80, 362, 18), "Test_7": (80, 422, 26), "Test_8": (80, 677, 34), "Test_9": (100, 1109, 42), "Test_10": (100, 1265, 50),
                 "Test_11": (200, 1443, 58), "Test_12": (200, 2745, 66), "Test_13": (400, 3025, 74), "Test_14": (425, 2807, 82)}


@pytest.mark.parametrize("test", sorted(ROUND2_SHAPES))
def test_every_round2_shape_takes_a_launch_class_at_depth_2_and_3_and_alone(test):
    """round 6: exact classes for the BASELINE configs, BIN classes (compile-time carving, agents an upper bound, the call's depth) for
    every other shape of the table -- with both builders at depth 2 and 3 and for the flatland_cutils builder alone"""
    A, R, U = ROUND2_SHAPES[test]
    tall = int(test in ("Test_3", "Test_6", "Test_9"))      # maps taller than wide: compact prediction keys, two stages (bin classes 16 / 20)
    for depth in (2, 3, 0):
        got = _config(A, R, U, depth, tall=tall)
        assert got["fix"] != 0 and got["lds"] <= 160 * 1024, (test, depth, got)
