"""CPU: the native reset-time generators (include/flatland_gen.h) against golden vectors captured from the real reference
(oracle/refharness/capture_generators.py): for every level of the five Round-2 parameter rows, the same MT19937 state before
reset() must give the same city positions, rail grid, train stations, agents' lines, timetable, max_episode_steps and the same
MT19937 state after reset() -- bit for bit."""
import ctypes
import glob
import os

import numpy as np
import pytest

from flatland_marl_amd import generators as gen
from tests import util

GEN = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(util.GOLD, "gen_*.npz")))


@pytest.fixture(scope="module", autouse=True)
def built():
    gen.build()


def _run(fx, order="numpy"):
    rg = gen.sparse_rail_generator(max_num_cities=int(fx["max_num_cities"]), grid_mode=bool(fx["grid_mode"]),
                                   max_rails_between_cities=int(fx["max_rails_between_cities"]),
                                   max_rail_pairs_in_city=int(fx["max_rail_pairs_in_city"]),
                                   seed=int(fx["rail_seed"]) if "rail_seed" in fx else None)   # a private rail stream (rail_generators.py:221-222)
    lg = gen.sparse_line_generator(dict(zip(fx["speed_values"].tolist(), fx["speed_probs"].tolist())))
    hints = {}
    st = gen.generate_env(int(fx["width"]), int(fx["height"]), int(fx["n_agents"]), rg, lg, fx["mt_key_before"], int(fx["mt_pos_before"]),
                          float(fx["malf_rate"]), int(fx["malf_min"]), int(fx["malf_max"]), neighbour_order=order, hints=hints)
    return st, hints


def test_header_symbols_exported():
    import re
    hdr = open(os.path.join(util.ROOT, "include", "flatland_gen.h")).read()
    declared = set(re.findall(r"\b(flg_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(gen.SYMBOLS)
    L = ctypes.CDLL(gen.build())
    for s in declared:
        assert hasattr(L, s), s


@pytest.mark.parametrize("name", GEN)
def test_generated_env_equals_the_reference(name):
    fx = util.load(name)
    st, hints = _run(fx)
    np.testing.assert_array_equal(np.array(hints["city_positions"]), fx["city_positions"], err_msg="city positions")
    np.testing.assert_array_equal(hints["neighbour_order"], fx["neighbour_order"], err_msg="np.argsort order of the cities")
    np.testing.assert_array_equal(np.array(hints["city_orientations"]), fx["city_orientations"], err_msg="city orientations")
    for c, lst in enumerate(hints["train_stations"]):
        exp = fx["stations"][c, :fx["n_stations"][c]]
        np.testing.assert_array_equal(np.array([(s[0][0], s[0][1], s[1]) for s in lst]), exp, err_msg=f"stations of city {c}")
    bad = np.argwhere(st["grid"] != fx["grid"])
    assert len(bad) == 0, f"{len(bad)} rail cells differ, first {bad[0].tolist()}: {st['grid'][tuple(bad[0])]:#06x} vs {fx['grid'][tuple(bad[0])]:#06x}"
    for k in ("init_pos", "init_dir", "target", "speed", "earliest", "latest", "T", "malf_rate", "malf_min", "malf_max"):
        np.testing.assert_array_equal(st[k], fx[k], err_msg=k)
    assert int(st["mt_pos"]) == int(fx["mt_pos"]), "MT19937 position after reset()"
    np.testing.assert_array_equal(st["mt_key"], fx["mt_key"], err_msg="MT19937 key after reset()")


def test_seed_to_state_restatement_matches_the_capture_stub():
    """gym's seed -> MT19937 mapping (parity-unpinned: gym is absent); at least it is the one the fixtures were captured with"""
    for name in GEN[:4]:
        fx = util.load(name)
        st = gen.np_random(int(fx["random_seed"])).get_state()
        first = "mt_key_first" in fx      # a later reset(): the seeded state is the one before the FIRST reset
        np.testing.assert_array_equal(st[1], fx["mt_key_first" if first else "mt_key_before"])
        assert st[2] == int(fx["mt_pos_first" if first else "mt_pos_before"])


def test_infeasible_map_raises_like_the_reference():
    rg = gen.sparse_rail_generator(max_num_cities=3)
    st = np.random.RandomState(1).get_state()
    with pytest.raises(gen.GeneratorError, match="Cannot fit more than one city"):
        gen.generate_env(12, 12, 2, rg, gen.sparse_line_generator(), st[1], st[2])


def test_stable_order_equals_numpy_order_without_ties():
    fx = util.load("gen_Test_0_L0")          # two cities: no ties possible
    a, _ = _run(fx, "numpy")
    b, _ = _run(fx, "stable")
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])



@pytest.mark.parametrize("name", [n for n in GEN if "_reset" in n])
def test_consecutive_resets_on_one_stream(name):
    """RailEnv.reset() called again draws a NEW map from the running MT19937 stream: starting from the state the env was
    seeded with, the n-th generate_env() has to land on the reference's n-th reset() (inputs of every reset but the last are
    only the stream state the one before left behind)"""
    fx = util.load(name)
    rg = gen.sparse_rail_generator(max_num_cities=int(fx["max_num_cities"]), grid_mode=bool(fx["grid_mode"]),
                                   max_rails_between_cities=int(fx["max_rails_between_cities"]),
                                   max_rail_pairs_in_city=int(fx["max_rail_pairs_in_city"]))
    lg = gen.sparse_line_generator(dict(zip(fx["speed_values"].tolist(), fx["speed_probs"].tolist())))
    key, pos = fx["mt_key_first"], int(fx["mt_pos_first"])
    st = None
    for k in range(int(fx["resets"])):
        if k == int(fx["resets"]) - 1:
            np.testing.assert_array_equal(key, fx["mt_key_before"])
            assert pos == int(fx["mt_pos_before"])
        st = gen.generate_env(int(fx["width"]), int(fx["height"]), int(fx["n_agents"]), rg, lg, key, pos, float(fx["malf_rate"]),
                              int(fx["malf_min"]), int(fx["malf_max"]))
        key, pos = st["mt_key"], int(st["mt_pos"])
    for k in ("grid", "init_pos", "init_dir", "target", "speed", "earliest", "latest", "T", "mt_key", "mt_pos"):
        np.testing.assert_array_equal(st[k], fx[k], err_msg=k)
