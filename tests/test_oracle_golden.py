"""CPU: the C oracle (oracle/) against the golden vectors captured from the real reference
(flatland-rl RailEnv.step, DistanceMap, MotionCheck).  Pins the oracle."""
import numpy as np
import pytest

from oracle import orc
from tests import util


@pytest.mark.parametrize("name", util.episode_fixtures())
def test_step_trajectory_matches_reference(name):
    fx = util.load(name)
    e = orc.OracleEnv(fx)
    acts = util.actions_of(fx)
    for t in range(len(acts)):
        rew, done, done_all = e.step(acts[t])
        np.testing.assert_array_equal(e.state(), util.golden_state(fx, t), err_msg=f"{name} step {t}")
        np.testing.assert_array_equal(rew, fx["s_reward"][t], err_msg=f"{name} reward step {t}")
        np.testing.assert_array_equal(done, fx["s_done"][t], err_msg=f"{name} done step {t}")
        assert done_all == bool(fx["done_all"][t])


@pytest.mark.parametrize("name", util.episode_fixtures() + util.base_fixtures())
def test_distance_map_matches_reference(name):
    fx = util.load(name)
    e = orc.OracleEnv(fx)
    dm, slot = e.distance_map()
    np.testing.assert_array_equal(slot, fx["target_slot"])
    np.testing.assert_array_equal(dm, fx["dm_u16"])


def test_distance_map_raw_float64():
    fx = util.load("cfg1_uniform")
    e = orc.OracleEnv(fx)
    dm, slot = e.distance_map()
    full = dm[slot].astype(np.float64)
    full[dm[slot] == 0xFFFF] = np.inf
    np.testing.assert_array_equal(full, fx["dm_f64"])


def test_motion_check_known_answers():
    z = util.load("motioncheck")
    off = z["offsets"]
    for k in range(len(off) - 1):
        s, e = off[k], off[k + 1]
        got = orc.motion_check(z["cur"][s:e], z["nxt"][s:e])
        np.testing.assert_array_equal(got, z["can_move"][s:e].astype(bool), err_msg=f"case {k}")


def test_step_after_done_raises():
    fx = util.load("cfg1_spfollow")
    e = orc.OracleEnv(fx)
    for a in util.actions_of(fx):
        e.step(a)
    with pytest.raises(RuntimeError, match="Episode is done"):
        e.step(fx["actions"][0])


def test_mt19937_matches_numpy():
    rs = np.random.RandomState(1234)
    st = rs.get_state()
    key = np.array(st[1], dtype=np.uint32)
    pos = np.array([st[2]], dtype=np.int32)
    import ctypes as C
    L = orc.lib()
    L.orc_mt_rand.restype = C.c_double
    L.orc_mt_rand.argtypes = [C.c_void_p, C.c_void_p]
    got = [L.orc_mt_rand(key.ctypes.data, pos.ctypes.data) for _ in range(2000)]
    np.testing.assert_array_equal(np.array(got), rs.rand(2000))
    # init_by_array seeding
    key2 = np.zeros(624, dtype=np.uint32)
    pos2 = np.zeros(1, dtype=np.int32)
    init = np.array([7, 11], dtype=np.uint32)
    L.orc_mt_seed_by_array(key2.ctypes.data, pos2.ctypes.data, init.ctypes.data, 2)
    ref = np.random.RandomState([7, 11]).get_state()
    np.testing.assert_array_equal(key2, ref[1])
    assert pos2[0] == ref[2]
