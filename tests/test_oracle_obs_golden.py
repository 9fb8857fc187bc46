"""CPU: the oracle's observation builders against golden tensors captured from the reference's own
flatland_cutils C++ module (get_many + get_properties) and from upstream flatland TreeObsForRailEnv.
Bit-exact, float32/float64 values included."""
import numpy as np
import pytest

from oracle import orc
from tests import util

CUTILS_KEYS = (("attr", "o_attr"), ("forest", "o_forest"), ("adjacency", "o_adjacency"),
               ("node_order", "o_node_order"), ("edge_order", "o_edge_order"), ("valid", "o_valid"))


@pytest.mark.parametrize("name", util.episode_fixtures())
def test_obs_match_reference(name):
    fx = util.load(name)
    e = orc.OracleEnv(fx)
    obs_steps = {int(t): k for k, t in enumerate(fx["obs_steps"])}
    py_steps = {int(t): k for k, t in enumerate(fx["py_steps"])} if "py_steps" in fx.files else {}
    pykeys = [k for k in fx.files if k.startswith("py_d")]
    n_checked = 0

    def check(t):
        nonlocal n_checked
        if t in obs_steps:
            k = obs_steps[t]
            o = e.obs_cutils(31, 500)
            for got, key in CUTILS_KEYS:
                np.testing.assert_array_equal(o[got], fx[key][k], err_msg=f"{name} t={t} {got}")
            np.testing.assert_array_equal(o["props"][:, 0], fx["o_p_dist_target"][k])
            np.testing.assert_array_equal(o["props"][:, 1], fx["o_p_deadlocked"][k])
            np.testing.assert_array_equal(o["props"][:, 2], fx["o_p_ready"][k])
            n_checked += 1
        if t in py_steps:
            k = py_steps[t]
            for pk in pykeys:
                depth, pdepth = int(pk.split("_")[1][1:]), int(pk.split("_")[2][1:])
                np.testing.assert_array_equal(e.obs_pytree(depth, pdepth), fx[pk][k], err_msg=f"{name} t={t} {pk}")

    check(0)
    for t, a in enumerate(util.actions_of(fx)):
        e.step(a)
        # the deadlock flags of flatland_cutils are sticky and updated once per get_many(), i.e. once per step
        if (t + 1) not in obs_steps:
            e.obs_cutils(31, 500)
        check(t + 1)
    assert n_checked == len(obs_steps)
