"""CPU, build container only (needs the reference mounted at /root/reference): from_reference_env on a REAL reference
RailEnv equals the capture script's extraction, i.e. the committed golden static arrays; the dynamic-state extraction equals
the golden per-step state."""
import os
import sys

import numpy as np
import pytest

from tests import util

REF = "/root/reference/flatland-rl"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF) or not os.path.exists(os.path.join(util.ROOT, "oracle", "_ref")),
                                reason="the reference is only mounted in the build container")


@pytest.fixture(scope="module")
def cap():
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(util.ROOT, "oracle", "refharness"))
    import capture_golden   # puts the stubs, the reference and oracle/_ref on sys.path
    return capture_golden


def test_from_reference_env_equals_golden_static(cap):
    from flatland_marl_amd import from_reference_env, dynamic_state_of_reference_env, synth
    fx = util.load("cfg1_uniform")
    env, mp = cap.make_env(cap.csv_row("Test_0", "Level_0"))
    env.reset()
    got = from_reference_env(env)
    exp = cap.static_arrays(env, mp)
    assert sorted(got) == sorted(exp)
    for k in exp:
        np.testing.assert_array_equal(got[k], exp[k], err_msg=k)
        np.testing.assert_array_equal(got[k], fx[k], err_msg="golden " + k)
    # a few steps of the golden stream: the dynamic state read from the reference objects equals the fixture
    for t in range(25):
        a = synth.uniform_actions(1, 0, t, env.get_num_agents())
        env.step({i: int(a[i]) for i in range(len(a))})
        st, aux, el, da = dynamic_state_of_reference_env(env)
        np.testing.assert_array_equal(st, util.golden_state(fx, t), err_msg=f"step {t}")
        assert el == t + 1 and not da
