"""Error-path parity on MALFORMED maps (fixtures captured from the real reference by oracle/refharness/capture_malformed.py):
a branch walk that enters a rail cell without a transition for its direction, and a transition that leaves the rail.
flatland_cutils raises std::invalid_argument there (treeobs.cpp:528-535) -> FL_ERR_ZERO_TRANSITION at fl_check(); the upstream
TreeObsForRailEnv prints and makes the node terminal (observations.py:420-425) -> the trees have to equal the golden ones."""
import numpy as np
import pytest

from tests import util

NAMES = ("malformed_zero_transition", "malformed_leaves_rail")


@pytest.mark.parametrize("name", NAMES)
def test_oracle_on_malformed_maps(name):
    from oracle import orc
    fx = util.load(name)
    e = orc.OracleEnv(fx)
    assert fx["cutils_raised"].any() and fx["python_printed"].any()
    for t in range(len(fx["state"])):
        if t > 0:
            rew, done, _ = e.step(fx["actions"][t - 1])
            np.testing.assert_array_equal(rew, fx["reward"][t - 1])
        np.testing.assert_array_equal(e.state(), fx["state"][t], err_msg=f"t={t}")
        for d in (2, 3):
            np.testing.assert_array_equal(e.obs_pytree(d, 10), fx["py_d%d_p10" % d][t], err_msg=f"t={t} depth {d}")
        if fx["cutils_raised"][t]:
            with pytest.raises(RuntimeError, match="WRONG CELL TYPE"):
                e.obs_cutils(31, 500)
        else:
            e.obs_cutils(31, 500)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("fused", [False, True])
def test_kernels_on_malformed_maps(name, fused):
    import torch
    from flatland_marl_amd.hip_backend import BatchedRailEnv, FlatlandHipError
    fx = util.load(name)
    env = BatchedRailEnv([util.static_of(fx)])
    assert str(fx["cutils_message"]).startswith("WRONG CELL TYPE detected in tree-search (0 transitions possible)")
    for t in range(len(fx["state"])):
        if t > 0:
            rew, done, _ = env.step(torch.from_numpy(fx["actions"][t - 1][None, :].copy()).cuda())
            np.testing.assert_array_equal(rew.cpu().numpy()[0], fx["reward"][t - 1])
            np.testing.assert_array_equal(done.cpu().numpy()[0], fx["done"][t - 1])
        np.testing.assert_array_equal(env.state()[0][0], fx["state"][t], err_msg=f"t={t}")
        env.check()                                   # the step itself raises nothing
        for d in (2, 3):                              # the upstream builder does not raise: terminal nodes, like the reference's
            if fused:
                _, tree = env.obs_both(d, 10)
            else:
                tree = env.obs_tree(d, 10)
            np.testing.assert_array_equal(tree.cpu().numpy()[0], fx["py_d%d_p10" % d][t], err_msg=f"t={t} depth {d}")
            if not fused:
                env.check()
        if not fused:
            env.obs_cutils()
        if fx["cutils_raised"][t]:                    # the cutils builder of the launch raised, as flatland_cutils does
            with pytest.raises(FlatlandHipError, match="FL_ERR_ZERO_TRANSITION.*WRONG CELL TYPE"):
                env.check()
        else:
            env.check()
    env.close()


@pytest.mark.parametrize("name", ["threeway_cfg2", "threeway_cfg3"])
def test_oracle_on_a_grid_with_a_three_way_cell(name):
    """(oracle/refharness/capture_threeway.py: the real reference on a generated map whose first switch got a third way on)"""
    from oracle import orc
    fx = util.load(name)
    e = orc.OracleEnv(fx)
    dm, slot = e.distance_map()
    np.testing.assert_array_equal(dm, fx["dm_u16"])
    for t in range(len(fx["state"])):
        if t > 0:
            e.step(fx["actions"][t - 1])
        np.testing.assert_array_equal(e.state(), fx["state"][t], err_msg=f"t={t}")
        o = e.obs_cutils(31, 500)
        for k in ("attr", "forest", "adjacency", "node_order", "edge_order", "valid"):
            np.testing.assert_array_equal(o[k], fx["o_" + k][t], err_msg=f"t={t} {k}")
        if t % 5 == 0:
            for d in (2, 3):
                np.testing.assert_array_equal(e.obs_pytree(d, 30), fx["py_d%d_p30" % d][t], err_msg=f"t={t} depth {d}")
