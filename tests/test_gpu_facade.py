"""GPU: the dict-shaped facade (RailEnv / TreeObsForRailEnv / LocalTestEnvWrapper counterparts) against the
golden episodes of the reference: same rewards_dict / dones / info / observation lists / final metric."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _action_dict(row):
    return {i: int(a) for i, a in enumerate(row) if a != 255}


@pytest.mark.parametrize("name", ["cfg1_sparse", "cfg1_spfollow"])
def test_facade_episode_matches_reference(name):
    from flatland_marl_amd.rail_env import RailEnv, TreeObsForRailEnv, LocalTestEnvWrapper, TrainState
    fx = util.load(name)
    env = RailEnv(util.static_of(fx), obs_builder_object=TreeObsForRailEnv(31, 500))
    wrapper = LocalTestEnvWrapper(env)
    obs_steps = {int(t): k for k, t in enumerate(fx["obs_steps"])}
    obs, info = env.reset()
    assert set(info) == {"action_required", "malfunction", "speed", "state"}
    np.testing.assert_array_equal(np.array(obs[0], dtype=np.float32), fx["o_attr"][0])
    A = env.get_num_agents()
    for t, row in enumerate(util.actions_of(fx)):
        # eval_env.parse_actions keeps only the actions of agents with action_required
        req = {i: (fx["s_state"][t - 1][i] == 1 or (fx["s_state"][t - 1][i] in (3, 4, 5) and fx["s_scount"][t - 1][i] == 0))
               if t > 0 else False for i in range(A)}
        ad = _action_dict(row)
        assert wrapper.parse_actions(dict(ad)) == {i: a for i, a in ad.items() if req[i]}
        obs, rew, dones, info = env.step(ad)   # the golden episode was produced with the unfiltered dict
        assert [rew[i] for i in range(A)] == fx["s_reward"][t].tolist()
        assert [dones[i] for i in range(A)] == fx["s_done"][t].astype(bool).tolist()
        assert dones["__all__"] == bool(fx["done_all"][t])
        assert [int(info["state"][i]) for i in range(A)] == fx["s_state"][t].tolist()
        assert [info["malfunction"][i] for i in range(A)] == fx["s_malf"][t].tolist()
        assert [a.position for a in env.agents] == [None if r < 0 else (int(r), int(c))
                                                    for r, c in zip(fx["s_row"][t], fx["s_col"][t])]
        if (t + 1) in obs_steps:
            k = obs_steps[t + 1]
            attr, (nodes, adj, node_order, edge_order) = obs
            np.testing.assert_array_equal(np.array(attr, dtype=np.float32), fx["o_attr"][k])
            np.testing.assert_array_equal(np.array(nodes, dtype=np.float32), fx["o_forest"][k])
            np.testing.assert_array_equal(np.array(adj), fx["o_adjacency"][k])
            np.testing.assert_array_equal(np.array(node_order), fx["o_node_order"][k])
            np.testing.assert_array_equal(np.array(edge_order), fx["o_edge_order"][k])
            wrapper.update_obs_properties()
            np.testing.assert_array_equal(np.array(wrapper.get_valid_actions(), dtype=np.uint8), fx["o_valid"][k])
            np.testing.assert_array_equal(np.array(wrapper.obs_properties["deadlocked"]), fx["o_p_deadlocked"][k])
            feats = wrapper.parse_features(obs, wrapper.obs_properties)
            assert feats["agent_attr"].shape == (A, 83) and feats["forest"].shape == (A, 31, 12)
            assert not np.isinf(feats["forest"]).any()
    assert env.dones["__all__"]
    np.testing.assert_allclose(np.array(wrapper.final_metric()), fx["final_metric"], rtol=0, atol=0)
    with pytest.raises(Exception, match="Episode is done"):
        env.step({})
    dm = env.distance_map.get()
    assert dm.shape == (A, env.height, env.width, 4)
    exp = fx["dm_u16"][fx["target_slot"]].astype(np.float64)
    exp[fx["dm_u16"][fx["target_slot"]] == 0xFFFF] = np.inf
    np.testing.assert_array_equal(dm, exp)


def test_upstream_tree_builder_and_positions_map():
    from flatland_marl_amd.rail_env import RailEnv, TreeObsUpstream
    fx = util.load("cfg1_uniform")
    env = RailEnv(util.static_of(fx), obs_builder_object=TreeObsUpstream(2, 30))
    obs, _ = env.reset()
    py_steps = {int(t): k for k, t in enumerate(fx["py_steps"])}
    np.testing.assert_array_equal(np.stack([obs[i] for i in range(env.get_num_agents())]), fx["py_d2_p30"][py_steps[0]])
    for t, row in enumerate(fx["actions"][:60]):
        obs, _, _, _ = env.step(_action_dict(row))
        if (t + 1) in py_steps:
            np.testing.assert_array_equal(np.stack([obs[i] for i in range(env.get_num_agents())]),
                                          fx["py_d2_p30"][py_steps[t + 1]])
    pm = env._batch.positions_map(0)
    exp = np.full((env.height, env.width), -1, dtype=np.int32)
    for i, (r, c) in enumerate(zip(fx["s_row"][59], fx["s_col"][59])):
        if r >= 0:
            exp[r, c] = i
    np.testing.assert_array_equal(pm, exp)
