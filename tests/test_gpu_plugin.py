"""GPU: the observation builders as PLUG-INS of a caller-owned env (flatland_marl_amd/plugin.py): a duck-typed env object
-- only the attributes flatland_cutils reads (loader.cpp:8-120, 207-219, 329-333), nothing of this library -- replays the
per-step agent states of a golden episode of the reference; the builder is called after every step, the way
RailEnv._get_observations does (rail_env.py:660-666), and has to reproduce the reference's observations at every snapshot,
including the DeadlockChecker's sticky flags that only the builder itself carries from call to call."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

CUTILS = (("o_attr", 0, None), ("o_forest", 1, 0), ("o_adjacency", 1, 1), ("o_node_order", 1, 2), ("o_edge_order", 1, 3))


def _same(got, exp, msg):
    got = np.asarray(got)
    if got.shape != exp.shape or not np.array_equal(got, exp):
        bad = np.argwhere(got != exp) if got.shape == exp.shape else [["shape", got.shape, exp.shape]]
        raise AssertionError(f"{msg}: {len(bad)} mismatches, first {bad[0]}")


@pytest.mark.parametrize("name,string_states", [("cfg3_spfollow_malf100", False), ("cfg1_sparse", True), ("cfg2_fwd", False),
                                                 ("cfg0_tall_spfollow", True)])
def test_cutils_plugin_on_a_duck_typed_env_replaying_a_reference_episode(name, string_states):
    from flatland_marl_amd.plugin import TreeObsForRailEnv
    fx = util.load(name)
    env = util.DuckEnv(fx, string_states)
    b = TreeObsForRailEnv(31, 500)
    b.set_env(env)
    b.reset()
    A = env.get_num_agents()
    handles = list(range(A))
    obs_steps = {int(t): k for k, t in enumerate(fx["obs_steps"])}
    steps = len(fx["s_row"])
    n, dead_seen = 0, 0
    for T in range(0, steps + 1):
        env.goto(T)
        out = b.get_many(handles)
        if T not in obs_steps:
            continue
        k = obs_steps[T]
        for key, i, j in CUTILS:
            got = out[i] if j is None else out[i][j]
            _same(np.array(got, dtype=fx[key].dtype), fx[key][k], f"{name} T={T} {key}")
        cfg, props, valid = b.get_properties()
        assert cfg == dict(curr_step=T, n_agents=A, max_timesteps=int(fx["T"]), height=env.height, width=env.width)
        _same(np.array(valid, dtype=np.uint8), fx["o_valid"][k], f"{name} T={T} valid_actions")
        _same(np.array(props["dist_target"]), fx["o_p_dist_target"][k], f"{name} T={T} dist_target")
        _same(np.array(props["deadlocked"]), fx["o_p_deadlocked"][k], f"{name} T={T} deadlocked")
        _same(np.array(props["ready_not_depart"]), fx["o_p_ready"][k], f"{name} T={T} ready")
        assert props["earliest_departure"] == [float(v) for v in fx["earliest"]]
        assert props["speed"] == [float(np.float32(v)) for v in fx["speed"]]
        dead_seen = max(dead_seen, int(fx["o_p_deadlocked"][k].sum()))
        n += 1
    assert n >= 3 and dead_seen > 0
    # one agent's observation (ObservationBuilder.get, env_observation_builder.py:56-73) = its row of get_many
    attr, (nodes, adj, no, eo) = b.get(1)
    assert attr == out[0][1] and nodes == out[1][0][1] and adj == out[1][1][1] and no == out[1][2][1] and eo == out[1][3][1]


def test_reset_clears_the_sticky_flags_and_follows_a_new_map():
    """reset() = a new DeadlockChecker (loader.cpp:186-199) and a new static read: the same builder object serves a second env of
    another shape (RailEnv.reset(regenerate_rail=True) calls set_env + reset on the same builder, rail_env.py:305, 346)."""
    from flatland_marl_amd.plugin import TreeObsForRailEnv
    b = TreeObsForRailEnv(31, 500)
    for name in ("cfg1_sparse", "cfg2_fwd", "cfg1_sparse"):
        fx = util.load(name)
        env = util.DuckEnv(fx)
        b.set_env(env)
        b.reset()
        handles = list(range(env.get_num_agents()))
        obs_steps = [int(t) for t in fx["obs_steps"]]
        last = obs_steps[-1]
        for T in range(0, last + 1):
            env.goto(T)
            out = b.get_many(handles)
        k = len(obs_steps) - 1
        _same(np.array(out[0], dtype=np.float32), fx["o_attr"][k], f"{name} attr")
        _same(np.array(out[1][0], dtype=np.float32), fx["o_forest"][k], f"{name} forest")
        assert fx["o_p_deadlocked"][k].sum() > 0
        _same(np.array(b.get_properties()[1]["deadlocked"]), fx["o_p_deadlocked"][k], f"{name} deadlocked")
        # back to the start of the episode WITHOUT reset(): the flags stay (sticky); after reset() they are gone
        env.goto(obs_steps[1])
        b.get_many(handles)
        assert np.array(b.get_properties()[1]["deadlocked"]).sum() >= fx["o_p_deadlocked"][k].sum()
        b.reset()
        b.get_many(handles)
        _same(np.array(b.get_properties()[1]["deadlocked"]), fx["o_p_deadlocked"][1], f"{name} deadlocked after reset")


@pytest.mark.parametrize("name", ["cfg3_uniform", "cfg2_uniform", "cfg1_uniform"])
def test_upstream_plugin_on_a_duck_typed_env(name):
    from flatland_marl_amd.plugin import TreeObsUpstream
    from flatland_marl_amd.rail_env import dense_from_nodes, Node
    fx = util.load(name)
    env = util.DuckEnv(fx)
    A = env.get_num_agents()
    py_steps = [int(t) for t in fx["py_steps"]]
    for pk in [k for k in fx.files if k.startswith("py_d")]:
        depth, pdepth = int(pk.split("_")[1][1:]), int(pk.split("_")[2][1:])
        b = TreeObsUpstream(depth, util._NS(max_depth=pdepth))
        b.set_env(env)
        env.goto(0)
        b.reset()
        for k, T in enumerate(py_steps):
            env.goto(T)
            got = b.get_many(list(range(A)))
            assert sorted(got) == list(range(A)) and isinstance(got[0], Node)
            dense = np.stack([dense_from_nodes(got[h], depth) for h in range(A)])
            _same(dense, fx[pk][k], f"{name} T={T} {pk}")
        assert b.get_many(None) == {}             # observations.py:66-67: no handles, no observations


def test_verify_distance_map_accepts_the_reference_map_and_refuses_another():
    from flatland_marl_amd.plugin import TreeObsForRailEnv
    fx = util.load("cfg1_uniform")
    env = util.DuckEnv(fx)
    dm = fx["dm_u16"][fx["target_slot"]].astype(np.float64)
    dm[fx["dm_u16"][fx["target_slot"]] == 0xFFFF] = np.inf
    env.distance_map = util._NS(get=lambda: dm)
    b = TreeObsForRailEnv(31, 500, verify_distance_map=True)
    b.set_env(env)
    b.reset()
    dm2 = dm.copy()
    dm2[np.isfinite(dm2)] += 1
    env2 = util.DuckEnv(fx)
    env2.agents[0].earliest_departure += 1                 # another timetable, so that the static side is read again
    env2.distance_map = util._NS(get=lambda: dm2)
    b.set_env(env2)
    with pytest.raises(ValueError, match="distance map"):
        b.reset()
