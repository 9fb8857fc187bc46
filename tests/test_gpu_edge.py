"""GPU parity on edge cases: non-default observation sizes, no predictor, single-agent envs, masked and
non-fresh resets, stepping a finished env inside a batch."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _env(envs, **kw):
    from flatland_marl_amd.hip_backend import BatchedRailEnv
    return BatchedRailEnv(envs, **kw)


def _same(got, exp, msg):
    got = np.asarray(got)
    if not np.array_equal(got, exp):
        bad = np.argwhere(got != exp)
        raise AssertionError(f"{msg}: {len(bad)} mismatches, first {bad[0].tolist()}: {got[tuple(bad[0])]} vs {exp[tuple(bad[0])]}")


@pytest.mark.parametrize("max_nodes,pred_depth", [(16, 100), (4, 1), (32, 500), (25, 37)])
def test_cutils_other_sizes_match_oracle(max_nodes, pred_depth):
    from oracle import orc
    from flatland_marl_amd import synth
    fx = util.load("cfg2_spfollow")
    st = util.static_of(fx)
    env = _env([st, st], max_nodes=max_nodes, pred_depth=pred_depth)
    o = orc.OracleEnv(st)
    A = env.A
    for t in range(160):
        env.step_synth(3, 0, 1, auto_reset=True)
        o.step(synth.forward_biased_actions(3, 0, t, A))
        got = {k: v.cpu().numpy() for k, v in env.obs_cutils().items()}
        exp = o.obs_cutils(max_nodes, pred_depth)
        for g, e in (("agent_attr", "attr"), ("forest", "forest"), ("adjacency", "adjacency"), ("node_order", "node_order"),
                     ("edge_order", "edge_order"), ("valid_actions", "valid")):
            _same(got[g][0], exp[e], f"t={t} {g} N={max_nodes} P={pred_depth}")
        if t % 16 == 5:      # the policy's int64 tensors from the launch itself (fl_obs_cutils_policy) at this tree size == fl_policy_pack of the int32 ones
            import torch
            ref = [x.clone() for x in env.policy_inputs(env._obs)[2:]]
            for x, y in zip(env.obs_policy()[2:], ref):
                assert x.shape == y.shape and torch.equal(x, y), (t, max_nodes)
    env.check()


@pytest.mark.parametrize("depth,pred", [(1, 30), (2, -1), (3, -1), (3, 10), (2, 500), (2, 0), (2, 1), (4, 30), (4, -1), (4, 500)])
def test_upstream_tree_variants_match_oracle(depth, pred):
    from oracle import orc
    from flatland_marl_amd import synth
    fx = util.load("cfg0_tall_spfollow")
    st = util.static_of(fx)
    env = _env([st])
    o = orc.OracleEnv(st)
    for t in range(120):
        env.step_synth(9, 0, 1, auto_reset=True)
        o.step(synth.forward_biased_actions(9, 0, t, env.A))
        if t % 3 == 0:
            _same(env.obs_tree(depth, pred).cpu().numpy()[0], o.obs_pytree(depth, pred), f"t={t} depth={depth} pred={pred}")
    env.check()


def test_single_agent_env():
    from oracle import orc
    from flatland_marl_amd import synth
    fx = util.load("cfg1_spfollow")
    st = util.static_of(fx)
    for k in ("init_pos", "init_dir", "target", "speed", "earliest", "latest"):
        st[k] = st[k][2:3]
    env = _env([st])
    o = orc.OracleEnv(st)
    assert env.A == 1
    for t in range(int(st["T"])):
        rew, done, done_all = env.step_synth(1, 0, 1, auto_reset=False)
        r_o, d_o, da_o = o.step(synth.forward_biased_actions(1, 0, t, 1))
        _same(env.state()[0][0], o.state(), f"t={t}")
        _same(rew.cpu().numpy()[0], r_o, f"t={t} reward")
        got = {k: v.cpu().numpy() for k, v in env.obs_cutils().items()}
        exp = o.obs_cutils(31, 500)
        _same(got["forest"][0], exp["forest"], f"t={t} forest")
        _same(got["agent_attr"][0], exp["attr"], f"t={t} attr")
        if da_o:
            break
    env.check()


def test_largest_supported_agent_count_matches_oracle():
    """A = 1024 (the cap of one lane per agent): more than 64 KiB of dynamic LDS in the step kernel.  A synthetic
    env: the 30x30 map of cfg2 with 1024 agents cycling through its agents' lines and staggered departures."""
    from oracle import orc
    from flatland_marl_amd import synth
    fx = util.load("cfg2_spfollow")
    st = util.static_of(fx)
    A0, A = len(st["init_dir"]), 1024
    idx = np.arange(A) % A0
    big = dict(st)
    for k in ("init_pos", "init_dir", "target", "speed", "latest"):
        big[k] = np.ascontiguousarray(np.asarray(st[k])[idx])
    big["earliest"] = (np.arange(A) // 4).astype(np.int32)
    big["malf_rate"] = 1 / 300.0
    env = _env([big], pred_depth=60)
    o = orc.OracleEnv(big)
    for t in range(150):
        rew, done, done_all = env.step_synth(77, 0, 1, auto_reset=True)
        r_o, d_o, da = o.step(synth.forward_biased_actions(77, 0, t, A))
        _same(env.state()[0][0], o.state(), f"t={t} state")
        _same(rew.cpu().numpy()[0], r_o, f"t={t} rewards")
        if t % 25 == 0:
            got = {k: v.cpu().numpy() for k, v in env.obs_cutils().items()}
            exp = o.obs_cutils(31, 60)
            for g, e in (("agent_attr", "attr"), ("forest", "forest"), ("adjacency", "adjacency"), ("valid_actions", "valid"), ("props", "props")):
                _same(got[g][0], exp[e], f"t={t} {g}")
            _same(env.obs_tree(2, 30).cpu().numpy()[0], o.obs_pytree(2, 30), f"t={t} tree")
    env.check()
    key, pos = env.rng_state()
    k_o, p_o = o.get_rng()
    assert pos[0] == p_o and np.array_equal(key[0], k_o)


@pytest.mark.parametrize("A,max_nodes,depth", [(1, 31, 2), (2, 31, 1), (31, 31, 2), (32, 31, 2), (31, 13, 2), (31, 12, 2), (24, 20, 1)])
def test_fused_observations_at_the_limits_of_the_one_pass_mode(A, max_nodes, depth):
    """The fused launch builds the trees of both builders with ONE pass B on envs of at most 31 agents (upstream depth <= 2,
    max_nodes >= 13) and in two stages beyond (fl_obs.hip: trees_merged): the agent counts either side of the limit, the
    smallest counts, the smallest node budget, dense traffic on the 30x30 map of cfg2 (agents cycling through its lines,
    staggered departures, malfunctions)."""
    from oracle import orc
    from flatland_marl_amd import synth
    fx = util.load("cfg2_spfollow")
    st = util.static_of(fx)
    A0 = len(st["init_dir"])
    idx = np.arange(A) % A0
    big = dict(st)
    for k in ("init_pos", "init_dir", "target", "speed", "latest"):
        big[k] = np.ascontiguousarray(np.asarray(st[k])[idx])
    big["earliest"] = (np.arange(A) // 3).astype(np.int32)
    big["malf_rate"] = 1 / 200.0
    env = _env([big, big], max_nodes=max_nodes, pred_depth=120)
    o = orc.OracleEnv(big)
    keys = (("agent_attr", "attr"), ("forest", "forest"), ("adjacency", "adjacency"), ("node_order", "node_order"),
            ("edge_order", "edge_order"), ("valid_actions", "valid"), ("props", "props"))
    for t in range(140):
        env.step_synth(91, 0, 1, auto_reset=True)
        o.step(synth.forward_biased_actions(91, 0, t, A))
        _same(env.state()[0][0], o.state(), f"t={t} state")
        if t % 7 == 0:
            got, tree = env.obs_both(depth, 30)
            exp = o.obs_cutils(max_nodes, 120)
            for g, e in keys:
                _same(got[g].cpu().numpy()[0], exp[e], f"t={t} {g}")
            _same(tree.cpu().numpy()[0], o.obs_pytree(depth, 30), f"t={t} tree")
    env.check()


@pytest.mark.parametrize("name,A", [("cfg4_fwd_head", 24), ("cfg5_fwd_head", 20), ("cfg3_uniform", 31), ("cfg0_tall_spfollow", 3)])
def test_fused_observations_of_few_agents_on_other_maps(name, A):
    """The one-pass mode of the fused launch (fl_obs.hip: trees_merged) on other maps than cfg2's: the first agents of the
    60x60 / 150x150 / 35x30 envs (long paths, next-hop tables that do not fit the LDS, hundreds of keys) and a map that is
    taller than wide (its prediction keys collide, so it takes the two stages)."""
    from oracle import orc
    from flatland_marl_amd import synth
    st = util.static_of(util.load(name))
    sub = dict(st)
    A = min(A, len(st["init_dir"]))
    for k in ("init_pos", "init_dir", "target", "speed", "latest"):
        sub[k] = np.ascontiguousarray(np.asarray(st[k])[:A])
    sub["earliest"] = (np.arange(A) // 3).astype(np.int32)  # everybody on the map soon
    sub["malf_rate"] = 1 / 150.0
    env = _env([sub], pred_depth=300)
    o = orc.OracleEnv(sub)
    keys = (("agent_attr", "attr"), ("forest", "forest"), ("adjacency", "adjacency"), ("node_order", "node_order"),
            ("edge_order", "edge_order"), ("valid_actions", "valid"), ("props", "props"))
    for t in range(120):
        env.step_synth(5, 0, 1, auto_reset=True)
        o.step(synth.forward_biased_actions(5, 0, t, A))
        _same(env.state()[0][0], o.state(), f"t={t} state")
        if t % 10 == 0:
            got, tree = env.obs_both(2, 30)
            exp = o.obs_cutils(31, 300)
            for g, e in keys:
                _same(got[g].cpu().numpy()[0], exp[e], f"t={t} {g}")
            _same(tree.cpu().numpy()[0], o.obs_pytree(2, 30), f"t={t} tree")
    assert (env.state()[0][0][:, 0] >= 0).sum() >= min(A, 3) // 2      # agents are on the map at the end
    env.check()


def test_masked_and_non_fresh_reset():
    """fl_reset(mask, fresh=0) follows EnvAgent.reset() literally: arrival_time survives (agent_utils.py:90-105)."""
    import torch
    fx = util.load("cfg1_spfollow")
    st = util.static_of(fx)
    env = _env([st, st, st])
    acts = util.actions_of(fx)
    for a in acts:
        env.step(torch.from_numpy(np.stack([a, a, a])).cuda())
    s_end, el = env.state()
    assert (el == len(acts)).all() and (s_end[:, :, 3] == 6).all()      # every agent DONE in this episode
    env.reset(mask=[1, 0, 1], fresh=False)
    env.reset(mask=[0, 0, 1], fresh=True)
    s, el = env.state()
    assert el.tolist() == [0, len(acts), 0]
    np.testing.assert_array_equal(s[1], s_end[1])                        # untouched env
    assert (s[0][:, 3] == 0).all() and (s[0][:, 0] == -1).all()          # WAITING, off map
    np.testing.assert_array_equal(s[0][:, 8], s_end[0][:, 8])            # arrival_time kept by the literal reset
    assert (s[2][:, 8] == -1).all()                                      # fresh reset clears it
    # env 2 replays the episode after its fresh reset (the RNG kept running, malfunction draws may differ: compare
    # with an oracle that continues from the same RNG state)
    from oracle import orc
    key, pos = env.rng_state()
    o = orc.OracleEnv(st)
    o.set_rng(key[2], pos[2])
    from flatland_marl_amd.hip_backend import EpisodeDoneError
    for t, a in enumerate(acts[:50]):
        env.step(torch.from_numpy(np.stack([a, a, a])).cuda())
        o.step(a)
        _same(env.state()[0][2], o.state(), f"replay t={t}")
    with pytest.raises(EpisodeDoneError):      # env 1 was finished and not reset: its step raised inside the batch
        env.check()


def test_distance_map_rebuild_is_idempotent():
    fx = util.load("cfg4_fwd_head")
    env = _env([util.static_of(fx)])
    dm0, slot0 = env.distance_map(0)
    env.rebuild_distance_maps()
    env.rebuild_distance_maps()
    dm1, slot1 = env.distance_map(0)
    np.testing.assert_array_equal(dm0, dm1)
    np.testing.assert_array_equal(dm1, fx["dm_u16"])
    env.step_synth(1, 0, 1, auto_reset=True)
    env.obs_cutils()
    env.check()


def test_policy_inputs_match_reference_modify_adjacency():
    """golden: Network.modify_adjacency of the real reference on four cutils outputs stacked as a batch."""
    import torch
    from flatland_marl_amd.hip_backend import policy_pack
    fx = util.load("cfg2_uniform")
    g = util.load("policy_inputs")
    idx = g["obs_index"]
    adj = torch.from_numpy(np.stack([fx["o_adjacency"][k] for k in idx]).astype(np.int32)).cuda()
    no = torch.from_numpy(np.stack([fx["o_node_order"][k] for k in idx]).astype(np.int32)).cuda()
    eo = torch.from_numpy(np.stack([fx["o_edge_order"][k] for k in idx]).astype(np.int32)).cuda()
    B, A, E = adj.shape[:3]
    out = (torch.empty((B, A, E, 3), dtype=torch.int64, device="cuda"), torch.empty((B, A, E + 1), dtype=torch.int64, device="cuda"),
           torch.empty((B, A, E), dtype=torch.int64, device="cuda"))
    policy_pack(adj, no, eo, *out)
    np.testing.assert_array_equal(out[0].cpu().numpy(), g["adjacency_mod"])
    np.testing.assert_array_equal(out[1].cpu().numpy(), no.cpu().numpy().astype(np.int64))
    np.testing.assert_array_equal(out[2].cpu().numpy(), eo.cpu().numpy().astype(np.int64))
    # and through the env: same transformation of its own observation
    env = _env([util.static_of(fx)] * 3)
    attr, forest, adj64, no64, eo64 = env.policy_inputs()
    raw = env.obs_cutils()["adjacency"].cpu().numpy().astype(np.int64)
    exp = raw.copy()
    exp[exp == -2] = -3 * env.A * 31
    ids = (np.arange(3)[:, None] * env.A + np.arange(env.A)[None, :])[:, :, None] * 31
    exp[..., 0] += ids
    exp[..., 1] += ids
    exp[exp < 0] = -2
    np.testing.assert_array_equal(adj64.cpu().numpy(), exp)
    assert attr.dtype == torch.float32 and forest.dtype == torch.float32 and no64.dtype == torch.int64


def test_action_required_filter_info_and_scores_match_reference():
    """golden episode captured through eval_env.parse_actions: RAW policy actions + the fused filter must reproduce it;
    fl_info gives get_info_dict's tensors and the evaluator's normalized reward of the finished episode."""
    import torch
    fx = util.load("cfg2_filtered")
    env = _env([util.static_of(fx)])
    raw = np.array(fx["actions"], dtype=np.uint8)
    req = np.asarray(fx["action_required"])
    assert (raw[req == 0] != 255).any()                         # the raw stream really contains actions to drop
    info = env.info()
    np.testing.assert_array_equal(info["action_required"].cpu().numpy()[0], req[0])
    for t in range(len(raw)):
        rew, done, done_all = env.step(torch.from_numpy(raw[t][None, :].copy()).cuda(), filter_required=True)
        st, _ = env.state()
        np.testing.assert_array_equal(st[0], util.golden_state(fx, t), err_msg=f"step {t}")
        np.testing.assert_array_equal(rew.cpu().numpy()[0], fx["s_reward"][t])
        info = env.info()
        np.testing.assert_array_equal(info["state"].cpu().numpy()[0], fx["s_state"][t])
        np.testing.assert_array_equal(info["malfunction"].cpu().numpy()[0], fx["s_malf"][t])
        if t + 1 < len(raw):
            np.testing.assert_array_equal(info["action_required"].cpu().numpy()[0], req[t + 1], err_msg=f"required after {t}")
    env.check()
    assert bool(done_all.cpu().numpy()[0])
    scores = env.info()["scores"].cpu().numpy()[0]
    assert scores[0] == fx["final_metric"][2]                   # 1 + sum(rewards) / T / A (eval_env.py:92, service.py:875-879)
    # both evaluator scores as captured from the reference env objects (service.py:875-879, 900-913)
    np.testing.assert_array_equal(scores, fx["evaluator_scores"])


@pytest.mark.parametrize("name,B,depth,pred", [("cfg2_spfollow", 6, 2, 30), ("cfg0_tall_spfollow", 3, 3, 30),
                                               ("cfg3_uniform", 3, 2, 500), ("cfg5_fwd_head", 1, 2, 30),
                                               ("cfg2_spfollow", 2, 2, 500), ("cfg1_spfollow", 2, 3, 200),
                                               ("cfg3_uniform", 3, 4, 30)])      # depth 4: the fused entry points run the two builders one after the other
def test_fused_observation_launch_equals_separate_launches(name, B, depth, pred):
    fx = util.load(name)
    st = util.static_of(fx)
    env = _env([st] * B)
    for t in range(90):
        env.step_synth(4, 0, 1, auto_reset=True)
        if t % 6 == 0:
            sep = {k: v.clone() for k, v in env.obs_cutils().items()}
            sep_tree = env.obs_tree(depth, pred).clone()
            # the deadlock flags are sticky: a second cutils build in the same step must not change anything either
            fo, ft = env.obs_both(depth, pred)
            for k in sep:
                _same(fo[k].cpu().numpy(), sep[k].cpu().numpy(), f"{name} t={t} {k}")
            _same(ft.cpu().numpy(), sep_tree.cpu().numpy(), f"{name} t={t} tree")
        else:
            env.obs_both(depth, pred)
    env.check()


@pytest.mark.parametrize("name,B,depth,kind", [("cfg2_spfollow", 5, 2, 0), ("cfg1_malf50", 3, 0, 1), ("cfg0_tall_uniform", 2, 3, 0),
                                               ("cfg3_spfollow_malf100", 2, 2, 1), ("cfg5_fwd_head", 1, 2, 1)])
def test_fused_step_and_observation_launch_equals_separate_launches(name, B, depth, kind):
    """fl_step_obs (RailEnv.step() returning the observations, one launch) vs fl_step_synth + fl_obs_*: state, rewards,
    dones, RNG and every observation tensor identical on every step, through auto-resets."""
    fx = util.load(name)
    st = util.static_of(fx)
    e1, e2 = _env([st] * B), _env([st] * B)
    n_steps = min(int(st["T"]) + 25, 260)
    for t in range(n_steps):
        r1, d1, a1 = (x.clone() for x in e1.step_synth(5, 3, kind, auto_reset=True))
        o1 = {k: v.clone() for k, v in e1.obs_cutils().items()}
        t1 = e1.obs_tree(depth, 30).clone() if depth > 0 else None
        r2, d2, a2, o2, t2 = e2.step_obs(None, 5, 3, kind, auto_reset=True, tree_depth=depth, tree_pred=30)
        _same(r2.cpu().numpy(), r1.cpu().numpy(), f"{name} t={t} rewards")
        _same(d2.cpu().numpy(), d1.cpu().numpy(), f"{name} t={t} dones")
        _same(a2.cpu().numpy(), a1.cpu().numpy(), f"{name} t={t} done_all")
        if t % 7 == 0 or t > n_steps - 30:
            _same(e2.state()[0], e1.state()[0], f"{name} t={t} state")
            for k in o1:
                _same(o2[k].cpu().numpy(), o1[k].cpu().numpy(), f"{name} t={t} {k}")
            if depth > 0:
                _same(t2.cpu().numpy(), t1.cpu().numpy(), f"{name} t={t} tree")
    k1, p1 = e1.rng_state()
    k2, p2 = e2.rng_state()
    _same(k2, k1, "mt key"); _same(p2, p1, "mt pos")
    e1.check(); e2.check()


def test_fused_step_with_explicit_actions_and_filter():
    import torch
    fx = util.load("cfg2_filtered")
    st = util.static_of(fx)
    raw = fx["actions"]                       # RAW action stream of the fixture (the filter drops part of it)
    e1, e2 = _env([st, st]), _env([st, st])
    for t in range(min(len(raw), 150)):
        a = torch.from_numpy(np.stack([raw[t], raw[t]]).astype(np.uint8)).cuda()
        r1, d1, _ = (x.clone() for x in e1.step(a, filter_required=True))
        o1 = {k: v.clone() for k, v in e1.obs_cutils().items()}
        r2, d2, _, o2, _ = e2.step_obs(a, filter_required=True)
        _same(r2.cpu().numpy(), r1.cpu().numpy(), f"t={t} rewards")
        _same(d2.cpu().numpy(), d1.cpu().numpy(), f"t={t} dones")
        for k in o1:
            _same(o2[k].cpu().numpy(), o1[k].cpu().numpy(), f"t={t} {k}")
        if bool(e1.done_all.cpu().numpy().all()):
            break
    e1.check(); e2.check()


def test_fused_step_after_episode_end_raises_like_the_reference():
    from flatland_marl_amd.hip_backend import EpisodeDoneError
    fx = util.load("cfg1_spfollow")
    st = util.static_of(fx)
    env = _env([st])
    for a in util.actions_of(fx):
        env.step_obs(np.asarray(a, dtype=np.uint8)[None])
    env.check()
    env.step_obs(np.zeros((1, env.A), dtype=np.uint8))
    with pytest.raises(EpisodeDoneError):
        env.check()


def test_builder_sizes_beyond_the_limits_are_refused_with_a_message():
    """max_depth 5, max_depth 4 on a grid with a three-way cell (DFS-slot node tables stop at depth 3), 65 cutils nodes, a speed below
    1/64: FL_ERR_ARG with a message that names the limit, nothing launched"""
    from flatland_marl_amd.hip_backend import FlatlandHipError
    fx = util.load("cfg1_uniform")
    st = util.static_of(fx)
    env = _env([st])
    env.step_synth(1, 0, 0)
    assert env.obs_tree(4, 30).shape == (1, env.A, 341, 12)
    with pytest.raises(FlatlandHipError, match=r"max_depth must be in \[1,4\]"):
        env.obs_tree(5, 30)
    env.max_nodes = 65
    env._obs = None
    with pytest.raises(FlatlandHipError, match="max_nodes"):
        env.obs_cutils()
    slow = dict(st)
    slow["speed"] = np.array(st["speed"], dtype=np.float64)
    slow["speed"][0] = 1.0 / 65.0
    with pytest.raises(FlatlandHipError, match="max_count 64"):
        _env([slow])
    slow["speed"][0] = 1.0 / 64.0            # the slowest train this build takes
    _env([slow]).step_synth(1, 0, 0)
    tw = util.load("threeway_cfg2")
    e3 = _env([util.static_of(tw)])
    e3.step_synth(1, 0, 0)
    with pytest.raises(FlatlandHipError, match="more than two transitions"):
        e3.obs_tree(4, 30)
    assert e3.obs_tree(3, 30).shape[2] == 85
    # the fused entry points raise the same error BEFORE anything runs: the envs have not advanced, no sticky deadlock bit was set
    st0, el0 = e3.state()
    aux0 = e3.state_aux()
    with pytest.raises(FlatlandHipError, match="more than two transitions"):
        e3.step_obs(seed=1, tree_depth=4, tree_pred=30)
    with pytest.raises(FlatlandHipError, match="more than two transitions"):
        e3.obs_both(4, 30)
    st1, el1 = e3.state()
    np.testing.assert_array_equal(st0, st1)
    np.testing.assert_array_equal(el0, el1)
    np.testing.assert_array_equal(aux0, e3.state_aux())
    e3.step_obs(seed=1, tree_depth=3, tree_pred=30)          # a retry at depth 3 is the NEXT step, not the one after
    assert e3.state()[1][0] == el0[0] + 1
