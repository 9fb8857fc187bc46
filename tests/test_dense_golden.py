"""Dense-traffic goldens of the large maps (captured from the real reference by oracle/refharness/capture_golden.py
run_dense): cfg4 = 60x60 / 80 agents up to 67 agents on the map, cfg5 = 150x150 / 400 agents up to 250 on the map, dozens of
deadlocked agents, shortest-path-following actions; flatland_cutils tensors and depth-3 upstream trees (predictor 30) at
the snapshots.  CPU leg: the oracle replays the episode (pins the oracle in this regime).  GPU leg (-m gpu): the HIP path
replays it through the C-ABI, builds the observations separately AND fused (obs_both(3, 30), BASELINE's definition of
configs[2] / configs[4]), re-derives every snapshot from an injected state, and rebuilds the distance maps mid-episode."""
import numpy as np
import pytest

from tests import util

DENSE = ("dense_cfg4_spfollow", "dense_cfg5_spfollow")
HEAD = {"dense_cfg4_spfollow": "cfg4_fwd_head", "dense_cfg5_spfollow": "cfg5_fwd_head"}   # same CSV row: same static env
CPU_STEPS = {"dense_cfg4_spfollow": 800, "dense_cfg5_spfollow": 450}     # the CPU suite stops here (oracle: ~75 ms/step at cfg5)
CUTILS = (("attr", "agent_attr", "o_attr"), ("forest", "forest", "o_forest"), ("adjacency", "adjacency", "o_adjacency"),
          ("node_order", "node_order", "o_node_order"), ("edge_order", "edge_order", "o_edge_order"), ("valid", "valid_actions", "o_valid"))


def _same(got, exp, msg):
    got = np.asarray(got)
    if not np.array_equal(got, exp):
        bad = np.argwhere(got != exp)
        raise AssertionError(f"{msg}: {len(bad)} mismatches, first {bad[0].tolist()}: {got[tuple(bad[0])]} vs {exp[tuple(bad[0])]}")


@pytest.mark.parametrize("name", DENSE)
def test_dense_fixture_is_dense_and_shares_the_head_fixture_env(name):
    fx, head = util.load(name), util.load(HEAD[name])
    for k in ("grid", "init_pos", "init_dir", "target", "speed", "earliest", "latest", "T", "mt_key", "mt_pos", "target_slot"):
        np.testing.assert_array_equal(fx[k], head[k])
    A = len(fx["init_dir"])
    on = fx["on_map"][fx["obs_steps"] - 1]
    assert on.max() >= (100 if A == 400 else 40) and fx["o_p_deadlocked"].sum(axis=1).max() >= 10


@pytest.mark.parametrize("name", DENSE)
def test_oracle_replays_dense_reference_episode(name):
    from oracle import orc
    fx = util.load(name)
    e = orc.OracleEnv(fx)
    ss = {int(t): k for k, t in enumerate(fx["state_steps"])}
    os_ = {int(t): k for k, t in enumerate(fx["obs_steps"])}
    n = 0
    for t in range(min(CPU_STEPS[name], len(fx["actions"]))):
        rew, done, _ = e.step(fx["actions"][t])
        o = e.obs_cutils(31, 500)          # every step: the deadlock flags are sticky
        T = t + 1
        if T in ss:
            _same(e.state(), fx["states"][ss[T]], f"{name} T={T} state")
            _same(rew, fx["rewards"][ss[T]], f"{name} T={T} rewards")
            _same(done, fx["dones"][ss[T]], f"{name} T={T} dones")
        if T in os_:
            k = os_[T]
            for ok, _, fk in CUTILS:
                _same(o[ok], fx[fk][k], f"{name} T={T} {ok}")
            _same(o["props"][:, 0], fx["o_p_dist_target"][k], f"{name} T={T} dist_target")
            _same(o["props"][:, 1], fx["o_p_deadlocked"][k], f"{name} T={T} deadlocked")
            _same(o["props"][:, 2], fx["o_p_ready"][k], f"{name} T={T} ready")
            _same(e.obs_pytree(3, 30), fx["py_d3_p30"][k], f"{name} T={T} py_d3_p30")
            n += 1
    assert n >= 3


def _env(envs, **kw):
    from flatland_marl_amd.hip_backend import BatchedRailEnv
    return BatchedRailEnv(envs, **kw)


def _check_cutils(o, fx, k, b, msg):
    for _, gk, fk in CUTILS:
        _same(o[gk].cpu().numpy()[b], fx[fk][k], f"{msg} {gk}")
    pr = o["props"].cpu().numpy()[b]
    _same(pr[:, 0], fx["o_p_dist_target"][k], f"{msg} dist_target")
    _same(pr[:, 1], fx["o_p_deadlocked"][k], f"{msg} deadlocked")
    _same(pr[:, 2], fx["o_p_ready"][k], f"{msg} ready")


@pytest.mark.gpu
@pytest.mark.parametrize("name", DENSE)
def test_hip_path_replays_dense_reference_episode(name):
    """env 0 builds the two observations with separate launches, env 1 (same env, second batch) with the fused launch;
    the distance maps and static tables are rebuilt from the resident grid in the middle of the episode."""
    import torch
    fx, head = util.load(name), util.load(HEAD[name])
    st = util.static_of(fx)
    e_sep, e_fused = _env([st]), _env([st, st])
    ss = {int(t): k for k, t in enumerate(fx["state_steps"])}
    os_ = {int(t): k for k, t in enumerate(fx["obs_steps"])}
    rebuild_at = int(fx["obs_steps"][1]) - 3
    for t in range(len(fx["actions"])):
        a = torch.from_numpy(fx["actions"][t][None, :].copy()).cuda()
        rew, done, _ = e_sep.step(a)
        e_fused.step(torch.cat([a, a]))
        T = t + 1
        if T == rebuild_at:
            for e in (e_sep, e_fused):
                e.rebuild_distance_maps()
            np.testing.assert_array_equal(e_sep.distance_map(0)[0], head["dm_u16"])
        o = e_sep.obs_cutils()
        of, tf = e_fused.obs_both(3, 30)
        if T in ss:
            _same(e_sep.state()[0][0], fx["states"][ss[T]], f"{name} T={T} state")
            _same(e_fused.state()[0][1], fx["states"][ss[T]], f"{name} T={T} state (fused batch)")
            _same(rew.cpu().numpy()[0], fx["rewards"][ss[T]], f"{name} T={T} rewards")
            _same(done.cpu().numpy()[0], fx["dones"][ss[T]], f"{name} T={T} dones")
        if T in os_:
            k = os_[T]
            _check_cutils(o, fx, k, 0, f"{name} T={T} separate")
            _same(e_sep.obs_tree(3, 30).cpu().numpy()[0], fx["py_d3_p30"][k], f"{name} T={T} py_d3_p30 separate")
            for b in (0, 1):
                _check_cutils(of, fx, k, b, f"{name} T={T} fused[{b}]")
                _same(tf.cpu().numpy()[b], fx["py_d3_p30"][k], f"{name} T={T} py_d3_p30 fused[{b}]")
    e_sep.check(); e_fused.check()
    _same(e_sep.state_aux()[0], fx["o_aux"][-1], f"{name} aux columns at the last snapshot")


@pytest.mark.gpu
@pytest.mark.parametrize("name", DENSE)
def test_injected_dense_reference_states_reproduce_the_snapshots(name):
    fx = util.load(name)
    env = _env([util.static_of(fx)])
    ss = {int(t): k for k, t in enumerate(fx["state_steps"])}
    for k, T in enumerate(int(t) for t in fx["obs_steps"]):
        env.set_state(fx["states"][ss[T]][None], fx["o_aux"][k][None], np.array([T], dtype=np.int32))
        of, tf = env.obs_both(3, 30)
        _check_cutils(of, fx, k, 0, f"{name} T={T} injected")
        _same(tf.cpu().numpy()[0], fx["py_d3_p30"][k], f"{name} T={T} injected py_d3_p30")
    env.check()


def _replica_rng(b):
    st = np.random.RandomState([b]).get_state()
    return np.array(st[1], dtype=np.uint32), int(st[2])


@pytest.mark.gpu
@pytest.mark.parametrize("head,steps,every", [("cfg5_fwd_head", 420, 5), ("cfg4_fwd_head", 460, 4)])
def test_batched_dense_run_matches_oracle_with_depth3_fused_obs_and_autoreset(head, steps, every):
    """B = 3 replicas of a large map with the departures brought forward (>= 80 % of the agents on the map after ~200
    steps), forward-biased synthetic actions, one replica with a shortened episode (auto-reset crossed at least twice) and
    one with frequent malfunctions; every `every` steps the fused launch obs_both(3, 30) and the state are compared with
    the scalar oracle, which steps and builds its observations on the same schedule (sticky deadlock flags)."""
    from oracle import orc
    from flatland_marl_amd import synth
    fx = util.load(head)
    envs = []
    for b in range(3):
        key, pos = _replica_rng(300 + b)
        st = util.static_of(fx, key, pos)
        st["earliest"] = (np.asarray(st["earliest"]) // (8 if b < 2 else 4)).astype(np.int32)
        if b == 1:
            st["T"] = np.int32(150)
        if b == 2:
            st["malf_rate"] = 1 / 400.0
        envs.append(st)
    env = _env(envs)
    oracles = [orc.OracleEnv(st) for st in envs]
    A = env.A
    tcount, resets, peak = [0, 0, 0], 0, 0
    for it in range(steps):
        rew, done, done_all = env.step_synth(23, 7, 1, auto_reset=True)
        chk = it % every == 0
        if chk:
            o, tr = env.obs_both(3, 30)
            o = {k: v.cpu().numpy() for k, v in o.items()}
            tr = tr.cpu().numpy()
            st_g = env.state()[0]
            rew, done, done_all = rew.cpu().numpy(), done.cpu().numpy(), done_all.cpu().numpy()
        for b, oe in enumerate(oracles):
            r_o, d_o, da = oe.step(synth.forward_biased_actions(23, 7 + b, tcount[b], A))
            tcount[b] += 1
            if chk:
                st_o = oe.state()
                peak = max(peak, int((st_o[:, 0] >= 0).sum()))
                _same(st_g[b], st_o, f"{head} replica {b} iter {it} state")
                _same(rew[b], r_o, f"{head} replica {b} iter {it} rewards")
                _same(done[b], d_o, f"{head} replica {b} iter {it} dones")
                assert bool(done_all[b]) == da
                exp = oe.obs_cutils(31, 500)
                for ok, gk, _ in CUTILS:
                    _same(o[gk][b], exp[ok], f"{head} replica {b} iter {it} {gk}")
                _same(o["props"][b], exp["props"], f"{head} replica {b} iter {it} props")
                _same(tr[b], oe.obs_pytree(3, 30), f"{head} replica {b} iter {it} depth-3 tree")
            if da:
                key, pos = oe.get_rng()
                oracles[b] = orc.OracleEnv(envs[b])
                oracles[b].set_rng(key, pos)
                tcount[b] = 0
                resets += 1
    env.check()
    assert resets >= 2 and peak >= 0.8 * A, (resets, peak)
    key, pos = env.rng_state()
    for b, oe in enumerate(oracles):
        k_o, p_o = oe.get_rng()
        assert pos[b] == p_o
        np.testing.assert_array_equal(key[b], k_o)
