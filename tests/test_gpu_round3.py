"""GPU, round 3: the evaluator's aggregate scores as device sums (fl_scores), the refusal of a position off the rail
(fl_set_state), the on-device shortest-path-following action stream shadowed by the oracle, distinct generated maps in one
batch, and bench.py starting its own ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _env(envs, **kw):
    from flatland_marl_amd.hip_backend import BatchedRailEnv
    return BatchedRailEnv(envs, **kw)


def test_score_sums_equal_the_reference_evaluator_scores():
    """fl_scores = sum over finished episodes of (1 + R / (T * A), arrived / A) (flatland/evaluators/service.py:875-879, 900-913):
    two replicas of the golden `cfg2_filtered` episode, one of them run twice (auto-reset), against the scores the capture
    script computed from the reference env"""
    import torch
    fx = util.load("cfg2_filtered")
    st = util.static_of(fx)
    env = _env([st, dict(st)])
    raw = np.array(fx["actions"], dtype=np.uint8)
    for t in range(len(raw)):
        a = torch.from_numpy(np.stack([raw[t], raw[t]])).cuda()
        _, _, done_all = env.step(a, filter_required=True)
    assert done_all.cpu().numpy().tolist() == [1, 1]
    s = env.scores().cpu().numpy()
    exp = fx["evaluator_scores"]
    assert s[2] == 2 and s[0] == exp[0] + exp[0] and s[1] == exp[1] + exp[1]
    m = env.metrics().cpu().numpy()
    assert m[3] == 2
    # the all-reduce helper carries the two sums as 2**-32 fixed point beside the int64 metrics (world size 1 here)
    from flatland_marl_amd import dist_utils
    mm, ss = dist_utils.reduce_metrics(env.metrics().clone(), env.scores().clone())
    assert mm.cpu().numpy().tolist() == m.tolist()
    np.testing.assert_allclose(ss.cpu().numpy()[:2], s[:2], rtol=0, atol=2.0 ** -32)
    assert ss.cpu().numpy()[2] == 2
    # the scores keep an episode counter of their own: resetting the metrics alone leaves sums AND count of the same window
    env.metrics(reset=True)
    assert env.scores().cpu().numpy().tolist() == s.tolist() and env.metrics().cpu().numpy()[3] == 0
    env.scores(reset=True)
    assert env.scores().cpu().numpy().tolist() == [0.0, 0.0, 0.0]
    env.check()


def test_set_state_refuses_a_position_off_the_rail():
    from flatland_marl_amd.hip_backend import FlatlandHipError
    fx = util.load("cfg1_uniform")
    env = _env([util.static_of(fx)])
    grid = np.asarray(fx["grid"])
    empty = np.argwhere(grid == 0)[0]
    st, _ = env.state()
    st[0, 0, 0], st[0, 0, 1], st[0, 0, 3] = empty[0], empty[1], 3          # MOVING on a cell without rail
    with pytest.raises(FlatlandHipError, match="FL_ERR_STATE_SYNC"):
        env.set_state(st)
    st, _ = env.state()
    st[0, 1, 9], st[0, 1, 10] = empty[0], empty[1]                          # old_position off the rail
    with pytest.raises(FlatlandHipError, match="FL_ERR_STATE_SYNC"):
        env.set_state(st)
    env.step_synth(1, 0, 0, auto_reset=False)                               # the batch is untouched and keeps running
    env.obs_cutils()
    env.check()


@pytest.mark.parametrize("name,steps", [("cfg2_uniform", 400), ("cfg3_uniform", 250)])
def test_shortest_path_following_stream_is_shadowed_by_the_oracle(name, steps):
    """kind 2 of the on-device action stream: the oracle gets the actions synth.spfollow_actions derives on the host from ITS
    state; states, rewards, dones and the observations have to stay equal -- under dense traffic, with arrivals"""
    from flatland_marl_amd import synth
    from oracle import orc
    fx = util.load(name)
    st = util.static_of(fx)
    envs = [st, dict(st)]
    envs[1]["mt_key"], envs[1]["mt_pos"] = np.random.RandomState([5]).get_state()[1:3]
    env = _env(envs)
    oracles = [orc.OracleEnv(e) for e in envs]
    dm, slot = oracles[0].distance_map()
    seed, arrived, tc = 9, 0, [0, 0]
    for t in range(steps):
        acts = []
        for b, o in enumerate(oracles):
            s = o.state()
            acts.append(synth.spfollow_actions(seed, b, tc[b], s[:, 3], s[:, 0:2], s[:, 2], np.asarray(st["grid"]), dm, slot))
        rew, done, done_all = env.step_synth(seed, 0, 2, auto_reset=True)
        obs = env.obs_cutils()
        got, _ = env.state()
        for b, o in enumerate(oracles):
            r_o, d_o, da = o.step(acts[b])
            tc[b] += 1
            np.testing.assert_array_equal(got[b], o.state(), err_msg=f"env {b} step {t}")
            np.testing.assert_array_equal(rew.cpu().numpy()[b], r_o)
            assert bool(done_all.cpu().numpy()[b]) == da
            exp = o.obs_cutils(31, 500)
            if t % 10 == 0:
                np.testing.assert_array_equal(obs["forest"].cpu().numpy()[b], exp["forest"], err_msg=f"forest env {b} step {t}")
            arrived = max(arrived, int((o.state()[:, 3] == 6).sum()))
            if da:
                key, pos = o.get_rng()
                oracles[b] = orc.OracleEnv(envs[b])
                oracles[b].set_rng(key, pos)
                tc[b] = 0
    env.check()
    assert arrived >= 2, "the stream is supposed to bring agents to their targets"


def test_distinct_generated_maps_in_one_batch_match_the_oracle():
    from flatland_marl_amd import synth, workload as wl
    from oracle import orc
    envs, seed = wl.make_envs("cfg2", B=10, distinct=10)
    assert len({e["grid"].tobytes() for e in envs}) == 10
    env = _env(envs)
    oracles = [orc.OracleEnv(e) for e in envs]
    for t in range(60):
        rew, done, done_all = env.step_synth(seed, 0, 0, auto_reset=False)
        o, tree = env.obs_both(2, 30)
        st, _ = env.state()
        for b, orc_env in enumerate(oracles):
            orc_env.step(synth.uniform_actions(seed, b, t, env.A))
            np.testing.assert_array_equal(st[b], orc_env.state(), err_msg=f"env {b} step {t}")
            exp = orc_env.obs_cutils(31, 500)
            if t % 6 == 0:
                np.testing.assert_array_equal(o["forest"].cpu().numpy()[b], exp["forest"])
                np.testing.assert_array_equal(o["agent_attr"].cpu().numpy()[b], exp["attr"])
                np.testing.assert_array_equal(tree.cpu().numpy()[b], orc_env.obs_pytree(2, 30))
    env.check()


def test_two_hundred_agents_on_a_tall_map_match_the_oracle():
    """hundreds of agents on a map taller than wide: the prediction keys collide (flatland_cutils keys its maps by col * W + row,
    tool.h:391-398) AND the items live in HBM scratch, laid out bucket-major with two-piece conflict queries (the launch
    configuration of cfg5) -- the committed goldens have tall maps of three agents and large maps that are square"""
    from flatland_marl_amd import generators as gen, synth
    from oracle import orc
    rg = gen.sparse_rail_generator(max_num_cities=8, grid_mode=False, max_rails_between_cities=2, max_rail_pairs_in_city=2)
    lg = gen.sparse_line_generator({1.0: 0.25, 0.5: 0.25, 1.0 / 3.0: 0.25, 0.25: 0.25})
    envs = []
    for seed in (7, 8):
        st = gen.np_random(seed).get_state()
        envs.append(gen.generate_env(40, 76, 200, rg, lg, st[1], st[2], 1.0 / 200, 20, 50))
    assert envs[0]["grid"].shape == (76, 40)
    env = _env(envs)
    oracles = [orc.OracleEnv(e) for e in envs]
    keys = (("agent_attr", "attr"), ("forest", "forest"), ("adjacency", "adjacency"), ("node_order", "node_order"),
            ("edge_order", "edge_order"), ("valid_actions", "valid"))
    for t in range(64):
        env.step_synth(11, 0, 1, auto_reset=False)
        st, _ = env.state()
        for b, o in enumerate(oracles):
            o.step(synth.forward_biased_actions(11, b, t, env.A))
            np.testing.assert_array_equal(st[b], o.state(), err_msg=f"env {b} step {t}")
        got, tree = env.obs_both(3, 30) if t % 2 else (env.obs_cutils(), env.obs_tree(3, 30))
        for b, o in enumerate(oracles):
            exp = o.obs_cutils(31, 500)          # (every step: keeps the oracle's sticky deadlock flags in step)
            if t % 4 < 2:
                for g, e in keys:
                    np.testing.assert_array_equal(got[g].cpu().numpy()[b], exp[e], err_msg=f"env {b} step {t} {g}")
                np.testing.assert_array_equal(tree.cpu().numpy()[b], o.obs_pytree(3, 30), err_msg=f"env {b} step {t} tree")
    assert (env.state()[0][:, :, 0] >= 0).sum() > 10      # trains are on the map
    env.check()


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: the parent starts two ranks before any GPU call and relays rank 0's line
    (here both ranks share the one GPU of the box: FL_DIST_BACKEND=gloo)"""
    env = dict(os.environ, FL_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "2", "--envs", "8",
                        "--no-dephase", "--event-steps", "4"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and len(d["agent_steps_per_rank"]) == 2
    assert d["dist_backend"] == "gloo" and d["device_per_rank"] == [0, 0]       # the rehearsal: both ranks on the one GPU
    A = d["config"]["agents"]
    assert d["agent_steps_per_rank"] == [8 * A * 12, 8 * A * 12] and d["agent_steps"] == 2 * 8 * A * 12
    assert abs(d["value"] - 2 * 8 * A * 12 / (d["ms_per_step"] * 12 * 1e-3)) <= 2e-5 * d["value"]     # sum over ranks / max time (the line carries 6 significant digits)
    assert "cpu_baseline" not in d and d["scaling"] == "weak"
    assert d["roofline"]["frac"] > 0 and "roofline.note" in d["notes"] and len(lines[0]) < 8192


def test_bench_two_ranks_over_rccl_on_two_gpus():
    """`python bench.py --gpus 2` on RCCL (backend "nccl"), one rank per GPU: runs wherever at least two GPUs are visible (the
    driver's multi-GPU node; skipped on the one-GPU box, where test_bench_starts_its_own_ranks rehearses the same path on gloo).
    Every rank on a device of its own, equal shards, the job-level sums from the one all-reduce."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FL_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5", "--envs", "64",
                        "--event-steps", "8"], env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0 and any(k in r.stderr for k in ("ncclSystemError", "ncclUnhandledCudaError", "NCCL error", "unhandled system error")):
        pytest.skip("RCCL could not initialise on this box (not this library's code): " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    A = d["config"]["agents"]
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["dist_backend"] == "nccl"
    assert sorted(d["device_per_rank"]) == [0, 1]
    assert d["agent_steps_per_rank"] == [64 * A * 40] * 2 and d["agent_steps"] == 2 * 64 * A * 40
    assert abs(d["value"] - d["agent_steps"] / (d["ms_per_step"] * 40 * 1e-3)) <= 1e-6 * d["value"]
    assert d["scaling"] == "weak" and "cpu_baseline" not in d


def test_capacities_are_checked_before_anything_is_allocated():
    """fl_reserve / fl_commit refuse capacities the table kernels cannot hold (the reverse-BFS kernel keeps the env's neighbour
    table, visited bitmaps and queues in LDS: about 10 900 rail cells); an env that does not fit a live batch's capacities is
    FL_ERR_CAPACITY at fl_load_env (tests/test_gpu_reload.py).  The kernel-side FL_ERR_CAPACITY latches (BFS ring, 16-bit
    distances of a tree walk) cannot be reached below those limits: a BFS level of a map of 10 900 cells has at most ~1 000
    states (ring of 4 096), a 31-node tree walks at most 4 laps of the longest possible loop (43 600 < 65 535)."""
    from flatland_marl_amd.hip_backend import BatchedRailEnv, FlatlandHipError
    fx = util.load("cfg1_uniform")
    with pytest.raises(FlatlandHipError, match="FL_ERR_ARG.*distance-map kernel's LDS"):
        BatchedRailEnv([util.static_of(fx)], reserve=(4, 15000))
    with pytest.raises(FlatlandHipError, match="FL_ERR_ARG.*16383 rail cells"):
        BatchedRailEnv([util.static_of(fx)], reserve=(4, 20000))
    env = BatchedRailEnv([util.static_of(fx)], reserve=(8, 10000))      # fits the table kernels: the batch steps ...
    env.step_synth(1, 0, 0, auto_reset=False)
    env.check()
    with pytest.raises(FlatlandHipError, match="FL_ERR_ARG.*observation kernels' LDS"):   # ... the observation index of 10 000 cells does not fit
        env.obs_both(3, 30)
    env = BatchedRailEnv([util.static_of(fx)], reserve=(8, 5000))
    env.step_synth(1, 0, 0, auto_reset=False)
    env.obs_both(3, 30)
    env.check()


THREEWAY = ("threeway_cfg2", "threeway_cfg3")


@pytest.mark.parametrize("name", THREEWAY)
@pytest.mark.parametrize("fused", [False, True])
def test_grid_with_a_three_way_cell_matches_the_reference(name, fused):
    """a generated map in which one switch got a third way on for one direction of travel (oracle/refharness/capture_threeway.py ran
    the real reference on it): no Flatland rail cell type has that, the observation kernels take their DFS-slot node tables for
    such a batch (max_branch > 2) instead of the compact ones"""
    import torch
    fx = util.load(name)
    env = _env([util.static_of(fx)])
    dm, slot = env.distance_map(0)
    np.testing.assert_array_equal(dm, fx["dm_u16"])
    keys = (("agent_attr", "o_attr"), ("forest", "o_forest"), ("adjacency", "o_adjacency"), ("node_order", "o_node_order"),
            ("edge_order", "o_edge_order"), ("valid_actions", "o_valid"))
    for t in range(len(fx["state"])):
        if t > 0:
            rew, done, _ = env.step(torch.from_numpy(fx["actions"][t - 1][None, :].copy()).cuda())
            np.testing.assert_array_equal(rew.cpu().numpy()[0], fx["reward"][t - 1])
        np.testing.assert_array_equal(env.state()[0][0], fx["state"][t], err_msg=f"t={t}")
        assert not fx["cutils_raised"][t]
        if fused:
            got, tree3 = env.obs_both(3, 30)
            tree2 = env.obs_tree(2, 30) if t % 10 == 0 else None
        else:
            got = env.obs_cutils()
            tree3 = env.obs_tree(3, 30)
            tree2 = env.obs_both(2, 30)[1] if t % 10 == 0 else None
        for g, e in keys:
            np.testing.assert_array_equal(got[g].cpu().numpy()[0], fx[e][t], err_msg=f"t={t} {g}")
        np.testing.assert_array_equal(tree3.cpu().numpy()[0], fx["py_d3_p30"][t], err_msg=f"t={t} depth-3 tree")
        if tree2 is not None:
            np.testing.assert_array_equal(tree2.cpu().numpy()[0], fx["py_d2_p30"][t], err_msg=f"t={t} depth-2 tree")
    env.check()


def test_mixed_shapes_step_together_like_their_golden_episodes():
    """MixedBatch: envs of different (agents, height, width) -- the reference evaluates the Round-2 tests of different sizes back to
    back -- grouped into one handle per shape on streams of their own; every env follows its own golden episode (state, rewards,
    observations), whatever its neighbours in the list are."""
    from flatland_marl_amd.hip_backend import MixedBatch
    names = ["cfg1_sparse", "cfg2_fwd", "cfg3_uniform", "cfg1_malf50", "cfg0_tall_uniform", "cfg2_uniform"]
    fxs = [util.load(n) for n in names]
    mb = MixedBatch([util.static_of(fx) for fx in fxs])
    assert len(mb.groups) == 4 and sorted(g.B for g in mb.groups) == [1, 1, 2, 2]      # 30x30/7 twice, 30x30/20 twice, 35x30/80, 40x26/20
    acts = [util.actions_of(fx) for fx in fxs]
    obs_steps = [{int(t): k for k, t in enumerate(fx["obs_steps"])} for fx in fxs]
    n_obs = 0
    for t in range(150):
        res = mb.step([a[t] for a in acts])
        o = mb.obs_cutils()
        for i, fx in enumerate(fxs):
            st, el = mb.state(i)
            np.testing.assert_array_equal(st, util.golden_state(fx, t), err_msg=f"{names[i]} step {t}")
            rew = mb.pick(i, res)[0].cpu().numpy()
            np.testing.assert_array_equal(rew, fx["s_reward"][t], err_msg=f"{names[i]} rewards, step {t}")
            if (t + 1) in obs_steps[i]:
                k = obs_steps[i][t + 1]
                mine = mb.pick(i, o)
                np.testing.assert_array_equal(mine["forest"].cpu().numpy(), fx["o_forest"][k], err_msg=f"{names[i]} forest, step {t}")
                np.testing.assert_array_equal(mine["agent_attr"].cpu().numpy(), fx["o_attr"][k], err_msg=f"{names[i]} attr, step {t}")
                n_obs += 1
    mb.check()
    assert n_obs > 40
    mb.close()


def test_mixed_batch_results_are_ordered_with_the_callers_stream():
    """The groups of a MixedBatch run on non-blocking streams of their own; what a call returns is read by the CALLER on its current
    stream.  No state() / check() / sync() (which would drain the groups' streams) before the reads here: a `.cpu()` straight after the
    call has to see the finished results -- the caller's stream waits for the groups' streams inside every MixedBatch call."""
    import torch
    from flatland_marl_amd.hip_backend import MixedBatch
    names = ["cfg3_uniform", "cfg1_sparse", "cfg2_fwd", "cfg3_uniform", "cfg2_uniform"]
    fxs = [util.load(n) for n in names]
    mb = MixedBatch([util.static_of(fx) for fx in fxs])
    acts = [util.actions_of(fx) for fx in fxs]
    obs_steps = [{int(t): k for k, t in enumerate(fx["obs_steps"])} for fx in fxs]
    n_obs = 0
    for t in range(100):
        res = mb.step([a[t] for a in acts])
        rews = [mb.pick(i, res)[0].cpu().numpy() for i in range(len(fxs))]            # straight away, on the caller's stream
        o = mb.obs_cutils()
        forests = [mb.pick(i, o)["forest"].cpu().numpy() for i in range(len(fxs))]
        m = mb.metrics()
        assert int(m[2]) == sum(len(fx["init_dir"]) for fx in fxs) * (t + 1)
        for i, fx in enumerate(fxs):
            np.testing.assert_array_equal(rews[i], fx["s_reward"][t], err_msg=f"{names[i]} rewards, step {t}")
            if (t + 1) in obs_steps[i]:
                np.testing.assert_array_equal(forests[i], fx["o_forest"][obs_steps[i][t + 1]], err_msg=f"{names[i]} forest, step {t}")
                n_obs += 1
    torch.cuda.synchronize()
    mb.check()
    assert n_obs > 20
    mb.close()
