"""GPU: the BIN launch classes (round 6: compile-time carving, agents an upper bound, the call's depth) on BATCHES of the shapes they were made for -- 24 envs over two
generated levels of a Round-2 test, so that the per-env strides of the HBM scratch (work lists, items, bucket ends, paths) are exercised under the bins' carvings --, two
replicas (one per level) shadowed by the CPU oracle on every step: state, the flatland_cutils observation and the upstream tree from the fused launch at the given depth,
and the flatland_cutils builder alone.  The class every launch took is asserted.  (One env of every shape: tests/test_gpu_round2_table.py; the exact classes and
classes 15 / 21 at shard size: tests/test_gpu_fullsize.py.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CUTILS = (("agent_attr", "attr"), ("forest", "forest"), ("adjacency", "adjacency"), ("node_order", "node_order"),
          ("edge_order", "edge_order"), ("valid_actions", "valid"), ("props", "props"))


@pytest.mark.parametrize("test,depth,klass_both,klass_alone,steps", [
    ("Test_1", 3, (11, 0), (6, 0), 60),        # 10 agents: class 1's bin twin at depth 3
    ("Test_3", 2, (16, 0), (20, 0), 60),       # 50 agents on a map TALLER than wide: the two-stage bins with compact prediction keys
    ("Test_4", 2, (12, 0), (7, 0), 50),        # 80 agents at depth 2: class 2's bin twin (the builder alone: the exact class 7)
    ("Test_4/60", 3, (12, 0), (17, 0), 50),    # Test_4's map with 60 agents: fewer than the exact classes' 80 -- bins 12 and 17
    ("Test_5", 3, (13, 0), (8, 0), 50),        # 80 agents, 239 / 324 rail cells: class 3's bin twin at depth 3, class 8
    ("Test_10", 2, (15, 0), (18, 0), 40),      # 100 agents, 1 265 / 1 319 rail cells: rounds of 32 agents with HBM lists
    ("Test_11", 3, (4, 0), (9, 0), 30),        # 200 agents: the large-map classes as bins
    ("Test_13", 2, (4, 2), (9, 2), 24),      # 400 agents, 3 025 / 2 548 rail cells: split 2 -- class 4 / 9 for the level that fits, bins 14 / 19 (no LDS successor table) for the other
])
def test_bin_classes_on_batches_match_the_oracle(test, depth, klass_both, klass_alone, steps):
    from flatland_marl_amd import synth, workload as wl
    from flatland_marl_amd.hip_backend import BatchedRailEnv
    from oracle import orc
    if "/" in test:      # a Round-2 row with another number of agents (native host generators, as workload.generate_level)
        from flatland_marl_amd import generators as gen
        name, n_agents = test.split("/")
        p = wl.round2_params(name)
        levels = []
        for lv in (1, 2):
            rg = gen.sparse_rail_generator(max_num_cities=p["n_cities"], grid_mode=p["grid_mode"], max_rails_between_cities=p["max_rails_between_cities"],
                                           max_rail_pairs_in_city=p["max_rail_pairs_in_city"])
            lg = gen.sparse_line_generator(dict(zip(p["speed_values"], p["speed_probs"])))
            st0 = gen.np_random(p["seeds"][lv]).get_state()
            levels.append(gen.generate_env(p["width"], p["height"], int(n_agents), rg, lg, st0[1], st0[2], 1.0 / p["malfunction_interval"],
                                           p["malfunction_duration_min"], p["malfunction_duration_max"]))
    else:
        levels = [wl.generate_level(test, lv) for lv in (1, 2)]
    B = 24
    envs = []
    for b in range(B):
        e = dict(levels[b % 2])
        key, pos = wl.replica_rng(b)
        e["mt_key"], e["mt_pos"] = key, pos
        e["malf_rate"] = 1.0 / 200.0
        envs.append(e)
    env = BatchedRailEnv(envs)
    A = env.A
    shadow = (1, B - 2)
    oracles = {b: orc.OracleEnv(envs[b]) for b in shadow}
    dms = {b: o.distance_map() for b, o in oracles.items()}
    seen = set()
    for t in range(steps):
        kind = 2 if t % 10 else 0
        env.step_synth(9, 0, kind, auto_reset=False)
        o, tree = env.obs_both(depth, 30)
        seen.add(("both",) + tuple(env.last_obs_class()[:2]))
        st, _ = env.state()
        ob = {k: v.cpu().numpy() for k, v in o.items()}
        tr = tree.cpu().numpy()
        alone = None
        if t % 3 == 1:
            alone = {k: v.clone() for k, v in env.obs_cutils().items()}
            seen.add(("alone",) + tuple(env.last_obs_class()[:2]))
        for b, orc_env in oracles.items():
            if kind == 2:
                s = orc_env.state()
                acts = synth.spfollow_actions(9, b, t, s[:, 3], s[:, 0:2], s[:, 2], np.asarray(envs[b]["grid"]), *dms[b])
            else:
                acts = synth.uniform_actions(9, b, t, A)
            orc_env.step(acts)
            np.testing.assert_array_equal(st[b], orc_env.state(), err_msg=f"{test} replica {b} step {t} state")
            exp = orc_env.obs_cutils(31, 500)
            for key, okey in CUTILS:
                np.testing.assert_array_equal(ob[key][b], exp[okey], err_msg=f"{test} replica {b} step {t} {key}")
                if alone is not None:
                    np.testing.assert_array_equal(alone[key].cpu().numpy()[b], exp[okey], err_msg=f"{test} replica {b} step {t} {key} (builder alone)")
            if t % 2 == 0:
                np.testing.assert_array_equal(tr[b], orc_env.obs_pytree(depth, 30), err_msg=f"{test} replica {b} step {t} depth-{depth} tree")
    env.check()
    assert (env.state()[0][:, :, 0] >= 0).sum() > B
    assert seen == {("both",) + tuple(klass_both), ("alone",) + tuple(klass_alone)}, seen
