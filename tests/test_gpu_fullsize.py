"""GPU, BASELINE.json full sizes: cfg3 (1024 envs x 35x30 x 80 agents, depth-3 tree), the per-GPU shard of cfg4 (512 envs x
60x60 x 80 agents, depth-2 tree: HBM work lists + longest-first workgroup order) and the per-GPU shard of cfg5 (256 envs x
150x150 x 400 agents, depth-3 tree + masked distance-map rebuild of the envs that just reset) --
the batch sizes at which the per-env HBM scratch strides (predicted paths, prediction items, bucket offsets, work lists) are
actually exercised.  Size-independent properties: replica independence (an env inside the full batch == the same env stepped
alone, observations included) on picked replicas, and two replicas shadowed by the CPU oracle step by step."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CUTILS = (("agent_attr", "attr"), ("forest", "forest"), ("adjacency", "adjacency"), ("node_order", "node_order"),
          ("edge_order", "edge_order"), ("valid_actions", "valid"), ("props", "props"))


def _same(got, exp, msg):
    got = np.asarray(got)
    if not np.array_equal(got, exp):
        bad = np.argwhere(got != exp)
        raise AssertionError(f"{msg}: {len(bad)} mismatches, first {bad[0].tolist()}: {got[tuple(bad[0])]} vs {exp[tuple(bad[0])]}")


# fixed launch class of every case (fl_obs_layout.h ObsFixed<k>; BatchedRailEnv.last_obs_class): `klass` = (class, split) of the full
# batch's launch -- split 1: the batch's largest map exceeds the class's rail cells, the class's body builds the envs that fit it and
# the runtime-carving body of the same kernel the others (k_obs_split), so those cases cover BOTH bodies at shard size; split 2 (round 6,
# large maps): the others run the larger bin class's body (14 / 19: no LDS successor table) -- every env on a compile-time carving.
@pytest.mark.parametrize("workload,B,steps,picks,shadow,rebuild,distinct,depth,klass", [
    # (replicas with b % 7 == 3 have short episodes: 3, 766, 1018 / 3, 255 restart inside the window)
    ("cfg3", 1024, 60, (0, 257, 766, 1018, 1023), (3, 1022), False, 10, 3, (2, 0)),
    # the per-GPU shard of cfg4 as the bench runs it: rounds of 32 agents, pass-B work lists in HBM scratch (per-env stride) AND
    # the longest-first workgroup order of k_env_order (more envs than CUs)
    ("cfg4", 512, 40, (0, 129, 255, 256, 500, 511), (3, 510), False, 4, 2, (3, 0)),      # (class 3 holds every level of the row: 603 .. 677 rail cells)
    ("cfg5", 256, 40, (0, 85, 170, 255), (3, 254), True, 2, 3, (4, 2)),      # (split 2: class 4's body / bin class 14's for the 3 025-cell level)
    # the SINGLE-map shards of bench.py's EXTRA_WORKLOADS and of the profiles of record: every env fits its class -- k_obs<4,2,3>
    # with k_env_order and the per-env HBM work-list stride at B = 512, k_obs<2,2,4> with the masked rebuild at B = 256
    ("cfg4", 512, 40, (0, 129, 255, 256, 500, 511), (3, 510), False, 0, 2, (3, 0)),
    ("cfg5", 256, 40, (0, 85, 170, 255), (3, 254), True, 0, 3, (4, 0)),
    # four envs per CU at the north-star shape: class 5 -- 512-thread workgroups in rounds of 16 agents, two a CU (the solo runs of the
    # picks take class 1, the one-round kernel: two different kernels, the same bytes)
    ("cfg2", 1024, 90, (0, 257, 766, 1018, 1023), (3, 5, 1022), False, 0, 2, (5, 0)),
    # THE HEADLINE LAUNCH: bench.py's cfg2 step is obs_both(2, 30) at exactly B = 256 on class 1 (k_obs<3,0,1>, one env per CU) --
    # the launch the north-star number is quoted on, under the oracle
    ("cfg2", 256, 120, (0, 37, 101, 200, 255), (3, 5, 254), False, 0, 2, (1, 0)),
])
def test_full_size_batch_replicas_equal_solo_runs_and_the_oracle(workload, B, steps, picks, shadow, rebuild, distinct, depth, klass):
    from flatland_marl_amd import synth, workload as wl
    from flatland_marl_amd.hip_backend import BatchedRailEnv
    from oracle import orc
    envs, seed = wl.make_envs(workload, B=B, distinct=distinct)
    assert len({e["grid"].tobytes() for e in envs}) == (distinct or len(wl.WORKLOADS[workload]["bases"]))
    # short episodes for some replicas, so that auto-resets (and the masked rebuild) happen inside the window
    for b in range(B):
        if b % 7 == 3:
            envs[b] = dict(envs[b])
            envs[b]["T"] = np.int32(17 + b % 11)
    env = BatchedRailEnv(envs)
    A = env.A
    solo = {b: BatchedRailEnv([envs[b]]) for b in picks}
    oracles = {b: orc.OracleEnv(envs[b]) for b in shadow}
    tc = {b: 0 for b in shadow}
    for t in range(steps):
        rew, done, done_all = env.step_synth(seed, 0, 0, auto_reset=True)
        o, tree = env.obs_both(depth, 30)
        if t == 0:      # which kernel builds the batch: the class, for every env (split 0) or for the envs that fit it (split 1)
            fix, split, n_fit = env.last_obs_class()
            rails = np.array([int((np.asarray(e["grid"]) != 0).sum()) for e in envs])
            cap = {1: 256, 2: 232, 3: 680, 4: 2816, 5: 256}[klass[0]]
            assert (fix, split) == klass and n_fit == int((rails <= cap).sum()) and (split == 0) == (n_fit == B), (fix, split, n_fit)
            if split:   # both bodies are under test: replicas on either side of the class's capacity among the picks / shadows
                assert {bool(rails[b] <= cap) for b in picks} == {True, False}, rails[list(picks)]
        if rebuild:
            env.rebuild_distance_maps(env.done_all)
        st, el = env.state()
        check_obs = t % 8 == 0 or t == steps - 1
        ob = {k: v.cpu().numpy() for k, v in o.items()} if check_obs else None
        tr = tree.cpu().numpy() if check_obs else None
        for b, s_env in solo.items():
            s_rew, _, s_da = s_env.step_synth(seed, b, 0, auto_reset=True)
            s_o, s_tree = s_env.obs_both(depth, 30)
            if rebuild:
                s_env.rebuild_distance_maps(s_env.done_all)
            _same(s_env.state()[0][0], st[b], f"replica {b} step {t} state")
            _same(s_rew.cpu().numpy()[0], rew.cpu().numpy()[b], f"replica {b} step {t} rewards")
            if check_obs:
                for key, _ in CUTILS:
                    _same(s_o[key].cpu().numpy()[0], ob[key][b], f"replica {b} step {t} {key}")
                _same(s_tree.cpu().numpy()[0], tr[b], f"replica {b} step {t} depth-{depth} tree")
        for b, orc_env in oracles.items():
            r_o, d_o, da = orc_env.step(synth.uniform_actions(seed, b, tc[b], A))
            tc[b] += 1
            _same(st[b], orc_env.state(), f"oracle replica {b} step {t} state")
            _same(rew.cpu().numpy()[b], r_o, f"oracle replica {b} step {t} rewards")
            exp = orc_env.obs_cutils(31, 500)       # every step: the deadlock flags are sticky
            if check_obs:
                for key, okey in CUTILS:
                    _same(ob[key][b], exp[okey], f"oracle replica {b} step {t} {key}")
                _same(tr[b], orc_env.obs_pytree(depth, 30), f"oracle replica {b} step {t} depth-{depth} tree")
            if da:
                key, pos = orc_env.get_rng()
                oracles[b] = orc.OracleEnv(envs[b])
                oracles[b].set_rng(key, pos)
                tc[b] = 0
    env.check()
    m = env.metrics().cpu().numpy()
    assert m[2] == B * A * steps and m[3] >= B // 7       # the short episodes ended (and restarted) inside the window
    for s_env in solo.values():
        s_env.check()


@pytest.mark.parametrize("workload,B,steps,depth,rebuild", [("cfg5", 48, 60, 3, True), ("cfg3", 96, 100, 3, False), ("cfg2", 64, 160, 2, False)])
def test_keep_tree_rows_mode_gives_the_same_tensors(workload, B, steps, depth, rebuild):
    """FL_OBS_KEEP_TREE_ROWS (fl_obs_set_mode): the upstream-tree buffer is the previous call's, untouched -- the builder writes the
    real rows and sets only the rows that were real and are not any more, instead of pre-filling the whole slab with -inf per call.
    Two batches of the same envs in lock step, one with the mode: identical tensors (bit patterns) on every step -- through
    auto-resets, the separate fl_obs_tree launch, a launch at another depth in between (another buffer: a full fill, and the masks
    then describe THAT buffer) and a switch of the mode off and on."""
    import torch
    from flatland_marl_amd import workload as wl
    from flatland_marl_amd.hip_backend import BatchedRailEnv
    envs, seed = wl.make_envs(workload, B=B, distinct=2)
    for b in range(B):
        if b % 5 == 3:
            envs[b] = dict(envs[b])
            envs[b]["T"] = np.int32(19 + b % 13)
    plain, keep = BatchedRailEnv(envs), BatchedRailEnv(envs)
    keep.keep_tree_rows()
    other = 2 if depth == 3 else 3
    changed = 0
    prev = None
    for t in range(steps):
        for e in (plain, keep):
            e.step_synth(seed, 0, 2 if t % 40 < 30 else 0, auto_reset=True)
        if t == steps // 2:
            keep.keep_tree_rows(False)
        if t == steps // 2 + 3:
            keep.keep_tree_rows(True)
        res = []
        for e in (plain, keep):
            if t % 7 == 5:
                e.obs_cutils()
                tree = e.obs_tree(depth, 30)
            else:
                _, tree = e.obs_both(depth, 30)
            if t % 11 == 10:                      # another depth in between: its own buffer
                e.obs_tree(other, 30)
            if rebuild:
                e.rebuild_distance_maps(e.done_all)
            res.append(tree)
        assert torch.equal(res[0].view(torch.int64), res[1].view(torch.int64)), f"step {t}"
        cur = torch.isinf(res[0][..., 0]) & (res[0][..., 0] < 0)
        if prev is not None:
            changed += int((cur != prev).sum())
        prev = cur.clone()
    assert changed > 100            # rows did appear and disappear
    plain.check(); keep.check()
    assert plain.metrics().cpu().numpy()[3] >= B // 5


def _big_cfg4_env():
    """the Round-2 row of cfg4 (Test_8) with 20 cities instead of 17: 695 rail cells (the row's own levels: 603 .. 677)"""
    from flatland_marl_amd import generators as gen, workload as wl
    p = wl.round2_params("Test_8")
    rg = gen.sparse_rail_generator(max_num_cities=20, grid_mode=p["grid_mode"], max_rails_between_cities=p["max_rails_between_cities"],
                                   max_rail_pairs_in_city=p["max_rail_pairs_in_city"])
    lg = gen.sparse_line_generator(dict(zip(p["speed_values"], p["speed_probs"])))
    st = gen.np_random(3).get_state()
    e = gen.generate_env(60, 60, 80, rg, lg, st[1], st[2], 1.0 / p["malfunction_interval"], p["malfunction_duration_min"], p["malfunction_duration_max"])
    assert int((np.asarray(e["grid"]) != 0).sum()) == 695
    return e


def _cfg4_digest(steps=24, workload="cfg4", B=512, distinct=4, depth=2, want_class=None, big=False):
    """sha256 over the state, rewards and both observations of every step of the full cfg4 shard (or of another workload);
    want_class: (fixed launch class, split) the launches must have taken"""
    import hashlib
    from flatland_marl_amd import workload as wl
    from flatland_marl_amd.hip_backend import BatchedRailEnv
    envs, seed = wl.make_envs(workload, B=B, distinct=distinct)
    if big:        # every seventh env: a 60 x 60 / 80-agent map of 695 rail cells, beyond class 3's 680
        for b in range(5, B, 7):
            envs[b] = dict(_big_cfg4_env(), mt_key=envs[b]["mt_key"], mt_pos=envs[b]["mt_pos"])
    env = BatchedRailEnv(envs)
    h = hashlib.sha256()
    for t in range(steps):
        rew, done, done_all = env.step_synth(seed, 0, 2, auto_reset=True)      # dense traffic: envs of very different cost
        o, tree = env.obs_both(depth, 30)
        if want_class is not None:
            assert env.last_obs_class()[:2] == tuple(want_class), (workload, env.last_obs_class(), want_class)
        h.update(env.state()[0].tobytes())
        h.update(rew.cpu().numpy().tobytes())
        for k in sorted(o):
            h.update(o[k].cpu().numpy().tobytes())
        h.update(tree.cpu().numpy().tobytes())
    env.check()
    return h.hexdigest()


def _child(switches, *argv):
    import os
    import subprocess
    import sys
    from tests import util
    child = subprocess.run([sys.executable, os.path.abspath(__file__)] + list(argv), env=dict(os.environ, PYTHONPATH=util.ROOT, **{k: "1" for k in switches}),
                           capture_output=True, text=True, timeout=900)
    assert child.returncode == 0, child.stderr[-2000:]
    return dict(ln.split()[1:3] for ln in child.stdout.splitlines() if ln.startswith("DIGEST "))


def test_workgroup_order_and_the_split_of_a_launch_do_not_change_the_results_at_cfg4():
    """The cfg4 shard on four levels of its Round-2 row plus a larger map (695 rail cells, beyond class 3's 680) in every seventh env:
    the default launch is the class's SPLIT kernel (class body / runtime-carving body per env) with the envs handed to the
    workgroups longest first by k_env_order (B > CUs).  FL_OBS_NO_ORDER (read once per process, hence the children) makes workgroup
    k build env k; FL_OBS_NO_SPLIT runs every env on bin class 15; FL_OBS_NO_WL_HEAD (no LDS head of the HBM lists: neither class applies) and
    FL_OBS_NO_SPLIT + FL_OBS_NO_BINS on the runtime-carving kernel k_obs<4,2,0>.  Same bytes all five ways."""
    want = _cfg4_digest(want_class=(3, 1), big=True)
    assert _child(("FL_OBS_NO_ORDER",), "cfg4", "3", "1") == {"cfg4": want}
    # round 6: without the split (or without LDS heads, which class 3 has) the whole batch takes BIN class 15 (rounds of 32 agents, at most
    # 100 agents / 1 344 rail cells) -- k_obs<4,2,15> at shard size; without the bins too: the runtime carving
    assert _child(("FL_OBS_NO_SPLIT",), "cfg4", "15", "0") == {"cfg4": want}
    assert _child(("FL_OBS_NO_WL_HEAD",), "cfg4", "0", "0") == {"cfg4": want}     # (classes 3 and 15 both have an LDS head: runtime carving)
    assert _child(("FL_OBS_NO_SPLIT", "FL_OBS_NO_BINS"), "cfg4", "0", "0") == {"cfg4": want}


CASES = {"cfg3": dict(steps=60, workload="cfg3", B=24, distinct=3, depth=3), "cfg2": dict(steps=80, workload="cfg2", B=16, distinct=4, depth=2)}
CLASS_OF = {"cfg3": (2, 0), "cfg2": (1, 0)}


def test_other_launch_paths_give_the_same_bytes():
    """The launcher's alternative paths, each selected by an environment switch that is read once per process (hence children):
    the runtime LDS carving instead of the fixed launch classes (FL_OBS_NO_FIX) and the 512-thread kernel in rounds of 16 agents,
    two workgroups a CU (FL_OBS_ROUND16, MODE 5), one set of static tables per env instead of one per map (FL_NO_SHARED_TABLES) -- on
    cfg3 (depth 3) and cfg2 (depth 2) batches in dense traffic.  Same bytes
    as the default path, and every run asserts the launch class it took (default: classes 2 / 1; with a switch: none)."""
    want = {k: _cfg4_digest(want_class=CLASS_OF[k], **kw) for k, kw in CASES.items()}
    assert _child(("FL_OBS_NO_FIX",), "paths", "0") == want
    assert _child(("FL_OBS_ROUND16",), "paths", "r16") == want
    assert _child(("FL_NO_SHARED_TABLES",), "paths", "own") == want       # one set of static tables per env instead of one per map


# ---- the flatland_cutils builder ALONE (what the reference's solution launches: solution/eval_env.py:15-17 builds TreeCutils(31, 500) only):
# fl_obs_cutils on the one-pass kernels without the upstream builder (MODE 6 / 7 / 8) and their fixed launch classes 6 .. 10
CUTILS_ALONE_CAP = {6: 256, 7: 232, 8: 680, 9: 2816, 10: 256, 21: 232}


@pytest.mark.parametrize("workload,B,steps,picks,shadow,distinct,klass", [
    ("cfg2", 256, 100, (0, 101, 255), (3, 5, 254), 0, (6, 0)),            # the bench's cfg2_cutils launch: k_obs<6,0,6>, one env per CU
    ("cfg3", 256, 48, (0, 101, 255), (3, 254), 10, (7, 0)),              # one env per CU: k_obs<7,0,7>
    ("cfg3", 1024, 48, (0, 766, 1023), (3, 1022), 10, (21, 0)),          # four envs per CU: k_obs<8,0,21>, rounds of 16 agents, two workgroups a CU
    ("cfg4", 512, 40, (0, 256, 511), (3, 510), 4, (8, 0)),               # k_obs<7,0,8> with k_env_order
    ("cfg5", 256, 32, (0, 85, 255), (3, 254), 0, (9, 0)),                # k_obs<0,2,9>: the stand-alone kernel with its carving compiled in
    ("cfg5", 256, 32, (0, 85, 255), (3, 254), 2, (9, 2)),                # two levels of the row: k_obs_split<0,2,9,19>, both class bodies
    ("cfg2", 1024, 72, (0, 766, 1023), (3, 5, 1022), 0, (10, 0)),        # four envs a CU: k_obs<8,0,10>, two workgroups a CU
])
def test_cutils_alone_full_size_replicas_equal_solo_runs_and_the_oracle(workload, B, steps, picks, shadow, distinct, klass):
    import torch
    from flatland_marl_amd import synth, workload as wl
    from flatland_marl_amd.hip_backend import BatchedRailEnv
    from oracle import orc
    envs, seed = wl.make_envs(workload, B=B, distinct=distinct)
    for b in range(B):
        if b % 7 == 3:
            envs[b] = dict(envs[b])
            envs[b]["T"] = np.int32(17 + b % 11)
    env = BatchedRailEnv(envs)
    A = env.A
    solo = {b: BatchedRailEnv([envs[b]]) for b in picks}
    oracles = {b: orc.OracleEnv(envs[b]) for b in shadow}
    tc = {b: 0 for b in shadow}
    for t in range(steps):
        rew, done, done_all = env.step_synth(seed, 0, 2 if t % 3 else 0, auto_reset=True)      # mostly shortest-path following: trains on the map
        o = env.obs_cutils()
        if t == 0:
            fix, split, n_fit = env.last_obs_class()
            rails = np.array([int((np.asarray(e["grid"]) != 0).sum()) for e in envs])
            cap = CUTILS_ALONE_CAP[klass[0]]
            assert (fix, split) == klass and n_fit == int((rails <= cap).sum()) and (split == 0) == (n_fit == B), (fix, split, n_fit)
            if split:
                assert {bool(rails[b] <= cap) for b in picks} == {True, False}, rails[list(picks)]
        st, el = env.state()
        check_obs = t % 6 == 0 or t == steps - 1
        ob = {k: v.cpu().numpy() for k, v in o.items()} if check_obs else None
        if check_obs:       # the consumer's tensors: fl_policy_pack behind the launch, and the same written BY the launch (fl_obs_cutils_policy)
            _, _, adj, no, eo = env.policy_inputs(o)
            a32 = o["adjacency"].to(torch.int64)
            exp_adj = a32.clone()                       # Network.modify_adjacency (nn/net_tree.py:105-116) over the (env, agent)-flattened batch
            exp_adj[exp_adj == -2] = -B * A * env.max_nodes
            exp_adj[..., 0:2] += (torch.arange(B * A, device=a32.device) * env.max_nodes).view(B, A, 1, 1)
            exp_adj[exp_adj < 0] = -2
            assert torch.equal(adj, exp_adj) and torch.equal(no, o["node_order"].to(torch.int64)) and torch.equal(eo, o["edge_order"].to(torch.int64))
            keep = {k: o[k].clone() for k in ("agent_attr", "forest", "valid_actions", "props")}
            attr2, forest2, adj2, no2, eo2 = env.obs_policy()      # (idempotent on the state: the deadlock flags it sets were set by obs_cutils already)
            assert torch.equal(adj2, adj) and torch.equal(no2, no) and torch.equal(eo2, eo), f"step {t}: fl_obs_cutils_policy != fl_obs_cutils + fl_policy_pack"
            for k in keep:
                assert torch.equal(o[k].view(torch.uint8), keep[k].view(torch.uint8)), (t, k)
            assert env.last_obs_class()[:2] == klass
        for b, s_env in solo.items():
            s_rew, _, _ = s_env.step_synth(seed, b, 2 if t % 3 else 0, auto_reset=True)
            s_o = s_env.obs_cutils()
            _same(s_env.state()[0][0], st[b], f"replica {b} step {t} state")
            if check_obs:
                for key, _ in CUTILS:
                    _same(s_o[key].cpu().numpy()[0], ob[key][b], f"replica {b} step {t} {key}")
        for b, orc_env in oracles.items():
            if t % 3:
                dm = orc_env.distance_map()
                s_o = orc_env.state()
                acts = synth.spfollow_actions(seed, b, tc[b], s_o[:, 3], s_o[:, 0:2], s_o[:, 2], np.asarray(envs[b]["grid"]), *dm)
            else:
                acts = synth.uniform_actions(seed, b, tc[b], A)
            r_o, d_o, da = orc_env.step(acts)
            tc[b] += 1
            _same(st[b], orc_env.state(), f"oracle replica {b} step {t} state")
            exp = orc_env.obs_cutils(31, 500)
            if check_obs:
                for key, okey in CUTILS:
                    _same(ob[key][b], exp[okey], f"oracle replica {b} step {t} {key}")
            if da:
                key, pos = orc_env.get_rng()
                oracles[b] = orc.OracleEnv(envs[b])
                oracles[b].set_rng(key, pos)
                tc[b] = 0
    env.check()
    assert (env.state()[0][:, :, 0] >= 0).sum() > B      # trains are on the maps
    for s_env in solo.values():
        s_env.check()


def _cutils_digest(workload, B, distinct, steps, want_class):
    import hashlib
    from flatland_marl_amd import workload as wl
    from flatland_marl_amd.hip_backend import BatchedRailEnv
    envs, seed = wl.make_envs(workload, B=B, distinct=distinct)
    env = BatchedRailEnv(envs)
    h = hashlib.sha256()
    for t in range(steps):
        env.step_synth(seed, 0, 2, auto_reset=True)
        o = env.obs_cutils()
        assert env.last_obs_class()[:2] == tuple(want_class), (workload, env.last_obs_class(), want_class)
        h.update(env.state()[0].tobytes())
        for k in sorted(o):
            h.update(o[k].cpu().numpy().tobytes())
    env.check()
    return h.hexdigest()


CUTILS_CASES = {"cfg2": dict(workload="cfg2", B=16, distinct=4, steps=80), "cfg3": dict(workload="cfg3", B=24, distinct=3, steps=60),
                "cfg4": dict(workload="cfg4", B=12, distinct=4, steps=40), "cfg5": dict(workload="cfg5", B=4, distinct=2, steps=24)}
CUTILS_CLASS_OF = {"cfg2": (6, 0), "cfg3": (7, 0), "cfg4": (8, 0), "cfg5": (9, 2)}


def test_cutils_alone_other_launch_paths_give_the_same_bytes():
    """fl_obs_cutils three ways (switches read once per process, hence children): the default (one-pass kernels + classes 6 .. 9), the
    one-pass kernels with the runtime carving (FL_OBS_NO_FIX: MODE 6 / 7, MODE 0 at cfg5) and the stand-alone kernel every round before
    this one ran (FL_OBS_NO_CUTILS_MERGE: k_obs<0,*>, 32-lane pass A teams, no own-path filter).  Same bytes."""
    want = {k: _cutils_digest(want_class=CUTILS_CLASS_OF[k], **kw) for k, kw in CUTILS_CASES.items()}
    assert _child(("FL_OBS_NO_FIX",), "cutils", "nofix") == want
    assert _child(("FL_OBS_NO_CUTILS_MERGE",), "cutils", "alone") == want


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "cutils":
        for k, kw in CUTILS_CASES.items():
            # without the one-pass kernels (and the bins, which go with them) only the large-map class (MODE 0 anyway) still applies: its body
            # for the envs that fit, the runtime carving for the 3 025-cell level (split 1)
            klass = (0, 0) if sys.argv[2] == "nofix" else ((9, 1) if k == "cfg5" else (0, 0))
            print("DIGEST", k, _cutils_digest(want_class=klass, **kw))
    elif len(sys.argv) > 1 and sys.argv[1] == "paths":
        for k, kw in CASES.items():
            # rounds of 16 agents apply to envs of more than 32 agents only: cfg2 keeps its class there
            klass = CLASS_OF[k] if sys.argv[2] == "own" else (0, 0) if sys.argv[2] == "0" or k == "cfg3" else CLASS_OF[k]
            print("DIGEST", k, _cfg4_digest(want_class=klass, **kw))
    else:
        print("DIGEST cfg4", _cfg4_digest(want_class=(int(sys.argv[2]), int(sys.argv[3])), big=True))
