"""GPU parity of the live map replacement and the masked distance-map rebuild (RailEnv.reset(regenerate_rail=True, ...),
rail_env.py:288-320, DistanceMap.reset() + _compute(), distance_map.py:47-79): an env of a RUNNING batch gets a different
map / schedule / RNG state and matches the oracle from there, while the untouched envs keep matching theirs."""
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


def _rng(b):
    st = np.random.RandomState([b]).get_state()
    return np.array(st[1], dtype=np.uint32), int(st[2])


def _compare(env, oracles, tag, tree=(2, 30)):
    o = {k: v.cpu().numpy() for k, v in env.obs_cutils().items()}
    tr = env.obs_tree(*tree).cpu().numpy()
    st = env.state()[0]
    for b, oe in enumerate(oracles):
        _same(st[b], oe.state(), f"{tag} env {b} state")
        exp = oe.obs_cutils(31, 500)
        for got, key in (("agent_attr", "attr"), ("forest", "forest"), ("adjacency", "adjacency"), ("node_order", "node_order"),
                         ("edge_order", "edge_order"), ("valid_actions", "valid"), ("props", "props")):
            _same(o[got][b], exp[key], f"{tag} env {b} {got}")
        _same(tr[b], oe.obs_pytree(*tree), f"{tag} env {b} tree")


def test_env_of_a_live_batch_is_replaced_by_another_map_and_matches_the_oracle_from_there():
    from oracle import orc
    from flatland_marl_amd import synth
    from flatland_marl_amd.hip_backend import FlatlandHipError
    bases = [util.load("base_cfg2_L%d" % k) for k in range(1, 8)]
    U = max(len(fx["dm_targets"]) for fx in bases)
    R = max(int((fx["grid"] != 0).sum()) for fx in bases)
    envs = []
    for b in range(4):
        key, pos = _rng(700 + b)
        envs.append(util.static_of(bases[b], key, pos))
    env = _env(envs, reserve=(U, R))
    oracles = [orc.OracleEnv(e) for e in envs]
    A = env.A
    tc = [0] * 4
    seed = 31

    def run(n, tag):
        for it in range(n):
            rew, done, done_all = env.step_synth(seed, 50, 1, auto_reset=True)
            rew, done, done_all = rew.cpu().numpy(), done.cpu().numpy(), done_all.cpu().numpy()
            for b, oe in enumerate(oracles):
                r_o, d_o, da = oe.step(synth.forward_biased_actions(seed, 50 + b, tc[b], A))
                tc[b] += 1
                _same(rew[b], r_o, f"{tag} it {it} env {b} rewards")
                _same(done[b], d_o, f"{tag} it {it} env {b} dones")
                assert bool(done_all[b]) == da
                if da:
                    key, pos = oe.get_rng()
                    oracles[b] = orc.OracleEnv(envs[b])
                    oracles[b].set_rng(key, pos)
                    tc[b] = 0
            _compare(env, oracles, f"{tag} it {it}")

    run(60, "before")
    # replace env 1 (another map, other agents, fresh RNG) and env 3 in ONE commit; envs 0 and 2 keep running
    for b, src in ((1, 5), (3, 6)):
        key, pos = _rng(900 + b)
        envs[b] = util.static_of(bases[src], key, pos)
        env.replace_env(b, envs[b], commit=False)
        oracles[b] = orc.OracleEnv(envs[b])
        tc[b] = 0
    env.commit()
    dm, slot = env.distance_map(1)
    _same(dm, bases[5]["dm_u16"], "distance map of the replaced env")
    _same(slot, bases[5]["target_slot"], "target slots of the replaced env")
    _same(env.distance_map(0)[0], bases[0]["dm_u16"], "distance map of an untouched env")
    assert env.state()[1].tolist()[1] == 0 and env.state()[1].tolist()[0] == 60
    run(120, "after")
    env.check()
    # an env that does not fit the reserved capacity is refused and leaves the batch as it was
    big = dict(envs[0])
    big["grid"] = np.where(np.asarray(big["grid"]) == 0, np.uint16(0x8020), big["grid"]).astype(np.uint16)   # rail everywhere
    with pytest.raises(FlatlandHipError, match="FL_ERR_CAPACITY"):
        env.replace_env(0, big)
    run(10, "after refused load")
    env.check()


def test_replacement_without_reserve_is_limited_to_the_first_commit_sizes():
    from flatland_marl_amd.hip_backend import FlatlandHipError
    small, large = util.load("cfg1_uniform"), None
    st = util.static_of(small)
    env = _env([st, st])
    env.replace_env(1, st)       # same size: fine
    env.check()
    more = dict(st)
    g = np.array(more["grid"], dtype=np.uint16)
    free = np.argwhere(g == 0)
    g[tuple(free[0])] = 0x8020   # one more rail cell than the batch was committed for
    more["grid"] = g
    with pytest.raises(FlatlandHipError, match="FL_ERR_CAPACITY"):
        env.replace_env(0, more)


def test_masked_rebuild_touches_only_the_masked_envs_and_reset_takes_a_device_mask():
    import torch
    fx = util.load("cfg4_fwd_head")
    st = util.static_of(fx)
    env = _env([st, st, st])
    for _ in range(30):
        env.step_synth(3, 0, 1, auto_reset=True)
    mask = torch.tensor([0, 1, 0], dtype=torch.uint8, device="cuda")
    env.rebuild_distance_maps(mask)
    for b in range(3):
        _same(env.distance_map(b)[0], fx["dm_u16"], f"env {b} distance map")
    s0 = env.state()[0].copy()
    env.reset(mask)                               # device mask: only env 1 starts over
    s1, el = env.state()
    _same(s1[0], s0[0], "env 0 untouched"); _same(s1[2], s0[2], "env 2 untouched")
    assert el.tolist() == [30, 0, 30] and (s1[1][:, 3] == 0).all()
    e2 = _env([st])
    for t in range(40):
        env.step_synth(3, 0, 1, auto_reset=True)
    for t in range(40):
        e2.step_synth(3, 1, 1, auto_reset=True)   # env 1's stream, from its fresh state
    # the RNG of env 1 kept running through the reset, so only the RNG-independent part of the state is comparable here:
    # with malfunction rate 1/7200 the two runs differ only if a malfunction fired; compare the non-malfunctioning agents
    a, b = env.state()[0][1], e2.state()[0][0]
    same = (a[:, 5] == 0) & (b[:, 5] == 0)
    _same(a[same][:, :4], b[same][:, :4], "env 1 after the masked reset")
    env.check()


def test_envs_of_one_map_share_their_static_tables_through_replacements():
    """Envs with the same rail grid and unique targets read ONE set of device tables, the slabs of the first of them (FlDev::tab).
    Six envs on two maps (own RNG streams); the owner of a map's tables is replaced by a third map while its dependents keep
    running (one of them becomes the owner: its slabs are built at that commit), an env joins an existing map, and the masked
    distance-map rebuild goes through the owners.  Every env matches its oracle throughout.  (FL_NO_SHARED_TABLES -- one set of
    tables per env -- gives the same bytes: tests/test_gpu_fullsize.py::test_other_launch_paths_give_the_same_bytes.)"""
    import torch
    from oracle import orc
    from flatland_marl_amd import synth
    bases = [util.load("base_cfg2_L%d" % k) for k in range(1, 8)]
    U = max(len(fx["dm_targets"]) for fx in bases)
    R = max(int((fx["grid"] != 0).sum()) for fx in bases)
    which = [0, 1, 0, 1, 0, 1]
    envs = []
    for b, m in enumerate(which):
        key, pos = _rng(1700 + b)
        envs.append(util.static_of(bases[m], key, pos))
    env = _env(envs, reserve=(U, R))
    oracles = [orc.OracleEnv(e) for e in envs]
    A, seed = env.A, 47
    tc = [0] * len(envs)

    def run(n, tag, rebuild=False):
        for it in range(n):
            rew, done, done_all = env.step_synth(seed, 70, 1, auto_reset=True)
            if rebuild:
                env.rebuild_distance_maps(env.done_all)
            rew, done_all = rew.cpu().numpy(), done_all.cpu().numpy()
            for b, oe in enumerate(oracles):
                r_o, d_o, da = oe.step(synth.forward_biased_actions(seed, 70 + b, tc[b], A))
                tc[b] += 1
                _same(rew[b], r_o, f"{tag} it {it} env {b} rewards")
                if da:
                    key, pos = oe.get_rng()
                    oracles[b] = orc.OracleEnv(envs[b])
                    oracles[b].set_rng(key, pos)
                    tc[b] = 0
            if it % 5 == 4 or it == n - 1:
                _compare(env, oracles, f"{tag} it {it}")

    def replace(b, m, k):
        key, pos = _rng(k)
        envs[b] = util.static_of(bases[m], key, pos)
        env.replace_env(b, envs[b])
        oracles[b] = orc.OracleEnv(envs[b])
        tc[b] = 0

    run(40, "shared")
    replace(0, 2, 1800)          # the owner of map 0's tables gets map 2: env 2 now owns map 0's (never built so far)
    run(40, "owner replaced")
    for b in (2, 4):
        _same(env.distance_map(b)[0], bases[0]["dm_u16"], f"distance map of dependent {b}")
    replace(3, 2, 1801)          # env 3 joins map 2 (owner: env 0)
    replace(2, 1, 1802)          # the new owner of map 0 leaves too: env 4 is the last one on it
    run(40, "joined", rebuild=True)
    _same(env.distance_map(4)[0], bases[0]["dm_u16"], "distance map of the last env on map 0")
    _same(env.distance_map(3)[0], bases[2]["dm_u16"], "distance map of the env that joined map 2")
    env.rebuild_distance_maps(torch.tensor([0, 0, 0, 1, 1, 0], dtype=torch.uint8, device="cuda"))
    env.rebuild_distance_maps()
    run(20, "after rebuilds")
    env.check()
