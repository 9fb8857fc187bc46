"""GPU parity (through the C-ABI): HIP step / distance-map kernels vs the golden vectors from the real
reference and vs the CPU oracle on seeded inputs.  Bit-exact (integer work)."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _env(envs, **kw):
    from flatland_marl_amd.hip_backend import BatchedRailEnv
    return BatchedRailEnv(envs, **kw)


@pytest.mark.parametrize("name", util.episode_fixtures())
def test_step_matches_reference_golden(name):
    import torch
    fx = util.load(name)
    env = _env([util.static_of(fx)])
    acts = util.actions_of(fx)
    for t in range(len(acts)):
        rew, done, done_all = env.step(torch.from_numpy(acts[t][None, :].copy()).cuda())
        st, el = env.state()
        np.testing.assert_array_equal(st[0], util.golden_state(fx, t), err_msg=f"{name} step {t}")
        np.testing.assert_array_equal(rew.cpu().numpy()[0], fx["s_reward"][t], err_msg=f"{name} reward step {t}")
        np.testing.assert_array_equal(done.cpu().numpy()[0], fx["s_done"][t], err_msg=f"{name} done step {t}")
        assert bool(done_all.cpu().numpy()[0]) == bool(fx["done_all"][t])
        assert el[0] == t + 1
    env.check()


@pytest.mark.parametrize("name", util.episode_fixtures() + util.base_fixtures())
def test_distance_map_matches_reference_golden(name):
    fx = util.load(name)
    env = _env([util.static_of(fx)])
    dm, slot = env.distance_map(0)
    np.testing.assert_array_equal(slot, fx["target_slot"])
    np.testing.assert_array_equal(dm, fx["dm_u16"])


def test_step_after_done_raises_like_reference():
    import torch
    from flatland_marl_amd.hip_backend import EpisodeDoneError
    fx = util.load("cfg1_spfollow")
    env = _env([util.static_of(fx)])
    for a in util.actions_of(fx):
        env.step(torch.from_numpy(a[None, :].copy()).cuda())
    env.check()
    env.step(torch.from_numpy(fx["actions"][0][None, :].copy()).cuda())
    with pytest.raises(EpisodeDoneError, match="Episode is done"):
        env.check()


def _replica_rng(b):
    rs = np.random.RandomState([b])
    st = rs.get_state()
    return np.array(st[1], dtype=np.uint32), int(st[2])


@pytest.mark.parametrize("bases,B,steps,malf_rate", [
    (["base_cfg2_L1", "base_cfg2_L2", "base_cfg2_L3"], 24, 450, None),
    (["base_cfg2_L4", "base_cfg2_L5"], 16, 400, 1 / 25.0),     # heavy malfunctions: RNG replay path
    (["base_cfg3_L1", "base_cfg3_L2"], 6, 420, 1 / 60.0),
])
def test_batched_step_matches_oracle_with_synth_stream_and_autoreset(bases, B, steps, malf_rate):
    """many replicas, on-device synthetic action stream, per-replica MT19937 seeds, auto-reset at episode end:
    every replica must match the scalar oracle stepped with the same stream."""
    from oracle import orc
    from flatland_marl_amd import synth
    fxs = [util.load(n) for n in bases]
    envs, oracles = [], []
    for b in range(B):
        fx = fxs[b % len(fxs)]
        key, pos = _replica_rng(b)
        st = util.static_of(fx, key, pos)
        if malf_rate is not None:
            st["malf_rate"] = malf_rate
        envs.append(st)
        oracles.append(orc.OracleEnv(st))
    env = _env(envs)
    A = env.A
    seed = 99
    tcount = np.zeros(B, dtype=np.int64)
    for it in range(steps):
        rew, done, done_all = env.step_synth(seed, stream_base=1000, kind=0, auto_reset=True)
        st, el = env.state()
        rew, done, done_all = rew.cpu().numpy(), done.cpu().numpy(), done_all.cpu().numpy()
        for b in range(B):
            o = oracles[b]
            a = synth.uniform_actions(seed, 1000 + b, int(tcount[b]), A)
            r_o, d_o, da_o = o.step(a)
            tcount[b] += 1
            np.testing.assert_array_equal(st[b], o.state(), err_msg=f"replica {b} iter {it}")
            np.testing.assert_array_equal(rew[b], r_o, err_msg=f"replica {b} iter {it} reward")
            np.testing.assert_array_equal(done[b], d_o)
            assert bool(done_all[b]) == da_o
            if da_o:  # oracle side of the auto-reset: fresh agents, RNG keeps running
                key, pos = o.get_rng()
                oracles[b] = orc.OracleEnv(envs[b])
                oracles[b].set_rng(key, pos)
                tcount[b] = 0
    env.check()
    key, pos = env.rng_state()
    for b in range(B):
        k_o, p_o = oracles[b].get_rng()
        assert pos[b] == p_o
        np.testing.assert_array_equal(key[b], k_o)
