"""CPU, build container only (needs the reference mounted at /root/reference): from_reference_env on a REAL reference
RailEnv equals the capture script's extraction, i.e. the committed golden static arrays; the dynamic-state extraction equals
the golden per-step state."""
import os
import sys

import numpy as np
import pytest

from tests import util

REF = "/root/reference/flatland-rl"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF) or not os.path.exists(os.path.join(util.ROOT, "oracle", "_ref")),
                                reason="the reference is only mounted in the build container")


@pytest.fixture(scope="module")
def cap():
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(util.ROOT, "oracle", "refharness"))
    import capture_golden   # puts the stubs, the reference and oracle/_ref on sys.path
    return capture_golden


def test_from_reference_env_equals_golden_static(cap):
    from flatland_marl_amd import from_reference_env, dynamic_state_of_reference_env, synth
    fx = util.load("cfg1_uniform")
    env, mp = cap.make_env(cap.csv_row("Test_0", "Level_0"))
    env.reset()
    got = from_reference_env(env)
    exp = cap.static_arrays(env, mp)
    assert sorted(got) == sorted(exp)
    for k in exp:
        np.testing.assert_array_equal(got[k], exp[k], err_msg=k)
        np.testing.assert_array_equal(got[k], fx[k], err_msg="golden " + k)
    # a few steps of the golden stream: the dynamic state read from the reference objects equals the fixture
    for t in range(25):
        a = synth.uniform_actions(1, 0, t, env.get_num_agents())
        env.step({i: int(a[i]) for i in range(len(a))})
        st, aux, el, da = dynamic_state_of_reference_env(env)
        np.testing.assert_array_equal(st, util.golden_state(fx, t), err_msg=f"step {t}")
        assert el == t + 1 and not da


class _Recorder:
    """stands in for the device batch in the build container (no GPU): records what the plug-in hands to the C-ABI -- the static
    description (fl_load_env) and every injected state (fl_set_state).  It computes nothing."""
    made = []

    def __init__(self, envs, device=0, max_nodes=31, pred_depth=500):
        import torch
        self.static = envs[0]
        self.H, self.W = np.asarray(envs[0]["grid"]).shape
        self.A = len(envs[0]["init_dir"])
        self.max_nodes, self.pred_depth = max_nodes, pred_depth
        self.pushed = []
        self._t = torch
        _Recorder.made.append(self)

    def replace_env(self, b, st):
        self.static = st

    def set_state(self, state, aux, elapsed):
        self.pushed.append((state[0].copy(), aux[0].copy(), int(elapsed[0])))

    def obs_cutils(self, handles=None):
        t, A, N = self._t, self.A, self.max_nodes
        return dict(agent_attr=t.zeros(1, A, 83), forest=t.zeros(1, A, N, 12), adjacency=t.zeros(1, A, N - 1, 3, dtype=t.int32),
                    node_order=t.zeros(1, A, N, dtype=t.int32), edge_order=t.zeros(1, A, N - 1, dtype=t.int32),
                    valid_actions=t.zeros(1, A, 5, dtype=t.uint8), props=t.zeros(1, A, 3, dtype=t.float64))

    def check(self):
        pass

    def close(self):
        pass


def test_plugin_inside_the_real_reference_railenv(cap, monkeypatch):
    """RailEnv(obs_builder_object=plugin) of the REAL reference: the constructor's set_env (no rail yet), reset() -> set_env +
    reset + get_many, step() -> get_many (rail_env.py:178-179, 305, 346, 634, 660-666).  What the plug-in extracts from the
    reference's objects for the C-ABI equals the golden static arrays and the golden per-step states."""
    from flatland_marl_amd import plugin, synth
    monkeypatch.setattr(plugin._EnvBinding, "make_batch", _Recorder)
    _Recorder.made.clear()
    fx = util.load("cfg1_malf50")
    builder = plugin.TreeObsForRailEnv(31, 500)
    env, mp = cap.make_env(cap.csv_row("Test_0", "Level_1"), malfunction_interval=50, obs=builder)
    assert builder.env is env and not _Recorder.made          # set_env in the constructor reads nothing
    obs, info = env.reset()
    rec, = _Recorder.made
    for k in ("grid", "init_pos", "init_dir", "target", "speed", "earliest", "latest", "T"):
        np.testing.assert_array_equal(rec.static[k], fx[k], err_msg=k)
    A = env.get_num_agents()
    attr, (nodes, adj, no, eo) = obs
    assert (len(attr), len(attr[0]), len(nodes[0]), len(nodes[0][0]), len(adj[0]), len(no[0]), len(eo[0])) == (A, 83, 31, 12, 30, 31, 30)
    # reset() runs AgentsLoader::update itself (treeobs.cpp:22-28), then RailEnv.reset() asks for the observations: two reads, both at step 0
    assert len(rec.pushed) == 2 and rec.pushed[0][2] == 0 and rec.pushed[1][2] == 0
    cfg, props, valid = builder.get_properties()
    assert cfg == dict(curr_step=0, n_agents=A, max_timesteps=int(fx["T"]), height=30, width=30) and len(valid) == A
    cols = [i for i, n in enumerate(util.STATE_NAMES) if n != "saved"]       # flatland_cutils does not read the saved action
    malf_seen = 0
    for t in range(120):
        a = synth.uniform_actions(4, 0, t, A)
        env.step({i: int(a[i]) for i in range(A)})
        st, aux, el = rec.pushed[-1]
        assert el == t + 1 and len(rec.pushed) == t + 3
        np.testing.assert_array_equal(st[:, cols], util.golden_state(fx, t)[:, cols], err_msg=f"step {t}")
        np.testing.assert_array_equal(st[:, 7], fx["s_saved"][t], err_msg=f"saved action, step {t}")
        # the in_malfunction signal the reference module reads is the state machine's, as of this step (loader.cpp:16-18)
        exp_sig = [int(bool(ag.state_machine.st_signals.in_malfunction)) for ag in env.agents]
        assert aux[:, 1].tolist() == exp_sig
        malf_seen += int(sum(exp_sig))
        assert builder.get_properties()[0]["curr_step"] == t + 1
    assert malf_seen > 0
    # reset(regenerate_rail=True) on the same builder: the static side follows the new map
    env.reset(random_seed=7)
    assert not np.array_equal(_Recorder.made[-1].static["grid"], fx["grid"]) or len(_Recorder.made) > 1


def test_upstream_plugin_inside_the_real_reference_railenv(cap, monkeypatch):
    from flatland_marl_amd import plugin
    from flatland.envs.predictions import ShortestPathPredictorForRailEnv
    monkeypatch.setattr(plugin._EnvBinding, "make_batch", _Recorder)
    _Recorder.made.clear()
    import torch
    A_box = []
    _Recorder.obs_tree = lambda self, d, p, handles=None: A_box.append((d, p)) or torch.full((1, self.A, (4 ** (d + 1) - 1) // 3, 12), -np.inf, dtype=torch.float64)
    try:
        builder = plugin.TreeObsUpstream(2, ShortestPathPredictorForRailEnv(30))
        env, mp = cap.make_env(cap.csv_row("Test_0", "Level_0"), obs=builder)
        obs, _ = env.reset()
        assert builder.predictor.env is env and A_box == [(2, 30)]
        assert sorted(obs) == list(range(env.get_num_agents()))
        fx = util.load("cfg1_uniform")
        np.testing.assert_array_equal(_Recorder.made[-1].static["grid"], fx["grid"])
    finally:
        del _Recorder.obs_tree


def test_committed_goldens_are_what_the_reference_produces_here(cap):
    """The pin, verified: two small capture jobs re-run on the REAL reference into a temporary directory and compared with the
    committed fixtures, every array bit for bit and no key missing on either side (oracle/refharness/capture_golden.py --check
    does the same for all jobs; the whole set takes ~25 minutes of reference CPU time)."""
    assert cap.check(["cfg1_sparse", "cfg1_malf20_spfollow"], verbose=False) == []
