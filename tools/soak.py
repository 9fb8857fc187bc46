"""Long differential run: B replicas of a workload's base envs on the GPU (fused observation launch, auto-reset, high
malfunction rate), the first N replicas shadowed by the CPU oracle and compared bit for bit on every step.

  python tools/soak.py [workload] [steps] [shadowed replicas] [malfunction rate]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatland_marl_amd import synth, workload as wl  # noqa: E402
from flatland_marl_amd.hip_backend import BatchedRailEnv, malf_threshold  # noqa: E402
from oracle import orc  # noqa: E402  (checker)

workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
shadow = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rate = float(sys.argv[4]) if len(sys.argv) > 4 else 1 / 150.0
B = max(64, shadow)
envs, seed = wl.make_envs(workload, B=B)
for e in envs:
    e["malf_rate"] = rate
env = BatchedRailEnv(envs, device=0)
A = env.A
oracles = [orc.OracleEnv(envs[b]) for b in range(shadow)]
tc = [0] * shadow
keys = (("agent_attr", "attr"), ("forest", "forest"), ("adjacency", "adjacency"), ("node_order", "node_order"),
        ("edge_order", "edge_order"), ("valid_actions", "valid"), ("props", "props"))
episodes = 0
for it in range(steps):
    kind = (it // 400) % 2
    explicit = (it // 250) % 3 == 2          # every third block: explicit action tensors incl. absent / illegal values
    if explicit:
        rs = np.random.RandomState(it)
        acts = rs.choice(np.array([0, 1, 2, 3, 4, 5, 9, 255], dtype=np.uint8), size=(B, A),
                         p=[0.1, 0.15, 0.4, 0.15, 0.1, 0.03, 0.02, 0.05]).astype(np.uint8)
        filt = (it // 250) % 2 == 0
        rew, done, done_all, o, tree = env.step_obs(acts, auto_reset=True, filter_required=filt, tree_depth=2, tree_pred=30)
    else:
        rew, done, done_all = env.step_synth(seed, 0, kind, auto_reset=True)
        o, tree = env.obs_both(2, 30)
    rew, done = rew.cpu().numpy(), done.cpu().numpy()
    on = {k: v.cpu().numpy() for k, v in o.items()}
    tr = tree.cpu().numpy()
    state = env.state()[0]
    for b in range(shadow):
        fn = synth.forward_biased_actions if kind == 1 else synth.uniform_actions
        if explicit:
            a_b = acts[b].copy()
            if filt:      # eval_env.parse_actions: drop the actions of agents without action_required
                st = oracles[b].state()   # columns: row, col, dir, state, malf, nmalf, speed counter, ... (rail_env.py:243-258)
                required = (st[:, 3] == 1) | ((st[:, 3] >= 3) & (st[:, 3] <= 5) & (st[:, 6] == 0))
                a_b[~required] = 255
            r_o, d_o, da = oracles[b].step(a_b)
        else:
            r_o, d_o, da = oracles[b].step(fn(seed, b, tc[b], A))
        tc[b] += 1
        assert np.array_equal(rew[b], r_o) and np.array_equal(done[b], d_o), (it, b, "reward/done")
        assert np.array_equal(state[b], oracles[b].state()), (it, b, "state")
        exp = oracles[b].obs_cutils(31, 500)
        for g, k in keys:
            assert np.array_equal(on[g][b], exp[k], equal_nan=True), (it, b, g)
        assert np.array_equal(tr[b], oracles[b].obs_pytree(2, 30), equal_nan=True), (it, b, "tree")
        if da:
            key, pos = oracles[b].get_rng()
            oracles[b] = orc.OracleEnv(envs[b])
            oracles[b].set_rng(key, pos)
            tc[b] = 0
            episodes += 1
env.check()
print("soak ok: %s, %d steps, %d shadowed replicas of %d, %d shadowed episodes, malfunction rate %.4f" %
      (workload, steps, shadow, B, episodes, rate))
