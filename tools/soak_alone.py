"""Long differential run of the CONSUMER's path: B replicas of a workload on the GPU -- k_step on the dense shortest-path-following stream (auto-reset, a high
malfunction rate) + fl_obs_cutils_policy (the flatland_cutils builder alone writing the policy's int64 tensors) --, N replicas shadowed by the CPU oracle and
compared bit for bit on every step: state, rewards, dones, attribute rows, forest, valid actions, properties, and the int64 adjacency / orders against
Network.modify_adjacency of the oracle's int32 tensors (solution/nn/net_tree.py:105-116).

  python tools/soak_alone.py [workload=cfg2] [B=256] [steps=600] [shadowed=4] [malfunction rate=1/150] [distinct maps=0]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flatland_marl_amd import synth, workload as wl  # noqa: E402
from flatland_marl_amd.hip_backend import BatchedRailEnv  # noqa: E402
from oracle import orc  # noqa: E402  (checker)

workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 600
shadow = int(sys.argv[4]) if len(sys.argv) > 4 else 4
rate = float(sys.argv[5]) if len(sys.argv) > 5 else 1 / 150.0
distinct = int(sys.argv[6]) if len(sys.argv) > 6 else 0
envs, seed = wl.make_envs(workload, B=B, distinct=distinct)
for e in envs:
    e["malf_rate"] = rate
env = BatchedRailEnv(envs, device=0)
A, N = env.A, env.max_nodes
picks = sorted({int(round(k * (B - 1) / max(shadow - 1, 1))) for k in range(shadow)})
oracles = {b: orc.OracleEnv(envs[b]) for b in picks}
dms = {b: o.distance_map() for b, o in oracles.items()}
tc = {b: 0 for b in picks}
episodes, conflicts, klass = 0, 0, None
for it in range(steps):
    kind = 2 if it % 50 < 40 else 0
    rew, done, done_all = env.step_synth(seed, 0, kind, auto_reset=True)
    attr, forest, adj, no, eo = env.obs_policy()
    if klass is None:
        klass = env.last_obs_class()
    o = env._obs
    rew, done = rew.cpu().numpy(), done.cpu().numpy()
    got = dict(attr=attr.cpu().numpy(), forest=forest.cpu().numpy(), adj=adj.cpu().numpy(), no=no.cpu().numpy(), eo=eo.cpu().numpy(),
               valid=o["valid_actions"].cpu().numpy(), props=o["props"].cpu().numpy())
    state = env.state()[0]
    for b, orc_env in oracles.items():
        if kind == 2:
            s = orc_env.state()
            acts = synth.spfollow_actions(seed, b, tc[b], s[:, 3], s[:, 0:2], s[:, 2], np.asarray(envs[b]["grid"]), *dms[b])
        else:
            acts = synth.uniform_actions(seed, b, tc[b], A)
        r_o, d_o, da = orc_env.step(acts)
        tc[b] += 1
        assert np.array_equal(rew[b], r_o) and np.array_equal(done[b], d_o), (it, b, "reward/done")
        assert np.array_equal(state[b], orc_env.state()), (it, b, "state")
        exp = orc_env.obs_cutils(31, 500)
        for k, ek in (("attr", "attr"), ("forest", "forest"), ("valid", "valid"), ("props", "props")):
            assert np.array_equal(got[k][b], exp[ek], equal_nan=True), (it, b, k)
        a64 = exp["adjacency"].astype(np.int64)
        a64[a64 == -2] = -B * A * N
        a64[..., 0:2] += (np.arange(A, dtype=np.int64) + b * A).reshape(A, 1, 1) * N
        a64[a64 < 0] = -2
        assert np.array_equal(got["adj"][b], a64), (it, b, "adjacency (int64, modified)")
        assert np.array_equal(got["no"][b], exp["node_order"].astype(np.int64)) and np.array_equal(got["eo"][b], exp["edge_order"].astype(np.int64)), (it, b, "orders")
        conflicts += int((exp["forest"][:, :, 3] >= 0).sum())
        if da:
            episodes += 1
            key, pos = orc_env.get_rng()
            oracles[b] = orc.OracleEnv(envs[b])
            oracles[b].set_rng(key, pos)
            tc[b] = 0
    if it % 100 == 99:
        print("step %d / %d ok" % (it + 1, steps), flush=True)
env.check()
print("soak_alone %s: B=%d, %d steps, replicas %s shadowed, launch class %s, %d episodes ended, %d tree nodes with a potential conflict: all equal"
      % (workload, B, steps, picks, klass, episodes, conflicts))
