#!/usr/bin/env python3
"""Soak of the NON-BASELINE shapes: one level of every test of the Round-2 parameter table (Test_0 ... Test_14, fifteen shapes) as one
MixedBatch, stepped for hundreds of steps on the on-device SHORTEST-PATH-FOLLOWING action stream (kind 2: the large maps fill up,
agents arrive, queues and deadlocks form) and shadowed by the CPU oracle on every step: state, rewards, dones, the flatland_cutils
observation every step, the upstream depth-2 / depth-3 tree and the flatland_cutils builder alone every fourth each.  Round 6: every shape runs
a launch class (the BASELINE shapes their exact classes, the others bin classes).

  python tools/soak_round2.py [steps=320] [level=1] [out=profiles/r06_soak_round2.txt]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flatland_marl_amd import synth, workload as wl  # noqa: E402
from flatland_marl_amd.hip_backend import MixedBatch  # noqa: E402
from oracle import orc  # noqa: E402  (checker)

CUTILS = (("agent_attr", "attr"), ("forest", "forest"), ("adjacency", "adjacency"), ("node_order", "node_order"),
          ("edge_order", "edge_order"), ("valid_actions", "valid"), ("props", "props"))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 320
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    out = sys.argv[3] if len(sys.argv) > 3 else None
    tests = ["Test_%d" % k for k in range(15)]
    envs = [wl.generate_level(t, level) for t in tests]
    mb = MixedBatch(envs)
    oracles = [orc.OracleEnv(e) for e in envs]
    dms = [o.distance_map() for o in oracles]
    seed = 91
    n = len(envs)
    tc = [0] * n
    stat = [dict(max_on_map=0, arrived=0, max_deadlocked=0, episodes=0, conflicts=0) for _ in range(n)]
    t_start = time.time()
    cls_both, cls_alone = {}, None
    for t in range(steps):
        acts = []
        for i, o in enumerate(oracles):
            s = o.state()
            acts.append(synth.spfollow_actions(seed, mb.stream_of(i), tc[i], s[:, 3], s[:, 0:2], s[:, 2], np.asarray(envs[i]["grid"]), *dms[i]))
        res = mb.step_synth(seed, kind=2, auto_reset=True)
        depth = 3 if t % 4 == 1 else 2
        both = mb.obs_both(depth, 30)
        if t < 2:
            cls_both[depth] = [g.last_obs_class() for g in mb.groups]
        alone = mb.obs_cutils() if t % 4 == 2 else None       # the flatland_cutils builder alone: its own kernels and classes, same tensors
        if t == 2:
            cls_alone = [g.last_obs_class() for g in mb.groups]
        for i, o in enumerate(oracles):
            r_o, d_o, da = o.step(acts[i])
            tc[i] += 1
            rew, done, done_all = mb.pick(i, res)
            cut, tree = mb.pick(i, both)
            assert np.array_equal(rew.cpu().numpy(), r_o) and np.array_equal(done.cpu().numpy(), d_o), (tests[i], t, "reward / done")
            st, el = mb.state(i)
            assert np.array_equal(st, o.state()), (tests[i], t, "state")
            exp = o.obs_cutils(31, 500)
            for key, okey in CUTILS:
                assert np.array_equal(cut[key].cpu().numpy(), exp[okey], equal_nan=True), (tests[i], t, key)
                if alone is not None:
                    assert np.array_equal(mb.pick(i, alone)[key].cpu().numpy(), exp[okey], equal_nan=True), (tests[i], t, key, "builder alone")
            if t % 4 in (1, 3):
                assert np.array_equal(tree.cpu().numpy(), o.obs_pytree(depth, 30), equal_nan=True), (tests[i], t, "depth-%d tree" % depth)
            s = stat[i]
            s["max_on_map"] = max(s["max_on_map"], int((st[:, 0] >= 0).sum()))
            s["max_deadlocked"] = max(s["max_deadlocked"], int(exp["props"][:, 1].sum()))
            s["conflicts"] += int((exp["forest"][:, :, 3] >= 0).sum())       # nodes with a potential conflict
            if da:
                s["arrived"] += int((st[:, 3] == 6).sum())
                s["episodes"] += 1
                key, pos = o.get_rng()
                oracles[i] = orc.OracleEnv(envs[i])
                oracles[i].set_rng(key, pos)
                tc[i] = 0
        if t % 40 == 39:
            print("step %d / %d ok (%.0f s)" % (t + 1, steps, time.time() - t_start), flush=True)
    mb.check()
    lines = ["soak of the Round-2 table, level %d of every test: %d steps of shortest-path-following actions, every env shadowed by the oracle on every step "
             "(state, rewards, dones, cutils observation; depth-2 / depth-3 tree and the builder alone every 4th each): all equal" % (level, steps),
             "%-8s %5s %9s %6s %6s %11s %8s %10s %9s %s" % ("test", "agents", "grid", "rails", "on-map", "deadlocked", "arrived", "episodes", "conflicts", "launch class (class, split, envs on it): both builders depth 2 | depth 3 | the builder alone")]
    for i, e in enumerate(envs):
        g, b = mb.where[i]
        H, W = np.asarray(e["grid"]).shape
        s = stat[i]
        lines.append("%-8s %5d %9s %6d %6d %11d %8d %10d %9d %s" % (tests[i], len(e["init_dir"]), "%dx%d" % (H, W), int((np.asarray(e["grid"]) != 0).sum()),
                                                                  s["max_on_map"], s["max_deadlocked"], s["arrived"], s["episodes"], s["conflicts"],
                                                                  "%s | %s | %s" % (cls_both[2][g], cls_both[3][g], cls_alone[g])))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out if os.path.isabs(out) else os.path.join(ROOT, out), "w").write(txt + "\n")
    mb.close()


if __name__ == "__main__":
    main()
