#!/usr/bin/env python3
"""Golden vectors of `flatland_cutils.TreeObsForRailEnv.get_many(handles)` with a STRICT SUBSET of the handles (treeobs.cpp:50-62:
predicted_pos / predicted_dir then hold one entry per listed handle, in list order; the conflict test works on list positions).
Only lists that are a permutation of 0 .. n-1 are defined behaviour in the reference (get_possible_conflicting erases position
`agent.handle` of the list, tool.h:428-434); the real reference is run here on a dense shortest-path-following episode and asked,
at a few steps, for several such lists -> tests/golden/subset_cfg2.npz (agent states at those steps + handles + the returned forest).
Round 6: the UPSTREAM builder's get_many(handles) on the same lists at the same steps (observations.py:60-115: predicted_pos / predicted_dir
hold the listed handles' predictions in list order, the conflict test deletes list position `handle` and reads env.agents[position].state,
:337-366) -> pytree_<k> f64[snapshots, len(handles), 21, 12] (depth 2, predictor depth 30), rows in list order."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import capture_golden as cg  # noqa: E402  (sets up the reference's import path)


def main():
    row = cg.csv_row("Test_2", "Level_5")
    env, mp = cg.make_env(row, malfunction_interval=60)
    obs, _ = env.reset()
    A = env.get_num_agents()
    out = cg.static_arrays(env, mp)
    out.update(cg.dm_unique(env))
    lists = [list(range(A)), list(range(A // 2)), list(range(7))[::-1], [2, 0, 1], [0], [3, 1, 4, 0, 2], list(range(A - 1))]
    rng = np.random.default_rng(seed := 83)
    perm = rng.permutation(12).tolist()
    lists.append(perm)
    snaps, steps, actions = [], [], []
    up = cg.PyTreeObs(max_depth=2, predictor=cg.ShortestPathPredictorForRailEnv(30))
    up.set_env(env)
    up.reset()
    pytrees = {k: [] for k in range(len(lists))}
    t = 0
    forests = {k: [] for k in range(len(lists))}
    adjs = {k: [] for k in range(len(lists))}
    attrs = {k: [] for k in range(len(lists))}
    T = env._max_episode_steps
    while t < min(T, 330) and not env.dones["__all__"]:
        acts = cg.sp_follow_actions(env, rng)
        row_a = np.full(A, 255, dtype=np.uint8)
        for i, a in acts.items():
            row_a[i] = a
        actions.append(row_a)
        env.step(acts)
        t += 1
        if t % 55 == 0:
            snap = cg.agent_snapshot(env)
            snaps.append(np.stack([snap[k] for k in ("row", "col", "dir", "state", "malf", "nmalf", "scount", "saved", "arrival", "old_row", "old_col", "old_dir")], axis=1))
            steps.append(t)
            # the in_malfunction signal of the last step (loader.cpp:16-18) and the sticky deadlock flags BEFORE these calls
            out.setdefault("sig", []).append(np.array([int(bool(ag.state_machine.st_signals.in_malfunction)) for ag in env.agents], dtype=np.int32))
            for k, hs in enumerate(lists):
                attr, (nodes, adj, no, eo) = env.obs_builder.get_many(hs)
                forests[k].append(np.array(nodes, dtype=np.float32))
                adjs[k].append(np.array(adj, dtype=np.int32))
                attrs[k].append(np.array(attr, dtype=np.float32))
                got = up.get_many(hs)                 # (pure reads of the env: the cutils builder's sticky flags are not involved)
                rows = []
                for h in hs:
                    o = []
                    cg.flatten_pytree(got[h], 0, 2, o)
                    rows.append(o)
                pytrees[k].append(np.array(rows, dtype=np.float64))
            _, props, _ = env.obs_builder.get_properties()
            out.setdefault("deadlocked", []).append(np.array(props["deadlocked"], dtype=np.int32))
    out["sig"] = np.stack(out["sig"]); out["deadlocked"] = np.stack(out["deadlocked"])
    out["actions"] = np.stack(actions)
    out["snap_steps"] = np.array(steps, dtype=np.int32)
    out["snaps"] = np.stack(snaps).astype(np.int32)
    out["n_lists"] = np.int32(len(lists))
    for k, hs in enumerate(lists):
        out["handles_%d" % k] = np.array(hs, dtype=np.int32)
        out["forest_%d" % k] = np.stack(forests[k])
        out["adjacency_%d" % k] = np.stack(adjs[k])
        out["attr_%d" % k] = np.stack(attrs[k])
        out["pytree_%d" % k] = np.stack(pytrees[k])
    path = os.path.join(cg.GOLD, "subset_cfg2.npz")
    np.savez_compressed(path, **out)
    on = (out["snaps"][:, :, 0] >= 0).sum(1)
    diff = [int((out["forest_%d" % k][:, :len(hs)] != out["forest_0"][:, hs]).any(axis=(2, 3)).sum()) for k, hs in enumerate(lists)]
    full = out["pytree_0"]
    diff_py = [int((out["pytree_%d" % k] != full[:, hs]).any(axis=(2, 3)).sum()) for k, hs in enumerate(lists)]
    print("upstream trees that differ from the full-list call per list:", diff_py)
    print("subset_cfg2: A=%d snapshots at %s, on-map %s, trees that differ from the full-list call per list: %s -> %.0f KB" % (A, steps, on.tolist(), diff, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
