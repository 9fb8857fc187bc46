#!/usr/bin/env python3
"""Golden vectors on a grid that is NOT made of Flatland's rail cell types: a generated map (Round-2 Test_2 / Test_4, level 0)
in which one switch gets a third way on for one direction of travel (three transitions in one nibble).  The observation kernels
choose compact node tables for the upstream tree only when no direction of a cell has more than two transitions; this pins the
other path (and the oracle) on what the REAL reference does with such a cell: per step the agents' state, the upstream depth-2 /
depth-3 trees, the flatland_cutils tensors (or the fact that it raised).  Build-container only; data, no reference source."""
import contextlib
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import capture_golden as cg  # noqa: E402
from flatland_marl_amd import synth  # noqa: E402


def widen(grid):
    """the first switch (row-major, then direction N, E, S, W) with two ways on that can get a third onto a rail cell"""
    H, W = grid.shape
    for r in range(1, H - 1):
        for c in range(1, W - 1):
            for d in range(4):
                nib = (int(grid[r, c]) >> ((3 - d) * 4)) & 15
                if bin(nib).count("1") != 2:
                    continue
                for m in range(4):
                    if (nib >> (3 - m)) & 1 or m == (d + 2) % 4:
                        continue
                    nr, nc = r + (-1, 0, 1, 0)[m], c + (0, 1, 0, -1)[m]
                    if grid[nr, nc] != 0:
                        return (r, c, d, m)
    raise AssertionError("no switch to widen")


def capture(name, test_id, steps=90, seed=17):
    row = cg.csv_row(test_id, "Level_0")
    env, mp = cg.make_env(row)
    with contextlib.redirect_stdout(io.StringIO()):
        env.reset()
    r, c, d, m = widen(np.asarray(env.rail.grid))
    env.rail.grid[r, c] = np.uint16(int(env.rail.grid[r, c]) | (1 << ((3 - d) * 4 + (3 - m))))
    env.distance_map.reset(env.agents, env.rail)      # the distance map of the changed grid (recomputed at the next get())
    builders = {2: cg.PyTreeObs(max_depth=2, predictor=cg.ShortestPathPredictorForRailEnv(30)),
                3: cg.PyTreeObs(max_depth=3, predictor=cg.ShortestPathPredictorForRailEnv(30))}
    for b in builders.values():
        b.set_env(env)
        b.reset()
    cut = env.obs_builder        # the env's own flatland_cutils builder: reset() re-reads grid and distance map
    cut.reset()
    out = cg.static_arrays(env, mp)
    out.update(cg.dm_unique(env))
    out["widened"] = np.array([r, c, d, m], dtype=np.int32)
    A = env.get_num_agents()
    rec = {k: [] for k in ("state", "py_d2_p30", "py_d3_p30", "cutils_raised", "actions", "reward", "done")}
    cut_rec = {}

    def observe(cutils_obs_or_exc):
        with contextlib.redirect_stdout(io.StringIO()):
            rec["py_d2_p30"].append(cg.pytree_arrays(builders[2], env, 2))
            rec["py_d3_p30"].append(cg.pytree_arrays(builders[3], env, 3))
        raised = isinstance(cutils_obs_or_exc, Exception)
        rec["cutils_raised"].append(int(raised))
        arrs = None if raised else cg.cutils_arrays(cutils_obs_or_exc, env)
        for k in ("attr", "forest", "adjacency", "node_order", "edge_order", "valid"):
            shape_like = cut_rec[k][0] if k in cut_rec and cut_rec[k] else None
            if arrs is not None:
                cut_rec.setdefault(k, []).append(arrs[k])
            else:
                cut_rec.setdefault(k, []).append(None)
        s = cg.agent_snapshot(env)
        rec["state"].append(np.stack([s[k] for k in ("row", "col", "dir", "state", "malf", "nmalf", "scount", "saved", "arrival",
                                                      "old_row", "old_col", "old_dir")], axis=1).astype(np.int32))

    def get_cutils():
        try:
            return cut.get_many(list(range(A)))
        except ValueError as e:
            return e

    observe(get_cutils())
    for t in range(steps):
        acts = synth.forward_biased_actions(seed, 0, t, A)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                obs, rew, dones, _ = env.step({i: int(a) for i, a in enumerate(acts)})
        except ValueError as e:          # flatland_cutils raised inside step(): the dynamics of the step are done by then
            obs, rew, dones = e, env.rewards_dict, env.dones
        rec["actions"].append(acts.astype(np.uint8))
        rec["reward"].append(np.array([rew[i] for i in range(A)], dtype=np.int32))
        rec["done"].append(np.array([dones[i] for i in range(A)], dtype=np.uint8))
        observe(obs)
        if dones["__all__"]:
            break
    for k, v in rec.items():
        out[k] = np.stack(v)
    for k, v in cut_rec.items():
        proto = next(x for x in v if x is not None)
        out["o_" + k] = np.stack([x if x is not None else np.zeros_like(proto) for x in v])
    path = os.path.join(cg.GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "widened", (r, c, d, m), "steps", len(rec["actions"]), "cutils raised at", int(np.sum(out["cutils_raised"])), "of", len(out["cutils_raised"]),
          "->", os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    capture("threeway_cfg2", "Test_2")
    capture("threeway_cfg3", "Test_4", steps=60)
