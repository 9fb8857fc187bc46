#!/usr/bin/env python3
"""Golden vectors for the error paths of the tree builders on MALFORMED maps: the reference's 7x10 hand-made
rail (flatland/utils/simple_rail.py make_simple_rail) with one cell broken, run through the REAL reference:

  zero_transition : a horizontal straight replaced by a vertical one -- a branch walk enters a rail cell that has no transition
                    for its direction of travel;
  leaves_rail     : a straight removed -- the transition of its neighbour now points at an empty cell.

flatland_cutils raises std::invalid_argument -> ValueError("WRONG CELL TYPE detected in tree-search (0 transitions possible) ...")
when a tree walk gets there (treeobs.cpp:528-535); the upstream TreeObsForRailEnv prints the same words and makes the node a
terminal one (observations.py:420-425).  Stored per step: the agents' state, the upstream depth-2 / depth-3 trees, and whether
(and with which message) flatland_cutils raised.  Build-container only; data, no reference source."""
import contextlib
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import capture_golden as cg  # noqa: E402  (sets up sys.path for the reference)
from flatland.envs.line_generators import sparse_line_generator  # noqa: E402
from flatland.envs.malfunction_generators import MalfunctionParameters, ParamMalfunctionGen  # noqa: E402
from flatland.envs.rail_env import RailEnv  # noqa: E402
from flatland.envs.rail_generators import rail_from_grid_transition_map  # noqa: E402
from flatland.utils.simple_rail import make_simple_rail  # noqa: E402


def capture(name, breaker, n_agents=3, steps=24, seed=5):
    rail, rail_map, optionals = make_simple_rail()
    breaker(rail)
    mp = MalfunctionParameters(malfunction_rate=0.0, min_duration=0, max_duration=0)
    builders = {(2, 10): cg.PyTreeObs(max_depth=2, predictor=cg.ShortestPathPredictorForRailEnv(10)),
                (3, 10): cg.PyTreeObs(max_depth=3, predictor=cg.ShortestPathPredictorForRailEnv(10))}
    env = RailEnv(width=rail_map.shape[1], height=rail_map.shape[0], rail_generator=rail_from_grid_transition_map(rail, optionals),
                  line_generator=sparse_line_generator(), number_of_agents=n_agents, malfunction_generator=ParamMalfunctionGen(mp),
                  obs_builder_object=builders[(2, 10)], random_seed=seed)
    with contextlib.redirect_stdout(io.StringIO()):
        env.reset()
    if env._max_episode_steps < steps:      # (the timetable of a broken map can come out empty: keep the episode alive)
        env._max_episode_steps = steps + 6
    for b in builders.values():
        b.set_env(env)
        b.reset()
    cut = cg.TreeCutils(31, 500)
    cut.set_env(env)
    cut.reset()
    out = cg.static_arrays(env, mp)
    out.update(cg.dm_unique(env))
    A = env.get_num_agents()
    rec = {k: [] for k in ("state", "py_d2_p10", "py_d3_p10", "cutils_raised", "python_printed", "actions", "reward", "done")}
    msgs = []

    def observe():
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            rec["py_d2_p10"].append(cg.pytree_arrays(builders[(2, 10)], env, 2))
            rec["py_d3_p10"].append(cg.pytree_arrays(builders[(3, 10)], env, 3))
        rec["python_printed"].append(int("WRONG CELL TYPE" in buf.getvalue()))
        try:
            cut.get_many(list(range(A)))
            rec["cutils_raised"].append(0)
        except ValueError as e:
            rec["cutils_raised"].append(1)
            msgs.append(str(e))
        s = cg.agent_snapshot(env)
        rec["state"].append(np.stack([s[k] for k in ("row", "col", "dir", "state", "malf", "nmalf", "scount", "saved", "arrival",
                                                      "old_row", "old_col", "old_dir")], axis=1).astype(np.int32))

    observe()
    rng = np.random.default_rng(seed)
    for t in range(steps):
        acts = rng.integers(1, 4, size=A)          # LEFT / FORWARD / RIGHT: the agents explore the junctions
        acts[rng.random(A) < 0.1] = 4
        with contextlib.redirect_stdout(io.StringIO()):
            _, rew, dones, _ = env.step({i: int(a) for i, a in enumerate(acts)})
        rec["actions"].append(acts.astype(np.uint8))
        rec["reward"].append(np.array([rew[i] for i in range(A)], dtype=np.int32))
        rec["done"].append(np.array([dones[i] for i in range(A)], dtype=np.uint8))
        observe()
        if dones["__all__"]:
            break
    for k, v in rec.items():
        out[k] = np.stack(v)
    out["cutils_message"] = np.array(msgs[0] if msgs else "")
    path = os.path.join(cg.GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "steps", len(rec["actions"]), "cutils raised at", np.flatnonzero(out["cutils_raised"]).tolist(), "python printed at",
          np.flatnonzero(out["python_printed"]).tolist(), "|", str(out["cutils_message"])[:90], "->", os.path.getsize(path) // 1024, "KB")


def zero_transition(rail):
    # (3, 1): horizontal straight -> vertical straight: entered from (3, 0) / (3, 2) it has no transition
    rail.grid[3, 1] = rail.transitions.transition_list[1]


def leaves_rail(rail):
    # (3, 8) removed: the east exit of (3, 7) and the west exit of the dead end (3, 9) point at an empty cell
    rail.grid[3, 8] = 0


if __name__ == "__main__":
    capture("malformed_zero_transition", zero_transition)
    capture("malformed_leaves_rail", leaves_rail)
