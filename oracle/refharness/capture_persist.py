#!/usr/bin/env python3
"""Writes tests/golden/cfg1_persist.pkl with the REAL reference (RailEnvPersister.save, persistence.py:24-64) for the
same env as tests/golden/cfg1_uniform.npz (Test_0 / Level_0, after reset()), and persist_reset_cfg1.npz: what the reference's
RailEnvPersister.load_new(file) + env.reset() -- the path of solution/demo.py and eval_env.py:97-102 -- makes of that file from a
known MT19937 state: rail_from_file / line_from_file give the same rail and line back, timetable_generator DRAWS the timetable
again (no agents_hints: num_cities = 2, timetable_generators.py:36-40).  Build container only."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import capture_golden as cg  # noqa: E402  (sets up sys.path for the reference)
from flatland.envs.persistence import RailEnvPersister  # noqa: E402

if __name__ == "__main__":
    row = cg.csv_row("Test_0", "Level_0")
    env, mp = cg.make_env(row)
    env.reset()
    out = os.path.join(cg.GOLD, "cfg1_persist.pkl")
    RailEnvPersister.save(env, out, save_distance_maps=True)
    print(out, os.path.getsize(out) // 1024, "KB")
    import numpy as np
    rec = {}
    for tag, seed_words, kw in (("a", [5], {}), ("b", [77, 3], dict(regenerate_rail=False, regenerate_schedule=True))):
        env2, _ = RailEnvPersister.load_new(out)
        env2.np_random = np.random.RandomState(seed_words)
        st0 = env2.np_random.get_state()
        env2.reset(**kw)
        st1 = env2.np_random.get_state()
        mfp = env2.malfunction_generator.MFP
        rec.update({f"{tag}_mt_key0": np.asarray(st0[1], dtype=np.uint32), f"{tag}_mt_pos0": np.int32(st0[2]),
                    f"{tag}_mt_key1": np.asarray(st1[1], dtype=np.uint32), f"{tag}_mt_pos1": np.int32(st1[2]),
                    f"{tag}_earliest": np.array([a.earliest_departure for a in env2.agents], dtype=np.int32),
                    f"{tag}_latest": np.array([a.latest_arrival for a in env2.agents], dtype=np.int32),
                    f"{tag}_T": np.int32(env2._max_episode_steps),
                    f"{tag}_malf": np.array([mfp.malfunction_rate, mfp.min_duration, mfp.max_duration], dtype=np.float64),
                    f"{tag}_init_pos": np.array([a.initial_position for a in env2.agents], dtype=np.int32),
                    f"{tag}_target": np.array([a.target for a in env2.agents], dtype=np.int32),
                    f"{tag}_grid": np.asarray(env2.rail.grid, dtype=np.uint16)})
    out2 = os.path.join(cg.GOLD, "persist_reset_cfg1.npz")
    np.savez_compressed(out2, **rec)
    print(out2, os.path.getsize(out2) // 1024, "KB")
