#!/usr/bin/env python3
"""Writes tests/golden/cfg1_persist.pkl with the REAL reference (RailEnvPersister.save, persistence.py:24-64) for the
same env as tests/golden/cfg1_uniform.npz (Test_0 / Level_0, after reset()).  Build container only."""
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
