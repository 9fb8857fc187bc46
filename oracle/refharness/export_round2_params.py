#!/usr/bin/env python3
"""Exports the Round-2 evaluation parameters (solution/debug-environments/parameters_flatland_round_2_new.csv of the
reference: 15 tests x 10 levels, one random_seed per level) as package data flatland_marl_amd/data/round2_params.json, so that
bench.py / workload.py can generate distinct maps with the native host generators on a box without the reference.
Build-container only; data, no reference source."""
import json
import os
import sys

import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("REF", "/root/reference")
df = pd.read_csv(os.path.join(REF, "solution/debug-environments/parameters_flatland_round_2_new.csv"))
out = {}
for test_id, g in df.groupby("test_id", sort=False):
    r0 = g.iloc[0]
    speeds = eval(str(r0["speed_ratios"]))
    out[test_id] = dict(width=int(r0["x_dim"]), height=int(r0["y_dim"]), n_agents=int(r0["n_agents"]), n_cities=int(r0["n_cities"]),
                        max_rails_between_cities=int(r0["max_rails_between_cities"]), max_rail_pairs_in_city=int(r0["max_rail_pairs_in_city"]),
                        grid_mode=bool(eval(str(r0["grid_mode"]))), malfunction_interval=int(r0["malfunction_interval"]),
                        malfunction_duration_min=int(r0["malfunction_duration_min"]), malfunction_duration_max=int(r0["malfunction_duration_max"]),
                        speed_values=[float(k) for k in speeds], speed_probs=[float(v) for v in speeds.values()],
                        seeds=[int(s) for s in g["random_seed"]])
    for col in ("x_dim", "y_dim", "n_agents", "n_cities", "malfunction_interval"):
        assert g[col].nunique() == 1, (test_id, col)
path = os.path.join(REPO, "flatland_marl_amd", "data", "round2_params.json")
json.dump(out, open(path, "w"), indent=1)
print(path, len(out), "tests")
