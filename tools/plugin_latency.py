#!/usr/bin/env python3
"""Latency of the DROP-IN itself: `plugin.TreeObsForRailEnv.get_many` (the reference's pybind11 `flatland_cutils.TreeObsForRailEnv`
replaced behind its own plugin API) on a duck-typed env replaying a reference episode (tests/util.DuckEnv) -- milliseconds per call,
split into the Python attribute reads (Agent::Agent, loader.cpp:8-120), fl_set_state (H2D), the B = 1 kernel launch + check,
the read-back (D2H) and the conversion to nested Python lists -- beside BASELINE.md section 2's cost of the reference's own get_many
(1.7 / 4.5 / 17.9 ms at cfg1 / cfg2 / cfg3, one core of the build container's Xeon).

  python tools/plugin_latency.py [--out profiles/r05_plugin_latency.json]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REFERENCE_MS = {"cfg1": 1.7, "cfg2": 4.5, "cfg3": 17.9}      # get_many of the reference, ms per call
CASES = (("cfg1", "cfg1_uniform"), ("cfg2", "cfg2_uniform"), ("cfg3", "cfg3_uniform"), ("cfg4", "cfg4_fwd_head"), ("cfg5", "cfg5_fwd_head"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--calls", type=int, default=120)
    a = ap.parse_args()
    import numpy as np
    from tests import util
    from flatland_marl_amd.plugin import TreeObsForRailEnv
    res = {"unit": "ms per get_many call (mean over the calls)", "cases": {}}
    for cfg, name in CASES:
        fx = util.load(name)
        env = util.DuckEnv(fx)
        b = TreeObsForRailEnv(31, 500)
        b.set_env(env)
        b.reset()
        A = env.get_num_agents()
        handles = list(range(A))
        steps = len(fx["s_row"])
        n = min(a.calls, steps)
        first = max(1, steps - n) if cfg in ("cfg1", "cfg2", "cfg3") else 1     # late in the episode: agents on the map
        for T in range(first, first + 5):      # warm-up
            env.goto(min(T, steps))
            b.get_many(handles)
        row = {}
        for mode in ("lists", "arrays"):
            b._bind.profile = {}
            t0 = time.perf_counter()
            on_map = 0
            for k in range(n):
                env.goto(min(first + k, steps))
                t1 = time.perf_counter()
                b.get_many(handles, as_arrays=(mode == "arrays"))
                b._bind.profile["total"] = b._bind.profile.get("total", 0.0) + time.perf_counter() - t1
                on_map += sum(ag.position is not None for ag in env.agents)
            row[mode] = {k: round(v / n * 1e3, 4) for k, v in b._bind.profile.items()}
            row[mode]["on_map_agents"] = round(on_map / n, 1)
        b._bind.profile = None
        row["agents"] = A
        row["grid"] = [int(env.height), int(env.width)]
        row["reference_get_many_ms"] = REFERENCE_MS.get(cfg)
        res["cases"][cfg] = row
        print(cfg, json.dumps(row), file=sys.stderr, flush=True)
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.out:
        open(a.out if os.path.isabs(a.out) else os.path.join(ROOT, a.out), "w").write(txt + "\n")


if __name__ == "__main__":
    main()
