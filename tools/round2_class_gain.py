#!/usr/bin/env python3
"""What a launch class buys per shape of the Round-2 table (solution/debug-environments/parameters_flatland_round_2_new.csv, Test_0 ...
Test_14): for every test a batch of B envs over two generated levels, brought into traffic (shortest-path-following actions), then the
observation launches timed with HIP events -- both builders at depth 2 and depth 3 (fl_obs_cutils_tree) and the flatland_cutils builder alone
writing the policy's tensors (fl_obs_cutils_policy) -- three ways, each in a process of its own (the switches are read once):
    default              exact classes + bin classes (round 6)
    FL_OBS_NO_BINS=1     exact classes only: the launcher of rounds 4 / 5 (everything that is not a BASELINE shape: runtime carving)
    FL_OBS_NO_FIX=1      the runtime carving for every batch
  python tools/round2_class_gain.py [out=gpurun_out/round2_classes.txt] [tests=Test_0,...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODES = (("default", {}), ("no_bins", {"FL_OBS_NO_BINS": "1"}), ("no_fix", {"FL_OBS_NO_FIX": "1"}))
LAUNCHES = ("both_d2", "both_d3", "alone")


def measure(test):
    import numpy as np
    import torch
    from flatland_marl_amd import workload as wl
    from flatland_marl_amd.hip_backend import BatchedRailEnv
    levels = [wl.generate_level(test, lv) for lv in (1, 2)]
    A = len(levels[0]["init_dir"])
    B = 256 if A <= 100 else 128 if A <= 200 else 64
    envs = []
    for b in range(B):
        e = dict(levels[b % 2])
        e["mt_key"] = np.asarray(e["mt_key"]).copy()
        e["mt_key"][0] ^= np.uint32(b)            # (replicas differ in their malfunction streams)
        envs.append(e)
    env = BatchedRailEnv(envs)
    for t in range(90):                          # into traffic
        env.step_synth(5, 0, 2, auto_reset=True)
    out = {"agents": A, "envs": B, "rails": [int((np.asarray(e["grid"]) != 0).sum()) for e in levels]}
    calls = {"both_d2": lambda: env.obs_both(2, 30), "both_d3": lambda: env.obs_both(3, 30), "alone": lambda: env.obs_policy()}
    for name in LAUNCHES:
        for _ in range(4):
            env.step_synth(5, 0, 2, auto_reset=True); calls[name]()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(24)]
        for a, b2 in ev:
            env.step_synth(5, 0, 2, auto_reset=True)
            a.record(); calls[name](); b2.record()
        torch.cuda.synchronize()
        out[name] = {"us": float(np.mean([a.elapsed_time(b2) for a, b2 in ev]) * 1e3), "class": list(env.last_obs_class())}
    env.check()
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        print("RESULT " + json.dumps(measure(sys.argv[2])))
        return
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "round2_classes.txt")
    tests = sys.argv[2].split(",") if len(sys.argv) > 2 else ["Test_%d" % k for k in range(15)]
    res = {}
    for test in tests:
        res[test] = {}
        for mode, sw in MODES:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", test], env=dict(os.environ, **sw), capture_output=True, text=True, timeout=900)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
            if p.returncode != 0 or not line:
                print(test, mode, "FAILED", p.stderr[-500:], flush=True)
                continue
            res[test][mode] = json.loads(line[0][7:])
        print(test, {m: {k: (round(v[k]["us"], 1), v[k]["class"][0]) for k in LAUNCHES} for m, v in res[test].items()}, flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(res, open(os.path.splitext(out_path)[0] + ".json", "w"), indent=1)
    with open(out_path, "w") as f:
        f.write("Launch classes per shape of the Round-2 table (tools/round2_class_gain.py): mean launch time in us over 24 launches in traffic, (class) -- default | exact\n"
                "classes only (FL_OBS_NO_BINS, the launcher of rounds 4 / 5) | runtime carving (FL_OBS_NO_FIX); gain = default against exact-classes-only\n")
        f.write("%-8s %6s %5s %-11s" % ("test", "agents", "envs", "rails") + "".join("  %-44s" % k for k in LAUNCHES) + "\n")
        for test, r in res.items():
            if "default" not in r:
                continue
            d = r["default"]
            row = "%-8s %6d %5d %-11s" % (test, d["agents"], d["envs"], "/".join(map(str, d["rails"])))
            for k in LAUNCHES:
                cells = ["%7.1f (%2d)" % (r[m][k]["us"], r[m][k]["class"][0]) if m in r else "      -     " for m, _ in MODES]
                gain = (r["no_bins"][k]["us"] / d[k]["us"] - 1) * 100 if "no_bins" in r else float("nan")
                row += "  " + " | ".join(cells) + " %+5.1f%%" % gain
            f.write(row + "\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main()
