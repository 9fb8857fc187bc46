#!/usr/bin/env python3
"""agent-steps/s of the fused step (k_step + both observation builders, as bench.py times it) against the number of envs B at one
workload's shape: how the throughput moves once a CU holds more than one env's worth of work.  One child process per launcher mode
(the switches are read once per process):

  python tools/bsweep.py [--workload cfg2] [--bs 256,512,1024,2048] [--modes default,round16] [--out profiles/r05_cfg2_bsweep.json]

modes: default (the launcher's own choice by B), one_a_cu (FL_OBS_ROUND16=0: the 1024-thread one-round kernel, class 1, for every B),
two_a_cu (FL_OBS_ROUND16=2: 512-thread workgroups in rounds of 16 agents, at most 80 KB of LDS, two workgroups a CU: class 5, for every B),
nofix / nofix_two_a_cu (the same two with the runtime carving, FL_OBS_NO_FIX)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = {"default": {}, "one_a_cu": {"FL_OBS_ROUND16": "0"}, "two_a_cu": {"FL_OBS_ROUND16": "2"}, "nofix": {"FL_OBS_NO_FIX": "1", "FL_OBS_ROUND16": "0"},
         "nofix_two_a_cu": {"FL_OBS_NO_FIX": "1", "FL_OBS_ROUND16": "2"}}


def child(workload, bs, depth, steps, pack=0):
    sys.path.insert(0, ROOT)
    import bench
    out = {}
    for B in bs:
        r = bench.run_workload(workload, depth, 30, B, steps, 20, 0, 1, 0, event_steps=32, pack=pack)
        out[str(B)] = dict(value=r["value"], ms_per_step=r["ms_per_step"], kernel_ms=r["kernel_ms"], on_map=r["on_map_agents_per_env"], launch_class=r.get("launch_class"))
        print("B=%d %.1f M agent-steps/s, %.3f ms/step, kernels %s" % (B, r["value"] / 1e6, r["ms_per_step"], r["kernel_ms"]), file=sys.stderr, flush=True)
    print("RESULT " + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--bs", default="256,512,1024,2048")
    ap.add_argument("--modes", default="one_a_cu,two_a_cu,default")
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--pack", type=int, default=0, help="1 with --depth 0: the consumer's path (the flatland_cutils builder alone writing the policy's tensors)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--child", default=None)
    a = ap.parse_args()
    bs = [int(x) for x in a.bs.split(",")]
    if a.child:
        return child(a.workload, bs, a.depth, a.steps, a.pack)
    res = {"workload": a.workload, "tree_depth": a.depth, "pack": a.pack, "steps": a.steps, "modes": {}}
    for m in a.modes.split(","):
        env = dict(os.environ, **MODES[m])
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", m, "--workload", a.workload, "--bs", a.bs,
                            "--depth", str(a.depth), "--steps", str(a.steps), "--pack", str(a.pack)], env=env, capture_output=True, text=True)
        sys.stderr.write(p.stderr[-3000:])
        if p.returncode != 0:
            res["modes"][m] = {"error": p.stderr[-500:]}
            continue
        res["modes"][m] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.out:
        open(a.out if os.path.isabs(a.out) else os.path.join(ROOT, a.out), "w").write(txt + "\n")


if __name__ == "__main__":
    main()
