"""Diagnostic: time the fused observation launch of a workload under several forced LDS configurations
(FL_OBS_FORCE, see obs_pick_config in csrc/fl_obs.hip).  One child process per configuration.

  python tools/obs_sweep.py cfg5 2 "" "nt=1024,tmask=1" "nt=512"        (on the GPU box)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workload, depth = sys.argv[1], sys.argv[2]
for force in sys.argv[3:]:
    env = dict(os.environ, FL_OBS_FORCE=force, FL_OBS_VERBOSE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra-workloads", "--workload", workload, "--tree-depth", depth,
                        "--steps", "120", "--warmup", "20"], env=env, capture_output=True, text=True)
    cfg = [l for l in p.stderr.splitlines() if l.startswith("[fl_obs]")]
    try:
        j = json.loads(p.stdout.strip().splitlines()[-1])
        print("%s d%s %-40s %6.1f M  obs %.3f ms  | %s" % (workload, depth, force or "(default)", j["value"] / 1e6,
                                                          j["kernel_ms"].get("obs_cutils_tree_fused", 0.0), cfg[0][9:] if cfg else ""), flush=True)
    except Exception:
        print(workload, depth, force, "FAILED", p.stderr[-300:], flush=True)
