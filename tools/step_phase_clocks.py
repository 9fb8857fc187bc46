"""Diagnostic: clocks of lane 0 at the marks of k_step (RailEnv.step), from a -DFL_STEP_TIMING build of the library.

  git apply tools/step_timing.patch && tools/build_variant.sh steptiming -DFL_STEP_TIMING && git apply -R tools/step_timing.patch
  (the marks live in a patch so that the kernel sources of record -- bench.KERNEL_SOURCES -- stay as profiled)
  python tools/step_phase_clocks.py ab_libs/libfl_steptiming.so [workload]        (on the GPU box)

Prints the mean over envs / steps of the time since the workgroup's first instruction (us; wall_clock64 ticks at 100 MHz),
de-phased like bench.py.
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import flatland_marl_amd.hip_backend as hb  # noqa: E402

hb.LIB_PATH = os.path.abspath(sys.argv[1])
workload = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 0
from flatland_marl_amd import workload as wl  # noqa: E402

envs, seed = wl.make_envs(workload)
env = hb.BatchedRailEnv(envs, device=0)
L = hb.lib()
B = env.B
rs = np.random.RandomState(12345)
offs = np.array([rs.randint(0, int(e["T"])) for e in envs])
dephase_steps = int(max(int(e["T"]) for e in envs))
by_step = {}
for b, o in enumerate(offs):
    by_step.setdefault(int(o), []).append(b)
for s in range(dephase_steps):
    env.step_synth(seed, kind, 0, auto_reset=True)
    if s in by_step:
        m = np.zeros(B, dtype=np.uint8)
        m[by_step[s]] = 1
        env.reset(m, fresh=True)
out = np.zeros(16, dtype=np.uint64)
L.fl_debug_step_clocks.argtypes = [ctypes.c_void_p]
assert L.fl_debug_step_clocks(out.ctypes.data) == 0     # (reads and resets)
n = int(os.environ.get("STEPS", "200"))
for t in range(n):
    env.step_synth(seed, kind, 0, auto_reset=True)
    env.obs_both(2, 30)
assert L.fl_debug_step_clocks(out.ctypes.data) == 0
cnt = float(out[15])
names = ("state + MT block in LDS", "stream words of the speculative draws", "malfunction draws", "phase 1 (waits for the cell word)",
         "MotionCheck", "phase 2", "end of episode", "written back")
print("%s: %d workgroups, marks of lane 0 since the workgroup's start, us (mean)" % (workload, int(cnt)))
prev = 0.0
for k, nm in enumerate(names):
    v = out[k] / cnt / 100.0
    print("  %-42s %6.2f  (+%.2f)" % (nm, v, v - prev))
    prev = v
