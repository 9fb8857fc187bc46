"""Diagnostic: per-phase clocks of the fused observation kernel, from a -DFL_OBS_TIMING build of the library.

  mkdir -p gpurun_tmp; OUT=$PWD/gpurun_tmp/libfl_timing.so EXTRA_HIPCC_FLAGS=-DFL_OBS_TIMING FORCE=1 flatland_marl_amd/csrc/build.sh
  python tools/obs_phase_clocks.py gpurun_tmp/libfl_timing.so [workload]        (on the GPU box)

Prints the mean over envs / steps of the time between stamps (us; wall_clock64 ticks at 100 MHz).
"""
import ctypes
import sys

import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import flatland_marl_amd.hip_backend as hb

hb.LIB_PATH = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
from flatland_marl_amd import workload as wl  # noqa: E402

envs, seed = wl.make_envs(workload)
env = hb.BatchedRailEnv(envs, device=0)
L = hb.lib()
L.fl_debug_obs_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
acc = []
for t in range(120):
    env.step_synth(seed, 0, 0, auto_reset=True)
    env.obs_both(2, 30)
    if t >= 20:
        out = np.zeros((env.B, 32), dtype=np.int64)
        assert L.fl_debug_obs_clocks(env.h, out.ctypes.data) == 0
        acc.append(out)
c = np.stack(acc).astype(np.float64)  # [steps, B, 32]
names1 = ["p0 stage", "p1 attr/deadlock", "p2a walk+count", "p2b scan/fill/sort", "-", "passA", "passB", "rows", "orders(end)"]
def seg(a, b):
    return ((c[:, :, b] - c[:, :, a]) / 100.0).mean()
print("stage 1 (cutils):")
print("  p0 %.1f  p1 %.1f  p2a %.1f  p2b %.1f" % (seg(0, 1), seg(1, 2), seg(2, 3), seg(3, 4)))
print("  passA %.1f  passB %.1f  rows %.1f  orders %.1f   total %.1f" % (seg(4, 6), seg(6, 7), seg(7, 8), seg(8, 5), seg(0, 5)))
print("stage 2 (upstream tree):")
print("  prep %.1f  count %.1f  scan/fill/sort %.1f" % (seg(16, 18), seg(18, 19), seg(19, 20)))
print("  passA %.1f  passB %.1f  rows %.1f   total %.1f" % (seg(20, 22), seg(22, 23), seg(23, 21), seg(16, 21)))
print("kernel total %.1f us (slowest env %.1f)" % (seg(0, 21), ((c[:, :, 21] - c[:, :, 0]) / 100.0).max(1).mean()))
pb = c[:, :, 14].astype(np.uint64)
print("pass B (stage-1+2 max): slowest-lane loop %.1f us, cells/lane %.1f, total cells %.0f" %
      (((pb >> np.uint64(40)).astype(np.float64) / 100.0).mean(), ((pb >> np.uint64(20)) & np.uint64(0xFFFFF)).astype(np.float64).mean(), c[:, :, 15].mean()))
print("pass B slowest-lane setup (search + skip) %.1f us; cells skipped by the slowest lane %.1f" %
      ((c[:, :, 12] / 100.0).mean(), (pb & np.uint64(0xFFFFF)).astype(np.float64).mean()))
print("pass B step 1 (classify) %.1f / %.1f us, step 2 (work lists) %.1f / %.1f us; list entries occ %.0f / %.0f, conflict %.0f / %.0f  (cutils / upstream)" %
      (seg(6, 11), seg(22, 27), seg(11, 7), seg(27, 23), c[:, :, 9].mean(), c[:, :, 25].mean(), c[:, :, 10].mean(), c[:, :, 26].mean()))
print("pass B prelude (team_prepare, barrier, team search; wavefront 0) %.1f / %.1f us" % (seg(6, 13), seg(22, 29)))
