"""Diagnostic: per-phase clocks of the fused observation kernel, from a -DFL_OBS_TIMING build of the library.

  mkdir -p build_ab; OUT=$PWD/ab_libs/libfl_timing.so EXTRA_HIPCC_FLAGS=-DFL_OBS_TIMING FORCE=1 flatland_marl_amd/csrc/build.sh
  python tools/obs_phase_clocks.py ab_libs/libfl_timing.so [workload [tree depth]]        (on the GPU box)

Prints the mean over envs / steps (us; wall_clock64 ticks at 100 MHz).  The phases of the trees are accumulated over the
rounds of 32 (cutils) / 16-32 (upstream) trees.
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import flatland_marl_amd.hip_backend as hb  # noqa: E402

hb.LIB_PATH = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
from flatland_marl_amd import workload as wl  # noqa: E402

distinct = int(sys.argv[4]) if len(sys.argv) > 4 else 0
envs, seed = wl.make_envs(workload, distinct=distinct)
env = hb.BatchedRailEnv(envs, device=0)
L = hb.lib()
L.fl_debug_obs_clocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
acc = []
# The clocks are taken in the BENCH's regime (round 5): the replicas are de-phased exactly as bench.py does before its timed region
# (env b starts over once at an offset drawn from [0, T_b) by the same generator; the last 64 de-phasing steps with the observations),
# then the bench's default warm-up -- the same steady-state mix of episode phases, the same number of agents on the map.
# NO_DEPHASE=1: from a synchronised reset (what rounds 1 - 4 printed), WARM steps before the clocks are read.
B = env.B
if os.environ.get("NO_DEPHASE"):
    warm = int(os.environ.get("WARM", "200"))
else:
    rs = np.random.RandomState(12345)
    offs = np.array([rs.randint(0, int(e["T"])) for e in envs])
    dephase_steps = int(max(int(e["T"]) for e in envs))
    by_step = {}
    for b, o in enumerate(offs):
        by_step.setdefault(int(o), []).append(b)
    for s in range(dephase_steps):
        env.step_synth(seed, 0, 0, auto_reset=True)
        if s >= dephase_steps - 64:
            env.obs_both(depth, 30)
        if s in by_step:
            m = np.zeros(B, dtype=np.uint8)
            m[by_step[s]] = 1
            env.reset(m, fresh=True)
    warm = int(os.environ.get("WARM", "20"))
n_clock = int(os.environ.get("STEPS", "100"))
on_map = []
for t in range(warm + n_clock):
    env.step_synth(seed, 0, 0, auto_reset=True)
    env.obs_both(depth, 30)
    if t >= warm:
        out = np.zeros((env.B, 64), dtype=np.int64)
        assert L.fl_debug_obs_clocks(env.h, out.ctypes.data) == 0
        acc.append(out)
        if t % 10 == 0:
            on_map.append((env.state()[0][:, :, 0] >= 0).sum(1).mean())
c = np.stack(acc).astype(np.float64)  # [steps, B, 64]
# the kernels that build the trees of both builders in stage 1 have no stage 2: the kernel ends with stage 1's last stamp
c[:, :, 37] = np.where(c[:, :, 37] > 0, c[:, :, 37], c[:, :, 5])


def seg(a, b):
    return ((c[:, :, b] - c[:, :, a]) / 100.0).mean()


def dur(k):
    return (c[:, :, k] / 100.0).mean()


def report():

    print("%s, upstream depth %d%s, %s: on-map agents %.1f of %d (mean over the clocked steps), launch class %s" %
          (workload, depth, ", %d distinct maps" % distinct if distinct else "", "synchronised reset + %d steps" % warm if os.environ.get("NO_DEPHASE") else "de-phased like bench.py",
           float(np.mean(on_map)), env.A, env.last_obs_class()))
    print("stage 1 (cutils):")
    print("  p0 stage %.1f  p1 %.1f  p2a walk|phase1|passA + count %.1f  p2b scan/fill %.1f" % (seg(0, 1), seg(1, 2), seg(2, 3), seg(3, 4)))
    print("     since p2a start: deadlock wavefront done %.1f, walkers done %.1f, hoisted pass A done %.1f, rest of phase 1 done %.1f" % (seg(2, 20), seg(2, 21), seg(2, 19), seg(2, 22)))
    print("     hoisted pass A alone: slowest wavefront %.1f" % dur(23))
    if c[:, :, 56].max() > 0:
        print("     scan/fill: keys scanned + bucket offsets %.1f, items filled %.1f, rest (offsets to HBM) %.1f" % (seg(3, 56), seg(56, 57), seg(57, 4)))
    print("  trees (sum over rounds): passA %.1f  B classify %.1f  B work lists %.1f  rows %.1f  orders %.1f   stage total %.1f" %
          (dur(6), dur(11), dur(7), dur(8), dur(16), seg(0, 5)))
    if c[:, :, 32].max() > 0:   # (the kernels with one pass B for both builders have no second stage)
        print("stage 2 (upstream tree):")
        print("  prep %.1f  count %.1f  scan/fill %.1f" % (seg(32, 34), seg(34, 35), seg(35, 36)))
        print("  trees: passA %.1f  B classify %.1f  B work lists %.1f  rows %.1f   stage total %.1f" % (dur(38), dur(43), dur(39), dur(40), seg(32, 37)))
    print("kernel total %.1f us (slowest env %.1f)" % (seg(0, 37), ((c[:, :, 37] - c[:, :, 0]) / 100.0).max(1).mean()))
    pb = c[:, :, 25].astype(np.uint64)
    print("pass B (max over rounds / stages): slowest-lane loop %.1f us, cells/lane %.1f, cells per round %.0f; slowest-lane setup %.1f us (skipped %.1f cells)" %
          (((pb >> np.uint64(40)).astype(np.float64) / 100.0).mean(), ((pb >> np.uint64(20)) & np.uint64(0xFFFFF)).astype(np.float64).mean(),
           c[:, :, 26].mean(), (c[:, :, 24] / 100.0).mean(), (pb & np.uint64(0xFFFFF)).astype(np.float64).mean()))
    def rel(k, k0):  # absolute 40-bit marks of the timing build
        return ((c[:, :, k] - c[:, :, k0]) / 100.0).mean()

    def rel_min(k, k0):
        return (((1 << 40) - c[:, :, k] - c[:, :, k0]) / 100.0).mean()

    if (env.A <= 31 and c[:, :, 50].max() == 0) or os.environ.get("WL_MARKS"):  # marks of the LAST round of trees (of the last stage) only
        print("work-list step, since its start: occupants done %.1f, conflict scan done: earliest wavefront %.1f, latest %.1f" %
              (rel(12, 18), rel_min(17, 18), rel(13, 18)))
    if c[:, :, 51].max() > 0:   # mean wavefront, summed over the rounds of trees (and both stages)
        nw = float(os.environ.get("NWAVES", "16"))
        print("work-list step (mean wavefront, sum over rounds): occupants %.1f, first chunks of the conflict entries %.1f, late jobs %.1f, barrier %.1f, further chunks %.1f, end %.1f" %
              tuple((c[:, :, 48 if k == 5 else 51 + k] / 100.0 / nw).mean() for k in range(6)))
    if c[:, :, 27].max() > 0:  # a -DFL_OBS_COUNTS build
      print("conflict entries (cutils | upstream): items in their lists %.0f | %.0f, queried beyond the fine buckets %.0f | %.0f, somebody else there %.0f | %.0f, conflicts %.0f | %.0f, only the walking agent itself there %.0f | %.0f" %
            (c[:, :, 27].mean(), c[:, :, 59].mean(), c[:, :, 28].mean(), c[:, :, 60].mean(), c[:, :, 29].mean(), c[:, :, 61].mean(), c[:, :, 30].mean(), c[:, :, 62].mean(), c[:, :, 31].mean(), c[:, :, 63].mean()))
    print("work-list entries (sum over rounds): occupant %.0f / %.0f, conflict %.0f / %.0f  (cutils / upstream)" %
          (c[:, :, 9].mean(), c[:, :, 41].mean(), c[:, :, 10].mean(), c[:, :, 42].mean()))
    if c[:, :, 58].max() > 0:
        # scratch an env writes and reads back per launch, by kind (requested bytes): a work-list entry is 8 B, an item 4 B, a waypoint 2 B
        ent = c[:, :, 9].mean() + c[:, :, 10].mean() + (c[:, :, 41].mean() + c[:, :, 42].mean() if c[:, :, 32].max() > 0 else 0)
        print("scratch per env-step (requested bytes): prediction items %.0f (%.1f KB), predicted waypoints %.0f (%.1f KB), work-list entries %.0f (%.1f KB)" %
              (c[:, :, 58].mean(), c[:, :, 58].mean() * 4 / 1024, c[:, :, 59].mean(), c[:, :, 59].mean() * 2 / 1024, ent, ent * 8 / 1024))


report()
# the same for the slowest env of every step (it sets the launch time)
full = c
slow = ((full[:, :, 37] - full[:, :, 0])).argmax(1)
c = np.stack([full[t, slow[t]] for t in range(full.shape[0])])[:, None, :]
print("---- slowest env of each step:")
report()
