"""GPU: FL_OBS_KEEP_VERIFY (diagnostic of FL_OBS_KEEP_TREE_ROWS): the mode rests on the caller's promise that the tree buffer is the previous
launch's, untouched.  With the switch set (read once per process: a child) a kept launch first checks the promise -- an honest caller passes,
an in-place modification of the handed-out tensor is reported by check()."""
import os
import subprocess
import sys

import pytest

from tests import util

pytestmark = pytest.mark.gpu

CHILD = r"""
import numpy as np, torch
from flatland_marl_amd import workload as wl
from flatland_marl_amd.hip_backend import BatchedRailEnv, FlatlandHipError
envs, seed = wl.make_envs("cfg3", B=12, distinct=2)
env = BatchedRailEnv(envs)
env.keep_tree_rows()
for t in range(30):
    env.step_synth(seed, 0, 2, auto_reset=True)
    _, tree = env.obs_both(3, 30)
    env.check()                                   # an honest caller: the promise holds on every launch
tree[torch.isinf(tree) & (tree < 0)] = -1.0       # what a consumer might do before a network -- in place, on the handed-out tensor
env.step_synth(seed, 0, 2, auto_reset=True)
env.obs_both(3, 30)
try:
    env.check()
    print("RESULT not detected")
except FlatlandHipError as e:
    print("RESULT", e)
"""


def test_keep_verify_passes_an_honest_caller_and_reports_a_modified_buffer():
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, PYTHONPATH=util.ROOT, FL_OBS_KEEP_VERIFY="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][0]
    assert "FL_ERR_ARG" in line and "tree buffer" in line, line
