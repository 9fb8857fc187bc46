"""CPU: the bench's base envs (package data) are the static descriptions of the golden fixtures captured from the reference."""
import glob
import os

import numpy as np

from flatland_marl_amd import workload as wl
from tests import util


def test_package_data_equals_the_golden_fixtures():
    files = sorted(glob.glob(os.path.join(wl.DATA, "*.npz")))
    assert len(files) >= 15
    for f in files:
        name = os.path.basename(f)[:-4]
        z, fx = np.load(f), util.load(name)
        assert sorted(z.files) == sorted(wl.STATIC_KEYS)
        for k in wl.STATIC_KEYS:
            np.testing.assert_array_equal(z[k], fx[k], err_msg=f"{name} {k}")


def test_every_workload_resolves_and_replica_zero_keeps_the_fixture_rng():
    for name, w in wl.WORKLOADS.items():
        envs, seed = wl.make_envs(name, B=3)
        fx = util.load(w["pinned"][0])
        np.testing.assert_array_equal(envs[0]["mt_key"], fx["mt_key"])
        assert int(envs[0]["mt_pos"]) == int(fx["mt_pos"]) and seed == w["pinned"][1]
        assert not np.array_equal(envs[1]["mt_key"], envs[0]["mt_key"])
