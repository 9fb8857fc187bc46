"""CPU: ingest of the reference's RailEnvPersister .pkl format (a file written by the real reference) without the
reference: the restored static description equals the golden one of the same env."""
import os
import pickle

import numpy as np
import pytest

from flatland_marl_amd import persistence
from tests import util


def test_pkl_ingest_matches_golden_static():
    fx = util.load("cfg1_uniform")
    d = persistence.load_env_dict(os.path.join(util.GOLD, "cfg1_persist.pkl"))
    st = persistence.static_from_env_dict(d, fx["mt_key"], fx["mt_pos"])
    for k in ("grid", "init_pos", "init_dir", "target", "speed", "earliest", "latest"):
        np.testing.assert_array_equal(st[k], fx[k], err_msg=k)
    assert int(st["T"]) == int(fx["T"])
    assert float(st["malf_rate"]) == float(fx["malf_rate"])
    assert (int(st["malf_min"]), int(st["malf_max"])) == (int(fx["malf_min"]), int(fx["malf_max"]))
    np.testing.assert_array_equal(persistence.distance_map_from_env_dict(d), fx["dm_f64"])


def test_unpickler_refuses_foreign_globals():
    evil = pickle.dumps(os.system)
    with pytest.raises(pickle.UnpicklingError):
        persistence.load_env_dict(evil)


def test_ingested_env_steps_like_the_golden_episode():
    """the oracle stepped from the ingested description reproduces the golden trajectory."""
    from oracle import orc
    fx = util.load("cfg1_uniform")
    d = persistence.load_env_dict(os.path.join(util.GOLD, "cfg1_persist.pkl"))
    e = orc.OracleEnv(persistence.static_from_env_dict(d, fx["mt_key"], fx["mt_pos"]))
    for t, a in enumerate(fx["actions"][:80]):
        e.step(a)
        np.testing.assert_array_equal(e.state(), util.golden_state(fx, t))
