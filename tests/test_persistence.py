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


class _Reduce:
    def __init__(self, fn, args):
        self.fn, self.args = fn, args

    def __reduce__(self):
        return self.fn, self.args


@pytest.mark.parametrize("fn,args", [(np.savetxt, ("/tmp/fl_persistence_should_not_exist.txt", [1, 2])),
                                     (np.load, ("/tmp/fl_persistence_should_not_exist.npy", None, True)),
                                     (np.fromfile, ("/etc/hostname",)), (np.frombuffer, (b"abcd", "u1")),
                                     (eval, ("1+1",)), (getattr, ("abc", "upper"))])
def test_unpickler_refuses_numpy_callables_and_other_reduce_targets(fn, args, tmp_path):
    """a crafted env file must not reach numpy.savetxt / numpy.load (file write, unrestricted unpickle) or any
    other callable through REDUCE; only the array / scalar constructors are allowed."""
    evil = pickle.dumps({"grid": _Reduce(fn, args)})
    with pytest.raises(pickle.UnpicklingError, match="refusing to load"):
        persistence.load_env_dict(evil)
    assert not os.path.exists("/tmp/fl_persistence_should_not_exist.txt")


def test_unpickler_still_loads_arrays_scalars_and_dtypes():
    blob = pickle.dumps({"a": np.arange(6, dtype=np.uint16).reshape(2, 3), "s": np.float64(1.5), "i": np.int32(7),
                         "d": np.dtype("float32"), "b": np.array([True, False])})
    d = persistence.load_env_dict(blob)
    np.testing.assert_array_equal(d["a"], np.arange(6).reshape(2, 3))
    assert d["s"] == 1.5 and d["i"] == 7 and d["d"] == np.float32 and d["b"].tolist() == [True, False]


def test_ingested_env_steps_like_the_golden_episode():
    """the oracle stepped from the ingested description reproduces the golden trajectory."""
    from oracle import orc
    fx = util.load("cfg1_uniform")
    d = persistence.load_env_dict(os.path.join(util.GOLD, "cfg1_persist.pkl"))
    e = orc.OracleEnv(persistence.static_from_env_dict(d, fx["mt_key"], fx["mt_pos"]))
    for t, a in enumerate(fx["actions"][:80]):
        e.step(a)
        np.testing.assert_array_equal(e.state(), util.golden_state(fx, t))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_reset_of_a_file_env_redraws_the_timetable_like_the_reference(tag):
    """RailEnvPersister.load_new(file) + env.reset() of the REAL reference from a known MT19937 state (golden
    persist_reset_cfg1.npz; a: reset(), b: reset(regenerate_rail=False, regenerate_schedule=True)): same rail and line, a NEW
    timetable and max_episode_steps drawn by timetable_generator without hints (num_cities = 2), the stream advanced."""
    from flatland_marl_amd import generators
    g = util.load("persist_reset_cfg1")
    d = persistence.load_env_dict(os.path.join(util.GOLD, "cfg1_persist.pkl"))
    st = persistence.static_from_env_dict(d, g[tag + "_mt_key0"], g[tag + "_mt_pos0"])
    out = generators.redraw_timetable(st, st["mt_key"], st["mt_pos"], num_cities=2)
    np.testing.assert_array_equal(out["earliest"], g[tag + "_earliest"])
    np.testing.assert_array_equal(out["latest"], g[tag + "_latest"])
    assert int(out["T"]) == int(g[tag + "_T"])
    np.testing.assert_array_equal(out["mt_key"], g[tag + "_mt_key1"])
    assert int(out["mt_pos"]) == int(g[tag + "_mt_pos1"])
    np.testing.assert_array_equal(out["grid"], g[tag + "_grid"])
    np.testing.assert_array_equal(out["init_pos"], g[tag + "_init_pos"])
    np.testing.assert_array_equal(out["target"], g[tag + "_target"])
    # FileMalfunctionGen (malfunction_generators.py:63-74): the parameters of the file
    assert [float(st["malf_rate"]), float(st["malf_min"]), float(st["malf_max"])] == g[tag + "_malf"].tolist()
    # the file's own timetable differs (it was drawn with the generator's hints: 2 cities there too, but from another state)
    assert not np.array_equal(out["earliest"], st["earliest"])
