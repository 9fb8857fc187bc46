"""flatland.envs.persistence.RailEnvPersister.load_new (persistence.py:105-129) for .pkl env files"""
import numpy as np

from flatland_marl_amd import persistence as _p
from flatland_marl_amd.rail_env import RailEnv


class RailEnvPersister:
    @classmethod
    def load_env_dict(cls, filename, load_from_package=None):
        if load_from_package is not None:
            raise NotImplementedError("load_from_package is not supported: pass a file path")
        return _p.load_env_dict(filename)

    @classmethod
    def load_new(cls, filename, load_from_package=None):
        env_dict = cls.load_env_dict(filename, load_from_package)
        st = np.random.RandomState().get_state()       # the format carries no RNG state; the reference re-seeds at load too
        # reset() on this env draws the timetable again, like the reference's rail_from_file / line_from_file env does
        env = RailEnv.from_static(_p.static_from_env_dict(env_dict, st[1], st[2]), from_file=True)
        return env, env_dict
