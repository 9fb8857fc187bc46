"""flatland.envs.rail_env (rail_env.py:1-35): RailEnv, and the names callers import from here"""
from flatland_marl_amd.rail_env import RailEnv, RailEnvActions, TrainState  # noqa: F401
