from flatland_marl_amd.rail_env import TrainState  # noqa: F401
