from flatland_marl_amd.rail_env import EnvAgent  # noqa: F401
