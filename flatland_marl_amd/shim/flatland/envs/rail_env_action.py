from flatland_marl_amd.rail_env import RailEnvActions  # noqa: F401
