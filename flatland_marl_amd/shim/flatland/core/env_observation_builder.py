from flatland_marl_amd.rail_env import ObservationBuilder  # noqa: F401
