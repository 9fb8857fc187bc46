"""flatland.envs.malfunction_generators: MalfunctionParameters, ParamMalfunctionGen, NoMalfunctionGen (:19-61)"""
from flatland_marl_amd.rail_env import MalfunctionParameters, NoMalfunctionGen, ParamMalfunctionGen  # noqa: F401
