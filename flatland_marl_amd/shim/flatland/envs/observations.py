"""flatland.envs.observations: Node and the upstream TreeObsForRailEnv(max_depth, predictor) (observations.py:20-532);
works on this library's RailEnv and, through the state hand-over of flatland_marl_amd.plugin, on any other env object"""
from flatland_marl_amd.rail_env import Node  # noqa: F401
from flatland_marl_amd.plugin import TreeObsUpstream


class TreeObsForRailEnv(TreeObsUpstream):
    def __init__(self, max_depth, predictor=None):
        super().__init__(max_depth, predictor)
