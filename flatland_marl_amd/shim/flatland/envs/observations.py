"""flatland.envs.observations: Node and the upstream TreeObsForRailEnv(max_depth, predictor) (observations.py:20-532)"""
from flatland_marl_amd.rail_env import Node, TreeObsUpstream


class TreeObsForRailEnv(TreeObsUpstream):
    def __init__(self, max_depth, predictor=None):
        super().__init__(max_depth=max_depth, pred_depth=-1 if predictor is None else predictor.max_depth)
