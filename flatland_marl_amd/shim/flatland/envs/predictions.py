"""flatland.envs.predictions.ShortestPathPredictorForRailEnv(max_depth) (predictions.py:91-180): here only the holder of the
depth -- the prediction itself is part of the observation kernel"""


class ShortestPathPredictorForRailEnv:
    def __init__(self, max_depth=20):
        self.max_depth = max_depth
