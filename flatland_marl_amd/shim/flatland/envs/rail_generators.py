"""flatland.envs.rail_generators: the sparse generator of Round 2 (rail_generators.py:161-292), native host code here"""
from flatland_marl_amd.generators import SparseRailGen, sparse_rail_generator  # noqa: F401
