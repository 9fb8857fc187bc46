"""flatland.envs.line_generators: sparse_line_generator / SparseLineGen (line_generators.py:44-165)"""
from flatland_marl_amd.generators import SparseLineGen, sparse_line_generator  # noqa: F401
