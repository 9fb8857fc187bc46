"""`flatland_cutils` (flatland_cutils/src/main.cpp:9-27 of the reference): TreeObsForRailEnv(max_nodes, max_pred_depth) with
set_env / reset / get_many / get_properties, computed by the HIP kernels"""
from flatland_marl_amd.rail_env import TreeObsForRailEnv  # noqa: F401
