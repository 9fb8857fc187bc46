"""`flatland_cutils` (flatland_cutils/src/main.cpp:9-27 of the reference): TreeObsForRailEnv(max_nodes, max_pred_depth) with
set_env / reset / get_many / get_properties, computed by the HIP kernels; set_env takes ANY env object (treeobs.cpp:17-21)"""
from flatland_marl_amd.plugin import TreeObsForRailEnv  # noqa: F401
