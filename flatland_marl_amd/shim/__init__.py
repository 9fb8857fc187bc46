"""Opt-in import shim: exposes the module names the reference's harness imports -- `flatland_cutils`,
`flatland.envs.rail_env`, `flatland.envs.step_utils.states`, `flatland.envs.rail_generators`, ... (solution/eval_env.py:1-3,
solution/demo.py:4-12,39) -- on top of the MI355X library, so that `solution/eval_env.py`, `solution/demo.py` and
`solution/plfActor.py` run unchanged:

    import flatland_marl_amd.shim as shim; shim.install()        # or PYTHONPATH=<repo>/flatland_marl_amd/shim
    from flatland.envs.rail_env import RailEnv, TrainState
    from flatland_cutils import TreeObsForRailEnv as TreeCutils

Only the hot path and its callers' entry points exist here; the renderer, the redis evaluator and the
other generators of flatland-rl are not rebuilt -- their names raise a clear error when used."""
import os
import sys

SHIM_DIR = os.path.dirname(os.path.abspath(__file__))


def install():
    """put the shim packages first on sys.path (a real flatland-rl / flatland_cutils installation is shadowed)"""
    for name in [m for m in sys.modules if m == "flatland" or m.startswith("flatland.") or m == "flatland_cutils"]:
        del sys.modules[name]
    if SHIM_DIR in sys.path:
        sys.path.remove(SHIM_DIR)
    sys.path.insert(0, SHIM_DIR)
    return SHIM_DIR
