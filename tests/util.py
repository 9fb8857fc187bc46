import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
STATE_NAMES = ("row", "col", "dir", "state", "malf", "nmalf", "scount", "saved", "arrival",
               "old_row", "old_col", "old_dir")


def episode_fixtures():
    # cfg*: the five BASELINE configs; test14*: the largest Round-2 map (158x158, 425 agents, 41 cities)
    return sorted(os.path.basename(f)[:-4] for pat in ("cfg*.npz", "test14*.npz") for f in glob.glob(os.path.join(GOLD, pat)))


def base_fixtures(prefix="base_"):
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLD, prefix + "*.npz")))


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def golden_state(fx, t):
    """[A, 12] int32 expected agent state after step index t (0-based)."""
    return np.stack([fx["s_" + k][t] for k in STATE_NAMES], axis=1).astype(np.int32)


def static_of(fx, mt_key=None, mt_pos=None):
    d = {k: fx[k] for k in ("grid", "init_pos", "init_dir", "target", "speed", "earliest", "latest", "T",
                            "malf_rate", "malf_min", "malf_max", "mt_key", "mt_pos")}
    if mt_key is not None:
        d["mt_key"], d["mt_pos"] = mt_key, mt_pos
    return d


def actions_of(fx):
    """[steps, A] uint8 actions as the env saw them: fixtures captured through eval_env.parse_actions store the RAW
    policy actions plus the action_required mask; agents without action_required were dropped from the dict (255)."""
    a = np.array(fx["actions"], dtype=np.uint8)
    if "action_required" in getattr(fx, "files", fx):
        a = a.copy()
        a[np.asarray(fx["action_required"]) == 0] = 255
    return a
