import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
STATE_NAMES = ("row", "col", "dir", "state", "malf", "nmalf", "scount", "saved", "arrival",
               "old_row", "old_col", "old_dir")


def episode_fixtures():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLD, "cfg*.npz")))


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
