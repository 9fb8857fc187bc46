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


# ---- a duck-typed env that replays a golden episode's per-step agent states: the attribute surface flatland_cutils reads from a
# caller-owned env (flatland_cutils/src/loader.cpp:8-120, 207-219, 329-333) and nothing else of flatland
class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


STATE_STR = ("TrainState.WAITING", "TrainState.READY_TO_DEPART", "TrainState.MALFUNCTION_OFF_MAP", "TrainState.MOVING",
             "TrainState.STOPPED", "TrainState.MALFUNCTION", "TrainState.DONE")


class DuckEnv:
    def __init__(self, fx, string_states=False):
        self.fx = fx
        self.string_states = string_states
        grid = np.array(fx["grid"], dtype=np.uint16)
        self.rail = _NS(grid=grid)
        self.height, self.width = grid.shape
        self._max_episode_steps = int(fx["T"])
        A = len(fx["init_dir"])
        self.agents = []
        for i in range(A):
            speed = float(fx["speed"][i])
            self.agents.append(_NS(
                handle=i, initial_position=tuple(int(v) for v in fx["init_pos"][i]), initial_direction=int(fx["init_dir"][i]),
                target=tuple(int(v) for v in fx["target"][i]), earliest_departure=int(fx["earliest"][i]),
                latest_arrival=int(fx["latest"][i]), moving=False,
                speed_counter=_NS(speed=speed, max_count=int(1 / speed) - 1, counter=0),
                malfunction_handler=_NS(malfunction_down_counter=0, num_malfunctions=0),
                state_machine=_NS(st_signals=_NS(in_malfunction=False))))
        self.goto(0)

    def get_num_agents(self):
        return len(self.agents)

    def goto(self, T):
        """the agents as they are after T steps (T = 0: after reset())"""
        fx = self.fx
        if T == 0:
            # snap0: the agents after reset(), one row per field in the alphabetical order of the field names
            order = sorted(STATE_NAMES)
            rows = np.stack([fx["snap0"][order.index(k)] for k in STATE_NAMES], axis=1) if "snap0" in fx.files else None
        else:
            rows = golden_state(fx, T - 1)
        malf_before = fx["s_malf"][T - 2] if T >= 2 else np.zeros(len(self.agents), dtype=np.int32)
        for i, a in enumerate(self.agents):
            if rows is None:
                r = (-1, -1, a.initial_direction, 0, 0, 0, 0, 0, -1, -1, -1, -1)
            else:
                r = [int(v) for v in rows[i]]
            a.position = None if r[0] < 0 else (r[0], r[1])
            a.direction = r[2]
            a.state = STATE_STR[r[3]] if self.string_states else r[3]
            a.malfunction_handler.malfunction_down_counter = r[4]
            a.malfunction_handler.num_malfunctions = r[5]
            a.speed_counter.counter = r[6]
            a.arrival_time = None if r[8] < 0 else r[8]
            a.old_position = None if r[9] < 0 else (r[9], r[10])
            a.old_direction = None if r[11] < 0 else r[11]
            # st_signals.in_malfunction was evaluated before the counter ticked down (rail_env.py:369-395, 622-624)
            a.state_machine.st_signals.in_malfunction = bool(r[4] > 0 or (T >= 1 and malf_before[i] == 1))
        self._elapsed_steps = T
