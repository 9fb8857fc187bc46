#!/usr/bin/env python3
"""Known-answer vectors for MotionCheck (envs/agent_chains.py): runs the reference's own scenario
builders (create_test_agents :302-330, create_test_agents2 :333-415) and a seeded fuzz -- including
agents sharing a cell, which rail_env.py:599-602 can produce -- through the REAL reference class and
stores (cur node, next node, can_move) triples in tests/golden/motioncheck.npz.  Build-container only."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "stubs"), "/root/reference/flatland-rl", REPO]
import numpy as np  # noqa: E402
from flatland.envs import agent_chains as ac  # noqa: E402


class Recorder(ac.MotionCheck):
    def __init__(self):
        super().__init__()
        self.rec = []

    def addAgent(self, iAg, rc1, rc2, xlabel=None):
        self.rec.append((iAg, rc1, rc2))
        super().addAgent(iAg, rc1, rc2, xlabel=xlabel)


def node_id(rc, i):
    return 100000 + i if rc is None else rc[0] * 100 + rc[1]


def run_case(agents):
    """agents: list of (idx, rc1|None, rc2|None) in handle order -> (cur, nxt, can_move) arrays."""
    mc = ac.MotionCheck()
    for i, rc1, rc2 in agents:
        mc.addAgent(i, rc1, rc2)
    mc.find_conflicts()
    cur = [node_id(rc1, i) for i, rc1, rc2 in agents]
    nxt = [node_id(rc2, i) for i, rc1, rc2 in agents]
    can = [bool(mc.check_motion(i, rc1)) for i, rc1, rc2 in agents]
    return cur, nxt, can


def scenario(builder):
    r = Recorder()
    builder(r)
    # the builders use arbitrary agent ids; re-index densely in ascending id order = env handle order
    rec = sorted(r.rec, key=lambda x: x[0])
    remap = {old: new for new, (old, _, _) in enumerate(rec)}
    return [(remap[i], a, b) for i, a, b in rec]


def fuzz_case(rng, stacked):
    n = int(rng.integers(1, 14))
    side = int(rng.integers(2, 5))
    cells = [(r, c) for r in range(side) for c in range(side)]
    agents = []
    used = []
    for i in range(n):
        if rng.random() < 0.2:
            rc1 = None
        else:
            if stacked and used and rng.random() < 0.25:
                rc1 = used[int(rng.integers(len(used)))]
            else:
                free = [c for c in cells if c not in used]
                if not free:
                    rc1 = None
                else:
                    rc1 = free[int(rng.integers(len(free)))]
            if rc1 is not None:
                used.append(rc1)
        if rc1 is None:
            rc2 = None if rng.random() < 0.5 else cells[int(rng.integers(len(cells)))]
        else:
            if rng.random() < 0.3:
                rc2 = rc1
            else:
                d = [(-1, 0), (0, 1), (1, 0), (0, -1)][int(rng.integers(4))]
                rc2 = (rc1[0] + d[0], rc1[1] + d[1])
                if not (0 <= rc2[0] < side and 0 <= rc2[1] < side):
                    rc2 = rc1
        agents.append((i, rc1, rc2))
    return agents


if __name__ == "__main__":
    cases = [scenario(ac.create_test_agents), scenario(ac.create_test_agents2)]
    rng = np.random.default_rng(7)
    for k in range(3000):
        cases.append(fuzz_case(rng, stacked=(k % 3 == 0)))
    cur, nxt, can, off = [], [], [], [0]
    for ag in cases:
        c, n, m = run_case(ag)
        cur += c; nxt += n; can += m; off.append(len(cur))
    out = os.path.join(REPO, "tests", "golden", "motioncheck.npz")
    np.savez_compressed(out, cur=np.array(cur, np.int32), nxt=np.array(nxt, np.int32),
                        can_move=np.array(can, np.uint8), offsets=np.array(off, np.int32))
    print("cases", len(cases), "agents", len(cur), "->", os.path.getsize(out) // 1024, "KB")
