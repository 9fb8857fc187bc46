"""Synthetic bench workloads: K base envs generated once by the reference (their static descriptions
-- grid, agents, timetable, malfunction parameters, post-reset MT state -- are package data under flatland_marl_amd/data/,
exported from the golden fixtures of the same name by oracle/refharness/capture_golden.py), tiled to B replicas; each replica has its own MT19937 state
(numpy RandomState([replica_id])) and its own counter-hash action stream; envs auto-reset at episode end.

Replica 0 of rank 0 reuses the MT state and action-stream seed of a committed golden episode, so its
trajectory must equal that fixture bit for bit (tests/test_gpu_bench_replica.py).
"""
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

STATIC_KEYS = ("grid", "init_pos", "init_dir", "target", "speed", "earliest", "latest", "T",
               "malf_rate", "malf_min", "malf_max", "mt_key", "mt_pos")

WORKLOADS = {
    # BASELINE.json configs[1]: 256 envs x 30x30 / 20 agents / 3 cities, 1 MI355X
    "cfg2": dict(bases=["cfg2_uniform"] + ["base_cfg2_L%d" % i for i in range(1, 8)], B=256,
                 pinned=("cfg2_uniform", 11), desc="256 envs x 30x30 / 20 agents / 3 cities (Round-2 Test_2)"),
    # BASELINE.json configs[2]: 1024 envs x 35x30 / 80 agents / 5 cities, malfunctions on
    "cfg3": dict(bases=["cfg3_uniform"] + ["base_cfg3_L%d" % i for i in range(1, 4)], B=1024,
                 pinned=("cfg3_uniform", 21), desc="1024 envs x 35x30 / 80 agents / 5 cities (Round-2 Test_4)"),
    # BASELINE.json configs[3] / configs[4] per-GPU shards (512 resp. 256 envs per GPU on an 8-GPU node); one base env each
    "cfg4": dict(bases=["cfg4_fwd_head"], B=512, pinned=("cfg4_fwd_head", 31),
                 desc="512 envs/GPU x 60x60 / 80 agents / 17 cities (Round-2 Test_8)"),
    "cfg5": dict(bases=["cfg5_fwd_head"], B=256, pinned=("cfg5_fwd_head", 41),
                 desc="256 envs/GPU x 150x150 / 400 agents / 37 cities (Round-2 Test_13)"),
    # BASELINE.json configs[0]: the reference's own CPU-runnable case
    "cfg1": dict(bases=["cfg1_uniform"], B=1, pinned=("cfg1_uniform", 1), desc="1 env 30x30 / 7 agents / 2 cities"),
}


def load_static(name):
    z = np.load(os.path.join(DATA, name + ".npz"))
    return {k: z[k] for k in STATIC_KEYS}


def replica_rng(replica_id):
    st = np.random.RandomState([int(replica_id)]).get_state()
    return np.array(st[1], dtype=np.uint32), int(st[2])


def make_envs(workload="cfg2", B=None, rank=0):
    """list of B static env descriptions for this rank (weak scaling: every rank gets B envs) + the stream seed."""
    w = WORKLOADS[workload]
    B = B or w["B"]
    bases = [load_static(n) for n in w["bases"]]
    pinned_name, seed = w["pinned"]
    envs = []
    for b in range(B):
        gid = rank * B + b
        e = dict(bases[gid % len(bases)])
        if gid == 0:
            assert w["bases"][0] == pinned_name  # keeps the fixture's post-reset MT state
        else:
            e["mt_key"], e["mt_pos"] = replica_rng(gid)
        envs.append(e)
    return envs, seed
