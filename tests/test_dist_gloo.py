"""CPU, world_size 2, gloo: the N > 1 harness path -- contiguous env sharding, the int64 metrics (+ score sums)
all-reduce(SUM) and the max-over-ranks timing -- plus the per-replica seeding of the workload."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from flatland_marl_amd import dist_utils, workload as wl
    r, w, lr = dist_utils.init_from_env(backend="gloo")
    assert (r, w, lr) == (rank, world, rank)
    lo, hi = dist_utils.shard_range(11, rank, world)
    metrics = torch.tensor([-(rank + 1) * 10, rank + 1, (hi - lo) * 20, 1], dtype=torch.int64)
    dist_utils.barrier()
    red = dist_utils.reduce_metrics(metrics.clone())
    # with the evaluator's score sums in the same all-reduce: `metrics` is reduced IN PLACE on this path too
    m2 = metrics.clone()
    scores = torch.tensor([0.25 + rank, 0.5 * (rank + 1), float(rank + 1)], dtype=torch.float64)
    m2_ret, sc = dist_utils.reduce_metrics(m2, scores)
    assert m2_ret is m2
    tmax = dist_utils.max_over_ranks(1.0 + rank)
    per_rank = dist_utils.gather_agent_steps(metrics)
    envs, seed = wl.make_envs("cfg2", B=3, rank=rank)
    out[rank] = dict(shard=(lo, hi), red=red.tolist(), red2=m2.tolist(), sc=sc.tolist(), tmax=tmax, per_rank=per_rank, keys=[int(e["mt_key"][1]) for e in envs], seed=seed)
    dist_utils.shutdown()


def test_metrics_allreduce_and_sharding_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0]["shard"] == (0, 6) and out[1]["shard"] == (6, 11)
    assert out[0]["red"] == out[1]["red"] == [-30, 3, 11 * 20, 2]
    assert out[0]["red2"] == out[1]["red2"] == [-30, 3, 11 * 20, 2]
    assert out[0]["sc"] == out[1]["sc"] == [0.25 + 1.25, 0.5 + 1.0, 3.0]      # (exact: multiples of 2**-32)
    assert out[0]["tmax"] == out[1]["tmax"] == 2.0
    assert out[0]["per_rank"] == out[1]["per_rank"] == [6 * 20, 5 * 20]
    # weak scaling: rank r owns global replicas [3r, 3r+3), each with its own MT19937 state
    assert len(set(out[0]["keys"]) | set(out[1]["keys"])) == 6
    assert out[0]["seed"] == out[1]["seed"]


def test_shard_range_covers_everything():
    from flatland_marl_amd import dist_utils
    for n in (1, 7, 256, 4096):
        for world in (1, 2, 3, 8):
            spans = [dist_utils.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_replica_rng_matches_numpy_init_by_array():
    from flatland_marl_amd import workload as wl
    key, pos = wl.replica_rng(5)
    st = np.random.RandomState([5]).get_state()
    np.testing.assert_array_equal(key, st[1])
    assert pos == st[2]
