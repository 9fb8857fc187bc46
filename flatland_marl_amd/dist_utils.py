"""Multi-GPU harness pieces: one process per GPU, envs sharded contiguously across ranks (no data-path
collective: envs are independent), one all-reduce(SUM) of the int64[7] episode metrics + evaluator score sums and one
all-reduce(MAX) of the timed region per measurement window.  backend "nccl" is RCCL on ROCm; the
same code runs on "gloo"/CPU tensors in the tests."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """returns (rank, world_size, local_rank); initialises torch.distributed when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:   # FL_DIST_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks
            backend = os.environ.get("FL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_range(n_total, rank, world):
    """contiguous shard [lo, hi) of n_total envs for this rank (the remainder goes to the first ranks)."""
    q, r = divmod(n_total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


SCORE_SCALE = float(2 ** 32)   # fixed point of the evaluator's score sums inside the int64 metrics vector


def reduce_metrics(metrics, scores=None):
    """ONE all-reduce(SUM) of the int64 metrics vector (sum reward, arrived, agent-steps, episodes); `metrics` holds the job's
    totals afterwards (in place, with or without `scores`).  With `scores` (the float64[3] tensor of BatchedRailEnv.scores(): sum
    of normalized rewards, sum of completion ratios, their own episode count) the two score sums ride in the same vector as
    2**-32 fixed point -- integer sums do not depend on the order the ranks are added in, at the price of rounding every rank's
    sum to a multiple of 2**-32 (the job-level means are therefore within world_size * 2**-33 / episodes of the float64 sums
    fl_scores documents, not bit-equal to them) -- together with the scores' episode count, and the call returns
    (metrics int64[4], scores float64[3]) of the whole job (flatland/evaluators/service.py:900-913 divides the sums by the
    number of episodes: scores[0] / scores[2], scores[1] / scores[2])."""
    if scores is None:
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(metrics, op=dist.ReduceOp.SUM)
        return metrics
    sc = scores.to(torch.float64).to(metrics.device)
    fixed = torch.round(sc[:2] * SCORE_SCALE).to(torch.int64)
    vec = torch.cat([metrics.to(torch.int64), fixed, torch.round(sc[2:3]).to(torch.int64)])
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    metrics.copy_(vec[:4])
    return metrics, torch.cat([vec[4:6].to(torch.float64) / SCORE_SCALE, vec[6:7].to(torch.float64)])


def gather_agent_steps(metrics, device=None):
    """all-gather of every rank's agent-step counter (metrics[2]): lets a scaling run confirm that N ranks took part and
    that the shards were equal (weak scaling).  Reporting only, like reduce_metrics."""
    mine = metrics[2:3].clone()
    if dist.is_available() and dist.is_initialized():
        parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, mine)
        return [int(p.item()) for p in parts]
    return [int(mine.item())]


def max_over_ranks(seconds, device=None):
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
