#!/usr/bin/env python3
"""bench.py -- agent-steps/s of the fused hot path (RailEnv.step + tree observations) on MI355X.

One "step" = one lock-step tick of B envs: k_step (synthetic counter-hash actions generated on device,
envs auto-reset at episode end) + the flatland_cutils observation (31 nodes, predictor depth 500) + the
upstream TreeObsForRailEnv dense observation (depth 2, predictor depth 30) for every agent of every env.
Inputs are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).

  python bench.py --gpus 1 --steps 600 --warmup 50
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          (weak scaling: every rank runs B envs)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _cpu_worker(job):
    """one host thread: the CPU oracle (bit-exact C port of the reference path) over the workload's base envs."""
    workload, tree_depth, tree_pred, budget_s, worker = job
    from oracle import orc                      # checker/baseline only; never on the product path
    from flatland_marl_amd import synth, workload as wl
    envs, seed = wl.make_envs(workload, B=len(wl.WORKLOADS[workload]["bases"]))
    oracles = [orc.OracleEnv(e) for e in envs]
    tc = [0] * len(oracles)
    A = oracles[0].A
    agent_steps, t0, b = 0, time.perf_counter(), worker % len(oracles)
    while time.perf_counter() - t0 < budget_s:
        o = oracles[b]
        _, _, done_all = o.step(synth.uniform_actions(seed, b + 1000 * worker, tc[b], A))
        o.obs_cutils(31, 500)
        if tree_depth > 0:
            o.obs_pytree(tree_depth, tree_pred)
        tc[b] += 1
        agent_steps += A
        if done_all:
            key, pos = o.get_rng()
            oracles[b] = orc.OracleEnv(envs[b])
            oracles[b].set_rng(key, pos)
            tc[b] = 0
        b = (b + 1) % len(oracles)
    return agent_steps, time.perf_counter() - t0, A, len(oracles)


def cpu_baseline(workload, tree_depth, tree_pred, budget_s=12.0):
    """The CPU oracle on this host: one thread, then one worker per host core (independent envs, like the GPU's shards).
    Runs BEFORE the GPU is initialised (the workers are forked)."""
    import multiprocessing as mp
    steps1, dt1, A, nb = _cpu_worker((workload, tree_depth, tree_pred, budget_s, 0))
    cores = os.cpu_count() or 1
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(workload, tree_depth, tree_pred, budget_s, w) for w in range(cores)])
    total = sum(r[0] / r[1] for r in res)
    return dict(value=total, unit="agent-steps/s", cores=cores, kind="port",
                single_thread_value=steps1 / dt1,
                sample="one worker per host core (%d), each %.0f s over the %d base envs of %s (step + cutils obs + depth-%d tree), "
                       "%d env-steps in total; single thread: %d env-steps in %.1f s"
                       % (cores, budget_s, nb, workload, tree_depth, sum(r[0] for r in res) // A, steps1 // A, dt1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default: the workload's)")
    ap.add_argument("--tree-depth", type=int, default=2)
    ap.add_argument("--tree-pred", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--event-every", type=int, default=8, help="bracket the kernels of every n-th timed step with HIP events")
    ap.add_argument("--lib", default=None, help="diagnostic: load this build of the C-ABI library instead of the in-tree one (A/B runs)")
    ap.add_argument("--separate", action="store_true", help="launch the two observation builders separately")
    ap.add_argument("--dm-rebuild", action="store_true", help="also rebuild all distance maps every step (BASELINE configs[4])")
    args = ap.parse_args()

    cpu = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, args.tree_depth, args.tree_pred)   # before any GPU initialisation: it forks

    import torch
    from flatland_marl_amd import dist_utils, workload as wl
    from flatland_marl_amd import hip_backend
    if args.lib:
        hip_backend.LIB_PATH = os.path.abspath(args.lib)
    from flatland_marl_amd.hip_backend import BatchedRailEnv

    rank, world, local_rank = dist_utils.init_from_env()
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    W = wl.WORKLOADS[args.workload]
    B = args.envs or W["B"]
    envs, seed = wl.make_envs(args.workload, B=B, rank=rank)
    env = BatchedRailEnv(envs, device=local_rank)
    A = env.A
    stream_base = rank * B

    fused = args.tree_depth > 0 and 0 <= args.tree_pred <= env.pred_depth and not args.separate

    def step_all(ev=None):
        if ev: ev[0].record()
        env.step_synth(seed, stream_base, 0, auto_reset=True)
        if ev: ev[1].record()
        if fused:
            env.obs_both(args.tree_depth, args.tree_pred)
            if ev: ev[2].record()
        else:
            env.obs_cutils()
            if ev: ev[2].record()
            if args.tree_depth > 0:
                env.obs_tree(args.tree_depth, args.tree_pred)
        if ev: ev[3].record()
        if args.dm_rebuild:
            env.rebuild_distance_maps()

    for _ in range(args.warmup):
        step_all()
    env.metrics(reset=True)
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] if k % args.event_every == 0 else None
              for k in range(args.steps)]
    dist_utils.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step_all(events[k])
    torch.cuda.synchronize()
    dist_utils.barrier()
    dt = time.perf_counter() - t0
    dt = dist_utils.max_over_ranks(dt, device=dev)
    env.check()
    metrics = dist_utils.reduce_metrics(env.metrics().clone())
    torch.cuda.synchronize()

    if rank == 0:
        K = args.steps
        seg = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(3)] for e in events if e])  # ms
        ms_step, ms_cutils, ms_tree = seg.mean(0)
        names = ["k_step<synth>", "k_obs<cutils+tree>" if fused else "k_obs<cutils>", "k_obs<tree>"]
        b_step = env.algorithmic_bytes_per_agent_step(False, 0)
        b_cut = env.algorithmic_bytes_per_agent_step(True, 0) - b_step
        b_tree = (env.algorithmic_bytes_per_agent_step(False, args.tree_depth) - b_step) if args.tree_depth > 0 else 0.0
        # fused launch: both outputs, the rail bitmap is read once
        per_agent_bytes = [b_step, b_cut + b_tree - 2.0 * env.H * env.W / A, 0.0] if fused else [b_step, b_cut, b_tree]
        dom = int(np.argmax(seg.mean(0)))
        bytes_per_launch = per_agent_bytes[dom] * B * A
        achieved = bytes_per_launch / (seg.mean(0)[dom] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            ent = tj.get(args.workload, {}).get(names[dom])
            if ent and ent.get("envs") == B:
                traffic = ent["hbm_bytes_per_launch"]
        m = metrics.cpu().numpy()
        out = {
            "metric": "agent-steps/sec (batched envs) + tree-obs build ms/step",
            "value": float(B * A * K * world / dt), "unit": "agent-steps/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": float(dt / K * 1e3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, W["desc"]), "envs_per_gpu": B, "agents": A,
                       "grid": [env.H, env.W], "obs": "cutils(31 nodes, pred 500)" +
                       (" + upstream tree depth %d (pred %d)" % (args.tree_depth, args.tree_pred) if args.tree_depth > 0 else ""),
                       "actions": "counter-hash uniform 0..4 generated on device, auto-reset at episode end",
                       "parallelism": "envs sharded over %d GPU(s), metrics all-reduce only" % world},
            "tree_obs_ms_per_step": float(ms_cutils + ms_tree),
            "kernel_ms": ({"step": float(ms_step), "obs_cutils_tree_fused": float(ms_cutils)} if fused else
                          {"step": float(ms_step), "obs_cutils": float(ms_cutils), "obs_tree": float(ms_tree)}),
            "episodes": int(m[3]), "arrived_agents": int(m[1]), "sum_terminal_reward": int(m[0]), "agent_steps": int(m[2]),
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": float(achieved), "peak": 8000.0, "unit": "GB/s",
                         "frac": float(achieved / 8000.0), "traffic": traffic,
                         "algorithmic_bytes_per_agent_step": {"step": per_agent_bytes[0], names[1]: per_agent_bytes[1],
                                                              "obs_tree_separate": per_agent_bytes[2]},
                         "note": "dependent-gather/latency-bound integer kernel; achieved = algorithmic bytes per launch / mean launch time"},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    env.close()
    dist_utils.shutdown()


if __name__ == "__main__":
    main()
