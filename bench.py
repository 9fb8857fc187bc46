#!/usr/bin/env python3
"""bench.py -- agent-steps/s of the fused hot path (RailEnv.step + tree observations) on MI355X.

One "step" = one lock-step tick of B envs: k_step (synthetic counter-hash actions generated on device, envs
auto-reset at episode end) + the flatland_cutils observation (31 nodes, predictor depth 500) + the upstream
TreeObsForRailEnv dense observation (depth 2, predictor depth 30) for every agent of every env, the two observation
builders in one launch.  Inputs are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).

The headline (`value`) is cfg2 = BASELINE.json configs[1], the configuration the north-star target is quoted on.
At N = 1 the same line carries a `workloads` object with the other single-GPU configurations measured the same way:
cfg3 (configs[2]: 1024 envs, malfunctions on, DEPTH-3 tree), the per-GPU shards of cfg4 (configs[3]: 512 envs) and of
cfg5 (configs[4]: 256 envs, DEPTH-3 tree, and the distance maps of every env that just reset rebuilt on the GPU, as
RailEnv.reset() does).

Before the timed region the replicas are de-phased: env b is reset once at a step drawn from [0, T), so the batch is in
its steady-state mix of episode phases (not "every env in its first steps after a synchronised reset").

  python bench.py --gpus 1 --steps 600 --warmup 50
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          (weak scaling: every rank runs B envs)
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KERNEL_SOURCES = ("fl_obs.hip", "fl_obs.h", "fl_step.hip", "fl_step_body.h", "fl_dmap.hip", "fl_internal.h", "build.sh")  # sources + compile flags
EXTRA_WORKLOADS = (  # (workload, tree depth, distance-map rebuild at auto-reset, timed steps)
    ("cfg3", 3, False, 150), ("cfg4", 2, False, 150), ("cfg5", 3, True, 100))


def kernel_source_sha():
    """identifies the kernel build a stored PMC measurement (profiles/pmc_traffic.json) belongs to"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "flatland_marl_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


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


def run_workload(name, tree_depth, tree_pred, envs_per_gpu, steps, warmup, rank, world, local_rank, dephase=True,
                 dm_rebuild=False, event_every=8, separate=False):
    """time `steps` steps of one workload on this rank's GPU; returns the measurement (rank 0 only) or None"""
    import torch
    from flatland_marl_amd import dist_utils, workload as wl
    from flatland_marl_amd.hip_backend import BatchedRailEnv
    dev = torch.device("cuda", local_rank)
    W = wl.WORKLOADS[name]
    B = envs_per_gpu or W["B"]
    envs, seed = wl.make_envs(name, B=B, rank=rank)
    env = BatchedRailEnv(envs, device=local_rank)
    A = env.A
    stream_base = rank * B
    fused = tree_depth > 0 and 0 <= tree_pred <= env.pred_depth and not separate

    def step_all(ev=None):
        if ev: ev[0].record()
        env.step_synth(seed, stream_base, 0, auto_reset=True)
        if ev: ev[1].record()
        if fused:
            env.obs_both(tree_depth, tree_pred)
            if ev: ev[2].record()
        else:
            env.obs_cutils()
            if ev: ev[2].record()
            if tree_depth > 0:
                env.obs_tree(tree_depth, tree_pred)
        if ev: ev[3].record()
        if dm_rebuild:      # RailEnv.reset() recomputes the distance map: here for exactly the envs whose episode just ended
            env.rebuild_distance_maps(env.done_all)
        if ev: ev[4].record()

    # de-phase the replicas (untimed): env b starts over once, at its own offset in [0, T_b)
    dephase_steps = 0
    if dephase:
        rs = np.random.RandomState(12345 + rank)
        offs = np.array([rs.randint(0, int(e["T"])) for e in envs])
        dephase_steps = int(max(int(e["T"]) for e in envs))
        by_step = {}
        for b, o in enumerate(offs):
            by_step.setdefault(int(o), []).append(b)
        for s in range(dephase_steps):
            env.step_synth(seed, stream_base, 0, auto_reset=True)
            if s in by_step:
                m = np.zeros(B, dtype=np.uint8)
                m[by_step[s]] = 1
                env.reset(m, fresh=True)
    for _ in range(warmup):
        step_all()
    env.metrics(reset=True)
    st0, _ = env.state()
    on_map0 = float((st0[:, :, 0] >= 0).sum(1).mean())
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] if k % event_every == 0 else None for k in range(steps)]
    dist_utils.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step_all(events[k])
    torch.cuda.synchronize()
    dist_utils.barrier()
    dt = time.perf_counter() - t0
    dt = dist_utils.max_over_ranks(dt, device=dev)
    env.check()
    local_metrics = env.metrics().clone()
    per_rank = dist_utils.gather_agent_steps(local_metrics, device=dev)
    metrics = dist_utils.reduce_metrics(local_metrics.clone())
    torch.cuda.synchronize()
    out = None
    if rank == 0:
        seg = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(4)] for e in events if e])  # ms
        ms_step, ms_obs1, ms_obs2, ms_dm = seg.mean(0)
        names = ["k_step<synth>", "k_obs<cutils+tree>" if fused else "k_obs<cutils>", "k_obs<tree>", "dm rebuild (masked)"]
        b_step = env.algorithmic_bytes_per_agent_step(False, 0)
        b_cut = env.algorithmic_bytes_per_agent_step(True, 0) - b_step
        b_tree = (env.algorithmic_bytes_per_agent_step(False, tree_depth) - b_step) if tree_depth > 0 else 0.0
        # fused launch: both outputs, the rail bitmap is read once
        per_agent_bytes = [b_step, b_cut + b_tree - 2.0 * env.H * env.W / A, 0.0] if fused else [b_step, b_cut, b_tree]
        dom = int(np.argmax(seg.mean(0)[:3]))
        bytes_per_launch = per_agent_bytes[dom] * B * A
        achieved = bytes_per_launch / (seg.mean(0)[dom] * 1e-3) / 1e9
        # HBM traffic of the dominant kernel: a STORED rocprofv3 PMC measurement (profiles/pmc_traffic.json), valid only for the
        # kernel sources it was taken on -- null when the sources changed since or no entry exists for this workload
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            ent = json.load(open(tpath)).get("%s_d%d" % (name, tree_depth), {}).get(names[dom])
            if ent and ent.get("envs") == B and ent.get("kernel_source_sha") == kernel_source_sha():
                traffic = ent["hbm_bytes_per_launch"]
                traffic_src = "stored PMC measurement %s (FETCH_SIZE x2 + WRITE_SIZE, separate passes), kernel sources %s" % (
                    ent.get("tag"), ent["kernel_source_sha"])
        m = metrics.cpu().numpy()
        st1, _ = env.state()
        out = {
            "value": float(B * A * steps * world / dt), "unit": "agent-steps/s", "ms_per_step": float(dt / steps * 1e3),
            "steps": steps, "warmup": warmup, "dephase_steps": dephase_steps,
            "config": {"workload": "%s: %s" % (name, W["desc"]), "envs_per_gpu": B, "agents": A, "grid": [env.H, env.W],
                       "obs": "cutils(31 nodes, pred 500)" + (" + upstream tree depth %d (pred %d)" % (tree_depth, tree_pred) if tree_depth > 0 else ""),
                       "actions": "counter-hash uniform 0..4 generated on device, auto-reset at episode end",
                       "dm_rebuild": "distance maps + static tables of the envs that just reset rebuilt every step (masked, on device)" if dm_rebuild else "at commit only",
                       "parallelism": "envs sharded over %d GPU(s), metrics all-reduce only" % world},
            "tree_obs_ms_per_step": float(ms_obs1 + ms_obs2),
            "kernel_ms": dict({"step": float(ms_step)},
                              **({"obs_cutils_tree_fused": float(ms_obs1)} if fused else {"obs_cutils": float(ms_obs1), "obs_tree": float(ms_obs2)}),
                              **({"dm_rebuild_masked": float(ms_dm)} if dm_rebuild else {})),
            "episodes": int(m[3]), "arrived_agents": int(m[1]), "sum_terminal_reward": int(m[0]), "agent_steps": int(m[2]),
            "agent_steps_per_rank": [int(v) for v in per_rank], "world_size": world,
            "on_map_agents_per_env": [on_map0, float((st1[:, :, 0] >= 0).sum(1).mean())],
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": float(achieved), "peak": 8000.0, "unit": "GB/s",
                         "frac": float(achieved / 8000.0), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms": float(seg.mean(0)[dom]),
                         "algorithmic_bytes_per_agent_step": {"step": per_agent_bytes[0], names[1]: per_agent_bytes[1],
                                                              "obs_tree_separate": per_agent_bytes[2]},
                         "note": "dependent-gather/latency-bound integer kernel; achieved = algorithmic bytes per launch / mean launch "
                                 "time (HIP events on the launch stream, every %d-th timed step)" % event_every},
        }
    env.close()
    return out


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
    ap.add_argument("--no-extra-workloads", action="store_true", help="skip the `workloads` object (cfg3 / cfg4 / cfg5 at N = 1)")
    ap.add_argument("--no-dephase", action="store_true", help="start the timed region from a synchronised reset (as round 1 did)")
    ap.add_argument("--event-every", type=int, default=8, help="bracket the kernels of every n-th timed step with HIP events")
    ap.add_argument("--lib", default=None, help="diagnostic: load this build of the C-ABI library instead of the in-tree one (A/B runs)")
    ap.add_argument("--separate", action="store_true", help="launch the two observation builders separately")
    ap.add_argument("--dm-rebuild", action="store_true", help="rebuild the distance maps of the envs that just reset, every step (BASELINE configs[4])")
    args = ap.parse_args()

    single = int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1
    cpu = None
    if single and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, args.tree_depth, args.tree_pred)   # before any GPU initialisation: it forks

    import torch
    from flatland_marl_amd import dist_utils
    from flatland_marl_amd import hip_backend
    if args.lib:
        hip_backend.LIB_PATH = os.path.abspath(args.lib)

    rank, world, local_rank = dist_utils.init_from_env()
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    if os.environ.get("FL_DIST_BACKEND") == "gloo":      # rehearsal: several ranks share the GPUs that exist
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)

    head = run_workload(args.workload, args.tree_depth, args.tree_pred, args.envs, args.steps, args.warmup, rank, world, local_rank,
                        dephase=not args.no_dephase, dm_rebuild=args.dm_rebuild, event_every=args.event_every, separate=args.separate)
    extra = {}
    if single and not args.no_extra_workloads and args.workload == "cfg2":
        for name, depth, rebuild, steps in EXTRA_WORKLOADS:
            r = run_workload(name, depth, args.tree_pred, None, steps, 20, rank, world, local_rank, dephase=not args.no_dephase,
                             dm_rebuild=rebuild, event_every=args.event_every)
            extra["%s_d%d%s" % (name, depth, "_dmrebuild" if rebuild else "")] = r
    if rank == 0:
        out = {"metric": "agent-steps/sec (batched envs) + tree-obs build ms/step", "value": head["value"], "unit": head["unit"],
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic"}
        out.update({k: v for k, v in head.items() if k not in out})
        out["kernel_source_sha"] = kernel_source_sha()
        if extra:
            out["workloads"] = extra
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    dist_utils.shutdown()


if __name__ == "__main__":
    main()
