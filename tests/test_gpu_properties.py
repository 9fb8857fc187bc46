"""GPU, BASELINE.json full size (cfg2: 256 envs x 30x30 x 20 agents): size-independent properties of the hot path --
replica independence (an env inside the batch == the same env alone), determinism, structural invariants of the
observation tensors, metrics consistency -- plus an RCCL sanity check of the multi-GPU harness at world size 1."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(B, rank=0, workload="cfg2"):
    from flatland_marl_amd import workload as wl
    from flatland_marl_amd.hip_backend import BatchedRailEnv
    envs, seed = wl.make_envs(workload, B=B, rank=rank)
    return BatchedRailEnv(envs), envs, seed


def test_full_size_batch_properties_and_replica_independence():
    from flatland_marl_amd.hip_backend import BatchedRailEnv
    B, steps = 256, 240
    env, envs, seed = _mk(B)
    A = env.A
    picks = [0, 37, 101, 200, 255]
    solo = [BatchedRailEnv([envs[b]]) for b in picks]
    sum_rew = np.zeros(B, dtype=np.int64)
    episodes = 0
    arrived = 0
    for t in range(steps):
        rew, done, done_all = env.step_synth(seed, 0, 0, auto_reset=True)
        o = env.obs_cutils()
        tr = env.obs_tree(2, 30)
        st, el = env.state()
        rew_h, done_h, da_h = rew.cpu().numpy(), done.cpu().numpy(), done_all.cpu().numpy()
        # --- invariants
        on_map = st[:, :, 0] >= 0
        cells = st[:, :, 0] * env.W + st[:, :, 1]
        grids = np.stack([np.asarray(e["grid"]).reshape(-1) for e in envs])
        assert (np.take_along_axis(grids, np.where(on_map, cells, 0), axis=1)[on_map] != 0).all()      # agents sit on rail
        assert ((st[:, :, 3] >= 3) & (st[:, :, 3] <= 5))[on_map].all()                                  # on-map states only
        assert (st[:, :, 0][st[:, :, 3] <= 2] == -1).all() and (st[:, :, 0][st[:, :, 3] == 6] == -1).all()
        assert (rew_h[~da_h.astype(bool)] == 0).all()                                                  # sparse reward
        assert (done_h[da_h.astype(bool)] == 1).all()
        assert ((done_h == 1) == ((st[:, :, 3] == 6) | da_h.astype(bool)[:, None])).all()
        sum_rew += rew_h.sum(1)
        episodes += int(da_h.sum())
        arrived += int((st[:, :, 3] == 6)[da_h.astype(bool)].sum())
        if t % 40 == 0:
            adj = o["adjacency"].cpu().numpy()
            no = o["node_order"].cpu().numpy()
            eo = o["edge_order"].cpu().numpy()
            forest = o["forest"].cpu().numpy()
            real = adj[..., 0] >= 0
            assert (adj[..., 1][real] == np.broadcast_to(np.arange(1, 31), adj[..., 1].shape)[real]).all()
            assert (adj[..., 0][real] < adj[..., 1][real]).all()                                       # BFS numbering
            assert (adj[~real] == -2).all()
            par_order = np.take_along_axis(no, np.where(real, adj[..., 0], 0), axis=2)
            assert (eo[real] == par_order[real]).all() and (eo[~real] == -2).all()
            assert (no[:, :, 0] >= 1).all()                                                             # the root has children
            assert np.isfinite(forest).all() and (forest >= -1).all()
            tree = tr.cpu().numpy()
            assert (np.isneginf(tree).all(axis=-1) | ~np.isneginf(tree).any(axis=-1)).all()            # rows all -inf or none
        # --- replica independence: the same env alone gives the same state and observations
        for k, b in enumerate(picks):
            s_env = solo[k]
            s_env.step_synth(seed, b, 0, auto_reset=True)
            s_st, _ = s_env.state()
            np.testing.assert_array_equal(s_st[0], st[b], err_msg=f"replica {b} step {t}")
            if t % 20 == 0:
                s_o = s_env.obs_cutils()
                for key in ("agent_attr", "forest", "adjacency", "node_order", "edge_order", "valid_actions"):
                    np.testing.assert_array_equal(s_o[key].cpu().numpy()[0], o[key].cpu().numpy()[b], err_msg=f"{key} replica {b}")
                np.testing.assert_array_equal(s_env.obs_tree(2, 30).cpu().numpy()[0], tr.cpu().numpy()[b])
            else:
                s_env.obs_cutils()   # keeps the sticky deadlock flags in step
    env.check()
    m = env.metrics().cpu().numpy()
    assert m[2] == B * A * steps and m[3] == episodes and m[0] == sum_rew.sum() and m[1] == arrived


def test_determinism_two_runs_bitwise_identical():
    outs = []
    for _ in range(2):
        env, _, seed = _mk(64)
        for t in range(120):
            env.step_synth(seed, 0, 0, auto_reset=True)
            o = env.obs_cutils()
        st, _ = env.state()
        outs.append((st.copy(), {k: v.cpu().numpy().copy() for k, v in o.items()}, env.rng_state()))
        env.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        np.testing.assert_array_equal(outs[0][1][k], outs[1][1][k])
    np.testing.assert_array_equal(outs[0][2][0], outs[1][2][0])


def test_rccl_harness_world_size_one():
    """backend "nccl" is RCCL on ROCm: init, barrier, the int64[4] all-reduce and the max-over-ranks path of bench.py."""
    import torch
    import torch.distributed as dist
    from flatland_marl_amd import dist_utils
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        env, _, seed = _mk(8)
        for _ in range(5):
            env.step_synth(seed, 0, 0, auto_reset=True)
        dist_utils.barrier()
        m = dist_utils.reduce_metrics(env.metrics().clone())
        assert int(m[2].item()) == 8 * env.A * 5
        assert dist_utils.max_over_ranks(1.25, device=torch.device("cuda", 0)) == 1.25
    finally:
        dist.destroy_process_group()
