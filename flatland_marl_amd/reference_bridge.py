"""Hand-over from the reference's own objects: a `flatland.envs.rail_env.RailEnv` that has been reset() -> the static
description `BatchedRailEnv` / `RailEnv.from_static` / `fl_load_env` take.  Duck-typed: nothing of flatland is imported
here, the attributes read are the ones `flatland_cutils` itself reads from the env (flatland_cutils/src/loader.cpp:8-120,
207-219) plus the malfunction parameters and the MT19937 state of `env.np_random`."""
import numpy as np

STATIC_KEYS = ("grid", "init_pos", "init_dir", "target", "speed", "earliest", "latest", "T",
               "malf_rate", "malf_min", "malf_max", "mt_key", "mt_pos")


def _malfunction_parameters(env):
    """(rate, min_duration, max_duration) of the env's malfunction generator (malfunction_generators.py:36-53, 56-75)."""
    gen = getattr(env, "malfunction_generator", None)
    mfp = getattr(gen, "MFP", None)
    if mfp is None:
        mpd = getattr(env, "malfunction_process_data", None)   # NoMalfunctionGen / legacy closures: MalfunctionProcessData
        if mpd is not None and len(mpd) >= 3:
            return float(mpd[0]), int(mpd[1]), int(mpd[2])
        return 0.0, 0, 0
    return float(mfp.malfunction_rate), int(mfp.min_duration), int(mfp.max_duration)


def from_reference_env(env):
    """static description (dict of numpy arrays, keys STATIC_KEYS) of a reference RailEnv after reset()."""
    st = env.np_random.get_state()
    if st[0] != "MT19937" or st[3] != 0:
        raise ValueError("env.np_random must be a numpy RandomState (MT19937) without a cached gaussian")
    agents = env.agents
    rate, mn, mx = _malfunction_parameters(env)
    if any(a.initial_position is None or a.target is None for a in agents):
        raise ValueError("the env has to be reset() first (agents without initial position / target)")
    return dict(
        grid=np.asarray(env.rail.grid, dtype=np.uint16),
        init_pos=np.array([a.initial_position for a in agents], dtype=np.int32).reshape(len(agents), 2),
        init_dir=np.array([int(a.initial_direction) for a in agents], dtype=np.int32),
        target=np.array([a.target for a in agents], dtype=np.int32).reshape(len(agents), 2),
        speed=np.array([a.speed_counter.speed for a in agents], dtype=np.float64),
        earliest=np.array([a.earliest_departure for a in agents], dtype=np.int32),
        latest=np.array([a.latest_arrival for a in agents], dtype=np.int32),
        T=np.int32(env._max_episode_steps),
        malf_rate=np.float64(rate), malf_min=np.int32(mn), malf_max=np.int32(mx),
        mt_key=np.asarray(st[1], dtype=np.uint32), mt_pos=np.int32(st[2]),
    )


def dynamic_state_of_reference_env(env):
    """(state int32[A, 12], aux int32[A, 4], elapsed, done_all) of a reference RailEnv mid-episode, in the layout of
    fl_get_state / fl_get_state_aux -- what fl_set_state injects (AgentsLoader's per-call read, loader.cpp:221-327)."""
    A = len(env.agents)
    state = np.zeros((A, 12), dtype=np.int32)
    aux = np.zeros((A, 4), dtype=np.int32)
    for i, a in enumerate(env.agents):
        r, c = a.position if a.position is not None else (-1, -1)
        orow, ocol = a.old_position if a.old_position is not None else (-1, -1)
        sm = a.state_machine
        state[i] = (r, c, int(a.direction), int(a.state), a.malfunction_handler.malfunction_down_counter,
                    a.malfunction_handler.num_malfunctions, a.speed_counter.counter,
                    0 if a.action_saver.saved_action is None else int(a.action_saver.saved_action),
                    -1 if a.arrival_time is None else a.arrival_time, orow, ocol,
                    -1 if a.old_direction is None else int(a.old_direction))
        aux[i] = (-1 if sm.previous_state is None else int(sm.previous_state), int(bool(sm.st_signals.in_malfunction)), 0,
                  int(bool(env.dones[i])))
    return state, aux, int(env._elapsed_steps), bool(env.dones["__all__"])
