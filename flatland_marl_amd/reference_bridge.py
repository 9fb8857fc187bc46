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


def _rng_state_of(env, required):
    rs = getattr(env, "np_random", None)
    if rs is None or not hasattr(rs, "get_state"):
        if required:
            raise ValueError("env.np_random must be a numpy RandomState (MT19937)")
        return np.zeros(624, dtype=np.uint32), 624       # observation-only use: the stream is never drawn from
    st = rs.get_state()
    if st[0] != "MT19937" or st[3] != 0:
        if required:
            raise ValueError("env.np_random must be a numpy RandomState (MT19937) without a cached gaussian")
        return np.zeros(624, dtype=np.uint32), 624       # observation-only use: the stream is never drawn from
    return np.asarray(st[1], dtype=np.uint32), int(st[2])


AGENT_STATIC_KEYS = ("init_pos", "init_dir", "target", "speed", "earliest", "latest", "T")


def agents_static_of_env(env):
    """the agents' line and timetable and the episode length -- the part of the static description flatland_cutils reads again on
    every call (Agent::Agent, loader.cpp:19-73; AgentsLoader::update reads max_timesteps, loader.cpp:221-233); no grid, no RNG"""
    agents = env.agents
    if any(a.initial_position is None or a.target is None for a in agents):
        raise ValueError("the env has to be reset() first (agents without initial position / target)")
    n = len(agents)
    return dict(
        init_pos=np.array([a.initial_position for a in agents], dtype=np.int32).reshape(n, 2),
        init_dir=np.array([int(a.initial_direction) for a in agents], dtype=np.int32),
        target=np.array([a.target for a in agents], dtype=np.int32).reshape(n, 2),
        speed=np.array([a.speed_counter.speed for a in agents], dtype=np.float64),
        earliest=np.array([a.earliest_departure for a in agents], dtype=np.int32),
        latest=np.array([a.latest_arrival for a in agents], dtype=np.int32),
        T=np.int32(env._max_episode_steps))


def static_of_env(env, require_rng=False):
    """static description (dict of numpy arrays, keys STATIC_KEYS) of any env object that has what flatland_cutils reads from
    it at reset() (loader.cpp:207-219, 329-333) and per agent (loader.cpp:19-73): `rail.grid`, `_max_episode_steps`, `agents`
    with initial_position / initial_direction / target / speed_counter.speed / earliest_departure / latest_arrival.  The
    malfunction parameters and the MT19937 state matter to step() only; without require_rng an env that has none is accepted."""
    rate, mn, mx = _malfunction_parameters(env)
    st = agents_static_of_env(env)
    key, pos = _rng_state_of(env, require_rng)
    st.update(grid=np.asarray(env.rail.grid, dtype=np.uint16), malf_rate=np.float64(rate), malf_min=np.int32(mn), malf_max=np.int32(mx),
              mt_key=key, mt_pos=np.int32(pos))
    return st


def from_reference_env(env):
    """static description of a reference RailEnv after reset(), MT19937 state of `env.np_random` included (step() needs it)."""
    return static_of_env(env, require_rng=True)


_STATE_BY_NAME = {"WAITING": 0, "READY_TO_DEPART": 1, "MALFUNCTION_OFF_MAP": 2, "MOVING": 3, "STOPPED": 4, "MALFUNCTION": 5, "DONE": 6}


def _state_code(s):
    """TrainState as an int; flatland_cutils parses str(agent.state) = "TrainState.MOVING" (loader.cpp:10, tool.h:219-228)"""
    try:
        return int(s)
    except (TypeError, ValueError):
        return _STATE_BY_NAME[str(s).rsplit(".", 1)[-1]]


def dynamic_state_of_env(env):
    """(state int32[A, 12], aux int32[A, 4], elapsed) of any env object mid-episode, reading exactly what Agent::Agent reads per
    call (loader.cpp:8-73) -- attributes flatland_cutils does not read (saved action, previous state, dones) are optional."""
    A = len(env.agents)
    state = np.zeros((A, 12), dtype=np.int32)
    aux = np.zeros((A, 4), dtype=np.int32)
    dones = getattr(env, "dones", None)
    for i, a in enumerate(env.agents):
        r, c = a.position if a.position is not None else (-1, -1)
        orow, ocol = a.old_position if a.old_position is not None else (-1, -1)
        code = _state_code(a.state)
        saved = getattr(getattr(a, "action_saver", None), "saved_action", None)
        mh = a.malfunction_handler
        state[i] = (r, c, int(a.direction), code, int(mh.malfunction_down_counter), int(mh.num_malfunctions),
                    int(a.speed_counter.counter), 0 if saved is None else int(saved),
                    -1 if a.arrival_time is None else int(a.arrival_time), orow, ocol,
                    -1 if a.old_direction is None else int(a.old_direction))
        sm = getattr(a, "state_machine", None)
        prev = getattr(sm, "previous_state", None)
        sig = getattr(getattr(sm, "st_signals", None), "in_malfunction", None)
        done = dones[i] if dones is not None and i in dones else code == 6
        aux[i] = (-1 if prev is None else _state_code(prev), int(bool(mh.malfunction_down_counter > 0 if sig is None else sig)), 0,
                  int(bool(done)))
    return state, aux, int(env._elapsed_steps)


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
