#!/usr/bin/env python3
"""Golden vectors for the reset-time generators: runs the REAL reference
(sparse_rail_generator + sparse_line_generator + timetable_generator inside RailEnv.reset(), flatland-rl
rail_generators.py:196-292, line_generators.py:82-165, timetable_generators.py:21-96) in THIS container for several
levels of each of the five Round-2 parameter rows BASELINE.json's configs come from and stores
    inputs : width, height, number of agents, generator parameters, speed-ratio map, the MT19937 state BEFORE reset()
             (seeded through the gym-seeding stub), the np.argsort order of the city distances (the reference sorts
             them with an unstable sort: the tie order is this container's numpy's)
    outputs: city positions / orientations, train stations, the rail grid, agent start / direction / target / speed,
             earliest departure / latest arrival, max_episode_steps, the MT19937 state AFTER reset()
in tests/golden/gen_*.npz.  Build-container only; data, no reference source."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import capture_golden as cg  # noqa: E402  (sets up sys.path for the reference)
from flatland.core.grid.grid_utils import Vec2dOperations  # noqa: E402

ROWS = [("Test_0", 3), ("Test_2", 3), ("Test_3", 2), ("Test_4", 3), ("Test_8", 3), ("Test_13", 2)]   # Test_3 is taller than wide


def capture(test_id, level, grid_mode=False, rail_seed=None, resets=1):
    """rail_seed: sparse_rail_generator(seed=rail_seed) -- the rail from a private RandomState, lines and timetable from the env's
    (rail_generators.py:221-222).  resets = 2: the golden is the SECOND reset() of one env (the first one's MT19937 state is the
    input), i.e. a new map drawn from the running stream."""
    row = cg.csv_row(test_id, "Level_%d" % level)
    if grid_mode:
        row = dict(row)
        row["grid_mode"] = True
    env, mp = cg.make_env(row)
    if rail_seed is not None:
        env.rail_generator.seed = rail_seed
    st_first = env.np_random.get_state()
    for _ in range(resets - 1):
        env.reset()
    st0 = env.np_random.get_state()
    hints = {}
    gen = env.rail_generator
    orig = gen.generate

    def spy(*a, **k):
        rail, opt = orig(*a, **k)
        hints.update(opt["agents_hints"])
        return rail, opt
    gen.generate = spy
    env.reset()
    st1 = env.np_random.get_state()
    out = cg.static_arrays(env, mp)
    cp = np.array(hints["city_positions"], dtype=np.int32)
    order = np.stack([np.argsort([Vec2dOperations.get_manhattan_distance(tuple(a), tuple(b)) for b in cp]) for a in cp]).astype(np.int32)
    st = hints["train_stations"]
    ns = np.array([len(s) for s in st], dtype=np.int32)
    stations = np.full((len(st), ns.max(), 3), -1, dtype=np.int32)
    for c, lst in enumerate(st):
        for k, (cell, track) in enumerate(lst):
            stations[c, k] = (cell[0], cell[1], track)
    speeds = list(row["speed_ratios"].items())
    out.update(width=np.int32(row["x_dim"]), height=np.int32(row["y_dim"]), n_agents=np.int32(row["n_agents"]),
               max_num_cities=np.int32(row["n_cities"]), grid_mode=np.int32(bool(row["grid_mode"])),
               max_rails_between_cities=np.int32(row["max_rails_between_cities"]),
               max_rail_pairs_in_city=np.int32(row["max_rail_pairs_in_city"]),
               speed_values=np.array([s for s, _ in speeds], dtype=np.float64), speed_probs=np.array([p for _, p in speeds], dtype=np.float64),
               random_seed=np.uint64(int(row["random_seed"])),
               mt_key_before=np.asarray(st0[1], dtype=np.uint32), mt_pos_before=np.int32(st0[2]),
               city_positions=cp, city_orientations=np.array([int(o) for o in hints["city_orientations"]], dtype=np.int32),
               neighbour_order=order, n_stations=ns, stations=stations)
    if rail_seed is not None:
        out["rail_seed"] = np.int64(rail_seed)
    if resets > 1:   # the state the env was seeded with, before its first reset()
        out["mt_key_first"] = np.asarray(st_first[1], dtype=np.uint32)
        out["mt_pos_first"] = np.int32(st_first[2])
        out["resets"] = np.int32(resets)
    assert np.array_equal(out["mt_key"], st1[1]) and int(out["mt_pos"]) == st1[2]
    name = "gen_%s_L%d%s%s%s" % (test_id, level, "_grid" if grid_mode else "", "_railseed%d" % rail_seed if rail_seed is not None else "",
                                 "_reset%d" % resets if resets > 1 else "")
    path = os.path.join(cg.GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "%dx%d" % (row["x_dim"], row["y_dim"]), "cities", len(cp), "agents", row["n_agents"], "T", int(out["T"]),
          "rail cells", int((out["grid"] != 0).sum()), "->", os.path.getsize(path) // 1024, "KB", flush=True)


if __name__ == "__main__":
    only = sys.argv[1:]
    for test_id, n in ROWS:
        for level in range(n):
            if only and test_id not in only:
                continue
            capture(test_id, level)
    if not only or "grid" in only:
        capture("Test_2", 5, grid_mode=True)
        capture("Test_4", 4, grid_mode=True)
    if not only or "seeded" in only:      # sparse_rail_generator(seed=...)
        capture("Test_2", 6, rail_seed=7)
        capture("Test_4", 3, rail_seed=123)
    if not only or "reset2" in only:      # two consecutive reset() calls on one MT19937 stream
        capture("Test_2", 0, resets=2)
        capture("Test_4", 1, resets=2)
        capture("Test_0", 0, resets=3)
    if not only or "Test_14" in only:     # the largest Round-2 map: 158x158, 425 agents, 41 cities
        capture("Test_14", 0)
