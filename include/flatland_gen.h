/*
 * flatland_gen.h -- C-ABI of the reset-time generators (host code): what
 * RailEnv.reset(regenerate_rail=True, regenerate_schedule=True) (flatland-rl/flatland/envs/rail_env.py:260-357) computes
 * before the first step -- SparseRailGen.generate (envs/rail_generators.py:196-292), SparseLineGen.generate
 * (envs/line_generators.py:82-165), timetable_generator (envs/timetable_generators.py:21-96) -- bit for bit on the same
 * numpy RandomState (MT19937) stream, so that a batch can be (re)filled without the Python reference
 * (fl_load_env + fl_commit of include/flatland_hip.h take the result).
 *
 * Two calls, because the reference orders the cities by distance with an UNSTABLE sort (np.argsort,
 * rail_generators.py:768) whose tie order is numpy's business: flg_city_positions places the cities, the caller may
 * compute numpy's order of them, flg_generate does the rest.  Host pointers only; no GPU involved; thread-safe
 * (no global state besides the thread-local error string).
 */
#ifndef FLATLAND_GEN_H
#define FLATLAND_GEN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { FLG_OK = 0, FLG_ERR_ARG = 1, FLG_ERR_INFEASIBLE = 2 /* ValueError: cannot fit more than one city (rail_generators.py:247-249) */ };

const char *flg_last_error(void);

/* SparseRailGen._generate_random_city_positions / _generate_evenly_distr_city_positions (rail_generators.py:294-398).
 * mt_key[624] / mt_pos: the env's np_random state, advanced in place.  city_positions: int32[max_num_cities][2] (row, col). */
int flg_city_positions(int width, int height, int max_num_cities, int grid_mode, int max_rails_between_cities,
                       int max_rail_pairs_in_city, uint32_t *mt_key, int *mt_pos, int *n_cities, int32_t *city_positions);

/* The rest of reset(): rail grid, agents' lines, timetable.
 * neighbour_order: int32[n_cities][n_cities], row i = np.argsort of the Manhattan distances from city i to every city, or
 *   NULL (= a stable sort; equal to numpy's wherever the distances have no ties).
 * speed_values / speed_probs: the speed_ratio_map of sparse_line_generator in dict order (n_speeds = 0: every speed 1.0).
 * Outputs: grid u16[height][width]; city_orientations int32[n_cities]; n_stations int32[n_cities]; stations
 *   int32[n_cities][max_stations][3] = (row, col, track) padded with -1 (the three may be NULL); init_pos / target
 *   int32[n_agents][2]; init_dir, earliest, latest int32[n_agents]; speed float64[n_agents]; max_episode_steps.
 * mt_key / mt_pos end up as env.np_random.get_state() after reset(). */
int flg_generate(int width, int height, int n_agents, int grid_mode, int max_rails_between_cities, int max_rail_pairs_in_city,
                 int n_cities, const int32_t *city_positions, const int32_t *neighbour_order, int n_speeds,
                 const double *speed_values, const double *speed_probs, uint32_t *mt_key, int *mt_pos, uint16_t *grid,
                 int32_t *city_orientations, int32_t *n_stations, int32_t *stations, int max_stations, int32_t *init_pos,
                 int32_t *init_dir, int32_t *target, double *speed, int32_t *earliest, int32_t *latest,
                 int32_t *max_episode_steps);

/* flg_generate for sparse_rail_generator(seed=s): the reference draws the RAIL (city positions included) from a private
 * RandomState(s) (rail_generators.py:221-222) while lines, timetable and the env's stream afterwards stay on env.np_random.
 * rail_mt_key / rail_mt_pos: that private stream (as left by flg_city_positions), advanced in place; NULL, NULL = the env's
 * stream draws everything (= flg_generate). */
int flg_generate_seeded_rail(int width, int height, int n_agents, int grid_mode, int max_rails_between_cities, int max_rail_pairs_in_city,
                             int n_cities, const int32_t *city_positions, const int32_t *neighbour_order, int n_speeds,
                             const double *speed_values, const double *speed_probs, uint32_t *rail_mt_key, int *rail_mt_pos,
                             uint32_t *mt_key, int *mt_pos, uint16_t *grid, int32_t *city_orientations, int32_t *n_stations,
                             int32_t *stations, int max_stations, int32_t *init_pos, int32_t *init_dir, int32_t *target,
                             double *speed, int32_t *earliest, int32_t *latest, int32_t *max_episode_steps);

/* timetable_generator alone (envs/timetable_generators.py:21-96) on a finished rail and line: what RailEnv.reset() redoes for an
 * env loaded from a file (rail_from_file / line_from_file, rail_generators.py:116-145, line_generators.py:168-206: the rail and
 * the line come back unchanged, earliest_departure / latest_arrival / max_episode_steps are DRAWN AGAIN from env.np_random).
 * n_cities: len(agents_hints['city_positions']), or 2 without hints (rail_from_file gives none).  grid u16[height][width]. */
int flg_timetable(int width, int height, const uint16_t *grid, int n_agents, int n_cities, const int32_t *init_pos, const int32_t *init_dir,
                  const int32_t *target, const double *speed, uint32_t *mt_key, int *mt_pos, int32_t *earliest, int32_t *latest,
                  int32_t *max_episode_steps);

#ifdef __cplusplus
}
#endif
#endif
