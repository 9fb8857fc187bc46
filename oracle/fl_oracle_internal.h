/* Internal layout of the CPU oracle (test infrastructure only; see fl_oracle.h). */
#ifndef FL_ORACLE_INTERNAL_H
#define FL_ORACLE_INTERNAL_H
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fl_oracle.h"

enum { ST_WAITING = 0, ST_READY = 1, ST_MALF_OFF = 2, ST_MOVING = 3, ST_STOPPED = 4, ST_MALF = 5, ST_DONE = 6 };
enum { ACT_NOTHING = 0, ACT_LEFT = 1, ACT_FORWARD = 2, ACT_RIGHT = 3, ACT_STOP = 4 };

struct OrcEnv {
    int H, W, A, T, t;
    uint16_t *grid;
    /* static per agent */
    int *init_r, *init_c, *init_dir, *tgt_r, *tgt_c, *earliest, *latest, *max_count;
    double *speed;
    /* dynamic per agent: -1 = None for r, c, old_r, old_c, old_dir, arrival, prev_state; saved 0 = None */
    int *r, *c, *dir, *state, *prev_state, *saved, *scount, *malf, *nmalf, *old_r, *old_c, *old_dir, *arrival;
    uint8_t *done;
    uint8_t done_all;
    uint8_t *sig_in_malf; /* state_machine.st_signals.in_malfunction left behind by the last step */
    /* rng */
    uint32_t mt[624];
    int mti;
    uint64_t malf_threshold;
    int malf_min, malf_max;
    /* distance map per unique target */
    int U;
    int *tslot, *ut_r, *ut_c;
    uint16_t *dm; /* [U][H][W][4], 0xFFFF = inf */
    /* cutils persistent state */
    uint8_t *deadlocked;
};

static inline int orc_is_off_map(int s) { return s == ST_WAITING || s == ST_READY || s == ST_MALF_OFF; }
static inline int orc_is_on_map(int s) { return s == ST_MOVING || s == ST_STOPPED || s == ST_MALF; }

static const int ORC_DR[4] = {-1, 0, 1, 0};
static const int ORC_DC[4] = {0, 1, 0, -1};

/* Grid4Transitions.get_transitions (core/grid/grid4.py:66-87): 4-bit nibble, bit 3 = N ... bit 0 = W */
static inline int orc_nibble(uint16_t cell, int dir) { return (cell >> ((3 - dir) * 4)) & 15; }
/* Grid4Transitions.get_transition (grid4.py:127-149) */
static inline int orc_tbit(uint16_t cell, int dir, int m) { return (orc_nibble(cell, dir) >> (3 - m)) & 1; }
static inline int orc_popcount(unsigned x) { return __builtin_popcount(x); }
static inline uint16_t orc_cell(const OrcEnv *e, int r, int c) { return e->grid[r * e->W + c]; }
static inline int orc_in_bounds(const OrcEnv *e, int r, int c) { return r >= 0 && c >= 0 && r < e->H && c < e->W; }
/* distance as float: +inf for 0xFFFF */
static inline uint16_t orc_dm_at(const OrcEnv *e, int agent, int r, int c, int d) {
    return e->dm[(((size_t)e->tslot[agent] * e->H + r) * e->W + c) * 4 + d];
}
void orc_set_error(const char *fmt, ...);

#endif
