// fl_obs.hip -- host side of the observation kernels: scratch allocation, the choice of what a launch keeps in LDS
// (obs_pick_config; the carving itself is obs_layout_c in fl_obs_layout.h, shared with the kernels), the fixed launch classes
// and the three launch entry points.  The kernels are in fl_obs_m0 ... m5.hip (one unit per MODE) and fl_obs_f1 ... f4.hip (one
// per fixed launch class), all instantiating fl_obs_body.h and the phase-level headers it includes (fl_obs_ctx.h,
// fl_obs_passb.h, fl_obs_trees.h).
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/flatland_hip.h"
#include "fl_obs_layout.h"

int fl_obs_alloc(FlObsScratch &o, const FlDev &d, hipStream_t s, std::vector<void *> &allocs) {
    o.pred_cap = OBS_PRED_CAP;
    o.items_cap = (size_t)d.A * (o.pred_cap + 2 * OBS_FB_NB + 2);  // bucketed lists: an item sits in every time bucket it touches
    const size_t BA = (size_t)d.B * d.A;
    void *p = nullptr;
    if (hipMalloc(&p, BA * o.pred_cap * 2) != hipSuccess) return FL_ERR_HIP;
    o.path = (uint16_t *)p; allocs.push_back(p);
    if (hipMalloc(&p, (size_t)d.B * o.items_cap * 4 + 64) != hipSuccess) return FL_ERR_HIP;   // (+ 64: conflict_flags reads up to seven words behind a list)
    o.cell_items = (uint32_t *)p; allocs.push_back(p);
    if (hipMalloc(&p, (size_t)d.B * 64 * 8) != hipSuccess) return FL_ERR_HIP;
    o.dbg = (long long *)p; allocs.push_back(p);
    if (hipMalloc(&p, (size_t)d.B * (d.Rcap + 1) * OBS_BK_NB * 2 + 16) != hipSuccess) return FL_ERR_HIP;
    o.bk_rel = (uint16_t *)p; allocs.push_back(p);
    o.wl_cap = OBS_WL_HBM_ENTRIES;
    if (hipMalloc(&p, (size_t)d.B * o.wl_cap * 8) != hipSuccess) return FL_ERR_HIP;
    o.wl = (uint2 *)p; allocs.push_back(p);
    if (hipMalloc(&p, (size_t)d.A * sizeof(int16_t) + 16) != hipSuccess) return FL_ERR_HIP;
    o.label = (int16_t *)p; allocs.push_back(p);
    if (hipMalloc(&p, BA * sizeof(uint4)) != hipSuccess) return FL_ERR_HIP;
    o.rowmask = (uint4 *)p; allocs.push_back(p);
    if (hipMemsetAsync(o.rowmask, 0, BA * sizeof(uint4), s) != hipSuccess) return FL_ERR_HIP;
    o.rows_out = nullptr; o.rows_depth = 0; o.keep_rows = 0;
    if (hipMalloc(&p, (size_t)d.B * 4) != hipSuccess) return FL_ERR_HIP;
    o.cost = (uint32_t *)p; allocs.push_back(p);
    if (hipMemsetAsync(o.cost, 0, (size_t)d.B * 4, s) != hipSuccess) return FL_ERR_HIP;
    // every scratch array starts out zeroed: what a kernel reads of them it has written before in the same launch, but a handle's
    // behaviour must not depend on what a freed allocation of an earlier handle left in the memory it got
    if (hipMemsetAsync(o.path, 0, BA * o.pred_cap * 2, s) != hipSuccess || hipMemsetAsync(o.cell_items, 0, (size_t)d.B * o.items_cap * 4, s) != hipSuccess ||
        hipMemsetAsync(o.dbg, 0, (size_t)d.B * 64 * 8, s) != hipSuccess || hipMemsetAsync(o.bk_rel, 0, (size_t)d.B * (d.Rcap + 1) * OBS_BK_NB * 2 + 16, s) != hipSuccess ||
        hipMemsetAsync(o.wl, 0, (size_t)d.B * o.wl_cap * 8, s) != hipSuccess) return FL_ERR_HIP;
    if (hipMalloc(&p, (size_t)d.B * 4) != hipSuccess) return FL_ERR_HIP;
    o.order = (int *)p; allocs.push_back(p);
    if (hipMemsetAsync(o.order, 0, (size_t)d.B * 4, s) != hipSuccess) return FL_ERR_HIP;
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return FL_ERR_HIP;
    o.n_cu = n_cu;
    o.order_age = 0;
    return FL_OK;
}

// The launch is one workgroup per env and a CU holds one workgroup: with more envs than CUs the launch ends when the last CU has
// worked through its envs, and the envs differ (agents on the map, traffic around them) -- mean 187 us, slowest 315 us at cfg4.
// Longest first: the workgroups take the envs in descending order of what each took in the previous launch (a counting sort
// over 1024 bins by ONE workgroup; ties in any order -- the results of an env do not depend on which workgroup builds it).
__global__ __launch_bounds__(1024) void k_env_order(int B, const uint32_t *__restrict__ cost, int *__restrict__ order) {
    __shared__ unsigned int hist[1024];
    __shared__ unsigned int wsum[16];
    __shared__ unsigned int cmax;
    const int tid = threadIdx.x;
    hist[tid] = 0;
    if (tid == 0) cmax = 1;
    __syncthreads();
    unsigned int m = 0;
    for (int b = tid; b < B; b += 1024) m = max(m, cost[b]);
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned int)__shfl_down((int)m, off));
    if ((tid & 63) == 0) atomicMax(&cmax, m);
    __syncthreads();
    const unsigned long long top = cmax;
    for (int b = tid; b < B; b += 1024) atomicAdd(&hist[1023u - (unsigned int)((unsigned long long)cost[b] * 1023ull / top)], 1u);
    __syncthreads();
    // exclusive prefix over the bins (bin 0 = the longest envs)
    const unsigned int v = hist[tid];
    unsigned int incl = v;
    for (int off = 1; off < 64; off <<= 1) { const unsigned int u = (unsigned int)__shfl_up((int)incl, off); if ((tid & 63) >= off) incl += u; }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    unsigned int base = 0;
    for (int w = 0; w < (tid >> 6); w++) base += wsum[w];
    __syncthreads();
    hist[tid] = base + incl - v;
    __syncthreads();
    for (int b = tid; b < B; b += 1024) order[atomicAdd(&hist[1023u - (unsigned int)((unsigned long long)cost[b] * 1023ull / top)], 1u)] = b;
}

// FL_OBS_KEEP_VERIFY (diagnostic): FL_OBS_KEEP_TREE_ROWS rests on the caller's promise that the tree buffer is the previous launch's, untouched
// -- a buffer that was freed and re-allocated at the same address, or a tensor modified in place (-inf replaced before a network), breaks it
// silently.  With the switch set, every launch that skips the pre-fill first checks the promise: a row the masks call constant must still be
// -inf, a row they call real must not be; a violation latches FL_ERR_ARG for the env (fl_check: "tree buffer modified").
__global__ void k_keep_verify(int B, int A, int n_rows, const double *__restrict__ out, const uint4 *__restrict__ rowmask, int *__restrict__ err) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (long long)B * A * n_rows) return;
    const long long g = k / n_rows;
    const int row = (int)(k % n_rows);
    const uint4 m = rowmask[g];
    if (!m.w) return;
    const uint32_t w = row < 32 ? m.x : row < 64 ? m.y : m.z;
    const bool real = (w >> (row & 31)) & 1u;
    const double v = out[k * 12];
    const bool ninf = isinf(v) && v < 0;
    if (real == ninf) atomicCAS(&err[g / A], 0, FL_ERR_ARG);
}
static void obs_keep_verify(const FlDev &d, const ObsArgs &P, const uint4 *rowmask, hipStream_t s) {
    static const bool verify = getenv("FL_OBS_KEEP_VERIFY") != nullptr;
    if (!verify || !P.keep_rows || !rowmask || P.n_tree_nodes > 96) return;
    const long long n = (long long)d.B * d.A * P.n_tree_nodes;
    hipLaunchKernelGGL(k_keep_verify, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d.B, d.A, P.n_tree_nodes, P.tree_out, rowmask, d.err);
}

#ifndef OBS_ROUND16_DEFAULT
#define OBS_ROUND16_DEFAULT 0   // rounds of 16 agents on 512 threads, two workgroups a CU (MODE 5) for envs of more than 32 agents
#endif
#ifndef OBS_ORDER_EVERY
#define OBS_ORDER_EVERY 4   // what an env takes changes slowly from step to step: the order of every fourth launch serves the next three
#endif
FlObsScratch fl_obs_env_order(FlObsScratch &o, const FlDev &d, hipStream_t s) {
    static const bool off = getenv("FL_OBS_NO_ORDER") != nullptr;   // diagnostic: workgroup k builds env k
    FlObsScratch u = o;
    if (off || d.B <= o.n_cu) { u.order = nullptr; return u; }
    if (o.order_age++ % OBS_ORDER_EVERY == 0) hipLaunchKernelGGL(k_env_order, dim3(1), dim3(1024), 0, s, d.B, o.cost, o.order);
    return u;
}

// carve the dynamic LDS of a launch (obs_layout_c in fl_obs_layout.h: one function for the host and for the kernels with a
// compile-time layout)
static ObsLayout obs_layout(const FlDev &d, const ObsArgs &P, const ObsOptions &o) {
    return obs_layout_c(ObsDims{d.Rcap, d.A, d.Ucap, d.rkey != nullptr}, ObsShape{P.merged, P.tw_c, P.tw_t, P.tpw_t, P.tree_pred}, o);
}

// A FIXED launch class (compile-time carving, ObsFixed<k> in fl_obs_layout.h) is taken when the batch fits the class's capacities
// and the configuration just chosen for it has the class's options and shape; L becomes the class's carving (what the kernel has
// compiled in) with the next-hop tables, last in the carving, at the batch's size.
static bool obs_no_wl_head() {
    static const bool v = getenv("FL_OBS_NO_WL_HEAD") != nullptr;   // diagnostic: HBM work lists without their LDS head (rules out the classes that have one)
    return v;
}
template <int FIX>
static bool obs_fits_fixed(const FlDev &d, const ObsArgs &P, const ObsOptions &o, ObsLayout &L) {
    using F = ObsFixed<FIX>;
    if (F::opt.wl_head && obs_no_wl_head()) return false;
    if (P.label) return false;   // (get_many(handles) with a strict subset: the stand-alone kernel's runtime body)
    if ((FIX == 1 || FIX == 5) && P.keep_mode) return false;   // (the small-env classes carry no row-mask code: FL_OBS_KEEP_TREE_ROWS runs the runtime carving there)
    const size_t nh_bytes = F::opt.nh ? (((size_t)d.Ucap * d.Rcap * 2 + 15) & ~(size_t)15) : 0;
    if (!(d.A <= F::dims.A && d.Rcap <= F::dims.Rcap && d.rkey == nullptr && obs_same_options(o, F::opt) && P.merged == F::shape.merged &&
          P.tw_c == F::shape.tw_c && P.tw_t == F::shape.tw_t && P.tpw_t == F::shape.tpw_t)) return false;
    if (F::opt.dual && (size_t)d.A * (P.tree_pred + 2) > (size_t)F::L.items2_cap) return false;
    // the class's kernel has the builders' parameters compiled in
    if (P.max_nodes != F::max_nodes || P.pred_depth != F::pred_depth || P.tree_pred != F::shape.tree_pred || (F::max_depth != 0 && P.max_depth != F::max_depth)) return false;
    if (F::agents != 0 && d.A != F::agents) return false;
    // ... and what the launcher derives from the options (the kernel has the class's values: obs_fixed_* in fl_obs_layout.h)
    if (!P.compact_t || P.bk != obs_fixed_bk<FIX>() || P.wl_occ_div != obs_fixed_wl_occ_div<FIX>() || P.tshift != obs_fixed_tshift<FIX>(d.A)) return false;
    const size_t total = F::opt.nh ? F::L.off[L_NH] + nh_bytes : F::L.total;
    if (total > (size_t)160 * 1024) return false;
    L = F::L;
    L.total = (unsigned)total;
    return true;
}
// A BIN class (round 6): the batch fits the class's capacities, the call has the class's builder parameters (the depth is the call's) and the
// batch's own choice has the class's STRUCTURE (one round / rounds of 32 agents / two stages) -- the class's options then replace the batch's
// own: a compile-time carving with the options of the bin's largest shape instead of the runtime carving with options chosen for this one
// (per shape of the Round-2 table: profiles/r06_round2_classes.txt).  FL_OBS_NO_BINS: exact classes only.
template <int FIX> static int obs_split_fits(const FlDev &d, const ObsArgs &P, const int *h_R);
template <int FIX>
static bool obs_fits_bin(const FlDev &d, ObsArgs &P, const ObsOptions &own, ObsLayout &L) {
    using F = ObsFixed<FIX>;
    if (F::opt.wl_head && obs_no_wl_head()) return false;
    // the large-map bins have no LDS successor table: not for a batch that affords one itself (Test_12: 200 agents on 2 745 rail cells --
    // same box, its envs beyond class 4 on the runtime carving WITH the table 435 us, on class 14 without 442)
    if (!F::opt.snext && F::dims.rkey == 0 && own.snext) return false;
    if (P.label || !P.compact_t || (d.rkey != nullptr) != (F::dims.rkey != 0) || d.A > F::dims.A || d.Rcap > F::dims.Rcap) return false;
    if (P.merged != F::shape.merged || P.tw_c != F::shape.tw_c || P.tw_t != F::shape.tw_t || P.tpw_t != F::shape.tpw_t) return false;
    if (P.max_nodes != F::max_nodes || P.pred_depth != F::pred_depth || P.tree_pred != F::shape.tree_pred || (F::max_depth != 0 && P.max_depth != F::max_depth)) return false;
    if (F::shape.tw_t != 0 && (P.max_depth < 1 || P.max_depth > 3)) return false;
    if (FIX == 11 && P.keep_mode) return false;   // (class 1's code: no row masks)
    if (F::opt.dual && (size_t)d.A * (P.tree_pred + 2) > (size_t)F::L.items2_cap) return false;
    if (F::shape.merged != 0 && (long long)d.A * (P.pred_depth + 2) >= 65536) return false;
    const size_t nh_bytes = F::opt.nh ? (((size_t)d.Ucap * d.Rcap * 2 + 15) & ~(size_t)15) : 0;
    const size_t total = F::opt.nh ? F::L.off[L_NH] + nh_bytes : F::L.total;
    if (total > (size_t)160 * 1024) return false;
    // what the launcher derives from the options: the class's values (its kernel has them compiled in)
    P.use_tmask = F::opt.tmask; P.dual_index = F::opt.dual; P.bk = obs_fixed_bk<FIX>();
    P.bk_nb = F::shape.merged != 0 ? OBS_FB_NB : OBS_BK_NB; P.bk_shift = F::shape.merged != 0 ? OBS_FB_SHIFT : OBS_BK_SHIFT;
    P.wl_occ_div = obs_fixed_wl_occ_div<FIX>(); P.tshift = obs_fixed_tshift<FIX>(d.A);
    L = F::L;
    L.total = (unsigned)total;
    return true;
}
// an exact class's SPLIT launch (its body for the envs that fit, the runtime carving for the few larger maps) goes before a bin class for the
// whole batch when at least half the envs fit the exact class
template <int FIX>
static bool exact_split_covers_most(const FlDev &d, ObsArgs &P, const ObsLayout &L) {
    static const bool no_split = getenv("FL_OBS_NO_SPLIT") != nullptr;
    if (no_split || !P.h_R) return false;
    const ObsLayout keep = P.L;
    P.L = L;
    const int n = obs_split_fits<FIX>(d, P, P.h_R);
    P.L = keep;
    return 2 * n >= d.B;
}
static void obs_take_fixed_class(const FlDev &d, ObsArgs &P, const ObsOptions &o, ObsLayout &L, bool allowed) {
    static const bool no_fix = getenv("FL_OBS_NO_FIX") != nullptr;   // diagnostic: the runtime carving for every batch
    // diagnostic: exact classes only (the launcher of rounds 4 and 5).  A bin class REPLACES the batch's own options, so the switches that shape those options
    // (what a same-box experiment wants to measure) rule the bins out as well; an exact class only ever matches options it dominates.
    static const bool no_bins = [] {
        for (const char *v : {"FL_OBS_NO_BINS", "FL_OBS_FORCE", "FL_OBS_NO_FB", "FL_OBS_NO_BK", "FL_OBS_NO_OWN_FILTER", "FL_OBS_NO_MERGE", "FL_OBS_NO_TAB", "FL_OBS_ROUND16"})
            if (getenv(v) != nullptr) return true;
        return false;
    }();
    P.fix = 0; P.split = 0;
    if (no_fix || !allowed) return;
    if (P.tw_t == 0) {   // the flatland_cutils builder alone: classes 6 .. 10 (the counterparts of 1 .. 5)
        if (obs_fits_fixed<6>(d, P, o, L)) P.fix = 6;
        else if (obs_fits_fixed<7>(d, P, o, L)) P.fix = 7;
        else if (obs_fits_fixed<8>(d, P, o, L)) P.fix = 8;
        else if (obs_fits_fixed<9>(d, P, o, L)) P.fix = 9;
        else if (obs_fits_fixed<10>(d, P, o, L)) P.fix = 10;
        else if (obs_fits_fixed<21>(d, P, o, L)) P.fix = 21;
        else if (no_bins || !P.cutils_alone) return;   // (FL_OBS_NO_CUTILS_MERGE: the stand-alone kernel as it ran before round 6)
        else if (obs_fits_bin<21>(d, P, o, L)) P.fix = 21;
        else if (obs_fits_bin<17>(d, P, o, L)) P.fix = 17;
        else if (obs_fits_bin<8>(d, P, o, L)) P.fix = 8;
        else if (obs_fits_bin<18>(d, P, o, L)) P.fix = 18;
        else if (obs_fits_bin<9>(d, P, o, L)) P.fix = 9;
        else if (obs_fits_bin<19>(d, P, o, L)) P.fix = 19;
        else if (obs_fits_bin<20>(d, P, o, L)) P.fix = 20;
        return;
    }
    if (obs_fits_fixed<1>(d, P, o, L)) P.fix = 1;
    else if (obs_fits_fixed<2>(d, P, o, L)) P.fix = 2;
    else if (obs_fits_fixed<3>(d, P, o, L)) P.fix = 3;
    else if (obs_fits_fixed<4>(d, P, o, L)) P.fix = 4;
    else if (obs_fits_fixed<5>(d, P, o, L)) P.fix = 5;
    else if (no_bins) return;
    else if (exact_split_covers_most<2>(d, P, L) || exact_split_covers_most<3>(d, P, L)) return;   // (obs_take_split_class takes it)
    else if (obs_fits_bin<11>(d, P, o, L)) P.fix = 11;
    else if (obs_fits_bin<12>(d, P, o, L)) P.fix = 12;
    else if (obs_fits_bin<13>(d, P, o, L)) P.fix = 13;
    else if (obs_fits_bin<15>(d, P, o, L)) P.fix = 15;
    else if (obs_fits_bin<4>(d, P, o, L)) P.fix = 4;
    else if (obs_fits_bin<14>(d, P, o, L)) P.fix = 14;
    else if (obs_fits_bin<16>(d, P, o, L)) P.fix = 16;
}

// (ObsArgs::fix_allowed: this launch's configuration was chosen without the diagnostic overrides that rule the fixed launch classes out)
// A batch whose LARGEST map exceeds a class's rail cells still has the class's kind of envs in it (the levels of a Round-2 test differ
// by a few per cent in rail cells; the classes are the BASELINE maps' own sizes): when everything but the rail-cell capacity matches
// -- agents, builder parameters, and the runtime configuration just chosen for the batch runs the class's MODE and VAR -- the launch
// takes the class's SPLIT kernel: per workgroup the class's body for an env that fits, the runtime-carving body (P.L, unchanged) for
// the others.  h_R: the host's copy of the envs' rail cells (null: unknown, no split).  Returns the number of envs that fit.
static int obs_var(const ObsArgs &P);
template <int FIX>
static int obs_split_fits(const FlDev &d, const ObsArgs &P, const int *h_R) {
    using F = ObsFixed<FIX>;
    if (F::opt.wl_head && obs_no_wl_head()) return 0;
    if (!h_R || F::opt.nh) return 0;   // (class 1 keeps the next-hop tables, sized by the batch, behind its carving: not split)
    if ((F::agents != 0 ? d.A != F::agents : d.A > F::dims.A) || d.rkey != nullptr || !P.compact_t) return 0;
    if (P.max_nodes != F::max_nodes || P.pred_depth != F::pred_depth || P.tree_pred != F::shape.tree_pred || (F::max_depth != 0 && P.max_depth != F::max_depth)) return 0;
    if (P.merged != F::shape.merged || obs_var(P) != obs_fixed_var<FIX>() || P.L.nt != F::opt.nt) return 0;
    if (P.tw_c != F::shape.tw_c || P.tw_t != F::shape.tw_t || P.tpw_t != F::shape.tpw_t) return 0;
    if (F::opt.dual && (size_t)d.A * (P.tree_pred + 2) > (size_t)F::L.items2_cap) return 0;
    if (F::L.total > (size_t)160 * 1024) return 0;
    int n = 0;
    for (int b = 0; b < d.B; b++) n += h_R[b] <= F::dims.Rcap;
    return n;
}
static int obs_take_split_class(const FlDev &d, ObsArgs &P, const int *h_R) {
    static const bool no_fix = getenv("FL_OBS_NO_FIX") != nullptr, no_split = getenv("FL_OBS_NO_SPLIT") != nullptr;   // diagnostic
    if (no_fix || no_split || !P.fix_allowed) return 0;
    if (P.fix == 14 || P.fix == 19) {
        // the larger large-map bin (no LDS successor table) was taken because of the batch's largest map: the envs that fit the smaller class
        // (with the table) run ITS body, the others the bin's -- one kernel, every env on a compile-time carving (split 2)
        const int fix2 = P.fix;
        P.fix = 0;
        const int n2 = fix2 == 14 ? obs_split_fits<4>(d, P, h_R) : obs_split_fits<9>(d, P, h_R);
        const unsigned t2 = fix2 == 14 ? ObsFixed<4>::L.total : ObsFixed<9>::L.total;
        if (n2 > 0) { P.fix = fix2 == 14 ? 4 : 9; P.split = 2; if (t2 > P.L.total) P.L.total = t2; return n2; }
        P.fix = fix2;
        return 0;
    }
    if (P.fix != 0) return 0;
    int n = 0, k = 0;
    unsigned total = 0;
    if (P.tw_t == 0) {   // the flatland_cutils builder alone: the large-map class has a split kernel (the levels of cfg5's row differ by 13 % in rail cells)
        if ((n = obs_split_fits<9>(d, P, h_R)) > 0) { k = 9; total = ObsFixed<9>::L.total; }
    } else
    if ((n = obs_split_fits<2>(d, P, h_R)) > 0) { k = 2; total = ObsFixed<2>::L.total; }
    else if ((n = obs_split_fits<3>(d, P, h_R)) > 0) { k = 3; total = ObsFixed<3>::L.total; }
    else if ((n = obs_split_fits<4>(d, P, h_R)) > 0) { k = 4; total = ObsFixed<4>::L.total; }
    if (k == 0) return 0;
    P.fix = k; P.split = 1;
    if (total > P.L.total) P.L.total = total;   // dynamic LDS of the launch: the larger of the two carvings
    return n;
}

static thread_local ObsOptions g_last_options;   // diagnostic (FL_OBS_VERBOSE): the options of the last configuration obs_pick_config chose

// Choose what lives in LDS so that the workgroup fits 160 KiB.
static bool obs_pick_config(const FlDev &d, ObsArgs &P) {
    // diagnostic overrides (experiments on the LDS / occupancy trade-off): FL_OBS_NT, FL_OBS_LDS_LIMIT (bytes), FL_OBS_NO_TAB
    static const int force_nt = getenv("FL_OBS_NT") ? atoi(getenv("FL_OBS_NT")) : 0;
    static const size_t lds_limit = getenv("FL_OBS_LDS_LIMIT") ? (size_t)atol(getenv("FL_OBS_LDS_LIMIT")) : (size_t)160 * 1024;
    static const bool no_tab = getenv("FL_OBS_NO_TAB") != nullptr;
    // FL_OBS_FORCE="nt=512,wl=8192,tab=0,nh=1,tmask=1,dual=0,items=0,snext=1": only configurations with these values
    struct Force { int nt = -1, wl = -1, tab = -1, nh = -1, tmask = -1, dual = -1, items = -1, snext = -1; };
    static const Force force = [] {
        Force f;
        const char *e = getenv("FL_OBS_FORCE");
        if (!e) return f;
        const struct { const char *k; int *v; } keys[] = {{"nt=", &f.nt}, {"wl=", &f.wl}, {"tab=", &f.tab}, {"nh=", &f.nh},
                                                         {"tmask=", &f.tmask}, {"dual=", &f.dual}, {"items=", &f.items}, {"snext=", &f.snext}};
        for (const auto &kv : keys) {
            const char *q = strstr(e, kv.k);
            if (q && (q == e || q[-1] == ',')) *kv.v = atoi(q + strlen(kv.k));
        }
        return f;
    }();
    auto ok = [](int forced, int v) { return forced < 0 || forced == v; };
    const int nts[3] = {OBS_NT, 512, 256};
    const bool nh_fit = (size_t)d.Ucap * d.Rcap * 2 <= 24 * 1024;  // beyond that the next-hop tables stay in HBM / L2
    const bool dual_ok = P.tw_c != 0 && P.tw_t != 0 && P.tree_pred >= 0 &&
                         (long long)d.A * (P.pred_depth + 2) < 65536 && (long long)d.A * (P.tree_pred + 2) < 32768;
    // the flatland_cutils builder ALONE on the one-pass kernels (MODE 6 / 7 / 8: no second index, no upstream tables): 32-slot trees,
    // every agent listed (get_many(handles) with a strict subset stays on the stand-alone kernel)
    const bool up = P.tw_t != 0;
    const bool merge_ok = up ? dual_ok : (P.cutils_alone && P.tw_c == N_WORDS_C * OBS_CAP_C && P.label == nullptr && (long long)d.A * (P.pred_depth + 2) < 65536);
    static const int opts[5][3] = {{1, 1, 1}, {1, 0, 1}, {1, 0, 0}, {0, 0, 1}, {0, 0, 0}};  // masks, second index, items
    // Order of preference, from same-box A/B runs (tools/obs_sweep.py): the most wavefronts; the full work-list space; the LDS
    // copy of the items, the time masks and the second index; the successor table; the next-hop tables; own scan scratch.
    // The env's distance / segment / eight-hop tables join them when there is room left (small maps): they make no
    // difference in time (their gathers hit L2 and hide behind the rest) but replace narrow HBM gathers with one coalesced read.
    ObsOptions o = {};
    static const int force_tshift = getenv("FL_OBS_TSHIFT") ? atoi(getenv("FL_OBS_TSHIFT")) : -1;
    // Both builders: ONE pass B per round of 32 agents over the trees of both (trees_merged).  It needs compact upstream
    // trees, sixteen wavefronts, the second index with its items and the time masks in LDS, the successor table and keys =
    // rail indices (the fast classify loop).  Small envs (one round) also get the own-path filter of the classify loop.
    static const bool no_merge = getenv("FL_OBS_NO_MERGE") != nullptr;
    P.merged = 0;
    P.wl_occ_div = OBS_WL_OCC_DIV;
    // Rounds of 16 agents on 512 threads and at most 80 KB of LDS (MODE 5): a CU then holds TWO workgroups, and one env's barriers
    // and L2 round trips are filled by the other's issue.  FL_OBS_ROUND16=0 / 1 overrides the default.
    static const int round16_env = getenv("FL_OBS_ROUND16") ? atoi(getenv("FL_OBS_ROUND16")) : -1;
    // Envs of at most 32 agents (rounds of 16 instead of the one-round kernel): when the batch has several envs per CU (P.wide) -- same-box
    // sweep at the cfg2 shape, runtime carving on both sides: 512 envs +7 %, 1024 +13 %, 2048 +17 % with two workgroups a CU; at one
    // env per CU the 512-thread workgroup alone takes 75 us against 53 (profiles/r05_cfg2_bsweep.json).  FL_OBS_ROUND16=2 forces it
    // for any batch, =0 rules it out.
    const bool r16_small = d.A <= 32 && (round16_env >= 0 ? round16_env == 2 : P.wide != 0);
    // ... and, round 6, envs of more than 32 agents when the builder runs ALONE on small maps (cfg3's kind) in a wide batch: without the second index
    // and the upstream tables its LDS lists and items fit 80 KB (class 21; same box at 1 024 cfg3 envs, runtime carving: 0.467 -> 0.419 ms)
    // (only where class 21 holds the batch: beyond its capacities the one-a-CU bin classes are the better choice)
    const bool r16_alone = !up && P.cutils_alone && d.A > 32 && P.wide != 0 && d.Rcap <= ObsFixed<21>::dims.Rcap && d.A <= ObsFixed<21>::dims.A;
    const bool r16 = (r16_small || (d.A > 32 && (round16_env >= 0 ? round16_env != 0 : (OBS_ROUND16_DEFAULT != 0 || r16_alone)))) && (!force_nt || force_nt == 512) && ok(force.nt, 512);
    const int merged_nt = r16 ? 512 : OBS_NT;
    const size_t merged_limit = r16 ? std::min(lds_limit, (size_t)80 * 1024) : lds_limit;
    if (!no_merge && merge_ok && P.compact_t && d.rkey == nullptr && (r16 || ((!force_nt || force_nt == OBS_NT) && ok(force.nt, OBS_NT))) &&
        ok(force.tmask, 1) && ok(force.dual, up ? 1 : 0) && ok(force.snext, 1) &&
        (!up || (size_t)d.A * (P.tree_pred + 2) <= OBS_ITEMS2_CAP)) {
        P.merged = r16 ? 3 : d.A <= 32 ? 1 : 2;
        o.nt = merged_nt; o.tmask = 1; o.dual = up ? 1 : 0; o.snext = 1; o.partial = 1; o.tab = 0; o.bk_room = 0;
        // Order of preference, from same-box sweeps (tools/gpu_env_sweep.sh).  One round (at most 32 agents, cfg2): 24 KB of LDS work
        // lists, the items in LDS, plain lists (the extra counting pass of the bucketed lists costs more than their short scans
        // save: 57.9 against 51.8 us).  Rounds of 32 agents: a round of 64 trees meets thousands of occupied cells, and a work
        // list that overflows is handled cell by cell inside the classify loop (cfg3: 1.04 ms with 24 KB, 0.82 ms with 36 KB) -- so
        // 36 KB of LDS lists, or else the lists in HBM scratch (no cap); the items in LDS before anything else (cfg4: 0.45 against
        // 0.52 ms); lists grouped by 32-step time buckets where their offsets fit too (cfg4: 0.48 -> 0.45 ms, cfg3 neutral).
        const bool no_head = obs_no_wl_head();
        static const bool no_own = getenv("FL_OBS_NO_OWN_FILTER") != nullptr;
        static const bool no_fb = getenv("FL_OBS_NO_FB") != nullptr;
        struct Pref { int fb, wl, items; };
        static const Pref one_round[] = {{0, 24 * 1024, 1}, {0, 16 * 1024, 1}, {0, 0, 1}, {0, 24 * 1024, 0}, {0, 0, 0}, {0, 8 * 1024, 0}};
        // (round 5) HBM lists without the LDS copy of the items BEFORE the copy: at 80 agents on 60 x 60 every env has more items than the
        // 4 096 entries that fit beside the rest (the copy is dead weight there), and what it occupied becomes the lists' LDS head
        static const Pref rounds[] = {{1, 36 * 1024, 1}, {0, 36 * 1024, 1}, {1, 0, 0}, {1, 0, 1}, {0, 0, 1}, {1, 24 * 1024, 1}, {0, 24 * 1024, 1},
                                      {0, 0, 0}, {0, 24 * 1024, 0}, {0, 8 * 1024, 0}};
        // rounds of 16 agents in 80 KB: half the trees a round meet half the cells -- 16 KB of LDS lists, else HBM scratch
        static const Pref rounds16[] = {{1, 16 * 1024, 1}, {0, 16 * 1024, 1}, {1, 0, 1}, {1, 12 * 1024, 1}, {0, 12 * 1024, 1}, {0, 0, 1}, {1, 16 * 1024, 0}, {1, 0, 0}, {0, 16 * 1024, 0},
                                        {0, 0, 0}, {0, 8 * 1024, 0}};
        // ... of small envs (17 .. 32 agents, two rounds): plain lists like the one-round kernel, so that the own-path filter fits too
        static const Pref rounds16_small[] = {{0, 16 * 1024, 1}, {0, 12 * 1024, 1}, {1, 16 * 1024, 1}, {0, 0, 1}, {0, 16 * 1024, 0}, {0, 0, 0}, {0, 8 * 1024, 0}};
        const Pref *prefs = P.merged == 1 ? one_round : P.merged == 3 ? (d.A <= 32 ? rounds16_small : rounds16) : rounds;
        const int n_prefs = P.merged == 1 ? (int)(sizeof one_round / sizeof one_round[0]) : P.merged == 3 ? (d.A <= 32 ? (int)(sizeof rounds16_small / sizeof rounds16_small[0]) : (int)(sizeof rounds16 / sizeof rounds16[0])) : (int)(sizeof rounds / sizeof rounds[0]);
        // what a configuration of this branch sets in P (and the fixed launch class that has exactly these options, if the batch fits one)
        auto accept = [&](const ObsOptions &oo, ObsLayout L) {
            g_last_options = oo;
            P.use_tmask = 1; P.dual_index = up ? 1 : 0;
            P.bk = oo.fb ? 2 : 0; P.bk_nb = OBS_FB_NB; P.bk_shift = OBS_FB_SHIFT;
            P.wl_occ_div = d.A <= 32 ? OBS_WL_OCC_DIV : 3;
            // small envs: 4-step buckets (same-box A/B on cfg2: 54.8 us against 55.1 with 2-step and 56.9 with 8-step buckets)
            P.tshift = force_tshift >= 0 ? force_tshift : (d.A <= 31 ? 2 : OBS_TSHIFT);
            P.fix_allowed = !force_nt && lds_limit == (size_t)160 * 1024 && force_tshift < 0;
            obs_take_fixed_class(d, P, oo, L, P.fix_allowed != 0);
            P.L = L;
        };
        for (int pk = 0; pk < n_prefs; pk++) {
            // the builder alone has the room for 36 KB of LDS lists AND the items' copy on maps where both builders never had it -- but there
            // (cfg4: 649 rail cells) every env has more items than the copy holds and the lists' entries are many: same box, runtime carving,
            // 0.268 ms with the LDS lists against 0.261 with HBM lists behind a 16 KB LDS head.  Small maps (cfg3) keep the LDS lists.
            if (!up && P.merged == 2 && prefs[pk].wl >= 36 * 1024 && d.Rcap > OBS_ALONE_LDS_LISTS_RCAP) continue;
            o.fb = prefs[pk].fb && P.pred_depth + 1 > 64; o.wl_bytes = prefs[pk].wl; o.items = prefs[pk].items;
            o.tab = force.tab == 1 && o.wl_bytes && nh_fit;   // diagnostic: the env's static tables in LDS too
            if ((o.fb && no_fb) || !ok(force.wl, o.wl_bytes) || !ok(force.items, o.items) || (o.items && d.A * 32 > OBS_ITEMS_LDS_CAP)) continue;
            // the own-path filter of the classify loop (a second set of time masks) before the full-size LDS copy of the items: on
            // sparse maps a third of the conflict entries are the walking agent's own prediction (cfg4: 34 %), and an env whose
            // items do not fit the smaller copy scans them in HBM scratch at nearly the same speed
            static const int caps[3] = {OBS_ITEMS_LDS_CAP, 4096, 2048};   // (2048: cfg3 0.86 against 0.77 ms -- most envs' items then sit in HBM; only in 80 KB)
            for (o.own_filter = no_own ? 0 : 1; o.own_filter >= 0; o.own_filter--)
                for (int ck = 0; ck < (o.items ? (P.merged == 3 ? 3 : 2) : 1); ck++)
                    for (o.raw = o.own_filter; o.raw >= 0; o.raw--)
                        for (o.nh = nh_fit ? 1 : 0; o.nh >= 0; o.nh--) {
                            if (!ok(force.nh, o.nh)) continue;
                            o.items_cap = caps[ck];
                            o.wl_head = 0;
                            ObsLayout L = obs_layout(d, P, o);
                            if (L.total > merged_limit) continue;
                            if (o.wl_bytes == 0 && !no_head) {   // lists in HBM scratch: an LDS head of what the carving leaves
                                const size_t room = (merged_limit - L.total) & ~(size_t)1023;
                                o.wl_head = (int)std::min(room, (size_t)OBS_WL_HEAD_MAX);
                                if (o.wl_head < OBS_WL_HEAD_MIN) o.wl_head = 0;
                                if (o.wl_head) L = obs_layout(d, P, o);
                            }
                            accept(o, L);
                            return true;
                        }
        }
        P.merged = 0;
    }
    o = ObsOptions();
    for (int k = 0; k < 3; k++) {
        o.nt = nts[k];
        if ((force_nt && o.nt != force_nt) || !ok(force.nt, o.nt)) continue;
        // work lists: in LDS when everything else fits beside them (small maps), else in HBM scratch (the LDS goes to the time
        // masks, the items and the second index; no cap on the entries), else LDS lists with whatever still fits
        for (int wk = 0; wk < 4; wk++) {
            o.wl_bytes = wk == 0 || wk == 2 ? 24 * 1024 : wk == 1 ? 0 : 8 * 1024;
            for (int opt = 0; opt < (wk == 0 ? 1 + !dual_ok : 5); opt++) {
                o.tmask = opts[opt][0]; o.dual = opts[opt][1]; o.items = opts[opt][2];
                if (o.dual && !dual_ok) continue;
                if (o.items && d.A * 32 > OBS_ITEMS_LDS_CAP) continue;  // hundreds of agents: their items never fit the LDS copy
                if (!ok(force.wl, o.wl_bytes) || !ok(force.tmask, o.tmask) || !ok(force.dual, o.dual) || !ok(force.items, o.items)) continue;
                // large maps: the cutils index grouped by time bucket, counted in the node tables' LDS (room for the counters)
                static const bool no_bk = getenv("FL_OBS_NO_BK") != nullptr;
                const bool want_bk = !no_bk && o.wl_bytes == 0 && !o.items && o.tmask && !o.dual && P.tw_c != 0 && P.pred_depth + 1 > 64 &&
                                     d.A * ((1 << OBS_BK_SHIFT) + 2) <= 65535;   // (offsets inside a bucket are 16 bits)
                for (o.bk_room = want_bk ? 1 : 0; o.bk_room >= 0; o.bk_room--)
                for (o.snext = 1; o.snext >= 0; o.snext--)
                    for (o.nh = nh_fit ? 1 : 0; o.nh >= 0; o.nh--)
                        for (o.partial = 1; o.partial >= 0; o.partial--) {  // 4 KB of scan scratch: borrowed when tight
                            if (!ok(force.snext, o.snext) || !ok(force.nh, o.nh)) continue;
                            ObsLayout L = obs_layout(d, P, o);
                            if (L.total > lds_limit) continue;
                            if (!no_tab && force.tab != 0 && nh_fit && o.wl_bytes) {
                                ObsOptions ot = o;
                                ot.tab = 1;
                                const ObsLayout Lt = obs_layout(d, P, ot);
                                if (Lt.total <= lds_limit) L = Lt;
                                else if (force.tab == 1) continue;
                            } else if (force.tab == 1) continue;
                            g_last_options = o;
                            P.use_tmask = o.tmask; P.dual_index = o.dual;
                            P.bk = o.bk_room; P.bk_nb = OBS_BK_NB; P.bk_shift = OBS_BK_SHIFT;
                            // 2-step buckets where the traffic is and one catch-all bucket for late times (8-step buckets over the
                            // whole horizon measured slower on every map size)
                            P.tshift = force_tshift >= 0 ? force_tshift : OBS_TSHIFT;
                            P.fix_allowed = !force_nt && lds_limit == (size_t)160 * 1024 && force_tshift < 0;
                            obs_take_fixed_class(d, P, o, L, P.fix_allowed != 0);
                            P.L = L;
                            return true;
                        }
            }
        }
    }
    return false;
}

static void obs_verbose(const ObsArgs &P) {
    static const bool verbose = getenv("FL_OBS_VERBOSE") != nullptr;  // diagnostic: the configuration obs_pick_config chose
    static int printed = 0;
    const ObsLayout &L = P.L;
    if (verbose && printed < 4) {
        printed++;
        fprintf(stderr, "[fl_obs] fixed launch class %d%s, %d threads, %u B LDS: static tables in LDS %d, next-hop in LDS %d, successor table %d, work lists %d B, time masks %d, second index %d, items in LDS %d, one pass B for both builders %d, compact upstream trees %d, bucketed index %d\n",
                P.fix, P.split ? " (split: the envs that fit it)" : "", L.nt, L.total, L.tab_lds, L.off[L_NH] != L_ABSENT, L.off[L_SNEXT] != L_ABSENT, L.wl_bytes, P.use_tmask, P.dual_index, L.off[L_ITEMS] != L_ABSENT, P.merged, P.compact_t, P.bk);
    }
}
static int obs_var(const ObsArgs &P) { return P.L.tab_lds ? 1 : P.L.wl_bytes == 0 ? 2 : 0; }

// node tables of the upstream builder: compact slots when no direction of a cell of the batch has more than two transitions
static void obs_tree_args(const FlDev &d, ObsArgs &P, int max_depth, int tree_pred, double *tree_out) {
    P.max_depth = max_depth; P.tree_pred = tree_pred; P.tree_out = tree_out;
    int n = 0, p = 1;
    for (int k = 0; k <= max_depth; k++) { n += p; p *= 4; }
    P.n_tree_nodes = n;
    static const bool no_compact = getenv("FL_OBS_NO_COMPACT") != nullptr;
    P.compact_t = d.max_branch <= 2 && !no_compact;
    if (P.compact_t && max_depth >= 4) { P.tw_t = N_WORDS_T * 32; P.tpw_t = 2; }   // depth 4: 30 compact slots on a 32-lane team (the stand-alone tree launch)
    else if (P.compact_t) { P.tw_t = N_WORDS_T * OBS_CAP_T_COMPACT; P.tpw_t = 4; }
    else { P.tw_t = max_depth <= 2 ? N_WORDS_T * 32 : N_WORDS_T * 88; P.tpw_t = max_depth <= 2 ? 2 : 1; }
}
int fl_launch_obs_cutils(FlObsScratch &o, const FlDev &d, int max_nodes, int pred_depth, float *attr, float *forest,
                         int32_t *adjacency, int32_t *node_order, int32_t *edge_order, uint8_t *valid, double *props,
                         hipStream_t s, const int16_t *label_dev, int out64) {
    if (d.A > 1024 || pred_depth + 2 > o.pred_cap || pred_depth > 510 || max_nodes > FL_OBS_MAX_NODES) return FL_ERR_ARG;
    ObsArgs P = {};
    P.label = label_dev;
    P.out64 = out64;
    P.max_nodes = max_nodes; P.pred_depth = pred_depth; P.attr = attr; P.forest = forest; P.adjacency = adjacency;
    P.node_order = node_order; P.edge_order = edge_order; P.valid = valid; P.props = props; P.dbg = o.dbg;
    P.tw_c = N_WORDS_C * (max_nodes > OBS_CAP_C ? 64 : OBS_CAP_C);   // (more than 32 nodes: 64-slot tables, one tree a wavefront)
    // The builder alone -- what the reference's solution launches (solution/eval_env.py:15-17) -- runs on the one-pass kernels' machinery
    // (MODE 6 / 7 / 8 = MODE 3 / 4 / 5 without the upstream builder; classes 6 .. 10) wherever those apply: 16-lane pass A teams need grids
    // whose cells have at most two transitions a direction (every Flatland rail cell type).  FL_OBS_NO_CUTILS_MERGE: the stand-alone kernel.
    static const bool no_compact = getenv("FL_OBS_NO_COMPACT") != nullptr, no_alone = getenv("FL_OBS_NO_CUTILS_MERGE") != nullptr;
    P.compact_t = d.max_branch <= 2 && !no_compact;
    P.cutils_alone = !no_alone;
    P.wide = obs_batch_is_wide(d.B, o.n_cu);
    if (!obs_pick_config(d, P)) return FL_ERR_ARG;
    const int n_split = P.label ? 0 : obs_take_split_class(d, P, o.h_R);
    if (!P.merged && !P.fix) P.compact_t = 0;    // (the stand-alone kernel's own pass A: teams of 32 lanes)
    o.last_fix = P.fix; o.last_split = P.split; o.last_fit = P.split ? n_split : P.fix ? d.B : 0;
    obs_verbose(P);
    FlObsScratch u = o;
    if (P.merged == 1) u.order = nullptr;   // small envs, one round: workgroup k builds env k
    else u = fl_obs_env_order(o, d, s);
    if (P.split == 2) return P.fix == 9 ? fl_obs_launch_s9b(d, u, P, s) : FL_ERR_ARG;
    if (P.split) return P.fix == 9 ? fl_obs_launch_s9(d, u, P, s) : FL_ERR_ARG;
    switch (P.fix) {
    case 17: return fl_obs_launch_f17(d, u, P, s);
    case 18: return fl_obs_launch_f18(d, u, P, s);
    case 19: return fl_obs_launch_f19(d, u, P, s);
    case 20: return fl_obs_launch_f20(d, u, P, s);
    case 21: return fl_obs_launch_f21(d, u, P, s);
    case 6: return fl_obs_launch_f6(d, u, P, s);
    case 7: return fl_obs_launch_f7(d, u, P, s);
    case 8: return fl_obs_launch_f8(d, u, P, s);
    case 9: return fl_obs_launch_f9(d, u, P, s);
    case 10: return fl_obs_launch_f10(d, u, P, s);
    default: break;
    }
    return P.merged == 1 ? fl_obs_launch_m6(obs_var(P), d, u, P, s) : P.merged == 2 ? fl_obs_launch_m7(obs_var(P), d, u, P, s) :
           P.merged == 3 ? fl_obs_launch_m8(obs_var(P), d, u, P, s) : fl_obs_launch_m0(obs_var(P), d, u, P, s);
}

int fl_launch_obs_both(FlObsScratch &o, const FlDev &d, int max_nodes, int pred_depth, float *attr, float *forest,
                       int32_t *adjacency, int32_t *node_order, int32_t *edge_order, uint8_t *valid, double *props,
                       int max_depth, int tree_pred, double *tree_out, hipStream_t s) {
    if (d.A > 1024 || pred_depth + 2 > o.pred_cap || pred_depth > 510 || max_nodes > OBS_CAP_C) return FL_ERR_ARG;   // (the fused kernels: 32-lane teams)
    if (max_depth > 3 || tree_pred > pred_depth || tree_pred < 0) return FL_ERR_ARG;  // the upstream path must be a prefix
    ObsArgs P = {};
    P.max_nodes = max_nodes; P.pred_depth = pred_depth; P.attr = attr; P.forest = forest; P.adjacency = adjacency;
    P.node_order = node_order; P.edge_order = edge_order; P.valid = valid; P.props = props; P.dbg = o.dbg;
    P.tw_c = N_WORDS_C * OBS_CAP_C;
    obs_tree_args(d, P, max_depth, tree_pred, tree_out);
    P.wide = obs_batch_is_wide(d.B, o.n_cu);
    P.keep_mode = o.keep_rows;
    P.h_R = o.h_R;
    if (!obs_pick_config(d, P)) return FL_ERR_ARG;
    // FL_OBS_KEEP_TREE_ROWS: the row masks of the previous launch describe this very buffer at this depth -> no pre-fill of the slab
    P.keep_rows = o.keep_rows && o.rows_out == tree_out && o.rows_depth == max_depth;
    o.rows_out = tree_out; o.rows_depth = max_depth;
    uint4 *const rowmask = o.rowmask;
    if (!o.keep_rows) o.rowmask = nullptr;   // (mode off: the kernels keep no row masks; restored below -- `u` is a copy of o)
    struct Restore { FlObsScratch &o; uint4 *m; ~Restore() { o.rowmask = m; } } restore{o, rowmask};
    const int n_split = obs_take_split_class(d, P, o.h_R);
    o.last_fix = P.fix; o.last_split = P.split; o.last_fit = P.split ? n_split : P.fix ? d.B : 0;
    obs_verbose(P);
    obs_keep_verify(d, P, rowmask, s);
    FlObsScratch u = o;
    if (P.merged == 1) u.order = nullptr;   // small envs, one round: workgroup k builds env k
    else u = fl_obs_env_order(o, d, s);
    if (P.split == 2) return P.fix == 4 ? fl_obs_launch_s4b(d, u, P, s) : FL_ERR_ARG;
    if (P.split) {   // the class for the envs that fit it, the runtime carving for the others: one kernel, the choice per workgroup
        switch (P.fix) {
        case 2: return fl_obs_launch_s2(d, u, P, s);
        case 3: return fl_obs_launch_s3(d, u, P, s);
        case 4: return fl_obs_launch_s4(d, u, P, s);
        default: return FL_ERR_ARG;
        }
    }
    switch (P.fix) {   // a fixed launch class: its own kernel (MODE and VAR are the class's)
    case 1: return fl_obs_launch_f1(d, u, P, s);
    case 2: return fl_obs_launch_f2(d, u, P, s);
    case 3: return fl_obs_launch_f3(d, u, P, s);
    case 4: return fl_obs_launch_f4(d, u, P, s);
    case 5: return fl_obs_launch_f5(d, u, P, s);
    case 11: return fl_obs_launch_f11(d, u, P, s);
    case 12: return fl_obs_launch_f12(d, u, P, s);
    case 13: return fl_obs_launch_f13(d, u, P, s);
    case 14: return fl_obs_launch_f14(d, u, P, s);
    case 15: return fl_obs_launch_f15(d, u, P, s);
    case 16: return fl_obs_launch_f16(d, u, P, s);
    default: break;
    }
    return P.merged == 1 ? fl_obs_launch_m3(obs_var(P), d, u, P, s) : P.merged == 2 ? fl_obs_launch_m4(obs_var(P), d, u, P, s) :
           P.merged == 3 ? fl_obs_launch_m5(obs_var(P), d, u, P, s) : fl_obs_launch_m2(obs_var(P), d, u, P, s);
}

int fl_launch_obs_tree(FlObsScratch &o, const FlDev &d, int max_depth, int pred_depth, double *out, hipStream_t s, const int16_t *label_dev) {
    if (d.A > 1024 || pred_depth + 2 > o.pred_cap || pred_depth > 510) return FL_ERR_ARG;
    if (max_depth > FL_MAX_TREE_DEPTH) return FL_ERR_ARG;
    // depth 4: compact node tables only (level L of the tree has at most 2^L nodes when no direction of a cell has more than two
    // transitions -- every Flatland rail cell type): 30 slots; the DFS-slot tables of other grids stop at depth 3 (85 slots)
    if (max_depth > 3 && d.max_branch > 2) return FL_ERR_ARG;
    ObsArgs P = {};
    P.dbg = o.dbg;
    P.label = label_dev;
    obs_tree_args(d, P, max_depth, pred_depth, out);
    if (max_depth > 3 && !P.compact_t) return FL_ERR_ARG;   // (FL_OBS_NO_COMPACT)
    if (!obs_pick_config(d, P)) return FL_ERR_ARG;
    P.keep_rows = o.keep_rows && o.rows_out == out && o.rows_depth == max_depth;
    o.rows_out = out; o.rows_depth = max_depth;
    uint4 *const rowmask = o.rowmask;
    if (!o.keep_rows || max_depth > 3) { o.rowmask = nullptr; P.keep_rows = 0; o.rows_out = nullptr; }   // (mode off, or more rows than the masks' 96 bits: no row masks)
    struct Restore { FlObsScratch &o; uint4 *m; ~Restore() { o.rowmask = m; } } restore{o, rowmask};
    o.last_fix = 0; o.last_split = 0; o.last_fit = 0;
    obs_verbose(P);
    obs_keep_verify(d, P, rowmask, s);
    return fl_obs_launch_m1(obs_var(P), d, fl_obs_env_order(o, d, s), P, s);
}

// diagnostic: the configuration obs_pick_config chooses for the fused launch (cutils + upstream tree of max_depth):
// threads, LDS bytes, static tables in LDS, next-hop in LDS, work-list bytes, time masks, second index, items in LDS
int fl_obs_config_of_fused(const FlDev &d, int pred_depth, int max_depth, int tree_pred, int out[11], int wide) {
    ObsArgs P = {};
    P.wide = wide;
    P.pred_depth = pred_depth;
    P.max_nodes = 31;   // (the solution's tree size; the fixed launch classes are for exactly that)
    P.tw_c = N_WORDS_C * OBS_CAP_C;
    if (max_depth > 0) obs_tree_args(d, P, max_depth, tree_pred, nullptr);
    else {   // max_depth 0: the flatland_cutils builder alone (fl_launch_obs_cutils)
        P.compact_t = d.max_branch <= 2 && getenv("FL_OBS_NO_COMPACT") == nullptr;
        P.cutils_alone = getenv("FL_OBS_NO_CUTILS_MERGE") == nullptr;
    }
    if (!obs_pick_config(d, P)) return FL_ERR_ARG;
    if (getenv("FL_OBS_VERBOSE")) {   // diagnostic: the carving of the LDS, array by array (enum L_* of fl_obs_layout.h)
        const ObsOptions &q = g_last_options;
        fprintf(stderr, "  options {nt %d, wl_bytes %d, tab %d, nh %d, tmask %d, dual %d, items %d, snext %d, partial %d, bk_room %d, own_filter %d, fb %d, raw %d, items_cap %d, wl_head %d}; "
                        "shape {merged %d, tw_c %d, tw_t %d, tpw_t %d, tree_pred %d}; bk %d tshift %d wl_occ_div %d\n",
                q.nt, q.wl_bytes, q.tab, q.nh, q.tmask, q.dual, q.items, q.snext, q.partial, q.bk_room, q.own_filter, q.fb, q.raw, q.items_cap, q.wl_head,
                P.merged, P.tw_c, P.tw_t, P.tpw_t, P.tree_pred, P.bk, P.tshift, P.wl_occ_div);
        static const char *names[L_COUNT] = {"cellw", "nbr", "snext", "rkey", "slot_agent", "slot_ready", "cell_target", "a_speed", "a_vpos", "a_pos", "a_tslot",
            "a_target", "a_malf", "a_tpc", "a_tq", "a_tq2", "a_raw", "rtype", "a_lp", "a_n", "a_srank", "a_dir", "a_state", "a_free", "a_dead", "misc", "team_meta", "node_tables",
            "csr", "items", "wl", "partial", "tmask", "tmask2", "nh", "csr2", "tmaskb", "tmaskb2", "items2", "a_lp2", "a_tpc2", "bkrel", "seg", "dm", "hop8"};
        for (int k = 0; k < L_COUNT; k++) {
            if (P.L.off[k] == L_ABSENT) continue;
            unsigned next = P.L.total;
            for (int j = 0; j < L_COUNT; j++) if (P.L.off[j] != L_ABSENT && P.L.off[j] > P.L.off[k] && P.L.off[j] < next) next = P.L.off[j];
            fprintf(stderr, "  %-12s %7u B\n", names[k], next - P.L.off[k]);
        }
    }
    out[0] = P.L.nt; out[1] = (int)P.L.total; out[2] = P.L.tab_lds; out[3] = P.L.off[L_NH] != L_ABSENT; out[4] = P.L.wl_bytes;
    out[5] = P.use_tmask + 2 * (P.L.off[L_TMASK2] != L_ABSENT);                    // 3: time masks + the own-path filter's second set
    out[6] = P.dual_index; out[7] = P.L.off[L_ITEMS] != L_ABSENT ? P.L.items_cap : 0;  // entries of the LDS copy of the items
    out[8] = P.merged; out[9] = P.compact_t; out[10] = P.fix;   // fixed launch class (compile-time LDS carving), 0 = none
    return FL_OK;
}

