// fl_host.hip -- host side of the C-ABI (include/flatland_hip.h): device-resident SoA state, env hand-over,
// kernel launches on the handle's HIP stream.  No torch types; plain pointers and sizes.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/flatland_hip.h"
#include "fl_internal.h"
#include "fl_obs.h"

static thread_local char g_err[512] = "";
static void set_err(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *fl_last_error(void) { return g_err; }
int fl_version(void) { return 100; }
int fl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

#define HIPCHK(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_err("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return FL_ERR_HIP;                                                         \
        }                                                                              \
    } while (0)

struct fl_batch {
    int B, A, H, W, device;
    FlDev d;
    hipStream_t own_stream, stream;
    bool committed;
    int reserve_U, reserve_R;  // fl_reserve: capacity for envs loaded after the first commit
    std::vector<void *> allocs;
    // host copy of every env's static description (filled by fl_load_env; kept after commit: live map replacement,
    // fl_distance_map's expansion to the dense layout)
    std::vector<uint32_t> h_gridx;  // the device's copy of the grid: transitions | neighbour-has-rail bits (FlDev::grid)
    std::vector<uint16_t> h_grid, h_ridx, h_rgrid, h_nbr, h_rkey, h_init_r, h_target_r, h_ut_r, h_srank;
    std::vector<int> h_maxbr;  // per env: most transitions of any (cell, direction)
    std::vector<uint32_t> h_rcell;
    std::vector<int> h_init_pos, h_target, h_earliest, h_latest, h_tslot, h_ut, h_U, h_R, h_K, h_T, h_mt_pos, h_malf_min, h_malf_max;
    std::vector<uint32_t> h_spk, h_mt;
    std::vector<double> h_speed;
    std::vector<uint64_t> h_thr;
    std::vector<uint8_t> h_loaded, h_dirty;  // dirty: loaded since the last commit
    std::vector<uint8_t> h_rtype;
    uint8_t *mask_dev;   // [B] staging of host-side env masks (fl_reset, fl_commit after a live replacement)
    uint8_t *need_dev;   // [B] the envs whose slabs a commit / full rebuild (re)builds: the OWNERS of shared static tables (FlDev::tab)
    // shared static tables: h_key = content id of env b's map side (rail grid + unique targets), h_tab = the owner of its tables,
    // h_built = content id of what env b's slabs hold on the device (0 = nothing yet)
    std::vector<uint64_t> h_key, h_built;
    std::vector<int> h_tab;
    std::vector<uint8_t> h_need;
    FlObsScratch obs;
};

template <typename T>
static int dev_alloc(fl_batch *h, T **p, size_t n) {
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, n * sizeof(T) + 16));
    HIPCHK(hipMemsetAsync(q, 0, n * sizeof(T) + 16, h->stream));
    h->allocs.push_back(q);
    *p = (T *)q;
    return FL_OK;
}
#define DALLOC(ptr, n)                                   \
    do {                                                 \
        int rc_ = dev_alloc(h, &(ptr), (size_t)(n));     \
        if (rc_ != FL_OK) return rc_;                    \
    } while (0)

int fl_create(int B, int A, int H, int W, int device, fl_batch **out) {
    if (!out || B <= 0 || A <= 0 || H <= 0 || W <= 0) { set_err("fl_create: bad sizes"); return FL_ERR_ARG; }
    if (A > 1024) { set_err("fl_create: at most 1024 agents per env (one lane per agent), got %d", A); return FL_ERR_ARG; }
    if ((long long)H * W * 4 >= (1ll << 30)) { set_err("fl_create: map too large"); return FL_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_err("fl_create: no HIP device visible (the HIP path has no CPU fallback)");
        return FL_ERR_HIP;
    }
    HIPCHK(hipSetDevice(device));
    fl_batch *h = new fl_batch();
    h->B = B; h->A = A; h->H = H; h->W = W; h->device = device;
    h->committed = false;
    h->mask_dev = nullptr;
    if (fl_step_lds_bytes(A) > 160 * 1024) {
        delete h;
        set_err("fl_create: %d agents need %zu B of LDS per workgroup in the step kernel (limit 160 KiB)", A, fl_step_lds_bytes(A));
        return FL_ERR_ARG;
    }
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        set_err("hipStreamCreate failed");
        return FL_ERR_HIP;
    }
    h->stream = h->own_stream;
    memset(&h->d, 0, sizeof h->d);
    h->d.B = B; h->d.A = A; h->d.H = H; h->d.W = W; h->d.Ucap = 0; h->d.Rcap = 0;
    const size_t BA = (size_t)B * A, HW = (size_t)H * W;
    h->reserve_U = 0; h->reserve_R = 0;
    h->h_grid.assign(B * HW, 0); h->h_gridx.assign(B * HW, 0); h->h_ridx.assign(B * HW, FL_R_NONE);
    h->h_init_pos.assign(BA, 0); h->h_target.assign(BA, 0); h->h_earliest.assign(BA, 0); h->h_latest.assign(BA, 0);
    h->h_init_r.assign(BA, 0); h->h_target_r.assign(BA, 0); h->h_srank.assign(BA, 0); h->h_maxbr.assign(B, 0);
    h->h_tslot.assign(BA, 0); h->h_spk.assign(BA, 0); h->h_speed.assign(BA, 1.0);
    h->h_ut.assign(BA, 0); h->h_U.assign(B, 0); h->h_R.assign(B, 0); h->h_K.assign(B, 0); h->h_T.assign(B, 0); h->h_mt_pos.assign(B, 624);
    h->h_malf_min.assign(B, 0); h->h_malf_max.assign(B, 0); h->h_thr.assign(B, 0);
    h->h_mt.assign((size_t)B * 624, 0);
    h->h_loaded.assign(B, 0); h->h_dirty.assign(B, 0);
    h->h_key.assign(B, 0); h->h_built.assign(B, 0); h->h_tab.assign(B, 0); h->h_need.assign(B, 0);
    h->need_dev = nullptr;
    memset(&h->obs, 0, sizeof h->obs);
    *out = h;
    return FL_OK;
}

void fl_destroy(fl_batch *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    for (void *p : h->allocs) (void)hipFree(p);
    (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int fl_set_stream(fl_batch *h, void *hip_stream) {
    if (!h) return FL_ERR_ARG;
    h->stream = (hipStream_t)hip_stream;  // NULL = HIP's default (null) stream, which is what torch uses by default
    return FL_OK;
}

int fl_sync(fl_batch *h) {
    if (!h) return FL_ERR_ARG;
    HIPCHK(hipStreamSynchronize(h->stream));
    return FL_OK;
}

int fl_reserve(fl_batch *h, int max_targets, int max_rail_cells) {
    if (!h || max_targets < 0 || max_rail_cells < 0) { set_err("fl_reserve: bad argument"); return FL_ERR_ARG; }
    if (h->committed) { set_err("fl_reserve: the capacities are fixed at the first fl_commit"); return FL_ERR_ARG; }
    if ((long long)max_rail_cells * 4 > 65532) { set_err("fl_reserve: at most 16383 rail cells per env (u16 rail states)"); return FL_ERR_ARG; }
    {   // the reverse-BFS kernel keeps the env's neighbour table, four visited bitmaps and four queues in LDS (fl_dmap.hip)
        FlDev probe = h->d;
        probe.Rcap = max_rail_cells > 1 ? max_rail_cells : 1;
        probe.Ucap = 1;
        if (fl_dmap_fits(probe) != FL_OK) { set_err("fl_reserve: %d rail cells per env do not fit the distance-map kernel's LDS", max_rail_cells); return FL_ERR_ARG; }
    }
    h->reserve_U = max_targets > h->A ? h->A : max_targets;
    h->reserve_R = max_rail_cells;
    return FL_OK;
}

template <typename T>
static int upload_range(fl_batch *h, T *dst, const std::vector<T> &src, size_t first, size_t count) {
    HIPCHK(hipMemcpyAsync(dst + first, src.data() + first, count * sizeof(T), hipMemcpyHostToDevice, h->stream));
    return FL_OK;
}
#define UPLOAD_RANGE(dst, src, first, count)                    \
    do {                                                        \
        int rc_ = upload_range(h, dst, src, (size_t)(first), (size_t)(count)); \
        if (rc_ != FL_OK) return rc_;                           \
    } while (0)

// RailEnvTransitions.transition_list (core/grid/rail_env_grid.py:28-38)
static const uint16_t k_transition_list[11] = {0x0000, 0x8020, 0x9220, 0x8421, 0x9621, 0xCC33, 0x5202, 0x2000, 0x4002, 0x1200, 0xC022};
// rotate_transition (flatland_cutils/src/tool.h:300-335): each nibble rotated right by k, then the word by 4k
static uint32_t rotate_transition(uint32_t cell, int k) {
    uint32_t v = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t nib = (cell >> ((3 - i) * 4)) & 15u;
        nib = ((nib >> k) | (nib << (4 - k))) & 15u;
        v |= nib << ((3 - i) * 4);
    }
    return ((v >> (4 * k)) | (v << (16 - 4 * k))) & 0xFFFFu;
}
// road_type of a cell (flatland_cutils/src/loader.cpp:122-161): the first basic transition one of its four rotations equals
static int road_type_of(uint32_t cell) {
    for (int rot = 0; rot < 4; rot++) {
        const uint32_t t = rot == 0 ? cell : rotate_transition(cell, rot);
        for (int k = 0; k < 11; k++)
            if (k_transition_list[k] == t) return k;
    }
    return 0;
}

// rail-cell index space of env b (fl_internal.h): ridx / rcell / rgrid / nbr / rkey, the agents' and targets' rail indices
static void build_rail_tables(fl_batch *h, int b) {
    const int A = h->A, H = h->H, W = h->W, Rcap = h->d.Rcap, Ucap = h->d.Ucap;
    const size_t HW = (size_t)H * W;
    const uint16_t *grid = &h->h_grid[b * HW];
    uint16_t *ridx = &h->h_ridx[b * HW];
    uint32_t *rcell = &h->h_rcell[(size_t)b * Rcap];
    uint16_t *rgrid = &h->h_rgrid[(size_t)b * Rcap], *nbr = &h->h_nbr[(size_t)b * Rcap * 4];
    int R = 0;
    for (size_t c = 0; c < HW; c++) {
        if (grid[c]) { ridx[c] = (uint16_t)R; rcell[R] = (uint32_t)c; rgrid[R] = grid[c]; h->h_rtype[(size_t)b * Rcap + R] = (uint8_t)road_type_of(grid[c]); R++; }
        else ridx[c] = FL_R_NONE;
    }
    for (int r = R; r < Rcap; r++) { rcell[r] = 0; rgrid[r] = 0; }
    // the step kernel's copy of the grid: one load answers "which transitions" and "does the cell towards m have rail"
    // (check_valid_action, transition_utils.py:47-82)
    uint32_t *gridx = &h->h_gridx[b * HW];
    for (size_t c = 0; c < HW; c++) {
        const int row = (int)(c / W), col = (int)(c % W);
        uint32_t v = grid[c];
        for (int m = 0; m < 4; m++) {
            const int nr = row + (m == 0 ? -1 : m == 2 ? 1 : 0), nc = col + (m == 1 ? 1 : m == 3 ? -1 : 0);
            if (nr >= 0 && nr < H && nc >= 0 && nc < W && grid[(size_t)nr * W + nc] != 0) v |= 1u << (16 + m);
        }
        gridx[c] = v;
    }
    for (int r = 0; r < Rcap; r++) {
        const int row = r < R ? (int)(rcell[r] / W) : 0, col = r < R ? (int)(rcell[r] % W) : 0;
        for (int m = 0; m < 4; m++) {
            const int nr = row + (m == 0 ? -1 : m == 2 ? 1 : 0), nc = col + (m == 1 ? 1 : m == 3 ? -1 : 0);
            nbr[r * 4 + m] = (r < R && nr >= 0 && nr < H && nc >= 0 && nc < W) ? ridx[(size_t)nr * W + nc] : FL_R_NONE;
        }
    }
    int K = R;
    if (H > W) {  // flatland_cutils keys its prediction maps by col * W + row (tool.h:391-398), which collides on tall maps:
                  // rail cells with equal keys share one compact key
        std::vector<uint32_t> keys(R);
        for (int r = 0; r < R; r++) keys[r] = (rcell[r] % W) * (uint32_t)W + rcell[r] / W;
        std::vector<uint32_t> uniq(keys);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        K = (int)uniq.size();
        uint16_t *rkey = &h->h_rkey[(size_t)b * Rcap];
        for (int r = 0; r < R; r++) rkey[r] = (uint16_t)(std::lower_bound(uniq.begin(), uniq.end(), keys[r]) - uniq.begin());
        for (int r = R; r < Rcap; r++) rkey[r] = 0;
    }
    h->h_K[b] = K;
    for (int i = 0; i < A; i++) {
        const size_t g = (size_t)b * A + i;
        h->h_init_r[g] = ridx[h->h_init_pos[g]];
        h->h_target_r[g] = ridx[h->h_target[g]];
    }
    for (int u = 0; u < Ucap; u++) h->h_ut_r[(size_t)b * Ucap + u] = u < h->h_U[b] ? ridx[h->h_ut[(size_t)b * A + u]] : 0;
}

// upload everything fl_load_env staged for envs [b0, b0 + nb) (device arrays exist): one copy per array
static int upload_envs(fl_batch *h, int b0, int nb) {
    const size_t A = h->A, Rcap = h->d.Rcap, Ucap = h->d.Ucap, n = (size_t)nb, b = (size_t)b0;
    const size_t HW = (size_t)h->H * h->W;
    FlDev &d = h->d;
    UPLOAD_RANGE(d.T, h->h_T, b, n); UPLOAD_RANGE(d.mt_pos, h->h_mt_pos, b, n); UPLOAD_RANGE(d.mt, h->h_mt, b * 624, n * 624);
    UPLOAD_RANGE(d.malf_thr, h->h_thr, b, n); UPLOAD_RANGE(d.malf_min, h->h_malf_min, b, n); UPLOAD_RANGE(d.malf_max, h->h_malf_max, b, n);
    UPLOAD_RANGE(d.U, h->h_U, b, n); UPLOAD_RANGE(d.R, h->h_R, b, n); UPLOAD_RANGE(d.K, h->h_K, b, n);
    UPLOAD_RANGE(d.grid, h->h_gridx, b * HW, n * HW); UPLOAD_RANGE(d.ridx, h->h_ridx, b * HW, n * HW);
    UPLOAD_RANGE(d.rgrid, h->h_rgrid, b * Rcap, n * Rcap); UPLOAD_RANGE(d.rtype, h->h_rtype, b * Rcap, n * Rcap); UPLOAD_RANGE(d.nbr, h->h_nbr, b * Rcap * 4, n * Rcap * 4);
    if (d.rkey) UPLOAD_RANGE(d.rkey, h->h_rkey, b * Rcap, n * Rcap);
    UPLOAD_RANGE(d.ut_r, h->h_ut_r, b * Ucap, n * Ucap);
    const size_t g0 = b * A, ng = n * A;
    UPLOAD_RANGE(d.init_pos, h->h_init_pos, g0, ng); UPLOAD_RANGE(d.target, h->h_target, g0, ng);
    UPLOAD_RANGE(d.init_r, h->h_init_r, g0, ng); UPLOAD_RANGE(d.target_r, h->h_target_r, g0, ng); UPLOAD_RANGE(d.srank, h->h_srank, g0, ng);
    UPLOAD_RANGE(d.earliest, h->h_earliest, g0, ng); UPLOAD_RANGE(d.latest, h->h_latest, g0, ng);
    UPLOAD_RANGE(d.tslot, h->h_tslot, g0, ng); UPLOAD_RANGE(d.spk, h->h_spk, g0, ng); UPLOAD_RANGE(d.speed, h->h_speed, g0, ng);
    return FL_OK;
}

int fl_load_env(fl_batch *h, int b, const uint16_t *grid, const int32_t *init_pos, const int32_t *init_dir,
                const int32_t *target, const double *speed, const int32_t *earliest, const int32_t *latest,
                int max_episode_steps, uint64_t malf_threshold, int malf_min, int malf_max, const uint32_t *mt_key,
                int mt_pos) {
    if (!h || b < 0 || b >= h->B) { set_err("fl_load_env: env index out of range"); return FL_ERR_ARG; }
    if (!grid || !init_pos || !init_dir || !target || !speed || !earliest || !latest || !mt_key) { set_err("fl_load_env: null argument"); return FL_ERR_ARG; }
    const int A = h->A, H = h->H, W = h->W;
    const size_t HW = (size_t)H * W;
    if (mt_pos < 0 || mt_pos > 624) { set_err("fl_load_env: mt_pos out of range"); return FL_ERR_ARG; }
    if (malf_max < malf_min || malf_min < 0 || malf_max > 60000) { set_err("fl_load_env: bad malfunction duration range"); return FL_ERR_ARG; }
    size_t rail_cells = 0;
    for (size_t c = 0; c < HW; c++) rail_cells += grid[c] != 0;
    if (rail_cells * 4 > 65532) { set_err("fl_load_env: %zu rail cells exceed the u16 rail-state / distance range", rail_cells); return FL_ERR_ARG; }
    // validate everything before touching the staged copy of env b (a refused call leaves it as it was)
    std::vector<int> ut;
    std::vector<int> tslot(A);
    for (int i = 0; i < A; i++) {
        const int ir = init_pos[2 * i], ic = init_pos[2 * i + 1], tr = target[2 * i], tc = target[2 * i + 1];
        if (ir < 0 || ir >= H || ic < 0 || ic >= W || tr < 0 || tr >= H || tc < 0 || tc >= W || init_dir[i] < 0 || init_dir[i] > 3) {
            set_err("fl_load_env: agent %d position/direction out of range", i);
            return FL_ERR_ARG;
        }
        if (grid[(size_t)ir * W + ic] == 0 || grid[(size_t)tr * W + tc] == 0) { set_err("fl_load_env: agent %d starts or ends on a cell without rail", i); return FL_ERR_ARG; }
        if (!(speed[i] > 0.0) || speed[i] > 1.0) { set_err("fl_load_env: agent %d speed %g not in (0, 1]", i, speed[i]); return FL_ERR_ARG; }
        const int max_count = (int)(1.0 / speed[i]) - 1;  // SpeedCounter.max_count (step_utils/speed_counter.py:39-41)
        if (max_count < 0 || max_count > FL_MAX_SPEED_COUNT) { set_err("fl_load_env: agent %d speed %g unsupported (max_count %d)", i, speed[i], max_count); return FL_ERR_ARG; }
        // unique targets in first-seen order (distance_map.py:71-79)
        const int tcell = tr * W + tc;
        size_t u = 0;
        for (; u < ut.size(); u++)
            if (ut[u] == tcell) break;
        if (u == ut.size()) ut.push_back(tcell);
        tslot[i] = (int)u;
    }
    if (h->committed && ((int)ut.size() > h->d.Ucap || (int)rail_cells > h->d.Rcap)) {
        set_err("fl_load_env: env %d has %zu unique targets / %zu rail cells, the batch was committed for %d / %d (fl_reserve)", b,
                ut.size(), rail_cells, h->d.Ucap, h->d.Rcap);
        return FL_ERR_CAPACITY;
    }
    {   // another map or other unique targets than the staged ones: whatever env b's slabs hold is stale, whatever its content key says
        bool same = h->h_loaded[b] && h->h_U[b] == (int)ut.size() && memcmp(&h->h_grid[b * HW], grid, HW * 2) == 0;
        for (size_t u = 0; same && u < ut.size(); u++) same = h->h_ut[(size_t)b * A + u] == ut[u];
        if (!same) h->h_built[b] = 0;
    }
    memcpy(&h->h_grid[b * HW], grid, HW * 2);
    for (int i = 0; i < A; i++) {
        const size_t g = (size_t)b * A + i;
        h->h_init_pos[g] = init_pos[2 * i] * W + init_pos[2 * i + 1];
        h->h_target[g] = target[2 * i] * W + target[2 * i + 1];
        h->h_earliest[g] = earliest[i];
        h->h_latest[g] = latest[i];
        h->h_speed[g] = speed[i];
        h->h_spk[g] = (uint32_t)init_dir[i] | ((uint32_t)((int)(1.0 / speed[i]) - 1) << 2);
        h->h_tslot[g] = tslot[i];
        h->h_ut[g] = i < (int)ut.size() ? ut[i] : 0;
        int rank = 0;
        for (int j = 0; j < A; j++) rank += speed[j] < speed[i];
        h->h_srank[g] = (uint16_t)rank;
    }
    {
        int mb = 0;
        for (size_t c = 0; c < HW; c++)
            for (int dd = 0; dd < 4; dd++) mb = std::max(mb, __builtin_popcount((grid[c] >> ((3 - dd) * 4)) & 15u));
        h->h_maxbr[b] = mb;
    }
    h->h_U[b] = (int)ut.size();
    h->h_R[b] = (int)rail_cells;
    h->h_T[b] = max_episode_steps;
    h->h_thr[b] = malf_threshold;
    h->h_malf_min[b] = malf_min;
    h->h_malf_max[b] = malf_max;
    memcpy(&h->h_mt[(size_t)b * 624], mt_key, 624 * 4);
    h->h_mt_pos[b] = mt_pos;
    h->h_loaded[b] = 1;
    h->h_dirty[b] = 1;
    return FL_OK;
}

// content id of env b's map side: FNV-1a over the rail grid and the unique targets (what the static tables are a function of)
static uint64_t map_content_key(const fl_batch *h, int b) {
    const size_t HW = (size_t)h->H * h->W;
    uint64_t k = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) {
        const unsigned char *c = (const unsigned char *)p;
        for (size_t i = 0; i < n; i++) { k ^= c[i]; k *= 1099511628211ull; }
    };
    mix(&h->h_grid[b * HW], HW * 2);
    const int U = h->h_U[b];
    mix(&U, sizeof U);
    mix(&h->h_ut[(size_t)b * h->A], (size_t)U * sizeof(int));
    return k ? k : 1;
}
static bool same_map_content(const fl_batch *h, int a, int b) {
    const size_t HW = (size_t)h->H * h->W;
    return h->h_U[a] == h->h_U[b] && memcmp(&h->h_grid[a * HW], &h->h_grid[b * HW], HW * 2) == 0 &&
           memcmp(&h->h_ut[(size_t)a * h->A], &h->h_ut[(size_t)b * h->A], (size_t)h->h_U[a] * sizeof(int)) == 0;
}
// owners of the static tables (h_tab -> FlDev::tab) and the owners whose slabs have to be (re)built (h_need -> need_dev)
static int share_tables(fl_batch *h) {
    static const bool no_share = getenv("FL_NO_SHARED_TABLES") != nullptr;   // diagnostic: every env reads its own slabs
    const int B = h->B;
    for (int b = 0; b < B; b++)
        if (h->h_dirty[b]) h->h_key[b] = map_content_key(h, b);
    std::vector<std::pair<uint64_t, int>> order(B);
    for (int b = 0; b < B; b++) order[b] = {h->h_key[b], b};
    std::sort(order.begin(), order.end());
    for (int i = 0; i < B;) {
        int j = i;
        while (j < B && order[j].first == order[i].first) j++;
        // envs of one key, ascending: the first env with identical content is the owner (a hash collision keeps its own tables)
        for (int k = i; k < j; k++) {
            const int b = order[k].second;
            int owner = b;
            if (!no_share)
                for (int q = i; q < k; q++)
                    if (h->h_tab[order[q].second] == order[q].second && same_map_content(h, order[q].second, b)) { owner = order[q].second; break; }
            h->h_tab[b] = owner;
        }
        i = j;
    }
    // (h_built is set by fl_commit once the table kernels of the h_need envs have run: a commit that fails half way rebuilds them)
    for (int b = 0; b < B; b++) h->h_need[b] = h->h_tab[b] == b && h->h_built[b] != h->h_key[b];
    HIPCHK(hipMemcpyAsync(h->d.tab, h->h_tab.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->need_dev, h->h_need.data(), B, hipMemcpyHostToDevice, h->stream));
    return FL_OK;
}

int fl_commit(fl_batch *h) {
    if (!h) return FL_ERR_ARG;
    for (int b = 0; b < h->B; b++)
        if (!h->h_loaded[b]) { set_err("fl_commit: env %d was never loaded", b); return FL_ERR_ARG; }
    HIPCHK(hipSetDevice(h->device));
    const int B = h->B, A = h->A;
    const size_t BA = (size_t)B * A, HW = (size_t)h->H * h->W;
    FlDev &d = h->d;
    if (!h->committed) {
        // a first commit that failed half way (allocation, LDS fit) left device pointers behind: release them, a retry starts clean
        if (!h->allocs.empty()) {
            for (void *p : h->allocs) (void)hipFree(p);
            h->allocs.clear();
            const int B0 = h->d.B, A0 = h->d.A, H0 = h->d.H, W0 = h->d.W;
            memset(&h->d, 0, sizeof h->d);
            h->d.B = B0; h->d.A = A0; h->d.H = H0; h->d.W = W0;
            memset(&h->obs, 0, sizeof h->obs);
            h->mask_dev = nullptr;
            h->need_dev = nullptr;
            std::fill(h->h_built.begin(), h->h_built.end(), 0);
        }
        int Ucap = h->reserve_U > 1 ? h->reserve_U : 1, Rcap = h->reserve_R > 1 ? h->reserve_R : 1;
        for (int b = 0; b < B; b++) {
            Ucap = h->h_U[b] > Ucap ? h->h_U[b] : Ucap;
            Rcap = h->h_R[b] > Rcap ? h->h_R[b] : Rcap;
        }
        d.Ucap = Ucap; d.Rcap = Rcap;
        const size_t Scap = (size_t)Rcap * 4;
        {   // before anything is allocated: do the capacities fit the table kernels' LDS?
            const int rc_dm = fl_dmap_prepare(d);
            if (rc_dm != FL_OK) { set_err("fl_commit: %d rail cells per env do not fit the distance-map kernel's LDS", Rcap); return rc_dm; }
        }
        h->h_rcell.assign((size_t)B * Rcap, 0); h->h_rgrid.assign((size_t)B * Rcap, 0); h->h_rtype.assign((size_t)B * Rcap, 0); h->h_nbr.assign((size_t)B * Scap, FL_R_NONE);
        h->h_ut_r.assign((size_t)B * Ucap, 0);
        if (h->H > h->W) h->h_rkey.assign((size_t)B * Rcap, 0);
        DALLOC(d.t, B); DALLOC(d.T, B); DALLOC(d.done_all, B); DALLOC(d.mt_pos, B); DALLOC(d.mt, (size_t)B * 624);
        DALLOC(d.malf_thr, B); DALLOC(d.malf_min, B); DALLOC(d.malf_max, B); DALLOC(d.U, B); DALLOC(d.R, B); DALLOC(d.K, B);
        DALLOC(d.err, B); DALLOC(d.env_list, B + 1); DALLOC(d.metrics, (size_t)B * 4); DALLOC(d.last_episode, (size_t)B * 2); DALLOC(d.score_sums, (size_t)B * 3);
        DALLOC(d.grid, B * HW); DALLOC(d.ridx, B * HW);
        DALLOC(d.rgrid, (size_t)B * Rcap); DALLOC(d.rtype, (size_t)B * Rcap); DALLOC(d.nbr, (size_t)B * Scap); DALLOC(d.snext, (size_t)B * Scap);
        d.rkey = nullptr;
        if (h->H > h->W) DALLOC(d.rkey, (size_t)B * Rcap);
        DALLOC(d.ut_r, (size_t)B * Ucap);
        DALLOC(d.dm, (size_t)B * Ucap * Scap); DALLOC(d.seg, (size_t)B * Scap);
        DALLOC(d.nh, (size_t)B * Ucap * Rcap); DALLOC(d.hop8, (size_t)B * Ucap * Scap);
        DALLOC(d.init_pos, BA); DALLOC(d.target, BA); DALLOC(d.init_r, BA); DALLOC(d.target_r, BA); DALLOC(d.srank, BA);
        DALLOC(d.earliest, BA); DALLOC(d.latest, BA); DALLOC(d.tslot, BA);
        DALLOC(d.spk, BA); DALLOC(d.speed, BA);
        DALLOC(d.pos, BA); DALLOC(d.old_pos, BA); DALLOC(d.arrival, BA); DALLOC(d.malf, BA); DALLOC(d.pk, BA);
        DALLOC(h->mask_dev, B); DALLOC(h->need_dev, B); DALLOC(d.tab, B);
        if (fl_step_prepare() != FL_OK) { set_err("fl_commit: hipFuncSetAttribute failed"); return FL_ERR_HIP; }
        int rc = fl_obs_alloc(h->obs, d, h->stream, h->allocs);
        if (rc != FL_OK) { set_err("fl_commit: observation scratch allocation failed"); return rc; }
    }
    h->obs.h_R = h->h_R.data();   // (host copy: which envs fit a fixed launch class, fl_obs.hip)
    d.max_branch = 0;
    for (int b = 0; b < B; b++) d.max_branch = std::max(d.max_branch, h->h_maxbr[b]);
    // (re)build the host tables of every env loaded since the last commit and upload them; the device tables of exactly
    // those envs are rebuilt below, their agents reset (fresh)
    bool any = false;
    for (int b = 0; b < B; b++) {
        if (!h->h_dirty[b]) continue;
        any = true;
        build_rail_tables(h, b);
    }
    if (!any) return FL_OK;
    for (int b = 0; b < B;) {  // one set of copies per run of consecutive dirty envs (the first commit: one set in all)
        if (!h->h_dirty[b]) { b++; continue; }
        int e = b;
        while (e < B && h->h_dirty[e]) e++;
        const int rc = upload_envs(h, b, e - b);
        if (rc != FL_OK) return rc;
        b = e;
    }
    const bool all = !h->committed;
    if (!all) HIPCHK(hipMemcpyAsync(h->mask_dev, h->h_dirty.data(), B, hipMemcpyHostToDevice, h->stream));
    const uint8_t *mask = all ? nullptr : h->mask_dev;
    // Shared static tables (FlDev::tab): envs with the same rail grid and the same unique targets read ONE set of tables, the
    // slabs of the first such env (the owner).  Built here: the owners whose slabs do not hold their content yet -- a new map, or
    // an env that becomes an owner because the previous one was replaced.
    {
        int rc_t = share_tables(h);
        if (rc_t != FL_OK) return rc_t;
    }
    fl_launch_env_list(d, h->need_dev, h->stream);
    fl_launch_distance_maps(d, h->need_dev, h->stream);
    HIPCHK(hipGetLastError());
    fl_launch_segments(d, h->need_dev, h->stream);
    HIPCHK(hipGetLastError());
    fl_launch_nexthop(d, h->need_dev, h->stream);
    HIPCHK(hipGetLastError());
    fl_launch_hop8(d, h->need_dev, h->stream);
    HIPCHK(hipGetLastError());
    fl_launch_reset(d, mask, 1, h->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int b = 0; b < B; b++)
        if (h->h_need[b]) h->h_built[b] = h->h_key[b];   // the owners' slabs hold their content now
    std::fill(h->h_dirty.begin(), h->h_dirty.end(), 0);
    h->committed = true;
    return fl_check(h);
}

#define NEED_COMMIT(h)                                                         \
    do {                                                                       \
        if (!(h) || !(h)->committed) { set_err("handle not committed"); return FL_ERR_ARG; } \
        HIPCHK(hipSetDevice((h)->device));                                     \
    } while (0)

int fl_set_rng(fl_batch *h, const uint32_t *mt_key, const int32_t *mt_pos) {
    NEED_COMMIT(h);
    for (int b = 0; b < h->B; b++)
        if (mt_pos[b] < 0 || mt_pos[b] > 624) { set_err("fl_set_rng: mt_pos out of range"); return FL_ERR_ARG; }
    HIPCHK(hipMemcpyAsync(h->d.mt, mt_key, (size_t)h->B * 624 * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d.mt_pos, mt_pos, (size_t)h->B * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return FL_OK;
}

int fl_get_rng(fl_batch *h, uint32_t *mt_key, int32_t *mt_pos) {
    NEED_COMMIT(h);
    HIPCHK(hipMemcpyAsync(mt_key, h->d.mt, (size_t)h->B * 624 * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(mt_pos, h->d.mt_pos, (size_t)h->B * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return FL_OK;
}

int fl_reset(fl_batch *h, const uint8_t *mask, int fresh) {
    NEED_COMMIT(h);
    // the mask goes through a device buffer owned by the handle (no allocation, no host synchronisation here: a copy from
    // pageable memory returns once the source has been staged)
    if (mask) HIPCHK(hipMemcpyAsync(h->mask_dev, mask, h->B, hipMemcpyHostToDevice, h->stream));
    return fl_reset_dev(h, mask ? h->mask_dev : nullptr, fresh);
}

int fl_reset_dev(fl_batch *h, const uint8_t *mask_dev, int fresh) {
    NEED_COMMIT(h);
    // (the observation side keeps no state of its own across steps: the sticky deadlock flags live in the agents' packed word and
    // are cleared by k_reset -- flatland_cutils rebuilds its DeadlockChecker in TreeObsForRailEnv::reset(), treeobs.cpp:22-28)
    fl_launch_reset(h->d, mask_dev, fresh, h->stream);
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_step(fl_batch *h, const uint8_t *actions_dev, int32_t *rewards_dev, uint8_t *dones_dev, uint8_t *done_all_dev,
            int auto_reset) {
    NEED_COMMIT(h);
    if (!actions_dev || !rewards_dev || !dones_dev || !done_all_dev) { set_err("fl_step: null buffer"); return FL_ERR_ARG; }
    fl_launch_step(h->d, actions_dev, 0, 0, 0, rewards_dev, dones_dev, done_all_dev, auto_reset, h->stream);
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_step_synth(fl_batch *h, uint32_t seed, uint32_t stream_base, int kind, int32_t *rewards_dev, uint8_t *dones_dev,
                  uint8_t *done_all_dev, int auto_reset) {
    NEED_COMMIT(h);
    if (!rewards_dev || !dones_dev || !done_all_dev || kind < 0 || kind > 2) { set_err("fl_step_synth: bad argument"); return FL_ERR_ARG; }
    fl_launch_step(h->d, nullptr, seed, stream_base, kind, rewards_dev, dones_dev, done_all_dev, auto_reset, h->stream);
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_step_obs(fl_batch *h, const uint8_t *actions_dev, uint32_t seed, uint32_t stream_base, int kind, int32_t *rewards_dev,
                uint8_t *dones_dev, uint8_t *done_all_dev, int flags, int max_nodes, int pred_depth, float *attr_dev,
                float *forest_dev, int32_t *adjacency_dev, int32_t *node_order_dev, int32_t *edge_order_dev,
                uint8_t *valid_actions_dev, double *props_dev, int tree_max_depth, int tree_pred_depth, double *tree_out_dev) {
    NEED_COMMIT(h);
    if (!rewards_dev || !dones_dev || !done_all_dev || kind < 0 || kind > 2) { set_err("fl_step_obs: bad step argument"); return FL_ERR_ARG; }
    if (max_nodes < 4 || max_nodes > FL_OBS_MAX_NODES || pred_depth < 1 || pred_depth > FL_OBS_MAX_PRED || tree_max_depth < 0 ||
        tree_max_depth > FL_MAX_TREE_DEPTH || (tree_max_depth > 0 && (tree_pred_depth < 0 || tree_pred_depth > pred_depth || !tree_out_dev))) {
        set_err("fl_step_obs: max_nodes in [4,%d], pred_depth in [1,%d], tree depth in [0,%d], 0 <= tree_pred_depth <= pred_depth",
                FL_OBS_MAX_NODES, FL_OBS_MAX_PRED, FL_MAX_TREE_DEPTH);
        return FL_ERR_ARG;
    }
    if (!attr_dev || !forest_dev || !adjacency_dev || !node_order_dev || !edge_order_dev || !valid_actions_dev) {
        set_err("fl_step_obs: null output buffer");
        return FL_ERR_ARG;
    }
    // every argument error is raised BEFORE anything is launched: a depth-4 tree needs compact node tables (fl_obs_tree's own check,
    // which would otherwise fire after the envs have advanced a step and the cutils launch has set its sticky deadlock bits)
    if (tree_max_depth > 3 && (h->d.max_branch > 2 || getenv("FL_OBS_NO_COMPACT") != nullptr)) {
        set_err("fl_step_obs: max_depth 4 needs a grid on which no direction of a cell has more than two transitions (every Flatland rail cell type) and the compact node tables; this batch has %d (nothing was launched: the envs have not advanced)", h->d.max_branch);
        return FL_ERR_ARG;
    }
    // TWO launches (k_step, then the observation kernel) back to back on the handle's stream behind this one call.  A single fused launch was
    // measured and is slower: the step wants one lane per agent and few wavefronts, the builders 16 wavefronts, and the second launch's dispatch overlaps the first.
    fl_launch_step(h->d, actions_dev, seed, stream_base, kind, rewards_dev, dones_dev, done_all_dev, flags, h->stream);
    HIPCHK(hipGetLastError());
    if (tree_max_depth > 3 || (tree_max_depth > 0 && max_nodes > 32)) return fl_obs_cutils_tree(h, max_nodes, pred_depth, attr_dev, forest_dev, adjacency_dev, node_order_dev, edge_order_dev,
                                                      valid_actions_dev, props_dev, tree_max_depth, tree_pred_depth, tree_out_dev);
    const int rc = tree_max_depth > 0 ? fl_launch_obs_both(h->obs, h->d, max_nodes, pred_depth, attr_dev, forest_dev, adjacency_dev, node_order_dev,
                                                           edge_order_dev, valid_actions_dev, props_dev, tree_max_depth, tree_pred_depth,
                                                           tree_out_dev, h->stream)
                                      : fl_launch_obs_cutils(h->obs, h->d, max_nodes, pred_depth, attr_dev, forest_dev, adjacency_dev, node_order_dev,
                                                             edge_order_dev, valid_actions_dev, props_dev, h->stream);
    if (rc != FL_OK) { set_err("fl_step_obs: no launch configuration: %d rail cells and %d agents per env do not fit the observation kernels' LDS (160 KiB a workgroup), or the sizes are out of range", h->d.Rcap, h->d.A); return rc; }
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_check(fl_batch *h) {
    NEED_COMMIT(h);
    std::vector<int> err(h->B);
    HIPCHK(hipMemcpyAsync(err.data(), h->d.err, (size_t)h->B * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int b = 0; b < h->B; b++) {
        if (err[b]) {
            const int e = err[b];
            HIPCHK(hipMemsetAsync(h->d.err, 0, (size_t)h->B * 4, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            const char *msg = e == FL_ERR_EPISODE_DONE ? "Episode is done, cannot call step()"
                              : e == FL_ERR_STATE_SYNC ? "agent state / position desync"
                              : e == FL_ERR_ZERO_TRANSITION ? "WRONG CELL TYPE detected in tree-search (0 transitions possible)"
                              : e == FL_ERR_CAPACITY ? "internal capacity exceeded"
                              : e == FL_ERR_ARG ? "FL_OBS_KEEP_TREE_ROWS: the tree buffer is not the previous launch's, or was modified in between (FL_OBS_KEEP_VERIFY)" : "kernel error";
            set_err("env %d: %s", b, msg);
            return e;
        }
    }
    return FL_OK;
}

int fl_metrics(fl_batch *h, int64_t *out4_dev, int reset) {
    NEED_COMMIT(h);
    if (!out4_dev) { set_err("fl_metrics: null buffer"); return FL_ERR_ARG; }
    fl_launch_metrics(h->d, (long long *)out4_dev, reset, h->stream);
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_scores(fl_batch *h, double *out3_dev, int reset) {
    NEED_COMMIT(h);
    if (!out3_dev) { set_err("fl_scores: null buffer"); return FL_ERR_ARG; }
    // (call it BEFORE fl_metrics(reset): the episode count it reports is fl_metrics' counter)
    fl_launch_scores(h->d, out3_dev, reset, h->stream);
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_info(fl_batch *h, uint8_t *action_required_dev, int32_t *malfunction_dev, uint8_t *state_dev, double *scores_dev) {
    NEED_COMMIT(h);
    fl_launch_info(h->d, action_required_dev, malfunction_dev, state_dev, scores_dev, h->stream);
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_policy_pack(int B, int A, int E, const int32_t *adjacency_dev, const int32_t *node_order_dev,
                   const int32_t *edge_order_dev, int64_t *adjacency_out_dev, int64_t *node_order_out_dev,
                   int64_t *edge_order_out_dev, void *hip_stream) {
    if (B <= 0 || A <= 0 || E <= 0 || !adjacency_dev || !node_order_dev || !edge_order_dev || !adjacency_out_dev ||
        !node_order_out_dev || !edge_order_out_dev) { set_err("fl_policy_pack: bad argument"); return FL_ERR_ARG; }
    fl_launch_policy_pack(B, A, E, adjacency_dev, node_order_dev, edge_order_dev, (long long *)adjacency_out_dev,
                          (long long *)node_order_out_dev, (long long *)edge_order_out_dev, (hipStream_t)hip_stream);
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_get_state(fl_batch *h, int32_t *state, int32_t *elapsed) {
    NEED_COMMIT(h);
    const size_t BA = (size_t)h->B * h->A;
    std::vector<int> pos(BA), old_pos(BA), arrival(BA);
    std::vector<uint32_t> malf(BA), pk(BA);
    HIPCHK(hipMemcpyAsync(pos.data(), h->d.pos, BA * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(old_pos.data(), h->d.old_pos, BA * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(arrival.data(), h->d.arrival, BA * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(malf.data(), h->d.malf, BA * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(pk.data(), h->d.pk, BA * 4, hipMemcpyDeviceToHost, h->stream));
    if (elapsed) HIPCHK(hipMemcpyAsync(elapsed, h->d.t, (size_t)h->B * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int W = h->W;
    if (state) {
        for (size_t g = 0; g < BA; g++) {
            int32_t *o = state + g * FL_STATE_COLS;
            const uint32_t p = pk[g];
            o[0] = pos[g] < 0 ? -1 : pos[g] / W;
            o[1] = pos[g] < 0 ? -1 : pos[g] % W;
            o[2] = (int)PK_DIR(p);
            o[3] = (int)PK_STATE(p);
            o[4] = (int)(malf[g] & 0xFFFF);
            o[5] = (int)(malf[g] >> 16);
            o[6] = (int)PK_SCOUNT(p);
            o[7] = (int)PK_SAVED(p);
            o[8] = arrival[g];
            o[9] = old_pos[g] < 0 ? -1 : old_pos[g] / W;
            o[10] = old_pos[g] < 0 ? -1 : old_pos[g] % W;
            o[11] = PK_OLD_DIR(p) == 4 ? -1 : (int)PK_OLD_DIR(p);
        }
    }
    return FL_OK;
}

int fl_get_state_aux(fl_batch *h, int32_t *aux) {
    NEED_COMMIT(h);
    if (!aux) { set_err("fl_get_state_aux: null buffer"); return FL_ERR_ARG; }
    const size_t BA = (size_t)h->B * h->A;
    std::vector<uint32_t> pk(BA);
    HIPCHK(hipMemcpyAsync(pk.data(), h->d.pk, BA * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (size_t g = 0; g < BA; g++) {
        int32_t *o = aux + g * FL_AUX_COLS;
        o[0] = PK_PREV(pk[g]) == 7 ? -1 : (int)PK_PREV(pk[g]);
        o[1] = (int)PK_SIGMALF(pk[g]);
        o[2] = (int)PK_DEADLOCK(pk[g]);
        o[3] = (int)PK_DONE(pk[g]);
    }
    return FL_OK;
}

int fl_set_state(fl_batch *h, const int32_t *state, const int32_t *aux, const int32_t *elapsed, const uint8_t *done_all) {
    NEED_COMMIT(h);
    if (!state) { set_err("fl_set_state: null state"); return FL_ERR_ARG; }
    const size_t BA = (size_t)h->B * h->A;
    const int H = h->H, W = h->W;
    std::vector<int> pos(BA), old_pos(BA), arrival(BA);
    std::vector<uint32_t> malf(BA), pk(BA);
    for (size_t g = 0; g < BA; g++) {
        const int32_t *o = state + g * FL_STATE_COLS;
        const int r = o[0], c = o[1], dir = o[2], st = o[3], mf = o[4], nmf = o[5], sc = o[6], sv = o[7], orow = o[9], ocol = o[10], od = o[11];
        const bool on = r >= 0;
        if ((on && (r >= H || c < 0 || c >= W)) || dir < 0 || dir > 3 || st < ST_WAITING || st > ST_DONE || mf < 0 || mf > 0xFFFF ||
            nmf < 0 || nmf > 0xFFFF || sc < 0 || sc > FL_MAX_SPEED_COUNT || sv < 0 || sv > 3 || od < -1 || od > 3 ||
            (orow >= 0 && (orow >= H || ocol < 0 || ocol >= W))) {
            set_err("fl_set_state: env %zu agent %zu: value out of range", g / h->A, g % h->A);
            return FL_ERR_ARG;
        }
        // env_utils.state_position_sync_check (step_utils/env_utils.py:45-52)
        if ((st >= ST_MOVING && st <= ST_MALF && !on) || (st <= ST_MALF_OFF && on)) {
            set_err("fl_set_state: env %zu agent %zu: state %d does not match position", g / h->A, g % h->A, st);
            return FL_ERR_STATE_SYNC;
        }
        pos[g] = on ? r * W + c : -1;
        old_pos[g] = orow >= 0 ? orow * W + ocol : -1;
        // an agent on the map stands on a rail cell: the kernels index the env's rail tables with the rail index of its position
        // (a cell without rail has none)
        const size_t hw0 = (g / h->A) * (size_t)H * W;
        if ((on && h->h_ridx[hw0 + pos[g]] == FL_R_NONE) || (orow >= 0 && h->h_ridx[hw0 + old_pos[g]] == FL_R_NONE)) {
            set_err("fl_set_state: env %zu agent %zu: position (%d, %d) / old position (%d, %d) is not a rail cell", g / h->A, g % h->A, r, c, orow, ocol);
            return FL_ERR_STATE_SYNC;
        }
        arrival[g] = o[8];
        malf[g] = (uint32_t)mf | ((uint32_t)nmf << 16);
        int prev = 7, sig = mf > 0, dead = 0, done = st == ST_DONE;
        if (aux) {
            const int32_t *x = aux + g * FL_AUX_COLS;
            if (x[0] < -1 || x[0] > ST_DONE || (x[1] | x[2] | x[3]) < 0 || x[1] > 1 || x[2] > 1 || x[3] > 1) {
                set_err("fl_set_state: env %zu agent %zu: aux value out of range", g / h->A, g % h->A);
                return FL_ERR_ARG;
            }
            prev = x[0] < 0 ? 7 : x[0]; sig = x[1]; dead = x[2]; done = x[3];
        }
        pk[g] = pk_make((uint32_t)dir, od < 0 ? 4u : (uint32_t)od, (uint32_t)st, (uint32_t)prev, (uint32_t)sv, (uint32_t)sc,
                        (uint32_t)sig, (uint32_t)dead, (uint32_t)done);
    }
    if (elapsed)
        for (int b = 0; b < h->B; b++)
            if (elapsed[b] < 0) { set_err("fl_set_state: negative elapsed steps"); return FL_ERR_ARG; }
    HIPCHK(hipMemcpyAsync(h->d.pos, pos.data(), BA * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d.old_pos, old_pos.data(), BA * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d.arrival, arrival.data(), BA * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d.malf, malf.data(), BA * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d.pk, pk.data(), BA * 4, hipMemcpyHostToDevice, h->stream));
    if (elapsed) HIPCHK(hipMemcpyAsync(h->d.t, elapsed, (size_t)h->B * 4, hipMemcpyHostToDevice, h->stream));
    if (done_all) HIPCHK(hipMemcpyAsync(h->d.done_all, done_all, (size_t)h->B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));  // the staging vectors are locals
    return FL_OK;
}

int fl_motion_check(int device, int n_cases, const int32_t *offsets, const int32_t *cur, const int32_t *nxt, uint8_t *can_move) {
    if (n_cases <= 0 || !offsets || !cur || !nxt || !can_move) { set_err("fl_motion_check: bad argument"); return FL_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_err("fl_motion_check: no HIP device visible"); return FL_ERR_HIP; }
    int max_agents = 0;
    for (int c = 0; c < n_cases; c++) {
        const int n = offsets[c + 1] - offsets[c];
        if (n < 0 || n > 1024) { set_err("fl_motion_check: case %d has %d agents (0..1024 supported)", c, n); return FL_ERR_ARG; }
        max_agents = n > max_agents ? n : max_agents;
    }
    const int total = offsets[n_cases] - offsets[0];
    for (int k = offsets[0]; k < offsets[n_cases]; k++)
        if (cur[k] < -1 || nxt[k] < -1 || cur[k] >= (1 << 28) || nxt[k] >= (1 << 28)) { set_err("fl_motion_check: cell id out of range"); return FL_ERR_ARG; }
    if (offsets[0] != 0) { set_err("fl_motion_check: offsets must start at 0"); return FL_ERR_ARG; }
    HIPCHK(hipSetDevice(device));
    int *d_off = nullptr, *d_cur = nullptr, *d_nxt = nullptr;
    uint8_t *d_out = nullptr;
    const size_t nb = (size_t)(total > 0 ? total : 1);
    hipError_t e = hipMalloc((void **)&d_off, (size_t)(n_cases + 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cur, nb * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_nxt, nb * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, nb);
    if (e == hipSuccess) e = hipMemcpy(d_off, offsets, (size_t)(n_cases + 1) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && total > 0) e = hipMemcpy(d_cur, cur, nb * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && total > 0) e = hipMemcpy(d_nxt, nxt, nb * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        fl_launch_motion_check(n_cases, max_agents, d_off, d_cur, d_nxt, d_out, nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && total > 0) e = hipMemcpy(can_move, d_out, nb, hipMemcpyDeviceToHost);
    (void)hipFree(d_off); (void)hipFree(d_cur); (void)hipFree(d_nxt); (void)hipFree(d_out);
    if (e != hipSuccess) { set_err("fl_motion_check: %s", hipGetErrorString(e)); return FL_ERR_HIP; }
    return FL_OK;
}

int fl_distance_map(fl_batch *h, int b, int *n_targets, uint16_t *dm, int32_t *target_slot) {
    NEED_COMMIT(h);
    if (b < 0 || b >= h->B || !n_targets) { set_err("fl_distance_map: bad argument"); return FL_ERR_ARG; }
    const int U = h->h_U[b], R = h->h_R[b];
    *n_targets = U;
    const size_t HW = (size_t)h->H * h->W, Scap = (size_t)h->d.Rcap * 4;
    if (target_slot) memcpy(target_slot, &h->h_tslot[(size_t)b * h->A], (size_t)h->A * 4);
    if (dm) {  // the resident map holds rail states only: expand it to the reference's dense [U][H][W][4]
        std::vector<uint16_t> rs((size_t)U * Scap);
        HIPCHK(hipMemcpyAsync(rs.data(), h->d.dm + (size_t)h->h_tab[b] * h->d.Ucap * Scap, rs.size() * 2, hipMemcpyDeviceToHost, h->stream));   // (the owner's slab)
        HIPCHK(hipStreamSynchronize(h->stream));
        const uint32_t *rcell = &h->h_rcell[(size_t)b * h->d.Rcap];
        for (size_t k = 0; k < (size_t)U * HW * 4; k++) dm[k] = FL_INF16;
        for (int u = 0; u < U; u++)
            for (int r = 0; r < R; r++)
                memcpy(&dm[((size_t)u * HW + rcell[r]) * 4], &rs[(size_t)u * Scap + (size_t)r * 4], 8);
    }
    return FL_OK;
}

static int rebuild_tables(fl_batch *h, const uint8_t *mask_dev) {
    // every env: each OWNER once (host mask); a device mask of envs: the owners of the masked envs (FlDev::tab), an owner once per
    // masked env that reads it -- DistanceMap.reset() + _compute() of every env that resets, as the reference does it
    if (!mask_dev) {
        for (int b = 0; b < h->B; b++) h->h_need[b] = h->h_tab[b] == b;
        HIPCHK(hipMemcpyAsync(h->need_dev, h->h_need.data(), h->B, hipMemcpyHostToDevice, h->stream));
    }
    fl_launch_env_list(h->d, mask_dev ? mask_dev : h->need_dev, h->stream, mask_dev != nullptr);
    mask_dev = mask_dev ? mask_dev : h->need_dev;
    fl_launch_distance_maps(h->d, mask_dev, h->stream);
    HIPCHK(hipGetLastError());
    fl_launch_segments(h->d, mask_dev, h->stream);
    HIPCHK(hipGetLastError());
    fl_launch_nexthop(h->d, mask_dev, h->stream);
    fl_launch_hop8(h->d, mask_dev, h->stream);
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_distance_map_rebuild(fl_batch *h) {
    NEED_COMMIT(h);
    return rebuild_tables(h, nullptr);
}

int fl_distance_map_rebuild_masked(fl_batch *h, const uint8_t *mask_dev) {
    NEED_COMMIT(h);
    if (!mask_dev) { set_err("fl_distance_map_rebuild_masked: null mask"); return FL_ERR_ARG; }
    return rebuild_tables(h, mask_dev);
}

int fl_positions_map(fl_batch *h, int b, int32_t *out) {
    NEED_COMMIT(h);
    if (b < 0 || b >= h->B || !out) { set_err("fl_positions_map: bad argument"); return FL_ERR_ARG; }
    // RailEnv._update_agent_positions_map (rail_env.py:360-367): later agents overwrite earlier ones on a shared cell
    const int A = h->A;
    std::vector<int> pos(A);
    HIPCHK(hipMemcpyAsync(pos.data(), h->d.pos + (size_t)b * A, (size_t)A * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (size_t c = 0; c < (size_t)h->H * h->W; c++) out[c] = -1;
    for (int i = 0; i < A; i++)
        if (pos[i] >= 0) out[pos[i]] = i;
    return FL_OK;
}

int fl_obs_cutils(fl_batch *h, int max_nodes, int pred_depth, float *attr_dev, float *forest_dev, int32_t *adjacency_dev,
                  int32_t *node_order_dev, int32_t *edge_order_dev, uint8_t *valid_actions_dev, double *props_dev) {
    NEED_COMMIT(h);
    if (max_nodes < 4 || max_nodes > FL_OBS_MAX_NODES || pred_depth < 1 || pred_depth > FL_OBS_MAX_PRED) {
        set_err("fl_obs_cutils: max_nodes must be in [4,%d] and pred_depth in [1,%d]", FL_OBS_MAX_NODES, FL_OBS_MAX_PRED);
        return FL_ERR_ARG;
    }
    if (!attr_dev || !forest_dev || !adjacency_dev || !node_order_dev || !edge_order_dev || !valid_actions_dev) {
        set_err("fl_obs_cutils: null output buffer");
        return FL_ERR_ARG;
    }
    int rc = fl_launch_obs_cutils(h->obs, h->d, max_nodes, pred_depth, attr_dev, forest_dev, adjacency_dev, node_order_dev,
                                  edge_order_dev, valid_actions_dev, props_dev, h->stream);
    if (rc != FL_OK) { set_err("fl_obs_cutils: no launch configuration: %d rail cells and %d agents per env do not fit the observation kernels' LDS (160 KiB a workgroup), or the sizes are out of range", h->d.Rcap, h->d.A); return rc; }
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_obs_cutils_policy(fl_batch *h, int max_nodes, int pred_depth, float *attr_dev, float *forest_dev, int64_t *adjacency_dev,
                         int64_t *node_order_dev, int64_t *edge_order_dev, uint8_t *valid_actions_dev, double *props_dev) {
    NEED_COMMIT(h);
    if (max_nodes < 4 || max_nodes > FL_OBS_MAX_NODES || pred_depth < 1 || pred_depth > FL_OBS_MAX_PRED) {
        set_err("fl_obs_cutils_policy: max_nodes must be in [4,%d] and pred_depth in [1,%d]", FL_OBS_MAX_NODES, FL_OBS_MAX_PRED);
        return FL_ERR_ARG;
    }
    if (!attr_dev || !forest_dev || !adjacency_dev || !node_order_dev || !edge_order_dev || !valid_actions_dev) {
        set_err("fl_obs_cutils_policy: null output buffer");
        return FL_ERR_ARG;
    }
    int rc = fl_launch_obs_cutils(h->obs, h->d, max_nodes, pred_depth, attr_dev, forest_dev, reinterpret_cast<int32_t *>(adjacency_dev),
                                  reinterpret_cast<int32_t *>(node_order_dev), reinterpret_cast<int32_t *>(edge_order_dev), valid_actions_dev, props_dev,
                                  h->stream, nullptr, 1);
    if (rc != FL_OK) { set_err("fl_obs_cutils_policy: no launch configuration: %d rail cells and %d agents per env do not fit the observation kernels' LDS (160 KiB a workgroup), or the sizes are out of range", h->d.Rcap, h->d.A); return rc; }
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_obs_cutils_handles(fl_batch *h, int max_nodes, int pred_depth, const int32_t *handles, int n_handles, float *attr_dev,
                          float *forest_dev, int32_t *adjacency_dev, int32_t *node_order_dev, int32_t *edge_order_dev,
                          uint8_t *valid_actions_dev, double *props_dev) {
    NEED_COMMIT(h);
    const int A = h->A;
    if (!handles || n_handles < 1 || n_handles > A) { set_err("fl_obs_cutils_handles: 1 <= n_handles <= %d agents", A); return FL_ERR_ARG; }
    // the reference's conflict test erases position `agent.handle` from a list of len(handles) entries (tool.h:428-434): a listed
    // handle >= len(handles) is undefined behaviour there; what remains are the permutations of 0 .. n-1
    std::vector<int16_t> label(A, (int16_t)-1);
    bool identity = n_handles == A;
    for (int j = 0; j < n_handles; j++) {
        const int a = handles[j];
        if (a < 0 || a >= n_handles || label[a] >= 0) {
            set_err("fl_obs_cutils_handles: handles has to be a permutation of 0 .. %d (handle %d at position %d): the reference's get_many is undefined for any other strict subset "
                    "(treeobs.cpp:393-401 erases list position `handle`)", n_handles - 1, a, j);
            return FL_ERR_ARG;
        }
        label[a] = (int16_t)j;
        identity = identity && a == j;
    }
    if (identity)
        return fl_obs_cutils(h, max_nodes, pred_depth, attr_dev, forest_dev, adjacency_dev, node_order_dev, edge_order_dev, valid_actions_dev, props_dev);
    if (max_nodes < 4 || max_nodes > FL_OBS_MAX_NODES || pred_depth < 1 || pred_depth > FL_OBS_MAX_PRED) {
        set_err("fl_obs_cutils_handles: max_nodes must be in [4,%d] and pred_depth in [1,%d]", FL_OBS_MAX_NODES, FL_OBS_MAX_PRED);
        return FL_ERR_ARG;
    }
    if (!attr_dev || !forest_dev || !adjacency_dev || !node_order_dev || !edge_order_dev || !valid_actions_dev) {
        set_err("fl_obs_cutils_handles: null output buffer");
        return FL_ERR_ARG;
    }
    HIPCHK(hipMemcpyAsync(h->obs.label, label.data(), (size_t)A * sizeof(int16_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));      // (the staging vector is a local)
    int rc = fl_launch_obs_cutils(h->obs, h->d, max_nodes, pred_depth, attr_dev, forest_dev, adjacency_dev, node_order_dev,
                                  edge_order_dev, valid_actions_dev, props_dev, h->stream, h->obs.label);
    if (rc != FL_OK) { set_err("fl_obs_cutils_handles: no launch configuration"); return rc; }
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_obs_cutils_tree(fl_batch *h, int max_nodes, int pred_depth, float *attr_dev, float *forest_dev, int32_t *adjacency_dev,
                       int32_t *node_order_dev, int32_t *edge_order_dev, uint8_t *valid_actions_dev, double *props_dev,
                       int tree_max_depth, int tree_pred_depth, double *tree_out_dev) {
    NEED_COMMIT(h);
    if (max_nodes < 4 || max_nodes > FL_OBS_MAX_NODES || pred_depth < 1 || pred_depth > FL_OBS_MAX_PRED || tree_max_depth < 1 ||
        tree_max_depth > FL_MAX_TREE_DEPTH || tree_pred_depth < 0 || tree_pred_depth > pred_depth) {
        set_err("fl_obs_cutils_tree: max_nodes in [4,%d], pred_depth in [1,%d], tree depth in [1,%d], 0 <= tree_pred_depth <= pred_depth",
                FL_OBS_MAX_NODES, FL_OBS_MAX_PRED, FL_MAX_TREE_DEPTH);
        return FL_ERR_ARG;
    }
    if (!attr_dev || !forest_dev || !adjacency_dev || !node_order_dev || !edge_order_dev || !valid_actions_dev || !tree_out_dev) {
        set_err("fl_obs_cutils_tree: null output buffer");
        return FL_ERR_ARG;
    }
    if (tree_max_depth > 3 && (h->d.max_branch > 2 || getenv("FL_OBS_NO_COMPACT") != nullptr)) {   // (before the cutils launch sets its sticky deadlock bits)
        set_err("fl_obs_cutils_tree: max_depth 4 needs a grid on which no direction of a cell has more than two transitions (every Flatland rail cell type) and the compact node tables; this batch has %d", h->d.max_branch);
        return FL_ERR_ARG;
    }
    if (tree_max_depth > 3 || max_nodes > 32) {   // beyond the fused kernels' node tables: the two builders one after the other (same outputs)
        int rc2 = fl_obs_cutils(h, max_nodes, pred_depth, attr_dev, forest_dev, adjacency_dev, node_order_dev, edge_order_dev, valid_actions_dev, props_dev);
        return rc2 != FL_OK ? rc2 : fl_obs_tree(h, tree_max_depth, tree_pred_depth, tree_out_dev);
    }
    int rc = fl_launch_obs_both(h->obs, h->d, max_nodes, pred_depth, attr_dev, forest_dev, adjacency_dev, node_order_dev,
                                edge_order_dev, valid_actions_dev, props_dev, tree_max_depth, tree_pred_depth, tree_out_dev, h->stream);
    if (rc != FL_OK) { set_err("fl_obs_cutils_tree: no launch configuration: %d rail cells and %d agents per env do not fit the observation kernels' LDS (160 KiB a workgroup), or the sizes are out of range", h->d.Rcap, h->d.A); return rc; }
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_obs_set_mode(fl_batch *h, int flags) {
    NEED_COMMIT(h);
    if (flags & ~FL_OBS_KEEP_TREE_ROWS) { set_err("fl_obs_set_mode: unknown flag"); return FL_ERR_ARG; }
    h->obs.keep_rows = (flags & FL_OBS_KEEP_TREE_ROWS) != 0;
    h->obs.rows_out = nullptr;      // the next launch fills its whole slab and starts the row masks afresh
    return FL_OK;
}

int fl_obs_tree(fl_batch *h, int max_depth, int pred_depth, double *out_dev) {
    NEED_COMMIT(h);
    if (max_depth < 1 || max_depth > FL_MAX_TREE_DEPTH || pred_depth > FL_OBS_MAX_PRED || !out_dev) {
        set_err("fl_obs_tree: max_depth must be in [1,%d], pred_depth <= %d", FL_MAX_TREE_DEPTH, FL_OBS_MAX_PRED);
        return FL_ERR_ARG;
    }
    if (max_depth > 3 && h->d.max_branch > 2) {
        set_err("fl_obs_tree: max_depth 4 needs a grid on which no direction of a cell has more than two transitions (every Flatland rail cell type); this batch has %d", h->d.max_branch);
        return FL_ERR_ARG;
    }
    int rc = fl_launch_obs_tree(h->obs, h->d, max_depth, pred_depth, out_dev, h->stream);
    if (rc != FL_OK) { set_err("fl_obs_tree: no launch configuration: %d rail cells and %d agents per env do not fit the observation kernels' LDS (160 KiB a workgroup), or the sizes are out of range", h->d.Rcap, h->d.A); return rc; }
    HIPCHK(hipGetLastError());
    return FL_OK;
}

int fl_obs_tree_handles(fl_batch *h, int max_depth, int pred_depth, const int32_t *handles, int n_handles, double *out_dev) {
    NEED_COMMIT(h);
    const int A = h->A;
    if (!handles || n_handles < 1 || n_handles > A) { set_err("fl_obs_tree_handles: 1 <= n_handles <= %d agents", A); return FL_ERR_ARG; }
    // the reference's conflict test deletes position `handle` from the arrays of the listed handles' predictions and reads
    // env.agents[position].state (observations.py:337-366): a listed handle >= len(handles) is an IndexError there; what remains are
    // the permutations of 0 .. n-1
    std::vector<int16_t> label(A, (int16_t)-1);
    bool identity = n_handles == A;
    for (int j = 0; j < n_handles; j++) {
        const int a = handles[j];
        if (a < 0 || a >= n_handles || label[a] >= 0) {
            set_err("fl_obs_tree_handles: handles has to be a permutation of 0 .. %d (handle %d at position %d): the reference's get_many raises IndexError for a handle >= len(handles) "
                    "(observations.py:337 deletes list position `handle`)", n_handles - 1, a, j);
            return FL_ERR_ARG;
        }
        label[a] = (int16_t)j;
        identity = identity && a == j;
    }
    if (identity || pred_depth < 0) return fl_obs_tree(h, max_depth, pred_depth, out_dev);   // (no predictor: no conflict test, the list does not matter)
    if (max_depth < 1 || max_depth > FL_MAX_TREE_DEPTH || pred_depth > FL_OBS_MAX_PRED || !out_dev) {
        set_err("fl_obs_tree_handles: max_depth must be in [1,%d], pred_depth <= %d", FL_MAX_TREE_DEPTH, FL_OBS_MAX_PRED);
        return FL_ERR_ARG;
    }
    if (max_depth > 3 && h->d.max_branch > 2) {
        set_err("fl_obs_tree_handles: max_depth 4 needs a grid on which no direction of a cell has more than two transitions; this batch has %d", h->d.max_branch);
        return FL_ERR_ARG;
    }
    HIPCHK(hipMemcpyAsync(h->obs.label, label.data(), (size_t)A * sizeof(int16_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));      // (the staging vector is a local)
    int rc = fl_launch_obs_tree(h->obs, h->d, max_depth, pred_depth, out_dev, h->stream, h->obs.label);
    if (rc != FL_OK) { set_err("fl_obs_tree_handles: no launch configuration"); return rc; }
    HIPCHK(hipGetLastError());
    return FL_OK;
}

// diagnostic (not part of the public header): copy the per-env phase clocks of a -DFL_OBS_TIMING build
extern "C" int fl_debug_obs_clocks(fl_batch *h, long long *out /* [B][64] */) {
    NEED_COMMIT(h);
    HIPCHK(hipMemcpyAsync(out, h->obs.dbg, (size_t)h->B * 64 * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return FL_OK;
}

// diagnostic (not part of the public header): threads, LDS bytes, keys in LDS, next-hop in LDS, work-list bytes, time masks,
// second index, items in LDS, one pass B for both builders, compact upstream trees of the fused observation launch on this batch
extern "C" int fl_debug_obs_config(fl_batch *h, int pred_depth, int max_depth, int tree_pred, int *out11) {
    NEED_COMMIT(h);
    return fl_obs_config_of_fused(h->d, pred_depth, max_depth, tree_pred, out11, obs_batch_is_wide(h->d.B, h->obs.n_cu));
}

// diagnostic (not part of the public header), no GPU needed: does the launcher take a batch of B small envs on n_cu CUs as two workgroups a CU
extern "C" int fl_debug_batch_is_wide(int B, int n_cu) { return obs_batch_is_wide(B, n_cu) ? 1 : 0; }

// diagnostic (not part of the public header): what the last fused observation launch (fl_obs_cutils_tree / fl_step_obs with a tree)
// of this handle ran -- out[0] fixed launch class (0 = the runtime-carving kernel), out[1] 1 = the class's split kernel (the class's
// body only for the envs that fit it), out[2] envs that took the class's body
extern "C" int fl_debug_last_obs_class(fl_batch *h, int *out3) {
    if (!h || !out3) return FL_ERR_ARG;
    out3[0] = h->obs.last_fix; out3[1] = h->obs.last_split; out3[2] = h->obs.last_fit;
    return FL_OK;
}

// diagnostic (not part of the public header), no GPU needed: the same for a batch of the given sizes -- agents, rail-cell and
// unique-target capacities, maps taller than wide (compact prediction keys), most transitions of a (cell, direction)
extern "C" int fl_debug_obs_config_of(int A, int Rcap, int Ucap, int tall, int max_branch, int pred_depth, int max_depth, int tree_pred,
                                      int *out11) {
    FlDev d;
    memset(&d, 0, sizeof d);
    d.A = A; d.Rcap = Rcap; d.Ucap = Ucap; d.max_branch = max_branch;
    static uint16_t dummy_key;
    d.rkey = tall ? &dummy_key : nullptr;
    return fl_obs_config_of_fused(d, pred_depth, max_depth, tree_pred, out11);
}
// ... of a WIDE batch (several envs per CU)
extern "C" int fl_debug_obs_config_of_wide(int A, int Rcap, int Ucap, int tall, int max_branch, int pred_depth, int max_depth, int tree_pred,
                                           int *out11) {
    FlDev d;
    memset(&d, 0, sizeof d);
    d.A = A; d.Rcap = Rcap; d.Ucap = Ucap; d.max_branch = max_branch;
    static uint16_t dummy_key;
    d.rkey = tall ? &dummy_key : nullptr;
    return fl_obs_config_of_fused(d, pred_depth, max_depth, tree_pred, out11, 1);
}

double fl_algorithmic_bytes_per_agent_step(fl_batch *h, int with_cutils_obs, int tree_depth) {
    if (!h) return 0.0;
    // DESIGN.md "algorithmic bytes": compulsory HBM traffic per agent-step with this SoA.
    //   dynamics: 20 B state read + 20 B state write + 16 B static read (init_pos, target, earliest, spk; latest/speed/tslot
    //             only on the terminal step) + 1 B action + 5 B reward/done + 8 B RNG words + MT block write amortised
    //             (2496 B / 312 agent-steps = 8 B)
    //   per env amortised over A: rail grid 2*H*W read once per obs build
    //   cutils obs out: 31*12*4 + 30*3*4 + 31*4 + 30*4 + 83*4 + 5 = 2429 B; depth-d tree out: 12*8*N(d) (f64)
    double bytes = 20 + 20 + 16 + 1 + 5 + 8 + 8;
    if (with_cutils_obs || tree_depth > 0) bytes += 2.0 * h->H * h->W / h->A;
    if (with_cutils_obs) bytes += 2429.0 + 20 + 36 + 2 * 31;  // + state/static re-read by the obs kernel + 31 distance-map gathers
    if (tree_depth > 0) {
        int n = 1, p = 1;
        for (int k = 0; k < tree_depth; k++) { p *= 4; n += p; }
        bytes += 96.0 * n + 2.0 * n;
    }
    return bytes;
}
