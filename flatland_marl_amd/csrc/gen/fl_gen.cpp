// fl_gen.cpp -- reset-time generators on the host (C-ABI in include/flatland_gen.h): what RailEnv.reset(regenerate_rail=True,
// regenerate_schedule=True) computes before the first step, bit-exact on the same numpy RandomState (MT19937) stream.
//
// Replaces (paths relative to /root/reference/flatland-rl/flatland):
//   envs/rail_generators.py:196-854        SparseRailGen.generate and its helpers
//   envs/grid4_generators_utils.py:18-175  connect_rail_in_grid_map, connect_straight_line_in_grid_map, fix_inner_nodes, align_cell_to_city
//   core/grid/grid4_astar.py:40-150        a_star (insertion-ordered open set, first minimum of f)
//   core/transition_map.py:386-457,511-589 cell_neighbours_valid, fix_transitions (its own RandomState seeded with 12, :139-143)
//   core/grid/rail_env_grid.py:28-78       RailEnvTransitions.transition_list / is_valid; grid4.py:190-215 rotate_transition
//   envs/line_generators.py:18-165         speed_initialization_helper, SparseLineGen.generate / decide_orientation
//   envs/timetable_generators.py:21-96     timetable_generator (+ the shortest-path length of rail_env_shortest_paths.py:203-274)
//   envs/distance_map.py:57-160            the distance map the timetable needs (host BFS; the batch builds its own on the GPU)
// and numpy's legacy RandomState draws the reference makes on the way (randint, choice with and without p, permutation /
// shuffle, random_sample: numpy/random/_legacy + _bounded_integers, frozen by NEP 19).
//
// Plain C++17, no GPU: the generators run once per reset and are serial by nature (A* on a shared grid, one RNG stream).
#include "../../../include/flatland_gen.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace {

thread_local char g_err[512] = "";
void set_err(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------------------------- numpy legacy RandomState
struct Rng {
    uint32_t key[624];
    int pos;
    void twist() {
        const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;
        int i;
        for (i = 0; i < 624 - 397; i++) { const uint32_t y = (key[i] & UP) | (key[i + 1] & LO); key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u); }
        for (; i < 623; i++) { const uint32_t y = (key[i] & UP) | (key[i + 1] & LO); key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u); }
        const uint32_t y = (key[623] & UP) | (key[0] & LO);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        pos = 0;
    }
    uint32_t next32() {
        if (pos == 624) twist();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double random_sample() {  // rk_double
        const uint32_t a = next32() >> 5, b = next32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    // RandomState.randint(low, high): masked rejection on 32-bit words; no draw when the range is a single value
    long long randint(long long low, long long high) {
        const unsigned long long rng = (unsigned long long)(high - 1 - low);
        if (rng == 0) return low;
        unsigned long long mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        if (rng <= 0xFFFFFFFFull) {
            if (rng == 0xFFFFFFFFull) return low + (long long)next32();
            uint32_t v;
            do v = next32() & (uint32_t)mask; while (v > rng);
            return low + (long long)v;
        }
        unsigned long long v;
        do v = (((unsigned long long)next32() << 32) | next32()) & mask; while (v > rng);
        return low + (long long)v;
    }
    // random_interval (legacy shuffle): uniform in [0, max]
    unsigned long long interval(unsigned long long max) {
        if (max == 0) return 0;
        unsigned long long mask = max;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        unsigned long long v;
        if (max <= 0xffffffffull) { do v = next32() & mask; while (v > max); }
        else { do v = (((unsigned long long)next32() << 32) | next32()) & mask; while (v > max); }
        return v;
    }
    // RandomState.permutation(n): shuffle(arange(n))
    std::vector<int> permutation(int n) {
        std::vector<int> a(n);
        for (int i = 0; i < n; i++) a[i] = i;
        for (int i = n - 1; i >= 1; i--) {
            const int j = (int)interval((unsigned long long)i);
            std::swap(a[i], a[j]);
        }
        return a;
    }
    // seed(int): init_genrand
    void seed_int(uint32_t s) {
        for (int i = 0; i < 624; i++) {
            key[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
    }
};

// ---------------------------------------------------------------------------------------------- transitions
typedef std::pair<int, int> Cell;  // (row, col)
const int DR[4] = {-1, 0, 1, 0}, DC[4] = {0, 1, 0, -1};
inline int mirror(int d) { return (d + 2) & 3; }
inline uint16_t set_transition(uint16_t cell, int orientation, int direction) {
    return (uint16_t)(cell | (1u << ((3 - orientation) * 4 + (3 - direction))));
}
inline int get_transition(uint16_t cell, int orientation, int direction) {
    return (cell >> ((3 - orientation) * 4 + (3 - direction))) & 1;
}
// Grid4Transitions.rotate_transition (grid4.py:190-215), rotation in quarter turns
uint16_t rotate_transition(uint16_t cell, int rot) {
    rot &= 3;
    if (rot == 0) return cell;
    uint16_t v = 0;
    for (int i = 0; i < 4; i++) {
        uint16_t nib = (cell >> ((3 - i) * 4)) & 15u;
        nib = (uint16_t)(((nib >> rot) | (nib << (4 - rot))) & 15u);  // block_tuple[(4 - rot):] + block_tuple[:(4 - rot)]
        v |= (uint16_t)(nib << ((3 - i) * 4));
    }
    return (uint16_t)(((v & ((1u << (rot * 4)) - 1u)) << ((4 - rot) * 4)) | (v >> (rot * 4)));
}
const uint16_t TRANSITION_LIST[11] = {0x0000, 0x8020, 0x9220, 0x8421, 0x9621, 0xCC33, 0x5202, 0x2000, 0x4002, 0x1200, 0xC022};
struct ValidSet {
    bool ok[65536];
    ValidSet() {
        memset(ok, 0, sizeof ok);
        for (int idx = 0; idx < 11; idx++) {
            uint16_t t = TRANSITION_LIST[idx];
            ok[t] = true;
            const int rots = (idx == 2 || idx == 4 || idx == 6 || idx == 7 || idx == 8 || idx == 9 || idx == 10) ? 3 : (idx == 1 || idx == 5) ? 1 : 0;
            for (int k = 0; k < rots; k++) { t = rotate_transition(t, 1); ok[t] = true; }
        }
    }
};
const ValidSet &valid_set() { static const ValidSet v; return v; }

struct Grid {
    int H, W;
    std::vector<uint16_t> g;
    uint16_t &at(int r, int c) { return g[(size_t)r * W + c]; }
    uint16_t at(int r, int c) const { return g[(size_t)r * W + c]; }
    bool inside(int r, int c) const { return r >= 0 && r < H && c >= 0 && c < W; }
};

// grid4_utils.direction_to_point (:37-52): np.argmax of the squared differences (first maximum), then the sign
int direction_to_point(Cell p1, Cell p2) {
    const long long d0 = p1.first - p2.first, d1 = p1.second - p2.second;
    const int axis = (d1 * d1 > d0 * d0) ? 1 : 0;
    const long long dv = axis == 0 ? d0 : d1;
    if (axis == 0) return dv > 0 ? 0 : 2;
    return dv > 0 ? 3 : 1;
}
int get_direction(Cell a, Cell b) {  // adjacent cells
    if (b.first < a.first) return 0;
    if (b.first > a.first) return 2;
    if (b.second > a.second) return 1;
    return 3;
}
inline int manhattan(Cell a, Cell b) { return abs(a.first - b.first) + abs(a.second - b.second); }

// ---------------------------------------------------------------------------------------------- A*
// grid4_astar.a_star with respect_transition_validity=False: the open set is insertion-ordered and the node taken is the
// FIRST one with the lowest f; a child already open is never updated; children are generated W, E, N, S.
std::vector<Cell> a_star(const Grid &G, Cell start, Cell end, bool avoid_rails, const std::vector<uint8_t> &forbidden) {
    const int H = G.H, W = G.W;
    struct Node { int parent; double g, h, f; };
    std::vector<Node> node((size_t)H * W, Node{-1, 0.0, 0.0, 0.0});
    std::vector<uint8_t> state((size_t)H * W, 0);  // 1 = open, 2 = closed
    std::vector<int> open;                         // insertion order (removal keeps the order of the rest)
    const int s_id = start.first * W + start.second, e_id = end.first * W + end.second;
    open.push_back(s_id);
    state[s_id] = 1;
    static const int NR[4] = {0, 0, -1, 1}, NC[4] = {-1, 1, 0, 0};
    while (!open.empty()) {
        size_t best = 0;
        for (size_t k = 1; k < open.size(); k++)
            if (node[open[k]].f < node[open[best]].f) best = k;
        const int cur = open[best];
        open.erase(open.begin() + (long)best);
        state[cur] = 2;
        if (cur == e_id) {
            std::vector<Cell> path;
            for (int c = cur; c >= 0; c = node[c].parent) path.push_back(Cell(c / W, c % W));
            std::reverse(path.begin(), path.end());
            return path;
        }
        const int cr = cur / W, cc = cur % W;
        for (int k = 0; k < 4; k++) {
            const int r = cr + NR[k], c = cc + NC[k];
            if (r >= H || r < 0 || c >= W || c < 0) continue;
            const int id = r * W + c;
            if (forbidden[id] && id != s_id && id != e_id) continue;
            if (state[id] == 2) continue;
            // (the reference fills g / h / f of the fresh child object before it finds the child in the open set and drops it)
            if (state[id] == 1) continue;
            Node &ch = node[id];
            ch.parent = cur;
            ch.g = node[cur].g + 1.0;
            ch.h = (double)manhattan(Cell(r, c), end) + (avoid_rails ? (G.at(r, c) > 0 ? 1.0 : 0.0) : 0.0);
            ch.f = ch.g + ch.h;
            open.push_back(id);
            state[id] = 1;
        }
    }
    return std::vector<Cell>();
}

// grid4_generators_utils.connect_rail_in_grid_map with flip_start/end False, respect_transition_validity False, avoid_rail True
std::vector<Cell> connect_rail(Grid &G, Cell start, Cell end, const std::vector<uint8_t> &forbidden) {
    std::vector<Cell> path = a_star(G, start, end, true, forbidden);
    if (path.size() < 2) return std::vector<Cell>();
    int current_dir = get_direction(path[0], path[1]);
    const Cell end_pos = path.back();
    for (size_t index = 0; index + 1 < path.size(); index++) {
        const Cell cur = path[index], nxt = path[index + 1];
        const int new_dir = get_direction(cur, nxt);
        uint16_t t = G.at(cur.first, cur.second);
        if (index == 0) {
            if (t == 0) t = 0;                                      // end point, no flip
            else t = set_transition(t, current_dir, new_dir);       // into existing rail
        } else {
            t = set_transition(t, current_dir, new_dir);
            t = set_transition(t, mirror(new_dir), mirror(current_dir));
        }
        G.at(cur.first, cur.second) = t;
        if (nxt == end_pos) {
            uint16_t te = G.at(end_pos.first, end_pos.second);
            if (te == 0) te = 0;
            else te = set_transition(te, new_dir, new_dir);
            G.at(end_pos.first, end_pos.second) = te;
        }
        current_dir = new_dir;
    }
    return path;
}

// connect_straight_line_in_grid_map: the path runs in ascending row / column order whatever the direction
std::vector<Cell> connect_straight(Grid &G, Cell start, Cell end) {
    std::vector<Cell> path;
    if (!(start.first == end.first || start.second == end.second)) return path;
    const int direction = direction_to_point(start, end);
    if (direction == 0 || direction == 2) {
        for (int r = std::min(start.first, end.first); r <= std::max(start.first, end.first); r++) path.push_back(Cell(r, start.second));
    } else {
        for (int c = std::min(start.second, end.second); c <= std::max(start.second, end.second); c++) path.push_back(Cell(start.first, c));
    }
    for (const Cell &cell : path) {
        uint16_t t = G.at(cell.first, cell.second);
        t = set_transition(t, direction, direction);
        t = set_transition(t, mirror(direction), mirror(direction));
        G.at(cell.first, cell.second) = t;
    }
    return path;
}

void fix_inner_nodes(Grid &G, Cell pos) {
    int corner[4], n = 0;
    for (int d = 0; d < 4; d++) {
        const int r = pos.first + DR[d], c = pos.second + DC[d];
        if (G.inside(r, c) && G.at(r, c) > 0) corner[n++] = d;
    }
    if (n != 2) return;
    uint16_t t = 0;
    t = set_transition(t, mirror(corner[0]), corner[1]);
    t = set_transition(t, mirror(corner[1]), corner[0]);
    G.at(pos.first, pos.second) = t;
    for (int k = 0; k < 2; k++) {
        const int r = pos.first + DR[corner[k]], c = pos.second + DC[corner[k]];
        G.at(r, c) = set_transition(G.at(r, c), corner[k], mirror(corner[k]));
    }
}

// GridTransitionMap.cell_neighbours_valid(rcPos, check_this_cell=True)
bool cell_neighbours_valid(const Grid &G, Cell p) {
    const uint16_t t = G.at(p.first, p.second);
    if (!valid_set().ok[t]) return false;
    for (int d = 0; d < 4; d++) {
        bool out = false;
        for (int o = 0; o < 4; o++) out = out || get_transition(t, o, d);
        if (!out) continue;
        const int r = p.first + DR[d], c = p.second + DC[d];
        if (!G.inside(r, c)) return false;
        const uint16_t n = G.at(r, c);
        if (((n >> ((3 - d) * 4)) & 15u) == 0) return false;  // nothing leads on for an agent entering it facing d
    }
    if (t < 1) {  // an empty cell with incoming connections is invalid
        int connected = 0;
        for (int d = 0; d < 4; d++) {
            const int r = p.first + DR[d], c = p.second + DC[d];
            if (!G.inside(r, c)) continue;
            for (int o = 0; o < 4; o++) connected += get_transition(G.at(r, c), o, mirror(d));
        }
        if (connected > 0) return false;
    }
    return true;
}

// GridTransitionMap.fix_transitions(rcPos, direction); map_rng = the map's own RandomState(12)
void fix_transitions(Grid &G, Cell p, int direction, Rng &map_rng) {
    const uint16_t simple_switch_east_south = rotate_transition(TRANSITION_LIST[10], 1);
    const uint16_t simple_switch_west_south = rotate_transition(TRANSITION_LIST[2], 3);
    const uint16_t double_slip = TRANSITION_LIST[5];
    const uint16_t three_way[2] = {simple_switch_east_south, simple_switch_west_south};
    int incoming[4] = {0, 0, 0, 0}, n_in = 0;
    for (int d = 0; d < 4; d++) {
        const int r = p.first + DR[d], c = p.second + DC[d];
        if (!G.inside(r, c)) continue;
        int connected = 0;
        for (int o = 0; o < 4; o++) connected += get_transition(G.at(r, c), o, mirror(d));
        if (connected > 0) { incoming[d] = 1; n_in++; }
    }
    uint16_t &cell = G.at(p.first, p.second);
    if (n_in == 1) {  // one incoming direction: dead end (an empty cell stays empty)
        const bool was_empty = cell == 0;
        cell = 0;
        if (!was_empty)
            for (int d = 0; d < 4; d++)
                if (incoming[d]) cell = set_transition(cell, mirror(d), d);
    }
    if (n_in == 2) {
        cell = 0;
        int cd[2], k = 0;
        for (int d = 0; d < 4; d++)
            if (incoming[d]) cd[k++] = d;
        cell = set_transition(cell, mirror(cd[0]), cd[1]);
        cell = set_transition(cell, mirror(cd[1]), cd[0]);
    }
    if (n_in == 3) {
        cell = 0;
        int hole = 0;
        while (incoming[hole]) hole++;
        uint16_t t;
        const int idx = direction >= 0 ? (direction - hole + 3) % 4 : -1;
        if (idx == 0) t = simple_switch_west_south;
        else if (idx == 2) t = simple_switch_east_south;
        else t = three_way[map_rng.randint(0, 2)];  // random_generator.choice(three_way_transitions, 1)
        cell = rotate_transition(t, hole);
    }
    if (n_in == 4) {
        const int rotation = (int)map_rng.randint(0, 2);
        cell = rotate_transition(double_slip, rotation);
    }
}

// ---------------------------------------------------------------------------------------------- sparse rail generator
struct CityPlan {
    int radius, rail_pairs_in_city, rails_between_cities;
};
CityPlan city_plan(int max_rails_between_cities, int max_rail_pairs_in_city) {
    CityPlan p;
    p.rail_pairs_in_city = max_rail_pairs_in_city < 1 ? 1 : max_rail_pairs_in_city;
    p.rails_between_cities = max_rails_between_cities > p.rail_pairs_in_city * 2 ? p.rail_pairs_in_city * 2 : max_rails_between_cities;
    p.radius = (int)ceil((p.rail_pairs_in_city * 2) / 2.0) + 2;
    return p;
}

std::vector<Cell> evenly_distributed_cities(int num_cities, int radius, int W, int H) {
    const double aspect_ratio = (double)H / (double)W;
    const int city_size = 2 * (radius + 1);
    const int max_per_row = (H - 2) / city_size, max_per_col = (W - 2) / city_size;
    const int per_row = std::min((int)ceil(sqrt(num_cities * aspect_ratio)), max_per_row);
    const int per_col = std::min((int)ceil((double)num_cities / per_row), max_per_col);
    const int n_build = std::min(num_cities, per_col * per_row);
    auto linspace_int = [](double start, double stop, int num) {  // np.linspace(start, stop, num, dtype=int)
        std::vector<int> out(num);
        if (num == 1) { out[0] = (int)start; return out; }
        const double step = (stop - start) / (num - 1);
        for (int i = 0; i < num; i++) out[i] = (int)(i * step + start);
        out[num - 1] = (int)stop;
        return out;
    };
    const std::vector<int> rows = linspace_int(radius + 2, H - (radius + 2), per_row), cols = linspace_int(radius + 2, W - (radius + 2), per_col);
    std::vector<Cell> out;
    for (int i = 0; i < n_build; i++) out.push_back(Cell(rows[i % per_row], cols[i / per_row]));
    return out;
}

std::vector<Cell> random_cities(int num_cities, int radius, int W, int H, Rng &rng) {
    std::vector<Cell> out;
    std::vector<uint8_t> allowed((size_t)H * W, 0);
    const int pad = radius + 1;
    for (int r = pad; r < H - pad; r++)
        for (int c = pad; c < W - pad; c++) allowed[(size_t)r * W + c] = 1;
    std::vector<int> idx;
    for (int k = 0; k < num_cities; k++) {
        idx.clear();
        for (int i = 0; i < H * W; i++)
            if (allowed[i]) idx.push_back(i);
        if (idx.empty()) break;
        const int pick = idx[(size_t)rng.randint(0, (long long)idx.size())];
        const int row = pick / W, col = pick % W;
        for (int r = std::max(0, row - 2 * pad); r < std::min(H, row + 2 * pad + 1); r++)
            for (int c = std::max(0, col - 2 * pad); c < std::min(W, col + 2 * pad + 1); c++) allowed[(size_t)r * W + c] = 0;
        out.push_back(Cell(row, col));
    }
    return out;
}

struct RailOut {
    Grid G;
    std::vector<int> orientations;
    std::vector<std::vector<std::pair<Cell, int>>> stations;
};

// everything of SparseRailGen.generate after the city positions are known.  order[i] = indices of all cities sorted by
// their distance from city i (the reference's np.argsort, unstable: the caller supplies numpy's order; nullptr = stable)
int build_rail(int W, int H, const CityPlan &plan, bool grid_mode, const std::vector<Cell> &cities, const int32_t *order, Rng &rng, RailOut &out) {
    const int n = (int)cities.size(), radius = plan.radius;
    out.G.H = H; out.G.W = W; out.G.g.assign((size_t)H * W, 0);
    Grid &G = out.G;
    std::vector<double> vector_field((size_t)H * W, -1.0);
    std::vector<std::vector<std::vector<Cell>>> inner(n, std::vector<std::vector<Cell>>(4)), outer(n, std::vector<std::vector<Cell>>(4));
    std::vector<Cell> city_cells;
    // ---- _generate_city_connection_points
    for (int ci = 0; ci < n; ci++) {
        const Cell cp = cities[ci];
        std::vector<int> by_dist(n);
        for (int k = 0; k < n; k++) by_dist[k] = k;
        std::stable_sort(by_dist.begin(), by_dist.end(), [&](int a, int b) { return manhattan(cp, cities[a]) < manhattan(cp, cities[b]); });
        int dir;
        if (grid_mode) dir = (int)rng.randint(0, 4);
        else dir = direction_to_point(cp, cities[by_dist[1]]);
        out.orientations.push_back(dir);
        for (int r = cp.first - radius; r <= cp.first + radius; r++)       // _get_cells_in_city
            for (int c = cp.second - radius; c <= cp.second + radius; c++) {
                city_cells.push_back(Cell(r, c));
                if (r < 0 || r >= H || c < 0 || c >= W) { set_err("city at (%d,%d) does not fit the map", cp.first, cp.second); return FLG_ERR_ARG; }
                const int clip0 = std::min(std::max(r - cp.first, 0), 1), clip1 = std::min(std::max(cp.second - c, 0), 1);
                vector_field[(size_t)r * W + c] = (dir % 2 == 0) ? 2 * clip0 : 2 * clip1 + 1;  // align_cell_to_city
            }
        int per_dir[4] = {0, 0, 0, 0};
        const int nr = (int)rng.randint(1, plan.rail_pairs_in_city + 1) * 2;
        per_dir[dir] = nr; per_dir[(dir + 2) % 4] = nr;
        const int n_out = (int)rng.randint(1, std::min(plan.rails_between_cities, nr) + 1);
        const int start_idx = (nr - n_out) / 2;
        for (int d = 0; d < 4; d++) {
            for (int k = 0; k < per_dir[d]; k++) {
                const int slot = k - start_idx, off = k - nr / 2;
                const int inner_off = abs(off) + std::min(std::max(off, 0), 1) + 1;
                Cell in, ou;
                if (d == 0) { in = Cell(cp.first - radius + inner_off, cp.second + slot); ou = Cell(cp.first - radius, cp.second + slot); }
                else if (d == 1) { in = Cell(cp.first + slot, cp.second + radius - inner_off); ou = Cell(cp.first + slot, cp.second + radius); }
                else if (d == 2) { in = Cell(cp.first + radius - inner_off, cp.second + slot); ou = Cell(cp.first + radius, cp.second + slot); }
                else { in = Cell(cp.first + slot, cp.second - radius + inner_off); ou = Cell(cp.first + slot, cp.second - radius); }
                inner[ci][d].push_back(in);
                if (k >= start_idx && k < start_idx + n_out) outer[ci][d].push_back(ou);
            }
        }
    }
    std::vector<uint8_t> forbidden((size_t)H * W, 0);
    for (const Cell &c : city_cells) forbidden[(size_t)c.first * W + c.second] = 1;
    // ---- _connect_cities
    std::vector<Cell> inter_city;
    for (int ci = 0; ci < n; ci++) {
        // _closest_neighbour_in_grid4_directions
        int closest[4] = {-1, -1, -1, -1};
        std::vector<int> sorted(n);
        if (order) for (int k = 0; k < n; k++) sorted[k] = order[(size_t)ci * n + k];
        else {
            for (int k = 0; k < n; k++) sorted[k] = k;
            std::stable_sort(sorted.begin(), sorted.end(), [&](int a, int b) { return manhattan(cities[ci], cities[a]) < manhattan(cities[ci], cities[b]); });
        }
        for (int k = 1; k < n; k++) {
            const int nb = sorted[k];
            if (nb < 0 || nb >= n) { set_err("neighbour_order holds an invalid city index"); return FLG_ERR_ARG; }
            const int d = direction_to_point(cities[ci], cities[nb]);
            if (closest[d] < 0) closest[d] = nb;
            if (closest[0] >= 0 && closest[1] >= 0 && closest[2] >= 0 && closest[3] >= 0) break;
        }
        for (int od = 0; od < 4; od++) {
            int nb = closest[od];                                    // get_closest_neighbour_for_direction
            if (nb < 0) nb = closest[(od + 3) % 4];
            if (nb < 0) nb = closest[(od + 1) % 4];
            if (nb < 0) nb = closest[(od + 2) % 4];
            for (const Cell &op : outer[ci][od]) {
                if (nb < 0) { set_err("city %d has no neighbour to connect to", ci); return FLG_ERR_ARG; }
                int best = 0x7fffffff;
                Cell target(-1, -1);
                for (int d = 0; d < 4; d++)
                    for (const Cell &ip : outer[nb][d]) {
                        const int dist = manhattan(op, ip);
                        if (dist < best) { best = dist; target = ip; }
                    }
                if (target.first < 0) { set_err("city %d has no connection point", nb); return FLG_ERR_ARG; }
                const std::vector<Cell> line = connect_rail(G, op, target, forbidden);
                inter_city.insert(inter_city.end(), line.begin(), line.end());
            }
        }
    }
    // ---- _build_inner_cities
    std::vector<std::vector<std::vector<Cell>>> free_rails(n);
    for (int ci = 0; ci < n; ci++) {
        int boarder = 0;
        while (boarder < 4 && inner[ci][boarder].empty()) boarder++;
        const int opp = (boarder + 2) % 4;
        const int nr = (int)inner[ci][boarder].size(), n_out = (int)outer[ci][boarder].size();
        const int start_idx = (nr - n_out) / 2;
        for (int t = 0; t < nr; t++) free_rails[ci].push_back(connect_straight(G, inner[ci][boarder][t], inner[ci][opp][t]));
        for (int t = 0; t < nr; t++) {
            const Cell source = inner[ci][boarder][t], target = inner[ci][opp][t];
            fix_inner_nodes(G, source);
            fix_inner_nodes(G, target);
            if (t >= start_idx && t < start_idx + n_out) {
                connect_straight(G, source, outer[ci][boarder][t - start_idx]);
                connect_straight(G, target, outer[ci][opp][t - start_idx]);
            }
        }
    }
    // ---- _set_trainstation_positions
    out.stations.assign(n, std::vector<std::pair<Cell, int>>());
    for (int ci = 0; ci < n; ci++)
        for (int t = 0; t < (int)free_rails[ci].size(); t++) {
            const std::vector<Cell> &track = free_rails[ci][t];
            if (track.empty()) { set_err("empty city track"); return FLG_ERR_ARG; }
            out.stations[ci].push_back(std::make_pair(track[track.size() / 2], t));
        }
    // ---- _fix_transitions: validity of every city / inter-city cell on the unfixed grid first, then the fixes in that order
    Rng map_rng;
    map_rng.seed_int(12);  // GridTransitionMap.random_generator (transition_map.py:139-143)
    std::vector<std::pair<Cell, int>> to_fix;
    auto consider = [&](const Cell &c) {
        if (!cell_neighbours_valid(G, c)) to_fix.push_back(std::make_pair(c, (int)vector_field[(size_t)c.first * W + c.second]));
    };
    for (const Cell &c : city_cells) consider(c);
    for (const Cell &c : inter_city) consider(c);
    for (const auto &f : to_fix) fix_transitions(G, f.first, f.second, map_rng);
    return FLG_OK;
}

// ---------------------------------------------------------------------------------------------- line generator
// GridTransitionMap.check_path_exists: can `end` be reached from (start, direction)?
bool check_path_exists(const Grid &G, Cell start, int direction, Cell end) {
    std::vector<uint8_t> visited((size_t)G.H * G.W * 4, 0);
    std::vector<int> stack;
    stack.push_back((start.first * G.W + start.second) * 4 + direction);
    while (!stack.empty()) {
        const int node = stack.back();
        stack.pop_back();
        const int cell = node >> 2, d = node & 3, r = cell / G.W, c = cell % G.W;
        if (r == end.first && c == end.second) return true;
        if (visited[node]) continue;
        visited[node] = 1;
        const uint32_t bits = (G.at(r, c) >> ((3 - d) * 4)) & 15u;
        for (int m = 0; m < 4; m++)
            if ((bits >> (3 - m)) & 1) {
                const int nr = r + DR[m], nc = c + DC[m];
                if (G.inside(nr, nc)) stack.push_back((nr * G.W + nc) * 4 + m);
            }
    }
    return false;
}

// ---------------------------------------------------------------------------------------------- distance map (host)
// DistanceMap._compute for one target: u16 distances per (cell, orientation), 0xFFFF = unreachable
void distance_map_bfs(const Grid &G, Cell target, std::vector<uint16_t> &out) {
    const int H = G.H, W = G.W;
    out.assign((size_t)H * W * 4, 0xFFFF);
    std::vector<int> cur, nxt;
    for (int o = 0; o < 4; o++) out[((size_t)target.first * W + target.second) * 4 + o] = 0;
    auto visit = [&](int r, int c, int a, int dist) {
        uint16_t &v = out[((size_t)r * W + c) * 4 + a];
        if (v == 0xFFFF) { v = (uint16_t)dist; nxt.push_back((r * W + c) * 4 + a); }
    };
    for (int nd = 0; nd < 4; nd++) {
        const int r = target.first + DR[nd], c = target.second + DC[nd];
        if (!G.inside(r, c)) continue;
        for (int a = 0; a < 4; a++)
            if (get_transition(G.at(r, c), a, mirror(nd))) visit(r, c, a, 1);
    }
    int dist = 1;
    while (!nxt.empty()) {
        cur.swap(nxt);
        nxt.clear();
        for (int s : cur) {
            const int cell = s >> 2, o = s & 3, back = mirror(o);
            const int r = cell / W + DR[back], c = cell % W + DC[back];
            if (!G.inside(r, c)) continue;
            const uint16_t g = G.at(r, c);
            if (!g) continue;
            for (int a = 0; a < 4; a++)
                if (get_transition(g, a, o)) visit(r, c, a, dist + 1);
        }
        dist++;
    }
}

// numpy's pairwise summation of a contiguous float64 array (umath loops: PW_BLOCKSIZE 128, 8 accumulators)
double pairwise_sum(const double *a, size_t n) {
    if (n < 8) {
        double res = 0.;
        for (size_t i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; k++) r[k] = a[k];
        size_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

// timetable_generator (timetable_generators.py:21-96) on a finished rail and line; n_cities = len(agents_hints['city_positions']),
// or 2 when the caller has no hints (a rail loaded from a file: rail_generators.py:131-143 returns none)
int make_timetable(const Grid &G, int n_agents, int n_cities, const int32_t *init_pos, const int32_t *init_dir, const int32_t *target,
                   const double *speed, Rng &rng, int32_t *earliest, int32_t *latest, int32_t *max_episode_steps) {
    const int width = G.W, height = G.H;
    std::vector<double> times(n_agents);
    {
        std::vector<Cell> targets;
        std::vector<std::vector<uint16_t>> dms;
        for (int a = 0; a < n_agents; a++) {
            const Cell t(target[2 * a], target[2 * a + 1]);
            size_t u = 0;
            for (; u < targets.size(); u++)
                if (targets[u] == t) break;
            if (u == targets.size()) { targets.push_back(t); dms.push_back(std::vector<uint16_t>()); distance_map_bfs(G, t, dms.back()); }
            const uint16_t dv = dms[u][((size_t)init_pos[2 * a] * width + init_pos[2 * a + 1]) * 4 + init_dir[a]];
            const int len = dv == 0xFFFF ? 0 : (int)dv + 1;  // len(shortest path), None -> 0
            times[a] = (double)len / speed[a];
        }
    }
    int T = (int)(4 * 2 * ((double)(width + height) + ((double)n_agents / (double)n_cities)));
    const double mean_time = pairwise_sum(times.data(), times.size()) / (double)n_agents;
    double longest = times[0];
    for (int a = 1; a < n_agents; a++) longest = times[a] > longest ? times[a] : longest;
    const double mean_path_delay = mean_time * 0.2;
    const int T_new = (int)(ceil(longest * 1.5) + mean_path_delay);
    const int T_old = (int)(T * 3.0);
    T = std::min(T_new, T_old);
    const int end_buffer = (int)(T * 0.05);
    const int latest_arrival_max = T - end_buffer;
    for (int a = 0; a < n_agents; a++) {
        const int travel_max = (int)(ceil((times[a] * 1.3) + mean_path_delay));
        const int window = std::max(latest_arrival_max - travel_max, 1);
        const int e = (int)rng.randint(0, window);
        earliest[a] = e;
        latest[a] = e + travel_max;
    }
    *max_episode_steps = T;
    return FLG_OK;
}

// SparseLineGen.generate + timetable_generator on a finished rail (what reset(regenerate_schedule=True) redoes)
int make_schedule(const Grid &G, int n_agents, int n_cities, const std::vector<int> &orientations,
                  const std::vector<std::vector<std::pair<Cell, int>>> &stations, int n_speeds, const double *speed_values,
                  const double *speed_probs, Rng &rng, int32_t *init_pos, int32_t *init_dir, int32_t *target, double *speed,
                  int32_t *earliest, int32_t *latest, int32_t *max_episode_steps) {
        struct { const std::vector<int> &orientations; const std::vector<std::vector<std::pair<Cell, int>>> &stations; } R = {orientations, stations};
    // ---- SparseLineGen.generate (line_generators.py:82-165)
    int city1 = 0, city2 = 0;
    for (int a = 0; a < n_agents; a++) {
        Cell start, goal;
        int possible[2];
        if (a % 2 == 0) {
            const std::vector<int> perm = rng.permutation(n_cities);  // choice(len(city_positions), 2, replace=False)
            city1 = perm[0]; city2 = perm[1];
            const int n1 = (int)R.stations[city1].size(), n2 = (int)R.stations[city2].size();
            if (n1 == 0 || n2 == 0) { set_err("city without train stations"); return FLG_ERR_ARG; }
            const int si = (int)((2 * rng.randint(0, 10)) % n1), ti = (int)((2 * rng.randint(0, 10) + 1) % n2);
            start = R.stations[city1][si].first; goal = R.stations[city2][ti].first;
            possible[0] = R.orientations[city1]; possible[1] = (R.orientations[city1] + 2) % 4;
        } else {
            const int n1 = (int)R.stations[city1].size(), n2 = (int)R.stations[city2].size();
            const int si = (int)((2 * rng.randint(0, 10)) % n2), ti = (int)((2 * rng.randint(0, 10) + 1) % n1);
            start = R.stations[city2][si].first; goal = R.stations[city1][ti].first;
            possible[0] = R.orientations[city2]; possible[1] = (R.orientations[city2] + 2) % 4;
        }
        int feasible[2], nf = 0;  // decide_orientation
        for (int k = 0; k < 2; k++)
            if (check_path_exists(G, start, possible[k], goal)) feasible[nf++] = possible[k];
        const int orientation = nf > 0 ? feasible[rng.randint(0, nf)] : 0;
        init_pos[2 * a] = start.first; init_pos[2 * a + 1] = start.second;
        target[2 * a] = goal.first; target[2 * a + 1] = goal.second;
        init_dir[a] = orientation;
    }
    // speed_initialization_helper: np_random.choice(nb_classes, nb_agents, p=speed_ratios)
    if (n_speeds > 0) {
        std::vector<double> cdf(n_speeds);
        double run = 0.0;
        for (int k = 0; k < n_speeds; k++) { run += speed_probs[k]; cdf[k] = run; }
        const double total = run;
        for (int k = 0; k < n_speeds; k++) cdf[k] /= total;  // cdf /= cdf[-1]
        std::vector<double> u(n_agents);
        for (int a = 0; a < n_agents; a++) u[a] = rng.random_sample();
        for (int a = 0; a < n_agents; a++) {
            const int idx = (int)(std::upper_bound(cdf.begin(), cdf.end(), u[a]) - cdf.begin());  // searchsorted(side='right')
            speed[a] = speed_values[std::min(idx, n_speeds - 1)];
        }
    } else {
        for (int a = 0; a < n_agents; a++) speed[a] = 1.0;
    }
    return make_timetable(G, n_agents, n_cities, init_pos, init_dir, target, speed, rng, earliest, latest, max_episode_steps);
}

Rng load_rng(const uint32_t *key, int pos) {
    Rng r;
    memcpy(r.key, key, sizeof r.key);
    r.pos = pos;
    return r;
}
void store_rng(const Rng &r, uint32_t *key, int *pos) {
    memcpy(key, r.key, sizeof r.key);
    *pos = r.pos;
}

}  // namespace

extern "C" {

const char *flg_last_error(void) { return g_err; }

int flg_city_positions(int width, int height, int max_num_cities, int grid_mode, int max_rails_between_cities,
                       int max_rail_pairs_in_city, uint32_t *mt_key, int *mt_pos, int *n_cities, int32_t *city_positions) {
    if (!mt_key || !mt_pos || !n_cities || !city_positions || width <= 0 || height <= 0 || *mt_pos < 0 || *mt_pos > 624) { set_err("flg_city_positions: bad argument"); return FLG_ERR_ARG; }
    const CityPlan plan = city_plan(max_rails_between_cities, max_rail_pairs_in_city);
    const int r = plan.radius;
    const int feasible = std::min(max_num_cities, ((height - 2) / (2 * (r + 1))) * ((width - 2) / (2 * (r + 1))));
    if (feasible < 2) { set_err("ERROR: Cannot fit more than one city in this map, no feasible environment possible!"); return FLG_ERR_INFEASIBLE; }
    Rng rng = load_rng(mt_key, *mt_pos);
    std::vector<Cell> cities = grid_mode ? evenly_distributed_cities(feasible, r, width, height) : random_cities(feasible, r, width, height, rng);
    if ((int)cities.size() < 2) cities = evenly_distributed_cities(feasible, r, width, height);  // "Changing to Grid mode to place at least 2 cities"
    *n_cities = (int)cities.size();
    for (size_t k = 0; k < cities.size(); k++) { city_positions[2 * k] = cities[k].first; city_positions[2 * k + 1] = cities[k].second; }
    store_rng(rng, mt_key, mt_pos);
    return FLG_OK;
}

int flg_generate(int width, int height, int n_agents, int grid_mode, int max_rails_between_cities, int max_rail_pairs_in_city,
                 int n_cities, const int32_t *city_positions, const int32_t *neighbour_order, int n_speeds,
                 const double *speed_values, const double *speed_probs, uint32_t *mt_key, int *mt_pos, uint16_t *grid,
                 int32_t *city_orientations, int32_t *n_stations, int32_t *stations, int max_stations, int32_t *init_pos,
                 int32_t *init_dir, int32_t *target, double *speed, int32_t *earliest, int32_t *latest,
                 int32_t *max_episode_steps) {
    return flg_generate_seeded_rail(width, height, n_agents, grid_mode, max_rails_between_cities, max_rail_pairs_in_city, n_cities,
                                    city_positions, neighbour_order, n_speeds, speed_values, speed_probs, nullptr, nullptr, mt_key, mt_pos,
                                    grid, city_orientations, n_stations, stations, max_stations, init_pos, init_dir, target, speed,
                                    earliest, latest, max_episode_steps);
}

int flg_generate_seeded_rail(int width, int height, int n_agents, int grid_mode, int max_rails_between_cities, int max_rail_pairs_in_city,
                             int n_cities, const int32_t *city_positions, const int32_t *neighbour_order, int n_speeds,
                             const double *speed_values, const double *speed_probs, uint32_t *rail_mt_key, int *rail_mt_pos,
                             uint32_t *mt_key, int *mt_pos, uint16_t *grid,
                             int32_t *city_orientations, int32_t *n_stations, int32_t *stations, int max_stations, int32_t *init_pos,
                             int32_t *init_dir, int32_t *target, double *speed, int32_t *earliest, int32_t *latest,
                             int32_t *max_episode_steps) {
    if ((rail_mt_key != nullptr) != (rail_mt_pos != nullptr) || (rail_mt_pos && (*rail_mt_pos < 0 || *rail_mt_pos > 624))) { set_err("flg_generate_seeded_rail: bad rail stream"); return FLG_ERR_ARG; }
    if (!city_positions || !mt_key || !mt_pos || !grid || !init_pos || !init_dir || !target || !speed || !earliest || !latest ||
        !max_episode_steps || n_cities < 2 || n_agents <= 0 || width <= 0 || height <= 0 || *mt_pos < 0 || *mt_pos > 624 ||
        (n_speeds > 0 && (!speed_values || !speed_probs))) { set_err("flg_generate: bad argument"); return FLG_ERR_ARG; }
    const CityPlan plan = city_plan(max_rails_between_cities, max_rail_pairs_in_city);
    std::vector<Cell> cities(n_cities);
    for (int k = 0; k < n_cities; k++) cities[k] = Cell(city_positions[2 * k], city_positions[2 * k + 1]);
    Rng rng = load_rng(mt_key, *mt_pos);
    RailOut R;
    int rc;
    if (rail_mt_key) {  // SparseRailGen(seed=...): the rail is drawn from its own RandomState(seed) (rail_generators.py:221-222),
                        // lines and timetable stay on the env's stream
        Rng rail_rng = load_rng(rail_mt_key, *rail_mt_pos);
        rc = build_rail(width, height, plan, grid_mode != 0, cities, neighbour_order, rail_rng, R);
        store_rng(rail_rng, rail_mt_key, rail_mt_pos);
    } else {
        rc = build_rail(width, height, plan, grid_mode != 0, cities, neighbour_order, rng, R);
    }
    if (rc != FLG_OK) return rc;
    const Grid &G = R.G;
    memcpy(grid, G.g.data(), G.g.size() * 2);
    for (int c = 0; c < n_cities; c++) {
        if (city_orientations) city_orientations[c] = R.orientations[c];
        if (n_stations) n_stations[c] = (int)R.stations[c].size();
        if (stations)
            for (int k = 0; k < max_stations; k++) {
                int32_t *s = stations + ((size_t)c * max_stations + k) * 3;
                if (k < (int)R.stations[c].size()) { s[0] = R.stations[c][k].first.first; s[1] = R.stations[c][k].first.second; s[2] = R.stations[c][k].second; }
                else s[0] = s[1] = s[2] = -1;
            }
    }
    rc = make_schedule(G, n_agents, n_cities, R.orientations, R.stations, n_speeds, speed_values, speed_probs, rng, init_pos, init_dir,
                       target, speed, earliest, latest, max_episode_steps);
    if (rc != FLG_OK) return rc;
    store_rng(rng, mt_key, mt_pos);
    return FLG_OK;
}

int flg_timetable(int width, int height, const uint16_t *grid, int n_agents, int n_cities, const int32_t *init_pos, const int32_t *init_dir,
                  const int32_t *target, const double *speed, uint32_t *mt_key, int *mt_pos, int32_t *earliest, int32_t *latest,
                  int32_t *max_episode_steps) {
    if (!grid || !init_pos || !init_dir || !target || !speed || !mt_key || !mt_pos || !earliest || !latest || !max_episode_steps ||
        width <= 0 || height <= 0 || n_agents <= 0 || n_cities <= 0 || *mt_pos < 0 || *mt_pos > 624) { set_err("flg_timetable: bad argument"); return FLG_ERR_ARG; }
    Grid G;
    G.H = height; G.W = width;
    G.g.assign(grid, grid + (size_t)width * height);
    for (int a = 0; a < n_agents; a++) {
        if (!G.inside(init_pos[2 * a], init_pos[2 * a + 1]) || !G.inside(target[2 * a], target[2 * a + 1]) || init_dir[a] < 0 || init_dir[a] > 3 ||
            !(speed[a] > 0.0)) { set_err("flg_timetable: agent %d: position / direction / speed out of range", a); return FLG_ERR_ARG; }
    }
    Rng rng = load_rng(mt_key, *mt_pos);
    const int rc = make_timetable(G, n_agents, n_cities, init_pos, init_dir, target, speed, rng, earliest, latest, max_episode_steps);
    if (rc != FLG_OK) return rc;
    store_rng(rng, mt_key, mt_pos);
    return FLG_OK;
}

}  // extern "C"
