// pe_reset.cpp -- host side of Pursuit_Env.reset(): map, boundary obstacles, target, defenders, evader.
//
// Replaces (reference paths) pursuit_env.py:60-73, base_env.py:37-162, Occupied_Grid_Map.py:46-62,119-166 and the
// skimage 'inner' boundary of pursuit_env.py:18-27 for N environments, multi-threaded.  Each environment owns the two
// generator streams the reference consumes -- Python's `random` (Mersenne Twister + randbelow) and numpy's legacy
// RandomState (rand / normal) -- re-implemented bit-exactly, so environment n reset from `random.seed(s);
// np.random.seed(s)` starts exactly like the reference seeded with s, and keeps consuming its streams over
// episodes the way one reference Worker does (the targets the evader did not reach are un-drawn again).
// C ABI: include/pe_env.h (pe_resetter_*).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

#include "pe_env.h"
#include "rng_replica.hpp"

namespace {

using rngrep::NpRandom;
using rngrep::PyRandom;

inline int py_round(double v) { return (int)nearbyint(v); }
inline double norm2(double a, double b) { return sqrt(fma(b, b, a * a)); }

struct EnvRng {
    PyRandom py;
    NpRandom np;
    PyRandom py_before_tape;  // snapshot taken before the tape of the running episode was drawn
    bool has_tape = false;
};

struct Resetter {
    pe_config cfg;
    pe_reset_params prm;
    int N;
    std::vector<EnvRng> rng;
    std::vector<uint8_t> inflated;  // [N][W*H] static map inflated by 2 of the running episode (target re-draws)
};

void inflate(int W, int H, uint8_t *g, int x, int y, int ext) {
    for (int xx = x - ext; xx <= x + ext; xx++)
        for (int yy = y - ext; yy <= y + ext; yy++)
            if (xx >= 0 && xx < W && yy >= 0 && yy < H) g[xx * H + yy] = 1;
}

// base_env.py:52-70; false when PE_RESET_MAX_DRAWS candidates were all occupied (the last one is kept)
bool draw_target(int W, int H, const uint8_t *infl, PyRandom &py, int32_t *out) {
    for (int draws = 0; draws < PE_RESET_MAX_DRAWS; draws++) {
        int tx = py.randint(0, W - 1), ty = py.randint(0, H - 1);
        out[0] = tx; out[1] = ty;
        if (infl[tx * H + ty] == 0) return true;
    }
    return false;
}

// returns false when a placement loop gave up (PE_RESET_MAX_DRAWS); the arrays then hold the last candidates
bool reset_one(Resetter &R, int n, int consumed, const pe_host_init_out &o) {
    const pe_config &c = R.cfg;
    const int W = c.W, H = c.H, WH = W * H, P = c.P, O = c.O;
    EnvRng &rg = R.rng[n];
    uint8_t *infl_static = R.inflated.data() + (size_t)n * WH;
    if (rg.has_tape) {
        // the reference draws a new target only when the evader reaches one: replay exactly `consumed` draws
        rg.py = rg.py_before_tape;
        int32_t tmp[2];
        for (int k = 0; k < consumed; k++) draw_target(W, H, infl_static, rg.py, tmp);
    }
    bool all_ok = true;
    uint8_t *grid = o.grid + (size_t)n * WH;
    memset(grid, 0, WH);
    // init_map -> initailize_obstacle -> add_blocker_type('r', (6, 7)) : x, y in [-3, 3) (Occupied_Grid_Map.py:46-62)
    for (int b = 0; b < R.prm.num_blocks; b++) {
        rg.py.randbelow(1);  // random.randrange(len(shape)) with one shape
        double cx = rg.np.normal(R.prm.center[0], R.prm.variance), cy = rg.np.normal(R.prm.center[1], R.prm.variance);
        for (int x = -3; x < 3; x++)
            for (int y = -3; y < 3; y++) {
                int px = py_round((double)x + cx), py = py_round((double)y + cy);
                if (px >= 0 && px < W && py >= 0 && py < H) grid[px * H + py] = 1;
            }
    }
    std::vector<uint8_t> infl(WH);
    memcpy(infl.data(), grid, WH);
    for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++)
            if (grid[x * H + y]) inflate(W, H, infl.data(), x, y, 2);
    memcpy(infl_static, infl.data(), WH);
    // inner boundary (find_boundaries mode='inner', connectivity 1, mirrored edges): obstacle cell with a free 4-neighbour
    int32_t *obs = o.obs_xy + (size_t)n * O * 2;
    memset(obs, 0, sizeof(int32_t) * O * 2);
    int n_obs = 0;
    for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
            if (!grid[x * H + y]) continue;
            bool fr = (x > 0 && !grid[(x - 1) * H + y]) || (x < W - 1 && !grid[(x + 1) * H + y]) || (y > 0 && !grid[x * H + y - 1]) ||
                      (y < H - 1 && !grid[x * H + y + 1]);
            if (fr) {
                if (n_obs < O) { obs[2 * n_obs] = x; obs[2 * n_obs + 1] = y; }
                n_obs++;
            }
        }
    o.n_obs[n] = n_obs;  // > O is reported to the caller (the reference's buffer would not fit it either)
    all_ok = draw_target(W, H, infl.data(), rg.py, o.target + 2 * n) && all_ok;
    // init_defender (base_env.py:72-120)
    double *def = o.def + (size_t)n * P * 4;
    std::vector<int> cells;
    int placed = 0, draws = 0;
    while (placed < P) {
        double px = rg.np.random_sample() * (double)(W - 1), py = rg.np.random_sample() * (double)(H - 1);
        bool ok = false;
        if (++draws > PE_RESET_MAX_DRAWS) { ok = true; all_ok = false; }  // give up: keep this candidate, flag the environment
        else
        if (infl[py_round(px) * H + py_round(py)] == 0) {
            if (placed == 0) {
                ok = true;
            } else {
                int collision = 0, connectivity = 0;
                for (int k = 0; k < placed; k++) {
                    double d = norm2(px - def[k * 4], py - def[k * 4 + 1]);
                    if (d < (double)R.prm.min_dist) collision++;
                    if (d < c.def_comm_range) connectivity++;
                }
                ok = collision == 0 && connectivity > 0 && connectivity <= 2;
            }
        }
        if (ok) {
            def[placed * 4] = px; def[placed * 4 + 1] = py; def[placed * 4 + 2] = 0.0; def[placed * 4 + 3] = 0.0;
            placed++;
            int cxi = py_round(px), cyi = py_round(py);
            infl[cxi * H + cyi] = 1;
            bool seen = false;
            for (size_t k = 0; k < cells.size(); k += 2) seen = seen || (cells[k] == cxi && cells[k + 1] == cyi);
            if (!seen) { cells.push_back(cxi); cells.push_back(cyi); }
            for (size_t k = 0; k < cells.size(); k += 2) inflate(W, H, infl.data(), cells[k], cells[k + 1], 2);
        }
    }
    // init_attacker (base_env.py:122-162), is_percepted=True
    double *eva = o.eva + (size_t)n * 4;
    draws = 0;
    for (bool done = false; !done;) {
        double px = rg.np.random_sample() * (double)(W - 1), py = rg.np.random_sample() * (double)(H - 1);
        if (++draws > PE_RESET_MAX_DRAWS) {
            eva[0] = px; eva[1] = py; eva[2] = 0.0; eva[3] = 0.0;
            all_ok = false;
            break;
        }
        if (infl[py_round(px) * H + py_round(py)] != 0) continue;
        for (size_t k = 0; k < cells.size(); k += 2)
            if (norm2((double)cells[k] - px, (double)cells[k + 1] - py) < c.def_sen_range) {
                eva[0] = px; eva[1] = py; eva[2] = 0.0; eva[3] = 0.0;
                done = true;
                break;
            }
    }
    // target tape: what init_target would return on the evader's next arrivals (pursuit_env.py:98-100)
    rg.py_before_tape = rg.py;
    rg.has_tape = true;
    int32_t *tape = o.tape + (size_t)n * c.tape_len * 2;
    for (int k = 0; k < c.tape_len; k++) all_ok = draw_target(W, H, infl_static, rg.py, tape + 2 * k) && all_ok;
    return all_ok;
}

}  // namespace

extern "C" {

void *pe_resetter_create(const pe_config *cfg, const pe_reset_params *prm, int32_t N, const uint64_t *seeds) {
    if (prm && prm->fixed_grid) return nullptr;  // a map-bank slot is a device pointer: pe_env_reset only

    if (!cfg || !prm || !seeds || N < 1 || pe_config_check(cfg) != 0) return nullptr;
    Resetter *R = new Resetter();
    R->cfg = *cfg;
    R->prm = *prm;
    R->N = N;
    R->rng.resize(N);
    R->inflated.assign((size_t)N * cfg->W * cfg->H, 0);
    for (int n = 0; n < N; n++) {
        R->rng[n].py.seed(seeds[n]);
        R->rng[n].np.seed((uint32_t)seeds[n]);
    }
    return R;
}

void pe_resetter_destroy(void *h) { delete (Resetter *)h; }

// Resume support: the generator streams (and the inflated maps the running episode's target re-draws use) as one blob.
int64_t pe_resetter_state_bytes(void *h) {
    if (!h) return 0;
    Resetter &R = *(Resetter *)h;
    return (int64_t)(sizeof(EnvRng) * R.rng.size() + R.inflated.size());
}

int pe_resetter_get_state(void *h, void *out) {
    if (!h || !out) return PE_ERR_NULL;
    Resetter &R = *(Resetter *)h;
    memcpy(out, R.rng.data(), sizeof(EnvRng) * R.rng.size());
    memcpy((char *)out + sizeof(EnvRng) * R.rng.size(), R.inflated.data(), R.inflated.size());
    return 0;
}

int pe_resetter_set_state(void *h, const void *in) {
    if (!h || !in) return PE_ERR_NULL;
    Resetter &R = *(Resetter *)h;
    memcpy(R.rng.data(), in, sizeof(EnvRng) * R.rng.size());
    memcpy(R.inflated.data(), (const char *)in + sizeof(EnvRng) * R.rng.size(), R.inflated.size());
    return 0;
}

int pe_resetter_reset(void *h, const int32_t *consumed_targets, const pe_host_init_out *out, int32_t n_threads) {
    if (!h || !out) return PE_ERR_NULL;
    Resetter &R = *(Resetter *)h;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > R.N) n_threads = R.N;
    std::vector<int> failed(n_threads, 0);  // one slot per thread: no shared writes
    auto work = [&](int t) {
        for (int n = t; n < R.N; n += n_threads)
            if (!reset_one(R, n, consumed_targets ? consumed_targets[n] : 0, *out)) failed[t] = 1;
    };
    if (n_threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < n_threads; t++)
        if (failed[t]) return PE_ERR_RESET_FAILED;
    return 0;
}

}  // extern "C"
