// n2n_env.hip -- batched env_n2n (continuous 2-D pursuit, no obstacles) for MI355X (gfx950).  C ABI: include/n2n_env.h.
// Several environments per wavefront (lane = (environment, agent slot), see k_n2n); the environment's record (5 x (P + E)
// doubles) is contiguous in HBM, every agent lives in its lane's registers, pairwise kill-radius / range tests go through
// wave shuffles.  f64 state like the reference; headings go through
// the device cos/sin (agreement with the reference's libm: <= 1e-9 on positions over an episode, see tests).
// Build with -ffp-contract=off; the only fused multiply-add is the explicit one in norm2 (numpy's 2-vector norm).
#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include <stdint.h>
#include <string.h>

#include <thread>
#include <vector>

#include "n2n_env.h"
#include "rng_replica.hpp"

namespace {

constexpr int WAVE = 64;
constexpr double PI = 3.14159265358979323846;

__host__ __device__ inline double norm2(double a, double b) { return sqrt(fma(b, b, a * a)); }
// `norm2(a, b) <= r` without the square root: sqrt is correctly rounded and monotonic, so it holds exactly when the squared norm
// (the same fma the norm takes the root of) is <= the largest double t with sqrt(t) <= r, computed once per launch on the host.
__device__ __forceinline__ double sq2(double a, double b) { return fma(b, b, a * a); }
double sq_threshold(double r) {
    auto ok = [&](double t) { return sqrt(t) <= r; };
    if (!(r >= 0.0) || !ok(0.0)) return -1.0;  // a squared norm is >= 0: nothing qualifies
    if (std::isinf(r)) return r;
    double t = r * r;
    while (!ok(t)) t = nextafter(t, 0.0);
    for (;;) {
        const double n = nextafter(t, INFINITY);
        if (std::isinf(n) || !ok(n)) break;
        t = n;
    }
    return t;
}
struct N2nThr { double kill, comm, sen; };
__device__ __forceinline__ double sgn(double v) { return (double)((v > 0) - (v < 0)); }

// particle_env.py:41-57 / :78-90 : signed heading change towards the commanded heading a, limited to ang_lmt
__device__ __forceinline__ double turn(double a, double phi, double lim) {
    double sign, delta;
    const double d = fabs(a - phi);
    if (sgn(a * phi) >= 0) { delta = d; sign = sgn(a - phi); }
    else if (d < 2 * PI - d) { delta = d; sign = sgn(a - phi); }
    else { delta = 2 * PI - d; sign = -sgn(a - phi); }
    delta = delta > lim ? lim : (delta < 0 ? 0 : delta);
    return sign * delta;
}
__device__ __forceinline__ double wrap(double phi) { return phi > PI ? phi - 2 * PI : (phi < -PI ? phi + 2 * PI : phi); }

// lane = (environment, agent slot): a group of PT lanes owns one environment (PT = power of two >= max(P, E)), G = 64 / PT
// environments per wavefront, four wavefronts per workgroup.  Slot a holds pursuer a (a < P) AND evader a (a < E) in
// registers; partners are read through wave shuffles inside the group.  (One wavefront per environment left 48 of 64 lanes
// idle at P = 16 and the launch bound by per-wave latency: 9 % of the HBM roofline in round 1.)
constexpr int WPB = 4;

template <int PT, bool TICK>
__global__ __launch_bounds__(WAVE * WPB) void k_n2n(const n2n_config c, const n2n_state st, const int32_t *actions, const double *e_cmd, float *reward,
                                                    uint8_t *active, uint8_t *done, const n2n_obs_out o, const N2nThr th) {
    constexpr int G = WAVE / PT;
    constexpr unsigned long long GM = (PT == 64) ? ~0ull : ((1ull << PT) - 1ull);
    const int lane = threadIdx.x & (WAVE - 1), wave = blockIdx.x * WPB + (threadIdx.x >> 6);
    const int g = lane / PT, a = lane - g * PT, base = lane - a;
    const int env = wave * G + g, P = c.P, E = c.E;
    const bool ev = env < st.N, pv = ev && a < P, evv = ev && a < E;
    double px = 0, py = 0, pphi = 0, pvel = 0, pact = 0, ex = 0, ey = 0, ephi = 0, evel = 0, eact = 0;
    double *gp = st.p + (size_t)(ev ? env : 0) * 5 * P, *ge = st.e + (size_t)(ev ? env : 0) * 5 * E;
    if (pv) { px = gp[a]; py = gp[P + a]; pphi = gp[2 * P + a]; pvel = gp[3 * P + a]; pact = gp[4 * P + a]; }
    if (evv) { ex = ge[a]; ey = ge[E + a]; ephi = ge[2 * E + a]; evel = ge[3 * E + a]; eact = ge[4 * E + a]; }
    if (TICK) {
        // Evader.step (:74-99): position with the OLD heading, then the heading turns towards the command
        if (evv && eact != 0.0) {
            const double d = turn(e_cmd[(size_t)env * E + a] * PI, ephi, c.ang_lmt);
            ex += evel * cos(ephi) * c.step_size;
            ey += evel * sin(ephi) * c.step_size;
            ephi = wrap(ephi + d);
        }
        // Pursuer.step (:34-67): the heading turns even when the pursuer is inactive, the position only moves when active
        if (pv) {
            const int a_i = actions[(size_t)env * P + a];
            double v = 0.0;
            if (a_i != 0) {
                v = c.p_vmax;
                double ang = (double)a_i * PI / 4;
                if (ang > PI) ang -= 2 * PI;
                pphi = wrap(pphi + turn(ang, pphi, c.ang_lmt));
            }
            if (pact != 0.0) {
                px += v * cos(pphi) * c.step_size;
                py += v * sin(pphi) * c.step_size;
                pvel = v;
            }
        }
        // reward (:316-334) and update_agent_active (:336-365) are both evaluated on the moved, not yet culled state
        int ce = 0, cp = 0, chit = 0;
        for (int k = 0; k < E; k++) {
            const double kx = __shfl(ex, base + k), ky = __shfl(ey, base + k), ka = __shfl(eact, base + k);
            ce += ka != 0.0 && sq2(px - kx, py - ky) <= th.kill;
        }
        for (int k = 0; k < P; k++) {
            const double kx = __shfl(px, base + k), ky = __shfl(py, base + k), ka = __shfl(pact, base + k);
            cp += ka != 0.0 && sq2(px - kx, py - ky) <= th.kill;
            chit += ka != 0.0 && sq2(ex - kx, ey - ky) <= th.kill;   // the evader of this slot against pursuer k
        }
        const bool p_on = pv && pact != 0.0, e_on = evv && eact != 0.0;
        if (pv) reward[(size_t)env * P + a] = p_on ? (float)(ce - (cp - 1)) : 0.f;
        if (p_on && (cp + ce - 1) != 0) { px = 1000; py = 1000; pphi = 0; pact = 0; }
        if (e_on && chit != 0) { ex = 1000; ey = 1000; ephi = 0; eact = 0; }
        const bool pact_b = pv && pact != 0.0, eact_b = evv && eact != 0.0;
        double tx = 0, ty = 0;
        if (ev) { tx = st.target[2 * env]; ty = st.target[2 * env + 1]; }
        const bool reach = evv && sq2(ex - tx, ey - ty) <= th.kill;  // get_done (:283-304), all evaders
        const int pa = __popcll((__ballot(pact_b) >> base) & GM), ea = __popcll((__ballot(eact_b) >> base) & GM);
        const bool rc = ((__ballot(reach) >> base) & GM) != 0ull;
        if (pv) {
            active[(size_t)env * P + a] = pact_b;
            gp[a] = px; gp[P + a] = py; gp[2 * P + a] = pphi; gp[3 * P + a] = pvel; gp[4 * P + a] = pact;
        }
        if (evv) { ge[a] = ex; ge[E + a] = ey; ge[2 * E + a] = ephi; ge[3 * E + a] = evel; ge[4 * E + a] = eact; }
        if (ev && a == 0) {
            const int t = st.time_step[env] + 1;
            st.time_step[env] = t;
            done[env] = (uint8_t)(rc || pa == 0 || ea == 0 || t >= c.episode_limit);
        }
    }
    // observations (get_team_state rules=False, get_adj_mat :386-397: rows of inactive pursuers are zero)
    if (o.p_state && pv) {
        float *d = o.p_state + (int64_t)env * o.p_state_stride + 3 * a;
        d[0] = (float)px; d[1] = (float)py; d[2] = (float)pphi;
    }
    if (o.e_state && evv) {
        float *d = o.e_state + (int64_t)env * o.e_state_stride + 3 * a;
        d[0] = (float)ex; d[1] = (float)ey; d[2] = (float)ephi;
    }
    if (o.pp_adj)
        for (int k = 0; k < P; k++) {  // row k, column a: the lanes of a group store consecutive floats
            const double kx = __shfl(px, base + k), ky = __shfl(py, base + k), ka = __shfl(pact, base + k);
            if (pv) o.pp_adj[(int64_t)env * o.pp_adj_stride + k * P + a] = (ka != 0.0 && sq2(kx - px, ky - py) <= th.comm) ? 1.f : 0.f;
        }
    if (o.pe_adj)
        for (int k = 0; k < E; k++) {
            const double kx = __shfl(ex, base + k), ky = __shfl(ey, base + k);
            if (pv) o.pe_adj[(int64_t)env * o.pe_adj_stride + a * E + k] = (pact != 0.0 && sq2(px - kx, py - ky) <= th.sen) ? 1.f : 0.f;
        }
}

template <bool TICK>
int launch_n2n(const n2n_config *c, const n2n_state *st, const int32_t *actions, const double *e_cmd, float *reward, uint8_t *active, uint8_t *done,
               const n2n_obs_out &o, hipStream_t s) {
    const int m = c->P > c->E ? c->P : c->E;
    const int pt = m <= 8 ? 8 : (m <= 16 ? 16 : (m <= 32 ? 32 : 64));
    const int envs_per_block = (WAVE / pt) * WPB, blocks = (st->N + envs_per_block - 1) / envs_per_block;
    const N2nThr th{sq_threshold(c->kill_radius), sq_threshold(c->p_comm_range), sq_threshold(c->p_sen_range)};
#define N2N_GO(PT) hipLaunchKernelGGL((k_n2n<PT, TICK>), dim3(blocks), dim3(WAVE * WPB), 0, s, *c, *st, actions, e_cmd, reward, active, done, o, th)
    if (pt == 8) N2N_GO(8); else if (pt == 16) N2N_GO(16); else if (pt == 32) N2N_GO(32); else N2N_GO(64);
#undef N2N_GO
    return (int)hipGetLastError();
}

// [N][A][5] host order -> [N][5][A] records
__global__ void k_aos_to_soa(int N, int A, const double *aos, double *soa) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * A * 5) return;
    const int n = i / (5 * A), r = i - n * 5 * A, k = r / A, a = r - k * A;
    soa[i] = aos[(size_t)n * 5 * A + a * 5 + k];
}

struct N2nResetter { n2n_config cfg; int N; std::vector<rngrep::NpRandom> rng; };

void sample_points(rngrep::NpRandom &g, int n, double lx, double ly, double lo, double hi, std::vector<double> &pts) {
    // gen_init_p_pos / gen_init_e_pos (:239-281): normal(loc, 2, size 2).clip(lo, hi), rejected when < 2 from an earlier point
    pts.clear();
    while ((int)pts.size() < 2 * n) {
        double x = g.normal(lx, 2.0), y = g.normal(ly, 2.0);
        x = x < lo ? lo : (x > hi ? hi : x);
        y = y < lo ? lo : (y > hi ? hi : y);
        bool ok = true;
        for (size_t k = 0; k < pts.size() && ok; k += 2) ok = !(norm2(x - pts[k], y - pts[k + 1]) < 2.0);
        if (ok) { pts.push_back(x); pts.push_back(y); }
    }
}

}  // namespace

extern "C" {

int n2n_config_check(const n2n_config *c) {
    if (!c) return N2N_ERR_NULL;
    if (c->P < 1 || c->P > N2N_MAX_P || c->E < 1 || c->E > N2N_MAX_E || c->episode_limit < 1) return N2N_ERR_BAD_CONFIG;
    return 0;
}

int n2n_env_load(const n2n_config *cfg, const n2n_state *st, const double *p, const double *e, const double *target, void *stream) {
    if (!cfg || !st || !p || !e || !target) return N2N_ERR_NULL;
    int rc = n2n_config_check(cfg);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t N = st->N, P = cfg->P, E = cfg->E;
    double *dp = nullptr, *de = nullptr;
    hipError_t err;
#define TRY(x) do { err = (x); if (err != hipSuccess) return (int)err; } while (0)
    TRY(hipMallocAsync((void **)&dp, N * P * 5 * sizeof(double), s));
    TRY(hipMallocAsync((void **)&de, N * E * 5 * sizeof(double), s));
    TRY(hipMemcpyAsync(dp, p, N * P * 5 * sizeof(double), hipMemcpyHostToDevice, s));
    TRY(hipMemcpyAsync(de, e, N * E * 5 * sizeof(double), hipMemcpyHostToDevice, s));
    TRY(hipMemcpyAsync(st->target, target, N * 2 * sizeof(double), hipMemcpyHostToDevice, s));
    TRY(hipMemsetAsync(st->time_step, 0, N * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_aos_to_soa, dim3((unsigned)((N * P * 5 + 255) / 256)), dim3(256), 0, s, (int)N, (int)P, dp, st->p);
    hipLaunchKernelGGL(k_aos_to_soa, dim3((unsigned)((N * E * 5 + 255) / 256)), dim3(256), 0, s, (int)N, (int)E, de, st->e);
    TRY(hipFreeAsync(dp, s));
    TRY(hipFreeAsync(de, s));
#undef TRY
    return (int)hipGetLastError();
}

int n2n_env_observe(const n2n_config *cfg, const n2n_state *st, const n2n_obs_out *out, void *stream) {
    if (!cfg || !st || !out) return N2N_ERR_NULL;
    return launch_n2n<false>(cfg, st, nullptr, nullptr, nullptr, nullptr, nullptr, *out, (hipStream_t)stream);
}

int n2n_env_tick(const n2n_config *cfg, const n2n_state *st, const int32_t *actions, const double *e_cmd, float *reward, uint8_t *active,
                 uint8_t *done, const n2n_obs_out *out, void *stream) {
    if (!cfg || !st || !actions || !e_cmd || !reward || !active || !done) return N2N_ERR_NULL;
    n2n_obs_out o0;
    memset(&o0, 0, sizeof o0);
    return launch_n2n<true>(cfg, st, actions, e_cmd, reward, active, done, out ? *out : o0, (hipStream_t)stream);
}

void *n2n_resetter_create(const n2n_config *cfg, int32_t N, const uint32_t *seeds) {
    if (!cfg || !seeds || N < 1 || n2n_config_check(cfg)) return nullptr;
    N2nResetter *R = new N2nResetter();
    R->cfg = *cfg;
    R->N = N;
    R->rng.resize(N);
    for (int n = 0; n < N; n++) R->rng[n].seed(seeds[n]);
    return R;
}

void n2n_resetter_destroy(void *h) { delete (N2nResetter *)h; }

int n2n_resetter_reset(void *h, double *p, double *e, double *target, int32_t n_threads) {
    if (!h || !p || !e || !target) return N2N_ERR_NULL;
    N2nResetter &R = *(N2nResetter *)h;
    const int P = R.cfg.P, E = R.cfg.E;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > R.N) n_threads = R.N;
    auto work = [&](int t) {
        std::vector<double> pts;
        for (int n = t; n < R.N; n += n_threads) {
            rngrep::NpRandom &g = R.rng[n];
            const double tx = g.random_sample() * 20, ty = g.random_sample() * 20;  // reset (:200-204)
            target[2 * n] = tx; target[2 * n + 1] = ty;
            sample_points(g, P, 0.0, 0.0, -8.0, 8.0, pts);
            for (int i = 0; i < P; i++) {
                double *s = p + ((size_t)n * P + i) * 5;
                s[0] = pts[2 * i] + 10; s[1] = pts[2 * i + 1] + 10; s[2] = PI / 4; s[3] = 0.0; s[4] = 1.0;
            }
            sample_points(g, E, 20 - tx, 20 - ty, 0.0, 20.0, pts);
            for (int i = 0; i < E; i++) {
                double *s = e + ((size_t)n * E + i) * 5;
                s[0] = pts[2 * i]; s[1] = pts[2 * i + 1]; s[2] = PI / 4; s[3] = R.cfg.e_vmax; s[4] = 1.0;
            }
        }
    };
    if (n_threads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    return 0;
}

}  // extern "C"
