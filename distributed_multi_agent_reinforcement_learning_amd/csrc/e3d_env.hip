// e3d_env.hip -- batched env_3d (continuous 3-D pursuit, continuous actions) for MI355X (gfx950).  C ABI: include/e3d_env.h.
//
// The environment is tiny (7 (P + 1) + 3 doubles), so a wavefront per environment would leave 56 of 64 lanes idle and the
// launch bound by per-wave latency (that is where the env_n2n kernel sits: 9 % of HBM).  Here lane = (environment, pursuer):
// a group of PT = 8 / 16 / 32 / 64 lanes owns one environment (8 environments per wave for P <= 8), four waves per
// workgroup.  Every lane keeps its pursuer in registers; the evader (one per environment) is replicated in the group's lanes
// and advanced redundantly, so nothing about it needs an exchange; the pairwise kill-radius / range tests read the other
// pursuers through wave shuffles inside the group.  f64 state like the reference; headings go through the device cos/sin
// (agreement with the reference's libm: <= 1e-9 on positions over an episode, see tests).  Build with -ffp-contract=off;
// the only fused multiply-adds are the explicit ones in norm3 (how numpy evaluates the norm of a 3-vector).
#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include <stdint.h>
#include <string.h>

#include <thread>
#include <vector>

#include "e3d_env.h"
#include "rng_replica.hpp"

namespace {

constexpr int WAVE = 64, WPB = 4;
constexpr double PI = 3.14159265358979323846;

__host__ __device__ inline double norm3(double a, double b, double c) { return sqrt(fma(c, c, fma(b, b, a * a))); }
// `norm3(..) <= r` without the square root (sqrt is correctly rounded and monotonic): the squared norm -- the same fma chain the
// norm takes the root of -- against the largest double t with sqrt(t) <= r, computed once per launch on the host
__device__ __forceinline__ double sq3(double a, double b, double c) { return fma(c, c, fma(b, b, a * a)); }
double sq_threshold(double r) {
    auto ok = [&](double t) { return sqrt(t) <= r; };
    if (!(r >= 0.0) || !ok(0.0)) return -1.0;
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
struct E3dThr { double kill, comm, sen; };
__device__ __forceinline__ double sgn(double v) { return (double)((v > 0) - (v < 0)); }
__device__ __forceinline__ double clip(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Agent { double x, y, z, phi, gamma, v, act; };

// particle_env.py:25-55 Point.step
__device__ __forceinline__ void point_step(Agent &s, double a0, double a1, double a2, double v_max, double ang, double vlmt, double h) {
    if (s.act == 0.0) return;
    const double phi = a0 * PI, gamma = a1 * PI / 2, v = (a2 + 1) / 2 * v_max;
    s.gamma += clip(gamma - s.gamma, -ang, ang);
    s.v += clip(v - s.v, -vlmt, vlmt);
    const double d = phi - s.phi, ad = fabs(d);
    double dphi;
    if (sgn(phi * s.phi) >= 0) dphi = clip(d, -ang, ang);
    else if (ad < 2 * PI - ad) dphi = clip(d, -ang, ang);
    else dphi = clip(2 * PI - ad, 0, ang) * -sgn(d);
    s.phi += dphi;
    if (s.phi > PI) s.phi -= 2 * PI; else if (s.phi < -PI) s.phi += 2 * PI;
    s.x += s.v * cos(s.gamma) * cos(phi) * h;
    s.y += s.v * cos(s.gamma) * sin(phi) * h;
    s.z += s.v * sin(s.gamma) * h;
}

template <int PT, bool TICK>
__global__ __launch_bounds__(WAVE * WPB) void k_e3d(const e3d_config c, const e3d_state st, const double *actions, const double *e_cmd, float *reward,
                                                    uint8_t *active, uint8_t *done, const e3d_obs_out o, const E3dThr th) {
    constexpr int G = WAVE / PT;  // environments per wavefront
    const int lane = threadIdx.x & (WAVE - 1), wave = blockIdx.x * WPB + (threadIdx.x >> 6);
    const int g = lane / PT, a = lane - g * PT, base = lane - a;  // base: first lane of this lane's group
    const int env = wave * G + g, P = c.P;
    const bool ev = env < st.N, pv = ev && a < P;
    Agent s = {0, 0, 0, 0, 0, 0, 0}, e = {0, 0, 0, 0, 0, 0, 0};
    double tx = 0, ty = 0, tz = 0;
    if (pv) {
        const double *gp = st.p + (size_t)env * 7 * P + a;
        s.x = gp[0]; s.y = gp[P]; s.z = gp[2 * P]; s.phi = gp[3 * P]; s.gamma = gp[4 * P]; s.v = gp[5 * P]; s.act = gp[6 * P];
    }
    if (ev) {
        const double *ge = st.e + (size_t)env * 7;
        e.x = ge[0]; e.y = ge[1]; e.z = ge[2]; e.phi = ge[3]; e.gamma = ge[4]; e.v = ge[5]; e.act = ge[6];
    }
    if (TICK) {
        if (ev) { tx = st.target[3 * env]; ty = st.target[3 * env + 1]; tz = st.target[3 * env + 2]; }
        // evader_step (:354-378): the SLSQP command moves the evader; the reference's driver calls it with the ACTIVE pursuers
        // and cannot when none is left.  Every lane of the group advances its own copy of the evader.
        const bool any_p = (__ballot(pv && s.act != 0.0) >> base) & ((PT == 64) ? ~0ull : ((1ull << PT) - 1ull));
        if (ev && any_p && e.act != 0.0) {
            const double *cm = e_cmd + (size_t)env * 3;
            point_step(e, cm[0], cm[1], cm[2], c.e_vmax, c.ang_lmt, c.v_lmt, c.step_size);
        }
        if (pv) {
            const double *ac = actions + ((size_t)env * P + a) * 3;
            point_step(s, ac[0], ac[1], ac[2], c.p_vmax, c.ang_lmt, c.v_lmt, c.step_size);
        }
        // reward (:267-284) and update_agent_active (:286-326) are both evaluated on the moved, not yet culled state
        int cp = 0;
        for (int k = 0; k < P; k++) {  // wave-uniform trip count; partners through shuffles inside the group
            const double kx = __shfl(s.x, base + k), ky = __shfl(s.y, base + k), kz = __shfl(s.z, base + k), ka = __shfl(s.act, base + k);
            cp += ka != 0.0 && sq3(s.x - kx, s.y - ky, s.z - kz) <= th.kill;
        }
        const bool me = pv && s.act != 0.0;
        const int ce = me && e.act != 0.0 && sq3(s.x - e.x, s.y - e.y, s.z - e.z) <= th.kill;
        const bool e_hit = me && e.act != 0.0 && sq3(e.x - s.x, e.y - s.y, e.z - s.z) <= th.kill;
        const bool pdie = me && (cp + ce - 1) != 0;
        const bool edie = ((__ballot(e_hit) >> base) & ((PT == 64) ? ~0ull : ((1ull << PT) - 1ull))) != 0ull;
        if (pv) reward[(size_t)env * P + a] = me ? (float)(ce - (cp - 1)) : 0.f;
        if (pdie) { s.x = s.y = s.z = 1000; s.phi = s.gamma = s.v = 0; s.act = 0; }
        if (edie) { e.x = e.y = e.z = 1000; e.phi = e.gamma = e.v = 0; e.act = 0; }
        const bool pact = pv && s.act != 0.0;
        const int pa = __popcll((__ballot(pact) >> base) & ((PT == 64) ? ~0ull : ((1ull << PT) - 1ull)));
        if (pv) {
            active[(size_t)env * P + a] = pact;
            double *gp = st.p + (size_t)env * 7 * P + a;
            gp[0] = s.x; gp[P] = s.y; gp[2 * P] = s.z; gp[3 * P] = s.phi; gp[4 * P] = s.gamma; gp[5 * P] = s.v; gp[6 * P] = s.act;
        }
        if (ev && a == 0) {
            double *ge = st.e + (size_t)env * 7;
            ge[0] = e.x; ge[1] = e.y; ge[2] = e.z; ge[3] = e.phi; ge[4] = e.gamma; ge[5] = e.v; ge[6] = e.act;
            const int t = st.time_step[env] + 1;
            st.time_step[env] = t;
            const bool reach = sq3(e.x - tx, e.y - ty, e.z - tz) <= th.kill;   // get_done (:221-241)
            done[env] = (uint8_t)(reach || pa == 0 || e.act == 0.0 || t >= c.max_step);
        }
    }
    // observations (get_team_state rules=False :247-265, get_adj_mat :328-340: rows of inactive pursuers are zero)
    if (o.p_state && pv) {
        float *d = o.p_state + (int64_t)env * o.p_state_stride + a * 6;
        d[0] = (float)s.x; d[1] = (float)s.y; d[2] = (float)s.z; d[3] = (float)s.phi; d[4] = (float)s.gamma; d[5] = (float)s.v;
    }
    if (o.e_state && ev && a == 0) {
        float *d = o.e_state + (int64_t)env * o.e_state_stride;
        d[0] = (float)e.x; d[1] = (float)e.y; d[2] = (float)e.z; d[3] = (float)e.phi; d[4] = (float)e.gamma; d[5] = (float)e.v;
    }
    if (o.pp_adj) {
        for (int k = 0; k < P; k++) {  // row k, column a: the lanes of a group store consecutive floats
            const double kx = __shfl(s.x, base + k), ky = __shfl(s.y, base + k), kz = __shfl(s.z, base + k), ka = __shfl(s.act, base + k);
            if (pv) o.pp_adj[(int64_t)env * o.pp_adj_stride + k * P + a] = (ka != 0.0 && sq3(kx - s.x, ky - s.y, kz - s.z) <= th.comm) ? 1.f : 0.f;
        }
    }
    if (o.pe_adj && pv)
        o.pe_adj[(int64_t)env * o.pe_adj_stride + a] = (s.act != 0.0 && sq3(s.x - e.x, s.y - e.y, s.z - e.z) <= th.sen) ? 1.f : 0.f;
}

// [N][P][7] host order -> [N][7][P] records
__global__ void k_aos_to_soa7(int N, int A, const double *aos, double *soa) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * A * 7) return;
    const int n = i / (7 * A), r = i - n * 7 * A, k = r / A, a = r - k * A;
    soa[i] = aos[(size_t)n * 7 * A + a * 7 + k];
}

template <bool TICK>
int launch(const e3d_config *c, const e3d_state *st, const double *actions, const double *e_cmd, float *reward, uint8_t *active, uint8_t *done,
           const e3d_obs_out &o, hipStream_t s) {
    const int pt = c->P <= 8 ? 8 : (c->P <= 16 ? 16 : (c->P <= 32 ? 32 : 64));
    const int envs_per_block = (WAVE / pt) * WPB, blocks = (st->N + envs_per_block - 1) / envs_per_block;
    const E3dThr th{sq_threshold(c->kill_radius), sq_threshold(c->p_comm_range), sq_threshold(c->p_sen_range)};
#define E3D_GO(PT) hipLaunchKernelGGL((k_e3d<PT, TICK>), dim3(blocks), dim3(WAVE * WPB), 0, s, *c, *st, actions, e_cmd, reward, active, done, o, th)
    if (pt == 8) E3D_GO(8); else if (pt == 16) E3D_GO(16); else if (pt == 32) E3D_GO(32); else E3D_GO(64);
#undef E3D_GO
    return (int)hipGetLastError();
}

struct E3dResetter { e3d_config cfg; int N; std::vector<rngrep::NpRandom> rng; };

}  // namespace

extern "C" {

int e3d_config_check(const e3d_config *c) {
    if (!c) return E3D_ERR_NULL;
    if (c->P < 1 || c->P > E3D_MAX_P || c->max_step < 1) return E3D_ERR_BAD_CONFIG;
    return 0;
}

int e3d_env_load(const e3d_config *cfg, const e3d_state *st, const double *p, const double *e, const double *target, void *stream) {
    if (!cfg || !st || !p || !e || !target) return E3D_ERR_NULL;
    int rc = e3d_config_check(cfg);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t N = st->N, P = cfg->P;
    double *dp = nullptr;
    hipError_t err;
#define TRY(x) do { err = (x); if (err != hipSuccess) return (int)err; } while (0)
    TRY(hipMallocAsync((void **)&dp, N * P * 7 * sizeof(double), s));
    TRY(hipMemcpyAsync(dp, p, N * P * 7 * sizeof(double), hipMemcpyHostToDevice, s));
    TRY(hipMemcpyAsync(st->e, e, N * 7 * sizeof(double), hipMemcpyHostToDevice, s));
    TRY(hipMemcpyAsync(st->target, target, N * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    TRY(hipMemsetAsync(st->time_step, 0, N * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_aos_to_soa7, dim3((unsigned)((N * P * 7 + 255) / 256)), dim3(256), 0, s, (int)N, (int)P, dp, st->p);
    TRY(hipFreeAsync(dp, s));
#undef TRY
    return (int)hipGetLastError();
}

int e3d_env_observe(const e3d_config *cfg, const e3d_state *st, const e3d_obs_out *out, void *stream) {
    if (!cfg || !st || !out) return E3D_ERR_NULL;
    int rc = e3d_config_check(cfg);
    if (rc) return rc;
    return launch<false>(cfg, st, nullptr, nullptr, nullptr, nullptr, nullptr, *out, (hipStream_t)stream);
}

int e3d_env_tick(const e3d_config *cfg, const e3d_state *st, const double *actions, const double *e_cmd, float *reward, uint8_t *active,
                 uint8_t *done, const e3d_obs_out *out, void *stream) {
    if (!cfg || !st || !actions || !e_cmd || !reward || !active || !done) return E3D_ERR_NULL;
    int rc = e3d_config_check(cfg);
    if (rc) return rc;
    e3d_obs_out o0;
    memset(&o0, 0, sizeof o0);
    return launch<true>(cfg, st, actions, e_cmd, reward, active, done, out ? *out : o0, (hipStream_t)stream);
}

void *e3d_resetter_create(const e3d_config *cfg, int32_t N, const uint32_t *seeds) {
    if (!cfg || !seeds || N < 1 || e3d_config_check(cfg)) return nullptr;
    E3dResetter *R = new E3dResetter();
    R->cfg = *cfg;
    R->N = N;
    R->rng.resize(N);
    for (int n = 0; n < N; n++) R->rng[n].seed(seeds[n]);
    return R;
}

void e3d_resetter_destroy(void *h) { delete (E3dResetter *)h; }

int e3d_resetter_reset(void *h, double *p, double *e, double *target, int32_t n_threads) {
    if (!h || !p || !e || !target) return E3D_ERR_NULL;
    E3dResetter &R = *(E3dResetter *)h;
    const int P = R.cfg.P;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > R.N) n_threads = R.N;
    std::vector<int> failed(n_threads, 0);
    auto work = [&](int t) {
        std::vector<double> pts;
        for (int n = t; n < R.N; n += n_threads) {
            rngrep::NpRandom &g = R.rng[n];
            double *tg = target + 3 * (size_t)n;
            for (int k = 0; k < 3; k++) tg[k] = g.random_sample() * 20;   // reset (:138-142)
            // gen_init_p_pos (:151-164): normal(10, 2, size 3).clip(5, 15), rejected when < 4 from an earlier point
            pts.clear();
            int draws = 0;
            while ((int)pts.size() < 3 * P) {
                double q[3];
                for (int k = 0; k < 3; k++) { const double v = g.normal(10.0, 2.0); q[k] = v < 5.0 ? 5.0 : (v > 15.0 ? 15.0 : v); }
                bool ok = true;
                if (++draws > E3D_RESET_MAX_DRAWS) failed[t] = 1;   // give up: keep this candidate, report the environment
                else for (size_t k = 0; k < pts.size() && ok; k += 3) ok = !(norm3(q[0] - pts[k], q[1] - pts[k + 1], q[2] - pts[k + 2]) < 4.0);
                if (ok) { pts.push_back(q[0]); pts.push_back(q[1]); pts.push_back(q[2]); }
            }
            for (int i = 0; i < P; i++) {
                double *s = p + ((size_t)n * P + i) * 7;
                s[0] = pts[3 * i]; s[1] = pts[3 * i + 1]; s[2] = pts[3 * i + 2];
                s[3] = (2 * g.random_sample() - 1) * PI;
                s[4] = (2 * g.random_sample() - 1) * PI / 2;
                s[5] = 0.0; s[6] = 1.0;
            }
            double *s = e + (size_t)n * 7;
            s[0] = 20 - tg[0]; s[1] = 20 - tg[1]; s[2] = 20 - tg[2];
            s[3] = (2 * g.random_sample() - 1) * PI;
            s[4] = (2 * g.random_sample() - 1) * PI / 2;
            s[5] = 0.0; s[6] = 1.0;
        }
    };
    if (n_threads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < n_threads; t++)
        if (failed[t]) return E3D_ERR_RESET_FAILED;
    return 0;
}

}  // extern "C"
