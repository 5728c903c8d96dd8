// n2n_env.hip -- batched env_n2n (continuous 2-D pursuit, no obstacles) for MI355X (gfx950).  C ABI: include/n2n_env.h.
// One wavefront per environment, lane = pursuer; the environment's record (5 x (P + E) doubles) is contiguous in HBM and
// staged in LDS; all pairwise kill-radius / range tests run out of LDS.  f64 state like the reference; headings go through
// the device cos/sin (agreement with the reference's libm: <= 1e-9 on positions over an episode, see tests).
// Build with -ffp-contract=off; the only fused multiply-add is the explicit one in norm2 (numpy's 2-vector norm).
#include <hip/hip_runtime.h>
#include <math.h>
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

struct Lds { double *p, *e; };  // p: [5][P], e: [5][E]

__device__ void dev_observe(const n2n_config &c, const Lds &l, int lane, int env, const n2n_obs_out &o) {
    const int P = c.P, E = c.E;
    if (o.p_state) for (int i = lane; i < 3 * P; i += WAVE) o.p_state[(int64_t)env * o.p_state_stride + i] = (float)l.p[(i % 3) * P + i / 3];
    if (o.e_state) for (int i = lane; i < 3 * E; i += WAVE) o.e_state[(int64_t)env * o.e_state_stride + i] = (float)l.e[(i % 3) * E + i / 3];
    if (o.pp_adj)
        for (int idx = lane; idx < P * P; idx += WAVE) {
            const int i = idx / P, j = idx - i * P;
            const bool v = l.p[4 * P + i] != 0.0 && norm2(l.p[i] - l.p[j], l.p[P + i] - l.p[P + j]) <= c.p_comm_range;
            o.pp_adj[(int64_t)env * o.pp_adj_stride + idx] = v ? 1.f : 0.f;
        }
    if (o.pe_adj)
        for (int idx = lane; idx < P * E; idx += WAVE) {
            const int i = idx / E, j = idx - i * E;
            const bool v = l.p[4 * P + i] != 0.0 && norm2(l.p[i] - l.e[j], l.p[P + i] - l.e[E + j]) <= c.p_sen_range;
            o.pe_adj[(int64_t)env * o.pe_adj_stride + idx] = v ? 1.f : 0.f;
        }
}

template <bool TICK>
__global__ __launch_bounds__(WAVE) void k_n2n(const n2n_config c, const n2n_state st, const int32_t *actions, const double *e_cmd, float *reward,
                                              uint8_t *active, uint8_t *done, const n2n_obs_out o) {
    __shared__ double sp[5 * N2N_MAX_P], se[5 * N2N_MAX_E];
    __shared__ uint8_t pdie[N2N_MAX_P], edie[N2N_MAX_E];
    const int env = blockIdx.x, lane = threadIdx.x;
    if (env >= st.N) return;
    const int P = c.P, E = c.E;
    double *gp = st.p + (size_t)env * 5 * P, *ge = st.e + (size_t)env * 5 * E;
    for (int i = lane; i < 5 * P; i += WAVE) sp[i] = gp[i];
    for (int i = lane; i < 5 * E; i += WAVE) se[i] = ge[i];
    __syncthreads();
    Lds l{sp, se};
    if (TICK) {
        const double tx = st.target[2 * env], ty = st.target[2 * env + 1];
        // Evader.step (:74-99): position with the OLD heading, then the heading turns towards the command
        if (lane < E && se[4 * E + lane] != 0.0) {
            const double phi = se[2 * E + lane], v = se[3 * E + lane];
            const double d = turn(e_cmd[(size_t)env * E + lane] * PI, phi, c.ang_lmt);
            se[lane] += v * cos(phi) * c.step_size;
            se[E + lane] += v * sin(phi) * c.step_size;
            se[2 * E + lane] = wrap(phi + d);
        }
        // Pursuer.step (:34-67): the heading turns even when the pursuer is inactive, the position only moves when active
        if (lane < P) {
            const int a_i = actions[(size_t)env * P + lane];
            double v = 0.0, phi = sp[2 * P + lane];
            if (a_i != 0) {
                v = c.p_vmax;
                double a = (double)a_i * PI / 4;
                if (a > PI) a -= 2 * PI;
                phi = wrap(phi + turn(a, phi, c.ang_lmt));
                sp[2 * P + lane] = phi;
            }
            if (sp[4 * P + lane] != 0.0) {
                sp[lane] += v * cos(phi) * c.step_size;
                sp[P + lane] += v * sin(phi) * c.step_size;
                sp[3 * P + lane] = v;
            }
        }
        __syncthreads();
        // reward (:316-334) and update_agent_active (:336-365) are both evaluated on the moved, not yet culled state
        if (lane < P) {
            float r = 0.f;
            bool die = false;
            if (sp[4 * P + lane] != 0.0) {
                int ce = 0, cp = 0;
                for (int k = 0; k < E; k++) ce += se[4 * E + k] != 0.0 && norm2(sp[lane] - se[k], sp[P + lane] - se[E + k]) <= c.kill_radius;
                for (int k = 0; k < P; k++) cp += sp[4 * P + k] != 0.0 && norm2(sp[lane] - sp[k], sp[P + lane] - sp[P + k]) <= c.kill_radius;
                r = (float)(ce - (cp - 1));
                die = (cp + ce - 1) != 0;
            }
            reward[(size_t)env * P + lane] = r;
            pdie[lane] = die;
        }
        if (lane < E) {
            bool die = false;
            if (se[4 * E + lane] != 0.0) {
                int cnt = 0;
                for (int i = 0; i < P; i++) cnt += sp[4 * P + i] != 0.0 && norm2(se[lane] - sp[i], se[E + lane] - sp[P + i]) <= c.kill_radius;
                die = cnt != 0;
            }
            edie[lane] = die;
        }
        __syncthreads();
        if (lane < P && pdie[lane]) { sp[lane] = 1000; sp[P + lane] = 1000; sp[2 * P + lane] = 0; sp[4 * P + lane] = 0; }
        if (lane < E && edie[lane]) { se[lane] = 1000; se[E + lane] = 1000; se[2 * E + lane] = 0; se[4 * E + lane] = 0; }
        __syncthreads();
        const bool pact = lane < P && sp[4 * P + lane] != 0.0;
        const bool eact = lane < E && se[4 * E + lane] != 0.0;
        const bool reach = lane < E && norm2(se[lane] - tx, se[E + lane] - ty) <= c.kill_radius;  // get_done (:283-304), all evaders
        const int pa = __popcll(__ballot(pact)), ea = __popcll(__ballot(eact)), rc = __ballot(reach) != 0ull;
        if (lane < P) active[(size_t)env * P + lane] = pact;
        if (lane == 0) {
            const int t = st.time_step[env] + 1;
            st.time_step[env] = t;
            done[env] = (uint8_t)(rc || pa == 0 || ea == 0 || t >= c.episode_limit);
        }
        for (int i = lane; i < 5 * P; i += WAVE) gp[i] = sp[i];
        for (int i = lane; i < 5 * E; i += WAVE) ge[i] = se[i];
    }
    dev_observe(c, l, lane, env, o);
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
    hipLaunchKernelGGL(k_n2n<false>, dim3(st->N), dim3(WAVE), 0, (hipStream_t)stream, *cfg, *st, (const int32_t *)nullptr, (const double *)nullptr,
                       (float *)nullptr, (uint8_t *)nullptr, (uint8_t *)nullptr, *out);
    return (int)hipGetLastError();
}

int n2n_env_tick(const n2n_config *cfg, const n2n_state *st, const int32_t *actions, const double *e_cmd, float *reward, uint8_t *active,
                 uint8_t *done, const n2n_obs_out *out, void *stream) {
    if (!cfg || !st || !actions || !e_cmd || !reward || !active || !done) return N2N_ERR_NULL;
    n2n_obs_out o0;
    memset(&o0, 0, sizeof o0);
    hipLaunchKernelGGL(k_n2n<true>, dim3(st->N), dim3(WAVE), 0, (hipStream_t)stream, *cfg, *st, actions, e_cmd, reward, active, done, out ? *out : o0);
    return (int)hipGetLastError();
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
