/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's env_3d step (continuous 3-D pursuit).
 * Follows environment/env_3d/particle_env.py: Point.step :25-55 (continuous action a in [-1, 1]^3: heading, pitch, speed, each
 * rate-limited; the position moves with the NEW pitch / speed and the COMMANDED heading), ParticleEnv.step :205-219,
 * get_done :221-241, reward / agent_reward :267-284, update_agent_active :286-326, get_adj_mat :328-340,
 * collision_detection :342-352.  The evader's command (eva.e_f, scipy SLSQP, eva.py:87-148) is an INPUT (parity of the
 * minimiser unpinned).  Parity status: PINNED by tests/golden/e3d_*.npz captured from the reference.
 * cos/sin are libm's (== numpy scalar cos/sin in the build container); np.linalg.norm of a 3-vector ==
 * sqrt(fma(c, c, fma(b, b, a*a))) (measured in the build container on 200 000 random vectors, 0 mismatches).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846 /* == numpy.pi */
#endif

typedef struct e3d_cfg {
    int32_t P, max_step;
    double p_vmax, e_vmax, p_sen_range, p_comm_range, kill_radius, ang_lmt, v_lmt, step_size;
} e3d_cfg;

static inline double norm3(double a, double b, double c) { return sqrt(fma(c, c, fma(b, b, a * a))); }
static inline double sgn(double v) { return (double)((v > 0) - (v < 0)); }
static inline double clip(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* state row: x, y, z, phi, gamma, v, active */
static void point_step(double *s, const double *a, double v_max, double ang, double vlmt, double h) {
    if (s[6] == 0.0) return;
    const double phi = a[0] * M_PI, gamma = a[1] * M_PI / 2, v = (a[2] + 1) / 2 * v_max;
    s[4] += clip(gamma - s[4], -ang, ang);
    s[5] += clip(v - s[5], -vlmt, vlmt);
    double dphi;
    if (sgn(phi * s[3]) >= 0) dphi = clip(phi - s[3], -ang, ang);
    else if (fabs(phi - s[3]) < 2 * M_PI - fabs(phi - s[3])) dphi = clip(phi - s[3], -ang, ang);
    else dphi = clip(2 * M_PI - fabs(phi - s[3]), 0, ang) * -sgn(phi - s[3]);
    s[3] += dphi;
    if (s[3] > M_PI) s[3] -= 2 * M_PI; else if (s[3] < -M_PI) s[3] += 2 * M_PI;
    s[0] += s[5] * cos(s[4]) * cos(phi) * h;
    s[1] += s[5] * cos(s[4]) * sin(phi) * h;
    s[2] += s[5] * sin(s[4]) * h;
}

void e3d_evader_step(const e3d_cfg *c, double *e, const double *cmd) { point_step(e, cmd, c->e_vmax, c->ang_lmt, c->v_lmt, c->step_size); }

/* returns done; reward[P], active_out[P] */
int e3d_step(const e3d_cfg *c, double *p, double *e, const double *target, const double *action, int32_t *time_step, double *reward,
             uint8_t *active_out) {
    const int P = c->P;
    *time_step += 1;
    for (int i = 0; i < P; i++) point_step(p + 7 * i, action + 3 * i, c->p_vmax, c->ang_lmt, c->v_lmt, c->step_size);
    uint8_t pdie[64];
    int near_e = 0;
    for (int i = 0; i < P; i++) {
        const double *s = p + 7 * i;
        reward[i] = 0.0; pdie[i] = 0;
        if (s[6] == 0.0) continue;
        int ce = 0, cp = 0;
        if (e[6] != 0.0 && norm3(s[0] - e[0], s[1] - e[1], s[2] - e[2]) <= c->kill_radius) ce++;
        for (int k = 0; k < P; k++) if (p[7 * k + 6] != 0.0 && norm3(s[0] - p[7 * k], s[1] - p[7 * k + 1], s[2] - p[7 * k + 2]) <= c->kill_radius) cp++;
        reward[i] = (double)ce - (double)(cp - 1);
        pdie[i] = (cp + ce - 1) != 0;
        if (e[6] != 0.0 && norm3(e[0] - s[0], e[1] - s[1], e[2] - s[2]) <= c->kill_radius) near_e++;
    }
    const int edie = e[6] != 0.0 && near_e != 0;   /* one evader: its own inner collision is itself (sum - 1 == pursuers in reach) */
    for (int i = 0; i < P; i++) if (pdie[i]) { double *s = p + 7 * i; s[0] = s[1] = s[2] = 1000; s[3] = s[4] = s[5] = 0; s[6] = 0; }
    if (edie) { e[0] = e[1] = e[2] = 1000; e[3] = e[4] = e[5] = 0; e[6] = 0; }
    int pa = 0;
    for (int i = 0; i < P; i++) { active_out[i] = p[7 * i + 6] != 0.0; pa += active_out[i]; }
    const int reached = norm3(e[0] - target[0], e[1] - target[1], e[2] - target[2]) <= c->kill_radius;
    return reached || pa == 0 || e[6] == 0.0 || *time_step >= c->max_step;
}

void e3d_observe(const e3d_cfg *c, const double *p, const double *e, float *p_state, float *e_state, float *pp_adj, float *pe_adj) {
    const int P = c->P;
    for (int i = 0; i < P; i++) for (int k = 0; k < 6; k++) p_state[6 * i + k] = (float)p[7 * i + k];
    for (int k = 0; k < 6; k++) e_state[k] = (float)e[k];
    for (int i = 0; i < P; i++) {
        const double *s = p + 7 * i;
        for (int j = 0; j < P; j++)
            pp_adj[i * P + j] = (s[6] != 0.0 && norm3(s[0] - p[7 * j], s[1] - p[7 * j + 1], s[2] - p[7 * j + 2]) <= c->p_comm_range) ? 1.f : 0.f;
        pe_adj[i] = (s[6] != 0.0 && norm3(s[0] - e[0], s[1] - e[1], s[2] - e[2]) <= c->p_sen_range) ? 1.f : 0.f;
    }
}
