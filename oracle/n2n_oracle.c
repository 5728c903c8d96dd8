/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's env_n2n step (continuous 2-D pursuit).
 * Follows environment/env_n2n/particle_env.py: Pursuer.step :34-67, Evader.step :74-99, ParticleEnv.step :164-177,
 * reward/agent_reward :316-334, update_agent_active :336-365, get_done :283-304, get_adj_mat :386-397,
 * collision_detection :399-409.  The evader's heading command (eva.e_f, scipy SLSQP) is an INPUT (parity of the
 * minimiser unpinned).  Parity status: PINNED by tests/golden/n2n_*.npz captured from the reference.
 * cos/sin are libm's (== numpy scalar cos/sin in the build container), norm == sqrt(fma(b,b,a*a)).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846 /* == numpy.pi */
#endif

typedef struct n2n_cfg {
    int32_t P, E, episode_limit, pad0;
    double p_vmax, e_vmax, p_sen_range, p_comm_range, kill_radius, ang_lmt, step_size;
} n2n_cfg;

static inline double norm2(double a, double b) { return sqrt(fma(b, b, a * a)); }
static inline double sgn(double v) { return (v > 0) - (v < 0); }

/* shared heading update: returns sign * clipped delta for a commanded heading a (radians) */
static double turn(double a, double phi, double lim) {
    double sign, delta;
    if (sgn(a * phi) >= 0) { delta = fabs(a - phi); sign = sgn(a - phi); }
    else if (fabs(a - phi) < 2 * M_PI - fabs(a - phi)) { delta = fabs(a - phi); sign = sgn(a - phi); }
    else { delta = 2 * M_PI - fabs(a - phi); sign = -sgn(a - phi); }
    if (delta > lim) delta = lim;
    if (delta < 0) delta = 0;
    return sign * delta;
}
static double wrap(double phi) { if (phi > M_PI) phi -= 2 * M_PI; else if (phi < -M_PI) phi += 2 * M_PI; return phi; }

/* state rows: x, y, phi, v, active */
void n2n_evader_step(const n2n_cfg *c, double *e, const double *cmd) {
    for (int i = 0; i < c->E; i++) {
        double *s = e + 5 * i;
        if (s[4] == 0.0) continue;                      /* evader_step only moves active evaders (:180-198) */
        double a = cmd[i] * M_PI;
        double d = turn(a, s[2], c->ang_lmt);
        s[0] += s[3] * cos(s[2]) * c->step_size;          /* position with the OLD heading (:93-95) */
        s[1] += s[3] * sin(s[2]) * c->step_size;
        s[2] = wrap(s[2] + d);
    }
}

/* returns done; reward[P], active_out[P] */
int n2n_step(const n2n_cfg *c, double *p, double *e, const double *target, const int32_t *action, int32_t *time_step, double *reward,
             uint8_t *active_out) {
    const int P = c->P, E = c->E;
    *time_step += 1;
    for (int i = 0; i < P; i++) {
        double *s = p + 5 * i;
        double v = 0.0;
        if (action[i] != 0) {
            v = c->p_vmax;
            double a = (double)action[i] * M_PI / 4;
            if (a > M_PI) a -= 2 * M_PI;
            s[2] = wrap(s[2] + turn(a, s[2], c->ang_lmt));  /* heading changes even for an inactive pursuer */
        }
        if (s[4] != 0.0) { s[0] += v * cos(s[2]) * c->step_size; s[1] += v * sin(s[2]) * c->step_size; s[3] = v; }
    }
    for (int i = 0; i < P; i++) {
        double r = 0.0;
        if (p[5 * i + 4] != 0.0) {
            int ce = 0, cp = 0;
            for (int k = 0; k < E; k++) if (e[5 * k + 4] != 0.0 && norm2(p[5 * i] - e[5 * k], p[5 * i + 1] - e[5 * k + 1]) <= c->kill_radius) ce++;
            for (int k = 0; k < P; k++) if (p[5 * k + 4] != 0.0 && norm2(p[5 * i] - p[5 * k], p[5 * i + 1] - p[5 * k + 1]) <= c->kill_radius) cp++;
            r = (double)ce - (double)(cp - 1);
        }
        reward[i] = r;
    }
    uint8_t pdie[64], edie[16];
    for (int i = 0; i < P; i++) {
        pdie[i] = 0;
        if (p[5 * i + 4] == 0.0) continue;
        int cnt = 0;
        for (int k = 0; k < P; k++) if (p[5 * k + 4] != 0.0 && norm2(p[5 * i] - p[5 * k], p[5 * i + 1] - p[5 * k + 1]) <= c->kill_radius) cnt++;
        for (int k = 0; k < E; k++) if (e[5 * k + 4] != 0.0 && norm2(p[5 * i] - e[5 * k], p[5 * i + 1] - e[5 * k + 1]) <= c->kill_radius) cnt++;
        pdie[i] = (cnt - 1) != 0;
    }
    for (int k = 0; k < E; k++) {
        edie[k] = 0;
        if (e[5 * k + 4] == 0.0) continue;
        int cnt = 0;
        for (int i = 0; i < P; i++) if (p[5 * i + 4] != 0.0 && norm2(e[5 * k] - p[5 * i], e[5 * k + 1] - p[5 * i + 1]) <= c->kill_radius) cnt++;
        edie[k] = cnt != 0;
    }
    for (int i = 0; i < P; i++) if (pdie[i]) { p[5 * i] = 1000; p[5 * i + 1] = 1000; p[5 * i + 2] = 0; p[5 * i + 4] = 0; }
    for (int k = 0; k < E; k++) if (edie[k]) { e[5 * k] = 1000; e[5 * k + 1] = 1000; e[5 * k + 2] = 0; e[5 * k + 4] = 0; }
    int pa = 0, ea = 0, reached = 0;
    for (int i = 0; i < P; i++) { active_out[i] = p[5 * i + 4] != 0.0; pa += active_out[i]; }
    for (int k = 0; k < E; k++) {
        ea += e[5 * k + 4] != 0.0;
        if (norm2(e[5 * k] - target[0], e[5 * k + 1] - target[1]) <= c->kill_radius) reached = 1;
    }
    return reached || pa == 0 || ea == 0 || *time_step >= c->episode_limit;
}

/* get_adj_mat for (pursuers -> pursuers, comm range) and (pursuers -> evaders, sensing range); rows of inactive pursuers are zero */
void n2n_observe(const n2n_cfg *c, const double *p, const double *e, float *p_state, float *e_state, float *pp_adj, float *pe_adj) {
    const int P = c->P, E = c->E;
    for (int i = 0; i < P; i++) for (int k = 0; k < 3; k++) p_state[3 * i + k] = (float)p[5 * i + k];
    for (int i = 0; i < E; i++) for (int k = 0; k < 3; k++) e_state[3 * i + k] = (float)e[5 * i + k];
    for (int i = 0; i < P; i++) {
        for (int j = 0; j < P; j++) pp_adj[i * P + j] = (p[5 * i + 4] != 0.0 && norm2(p[5 * i] - p[5 * j], p[5 * i + 1] - p[5 * j + 1]) <= c->p_comm_range) ? 1.f : 0.f;
        for (int j = 0; j < E; j++) pe_adj[i * E + j] = (p[5 * i + 4] != 0.0 && norm2(p[5 * i] - e[5 * j], p[5 * i + 1] - e[5 * j + 1]) <= c->p_sen_range) ? 1.f : 0.f;
    }
}
