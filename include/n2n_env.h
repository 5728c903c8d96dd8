/*
 * n2n_env.h -- C ABI of the MI355X (gfx950) batched env_n2n environment (continuous 2-D pursuit).
 *
 * Replaces, for N independent environments, the methods of the reference class
 * environment/env_n2n/particle_env.py:105 `ParticleEnv` cited per entry point.  The evader's heading command -- in the
 * reference the result of eva.e_f (scipy SLSQP, eva.py:36-53) -- is an INPUT here.  Conventions as in pe_env.h:
 * device pointers owned by the caller, caller's hipStream_t as void*, 0 == success.
 * Several environments share one 64-lane wavefront (lane = (environment, agent slot); 4 environments per wave at P = 16).
 */
#ifndef N2N_ENV_H
#define N2N_ENV_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define N2N_MAX_P 64
#define N2N_MAX_E 8
#define N2N_ERR_BAD_CONFIG 30001
#define N2N_ERR_NULL 30002

typedef struct n2n_config {      /* particle_env.py:108-121,147 */
    int32_t P, E, episode_limit, pad0;
    double p_vmax, e_vmax, p_sen_range, p_comm_range, kill_radius, ang_lmt, step_size;
} n2n_config;

typedef struct n2n_state {
    int32_t N, pad0;
    double *p;           /* [N][5][P]  x[P], y[P], phi[P], v[P], active[P]  (SoA over agents inside the record) */
    double *e;           /* [N][5][E]                                                                          */
    double *target;      /* [N][2]                                                                             */
    int32_t *time_step;  /* [N]                                                                                */
} n2n_state;

typedef struct n2n_obs_out {     /* fp32, NULL skips; *_stride = elements between environments */
    float *p_state; int64_t p_state_stride;   /* [N][P][3]  get_team_state(True, rules=False)  (:367-384)          */
    float *e_state; int64_t e_state_stride;   /* [N][E][3]                                                         */
    float *pp_adj;  int64_t pp_adj_stride;    /* [N][P][P]  get_adj_mat(p, p, p_comm_range)  (:386-397)            */
    float *pe_adj;  int64_t pe_adj_stride;    /* [N][P][E]  get_adj_mat(p, e, p_sen_range)                         */
} n2n_obs_out;

int n2n_config_check(const n2n_config *cfg);
/* ParticleEnv.reset hand-over: host p [N][P][5], e [N][E][5] (x, y, phi, v, active), target [N][2] -> device records */
int n2n_env_load(const n2n_config *cfg, const n2n_state *st, const double *p, const double *e, const double *target, void *stream);
int n2n_env_observe(const n2n_config *cfg, const n2n_state *st, const n2n_obs_out *out, void *stream);
/* One fused tick: Evader.step with the commanded heading e_cmd [N][E] in [-1, 1] (:74-99, driven by evader_step :179-198)
 * -> ParticleEnv.step(actions [N][P]) (:164-177: Pursuer.step :34-67, reward :316-334, update_agent_active :336-365,
 * get_done :283-304) -> observations of the new state.  reward [N][P] fp32, active [N][P] u8, done [N] u8. */
int n2n_env_tick(const n2n_config *cfg, const n2n_state *st, const int32_t *actions, const double *e_cmd, float *reward,
                 uint8_t *active, uint8_t *done, const n2n_obs_out *out, void *stream);

/* Host side of ParticleEnv.reset (:200-281) with a bit-exact replica of numpy's legacy RandomState per environment
 * (np.random.seed(seeds[n])).  Fills host arrays p [N][P][5], e [N][E][5], target [N][2]. */
void *n2n_resetter_create(const n2n_config *cfg, int32_t N, const uint32_t *seeds);
void n2n_resetter_destroy(void *resetter);
int n2n_resetter_reset(void *resetter, double *p, double *e, double *target, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif
