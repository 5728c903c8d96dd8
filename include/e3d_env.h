/*
 * e3d_env.h -- C ABI of the MI355X (gfx950) batched env_3d environment (continuous 3-D pursuit; SURVEY 8f row 4, BASELINE
 * config 5).
 *
 * Replaces, for N independent environments, the methods of the reference class
 * environment/env_3d/particle_env.py:76 `ParticleEnv` cited per entry point.  Pursuer actions are CONTINUOUS
 * (a in [-1, 1]^3: heading, pitch, speed; Point.step :25-55).  The evader's command -- in the reference the result of eva.e_f
 * (scipy SLSQP, eva.py:87-148) -- is an INPUT.  Conventions as in pe_env.h: device pointers owned by the caller, caller's
 * hipStream_t as void*, 0 == success.
 * Several environments share one 64-lane wavefront (lane = (environment, pursuer), 8 environments per wave for P <= 8).
 */
#ifndef E3D_ENV_H
#define E3D_ENV_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E3D_MAX_P 64
#define E3D_ERR_BAD_CONFIG 40001
#define E3D_ERR_NULL 40002
#define E3D_ERR_RESET_FAILED 40003   /* gen_init_p_pos (:151-164) found no placement within E3D_RESET_MAX_DRAWS draws */
#define E3D_RESET_MAX_DRAWS 100000

typedef struct e3d_config {      /* particle_env.py:78-121 */
    int32_t P, max_step;
    double p_vmax, e_vmax, p_sen_range, p_comm_range, kill_radius, ang_lmt, v_lmt, step_size;
} e3d_config;

typedef struct e3d_state {
    int32_t N, pad0;
    double *p;           /* [N][7][P]  x[P], y[P], z[P], phi[P], gamma[P], v[P], active[P] (SoA over agents inside the record) */
    double *e;           /* [N][7]     the evader (e_num == 1, initialize :134-135)                                        */
    double *target;      /* [N][3]                                                                                         */
    int32_t *time_step;  /* [N]                                                                                            */
} e3d_state;

typedef struct e3d_obs_out {     /* fp32, NULL skips; *_stride = elements between environments */
    float *p_state; int64_t p_state_stride;   /* [N][P][6]  get_team_state(True, rules=False)  (:247-265)                  */
    float *e_state; int64_t e_state_stride;   /* [N][1][6]                                                                 */
    float *pp_adj;  int64_t pp_adj_stride;    /* [N][P][P]  get_adj_mat(p, p, p_comm_range)  (:328-340)                    */
    float *pe_adj;  int64_t pe_adj_stride;    /* [N][P][1]  get_adj_mat(p, e, p_sen_range)                                 */
} e3d_obs_out;

int e3d_config_check(const e3d_config *cfg);
/* ParticleEnv.reset hand-over: host p [N][P][7], e [N][7] (x, y, z, phi, gamma, v, active), target [N][3] -> device records */
int e3d_env_load(const e3d_config *cfg, const e3d_state *st, const double *p, const double *e, const double *target, void *stream);
int e3d_env_observe(const e3d_config *cfg, const e3d_state *st, const e3d_obs_out *out, void *stream);
/* One fused tick: Point.step of the evader with the command e_cmd [N][3] in [-1, 1] (evader_step :354-378; skipped, as there,
 * when the evader is inactive or no pursuer is left) -> ParticleEnv.step(actions [N][P][3], f64) (:205-219: Point.step,
 * reward :267-284, update_agent_active :286-326, get_done :221-241) -> observations of the new state.
 * reward [N][P] fp32, active [N][P] u8, done [N] u8. */
int e3d_env_tick(const e3d_config *cfg, const e3d_state *st, const double *actions, const double *e_cmd, float *reward,
                 uint8_t *active, uint8_t *done, const e3d_obs_out *out, void *stream);

/* Host side of ParticleEnv.reset (:137-203) with a bit-exact replica of numpy's legacy RandomState per environment
 * (np.random.seed(seeds[n])).  Fills host arrays p [N][P][7], e [N][7], target [N][3]. */
void *e3d_resetter_create(const e3d_config *cfg, int32_t N, const uint32_t *seeds);
void e3d_resetter_destroy(void *resetter);
int e3d_resetter_reset(void *resetter, double *p, double *e, double *target, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif
