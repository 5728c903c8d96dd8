/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * Plain-C (gcc, -ffp-contract=off) CPU restatement of the reference's pursuit-evasion environment
 * tick.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Each function cites the reference file:line it follows (paths relative to the reference root).
 *
 * Parity status: PINNED by golden vectors captured from the reference itself in the build container
 * (tests/golden/env_trace_*.npz, astar_cases.npz, raser_*.npz; generator tests/golden/gen/make_goldens.py).
 * Exception: the evader heading uses cos(acos(c)) == c and sin(acos(c)) == sqrt((1-c)(1+c)) instead of libm
 * acos/cos/sin (numpy's arccos is not reproducible across CPUs either); evader f64 state therefore agrees
 * with the reference to <= 1e-9 absolute over an episode, every discrete output is identical on the goldens.
 */
#ifndef PE_ENV_ORACLE_H
#define PE_ENV_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PEO_MAX_BEAMS 64
#define PEO_MAX_PATH 4096

typedef struct peo_config {
    int32_t W, H;             /* map.map_size                       (config.yaml:32) */
    int32_t P;                /* env.num_defender                   (config.yaml:23) */
    int32_t O;                /* map.num_max_obstacle (padding)     (config.yaml:36) */
    int32_t max_steps;        /* env.max_steps                      (config.yaml:21) */
    int32_t difficulty;       /* env.difficulty                     (config.yaml:26) */
    int32_t extend_dis;       /* attacker.extend_dis                (config.yaml:46) */
    int32_t num_beams;        /* sensor.num_beams                   (config.yaml:28) */
    int32_t lidar_radius;     /* sensor.radius                      (config.yaml:29) */
    int32_t evader_view;      /* attacker.sen_range                 (config.yaml:42) */
    int32_t tape_len;         /* pre-drawn targets per env ("target tape") */
    int32_t pad0;
    double def_tau, def_dt, def_collision_radius, def_comm_range, def_sen_range;
    double eva_vmax, eva_tau, eva_dt, eva_collision_radius;
    double resolution;        /* map.resolution */
    double action_u[9][2];    /* agent.py:57-60 : vmax*(cos,sin)(k*pi/4), k<8 ; (0,0) */
    double beam_dir[PEO_MAX_BEAMS][2]; /* pursuit_env.py:37-39 */
} peo_config;

typedef struct peo_env {
    uint8_t *grid;            /* [W*H]  static occupancy, index x*H+y  (Occupied_Grid_Map.py:16) */
    int16_t *bidx;            /* [W*H]  boundary-obstacle index or -1  (pursuit_env.py:18-27)   */
    int32_t n_obs;
    double *def_state;        /* [P*4]  x,y,vx,vy per defender */
    double eva[4];
    int32_t target[2];
    int32_t *tape;            /* [tape_len*2] subsequent targets */
    int32_t tape_pos;
    int32_t t;                /* time_step */
    int32_t collision;        /* env.collision flag (pursuit_env.py:140) */
    int32_t path_len;
    int16_t path[PEO_MAX_PATH][2];  /* goal ... next waypoint (astar.py:130-146) */
    /* reward normaliser (DHGN/normalization.py:4-35), per defender */
    int64_t rn_n;
    double *rn_mean, *rn_S;   /* [P] */
    /* diagnostics */
    int32_t astar_expansions;
} peo_env;

/* astar.py:26-161.  obs: (W+1)*(H+1) bytes indexed x*(H+1)+y (cells with x==W or y==H are never obstacles).
 * Returns path length, path goal->start in out_path (capacity PEO_MAX_PATH). */
int peo_astar(int W, int H, const uint8_t *obs, int sx, int sy, int gx, int gy, int16_t (*out_path)[2],
              int *n_expanded);

/* agent.py:232-259 + :202-230 */
void peo_replan(const peo_config *c, peo_env *e);
/* pursuit_env.py:75-102 */
void peo_evader_step(const peo_config *c, peo_env *e);
/* pursuit_env.py:182-195, :197-209 (+ :29-53 on the fly), agent.py:157-169 ; outputs fp32 like
 * DHGN/mappo_parallel.py:767-771.  o_adj is [P*O] zero-padded (replay_buffer.py:52). */
void peo_observe(const peo_config *c, const peo_env *e, float *p_state, float *e_state, float *p_adj,
                 float *e_adj, float *o_adj);
/* pursuit_env.py:104-149 ; reward[P] raw (f64), can_apply[P]; returns done */
int peo_step(const peo_config *c, peo_env *e, const int32_t *actions, double *reward, uint8_t *can_apply);
/* DHGN/normalization.py:29-35 applied to a raw reward vector, in place semantic: out fp64 */
void peo_reward_norm(const peo_config *c, peo_env *e, const double *reward, double *out);
/* pursuit_env.py:29-53 for one cell: flags[n_obs] */
void peo_lidar_cell(const peo_config *c, const peo_env *e, int cx, int cy, uint8_t *flags);

/* handle API for ctypes */
peo_env *peo_create(const peo_config *c);
void peo_destroy(peo_env *e);
/* host-injected initial condition (the reset is host logic: oracle/reset_oracle.py) */
void peo_load(const peo_config *c, peo_env *e, const uint8_t *grid, const int32_t *obs_xy, int32_t n_obs,
              const double *def_state, const double *eva, const int32_t *target, const int32_t *tape);
void peo_get(const peo_config *c, const peo_env *e, double *def_state, double *eva, int32_t *target,
             int32_t *scalars /* t, path_len, tape_pos, collision, astar_expansions */);
void peo_get_path(const peo_env *e, int16_t *out /* [path_len*2] */);
void peo_get_rn(const peo_config *c, const peo_env *e, double *n_mean_S /* [1+2P] */);

/* Batched convenience used by the cpu_baseline: one full tick (observe -> evader -> step) for n envs
 * with the given actions [n*P]; returns a checksum so the work cannot be optimised away. */
double peo_tick_batch(const peo_config *c, peo_env **envs, int n, const int32_t *actions, float *scratch);

void peo_prims(int n, const double *a, const double *b, double *out);

#ifdef __cplusplus
}
#endif
#endif
