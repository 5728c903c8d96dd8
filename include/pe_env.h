/*
 * pe_env.h -- C ABI of the MI355X (gfx950) batched pursuit-evasion environment.
 *
 * Drop-in boundary for the reference's environment object on the MAPPO hot path.  The reference has no
 * native FFI: the seam is the Python class `Pursuit_Env` (environment/pursuit_evasion_game/pursuit_env.py:56)
 * called by `MAPPO.run_episode` (DHGN/mappo_parallel.py:742-827) and `evaluate` (evaluator.py:106-201).  Each
 * entry point below replaces the method(s) cited next to it for N independent environments at once; the
 * ctypes binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions: every pointer in pe_state / pe_obs_out / pe_step_out is a DEVICE pointer owned by the caller
 * (e.g. a torch tensor); kernels are enqueued on the caller's stream (a hipStream_t passed as void*); functions
 * return 0 on success or a non-zero hipError_t / PE_ERR_* code, never throw, and keep no hidden state.
 * One 64-lane wavefront steps one environment; the environment's record is contiguous in HBM.
 */
#ifndef PE_ENV_H
#define PE_ENV_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PE_MAX_BEAMS 64
#define PE_MAX_P 16          /* defenders per environment */
#define PE_META_INTS 8
#define PE_RASER_ROW_WORDS(O) (((((O) + 31) >> 5) + 3) & ~3)   /* 32-bit words per raser row (16-byte aligned rows) */
#define PE_ERR_BAD_CONFIG 10001
#define PE_ERR_NULL 10002
#define PE_ERR_RESET_FAILED 10003 /* a placement loop of Pursuit_Env.reset hit PE_RESET_MAX_DRAWS (see below) */
/* The reference's reset uses unbounded rejection loops (base_env.py:52-70, 72-120, 122-162); on a map that admits no
 * placement they never return.  Host resetter, device reset and the oracle all stop after this many draws per loop, flag
 * the environment (PE_STATUS_RESET_FAILED / PE_ERR_RESET_FAILED) and keep the last candidate, so a bad configuration is an
 * error instead of a hung launch.  Loops that terminate in the reference are unaffected (they need < 1e3 draws). */
#define PE_RESET_MAX_DRAWS 100000

/* meta[env][k] */
enum { PE_META_T = 0, PE_META_PATH_LEN = 1, PE_META_TAPE_POS = 2, PE_META_COLLISION = 3, PE_META_PATH_CNT = 4,
       PE_META_ASTAR_EXP = 5, PE_META_STATUS = 6, PE_META_WP_HEAD = 7 /* waypoints popped since the last replan */ };
/* meta[PE_META_STATUS] bits */
enum { PE_STATUS_TAPE_EXHAUSTED = 1, PE_STATUS_ASTAR_CAP = 2, PE_STATUS_PATH_UNDERFLOW = 4, PE_STATUS_RESET_FAILED = 8 };
/* The bits are STICKY on the device-reset path: pe_env_reset carries the finished episode's bits into the new episode's meta
 * record, so one read of meta[:, PE_META_STATUS] after any reset (or at any later time) reports every condition raised since
 * pe_env_reset_seed.  pe_env_load (host initial conditions) starts from 0. */

/* Values of config.yaml (reference config.yaml:13-54) the kernels need.  Plain data, passed by value. */
typedef struct pe_config {
    int32_t W, H;             /* map.map_size */
    int32_t P;                /* env.num_defender (2..PE_MAX_P) */
    int32_t O;                /* map.num_max_obstacle: padded obstacle slots of o_adj (multiple of 4) */
    int32_t max_steps;        /* env.max_steps */
    int32_t difficulty;       /* env.difficulty: evader replans when time_step % difficulty == 0 */
    int32_t extend_dis;       /* attacker.extend_dis */
    int32_t num_beams;        /* sensor.num_beams (<= PE_MAX_BEAMS) */
    int32_t lidar_radius;     /* sensor.radius */
    int32_t evader_view;      /* attacker.sen_range */
    int32_t tape_len;         /* pre-drawn evader targets per environment and episode */
    int32_t max_path;         /* stored tail of the evader path (>= difficulty + 2) */
    int32_t use_reward_norm;  /* algo.use_reward_norm */
    int32_t pad0;
    double def_tau, def_dt, def_collision_radius, def_comm_range, def_sen_range;
    double eva_vmax, eva_tau, eva_dt, eva_collision_radius;
    double resolution;
    double action_u[9][2];    /* desired velocity per discrete action (agent.py:57-60), already times vmax */
    double beam_dir[PE_MAX_BEAMS][2]; /* LiDAR beam directions (pursuit_env.py:37-39) */
} pe_config;

/* Environment state in HBM.  [N] leading dimension everywhere; records are contiguous per environment. */
typedef struct pe_state {
    int32_t N;
    int32_t pad0;
    uint8_t *grid;      /* [N][W*H]      static occupancy, cell (x,y) at x*H+y (Occupied_Grid_Map.py:16)      */
    int16_t *bidx;      /* [N][W*H]      index of the boundary obstacle in that cell, or -1 (pursuit_env.py:18-27) */
    int32_t *n_obs;     /* [N]                                                                                  */
    double *def;        /* [N][4][P]     x[P], y[P], vx[P], vy[P]: struct-of-arrays over agents inside the record */
    double *eva;        /* [N][4]        x, y, vx, vy of the evader                                              */
    int32_t *target;    /* [N][2]                                                                                */
    int32_t *tape;      /* [N][tape_len][2]  targets the evader is assigned after reaching the current one      */
    int32_t *meta;      /* [N][PE_META_INTS]                                                                     */
    int16_t *path;      /* [N][max_path][2]  tail of the A* path; path[cnt-1] is the next waypoint               */
    double *rn;         /* [N][1+2P]     reward normaliser: n, mean[P], S[P] (DHGN/normalization.py:4-22)        */
    uint32_t *wpw;      /* [N][16]       the evader's next waypoints (x<<16|y) as of the last replan: path[cnt-1], path[cnt-2], ...;
                           lets the tick prefetch them with everything else instead of a read that depends on meta        */
    uint32_t *raser;    /* [N][W*H][RW]  the episode's raser map (pursuit_env.py:29-53 get_raser_map), bit-packed: bit k of row
                           x*H+y = boundary obstacle k is the first hit of a LiDAR beam from cell (x, y); RW = PE_RASER_ROW_WORDS(O).
                           Written by pe_env_reset / pe_env_load, read by the tick: o_adj[i] = raser[int(x_i)][int(y_i)]        */
} pe_state;

/* Observations, fp32, reference layouts (DHGN/mappo_parallel.py:767-771, replay_buffer.py:28-33).  Every tensor
 * has its own per-environment element stride so the kernel can write straight into replay-buffer slices
 * buffer[key][:, t].  A NULL pointer skips that output. */
typedef struct pe_obs_out {
    float *p_state; int64_t p_state_stride;  /* [N][P][4]  */
    float *e_state; int64_t e_state_stride;  /* [N][1][4]  */
    float *p_adj;   int64_t p_adj_stride;    /* [N][P][P]  communicate() incl. its column-1 quirk */
    float *e_adj;   int64_t e_adj_stride;    /* [N][P][1]  find_attacker()                        */
    float *o_adj;   int64_t o_adj_stride;    /* [N][P][O]  LiDAR row, zero padded to O            */
    /* the same LiDAR rows bit-packed (bit k of row i = o_adj[i][k]), PE_RASER_ROW_WORDS(O) words per row: 1/29 of the bytes.
     * The product's rollout / replay buffer / msg-agg kernels consume this form (MO_ADJ_BITS); o_adj stays the reference layout. */
    uint32_t *o_adj_bits; int64_t o_adj_bits_stride;   /* [N][P][RW] */
} pe_obs_out;

typedef struct pe_step_out {
    float *reward;     int64_t reward_stride;      /* [N][P] normalised reward (or raw if !use_reward_norm), fp32 */
    float *reward_raw; int64_t reward_raw_stride;  /* [N][P] raw reward (pursuit_env.py:128-149), may be NULL      */
    uint8_t *done;                                 /* [N]    time_step >= max_steps, may be NULL                   */
} pe_step_out;

/* Host-side initial condition of every environment (what Pursuit_Env.reset produces, pursuit_env.py:60-73).
 * HOST pointers; pe_env_load copies them to the device and derives bidx.  rn is left untouched (the reward
 * normaliser persists across episodes, DHGN/mappo_parallel.py:579-580) unless reset_rn != 0. */
typedef struct pe_host_init {
    const uint8_t *grid;      /* [N][W*H]          */
    const int32_t *obs_xy;    /* [N][O][2]  boundary obstacles in index order, first n_obs[n] valid */
    const int32_t *n_obs;     /* [N]               */
    const double *def;        /* [N][P][4]  x,y,vx,vy (array-of-structs, as get_state returns it)   */
    const double *eva;        /* [N][4]            */
    const int32_t *target;    /* [N][2]            */
    const int32_t *tape;      /* [N][tape_len][2]  */
    int32_t reset_rn;
    int32_t pad0;
} pe_host_init;

/* Map / placement parameters of Pursuit_Env.reset that are not needed on the device (config.yaml:30-36,
 * pursuit_env.py:71 min_dist=4). */
typedef struct pe_reset_params {
    int32_t num_blocks;       /* map.num_obstacle_block */
    int32_t min_dist;         /* defender spacing, 4 in the reference */
    double center[2];         /* map.center   */
    double variance;          /* map.variance */
    const uint8_t *fixed_grid; /* pe_env_reset only: DEVICE [W*H] u8 occupancy grid every environment starts from instead of drawing
                                 obstacle blocks -- a slot of a pre-generated map bank (the older reference driver hands one `map_info`
                                 per node and iteration to all its workers, MAPPO_parallel_main.py:103-124); the map-generation draws
                                 are not taken, target / defender / attacker placement is unchanged.  NULL = draw the map. */
} pe_reset_params;

/* HOST arrays the resetter fills; same shapes as pe_host_init. */
typedef struct pe_host_init_out {
    uint8_t *grid; int32_t *obs_xy; int32_t *n_obs; double *def; double *eva; int32_t *target; int32_t *tape;
} pe_host_init_out;

/* Host side of Pursuit_Env.reset() (pursuit_env.py:60-73, base_env.py:37-162, Occupied_Grid_Map.py:46-62) for N
 * environments.  Environment n owns bit-exact re-implementations of the two generator streams the reference draws
 * from (Python `random`, numpy legacy RandomState), seeded like random.seed(seeds[n]); np.random.seed(seeds[n]).
 * consumed_targets[n] (may be NULL on the first call) = how many tape targets the evader used in the episode that
 * just ended; the un-used draws are returned to the stream before the next reset. */
void *pe_resetter_create(const pe_config *cfg, const pe_reset_params *prm, int32_t N, const uint64_t *seeds);
void pe_resetter_destroy(void *resetter);
int pe_resetter_reset(void *resetter, const int32_t *consumed_targets, const pe_host_init_out *out, int32_t n_threads);
/* Opaque snapshot of every environment's generator streams (for a resume bundle; the reference cannot resume at all). */
int64_t pe_resetter_state_bytes(void *resetter);
int pe_resetter_get_state(void *resetter, void *out);
int pe_resetter_set_state(void *resetter, const void *in);

/* Pursuit_Env.reset() on the DEVICE (pursuit_env.py:60-73, base_env.py:37-162, Occupied_Grid_Map.py:46-62,
 * 119-166): the same streams and draws as the host resetter above, one wavefront per environment, no host arrays and no
 * upload.  reset_state: device memory of pe_reset_state_bytes(cfg, N) bytes that holds every environment's two
 * generator states and the inflated map of its running episode between calls (opaque; copy it to save / restore).
 *   pe_env_reset_seed: random.seed(seeds[n]); np.random.seed(seeds[n]) for every environment (seeds: HOST array).
 *   pe_env_reset: next episode.  first != 0 on the first reset after seeding; afterwards the number of tape targets the
 *   finished episode consumed is read from state->meta and the unused draws go back to the stream.  o_state (may be
 *   NULL): [N][O][4] fp32 rows [x, y, 0, 0] of the boundary obstacles, zero padded (boundary_map.obstacle_agent).
 *   state->n_obs[n] may exceed cfg->O (the caller must check, as pe_env_load's caller does). */
int64_t pe_reset_state_bytes(const pe_config *cfg, int32_t N);
int pe_env_reset_seed(const pe_config *cfg, int32_t N, const uint64_t *seeds, void *reset_state, void *stream);
int pe_env_reset(const pe_config *cfg, const pe_state *st, const pe_reset_params *prm, void *reset_state, int32_t first, float *o_state,
                 int32_t reset_rn, void *stream);

/* Validates a configuration against the kernels' limits. */
int pe_config_check(const pe_config *cfg);
/* Bytes of dynamic LDS one workgroup of the fused tick uses (for occupancy reports). */
int64_t pe_tick_lds_bytes(const pe_config *cfg, int32_t with_replan);

/* Pursuit_Env.reset() hand-over: host initial conditions -> device records (pursuit_env.py:60-73). */
int pe_env_load(const pe_config *cfg, const pe_state *st, const pe_host_init *init, void *stream);

/* get_state + communicate + sensor (base_env.py:198-209, pursuit_env.py:182-209, agent.py:157-169, 319-341); the LiDAR
 * rows come from the episode's raser table (pursuit_env.py:29-53, built on the device by pe_env_reset / pe_env_load). */
int pe_env_observe(const pe_config *cfg, const pe_state *st, const pe_obs_out *out, void *stream);

/* Pursuit_Env.attacker_step (pursuit_env.py:75-102) incl. Evader.replan / rescan / A* (agent.py:202-271,
 * astar.py:26-161).  may_replan == 0 promises that no environment has time_step % difficulty == 0. */
int pe_evader_step(const pe_config *cfg, const pe_state *st, int32_t may_replan, void *stream);

/* Pursuit_Env.step (pursuit_env.py:104-149) + reward normalisation (DHGN/normalization.py:29-35). */
int pe_env_step(const pe_config *cfg, const pe_state *st, const int32_t *actions, const pe_step_out *out, void *stream);

/* Pursuit_Env.demon (pursuit_env.py:211-229): the scripted pursuit policy, one discrete action per defender -- the action whose
 * direction is closest to the bearing of the evader ((0, 0) within 0.01 of it).  unit_dirs: HOST [9][2] doubles, the reference's
 * `actions_mat` (cos, sin of k pi / 4, then (0, 0)); actions: DEVICE [N][P] int32. */
int pe_env_demon(const pe_config *cfg, const pe_state *st, const double *unit_dirs, int32_t *actions, void *stream);

/* Fused rollout tick: step(actions) -> observe -> attacker_step in ONE launch (the order of
 * DHGN/mappo_parallel.py:793 followed by :759-765 of the next loop iteration). */
int pe_env_tick(const pe_config *cfg, const pe_state *st, const int32_t *actions, const pe_step_out *sout,
                const pe_obs_out *oout, int32_t may_replan, void *stream);

/* The first two phases of the tick only (step -> observe).  Used on replan ticks so that the evader's rescan + A*
 * (pe_evader_step on a second stream) overlaps the policy forward of the next step, which needs the observations but
 * not the evader's new position. */
int pe_env_step_observe(const pe_config *cfg, const pe_state *st, const int32_t *actions, const pe_step_out *sout,
                        const pe_obs_out *oout, void *stream);

/* Weighted A* of the evader on one standalone problem per workgroup (astar.py:26-161); test/diagnostic entry.
 * obs: [n][(W+1)*(H+1)] device bytes, sg: [n][4] (sx,sy,gx,gy), out_path: [n][max_path][2], out_len: [n][2]
 * (true length, expansions). */
int pe_astar_batch(int32_t W, int32_t H, int32_t n, const uint8_t *obs, const int32_t *sg, int16_t *out_path,
                   int32_t *out_len, int32_t max_path, void *stream);

const char *pe_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif
