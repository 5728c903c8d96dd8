/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see pe_env_oracle.h).  CPU restatement, serial, one env at a time.
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).  No FMA contraction anywhere except
 * the explicit fma() at the np.linalg.norm sites (norm([a,b]) == sqrt(fma(b,b,a*a))).
 */
#include "pe_env_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Python round(): half-to-even, result int (Occupied_Grid_Map.py:65-69, :79-85) */
static inline int py_round(double v) { return (int)nearbyint(v); }
/* np.linalg.norm of a 2-vector as numpy computes it here */
static inline double norm2(double a, double b) { return sqrt(fma(b, b, a * a)); }

static inline int in_bound_i(const peo_config *c, int x, int y) { return x < c->W && x >= 0 && y < c->H && y >= 0; }
/* Occupied_Grid_Map.py:102-104 */
static inline int in_bound_f(const peo_config *c, double x, double y) { return in_bound_i(c, py_round(x), py_round(y)); }

/* agent.py:74-104, association exactly as written; theta is not observable and is omitted */
static void dynamic(double tau, double h, const double *s, double ux, double uy, double *out) {
    double vx0 = s[2], vy0 = s[3];
    double k1 = (ux - vx0) / tau;
    double k2 = (ux - (vx0 + h * k1 / 2)) / tau;
    double k3 = (ux - (vx0 + h * k2 / 2)) / tau;
    double k4 = (ux - (vx0 + h * k3)) / tau;
    double vx = vx0 + (k1 + 2 * k2 + 2 * k3 + k4) * h / 6;
    k1 = (uy - vy0) / tau;
    k2 = (uy - (vy0 + h * k1 / 2)) / tau;
    k3 = (uy - (vy0 + h * k2 / 2)) / tau;
    k4 = (uy - (vy0 + h * k3)) / tau;
    double vy = vy0 + (k1 + 2 * k2 + 2 * k3 + k4) * h / 6;
    out[0] = s[0] + vx * h;
    out[1] = s[1] + vy * h;
    out[2] = vx;
    out[3] = vy;
}

/* ------------------------------------------------------------------ A* (astar.py:26-161) */
typedef struct { double f; int x, y; } heap_item;

static inline int item_less(const heap_item *a, const heap_item *b) {
    /* Python tuple order of (f, (x, y)) */
    if (a->f != b->f) return a->f < b->f;
    if (a->x != b->x) return a->x < b->x;
    return a->y < b->y;
}

typedef struct { heap_item *v; int n, cap; } heap_t;

static void heap_push(heap_t *h, heap_item it) {
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 1024; h->v = (heap_item *)realloc(h->v, sizeof(heap_item) * h->cap); }
    int i = h->n++;
    h->v[i] = it;
    while (i > 0) {
        int p = (i - 1) >> 1;
        if (item_less(&h->v[i], &h->v[p])) { heap_item t = h->v[i]; h->v[i] = h->v[p]; h->v[p] = t; i = p; } else break;
    }
}
static heap_item heap_pop(heap_t *h) {
    heap_item top = h->v[0];
    h->v[0] = h->v[--h->n];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < h->n && item_less(&h->v[l], &h->v[m])) m = l;
        if (r < h->n && item_less(&h->v[r], &h->v[m])) m = r;
        if (m == i) break;
        heap_item t = h->v[i]; h->v[i] = h->v[m]; h->v[m] = t; i = m;
    }
    return top;
}

static const int U_SET[8][2] = {{-1, 0}, {-1, 1}, {0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1}}; /* astar.py:11-12 */

int peo_astar(int W, int H, const uint8_t *obs, int sx, int sy, int gx, int gy, int16_t (*out_path)[2], int *n_expanded) {
    const int SX = W + 1, SY = H + 1, NN = SX * SY;
    if (n_expanded) *n_expanded = 0;
    /* astar.py:46-47 : goal in obs -> [s_start] */
    if (gx >= 0 && gx <= W && gy >= 0 && gy <= H && obs[gx * SY + gy]) { out_path[0][0] = (int16_t)sx; out_path[0][1] = (int16_t)sy; return 1; }
    double *g = (double *)malloc(sizeof(double) * NN);
    int32_t *parent = (int32_t *)malloc(sizeof(int32_t) * NN);
    for (int i = 0; i < NN; i++) { g[i] = INFINITY; parent[i] = -1; }
    heap_t hp = {0, 0, 0};
    const double SQRT2 = hypot(1.0, 1.0); /* math.hypot(1,1) */
    const int s_id = sx * SY + sy, g_id = gx * SY + gy;
    parent[s_id] = s_id;
    g[s_id] = 0.0;
    g[g_id] = INFINITY; /* astar.py:40 : overrides g[start] when start == goal */
    {
        heap_item it = {g[s_id] + 2.5 * (double)(abs(gx - sx) + abs(gy - sy)), sx, sy};
        heap_push(&hp, it);
    }
    const int start_in_obs = obs[s_id];
    while (hp.n > 0) {
        heap_item cur = heap_pop(&hp);
        if (n_expanded) (*n_expanded)++;
        if (cur.x == gx && cur.y == gy) break;
        const int cid = cur.x * SY + cur.y;
        for (int k = 0; k < 8; k++) {
            int nx = cur.x + U_SET[k][0], ny = cur.y + U_SET[k][1];
            /* astar.py:98-118 is_collision(s, s_n) -> cost inf */
            int coll = 0;
            if (cid == s_id ? start_in_obs : obs[cid]) coll = 1;
            if (cur.x < 0 || cur.x > W || cur.y < 0 || cur.y > H) coll = 1;
            if (nx < 0 || nx > W || ny < 0 || ny > H) coll = 1;
            if (!coll && obs[nx * SY + ny]) coll = 1;
            if (coll) continue; /* new_cost = inf is never < g[s_n] */
            double cost = (U_SET[k][0] != 0 && U_SET[k][1] != 0) ? SQRT2 : 1.0;
            double new_cost = g[cid] + cost;
            int nid = nx * SY + ny;
            if (new_cost < g[nid]) {
                g[nid] = new_cost;
                parent[nid] = cid;
                heap_item it = {new_cost + 2.5 * (double)(abs(gx - nx) + abs(gy - ny)), nx, ny};
                heap_push(&hp, it);
            }
        }
    }
    int n = 0;
    if (parent[g_id] < 0) { /* KeyError in extract_path -> [s_start] (astar.py:67-71) */
        out_path[0][0] = (int16_t)sx; out_path[0][1] = (int16_t)sy; n = 1;
    } else {
        int s = g_id;
        out_path[n][0] = (int16_t)gx; out_path[n][1] = (int16_t)gy; n++;
        for (;;) {
            int p = parent[s];
            out_path[n][0] = (int16_t)(p / SY); out_path[n][1] = (int16_t)(p % SY); n++;
            s = p;
            if (s == s_id || n >= PEO_MAX_PATH) break;
        }
    }
    free(g); free(parent); free(hp.v);
    return n;
}

/* ------------------------------------------------------------------ replan (agent.py:232-259, 202-230) */
static void inflate_into(const peo_config *c, const uint8_t *src, uint8_t *dst, int ext) {
    /* Occupied_Grid_Map.py:157-166 / :126-135 : square [-ext, ext] around every set cell, clipped by in_bound */
    for (int x = 0; x < c->W; x++)
        for (int y = 0; y < c->H; y++)
            if (src[x * c->H + y])
                for (int xx = x - ext; xx <= x + ext; xx++)
                    for (int yy = y - ext; yy <= y + ext; yy++)
                        if (in_bound_i(c, xx, yy)) dst[xx * c->H + yy] = 1;
}

void peo_replan(const peo_config *c, peo_env *e) {
    const int W = c->W, H = c->H, WH = W * H, SY = H + 1;
    const int sx = py_round(e->eva[0]), sy = py_round(e->eva[1]);
    uint8_t *dyn = (uint8_t *)malloc(WH), *pred = (uint8_t *)malloc(WH), *defc = (uint8_t *)malloc(WH);
    uint8_t *obs = (uint8_t *)malloc((W + 1) * (H + 1));
    int ext = c->extend_dis;
    int n = 1;
    e->astar_expansions = 0;
    while (ext >= 0) {
        memcpy(dyn, e->grid, WH);
        inflate_into(c, e->grid, dyn, ext);             /* dynamic_map.extended_obstacles(ext) */
        memcpy(pred, dyn, WH);
        memset(defc, 0, WH);
        for (int i = 0; i < c->P; i++) {                /* pred_map.set_moving_obstacle (Occupied_Grid_Map.py:119-124) */
            int px = py_round(e->def_state[i * 4 + 0]), py = py_round(e->def_state[i * 4 + 1]);
            defc[px * H + py] = 1;
            pred[px * H + py] = 1;
        }
        inflate_into(c, defc, pred, ext);               /* pred_map.extended_moving_obstacles(ext) */
        memset(obs, 0, (W + 1) * (H + 1));
        for (int x = 0; x < W; x++)
            for (int y = 0; y < H; y++) obs[x * SY + y] = dyn[x * H + y];
        /* local_observation (Occupied_Grid_Map.py:177-191) : half-open square, disk radius view */
        const int vr = c->evader_view;
        for (int cx = sx - vr; cx < sx + vr; cx++)
            for (int cy = sy - vr; cy < sy + vr; cy++) {
                if (!in_bound_i(c, cx, cy)) continue;
                if (norm2((double)(sx - cx), (double)(sy - cy)) > (double)vr) continue;
                if (dyn[cx * H + cy] == 0 && pred[cx * H + cy] != 0) obs[cx * SY + cy] = 1; /* agent.py:223-229 */
            }
        int nexp = 0;
        n = peo_astar(W, H, obs, sx, sy, e->target[0], e->target[1], e->path, &nexp);
        e->astar_expansions += nexp;
        if (n >= 2) break;
        ext -= 1;
    }
    e->path_len = n;
    free(dyn); free(pred); free(defc); free(obs);
}

/* ------------------------------------------------------------------ evader tick (pursuit_env.py:75-102) */
void peo_evader_step(const peo_config *c, peo_env *e) {
    if (e->t % c->difficulty == 0) peo_replan(c, e);
    if (e->path_len >= 2) {
        const int16_t *last = e->path[e->path_len - 1];
        if (norm2(e->eva[0] - (double)last[0], e->eva[1] - (double)last[1]) < c->resolution) e->path_len--;
    }
    const int16_t *wp = e->path[e->path_len - 1];
    /* agent.py:261-271 waypoint2phi, then pursuit_env.py:93 action = vmax*(cos phi, sin phi).
     * phi = sign(dy)*acos(cc): cos(phi) = cc, sin(phi) = sign(dy)*sqrt((1-cc)(1+cc)); sign(0) = 0 -> phi = 0. */
    double dx = (double)wp[0] - e->eva[0], dy = (double)wp[1] - e->eva[1];
    double radius = norm2(dx, dy);
    double cphi = 1.0, sphi = 0.0;
    if (!(radius <= 0.01)) {
        if (dy != 0.0) {
            double cc = dx / (radius + 1e-3);
            double ss = sqrt((1.0 - cc) * (1.0 + cc));
            cphi = cc;
            sphi = dy > 0.0 ? ss : -ss;
        }
    }
    double ux = cphi * c->eva_vmax, uy = sphi * c->eva_vmax;
    double ns[4];
    dynamic(c->eva_tau, c->eva_dt, e->eva, ux, uy, ns);
    if (in_bound_f(c, ns[0], ns[1]) && e->grid[py_round(ns[0]) * c->H + py_round(ns[1])] == 0) memcpy(e->eva, ns, sizeof ns);
    /* target re-draw is tested on the PROPOSED position (pursuit_env.py:98-100) */
    if (norm2((double)e->target[0] - ns[0], (double)e->target[1] - ns[1]) <= c->eva_collision_radius) {
        int k = e->tape_pos < c->tape_len ? e->tape_pos : c->tape_len - 1;
        e->target[0] = e->tape[2 * k];
        e->target[1] = e->tape[2 * k + 1];
        e->tape_pos++;
    }
}

/* ------------------------------------------------------------------ LiDAR (pursuit_env.py:29-53) */
void peo_lidar_cell(const peo_config *c, const peo_env *e, int cx, int cy, uint8_t *flags) {
    for (int b = 0; b < c->num_beams; b++) {
        double bx = c->beam_dir[b][0], by = c->beam_dir[b][1];
        for (int r = 0; r < c->lidar_radius; r++) {
            double px = (double)cx + (double)r * bx;
            double py = (double)cy + (double)r * by;
            if (px < 0 || px >= (double)c->W || py < 0 || py >= (double)c->H) break;
            int ix = (int)px, iy = (int)py; /* int() truncation */
            int16_t id = e->bidx[ix * c->H + iy];
            if (id >= 0) { flags[id] = 1; break; }
        }
    }
}

/* agent.py:319-341 + :157-169 */
static int los_free(const peo_config *c, const peo_env *e, int x0, int y0, int x1, int y1) {
    int dx = abs(x1 - x0), dy = abs(y1 - y0);
    int sx = x0 > x1 ? -1 : 1, sy = y0 > y1 ? -1 : 1;
    int err = dx - dy;
    for (;;) {
        if (e->grid[x0 * c->H + y0] == 1) return 0;
        if (x0 == x1 && y0 == y1) break;
        int e2 = 2 * err;
        if (e2 > -dy) { err -= dy; x0 += sx; }
        if (e2 < dx) { err += dx; y0 += sy; }
    }
    return 1;
}

void peo_observe(const peo_config *c, const peo_env *e, float *p_state, float *e_state, float *p_adj, float *e_adj, float *o_adj) {
    const int P = c->P, O = c->O;
    for (int i = 0; i < P * 4; i++) p_state[i] = (float)e->def_state[i];
    for (int i = 0; i < 4; i++) e_state[i] = (float)e->eva[i];
    /* communicate (pursuit_env.py:182-195) incl. the adj[j,1] quirk */
    memset(p_adj, 0, sizeof(float) * P * P);
    for (int i = 0; i < P; i++)
        for (int j = 0; j < P; j++)
            if (i <= j && norm2(e->def_state[i * 4] - e->def_state[j * 4], e->def_state[i * 4 + 1] - e->def_state[j * 4 + 1]) <= c->def_comm_range) {
                p_adj[i * P + j] = 1.f;
                p_adj[j * P + 1] = 1.f;
            }
    /* sensor (pursuit_env.py:197-209) */
    memset(o_adj, 0, sizeof(float) * P * O);
    uint8_t *flags = (uint8_t *)malloc(O > e->n_obs ? O : e->n_obs);
    const int ex = py_round(e->eva[0]), ey = py_round(e->eva[1]);
    for (int i = 0; i < P; i++) {
        memset(flags, 0, O > e->n_obs ? O : e->n_obs);
        peo_lidar_cell(c, e, (int)e->def_state[i * 4], (int)e->def_state[i * 4 + 1], flags);
        for (int k = 0; k < e->n_obs && k < O; k++) o_adj[i * O + k] = (float)flags[k];
        int px = py_round(e->def_state[i * 4]), py = py_round(e->def_state[i * 4 + 1]);
        double d = norm2((double)(px - ex), (double)(py - ey));
        e_adj[i] = (d > c->def_sen_range) ? 0.f : (float)los_free(c, e, px, py, ex, ey);
    }
    free(flags);
}

/* ------------------------------------------------------------------ defender tick (pursuit_env.py:104-149) */
int peo_step(const peo_config *c, peo_env *e, const int32_t *actions, double *reward, uint8_t *can_apply) {
    const int P = c->P;
    double *prop = (double *)malloc(sizeof(double) * P * 4);
    e->t += 1;
    for (int i = 0; i < P; i++) dynamic(c->def_tau, c->def_dt, e->def_state + i * 4, c->action_u[actions[i]][0], c->action_u[actions[i]][1], prop + i * 4);
    for (int i = 0; i < P; i++) {
        double *s = prop + i * 4;
        int rew = 0;
        int cnt = 0;
        for (int j = 0; j < P; j++) /* collision_detection 'defender' on the (partly clipped) proposals, :165-177 */
            if (norm2(prop[j * 4] - s[0], prop[j * 4 + 1] - s[1]) <= c->def_collision_radius) cnt++;
        rew -= (cnt - 1);
        int col = 0; /* :152-163 */
        for (int a = -1; a < 2 && !col; a++)
            for (int b = -1; b < 2; b++) {
                double qx = s[0] + (double)a * c->def_collision_radius, qy = s[1] + (double)b * c->def_collision_radius;
                if (in_bound_f(c, qx, qy)) col = e->grid[py_round(qx) * c->H + py_round(qy)] != 0;
                if (col) break;
            }
        rew -= col;
        if (rew < 0) { can_apply[i] = 0; e->collision = 1; reward[i] = (double)rew; continue; }
        /* np.clip in place (:143-145) -- later defenders are scored against the clipped proposal */
        s[0] = s[0] < 0.0 ? 0.0 : (s[0] > (double)(c->W - 1) ? (double)(c->W - 1) : s[0]);
        s[1] = s[1] < 0.0 ? 0.0 : (s[1] > (double)(c->H - 1) ? (double)(c->H - 1) : s[1]);
        if (norm2(e->eva[0] - s[0], e->eva[1] - s[1]) <= c->def_collision_radius) rew += 1;
        can_apply[i] = 1;
        reward[i] = (double)rew;
    }
    for (int i = 0; i < P; i++)
        if (can_apply[i]) memcpy(e->def_state + i * 4, prop + i * 4, sizeof(double) * 4);
    free(prop);
    return e->t >= c->max_steps;
}

/* DHGN/normalization.py:12-35 */
void peo_reward_norm(const peo_config *c, peo_env *e, const double *reward, double *out) {
    e->rn_n += 1;
    for (int i = 0; i < c->P; i++) {
        double x = reward[i];
        if (e->rn_n == 1) {
            e->rn_mean[i] = x;
            out[i] = (x - x) / (x + 1e-8); /* std := x on the first sample */
        } else {
            double old = e->rn_mean[i];
            double mean = old + (x - old) / (double)e->rn_n;
            e->rn_S[i] = e->rn_S[i] + (x - old) * (x - mean);
            e->rn_mean[i] = mean;
            double sd = sqrt(e->rn_S[i] / (double)e->rn_n);
            out[i] = (x - mean) / (sd + 1e-8);
        }
    }
}

/* ------------------------------------------------------------------ handles */
peo_env *peo_create(const peo_config *c) {
    peo_env *e = (peo_env *)calloc(1, sizeof(peo_env));
    e->grid = (uint8_t *)calloc(c->W * c->H, 1);
    e->bidx = (int16_t *)malloc(sizeof(int16_t) * c->W * c->H);
    e->def_state = (double *)calloc(c->P * 4, sizeof(double));
    e->tape = (int32_t *)calloc(2 * (c->tape_len > 0 ? c->tape_len : 1), sizeof(int32_t));
    e->rn_mean = (double *)calloc(c->P, sizeof(double));
    e->rn_S = (double *)calloc(c->P, sizeof(double));
    return e;
}
void peo_destroy(peo_env *e) {
    if (!e) return;
    free(e->grid); free(e->bidx); free(e->def_state); free(e->tape); free(e->rn_mean); free(e->rn_S); free(e);
}
void peo_load(const peo_config *c, peo_env *e, const uint8_t *grid, const int32_t *obs_xy, int32_t n_obs, const double *def_state,
              const double *eva, const int32_t *target, const int32_t *tape) {
    memcpy(e->grid, grid, c->W * c->H);
    for (int i = 0; i < c->W * c->H; i++) e->bidx[i] = -1;
    for (int k = 0; k < n_obs; k++) e->bidx[obs_xy[2 * k] * c->H + obs_xy[2 * k + 1]] = (int16_t)k;
    e->n_obs = n_obs;
    memcpy(e->def_state, def_state, sizeof(double) * 4 * c->P);
    memcpy(e->eva, eva, sizeof(double) * 4);
    e->target[0] = target[0]; e->target[1] = target[1];
    if (c->tape_len > 0) memcpy(e->tape, tape, sizeof(int32_t) * 2 * c->tape_len);
    e->tape_pos = 0; e->t = 0; e->collision = 0; e->path_len = 0; /* reward normaliser persists across episodes */
}
void peo_get(const peo_config *c, const peo_env *e, double *def_state, double *eva, int32_t *target, int32_t *scalars) {
    memcpy(def_state, e->def_state, sizeof(double) * 4 * c->P);
    memcpy(eva, e->eva, sizeof(double) * 4);
    target[0] = e->target[0]; target[1] = e->target[1];
    scalars[0] = e->t; scalars[1] = e->path_len; scalars[2] = e->tape_pos; scalars[3] = e->collision; scalars[4] = e->astar_expansions;
}
void peo_get_path(const peo_env *e, int16_t *out) { memcpy(out, e->path, sizeof(int16_t) * 2 * e->path_len); }
void peo_get_rn(const peo_config *c, const peo_env *e, double *o) {
    o[0] = (double)e->rn_n;
    memcpy(o + 1, e->rn_mean, sizeof(double) * c->P);
    memcpy(o + 1 + c->P, e->rn_S, sizeof(double) * c->P);
}

double peo_tick_batch(const peo_config *c, peo_env **envs, int n, const int32_t *actions, float *scratch) {
    /* scratch: >= 4P + 4 + P*P + P + P*O floats */
    const int P = c->P;
    float *p_state = scratch, *e_state = p_state + 4 * P, *p_adj = e_state + 4, *e_adj = p_adj + P * P, *o_adj = e_adj + P;
    double rew[64], rn[64], acc = 0.0;
    uint8_t ok[64];
    for (int i = 0; i < n; i++) {
        peo_observe(c, envs[i], p_state, e_state, p_adj, e_adj, o_adj);
        peo_evader_step(c, envs[i]);
        peo_step(c, envs[i], actions + (size_t)i * P, rew, ok);
        peo_reward_norm(c, envs[i], rew, rn);
        for (int k = 0; k < P; k++) acc += rew[k] + (double)o_adj[k * c->O] + (double)e_adj[k];
    }
    return acc;
}

/* elementwise primitives for device-vs-host checks (tests only) */
void peo_prims(int n, const double *a, const double *b, double *out) {
    for (int i = 0; i < n; i++) { out[i] = norm2(a[i], b[i]); out[n + i] = a[i] / b[i]; out[2 * n + i] = (double)py_round(a[i]); }
}
