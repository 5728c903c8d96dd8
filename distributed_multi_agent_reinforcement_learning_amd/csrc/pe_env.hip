// pe_env.hip -- batched pursuit-evasion environment for MI355X (gfx950 / CDNA4).  C ABI: include/pe_env.h.
//
// One 64-lane wavefront (= one workgroup) owns one environment.  The environment's record (occupancy grid, agent
// state) is contiguous in HBM, so every global access of the wave is coalesced by construction; grid + agent state are
// staged once per launch in LDS and all neighbour / obstacle / line-of-sight / A* lookups run out of LDS.  The LiDAR
// row of a cell (pursuit_env.py:29-53 get_raser_map) is tabulated per episode by k_build_raser, bit-packed, and the tick
// gathers the P rows of the defenders' cells.  Launches keep the block -> environment map fixed (block b ==
// environment b), and blocks b, b+8, ... share an XCD, so an environment's record stays in the same XCD's L2 from tick
// to tick.
//
// Numerics: environment state is f64.  Build with -ffp-contract=off: the only fused multiply-add is the
// explicit one inside norm2(), which mirrors how numpy evaluates np.linalg.norm of a 2-vector in the
// reference.  Rounding to cells uses round-half-even (Python round), int() truncates.
//
// Reference citations (paths relative to the reference root) are given per function.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <cmath>

#include "pe_env.h"
#include "pe_env_diag.h"

namespace {

constexpr int WAVE = 64;
constexpr int PE_TAPE_LDS = 32;   // tape entries prefetched with the first batch of loads (longer tapes: a second, looped copy)
constexpr int PE_WP_WINDOW = 16;  // waypoints kept next to the meta record (>= pops between two replans)
constexpr double SQRT2 = 0x1.6a09e667f3bcdp+0;  // math.hypot(1, 1)  (astar.py:96)

__device__ __forceinline__ int py_round(double v) { return (int)__builtin_rint(v); }
__device__ __forceinline__ double norm2(double a, double b) { return __builtin_sqrt(__builtin_fma(b, b, a * a)); }
// norm2(a, b) <= r without the f64 square root on the device: sqrt is correctly rounded and monotone, so
// { x : sqrt(x) <= r } = { x : x <= t } for the largest double t with sqrt(t) <= r (likewise for <).  The thresholds of
// the configuration's radii are found on the host (launch()) with the same correctly rounded sqrt; a comparison of
// the squared norm against them decides exactly like the reference's comparison of the norm.
__device__ __forceinline__ double norm2sq(double a, double b) { return __builtin_fma(b, b, a * a); }
// a / b for a divisor known on the host, in three flops instead of the ~15-instruction IEEE sequence: with y = RN(1 / b),
// q = RN(a y), r = a - b q (exact in one fma) the corrected q' = RN(q + r y) is the correctly rounded quotient
// (Markstein) unless b's significand is all ones -- the host checks that and the exponent range and then passes y = 0,
// which selects the plain division.  r == 0 keeps q (also keeps the sign of a zero quotient).
__device__ __forceinline__ double div_const(double a, double b, double y) {
    if (y == 0.0) return a / b;  // wave-uniform
    const double q = a * y;
    const double r = __builtin_fma(-q, b, a);
    return r == 0.0 ? q : __builtin_fma(r, y, q);
}
struct SqThr {
    double inv_def_tau, inv_eva_tau, inv_six;  // RN(1 / b) for div_const, 0 = use the IEEE division
    double coll_le;     // sqrt(x) <= defender.collision_radius
    double comm_le;     // sqrt(x) <= defender.comm_range
    double sen_le;      // sqrt(x) <= defender.sen_range
    double evacoll_le;  // sqrt(x) <= attacker.collision_radius
    double res_lt;      // sqrt(x) <  map.resolution
    uint32_t o4_magic;  // floor(i / (O / 4)) == (i * o4_magic) >> 20 for every i < P * O / 4 (host-verified), 0 = divide
    int32_t rw_shift;   // log2 of the raser row length in words when it is a power of two, else -1
};
__host__ __device__ inline int raser_row_words(int O) { return (((O + 31) >> 5) + 3) & ~3; }  // 16-byte aligned rows

// One wave per workgroup: the lanes only ever exchange data with lanes of their own wave.  A wave's LDS and vector-memory
// instructions execute in program order through the same LDS / L1, so a wavefront-scope fence (no s_waitcnt, no s_barrier)
// plus the compiler scheduling barrier is all the ordering the exchange needs; __syncthreads() would drain every
// outstanding global load/store (vmcnt(0)) at each of the ~20 sync points of a tick.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL>
__device__ __forceinline__ unsigned int dpp_mov(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
// Wave-wide lexicographic arg-min of (f, sel), f >= 0 (or +inf); every lane ends up with the minimum pair.  Two short
// reductions instead of one over a 96-bit key: v_min_f64 over f, then v_min_u32 over the sel of the lanes that hold the
// minimum f.  Four row-rotate DPP steps reduce inside the 16-lane rows on the VALU, v_readlane combines the four rows.
// (An LDS atomic-min on one address would be serialised lane by lane by the compiler's atomic optimiser.)
__device__ __forceinline__ void wave_argmin_f(double &f, unsigned int &sel) {
    double m = f;
#define PE_DPP_STEP(CTRL)                                                                                     \
    {                                                                                                         \
        const unsigned long long k = (unsigned long long)__double_as_longlong(m);                             \
        const unsigned int oh = dpp_mov<CTRL>((unsigned int)(k >> 32)), ol = dpp_mov<CTRL>((unsigned int)k);  \
        m = __builtin_fmin(m, __longlong_as_double((long long)(((unsigned long long)oh << 32) | ol)));        \
    }
    PE_DPP_STEP(0x121)  // row_ror:1
    PE_DPP_STEP(0x122)  // row_ror:2
    PE_DPP_STEP(0x124)  // row_ror:4
    PE_DPP_STEP(0x128)  // row_ror:8
#undef PE_DPP_STEP
    {
        const unsigned long long k = (unsigned long long)__double_as_longlong(m);
        const int kh = (int)(k >> 32), kl = (int)(unsigned int)k;
#define PE_ROWF(L) __longlong_as_double((long long)(((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(kh, L) << 32) | (unsigned int)__builtin_amdgcn_readlane(kl, L)))
        m = __builtin_fmin(__builtin_fmin(PE_ROWF(0), PE_ROWF(16)), __builtin_fmin(PE_ROWF(32), PE_ROWF(48)));
#undef PE_ROWF
    }
    unsigned int s = (f == m) ? sel : ~0u;
#define PE_DPP_STEP(CTRL) { const unsigned int o = dpp_mov<CTRL>(s); s = o < s ? o : s; }
    PE_DPP_STEP(0x121)
    PE_DPP_STEP(0x122)
    PE_DPP_STEP(0x124)
    PE_DPP_STEP(0x128)
#undef PE_DPP_STEP
    {
        const unsigned int r0 = (unsigned int)__builtin_amdgcn_readlane((int)s, 0), r1 = (unsigned int)__builtin_amdgcn_readlane((int)s, 16),
                           r2 = (unsigned int)__builtin_amdgcn_readlane((int)s, 32), r3 = (unsigned int)__builtin_amdgcn_readlane((int)s, 48);
        const unsigned int a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
        s = a < b ? a : b;
    }
    f = m;
    sel = s;
}

struct Lds {
    uint8_t *grid;   // [W*H]
    double *def;     // [4][P]
    double *prop;    // [4][P] + [6]: the evader's desired velocity (2) and proposed state (4) of the merged tick
    double *eva;     // [4]
    uint32_t *rw;    // [P][raser_row_words(O)]  LiDAR rows (hit bits) of the defenders' cells
    uint8_t *cond;   // [P*P]
    int32_t *misc;   // [8 + 2*PE_MAX_P]: [8..] rounded defender cells (replan)
    // replan scratch
    uint8_t *obs;    // [(W+1)*(H+1)]
    uint8_t *open;   // [(W+1)*(H+1)]  node is in the OPEN list
    uint16_t *parent;  // [(W+1)*(H+1)]
    uint16_t *olist; // [(W+1)*(H+1)]  compact OPEN list (unordered)
    double *g;       // [(W+1)*(H+1)]
    // copies of the small per-environment records, prefetched with the grids in ONE batch of global loads at kernel start
    int32_t *m;      // [PE_META_INTS] meta | [PE_MAX_P] actions | [2] target | [PE_WP_WINDOW] waypoints | [2*tape_len] tape (whole)
    __device__ __forceinline__ int32_t *acts() const { return m + PE_META_INTS; }
    __device__ __forceinline__ int32_t *tg() const { return m + PE_META_INTS + PE_MAX_P; }
    __device__ __forceinline__ uint32_t *wp() const { return (uint32_t *)(m + PE_META_INTS + PE_MAX_P + 2); }
    __device__ __forceinline__ int32_t *tp() const { return m + PE_META_INTS + PE_MAX_P + 2 + PE_WP_WINDOW; }
    double *rnl;     // [1 + 2P]  reward normaliser
};

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

__host__ __device__ inline size_t lds_layout(const pe_config &c, bool with_obs, bool with_replan, unsigned char *base, Lds *l) {
    const int WH = c.W * c.H, P = c.P, NN = (c.W + 1) * (c.H + 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align16(off + bytes); return o; };
    size_t o_g = take(sizeof(double) * (with_replan ? NN : 0));
    size_t o_rnl = take(sizeof(double) * (1 + 2 * PE_MAX_P));
    size_t o_def = take(sizeof(double) * 4 * P);
    size_t o_prop = take(sizeof(double) * (4 * P + 6));
    size_t o_eva = take(sizeof(double) * 4);
    size_t o_rw = take(with_obs ? sizeof(uint32_t) * P * raser_row_words(c.O) : 0);
    size_t o_misc = take(sizeof(int32_t) * (8 + 2 * PE_MAX_P));
    size_t o_parent = take(sizeof(uint16_t) * (with_replan ? NN : 0));
    size_t o_olist = take(sizeof(uint16_t) * (with_replan ? NN : 0));
    size_t o_m = take(sizeof(int32_t) * (PE_META_INTS + PE_MAX_P + 2 + PE_WP_WINDOW + 2 * (c.tape_len > 0 ? c.tape_len : 1)));
    size_t o_grid = take(WH);
    size_t o_cond = take(P * P);
    size_t o_obs = take(with_replan ? NN : 0);
    size_t o_open = take(with_replan ? NN : 0);
    if (l) {
        l->g = (double *)(base + o_g);
        l->def = (double *)(base + o_def);
        l->prop = (double *)(base + o_prop);
        l->eva = (double *)(base + o_eva);
        l->rw = (uint32_t *)(base + o_rw);
        l->misc = (int32_t *)(base + o_misc);
        l->parent = (uint16_t *)(base + o_parent);
        l->olist = (uint16_t *)(base + o_olist);
        l->rnl = (double *)(base + o_rnl);
        l->m = (int32_t *)(base + o_m);
        l->grid = base + o_grid;
        l->cond = base + o_cond;
        l->obs = base + o_obs;
        l->open = base + o_open;
    }
    return off;
}

// ---- coalesced copies between HBM records and LDS ------------------------------------------------------
__device__ __forceinline__ void copy_in(void *dst, const void *src, int nbytes, int lane) {
    if ((nbytes & 15) == 0 && (((uintptr_t)src) & 15) == 0) {
        const uint4 *s = (const uint4 *)src;
        uint4 *d = (uint4 *)dst;
        for (int i = lane; i < (nbytes >> 4); i += WAVE) d[i] = s[i];
    } else if ((nbytes & 3) == 0 && (((uintptr_t)src) & 3) == 0) {
        const uint32_t *s = (const uint32_t *)src;
        uint32_t *d = (uint32_t *)dst;
        for (int i = lane; i < (nbytes >> 2); i += WAVE) d[i] = s[i];
    } else {
        const uint8_t *s = (const uint8_t *)src;
        uint8_t *d = (uint8_t *)dst;
        for (int i = lane; i < nbytes; i += WAVE) d[i] = s[i];
    }
}
__device__ __forceinline__ void copy_out_f32(float *dst, const float *src, int n, int lane) {
    if ((n & 3) == 0 && (((uintptr_t)dst) & 15) == 0) {
        const float4 *s = (const float4 *)src;
        float4 *d = (float4 *)dst;
        for (int i = lane; i < (n >> 2); i += WAVE) d[i] = s[i];
    } else {
        for (int i = lane; i < n; i += WAVE) dst[i] = src[i];
    }
}

// ---- agent.py:74-104 : first-order lag integrated with RK4, association exactly as written -------------
__device__ __forceinline__ void dynamic(double tau, double itau, double i6, double h, double x, double y, double vx0, double vy0, double ux,
                                        double uy, double *o) {
    double k1 = div_const(ux - vx0, tau, itau);
    double k2 = div_const(ux - (vx0 + h * k1 / 2), tau, itau);
    double k3 = div_const(ux - (vx0 + h * k2 / 2), tau, itau);
    double k4 = div_const(ux - (vx0 + h * k3), tau, itau);
    double vx = vx0 + div_const((k1 + 2 * k2 + 2 * k3 + k4) * h, 6.0, i6);
    k1 = div_const(uy - vy0, tau, itau);
    k2 = div_const(uy - (vy0 + h * k1 / 2), tau, itau);
    k3 = div_const(uy - (vy0 + h * k2 / 2), tau, itau);
    k4 = div_const(uy - (vy0 + h * k3), tau, itau);
    double vy = vy0 + div_const((k1 + 2 * k2 + 2 * k3 + k4) * h, 6.0, i6);
    o[0] = x + vx * h;
    o[1] = y + vy * h;
    o[2] = vx;
    o[3] = vy;
}

__device__ __forceinline__ bool in_bound_i(const pe_config &c, int x, int y) { return x < c.W && x >= 0 && y < c.H && y >= 0; }

// ---- defenders: Pursuit_Env.step + defender_reward + collision_detection (pursuit_env.py:104-177) ------
// The reference scores the defenders one after the other and clips an accepted proposal IN PLACE before the next one is
// scored against it.  The order only matters through accepted proposals that lie outside the clip box
// [0, W-1] x [0, H-1] (a defender pushing against the map border): for everybody else clipping is the identity.
// FAST8 (P <= 8, lane = 8 i + j): every pair distance is evaluated once against the unclipped AND the clipped proposal of j,
// every probe of every defender in two wave instructions; the in-place semantics are then resolved on 64-bit wave masks --
// for the (rare) out-of-box defenders b in index order: accepted(b) swaps column b of the pair mask to the clipped
// distances for the rows behind it.  Exactly the reference's sequential result, without the P-iteration loop.
// MERGE_EVA (fused tick without replan): lane P integrates the evader's first-order lag in the SAME instructions as the
// defenders' (agent.py:74-104 is one formula with per-agent tau / step size); its heading was computed by dev_evader_pre, its
// proposal is consumed by dev_evader_post after the observations were taken from the old evader state.
template <bool FAST8, bool MERGE_EVA>
__device__ void dev_step(const pe_config &c, const SqThr &th, const Lds &l, int lane, double *def_hbm, const pe_step_out &out, int env) {
    const int P = c.P;
    int32_t *meta = l.m;
    double *rn = l.rnl;
    double o[4] = {0.0, 0.0, 0.0, 0.0};
    {
        double tau = c.def_tau, itau = th.inv_def_tau, h = c.def_dt, x = 0.0, y = 0.0, vx = 0.0, vy = 0.0, ux = 0.0, uy = 0.0;
        const bool is_eva = MERGE_EVA && lane == P;
        if (lane < P) {
            int a = l.acts()[lane];
            a = a < 0 ? 0 : (a > 8 ? 8 : a);
            x = l.def[lane]; y = l.def[P + lane]; vx = l.def[2 * P + lane]; vy = l.def[3 * P + lane];
            ux = c.action_u[a][0]; uy = c.action_u[a][1];
        } else if (is_eva) {
            tau = c.eva_tau; itau = th.inv_eva_tau; h = c.eva_dt;
            x = l.eva[0]; y = l.eva[1]; vx = l.eva[2]; vy = l.eva[3];
            ux = l.prop[4 * P]; uy = l.prop[4 * P + 1];
        }
        if (lane < P || is_eva) dynamic(tau, itau, th.inv_six, h, x, y, vx, vy, ux, uy, o);
        if (lane < P) { l.prop[lane] = o[0]; l.prop[P + lane] = o[1]; l.prop[2 * P + lane] = o[2]; l.prop[3 * P + lane] = o[3]; }
        if (is_eva) { l.prop[4 * P + 2] = o[0]; l.prop[4 * P + 3] = o[1]; l.prop[4 * P + 4] = o[2]; l.prop[4 * P + 5] = o[3]; }
    }
    wave_sync();
    const double r = c.def_collision_radius;
    const double ex = l.eva[0], ey = l.eva[1];
    const double wmax = (double)(c.W - 1), hmax = (double)(c.H - 1);
    int my_rew = 0, my_ok = 0, any_coll = 0;
    if (FAST8) {
        const int i = lane >> 3, j = lane & 7;
        const bool iv = i < P, pv = iv && j < P;
        const double xi = l.prop[iv ? i : 0], yi = l.prop[P + (iv ? i : 0)];
        const double xj = l.prop[pv ? j : 0], yj = l.prop[P + (pv ? j : 0)];
        const double cxj = xj < 0.0 ? 0.0 : (xj > wmax ? wmax : xj), cyj = yj < 0.0 ? 0.0 : (yj > hmax ? hmax : yj);
        // inner collisions (pursuit_env.py:132-137): j against i's own (still unclipped) proposal, self included
        const unsigned long long Mu = __ballot(pv && norm2sq(xj - xi, yj - yi) <= th.coll_le);
        const unsigned long long Mc = __ballot(pv && norm2sq(cxj - xi, cyj - yi) <= th.coll_le);
        const unsigned int outb = (unsigned int)(__ballot(pv && i == 0 && (cxj != xj || cyj != yj)) & 0xFFull);
        // 3x3 probe of the static map at half-radius offsets; only in-bound probes count, any occupied one rejects
        // (pursuit_env.py:152-163): probes 0..7 of defender i on lanes 8 i + k, probe 8 (offset (+r, +r)) on lanes 8 i
        unsigned long long HA, HB;
        {
            const int a3 = (j * 11) >> 5;  // j / 3 for j < 8
            const double qx = xi + (double)(a3 - 1) * r, qy = yi + (double)(j - 3 * a3 - 1) * r;
            const int ix = py_round(qx), iy = py_round(qy);
            HA = __ballot(iv && in_bound_i(c, ix, iy) && l.grid[ix * c.H + iy] != 0);  // all eight probes, whatever P is
            const double q8x = xi + r, q8y = yi + r;  // (double)1 * r == r
            const int jx = py_round(q8x), jy = py_round(q8y);
            HB = __ballot(iv && j == 0 && in_bound_i(c, jx, jy) && l.grid[jx * c.H + jy] != 0);
        }
        unsigned long long M = Mu;
        for (unsigned int todo = outb; todo;) {  // wave-uniform; empty unless a proposal left the clip box
            const int b = __ffs(todo) - 1;
            todo &= todo - 1u;
            const int cnt = __popcll((M >> (8 * b)) & 0xFFull);
            const bool col = (((HA >> (8 * b)) & 0xFFull) | ((HB >> (8 * b)) & 1ull)) != 0ull;
            if (-(cnt - 1) - (col ? 1 : 0) >= 0 && b < 7) {  // accepted: later defenders are scored against its clipped position
                const unsigned long long colmask = (0x0101010101010101ull << b) & (~0ull << (8 * (b + 1)));
                M = (M & ~colmask) | (Mc & colmask);
            }
        }
        double cx = 0.0, cy = 0.0;
        if (lane < P) {
            const int cnt = __popcll((M >> (8 * lane)) & 0xFFull);
            const int col = ((((HA >> (8 * lane)) & 0xFFull) | ((HB >> (8 * lane)) & 1ull)) != 0ull) ? 1 : 0;
            my_rew = -(cnt - 1) - col;
            if (my_rew >= 0) {
                cx = o[0] < 0.0 ? 0.0 : (o[0] > wmax ? wmax : o[0]);
                cy = o[1] < 0.0 ? 0.0 : (o[1] > hmax ? hmax : o[1]);
                if (norm2sq(ex - cx, ey - cy) <= th.coll_le) my_rew += 1;
                my_ok = 1;
            }
        }
        any_coll = __ballot(lane < P && !my_ok) != 0ull;
        if (lane < P && my_ok) {
            l.def[lane] = cx; l.def[P + lane] = cy; l.def[2 * P + lane] = o[2]; l.def[3 * P + lane] = o[3];
            def_hbm[lane] = cx; def_hbm[P + lane] = cy; def_hbm[2 * P + lane] = o[2]; def_hbm[3 * P + lane] = o[3];
        }
    } else {
        for (int i = 0; i < P; i++) {  // sequential in the agent index: proposals are clipped in place
            const double sx = l.prop[i], sy = l.prop[P + i];
            // inner collisions against the current (possibly already clipped) proposals, self included
            bool near = (lane < P) && (norm2sq(l.prop[lane] - sx, l.prop[P + lane] - sy) <= th.coll_le);
            int cnt = __popcll(__ballot(near));
            // 3x3 probe of the static map at half-radius offsets; only in-bound probes count (pursuit_env.py:152-163)
            bool hit = false;
            if (lane < 9) {
                int a = lane / 3 - 1, b = lane % 3 - 1;
                double qx = sx + (double)a * r, qy = sy + (double)b * r;
                int ix = py_round(qx), iy = py_round(qy);
                if (in_bound_i(c, ix, iy)) hit = l.grid[ix * c.H + iy] != 0;
            }
            int col = __ballot(hit) != 0ull;
            int rew = -(cnt - 1) - col;
            int ok = 0;
            if (rew < 0) {
                any_coll = 1;
            } else {
                double cx = sx < 0.0 ? 0.0 : (sx > wmax ? wmax : sx);
                double cy = sy < 0.0 ? 0.0 : (sy > hmax ? hmax : sy);
                if (lane == 0) { l.prop[i] = cx; l.prop[P + i] = cy; }
                if (norm2sq(ex - cx, ey - cy) <= th.coll_le) rew += 1;
                ok = 1;
            }
            if (lane == i) { my_rew = rew; my_ok = ok; }
            wave_sync();
        }
        if (lane < P && my_ok) {
            double nx = l.prop[lane], ny = l.prop[P + lane], nvx = l.prop[2 * P + lane], nvy = l.prop[3 * P + lane];
            l.def[lane] = nx; l.def[P + lane] = ny; l.def[2 * P + lane] = nvx; l.def[3 * P + lane] = nvy;
            def_hbm[lane] = nx; def_hbm[P + lane] = ny; def_hbm[2 * P + lane] = nvx; def_hbm[3 * P + lane] = nvy;
        }
    }
    if (lane < P) {
        // DHGN/normalization.py:12-35 : per-environment running mean/std of the reward vector
        double x = (double)my_rew, outv = x;
        if (c.use_reward_norm) {
            double n = rn[0] + 1.0;
            if (n == 1.0) {
                rn[1 + lane] = x;
                outv = (x - x) / (x + 1e-8);
            } else {
                // the two divisions by the sample count share one reciprocal (div_const is exact: n is an integer far below
                // 2^53, so its significand is never all ones)
                const double yn = 1.0 / n;
                double old = rn[1 + lane];
                double mean = old + div_const(x - old, n, yn);
                double S = rn[1 + P + lane] + (x - old) * (x - mean);
                rn[1 + lane] = mean;
                rn[1 + P + lane] = S;
                outv = (x - mean) / (__builtin_sqrt(div_const(S, n, yn)) + 1e-8);
            }
        }
        if (out.reward) out.reward[(int64_t)env * out.reward_stride + lane] = (float)outv;
        if (out.reward_raw) out.reward_raw[(int64_t)env * out.reward_raw_stride + lane] = (float)my_rew;
    }
    wave_sync();
    if (lane == 0) {
        if (c.use_reward_norm) rn[0] = rn[0] + 1.0;
        int t = meta[PE_META_T] + 1;
        meta[PE_META_T] = t;
        if (any_coll) meta[PE_META_COLLISION] = 1;
        if (out.done) out.done[env] = (uint8_t)(t >= c.max_steps);
    }
    wave_sync();
}

// ---- observations: get_state, communicate, sensor (base_env.py:198-209, pursuit_env.py:182-209) --------
// o_adj[i] = raser_map[int(x_i)][int(y_i)] (pursuit_env.py:201): the P rows of the episode's bit-packed raser table
// (k_build_raser) are gathered with ONE global load issued up front and expanded to the fp32 rows the policy reads.
template <bool FAST8>
__device__ void dev_observe(const pe_config &c, const SqThr &th, const Lds &l, int lane, int env, const pe_obs_out &o, const uint32_t *raser_env) {
    const int P = c.P, O = c.O, RW = raser_row_words(O);
    uint32_t r_row = 0u;
    const bool one_load = P * RW <= WAVE;
    const bool want_rows = o.o_adj || o.o_adj_bits;
    if (want_rows && one_load && lane < P * RW) {
        const int i = th.rw_shift >= 0 ? lane >> th.rw_shift : lane / RW, w = lane - i * RW;
        const int cell = (int)l.def[i] * c.H + (int)l.def[P + i];
        r_row = raser_env[(size_t)cell * RW + w];  // in flight during the adjacency / line-of-sight work below
    }
    if (o.p_state && lane < 4 * P) o.p_state[(int64_t)env * o.p_state_stride + lane] = (float)l.def[(lane & 3) * P + (lane >> 2)];
    if (o.e_state && lane < 4) o.e_state[(int64_t)env * o.e_state_stride + lane] = (float)l.eva[lane];
    // communicate(): upper-triangular range test; `adj[j, 1] = 1` fires for every row j through the pair (j, j), whose
    // distance is 0 -- column 1 is all ones whenever comm_range >= 0
    const bool col1 = th.comm_le >= 0.0;
    if (o.p_adj) {
        if (FAST8) {
            const int i = lane >> 3, j = lane & 7;
            if (i < P && j < P) {
                const bool v = ((i <= j) && (norm2sq(l.def[i] - l.def[j], l.def[P + i] - l.def[P + j]) <= th.comm_le)) || (j == 1 && col1);
                o.p_adj[(int64_t)env * o.p_adj_stride + i * P + j] = v ? 1.f : 0.f;
            }
        } else {
            for (int idx = lane; idx < P * P; idx += WAVE) {
                const int i = idx / P, j = idx - i * P;
                const bool v = ((i <= j) && (norm2sq(l.def[i] - l.def[j], l.def[P + i] - l.def[P + j]) <= th.comm_le)) || (j == 1 && col1);
                o.p_adj[(int64_t)env * o.p_adj_stride + idx] = v ? 1.f : 0.f;
            }
        }
    }
    // find_attacker(): rounded-cell range test + Bresenham line of sight over the static map (agent.py:157-169, 319-341)
    if (o.e_adj && lane < P) {
        int x0 = py_round(l.def[lane]), y0 = py_round(l.def[P + lane]);
        int x1 = py_round(l.eva[0]), y1 = py_round(l.eva[1]);
        float seen = 0.f;
        if (norm2sq((double)(x0 - x1), (double)(y0 - y1)) <= th.sen_le) {
            int dx = abs(x1 - x0), dy = abs(y1 - y0);
            int sx = x0 > x1 ? -1 : 1, sy = y0 > y1 ? -1 : 1;
            int err = dx - dy;
            seen = 1.f;
            for (int it = 0; it < 4 * (c.W + c.H); it++) {  // bounded: the line has at most dx+dy+1 cells
                if (l.grid[x0 * c.H + y0] == 1) { seen = 0.f; break; }
                if (x0 == x1 && y0 == y1) break;
                int e2 = 2 * err;
                if (e2 > -dy) { err -= dy; x0 += sx; }
                if (e2 < dx) { err += dx; y0 += sy; }
            }
        }
        o.e_adj[(int64_t)env * o.e_adj_stride + lane] = seen;
    }
    if (want_rows) {
        uint32_t *bits = o.o_adj_bits ? o.o_adj_bits + (int64_t)env * o.o_adj_bits_stride : nullptr;
        if (one_load) {
            if (lane < P * RW) {
                if (o.o_adj) l.rw[lane] = r_row;
                if (bits) bits[lane] = r_row;
            }
        } else {
            for (int idx = lane; idx < P * RW; idx += WAVE) {
                const int i = idx / RW, w = idx - i * RW;
                const uint32_t v = raser_env[(size_t)((int)l.def[i] * c.H + (int)l.def[P + i]) * RW + w];
                l.rw[idx] = v;
                if (bits) bits[idx] = v;
            }
        }
    }
    if (o.o_adj) {
        wave_sync();
        // hit bits -> fp32 rows, four obstacles (16 bytes) per lane and store: the P rows of O floats are contiguous
        float *dst = o.o_adj + (int64_t)env * o.o_adj_stride;
        const int o4 = O >> 2, n4 = P * o4;
        if ((((uintptr_t)dst) & 15) == 0) {
            for (int idx = lane; idx < n4; idx += WAVE) {
                const int i = th.o4_magic ? (int)(((uint32_t)idx * th.o4_magic) >> 20) : idx / o4, j4 = idx - i * o4;
                const uint32_t wd = l.rw[i * RW + (j4 >> 3)];
                const int sh = (j4 & 7) << 2;
                ((float4 *)dst)[idx] = make_float4((float)((wd >> sh) & 1u), (float)((wd >> (sh + 1)) & 1u), (float)((wd >> (sh + 2)) & 1u),
                                                   (float)((wd >> (sh + 3)) & 1u));
            }
        } else {
            for (int idx = lane; idx < P * O; idx += WAVE) {
                const int i = idx / O, j = idx - i * O;
                dst[idx] = (float)((l.rw[i * RW + (j >> 5)] >> (j & 31)) & 1u);
            }
        }
    }
    wave_sync();
}

// ---- weighted A* (astar.py:26-161) on an LDS-resident problem ------------------------------------------
// OPEN is a compact unordered list of node ids in LDS (at most one live entry per node); the pop is a wave-wide
// arg-min of (f, x, y) over the list, which is the order heapq yields for (f, (x, y)) tuples: lanes scan the list,
// a wave-wide min over f finds the minimum, a min over the node ids of the lanes that hold it breaks ties (id order ==
// (x, y) tuple order; wave_argmin_f).  Stale duplicates of the
// reference's heap never change g/PARENT when popped (same g, same sums), so keeping only the live entry is
// equivalent.  Returns the true path length (goal -> start); stores the last min(len, max_path) nodes in path_out.
__device__ int dev_astar(int W, int H, const Lds &l, int lane, int sx, int sy, int gx, int gy, int16_t *path_out, int max_path,
                         int *n_stored, int *n_expanded, int *status) {
    const int SY = H + 1, NN = (W + 1) * SY;
    const int s_id = sx * SY + sy, g_id = gx * SY + gy;
    *n_expanded = 0;
    if (l.obs[g_id]) {  // astar.py:46-47
        if (lane == 0) { path_out[0] = (int16_t)sx; path_out[1] = (int16_t)sy; }
        *n_stored = 1;
        return 1;
    }
    // Reachability pre-check (maps up to 63x63): an 8-connected flood fill from the start on row bitmasks (lane = row y,
    // bit = column x).  The search only tests edge endpoints (astar.py:98-118), so 8-connectivity is exact.  When the goal
    // cannot be reached the reference's search drains its whole OPEN set and returns [s_start] (astar.py:67-71); that
    // answer does not depend on the pop order, so the exhaustive exploration (the straggler of a launch) is skipped.
    if (W + 1 <= 64 && H + 1 <= 64 && s_id != g_id) {
        unsigned long long freem = 0ull;
        if (lane <= H)
            for (int x = 0; x <= W; x++) freem |= (unsigned long long)(l.obs[x * SY + lane] == 0) << x;
        unsigned long long reach = (lane == sy) ? ((1ull << sx) & freem) : 0ull;
        bool found = false;
        for (int it = 0; it < NN; it++) {
            unsigned long long up = __shfl_up(reach, 1), dn = __shfl_down(reach, 1);
            if (lane == 0) up = 0ull;
            if (lane == WAVE - 1) dn = 0ull;
            unsigned long long m = reach | up | dn;
            m |= (m << 1) | (m >> 1);
            const unsigned long long nr = (reach | m) & freem;
            const bool changed = nr != reach;
            reach = nr;
            if (__ballot(lane == gy && ((reach >> gx) & 1ull)) != 0ull) { found = true; break; }
            if (__ballot(changed) == 0ull) break;
        }
        if (!found) {
            if (lane == 0) { path_out[0] = (int16_t)sx; path_out[1] = (int16_t)sy; }
            *n_stored = 1;
            return 1;
        }
    }
    for (int i = lane; i < NN; i += WAVE) { l.g[i] = __builtin_inf(); l.open[i] = 0; l.parent[i] = 0xFFFF; }
    wave_sync();
    if (lane == 0) {
        l.parent[s_id] = (uint16_t)s_id;
        l.g[s_id] = (s_id == g_id) ? __builtin_inf() : 0.0;  // astar.py:39-40: g[goal] = inf overrides g[start]
        l.open[s_id] = 1;
        l.olist[0] = (uint16_t)((sx << 8) | sy);
    }
    int cnt = 1;  // wave-uniform length of the OPEN list
    wave_sync();
    const int cap = 16 * NN;  // every wave leaves the loop: the open set drains or the cap trips
    int expanded = 0;
    for (int it = 0; it < cap && cnt > 0; it++) {
        // --- pop: arg-min over OPEN of (f = g + 2.5 * manhattan, id); f >= 0, so its bit pattern orders like f
        double fb = __builtin_inf();
        unsigned int sel = ~0u;
        for (int sl = lane; sl < cnt; sl += WAVE) {
            const int xy = l.olist[sl];  // x << 8 | y : no integer division in the scan
            const int x = xy >> 8, y = xy & 255, id = x * SY + y;
            const double f = l.g[id] + 2.5 * (double)(abs(gx - x) + abs(gy - y));
            const unsigned int se = ((unsigned int)id << 16) | (unsigned int)sl;
            if (f < fb || (f == fb && se < sel)) { fb = f; sel = se; }
        }
        wave_argmin_f(fb, sel);
        const int bid = (int)(sel >> 16), slot = (int)(sel & 0xFFFFu);
        expanded++;
        if (bid == g_id) break;
        const int cxy = l.olist[slot], cx = cxy >> 8, cy = cxy & 255;
        const double gc = l.g[bid];
        const bool cur_blocked = l.obs[bid] != 0;  // is_collision(s_start=cur, .) (astar.py:106-107)
        const int last = l.olist[cnt - 1];
        wave_sync();
        if (lane == 0) { l.open[bid] = 0; l.olist[slot] = (uint16_t)last; }
        cnt -= 1;
        wave_sync();
        bool push = false;
        int nxy = 0;
        if (lane < 8 && !cur_blocked) {
            // u_set order (-1,0),(-1,1),(0,1),(1,1),(1,0),(1,-1),(0,-1),(-1,-1) (astar.py:11-12); neighbours are distinct
            const int ux = (lane < 2 || lane == 7) ? -1 : ((lane >= 3 && lane <= 5) ? 1 : 0);
            const int uy = (lane >= 1 && lane <= 3) ? 1 : ((lane >= 5) ? -1 : 0);
            const int nx = cx + ux, ny = cy + uy;
            if (nx >= 0 && nx <= W && ny >= 0 && ny <= H) {  // '>' bounds: x == W and y == H are legal (astar.py:109-113)
                const int nid = nx * SY + ny;
                nxy = (nx << 8) | ny;
                // the three lookups are independent: one LDS round trip instead of three nested ones
                const uint8_t nb = l.obs[nid], nopen = l.open[nid];
                const double gn = l.g[nid];
                const double nc = gc + ((ux != 0 && uy != 0) ? SQRT2 : 1.0);
                if (!nb && nc < gn) {
                    l.g[nid] = nc;
                    l.parent[nid] = (uint16_t)bid;
                    l.open[nid] = 1;
                    push = nopen == 0;
                }
            }
        }
        const unsigned long long pm = __ballot(push);
        if (push) l.olist[cnt + __popcll(pm & ((1ull << lane) - 1ull))] = (uint16_t)nxy;
        cnt += __popcll(pm);
        wave_sync();
        if (it == cap - 1) *status |= PE_STATUS_ASTAR_CAP;
    }
    *n_expanded = expanded;
    // extract_path (astar.py:130-146); KeyError -> [s_start] (astar.py:67-71)
    int len = 1;
    if (l.parent[g_id] == 0xFFFF) {
        if (lane == 0) { path_out[0] = (int16_t)sx; path_out[1] = (int16_t)sy; }
        *n_stored = 1;
        return 1;
    }
    {
        int s = g_id;
        for (int k = 0; k < NN + 1; k++) { s = l.parent[s]; len++; if (s == s_id) break; }
    }
    const int cnt_out = len < max_path ? len : max_path;
    if (lane == 0) {
        int s = g_id, k = 0;
        const int skip = len - cnt_out;
        for (;;) {
            if (k >= skip) { path_out[2 * (k - skip)] = (int16_t)(s / SY); path_out[2 * (k - skip) + 1] = (int16_t)(s % SY); }
            k++;
            if (k >= len) break;
            s = l.parent[s];
        }
    }
    *n_stored = cnt_out;
    return len;
}

// ---- Evader.replan + rescan (agent.py:202-259, Occupied_Grid_Map.py:119-191) ----------------------------
__device__ void dev_replan(const pe_config &c, const Lds &l, int lane, int16_t *path, uint32_t *wp_hbm) {
    int32_t *meta = l.m;
    const int32_t *target = l.tg();
    const int W = c.W, H = c.H, P = c.P, SY = H + 1, NN = (W + 1) * SY;
    const int sx = py_round(l.eva[0]), sy = py_round(l.eva[1]);
    const int gx = target[0], gy = target[1];
    int ext = c.extend_dis;
    int len = 1, cnt = 1, total_exp = 0, status = 0;
    if (lane < P) { l.misc[8 + 2 * lane] = py_round(l.def[lane]); l.misc[8 + 2 * lane + 1] = py_round(l.def[P + lane]); }
    wave_sync();
    while (ext >= 0) {
        // OBS = static U inflate(static, ext) U {visible cells that only the defender-augmented map blocks}
        const int vr = c.evader_view;
        for (int i = lane; i < NN; i += WAVE) {
            int x = i / SY, y = i - x * SY;
            uint8_t v = 0;
            if (x < W && y < H) {
                bool dyn = false;
                for (int xx = x - ext; xx <= x + ext; xx++)
                    for (int yy = y - ext; yy <= y + ext; yy++)
                        if (in_bound_i(c, xx, yy) && l.grid[xx * H + yy]) dyn = true;
                v = dyn;
                if (!dyn) {
                    int dx = sx - x, dy = sy - y;
                    // np.linalg.norm of an integer pair <= view range  <=>  dx^2 + dy^2 <= vr^2 (exact in integers)
                    bool visible = (x >= sx - vr) && (x < sx + vr) && (y >= sy - vr) && (y < sy + vr) && (dx * dx + dy * dy <= vr * vr);
                    if (visible) {
                        bool pred = false;
                        for (int k = 0; k < P; k++)
                            if (abs(l.misc[8 + 2 * k] - x) <= ext && abs(l.misc[8 + 2 * k + 1] - y) <= ext) pred = true;
                        v = pred;
                    }
                }
            }
            l.obs[i] = v;
        }
        wave_sync();
        int nexp = 0;
        len = dev_astar(W, H, l, lane, sx, sy, gx, gy, path, c.max_path, &cnt, &nexp, &status);
        total_exp += nexp;
        wave_sync();
        if (len >= 2) break;
        ext -= 1;
    }
    if (lane == 0) {
        meta[PE_META_PATH_LEN] = len;
        meta[PE_META_PATH_CNT] = cnt;
        meta[PE_META_ASTAR_EXP] = total_exp;
        if (status) meta[PE_META_STATUS] |= status;
        meta[PE_META_WP_HEAD] = 0;
        // the next PE_WP_WINDOW waypoints, in the order the evader visits them (path[cnt-1], path[cnt-2], ...)
        for (int k = 0; k < PE_WP_WINDOW; k++) {
            const int idx = cnt - 1 - k >= 0 ? cnt - 1 - k : 0;
            const uint32_t v = ((uint32_t)(uint16_t)path[2 * idx] << 16) | (uint32_t)(uint16_t)path[2 * idx + 1];
            l.wp()[k] = v;
            wp_hbm[k] = v;
        }
    }
    wave_sync();
}

// path[n-1] (the next waypoint) packed x << 16 | y: from the prefetched window, or from HBM beyond it
__device__ __forceinline__ uint32_t dev_waypoint(const Lds &l, const int16_t *path, int head, int n) {
    if (head < PE_WP_WINDOW) return l.wp()[head];
    return ((uint32_t)(uint16_t)path[2 * (n - 1)] << 16) | (uint32_t)(uint16_t)path[2 * (n - 1) + 1];
}

// ---- Pursuit_Env.attacker_step (pursuit_env.py:75-102), waypoint2phi (agent.py:261-271) -----------------
// Three pieces on lane 0: waypoint bookkeeping + heading (pre), the lag integration (dynamic()), acceptance + target re-draw
// (post).  dev_evader runs them back to back; the fused no-replan tick runs `pre` before the defenders' step, integrates on
// lane P inside dev_step<.., MERGE_EVA> and runs `post` after the observations (which must see the old evader state, Q3).
__device__ __forceinline__ void dev_evader_pre(const pe_config &c, const SqThr &th, const Lds &l, const int16_t *path, double &ux_out, double &uy_out) {
    int32_t *meta = l.m;
    int len = meta[PE_META_PATH_LEN], cnt = meta[PE_META_PATH_CNT], head = meta[PE_META_WP_HEAD];
    const double ex = l.eva[0], ey = l.eva[1];
    int status = 0;
    if (cnt < 1) { status |= PE_STATUS_PATH_UNDERFLOW; cnt = 1; }
    // path[cnt-1] is the next waypoint; the window holds it without a dependent global read
    uint32_t wv = dev_waypoint(l, path, head, cnt);
    double wx = (double)(int16_t)(wv >> 16), wy = (double)(int16_t)(wv & 0xFFFFu);
    if (len >= 2 && norm2sq(ex - wx, ey - wy) <= th.res_lt) {
        len--;
        if (cnt > 1) {
            cnt--; head++;
            wv = dev_waypoint(l, path, head, cnt);
            wx = (double)(int16_t)(wv >> 16); wy = (double)(int16_t)(wv & 0xFFFFu);
        } else status |= PE_STATUS_PATH_UNDERFLOW;
    }
    // phi = sign(dy) * arccos(dx / (r + 1e-3)); u = vmax * (cos phi, sin phi).  cos(arccos(q)) == q and
    // sin(arccos(q)) == sqrt((1-q)(1+q)) are used in place of libm (same form in the CPU oracle); sign(0) == 0
    // gives phi == 0.
    double dx = wx - ex, dy = wy - ey;
    double radius = norm2(dx, dy);
    double cphi = 1.0, sphi = 0.0;
    if (!(radius <= 0.01) && dy != 0.0) {
        double q = dx / (radius + 1e-3);
        double s = __builtin_sqrt((1.0 - q) * (1.0 + q));
        cphi = q;
        sphi = dy > 0.0 ? s : -s;
    }
    ux_out = cphi * c.eva_vmax;
    uy_out = sphi * c.eva_vmax;
    meta[PE_META_PATH_LEN] = len;
    meta[PE_META_PATH_CNT] = cnt;
    meta[PE_META_WP_HEAD] = head;
    if (status) meta[PE_META_STATUS] |= status;
}

__device__ __forceinline__ void dev_evader_post(const pe_config &c, const SqThr &th, const Lds &l, double ns0, double ns1, double ns2, double ns3,
                                                int32_t *target_hbm, const int32_t *tape_hbm, double *eva_hbm) {
    const double ns[4] = {ns0, ns1, ns2, ns3};
    int32_t *meta = l.m;
    int status = 0;
    int ix = py_round(ns[0]), iy = py_round(ns[1]);
    if (in_bound_i(c, ix, iy) && l.grid[ix * c.H + iy] == 0) {
        l.eva[0] = ns[0]; l.eva[1] = ns[1]; l.eva[2] = ns[2]; l.eva[3] = ns[3];
        eva_hbm[0] = ns[0]; eva_hbm[1] = ns[1]; eva_hbm[2] = ns[2]; eva_hbm[3] = ns[3];
    }
    // the target is re-drawn when the PROPOSED position reaches it (pursuit_env.py:98-100); draws come from the tape
    if (norm2sq((double)l.tg()[0] - ns[0], (double)l.tg()[1] - ns[1]) <= th.evacoll_le) {
        int pos = meta[PE_META_TAPE_POS];
        int k = pos < c.tape_len ? pos : c.tape_len - 1;
        if (pos >= c.tape_len) status |= PE_STATUS_TAPE_EXHAUSTED;
        const int nx = l.tp()[2 * k], ny = l.tp()[2 * k + 1];  // the whole tape is LDS-resident
        l.tg()[0] = nx; l.tg()[1] = ny;
        target_hbm[0] = nx; target_hbm[1] = ny;
        meta[PE_META_TAPE_POS] = pos + 1;
    }
    if (status) meta[PE_META_STATUS] |= status;
}

template <bool REPLAN>
__device__ void dev_evader(const pe_config &c, const SqThr &th, const Lds &l, int lane, int16_t *path, uint32_t *wp_hbm, int32_t *target_hbm,
                           const int32_t *tape_hbm, double *eva_hbm) {
    int32_t *meta = l.m;
    const int t = meta[PE_META_T];
    if (REPLAN) {
        if (t % c.difficulty == 0) {  // wave-uniform
            dev_replan(c, l, lane, path, wp_hbm);
        }
    }
    if (lane == 0) {
        double ux, uy, ns[4];
        dev_evader_pre(c, th, l, path, ux, uy);
        dynamic(c.eva_tau, th.inv_eva_tau, th.inv_six, c.eva_dt, l.eva[0], l.eva[1], l.eva[2], l.eva[3], ux, uy, ns);
        dev_evader_post(c, th, l, ns[0], ns[1], ns[2], ns[3], target_hbm, tape_hbm, eva_hbm);
    }
    wave_sync();
}

// WPB wavefronts (= environments) per workgroup: the waves of a workgroup never synchronise with each other (wave_sync only),
// a fatter workgroup just lets the dispatcher start the grid in a quarter of the time (4096 one-wave workgroups take ~2.7 us
// to start, first to last; the whole tick runs ~12).  The replan variant keeps one wave per workgroup (30 KB of LDS each).
template <bool STEP, bool OBS, bool EVA, bool REPLAN, bool FAST8, int WPB>
__global__ __launch_bounds__(WAVE * WPB) void k_tick(const pe_config c, const pe_state st, const int32_t *actions, const pe_step_out sout,
                                                     const pe_obs_out oout, const SqThr th, const int lds_per_env) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int env = blockIdx.x * WPB + (WPB > 1 ? (int)(threadIdx.x >> 6) : 0), lane = threadIdx.x & (WAVE - 1);
    if (env >= st.N) return;
    Lds l;
    lds_layout(c, OBS, EVA && REPLAN, smem + (WPB > 1 ? (size_t)(threadIdx.x >> 6) * lds_per_env : 0), &l);
    const int WH = c.W * c.H, P = c.P;
    double *def_hbm = st.def + (size_t)env * 4 * P;
    double *eva_hbm = st.eva + (size_t)env * 4;
    int32_t *meta_hbm = st.meta + (size_t)env * PE_META_INTS;
    double *rn_hbm = st.rn + (size_t)env * (1 + 2 * P);
    uint32_t *wp_hbm = st.wpw + (size_t)env * PE_WP_WINDOW;
    int32_t *target_hbm = st.target + (size_t)env * 2;
    const int32_t *tape_hbm = st.tape + (size_t)env * c.tape_len * 2;
    // ---- ONE batch of global loads: every record this tick needs is requested before the first wait, so a wave pays the
    // HBM/L2 latency once instead of once per phase
    const uint8_t *gsrc = st.grid + (size_t)env * WH;
    const bool wide = (WH & 15) == 0 && WH <= 4096;
    const int ng = WH >> 4;
    const uint4 *g4 = (const uint4 *)gsrc;
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
    // named registers (not an array: an indexed array would live in scratch memory)
#define PE_LDG(k) uint4 rg##k = z4; if (wide && lane + WAVE * k < ng) rg##k = g4[lane + WAVE * k];
    PE_LDG(0) PE_LDG(1) PE_LDG(2) PE_LDG(3)
#undef PE_LDG
    double r_def = 0.0, r_eva = 0.0, r_rn = 0.0;
    int32_t r_meta = 0, r_act = 0, r_tg = 0, r_tp = 0;
    uint32_t r_wp = 0;
    if (lane < 4 * P) r_def = def_hbm[lane];
    if (lane < 4) r_eva = eva_hbm[lane];
    if (lane < PE_META_INTS) r_meta = meta_hbm[lane];
    if (lane < 2) r_tg = target_hbm[lane];
    if (STEP && lane < P) r_act = actions[(size_t)env * P + lane];
    if (STEP && lane < 1 + 2 * P) r_rn = rn_hbm[lane];
    if (EVA && lane < PE_WP_WINDOW) r_wp = wp_hbm[lane];
    if (EVA && lane < 2 * c.tape_len && lane < 2 * PE_TAPE_LDS) r_tp = tape_hbm[lane];
    if (wide) {
#define PE_STG(k) if (lane + WAVE * k < ng) ((uint4 *)l.grid)[lane + WAVE * k] = rg##k;
        PE_STG(0) PE_STG(1) PE_STG(2) PE_STG(3)
#undef PE_STG
    } else {
        copy_in(l.grid, gsrc, WH, lane);
    }
    if (lane < 4 * P) l.def[lane] = r_def;
    if (lane < 4) l.eva[lane] = r_eva;
    if (lane < PE_META_INTS) l.m[lane] = r_meta;
    if (lane < 2) l.tg()[lane] = r_tg;
    if (STEP && lane < P) l.acts()[lane] = r_act;
    if (STEP && lane < 1 + 2 * P) l.rnl[lane] = r_rn;
    if (EVA && lane < PE_WP_WINDOW) l.wp()[lane] = r_wp;
    if (EVA && lane < 2 * c.tape_len && lane < 2 * PE_TAPE_LDS) l.tp()[lane] = r_tp;
    if (EVA && c.tape_len > PE_TAPE_LDS) for (int i = 2 * PE_TAPE_LDS + lane; i < 2 * c.tape_len; i += WAVE) l.tp()[i] = tape_hbm[i];
    wave_sync();
    constexpr bool MERGE_EVA = STEP && EVA && !REPLAN;  // the evader's lag integration rides on lane P of the defenders' step
    int16_t *path = st.path + (size_t)env * c.max_path * 2;
    if (MERGE_EVA) {
        if (lane == 0) {
            double ux, uy;
            dev_evader_pre(c, th, l, path, ux, uy);
            l.prop[4 * P] = ux; l.prop[4 * P + 1] = uy;
        }
        wave_sync();
    }
    if (STEP) dev_step<FAST8, MERGE_EVA>(c, th, l, lane, def_hbm, sout, env);
    if (OBS) dev_observe<FAST8>(c, th, l, lane, env, oout, st.raser + (size_t)env * WH * raser_row_words(c.O));
    if (MERGE_EVA) {
        if (lane == 0) dev_evader_post(c, th, l, l.prop[4 * P + 2], l.prop[4 * P + 3], l.prop[4 * P + 4], l.prop[4 * P + 5], target_hbm, tape_hbm, eva_hbm);
        wave_sync();
    } else if (EVA) {
        dev_evader<REPLAN>(c, th, l, lane, path, wp_hbm, target_hbm, tape_hbm, eva_hbm);
    }
    // ---- write the small records back (def / eva / target were written where they changed)
    if ((STEP || EVA) && lane < PE_META_INTS) meta_hbm[lane] = l.m[lane];
    if (STEP && c.use_reward_norm && lane < 1 + 2 * P) rn_hbm[lane] = l.rnl[lane];
}

// ---- get_raser_map (pursuit_env.py:29-53): the LiDAR row of EVERY cell, once per episode -----------------------------------
// raser[cell][k] = 1 iff one of the num_beams beams from (x, y) = cell hits boundary obstacle k first: samples at the integer
// ranges r = 0 .. radius-1, px = x + r cos, py = y + r sin (separate f64 multiply and add, truncation toward zero), the beam
// ends at the first sample outside the map or on a boundary cell.  Bit-packed rows of raser_row_words(O) words.
// One wavefront per environment, lane = cell (64 cells per pass), beams in the (wave-uniform) outer loop.
__global__ __launch_bounds__(WAVE) void k_build_raser(const pe_config c, const pe_state st) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int env = blockIdx.x, lane = threadIdx.x;
    const int W = c.W, H = c.H, WH = W * H, O = c.O, RW = raser_row_words(O), RS = RW + 1;  // odd row stride: no bank conflicts
    int16_t *bidx = (int16_t *)smem;
    uint32_t *rows = (uint32_t *)(smem + align16(sizeof(int16_t) * WH));  // [WAVE][RS]
    copy_in(bidx, st.bidx + (size_t)env * WH, WH * 2, lane);
    const int n_obs = st.n_obs[env];
    uint32_t *dst = st.raser + (size_t)env * WH * RW;
    wave_sync();
    for (int base = 0; base < WH; base += WAVE) {
        const int cell = base + lane;
        const bool live = cell < WH;
        const int cx = live ? cell / H : 0, cy = live ? cell - cx * H : 0;
        for (int w = 0; w < RW; w++) rows[lane * RS + w] = 0u;
        for (int b = 0; b < c.num_beams; b++) {
            const double bx = c.beam_dir[b][0], by = c.beam_dir[b][1];
            bool going = live;
            for (int r = 0; r < c.lidar_radius; r++) {
                if (__ballot(going) == 0ull) break;  // wave-uniform
                if (going) {
                    const double px = (double)cx + (double)r * bx, py = (double)cy + (double)r * by;
                    if (px < 0 || px >= (double)W || py < 0 || py >= (double)H) {
                        going = false;
                    } else {
                        const int id = bidx[(int)px * H + (int)py];
                        if (id >= 0) {
                            if (id < O && id < n_obs) rows[lane * RS + (id >> 5)] |= 1u << (id & 31);
                            going = false;
                        }
                    }
                }
            }
        }
        if (live)
            for (int q = 0; q < RW; q += 4)
                *(uint4 *)(dst + (size_t)cell * RW + q) = make_uint4(rows[lane * RS + q], rows[lane * RS + q + 1], rows[lane * RS + q + 2], rows[lane * RS + q + 3]);
    }
}

// bidx from the obstacle list (pursuit_env.py:21: index == position in np.argwhere order)
struct DevRng;
__device__ int dev_rng_status(const DevRng *rng, int env);
__global__ void k_build_bidx(const pe_config c, const pe_state st, const int32_t *obs_xy, const DevRng *rng) {
    const int env = blockIdx.x;
    const int WH = c.W * c.H;
    int16_t *b = st.bidx + (size_t)env * WH;
    for (int i = threadIdx.x; i < WH; i += blockDim.x) b[i] = -1;
    __syncthreads();
    const int n = st.n_obs[env];
    for (int k = threadIdx.x; k < n && k < c.O; k += blockDim.x) {
        int x = obs_xy[((size_t)env * c.O + k) * 2], y = obs_xy[((size_t)env * c.O + k) * 2 + 1];
        if (x >= 0 && x < c.W && y >= 0 && y < c.H) b[x * c.H + y] = (int16_t)k;
    }
    // a fresh meta record; on the device-reset path the status bits raised so far stay (sticky, see pe_env.h)
    if (threadIdx.x < PE_META_INTS)
        st.meta[(size_t)env * PE_META_INTS + threadIdx.x] = (threadIdx.x == PE_META_STATUS && rng) ? dev_rng_status(rng, env) : 0;
    if (threadIdx.x < PE_WP_WINDOW) st.wpw[(size_t)env * PE_WP_WINDOW + threadIdx.x] = 0u;
}

// [N][P][4] (get_state order) -> [N][4][P] records
__global__ void k_def_aos_to_soa(int N, int P, const double *aos, double *soa) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * P * 4) return;
    int n = i / (4 * P), r = i - n * 4 * P, k = r / P, a = r - k * P;
    soa[i] = aos[(size_t)n * 4 * P + a * 4 + k];
}

__global__ __launch_bounds__(WAVE) void k_astar(int W, int H, int n, const uint8_t *obs, const int32_t *sg, int16_t *out_path,
                                                int32_t *out_len, int max_path) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= n) return;
    pe_config c;
    c.W = W; c.H = H; c.P = 2; c.O = 4;
    Lds l;
    lds_layout(c, false, true, smem, &l);
    const int NN = (W + 1) * (H + 1);
    for (int i = lane; i < NN; i += WAVE) l.obs[i] = obs[(size_t)b * NN + i];
    wave_sync();
    int cnt = 0, nexp = 0, status = 0;
    int len = dev_astar(W, H, l, lane, sg[4 * b], sg[4 * b + 1], sg[4 * b + 2], sg[4 * b + 3], out_path + (size_t)b * max_path * 2,
                        max_path, &cnt, &nexp, &status);
    if (lane == 0) { out_len[2 * b] = len; out_len[2 * b + 1] = nexp; }
}

// ---- Pursuit_Env.demon (pursuit_env.py:211-229): the scripted pursuer -- the discrete action whose direction is closest to the
// bearing of the evader.  One lane per defender.  The reference evaluates action = (cos(phi), sin(phi)), phi = sign(dy) *
// arccos(dx / (radius + 1e-3)); here cos(arccos q) = q and sin(arccos q) = sqrt((1 - q)(1 + q)) (the same restatement as the
// evader's heading: numpy's arccos / cos / sin are not reproducible to the last bit across CPUs).  The result is an
// arg-min over nine distances whose runner-up is >= 0.39 away except on the eight bisector bearings, so the 1-2 ulp between the
// two forms cannot change an action off a measure-zero set; pinned by the recorded demon actions of the reference traces.
// sign(0) = 0 gives phi = 0, i.e. action 0, also for an evader straight to the left (kept).
struct DemonDirs { double d[9][2]; };
__global__ void k_demon(int N, int P, const double *__restrict__ def, const double *__restrict__ eva, DemonDirs dirs, int32_t *__restrict__ actions) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * P) return;
    const int n = i / P, a = i - n * P;
    const double *rec = def + (size_t)n * 4 * P;
    const double x = rec[a], y = rec[P + a], ex = eva[(size_t)n * 4], ey = eva[(size_t)n * 4 + 1];
    const double radius = norm2(x - ex, y - ey);
    double ax = 0.0, ay = 0.0;
    if (!(radius <= 0.01)) {   // math.isclose(radius, 0.0, abs_tol=0.01) with the default rel_tol is radius <= 0.01
        const double dy = ey - y;
        const double sg = dy > 0.0 ? 1.0 : (dy < 0.0 ? -1.0 : 0.0);
        const double q = (ex - x) / (radius + 1e-3);
        if (sg == 0.0) { ax = 1.0; ay = 0.0; }
        else { ax = q; ay = sg * __builtin_sqrt((1.0 - q) * (1.0 + q)); }
    }
    int best = 0;
    double bd = norm2(dirs.d[0][0] - ax, dirs.d[0][1] - ay);
#pragma unroll
    for (int k = 1; k < 9; k++) {
        const double dk = norm2(dirs.d[k][0] - ax, dirs.d[k][1] - ay);
        if (dk < bd) { bd = dk; best = k; }   // list.index(min(..)): the first minimum
    }
    actions[i] = best;
}

__global__ void k_diag_norm2(int n, const double *a, const double *b, double *out, double c0, double y0, double c1, double y1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        out[i] = norm2(a[i], b[i]);
        out[n + i] = a[i] / b[i];
        out[2 * n + i] = (double)py_round(a[i]);
        out[3 * n + i] = div_const(a[i], c0, y0);   // vs a / c0 on the host
        out[4 * n + i] = div_const(b[i], c1, y1);   // vs b / c1
    }
}

// Host: the largest double t with sqrt(t) <= r (strict: sqrt(t) < r); -1 when no x >= 0 qualifies (a squared norm is >= 0).
double sq_threshold(double r, bool strict) {
    auto ok = [&](double t) { const double s = sqrt(t); return strict ? s < r : s <= r; };
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

// Host: y = RN(1 / b) if div_const is exact for every finite a in the simulator's range, else 0 (plain division).
double div_const_reciprocal(double b) {
    if (!(b > 0.0) || std::isinf(b)) return 0.0;
    int e;
    const double m = frexp(b, &e);                               // b = m 2^e, m in [0.5, 1)
    if (m == nextafter(1.0, 0.0) || e < -500 || e > 500) return 0.0;  // significand all ones / far exponents
    return 1.0 / b;
}

// Host: M with floor(i / d) == (i * M) >> 20 for every 0 <= i < n, or 0 when no 32-bit product qualifies (checked exhaustively)
uint32_t div_magic20(uint32_t d, uint32_t n) {
    if (d == 0 || n == 0 || (uint64_t)n * ((1u << 20) / d + 1) >= (1ull << 32)) return 0;
    const uint32_t M = ((1u << 20) + d - 1) / d;
    for (uint32_t i = 0; i < n; i++)
        if (((i * M) >> 20) != i / d) return 0;
    return M;
}

template <bool STEP, bool OBS, bool EVA, bool REPLAN, bool FAST8, int WPB>
int launch3(const pe_config *cfg, const pe_state *st, const int32_t *actions, const pe_step_out *so, const pe_obs_out *oo, void *stream) {
    const size_t lds_env = align16(lds_layout(*cfg, OBS, EVA && REPLAN, nullptr, nullptr));
    const size_t lds = lds_env * WPB;
    auto kern = k_tick<STEP, OBS, EVA, REPLAN, FAST8, WPB>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    pe_step_out s0;
    memset(&s0, 0, sizeof s0);
    pe_obs_out o0;
    memset(&o0, 0, sizeof o0);
    SqThr th;
    th.inv_def_tau = div_const_reciprocal(cfg->def_tau);
    th.inv_eva_tau = div_const_reciprocal(cfg->eva_tau);
    th.inv_six = div_const_reciprocal(6.0);
    th.coll_le = sq_threshold(cfg->def_collision_radius, false);
    th.comm_le = sq_threshold(cfg->def_comm_range, false);
    th.sen_le = sq_threshold(cfg->def_sen_range, false);
    th.evacoll_le = sq_threshold(cfg->eva_collision_radius, false);
    th.res_lt = sq_threshold(cfg->resolution, true);
    th.o4_magic = div_magic20((uint32_t)(cfg->O >> 2), (uint32_t)(cfg->P * (cfg->O >> 2)));
    const int rw = raser_row_words(cfg->O);
    th.rw_shift = -1;
    for (int sft = 0; sft < 16; sft++)
        if ((1 << sft) == rw) th.rw_shift = sft;
    hipLaunchKernelGGL(kern, dim3((st->N + WPB - 1) / WPB), dim3(WAVE * WPB), lds, (hipStream_t)stream, *cfg, *st, actions, so ? *so : s0, oo ? *oo : o0, th,
                       (int)lds_env);
    return (int)hipGetLastError();
}

template <bool STEP, bool OBS, bool EVA, bool REPLAN, bool FAST8>
int launch2(const pe_config *cfg, const pe_state *st, const int32_t *actions, const pe_step_out *so, const pe_obs_out *oo, void *stream) {
    // four environments per workgroup unless the per-environment LDS is large (replan scratch, very large maps)
    const bool fat = !(EVA && REPLAN) && 4 * align16(lds_layout(*cfg, OBS, false, nullptr, nullptr)) <= 48 * 1024;
    return fat ? launch3<STEP, OBS, EVA, REPLAN, FAST8, (EVA && REPLAN) ? 1 : 4>(cfg, st, actions, so, oo, stream)
               : launch3<STEP, OBS, EVA, REPLAN, FAST8, 1>(cfg, st, actions, so, oo, stream);
}

template <bool STEP, bool OBS, bool EVA, bool REPLAN>
int launch(const pe_config *cfg, const pe_state *st, const int32_t *actions, const pe_step_out *so, const pe_obs_out *oo, void *stream) {
    if (OBS && !st->raser) return PE_ERR_NULL;
    return cfg->P <= 8 ? launch2<STEP, OBS, EVA, REPLAN, true>(cfg, st, actions, so, oo, stream)
                       : launch2<STEP, OBS, EVA, REPLAN, false>(cfg, st, actions, so, oo, stream);
}

// the per-episode tables derived from the grid and the obstacle list: bidx, fresh meta / waypoint records, raser rows
int launch_episode_tables(const pe_config *cfg, const pe_state *st, const int32_t *d_obs, const void *rng, hipStream_t s, bool shared_map = false);

// ---- Pursuit_Env.reset on the device ---------------------------------------------------------------------
// The host resetter (pe_reset.cpp) restated for one wavefront per environment: the same two generator streams (CPython
// random = MT19937 + init_by_array + randbelow, numpy legacy RandomState = MT19937 + init_genrand + polar gauss), the same
// draws in the same order, so an environment reset here starts exactly like one reset on the host (and like the
// reference seeded with the same number).  The MT states live in HBM between episodes and in LDS during the launch;
// lane 0 advances the streams and broadcasts every draw, so control flow stays wave-uniform; the grid work (block stamps,
// inflation, inner boundary in argwhere order) runs on all lanes.  numpy's gauss goes through log(): the device libm is
// within 1 ulp of glibc's, which can only matter if a block centre lands within 1 ulp of a rounding boundary.
struct DevRng {
    uint32_t py_mt[624], np_mt[624], snap_mt[624];
    int32_t py_idx, np_idx, snap_idx, has_gauss, has_tape;
    int32_t status;  // sticky PE_STATUS_* bits of every episode since pe_env_reset_seed (k_build_bidx copies them into meta)
    double gauss;
};

__device__ int dev_rng_status(const DevRng *rng, int env) { return rng[env].status; }

__device__ uint32_t mt_next(uint32_t *mt, int &idx) {  // lane 0 only
    if (idx >= 624) {
        for (int k = 0; k < 624; k++) {
            const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
__device__ uint32_t py_randbelow(uint32_t *mt, int &idx, uint32_t n) {  // random._randbelow_with_getrandbits
    int k = 0;
    for (uint32_t v = n; v; v >>= 1) k++;
    uint32_t r = mt_next(mt, idx) >> (32 - k);
    while (r >= n) r = mt_next(mt, idx) >> (32 - k);
    return r;
}
__device__ double np_random_sample(uint32_t *mt, int &idx) {
    const uint32_t a = mt_next(mt, idx) >> 5, b = mt_next(mt, idx) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}
// base_env.py:52-70 on lane 0: first free cell of the inflated map
// false: PE_RESET_MAX_DRAWS candidates were all occupied (the last one is kept), like the host resetter
__device__ bool draw_target_l0(int W, int H, const uint8_t *infl, uint32_t *mt, int &idx, int &tx, int &ty) {
    for (int draws = 0; draws < PE_RESET_MAX_DRAWS; draws++) {
        tx = (int)py_randbelow(mt, idx, (uint32_t)W);
        ty = (int)py_randbelow(mt, idx, (uint32_t)H);
        if (infl[tx * H + ty] == 0) return true;
    }
    return false;
}

__host__ __device__ inline size_t reset_lds_bytes(const pe_config &c) {
    return align16(2 * 624 * sizeof(uint32_t)) + 3 * align16((size_t)c.W * c.H) + align16(sizeof(double) * 2 * PE_MAX_P) +
           align16(sizeof(int32_t) * (2 * PE_MAX_P + 8));
}

__global__ __launch_bounds__(64) void k_reset(const pe_config c, const pe_state st, const pe_reset_params prm, DevRng *rng, uint8_t *infl_bank,
                                              int first, int32_t *obs_xy, double *def_aos, float *o_state) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int env = blockIdx.x, lane = threadIdx.x;
    const int W = c.W, H = c.H, WH = W * H, P = c.P, O = c.O;
    size_t off = 0;
    uint32_t *pym = (uint32_t *)(smem + off); off += align16(624 * sizeof(uint32_t));
    uint32_t *npm = (uint32_t *)(smem + off); off += align16(624 * sizeof(uint32_t));
    uint8_t *grid = smem + off; off += align16(WH);
    uint8_t *infl = smem + off; off += align16(WH);      // static inflation, then + the defenders' blocks
    uint8_t *infs = smem + off; off += align16(WH);      // static inflation of this episode (target re-draws)
    double *dxy = (double *)(smem + off); off += align16(sizeof(double) * 2 * PE_MAX_P);
    int32_t *cells = (int32_t *)(smem + off);            // [2 * PE_MAX_P] unique rounded defender cells
    DevRng &R = rng[env];
    uint8_t *bank = infl_bank + (size_t)env * WH;
    const bool rewind = !first && R.has_tape;
    for (int i = lane; i < 624; i += WAVE) { pym[i] = rewind ? R.snap_mt[i] : R.py_mt[i]; npm[i] = R.np_mt[i]; }
    int pyi = rewind ? R.snap_idx : R.py_idx, npi = R.np_idx;
    int has_gauss = R.has_gauss;
    double gauss = R.gauss;
    if (rewind) for (int i = lane; i < WH; i += WAVE) infs[i] = bank[i];
    wave_sync();
    int fail = 0;  // lane 0: a placement loop gave up
    if (!first && lane == 0) fail = st.meta[(size_t)env * PE_META_INTS + PE_META_STATUS] | R.status;  // carried over (sticky)
    if (rewind && lane == 0) {  // the reference draws a new target only on arrival: give the unused tape draws back
        const int consumed = st.meta[(size_t)env * PE_META_INTS + PE_META_TAPE_POS];
        int tx, ty;
        for (int k = 0; k < consumed; k++) draw_target_l0(W, H, infs, pym, pyi, tx, ty);
    }
    for (int i = lane; i < WH; i += WAVE) grid[i] = prm.fixed_grid ? prm.fixed_grid[i] : 0;
    wave_sync();
    // init_map -> add_blocker_type('r', (6, 7)): x, y in [-3, 3) around a normal(center, variance) point (Occupied_Grid_Map.py:46-62)
    // (a map-bank slot in prm.fixed_grid replaces the blocks and their draws)
    for (int b = 0; b < (prm.fixed_grid ? 0 : prm.num_blocks); b++) {
        double cx = 0.0, cy = 0.0;
        if (lane == 0) {
            py_randbelow(pym, pyi, 1u);  // random.randrange(len(shape)) with one shape
            double g2[2];
            for (int q = 0; q < 2; q++) {
                double v;
                if (has_gauss) { has_gauss = 0; v = gauss; gauss = 0.0; }
                else {
                    double x1, x2, r2;
                    do {
                        x1 = 2.0 * np_random_sample(npm, npi) - 1.0;
                        x2 = 2.0 * np_random_sample(npm, npi) - 1.0;
                        r2 = x1 * x1 + x2 * x2;
                    } while (r2 >= 1.0 || r2 == 0.0);
                    const double f = __builtin_sqrt(-2.0 * log(r2) / r2);
                    gauss = f * x1;
                    has_gauss = 1;
                    v = f * x2;
                }
                g2[q] = v;
            }
            cx = prm.center[0] + prm.variance * g2[0];
            cy = prm.center[1] + prm.variance * g2[1];
        }
        cx = __shfl(cx, 0); cy = __shfl(cy, 0);
        if (lane < 36) {
            const int x = lane / 6 - 3, y = lane % 6 - 3;
            const int px = py_round((double)x + cx), pyy = py_round((double)y + cy);
            if (px >= 0 && px < W && pyy >= 0 && pyy < H) grid[px * H + pyy] = 1;
        }
        wave_sync();
    }
    for (int i = lane; i < WH; i += WAVE) {  // inflate every obstacle cell by 2 (Chebyshev)
        const int x = i / H, y = i - x * H;
        uint8_t v = 0;
        for (int xx = x - 2; xx <= x + 2; xx++)
            for (int yy = y - 2; yy <= y + 2; yy++)
                if (xx >= 0 && xx < W && yy >= 0 && yy < H) v |= grid[xx * H + yy];
        infl[i] = v; infs[i] = v;
        bank[i] = v;
        st.grid[(size_t)env * WH + i] = grid[i];
    }
    wave_sync();
    // inner boundary (find_boundaries mode='inner', connectivity 1): obstacle cell with a free 4-neighbour, argwhere order
    int n_obs = 0;
    for (int base = 0; base < WH; base += WAVE) {
        const int i = base + lane;
        bool fr = false;
        int x = 0, y = 0;
        if (i < WH && grid[i]) {
            x = i / H; y = i - x * H;
            fr = (x > 0 && !grid[i - H]) || (x < W - 1 && !grid[i + H]) || (y > 0 && !grid[i - 1]) || (y < H - 1 && !grid[i + 1]);
        }
        const unsigned long long m = __ballot(fr);
        const int pos = n_obs + __popcll(m & ((1ull << lane) - 1ull));
        if (fr && pos < O) {
            obs_xy[((size_t)env * O + pos) * 2] = x; obs_xy[((size_t)env * O + pos) * 2 + 1] = y;
            if (o_state) { float4 v = make_float4((float)x, (float)y, 0.f, 0.f); ((float4 *)o_state)[(size_t)env * O + pos] = v; }
        }
        n_obs += __popcll(m);
    }
    for (int k = (n_obs < O ? n_obs : O) + lane; k < O; k += WAVE) {
        obs_xy[((size_t)env * O + k) * 2] = 0; obs_xy[((size_t)env * O + k) * 2 + 1] = 0;
        if (o_state) ((float4 *)o_state)[(size_t)env * O + k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (lane == 0) st.n_obs[env] = n_obs;  // > O is reported to the caller (the reference's buffer would not fit it either)
    {
        int tx = 0, ty = 0;
        if (lane == 0) {
            if (!draw_target_l0(W, H, infl, pym, pyi, tx, ty)) fail |= PE_STATUS_RESET_FAILED;
            st.target[2 * env] = tx; st.target[2 * env + 1] = ty;
        }
    }
    // init_defender (base_env.py:72-120)
    int placed = 0, ncells = 0, draws = 0;  // wave-uniform
    double *da = def_aos + (size_t)env * P * 4;
    while (placed < P) {
        double px = 0.0, pyv = 0.0;
        if (lane == 0) { px = np_random_sample(npm, npi) * (double)(W - 1); pyv = np_random_sample(npm, npi) * (double)(H - 1); }
        px = __shfl(px, 0); pyv = __shfl(pyv, 0);
        const int cxi = py_round(px), cyi = py_round(pyv);
        bool ok = false;
        if (++draws > PE_RESET_MAX_DRAWS) { ok = true; fail |= PE_STATUS_RESET_FAILED; }  // give up: keep this candidate
        else if (infl[cxi * H + cyi] == 0) {
            if (placed == 0) {
                ok = true;
            } else {
                double d = 0.0;
                if (lane < placed) d = norm2(px - dxy[2 * lane], pyv - dxy[2 * lane + 1]);
                const int collision = __popcll(__ballot(lane < placed && d < (double)prm.min_dist));
                const int connectivity = __popcll(__ballot(lane < placed && d < c.def_comm_range));
                ok = collision == 0 && connectivity > 0 && connectivity <= 2;
            }
        }
        if (ok) {  // wave-uniform
            bool seen = false;
            for (int k = 0; k < ncells; k++) seen = seen || (cells[2 * k] == cxi && cells[2 * k + 1] == cyi);
            wave_sync();
            if (lane == 0) {
                dxy[2 * placed] = px; dxy[2 * placed + 1] = pyv;
                da[placed * 4] = px; da[placed * 4 + 1] = pyv; da[placed * 4 + 2] = 0.0; da[placed * 4 + 3] = 0.0;
                if (!seen) { cells[2 * ncells] = cxi; cells[2 * ncells + 1] = cyi; }
            }
            if (lane < 25) {  // the new cell's 5 x 5 block (re-inflating the earlier cells as the reference does changes nothing)
                const int xx = cxi + lane / 5 - 2, yy = cyi + lane % 5 - 2;
                if (xx >= 0 && xx < W && yy >= 0 && yy < H) infl[xx * H + yy] = 1;
            }
            placed++;
            if (!seen) ncells++;
            wave_sync();
        }
    }
    // init_attacker (base_env.py:122-162), is_percepted=True: free cell within sensing range of a defender cell
    draws = 0;
    for (;;) {
        double px = 0.0, pyv = 0.0;
        if (lane == 0) { px = np_random_sample(npm, npi) * (double)(W - 1); pyv = np_random_sample(npm, npi) * (double)(H - 1); }
        px = __shfl(px, 0); pyv = __shfl(pyv, 0);
        if (++draws > PE_RESET_MAX_DRAWS) {
            if (lane == 0) {
                double *e = st.eva + (size_t)env * 4;
                e[0] = px; e[1] = pyv; e[2] = 0.0; e[3] = 0.0;
            }
            fail |= PE_STATUS_RESET_FAILED;
            break;
        }
        if (infl[py_round(px) * H + py_round(pyv)] != 0) continue;
        const bool hit = lane < ncells && norm2((double)cells[2 * lane] - px, (double)cells[2 * lane + 1] - pyv) < c.def_sen_range;
        if (__ballot(hit) != 0ull) {
            if (lane == 0) {
                double *e = st.eva + (size_t)env * 4;
                e[0] = px; e[1] = pyv; e[2] = 0.0; e[3] = 0.0;
            }
            break;
        }
    }
    wave_sync();
    // target tape: what init_target would return on the evader's next arrivals (pursuit_env.py:98-100); the stream position
    // before the tape is kept so that the next reset can give the unused draws back
    for (int i = lane; i < 624; i += WAVE) R.snap_mt[i] = pym[i];
    if (lane == 0) {
        R.snap_idx = pyi;
        R.has_tape = 1;
        int32_t *tape = st.tape + (size_t)env * c.tape_len * 2;
        for (int k = 0; k < c.tape_len; k++) {
            int tx, ty;
            if (!draw_target_l0(W, H, infs, pym, pyi, tx, ty)) fail |= PE_STATUS_RESET_FAILED;
            tape[2 * k] = tx; tape[2 * k + 1] = ty;
        }
        R.py_idx = pyi; R.np_idx = npi; R.has_gauss = has_gauss; R.gauss = gauss;
        R.status = fail;
    }
    wave_sync();
    for (int i = lane; i < 624; i += WAVE) { R.py_mt[i] = pym[i]; R.np_mt[i] = npm[i]; }
}

// seeding: random.seed(s) (init_by_array) and np.random.seed(s) (init_genrand), one thread per environment
__global__ void k_reset_seed(int N, const uint64_t *seeds, DevRng *rng) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    DevRng &R = rng[n];
    const uint64_t a = seeds[n];
    {
        uint32_t *mt = R.np_mt;
        mt[0] = (uint32_t)a;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        R.np_idx = 624;
    }
    {
        uint32_t *mt = R.py_mt;
        mt[0] = 19650218u;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        const uint32_t key[2] = {(uint32_t)a, (uint32_t)(a >> 32)};
        const int len = key[1] ? 2 : 1;
        int i = 1, j = 0;
        for (int k = 624; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            i++; j++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
            if (j >= len) j = 0;
        }
        for (int k = 623; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            i++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
        R.py_idx = 624;
    }
    R.snap_idx = 624; R.has_gauss = 0; R.has_tape = 0; R.status = 0; R.gauss = 0.0;
}

// map-bank resets: every environment holds the same grid, hence the same raser table -- built once, copied N - 1 times
__global__ void k_bcast_raser(size_t quads_per_env, uint4 *raser) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < quads_per_env) raser[(size_t)(blockIdx.y + 1) * quads_per_env + i] = raser[i];
}

int launch_episode_tables(const pe_config *cfg, const pe_state *st, const int32_t *d_obs, const void *rng, hipStream_t s, bool shared_map) {
    if (!st->raser) return PE_ERR_NULL;
    hipLaunchKernelGGL(k_build_bidx, dim3(st->N), dim3(256), 0, s, *cfg, *st, d_obs, (const DevRng *)rng);
    const size_t lds = align16(sizeof(int16_t) * cfg->W * cfg->H) + sizeof(uint32_t) * WAVE * (raser_row_words(cfg->O) + 1);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_build_raser, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const size_t quads = (size_t)cfg->W * cfg->H * raser_row_words(cfg->O) / 4;   // rows are multiples of 16 bytes
    if (shared_map && st->N > 1 && st->N <= 65536) {
        hipLaunchKernelGGL(k_build_raser, dim3(1), dim3(WAVE), lds, s, *cfg, *st);
        hipLaunchKernelGGL(k_bcast_raser, dim3((unsigned)((quads + 255) / 256), (unsigned)(st->N - 1)), dim3(256), 0, s, quads, (uint4 *)st->raser);
    } else {
        hipLaunchKernelGGL(k_build_raser, dim3(st->N), dim3(WAVE), lds, s, *cfg, *st);
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int pe_config_check(const pe_config *c) {
    if (!c) return PE_ERR_NULL;
    if (c->W < 2 || c->H < 2 || c->W > 255 || c->H > 255 || (c->W + 1) * (c->H + 1) >= 65535) return PE_ERR_BAD_CONFIG;
    if (c->P < 2 || c->P > PE_MAX_P) return PE_ERR_BAD_CONFIG;  // the reference's communicate() needs P >= 2
    if (c->O < 4 || (c->O & 3) || c->O > 32767) return PE_ERR_BAD_CONFIG;
    if (c->num_beams < 1 || c->num_beams > PE_MAX_BEAMS || c->lidar_radius < 0) return PE_ERR_BAD_CONFIG;
    if (c->difficulty < 1 || c->extend_dis < 0 || c->extend_dis > 8 || c->evader_view < 0) return PE_ERR_BAD_CONFIG;
    if (c->tape_len < 1 || c->max_path < c->difficulty + 2) return PE_ERR_BAD_CONFIG;
    if (lds_layout(*c, true, true, nullptr, nullptr) > 160 * 1024) return PE_ERR_BAD_CONFIG;
    if (align16(sizeof(int16_t) * c->W * c->H) + sizeof(uint32_t) * WAVE * (raser_row_words(c->O) + 1) > 160 * 1024) return PE_ERR_BAD_CONFIG;
    return 0;
}

int64_t pe_tick_lds_bytes(const pe_config *cfg, int32_t with_replan) {
    return (int64_t)lds_layout(*cfg, true, with_replan != 0, nullptr, nullptr);
}

int64_t pe_reset_state_bytes(const pe_config *cfg, int32_t N) {
    if (!cfg || N < 1) return 0;
    return (int64_t)N * ((int64_t)sizeof(DevRng) + (int64_t)cfg->W * cfg->H);
}

int pe_env_reset_seed(const pe_config *cfg, int32_t N, const uint64_t *seeds, void *reset_state, void *stream) {
    if (!cfg || !seeds || !reset_state || N < 1) return PE_ERR_NULL;
    hipStream_t s = (hipStream_t)stream;
    uint64_t *d_seeds = nullptr;
    hipError_t e = hipMallocAsync((void **)&d_seeds, (size_t)N * sizeof(uint64_t), s);
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(d_seeds, seeds, (size_t)N * sizeof(uint64_t), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync((char *)reset_state + (size_t)N * sizeof(DevRng), 0, (size_t)N * cfg->W * cfg->H, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_reset_seed, dim3((N + 63) / 64), dim3(64), 0, s, (int)N, (const uint64_t *)d_seeds, (DevRng *)reset_state);
    e = hipStreamSynchronize(s);  // the caller's seed array may go away
    if (e != hipSuccess) return (int)e;
    e = hipFreeAsync(d_seeds, s);
    return e != hipSuccess ? (int)e : (int)hipGetLastError();
}

int pe_env_reset(const pe_config *cfg, const pe_state *st, const pe_reset_params *prm, void *reset_state, int32_t first, float *o_state,
                 int32_t reset_rn, void *stream) {
    if (!cfg || !st || !prm || !reset_state) return PE_ERR_NULL;
    int rc = pe_config_check(cfg);
    if (rc) return rc;
    if (prm->num_blocks < 0 || prm->min_dist < 0) return PE_ERR_BAD_CONFIG;
    const size_t lds = reset_lds_bytes(*cfg);
    if (lds > 160 * 1024) return PE_ERR_BAD_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    const size_t N = st->N, P = cfg->P;
    hipError_t e;
#define PE_TRY(x) do { e = (x); if (e != hipSuccess) return (int)e; } while (0)
    if (lds > 48 * 1024) PE_TRY(hipFuncSetAttribute((const void *)k_reset, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int32_t *d_obs = nullptr;
    double *d_def = nullptr;
    PE_TRY(hipMallocAsync((void **)&d_obs, N * cfg->O * 2 * sizeof(int32_t), s));
    PE_TRY(hipMallocAsync((void **)&d_def, N * P * 4 * sizeof(double), s));
    DevRng *rng = (DevRng *)reset_state;
    uint8_t *bank = (uint8_t *)reset_state + N * sizeof(DevRng);
    hipLaunchKernelGGL(k_reset, dim3(N), dim3(WAVE), lds, s, *cfg, *st, *prm, rng, bank, (int)first, d_obs, d_def, o_state);
    { const int rc2 = launch_episode_tables(cfg, st, d_obs, rng, s, prm->fixed_grid != nullptr); if (rc2) return rc2; }
    int tot = (int)(N * P * 4);
    hipLaunchKernelGGL(k_def_aos_to_soa, dim3((tot + 255) / 256), dim3(256), 0, s, (int)N, (int)P, (const double *)d_def, st->def);
    if (reset_rn) PE_TRY(hipMemsetAsync(st->rn, 0, N * (1 + 2 * P) * sizeof(double), s));
    PE_TRY(hipFreeAsync(d_obs, s));
    PE_TRY(hipFreeAsync(d_def, s));
#undef PE_TRY
    return (int)hipGetLastError();
}

int pe_env_load(const pe_config *cfg, const pe_state *st, const pe_host_init *h, void *stream) {
    if (!cfg || !st || !h) return PE_ERR_NULL;
    int rc = pe_config_check(cfg);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t N = st->N, WH = (size_t)cfg->W * cfg->H, P = cfg->P;
    hipError_t e;
#define PE_TRY(x) do { e = (x); if (e != hipSuccess) return (int)e; } while (0)
    PE_TRY(hipMemcpyAsync(st->grid, h->grid, N * WH, hipMemcpyHostToDevice, s));
    PE_TRY(hipMemcpyAsync(st->n_obs, h->n_obs, N * sizeof(int32_t), hipMemcpyHostToDevice, s));
    PE_TRY(hipMemcpyAsync(st->eva, h->eva, N * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    PE_TRY(hipMemcpyAsync(st->target, h->target, N * 2 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    PE_TRY(hipMemcpyAsync(st->tape, h->tape, N * cfg->tape_len * 2 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    // staging for the two layout-changing uploads lives in the path / rn-free scratch: allocate from the async pool
    int32_t *d_obs = nullptr;
    double *d_def = nullptr;
    PE_TRY(hipMallocAsync((void **)&d_obs, N * cfg->O * 2 * sizeof(int32_t), s));
    PE_TRY(hipMallocAsync((void **)&d_def, N * P * 4 * sizeof(double), s));
    PE_TRY(hipMemcpyAsync(d_obs, h->obs_xy, N * cfg->O * 2 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    PE_TRY(hipMemcpyAsync(d_def, h->def, N * P * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    { const int rc2 = launch_episode_tables(cfg, st, d_obs, nullptr, s); if (rc2) return rc2; }
    int tot = (int)(N * P * 4);
    hipLaunchKernelGGL(k_def_aos_to_soa, dim3((tot + 255) / 256), dim3(256), 0, s, (int)N, (int)P, d_def, st->def);
    if (h->reset_rn) PE_TRY(hipMemsetAsync(st->rn, 0, N * (1 + 2 * P) * sizeof(double), s));
    PE_TRY(hipFreeAsync(d_obs, s));
    PE_TRY(hipFreeAsync(d_def, s));
#undef PE_TRY
    return (int)hipGetLastError();
}

int pe_env_observe(const pe_config *cfg, const pe_state *st, const pe_obs_out *out, void *stream) {
    if (!cfg || !st || !out) return PE_ERR_NULL;
    return launch<false, true, false, false>(cfg, st, nullptr, nullptr, out, stream);
}

int pe_evader_step(const pe_config *cfg, const pe_state *st, int32_t may_replan, void *stream) {
    if (!cfg || !st) return PE_ERR_NULL;
    return may_replan ? launch<false, false, true, true>(cfg, st, nullptr, nullptr, nullptr, stream)
                      : launch<false, false, true, false>(cfg, st, nullptr, nullptr, nullptr, stream);
}

int pe_env_demon(const pe_config *cfg, const pe_state *st, const double *unit_dirs, int32_t *actions, void *stream) {
    if (!cfg || !st || !unit_dirs || !actions || !st->def || !st->eva) return PE_ERR_NULL;
    DemonDirs dd;
    memcpy(dd.d, unit_dirs, sizeof dd.d);
    const int n = st->N * cfg->P;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_demon, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, st->N, cfg->P, st->def, st->eva, dd, actions);
    return (int)hipGetLastError();
}

int pe_env_step(const pe_config *cfg, const pe_state *st, const int32_t *actions, const pe_step_out *out, void *stream) {
    if (!cfg || !st || !actions || !out) return PE_ERR_NULL;
    return launch<true, false, false, false>(cfg, st, actions, out, nullptr, stream);
}

int pe_env_tick(const pe_config *cfg, const pe_state *st, const int32_t *actions, const pe_step_out *sout, const pe_obs_out *oout,
                int32_t may_replan, void *stream) {
    if (!cfg || !st || !actions || !sout || !oout) return PE_ERR_NULL;
    return may_replan ? launch<true, true, true, true>(cfg, st, actions, sout, oout, stream)
                      : launch<true, true, true, false>(cfg, st, actions, sout, oout, stream);
}

int pe_env_step_observe(const pe_config *cfg, const pe_state *st, const int32_t *actions, const pe_step_out *sout, const pe_obs_out *oout,
                        void *stream) {
    if (!cfg || !st || !actions || !sout || !oout) return PE_ERR_NULL;
    return launch<true, true, false, false>(cfg, st, actions, sout, oout, stream);
}

int pe_astar_batch(int32_t W, int32_t H, int32_t n, const uint8_t *obs, const int32_t *sg, int16_t *out_path, int32_t *out_len,
                   int32_t max_path, void *stream) {
    pe_config c;
    memset(&c, 0, sizeof c);
    c.W = W; c.H = H; c.P = 2; c.O = 4;
    size_t lds = lds_layout(c, false, true, nullptr, nullptr);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_astar, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_astar, dim3(n), dim3(WAVE), lds, (hipStream_t)stream, W, H, n, obs, sg, out_path, out_len, max_path);
    return (int)hipGetLastError();
}

// diagnostic: device f64 norm / divide / round / constant-divisor divide (out [5][n]) against the host (tests only)
int pe_diag_norm2(int32_t n, const double *a, const double *b, double *out, double c0, double c1, void *stream) {
    hipLaunchKernelGGL(k_diag_norm2, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, a, b, out, c0, div_const_reciprocal(c0), c1,
                       div_const_reciprocal(c1));
    return (int)hipGetLastError();
}

// diagnostics (host only, no GPU needed): the comparison thresholds / reciprocals launch() hands to the tick kernel
double pe_diag_sq_threshold(double r, int32_t strict) { return sq_threshold(r, strict != 0); }
double pe_diag_div_reciprocal(double b) { return div_const_reciprocal(b); }

const char *pe_error_string(int code) {
    if (code == PE_ERR_BAD_CONFIG) return "pe_env: configuration outside kernel limits";
    if (code == PE_ERR_NULL) return "pe_env: null argument";
    if (code == PE_ERR_RESET_FAILED) return "pe_env: a placement loop of the episode reset gave up after PE_RESET_MAX_DRAWS draws (map too crowded for this configuration)";
    return hipGetErrorString((hipError_t)code);
}

}  // extern "C"
