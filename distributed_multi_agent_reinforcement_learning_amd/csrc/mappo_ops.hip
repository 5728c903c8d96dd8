// mappo_ops.hip -- fused fp32 ops of the MAPPO rollout/update path for MI355X (gfx950).  C ABI: include/mappo_ops.h.
//
// dhgn_msg_agg_{fwd,bwd}: the relation message ReLU(W (p_i - q_j) + b) and its adjacency-weighted mean are fused so
// the (rows, P, K, E) message tensor (865 MB per reference mini-batch) never exists; backward recomputes the
// pre-activation.  Lane = output feature; a row's positions / adjacency / neighbour coordinates are wave-uniform and live in
// registers read back with v_readlane (or come through scalar loads): no LDS, no barriers (section "relation message + mean").
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "mappo_ops.h"
#include "mappo_ops_diag.h"
#include "sb_common.hpp"

namespace {

constexpr int MAX_P = 16;

// d_j = W[:, :4] q_j with a fixed evaluation order (explicit fused multiply-adds: every loop variant rounds identically)
__device__ __forceinline__ float msg_dot4(const float *w, const float4 &qv) {
    return __builtin_fmaf(w[3], qv.w, __builtin_fmaf(w[2], qv.z, __builtin_fmaf(w[1], qv.y, w[0] * qv.x)));
}

// The pre-activation of the relation message is evaluated as z_ij = c_i - d_j with c_i = b + W[:, :4] p_i (+ W[:, 4:8] (p_i - e))
// and d_j = W[:, :4] q_j; lane = output feature, one accumulator per agent in registers (PT = P rounded up to 8 or 16).
// ---- relation message + mean (dhgn_msg_agg_*) -------------------------------------------------------------------------
// No LDS, no barriers.  Everything a row needs besides the per-lane feature weights is wave-uniform (positions, adjacency,
// neighbour coordinates): each of these small arrays is fetched with ONE coalesced load (lane l holds element l) and read back
// with v_readlane, or -- the obstacle coordinates, K x 16 bytes shared by the rows of an episode -- through uniform addresses
// (scalar loads); the loads of row r+1 are issued before row r is computed, and every wave walks its rows on its own.
// A packed LiDAR adjacency row (MO_ADJ_BITS: ~18 of 176 columns set per row, ~2 per agent) is walked bit by bit per agent
// (s_ff1), so the loop length is the number of edges, not K; float adjacencies and the all-ones forms keep the column loop
// with d_j shared by the agents.  Every accumulator receives its additions in ascending j in all forms, so packed and float
// adjacency, single and paired launches give bit-identical forward results.  Rows are dealt to workgroups in contiguous
// chunks (rows of one episode share their obstacle set: scalar-cache hits).
//   QS: the neighbour coordinates of a row fit one register (4 K <= 64);  AS: so does its adjacency (float: P K <= 64;
//   packed: P MO_ADJ_ROW_WORDS(K) <= 128, two registers); otherwise these are read through uniform addresses.
struct MsgDims {
    int R, P, K, E, din, q_div, adj_mode, rpb;  // rpb: rows per workgroup
    int64_t p_rs, q_rs, e_rs, adj_rs, o_is;
};

__device__ __forceinline__ float rl_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ uint32_t rl_u(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

struct MsgRow {
    float vp, ve, vq;   // p row [4 P], e row [4], q row [4 K] (QS)
    uint32_t va0, va1;  // adjacency (AS): float bits [P][K], or packed words [P][RWK] (va1: words 64..127)
    int kv;             // critic: number of neighbours averaged over
};

template <bool QS, bool AS>
__device__ __forceinline__ MsgRow msgw_load(const MsgDims &d, int r, const float *__restrict__ p, const float *__restrict__ q,
                                            const float *__restrict__ e, const void *__restrict__ adj, const int32_t *__restrict__ kvalid,
                                            bool use_kvalid) {
    const int lane = threadIdx.x & 63, qr = r / d.q_div;
    MsgRow m;
    m.vp = lane < 4 * d.P ? p[(size_t)r * d.p_rs + lane] : 0.f;
    m.ve = (d.din == 8 && lane < 4) ? e[(size_t)r * d.e_rs + lane] : 0.f;
    m.vq = 0.f;
    if (QS) m.vq = lane < 4 * d.K ? q[(size_t)qr * d.q_rs + lane] : 0.f;
    m.va0 = m.va1 = 0u;
    if (AS) {
        if (d.adj_mode == MO_ADJ_TENSOR) {
            m.va0 = lane < d.P * d.K ? __float_as_uint(((const float *)adj)[(size_t)r * d.adj_rs + lane]) : 0u;
        } else if (d.adj_mode == MO_ADJ_BITS) {
            const uint32_t *bw = (const uint32_t *)adj + (size_t)r * d.adj_rs;
            const int n = d.P * MO_ADJ_ROW_WORDS(d.K);
            m.va0 = lane < n ? bw[lane] : 0u;
            m.va1 = lane + 64 < n ? bw[lane + 64] : 0u;
        }
    }
    m.kv = use_kvalid ? kvalid[qr] : d.K;
    return m;
}

template <bool QS>
__device__ __forceinline__ float4 msgw_q(const MsgRow &m, const float4 *__restrict__ q4, int j) {
    if (QS) return make_float4(rl_f(m.vq, 4 * j), rl_f(m.vq, 4 * j + 1), rl_f(m.vq, 4 * j + 2), rl_f(m.vq, 4 * j + 3));
    return q4[j];
}
template <bool AS>
__device__ __forceinline__ uint32_t msgw_word(const MsgRow &m, const uint32_t *__restrict__ bw, int idx) {
    if (AS) return idx < 64 ? rl_u(m.va0, idx) : rl_u(m.va1, idx - 64);
    return bw[idx];
}

// non-zero pattern of a float adjacency held in va0 (lane i K + j): nz; col = the bits i K of all agents; zero_one: every
// non-zero entry is exactly 1.0f (the environments' adjacencies), so weights need no read-back and a row's L1 norm is its
// popcount (a sum of ones is exact in any order: the same value as the sequential sum)
struct MsgMask { uint64_t nz, col; bool zero_one; };
__device__ __forceinline__ MsgMask msgw_mask(const MsgRow &m, int P, int K) {
    const float a = __uint_as_float(m.va0);
    MsgMask k;
    k.nz = __ballot(a != 0.f);
    k.zero_one = __ballot(a != 0.f && a != 1.f) == 0ull;
    k.col = 0ull;
    for (int i = 0; i < P; i++) k.col |= 1ull << (i * K);
    return k;
}
__device__ __forceinline__ float msgw_row_norm(const MsgRow &m, const MsgMask &k, int i, int K) {
    if (k.zero_one) return (float)__popcll((k.nz >> (i * K)) & ((K >= 64) ? ~0ull : ((1ull << K) - 1ull)));
    float s = 0.f;
    for (int j = 0; j < K; j++) s += fabsf(__uint_as_float(rl_u(m.va0, i * K + j)));
    return s;
}

// EV output features per lane (f = lane + e * blockDim): with E = 128 and EV = 2 ONE wave owns a row, so the wave-uniform part of
// the work (read-backs, masks, bit walks, address arithmetic -- most of the instructions of a row) is issued once instead of twice
template <int EV>
struct FV {
    float v[EV];
    __device__ __forceinline__ FV() {}
    __device__ __forceinline__ FV(float s) {
#pragma unroll
        for (int e = 0; e < EV; e++) v[e] = s;
    }
};
#define FV_BIN(NAME, EXPR)                                                                                \
    template <int EV> __device__ __forceinline__ FV<EV> NAME(const FV<EV> &a, const FV<EV> &b) {         \
        FV<EV> r;                                                                                         \
        _Pragma("unroll") for (int e = 0; e < EV; e++) { const float x = a.v[e], y = b.v[e]; r.v[e] = EXPR; } \
        return r;                                                                                         \
    }                                                                                                     \
    template <int EV> __device__ __forceinline__ FV<EV> NAME(const FV<EV> &a, float y) {                  \
        FV<EV> r;                                                                                         \
        _Pragma("unroll") for (int e = 0; e < EV; e++) { const float x = a.v[e]; r.v[e] = EXPR; }          \
        return r;                                                                                         \
    }
FV_BIN(operator+, x + y)
FV_BIN(operator-, x - y)
FV_BIN(operator*, x *y)
FV_BIN(fv_pos, (x > 0.f) ? y : 0.f)  // y where x > 0
#undef FV_BIN
template <int EV> __device__ __forceinline__ FV<EV> fv_fma(const FV<EV> &a, float b, const FV<EV> &c) {
    FV<EV> r;
#pragma unroll
    for (int e = 0; e < EV; e++) r.v[e] = __builtin_fmaf(a.v[e], b, c.v[e]);
    return r;
}
template <int EV> __device__ __forceinline__ FV<EV> fv_relu(const FV<EV> &a) {
    FV<EV> r;
#pragma unroll
    for (int e = 0; e < EV; e++) r.v[e] = fmaxf(a.v[e], 0.f);
    return r;
}
// two features per lane as ONE 64-bit value: the compiler then keeps the pair in an aligned register pair and issues the packed fp32
// instructions (v_pk_add_f32 / v_pk_fma_f32: two lanes' worth per issue) for the differences as well as the sums
typedef float msg_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ msg_f2 fv2(const FV<2> &a) { return (msg_f2){a.v[0], a.v[1]}; }
__device__ __forceinline__ FV<2> fv2(msg_f2 x) { FV<2> r; r.v[0] = x.x; r.v[1] = x.y; return r; }
__device__ __forceinline__ FV<2> operator+(const FV<2> &a, const FV<2> &b) { return fv2(fv2(a) + fv2(b)); }
__device__ __forceinline__ FV<2> operator-(const FV<2> &a, const FV<2> &b) { return fv2(fv2(a) - fv2(b)); }
__device__ __forceinline__ FV<2> fv_relu(const FV<2> &a) { return fv2(__builtin_elementwise_max(fv2(a), (msg_f2){0.f, 0.f})); }
template <int EV> __device__ __forceinline__ FV<EV> fv_load(const float *p, int f, int stride) {
    FV<EV> r;
#pragma unroll
    for (int e = 0; e < EV; e++) r.v[e] = p[f + e * stride];
    return r;
}
template <int EV> __device__ __forceinline__ void fv_store(float *p, int f, int stride, const FV<EV> &a) {
#pragma unroll
    for (int e = 0; e < EV; e++) p[f + e * stride] = a.v[e];
}
// d_j = W[:, :4] q_j, fixed evaluation order
template <int EV> __device__ __forceinline__ FV<EV> msg_dot4v(const FV<EV> (&w)[8], const float4 &qv) {
    return fv_fma(w[3], qv.w, fv_fma(w[2], qv.z, fv_fma(w[1], qv.y, w[0] * qv.x)));
}
template <int EV> __device__ __forceinline__ void msgw_weights(const MsgDims &d, const float *__restrict__ W, const float *__restrict__ b, int f,
                                                               FV<EV> (&w)[8], FV<EV> &bias, int fs) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        w[k] = FV<EV>(0.f);
        if (k < d.din) {
#pragma unroll
            for (int e = 0; e < EV; e++) w[k].v[e] = W[(size_t)(f + e * fs) * d.din + k];
        }
    }
    bias = fv_load<EV>(b, f, fs);
}

template <int PT, int EV>
// (i0: the first of the PT agents this wave serves -- a row's agents may be divided between waves, see launch_msgw3)
__device__ __forceinline__ void msgw_center(const MsgDims &d, const MsgRow &m, const FV<EV> (&w)[8], const FV<EV> &bias, FV<EV> (&c)[PT], int i0 = 0) {
    const float e0 = rl_f(m.ve, 0), e1 = rl_f(m.ve, 1), e2 = rl_f(m.ve, 2), e3 = rl_f(m.ve, 3);
#pragma unroll
    for (int i = 0; i < PT; i++) {
        c[i] = FV<EV>(0.f);
        if (i0 + i < d.P) {
            const int ia = i0 + i;
            const float p0 = rl_f(m.vp, 4 * ia), p1 = rl_f(m.vp, 4 * ia + 1), p2 = rl_f(m.vp, 4 * ia + 2), p3 = rl_f(m.vp, 4 * ia + 3);
            c[i] = fv_fma(w[3], p3, fv_fma(w[2], p2, fv_fma(w[1], p1, fv_fma(w[0], p0, bias))));
            if (d.din == 8) c[i] = fv_fma(w[7], p3 - e3, fv_fma(w[6], p2 - e2, fv_fma(w[5], p1 - e1, fv_fma(w[4], p0 - e0, c[i]))));
        }
    }
}

// the actor's aggregate of one row: acc[i] = sum_j adj_ij relu(c_i - d_j), inv[i] = 1 / max(sum_j |adj_ij|, 1e-12), for agents i0 .. i0 + PT - 1
template <int PT, bool QS, bool AS, int EV>
__device__ __forceinline__ void msgw_actor_row(const MsgDims &d, int r, const MsgRow &m, const float4 *__restrict__ q4,
                                               const void *__restrict__ adj, const FV<EV> (&w)[8], const FV<EV> (&c)[PT], FV<EV> (&acc)[PT],
                                               float (&inv)[PT], int i0 = 0) {
    const int P = d.P, K = d.K;
    if (d.adj_mode == MO_ADJ_BITS) {
        const uint32_t *__restrict__ bw = (const uint32_t *)adj + (size_t)r * d.adj_rs;
        const int nw = (K + 31) >> 5, RWK = MO_ADJ_ROW_WORDS(K);
#pragma unroll
        for (int i = 0; i < PT; i++) {
            acc[i] = FV<EV>(0.f);
            inv[i] = 0.f;
            if (i0 + i < P) {
                int cnt = 0;
                for (int wd = 0; wd < nw; wd++) {
                    uint32_t bits = msgw_word<AS>(m, bw, (i0 + i) * RWK + wd);
                    if (wd == (K >> 5)) bits &= (1u << (K & 31)) - 1u;
                    cnt += __popc(bits);
                    while (bits) {
                        const int j = (wd << 5) + __builtin_ctz(bits);
                        bits &= bits - 1u;
                        acc[i] = acc[i] + fv_relu(c[i] - msg_dot4v(w, msgw_q<QS>(m, q4, j)));
                    }
                }
                inv[i] = 1.f / fmaxf((float)cnt, 1e-12f);
            }
        }
    } else if (AS) {  // MO_ADJ_TENSOR held in one register: the non-zero pattern as a 64-bit mask (bit i K + j)
        const MsgMask k = msgw_mask(m, P, K);
        uint64_t col = 0ull;                   // bit i K of this wave's agents
#pragma unroll
        for (int i = 0; i < PT; i++) {
            acc[i] = FV<EV>(0.f);
            inv[i] = 0.f;
            if (i0 + i < P) {
                inv[i] = 1.f / fmaxf(msgw_row_norm(m, k, i0 + i, K), 1e-12f);
                col |= 1ull << ((i0 + i) * K);
            }
        }
        for (int j = 0; j < K; j++) {
            const uint64_t cm = (k.nz >> j) & col;  // bit i K: agent i sees neighbour j
            if (cm == 0ull) continue;
            const FV<EV> dj = msg_dot4v(w, msgw_q<QS>(m, q4, j));
#pragma unroll
            for (int i = 0; i < PT; i++)
                if (i0 + i < P && ((cm >> ((i0 + i) * K)) & 1ull)) {
                    const float aij = k.zero_one ? 1.f : __uint_as_float(rl_u(m.va0, (i0 + i) * K + j));
                    acc[i] = fv_fma(fv_relu(c[i] - dj), aij, acc[i]);
                }
        }
    } else {  // MO_ADJ_TENSOR read through uniform addresses
        const float *__restrict__ ar = (const float *)adj + (size_t)r * d.adj_rs + (size_t)i0 * K;
        const int pn = P - i0 < PT ? P - i0 : PT;      // this wave's agents
#pragma unroll
        for (int i = 0; i < PT; i++) {
            acc[i] = FV<EV>(0.f);
            inv[i] = 0.f;
            if (i < pn) {
                float s = 0.f;
                for (int j = 0; j < K; j++) s += fabsf(ar[i * K + j]);
                inv[i] = 1.f / fmaxf(s, 1e-12f);
            }
        }
        for (int j = 0; j < K; j++) {
            bool any = false;
            for (int i = 0; i < pn; i++) any |= ar[i * K + j] != 0.f;
            if (!any) continue;
            const FV<EV> dj = msg_dot4v(w, msgw_q<QS>(m, q4, j));
#pragma unroll
            for (int i = 0; i < PT; i++)
                if (i < pn) {
                    const float aij = ar[i * K + j];
                    if (aij != 0.f) acc[i] = fv_fma(fv_relu(c[i] - dj), aij, acc[i]);
                }
        }
    }
}

// the critic's aggregate: ones over the first kv neighbours (coordinates fetched four columns at a time)
template <int PT, bool QS, int EV>
__device__ __forceinline__ void msgw_ones_row(int kv, const MsgRow &m, const float4 *__restrict__ q4, const FV<EV> (&w)[8], const FV<EV> (&c)[PT],
                                              FV<EV> (&acc)[PT]) {
#pragma unroll
    for (int i = 0; i < PT; i++) acc[i] = FV<EV>(0.f);
    int j = 0;
    if (!QS) {
        for (; j + 4 <= kv; j += 4) {
            const float4 qa = q4[j], qb = q4[j + 1], qc = q4[j + 2], qd = q4[j + 3];
            const FV<EV> da = msg_dot4v(w, qa), db = msg_dot4v(w, qb), dc = msg_dot4v(w, qc), dd = msg_dot4v(w, qd);
#pragma unroll
            for (int i = 0; i < PT; i++) {
                acc[i] = acc[i] + fv_relu(c[i] - da);
                acc[i] = acc[i] + fv_relu(c[i] - db);
                acc[i] = acc[i] + fv_relu(c[i] - dc);
                acc[i] = acc[i] + fv_relu(c[i] - dd);
            }
        }
    }
    for (; j < kv; j++) {
        const FV<EV> dj = msg_dot4v(w, msgw_q<QS>(m, q4, j));
#pragma unroll
        for (int i = 0; i < PT; i++) acc[i] = acc[i] + fv_relu(c[i] - dj);
    }
}

// one relation for this workgroup's rows.  out_c == nullptr: one network (d.adj_mode decides);  else actor (out) and critic
// (out_c; c_valid: over the first kvalid[row] neighbours) from the same messages.
template <int PT, bool QS, bool AS, int EV>
__device__ __forceinline__ void msgw_fwd_rows(const MsgDims &d, const float *__restrict__ p, const float *__restrict__ q,
                                              const float *__restrict__ e, const void *__restrict__ adj, const int32_t *__restrict__ kvalid,
                                              const float *__restrict__ W, const float *__restrict__ b, float *__restrict__ out,
                                              float *__restrict__ out_c, bool c_valid, int i0 = 0) {
    const int P = d.P, f = threadIdx.x, fs = blockDim.x;
    FV<EV> w[8], bias;
    msgw_weights<EV>(d, W, b, f, w, bias, fs);
    const int r0 = blockIdx.x * d.rpb, r1 = min(d.R, r0 + d.rpb);
    if (r0 >= r1) return;
    const bool use_kv = out_c != nullptr ? c_valid : d.adj_mode == MO_ADJ_VALID;
    MsgRow m = msgw_load<QS, AS>(d, r0, p, q, e, adj, kvalid, use_kv);
    for (int r = r0; r < r1; r++) {
        const MsgRow nxt = msgw_load<QS, AS>(d, r + 1 < r1 ? r + 1 : r, p, q, e, adj, kvalid, use_kv);
        const float4 *__restrict__ q4 = (const float4 *)(q + (size_t)(r / d.q_div) * d.q_rs);
        FV<EV> c[PT], acc[PT];
        float inv[PT];
        msgw_center<PT, EV>(d, m, w, bias, c, i0);
        if (out_c != nullptr) {
            FV<EV> acv[PT];
            msgw_ones_row<PT, QS, EV>(m.kv, m, q4, w, c, acv);
            const float inv_c = 1.f / fmaxf((float)m.kv, 1e-12f);
#pragma unroll
            for (int i = 0; i < PT; i++)
                if (i0 + i < P) fv_store<EV>(out_c + ((size_t)r * P + i0 + i) * d.o_is, f, fs, acv[i] * inv_c);
            msgw_actor_row<PT, QS, AS, EV>(d, r, m, q4, adj, w, c, acc, inv, i0);
        } else if (d.adj_mode == MO_ADJ_TENSOR || d.adj_mode == MO_ADJ_BITS) {
            msgw_actor_row<PT, QS, AS, EV>(d, r, m, q4, adj, w, c, acc, inv, i0);
        } else {
            msgw_ones_row<PT, QS, EV>(m.kv, m, q4, w, c, acc);
            const float iv = 1.f / fmaxf((float)m.kv, 1e-12f);
#pragma unroll
            for (int i = 0; i < PT; i++) inv[i] = iv;
        }
#pragma unroll
        for (int i = 0; i < PT; i++)
            if (i0 + i < P) fv_store<EV>(out + ((size_t)r * P + i0 + i) * d.o_is, f, fs, acc[i] * inv[i]);
        m = nxt;
    }
}

template <int PT, bool QS, bool AS, int EV>
__global__ __launch_bounds__(256) void k_msgw_fwd(MsgDims d, const float *__restrict__ p, const float *__restrict__ q, const float *__restrict__ e,
                           const void *__restrict__ adj, const int32_t *__restrict__ kvalid, const float *__restrict__ W,
                           const float *__restrict__ b, float *__restrict__ out) {
    msgw_fwd_rows<PT, QS, AS, EV>(d, p, q, e, adj, kvalid, W, b, out, nullptr, false);
}

// the three relations of DHGN.encoder (defender, evader: small neighbour sets; obstacle: large) of one network (out_c == nullptr)
// or of actor and critic together, one launch.  (kvalid2 is handed to all three calls: a literal null there crashes this
// hipcc's inliner; relations 0 and 1 never read it.)
template <int PT, bool S01, bool AS2, int EV>
__global__ __launch_bounds__(256) void k_msgw3_fwd(MsgDims d0, MsgDims d1, MsgDims d2, const float *__restrict__ p, const float *__restrict__ q0,
                            const float *__restrict__ e0, const void *__restrict__ adj0, const float *__restrict__ W0,
                            const float *__restrict__ b0, const float *__restrict__ q1, const void *__restrict__ adj1,
                            const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ q2,
                            const void *__restrict__ adj2, const int32_t *__restrict__ kvalid2, const float *__restrict__ W2,
                            const float *__restrict__ b2, float *__restrict__ out, float *__restrict__ out_c, int c_valid, int c_rel2,
                            const float *__restrict__ Wp, int64_t wp_rs, const float *__restrict__ bp, float *__restrict__ pos_a,
                            float *__restrict__ pos_c, int64_t pos_ld) {
    const int E = d0.E;
    // blockIdx.y: which PT agents of the row this wave serves (launch_msgw3 divides a row between waves when there are few rows: a
    // wave's work on a row is one long dependent stream -- 45 us per row at the rollout's shapes whatever the row count)
    const int i0 = blockIdx.y * PT;
    msgw_fwd_rows<PT, S01, S01, EV>(d0, p, q0, e0, adj0, kvalid2, W0, b0, out, out_c, false, i0);
    msgw_fwd_rows<PT, S01, S01, EV>(d1, p, q1, e0, adj1, kvalid2, W1, b1, out + E, out_c ? out_c + E : nullptr, false, i0);
    // c_rel2 == 0 (the update): the critic's obstacle relation is left to the sorted all-ones kernel (k_msg_ones_sorted_fwd)
    msgw_fwd_rows<PT, false, AS2, EV>(d2, p, q2, e0, adj2, kvalid2, W2, b2, out + 2 * E, (out_c && c_rel2) ? out_c + 2 * E : nullptr, c_valid != 0, i0);
    if (pos_a != nullptr) {
        // the position part of DHGN's semantic layer, bp + Wp p_i (Wp = the first four input columns, :284-303), for the same rows:
        // the addend the embedding part of that layer accumulates into; identical for actor and critic, written to both
        const int P = d0.P, f = threadIdx.x, fs = blockDim.x, lane = threadIdx.x & 63;
        FV<EV> wp[4], bias = fv_load<EV>(bp, f, fs);
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int e = 0; e < EV; e++) wp[k].v[e] = Wp[(size_t)(f + e * fs) * wp_rs + k];
        const int r0 = blockIdx.x * d0.rpb, r1 = min(d0.R, r0 + d0.rpb);
        for (int r = r0; r < r1; r++) {
            const float vp = lane < 4 * P ? p[(size_t)r * d0.p_rs + lane] : 0.f;
#pragma unroll
            for (int i = 0; i < PT; i++)
                if (i0 + i < P) {
                    const int ia = i0 + i;
                    const FV<EV> v = fv_fma(wp[3], rl_f(vp, 4 * ia + 3), fv_fma(wp[2], rl_f(vp, 4 * ia + 2), fv_fma(wp[1], rl_f(vp, 4 * ia + 1),
                                            fv_fma(wp[0], rl_f(vp, 4 * ia), bias))));
                    fv_store<EV>(pos_a + ((size_t)r * P + ia) * pos_ld, f, fs, v);
                    if (pos_c != nullptr) fv_store<EV>(pos_c + ((size_t)r * P + ia) * pos_ld, f, fs, v);
                }
        }
    }
}

// backward: per-thread partial sums over the workgroup's rows -> partials [gridDim.x][din + 1][E] (reduced by
// k_msg_agg_bwd_reduce).  dW[:, k<4] = sum_i G_i p_i[k] - sum_j H_j q_j[k] with G_i = sum_j g_ij, H_j = sum_i g_ij,
// g_ij = [z_ij > 0] abar_ij gout_i ; dW[:, 4+k] = sum_i G_i (p_i - e)[k] ; db = sum_i G_i.  For MO_ADJ_BITS the
// neighbour-coordinate term is summed edge by edge instead of column by column (fp32 reassociation only).
template <int PT, bool QS, bool AS, int EV, bool PAIR>
__global__ __launch_bounds__(256) void k_msgw_bwd(MsgDims d, const float *__restrict__ p, const float *__restrict__ q, const float *__restrict__ e,
                           const void *__restrict__ adj, const int32_t *__restrict__ kvalid, const float *__restrict__ W,
                           const float *__restrict__ b, const float *__restrict__ gout, const float *__restrict__ gout_c,
                           float *__restrict__ partials) {
    // PAIR (a separate instantiation: the second gradient costs 30 registers, which the single-network kernels must not pay),
    // gout_c != nullptr (MO_ADJ_TENSOR only): the SAME relation of the critic (adjacency = ones over all K neighbours, shared weights)
    // in the same pass -- the two networks' messages share z_ij, so g_ij = [z_ij > 0] (abar_ij gout_i + gout_c_i / K) and the sum of
    // both weight gradients lands in one set of partials
    const int P = d.P, K = d.K, fs = blockDim.x, f = threadIdx.x;
    FV<EV> w[8], gw[8], bias, gb(0.f);
    msgw_weights<EV>(d, W, b, f, w, bias, fs);
#pragma unroll
    for (int k = 0; k < 8; k++) gw[k] = FV<EV>(0.f);
    const int r0 = blockIdx.x * d.rpb, r1 = min(d.R, r0 + d.rpb);
    const bool use_kv = d.adj_mode == MO_ADJ_VALID;
    MsgRow m = msgw_load<QS, AS>(d, r0 < r1 ? r0 : 0, p, q, e, adj, kvalid, use_kv);
    FV<EV> go[PT], goc[PT];
    const float inv_k = 1.f / fmaxf((float)K, 1e-12f);
#pragma unroll
    for (int i = 0; i < PT; i++) {
        go[i] = (i < P && r0 < r1) ? fv_load<EV>(gout + ((size_t)r0 * P + i) * d.o_is, f, fs) : FV<EV>(0.f);
        goc[i] = (PAIR && i < P && r0 < r1) ? fv_load<EV>(gout_c + ((size_t)r0 * P + i) * d.o_is, f, fs) * inv_k : FV<EV>(0.f);
    }
    for (int r = r0; r < r1; r++) {
        const int rn = r + 1 < r1 ? r + 1 : r;
        const MsgRow nxt = msgw_load<QS, AS>(d, rn, p, q, e, adj, kvalid, use_kv);
        FV<EV> gon[PT], gocn[PT];
#pragma unroll
        for (int i = 0; i < PT; i++) {
            gon[i] = i < P ? fv_load<EV>(gout + ((size_t)rn * P + i) * d.o_is, f, fs) : FV<EV>(0.f);
            gocn[i] = (PAIR && i < P) ? fv_load<EV>(gout_c + ((size_t)rn * P + i) * d.o_is, f, fs) * inv_k : FV<EV>(0.f);
        }
        const float4 *__restrict__ q4 = (const float4 *)(q + (size_t)(r / d.q_div) * d.q_rs);
        FV<EV> c[PT], G[PT];
        msgw_center<PT, EV>(d, m, w, bias, c);
#pragma unroll
        for (int i = 0; i < PT; i++) G[i] = FV<EV>(0.f);
        FV<EV> hq0(0.f), hq1(0.f), hq2(0.f), hq3(0.f);
        if (d.adj_mode == MO_ADJ_BITS) {
            const uint32_t *__restrict__ bw = (const uint32_t *)adj + (size_t)r * d.adj_rs;
            const int nw = (K + 31) >> 5, RWK = MO_ADJ_ROW_WORDS(K);
#pragma unroll
            for (int i = 0; i < PT; i++)
                if (i < P) {
                    int cnt = 0;
                    for (int wd = 0; wd < nw; wd++) {
                        uint32_t bits = msgw_word<AS>(m, bw, i * RWK + wd);
                        if (wd == (K >> 5)) bits &= (1u << (K & 31)) - 1u;
                        cnt += __popc(bits);
                    }
                    const FV<EV> gi = go[i] * (1.f / fmaxf((float)cnt, 1e-12f));
                    for (int wd = 0; wd < nw; wd++) {
                        uint32_t bits = msgw_word<AS>(m, bw, i * RWK + wd);
                        if (wd == (K >> 5)) bits &= (1u << (K & 31)) - 1u;
                        while (bits) {
                            const int j = (wd << 5) + __builtin_ctz(bits);
                            bits &= bits - 1u;
                            const float4 qv = msgw_q<QS>(m, q4, j);
                            const FV<EV> g = fv_pos(c[i] - msg_dot4v(w, qv), gi);
                            G[i] = G[i] + g;
                            hq0 = fv_fma(g, qv.x, hq0); hq1 = fv_fma(g, qv.y, hq1); hq2 = fv_fma(g, qv.z, hq2); hq3 = fv_fma(g, qv.w, hq3);
                        }
                    }
                }
        } else if (d.adj_mode == MO_ADJ_TENSOR) {
            const float *__restrict__ ar = (const float *)adj + (size_t)r * d.adj_rs;   // (!AS: through uniform addresses)
            MsgMask k;
            if (AS) k = msgw_mask(m, P, K);
            FV<EV> gi[PT];
#pragma unroll
            for (int i = 0; i < PT; i++) {
                gi[i] = FV<EV>(0.f);
                if (i < P) {
                    float s = 0.f;
                    if (AS) s = msgw_row_norm(m, k, i, K);
                    else for (int j = 0; j < K; j++) s += fabsf(ar[i * K + j]);
                    gi[i] = go[i] * (1.f / fmaxf(s, 1e-12f));
                }
            }
            for (int j = 0; j < K; j++) {
                uint64_t cm = 0ull;
                if (AS) {
                    cm = (k.nz >> j) & k.col;
                    if (cm == 0ull && !PAIR) continue;
                } else if (!PAIR) {
                    bool any = false;
                    for (int i = 0; i < P; i++) any |= ar[i * K + j] != 0.f;
                    if (!any) continue;
                }
                const float4 qv = msgw_q<QS>(m, q4, j);
                const FV<EV> dj = msg_dot4v(w, qv);
                FV<EV> hj(0.f);
#pragma unroll
                for (int i = 0; i < PT; i++)
                    if (i < P) {
                        float aij;
                        if (AS) {
                            aij = ((cm >> (i * K)) & 1ull) ? (k.zero_one ? 1.f : __uint_as_float(rl_u(m.va0, i * K + j))) : 0.f;
                        } else {
                            aij = ar[i * K + j];
                        }
                        if (aij == 0.f && !PAIR) continue;
                        const FV<EV> g = fv_pos(c[i] - dj, PAIR ? fv_fma(gi[i], aij, goc[i]) : gi[i] * aij);   // actor: abar_ij gout_i; critic: gout_c_i / K
                        G[i] = G[i] + g;
                        hj = hj + g;
                    }
                hq0 = fv_fma(hj, qv.x, hq0); hq1 = fv_fma(hj, qv.y, hq1); hq2 = fv_fma(hj, qv.z, hq2); hq3 = fv_fma(hj, qv.w, hq3);
            }
        } else {
            const int kv = m.kv;
            const float iv = 1.f / fmaxf((float)kv, 1e-12f);
            FV<EV> gi[PT];
#pragma unroll
            for (int i = 0; i < PT; i++) gi[i] = go[i] * iv;
            for (int j = 0; j < kv; j++) {
                const float4 qv = msgw_q<QS>(m, q4, j);
                const FV<EV> dj = msg_dot4v(w, qv);
                FV<EV> hj(0.f);
#pragma unroll
                for (int i = 0; i < PT; i++) {
                    const FV<EV> g = fv_pos(c[i] - dj, gi[i]);
                    G[i] = G[i] + g;
                    hj = hj + g;
                }
                hq0 = fv_fma(hj, qv.x, hq0); hq1 = fv_fma(hj, qv.y, hq1); hq2 = fv_fma(hj, qv.z, hq2); hq3 = fv_fma(hj, qv.w, hq3);
            }
        }
        const float e0 = rl_f(m.ve, 0), e1 = rl_f(m.ve, 1), e2 = rl_f(m.ve, 2), e3 = rl_f(m.ve, 3);
#pragma unroll
        for (int i = 0; i < PT; i++)
            if (i < P) {
                const float p0 = rl_f(m.vp, 4 * i), p1 = rl_f(m.vp, 4 * i + 1), p2 = rl_f(m.vp, 4 * i + 2), p3 = rl_f(m.vp, 4 * i + 3);
                gb = gb + G[i];
                gw[0] = fv_fma(G[i], p0, gw[0]); gw[1] = fv_fma(G[i], p1, gw[1]); gw[2] = fv_fma(G[i], p2, gw[2]); gw[3] = fv_fma(G[i], p3, gw[3]);
                if (d.din == 8) {
                    gw[4] = fv_fma(G[i], p0 - e0, gw[4]); gw[5] = fv_fma(G[i], p1 - e1, gw[5]); gw[6] = fv_fma(G[i], p2 - e2, gw[6]);
                    gw[7] = fv_fma(G[i], p3 - e3, gw[7]);
                }
            }
        gw[0] = gw[0] - hq0; gw[1] = gw[1] - hq1; gw[2] = gw[2] - hq2; gw[3] = gw[3] - hq3;
        m = nxt;
#pragma unroll
        for (int i = 0; i < PT; i++) { go[i] = gon[i]; if (PAIR) goc[i] = gocn[i]; }
    }
    float *dst = partials + (size_t)blockIdx.x * (d.din + 1) * d.E;
    for (int k = 0; k < d.din; k++) fv_store<EV>(dst + k * d.E, f, fs, gw[k]);
    fv_store<EV>(dst + d.din * d.E, f, fs, gb);
}

// one workgroup per output element group: 64 lanes split the partial blocks, then a wave reduction (deterministic order)
__global__ void k_msg_agg_bwd_reduce(int nblk, int E, int din, const float *partials, float *dW, float *db) {
    const int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // over (din+1)*E, one wave each
    const int lane = threadIdx.x & 63;
    if (idx >= (din + 1) * E) return;
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += (double)partials[(size_t)b * (din + 1) * E + idx];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
        const int k = idx / E, f = idx - k * E;
        if (k < din) dW[(size_t)f * din + k] = (float)s; else db[f] = (float)s;
    }
}

// ---- all-ones adjacency over a STATIC neighbour set (the critic's obstacle relation in training) -------------------
// With adj = ones the aggregate of a relation is  out_i[f] = (1/K) sum_j relu(c_i[f] - d_j[f]),  c_i = b + W p_i,  d_j = W q_j.
// The obstacles q_j are per EPISODE (q_div = T rows share them), so for one (episode, feature) the K values d_j are fixed
// for all T x P pursuer rows:  sum_j relu(c - d_j) = m c - S[m]  with m = #{j : d_j < c} and S the prefix sums of the sorted
// d -- a piecewise-linear function of c.  One workgroup per (episode, 16 features): sorts d once (bitonic, LDS), builds the
// prefix tables of d and of q (sorted order), then evaluates every (step, pursuer) by an 8-step binary search: O(log K)
// instead of O(K) per pair.  Backward: dL/dc = g m, so db = sum g m, dW[:, k] = sum g (m p_i[k] - Q_k[m]) with Q_k the
// prefix sums of q_j[k] in sorted order; m and the Q tables are saved by the forward pass.
constexpr int SO_FC = 16, SO_TPB = 256, SO_PL = SO_TPB / SO_FC, SO_MAXK = 255, SO_LD = 260;  // 16 features x 16 pair lanes; rows 16-byte aligned

struct SortedArgs {
    int R, P, K, E, q_div;
    const float *p, *q, *W, *b;
    int64_t p_rs, q_rs, o_is;
};

__host__ __device__ inline size_t so_fwd_lds(int K) { return sizeof(float) * ((size_t)SO_FC * SO_LD + (size_t)SO_FC * (K + 2) + SO_FC * SO_PL * 5 + 4 * (K + 1)) + SO_FC * 256; }
__host__ __device__ inline size_t so_bwd_lds(int K) { return sizeof(float) * ((size_t)SO_FC * 4 * (K + 2) + SO_PL * SO_FC * 5); }

__global__ __launch_bounds__(SO_TPB) void k_msg_ones_sorted_fwd(SortedArgs a, float *out, uint8_t *save_m, float *qtab) {
    extern __shared__ __attribute__((aligned(16))) float so_smem[];
    const int K = a.K, P = a.P, KP = K + 2;
    float4 *s_q = (float4 *)so_smem;                                   // [K + 1]
    float (*s_key)[SO_LD] = (float (*)[SO_LD])(so_smem + 4 * (K + 1));  // [FC][257] d sorted ascending (+inf padding)
    float *s_pre = so_smem + 4 * (K + 1) + SO_FC * SO_LD;               // [FC][K + 2] prefix sums of the sorted d
    float *s_chunk = s_pre + SO_FC * KP;                                // [FC][PL][5]
    uint8_t (*s_idx)[256] = (uint8_t (*)[256])(s_chunk + SO_FC * SO_PL * 5);
    const int tid = threadIdx.x, f = tid & (SO_FC - 1), pl = tid >> 4;
    const int n = blockIdx.x, f0 = blockIdx.y * SO_FC;
    const float *wrow = a.W + (size_t)(f0 + f) * 4;
    const float w[4] = {wrow[0], wrow[1], wrow[2], wrow[3]};
    const float bias = a.b[f0 + f];
    for (int j = tid; j < K; j += SO_TPB) s_q[j] = ((const float4 *)(a.q + (size_t)n * a.q_rs))[j];
    __syncthreads();
    // Sort the 256 (padded) keys of each of the 16 features IN REGISTERS, one wave per feature at a time (four features per wave),
    // four keys per lane (element e = 64 r + lane): the bitonic network's exchanges at distance >= 64 are register moves, the rest
    // wave shuffles -- no LDS traffic and no workgroup barrier inside the sort (rounds 2-3 sorted in LDS: 36 passes of eight
    // compare-exchanges per thread with a barrier each were most of the kernel's 830 us).  Ties are broken by the obstacle index,
    // so the network is a total order and the result does not depend on the exchange pattern.
    {
        const int lane = tid & 63, wv = tid >> 6;
        for (int ff = 0; ff < SO_FC / 4; ff++) {
            const int fx = wv * (SO_FC / 4) + ff;
            const float *wr = a.W + (size_t)(f0 + fx) * 4;
            const float wx[4] = {wr[0], wr[1], wr[2], wr[3]};
            float key[4];
            int idx[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int j = 64 * r + lane;
                key[r] = j < K ? msg_dot4(wx, s_q[j]) : __builtin_inff();
                idx[r] = j;
            }
#pragma unroll
            for (int k = 2; k <= 256; k <<= 1) {
#pragma unroll
                for (int jj = k >> 1; jj > 0; jj >>= 1) {
                    if (jj >= 64) {
                        const int dr = jj >> 6;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            if (r & dr) continue;
                            const bool up = ((64 * r) & k) == 0;
                            const float x = key[r], y = key[r | dr];
                            const int ix = idx[r], iy = idx[r | dr];
                            const bool gt = x > y || (x == y && ix > iy);
                            if (gt == up) { key[r] = y; key[r | dr] = x; idx[r] = iy; idx[r | dr] = ix; }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float y = __shfl_xor(key[r], jj);
                            const int iy = __shfl_xor(idx[r], jj);
                            const bool up = ((64 * r + lane) & k) == 0, lower = (lane & jj) == 0;
                            const bool lt = key[r] < y || (key[r] == y && idx[r] < iy);
                            const bool keep = (up == lower) == lt;    // the lower slot of an ascending pair keeps the smaller element
                            key[r] = keep ? key[r] : y;
                            idx[r] = keep ? idx[r] : iy;
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                s_key[fx][64 * r + lane] = key[r];
                s_idx[fx][64 * r + lane] = (uint8_t)idx[r];
            }
        }
    }
    __syncthreads();
    // prefix tables: 16 lanes per feature, each sums a contiguous chunk, then offsets by the chunks before it; the prefix
    // sums of d stay in LDS, those of q_j[0..3] (sorted order) go to qtab for the backward pass: [n][E][4][K + 1]
    const int ch = (K + SO_PL - 1) / SO_PL, m0 = pl * ch, m1 = (m0 + ch < K) ? m0 + ch : K;
    {
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int m = m0; m < m1; m++) {
            const float4 qv = s_q[s_idx[f][m]];
            acc[0] += s_key[f][m]; acc[1] += qv.x; acc[2] += qv.y; acc[3] += qv.z; acc[4] += qv.w;
        }
        for (int k = 0; k < 5; k++) s_chunk[(f * SO_PL + pl) * 5 + k] = acc[k];
    }
    __syncthreads();
    {
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < pl; c++)
            for (int k = 0; k < 5; k++) acc[k] += s_chunk[(f * SO_PL + c) * 5 + k];
        // qtab [episode][feature block][4][K + 1][16 features]: the 16 feature lanes of a pair lane store 64 contiguous bytes
        // (rounds 2-3 kept [feature][4][K + 1]: every lane's 4-byte store its own transaction, 37 M of them per mini-batch)
        float *qt = qtab ? qtab + ((size_t)n * gridDim.y + blockIdx.y) * 4 * (K + 1) * SO_FC + f : nullptr;
        const int QS = (K + 1) * SO_FC;
        if (pl == 0) {
            s_pre[f * KP] = 0.f;
            if (qt) for (int k = 0; k < 4; k++) qt[k * QS] = 0.f;
        }
        for (int m = m0; m < m1; m++) {
            const float4 qv = s_q[s_idx[f][m]];
            acc[0] += s_key[f][m]; acc[1] += qv.x; acc[2] += qv.y; acc[3] += qv.z; acc[4] += qv.w;
            s_pre[f * KP + m + 1] = acc[0];
            if (qt) {
                float *q1 = qt + (m + 1) * SO_FC;
                q1[0] = acc[1]; q1[QS] = acc[2]; q1[2 * QS] = acc[3]; q1[3 * QS] = acc[4];
            }
        }
    }
    __syncthreads();
    // rank m = #{j : d_j < c} of a query in TWO steps instead of an 8-step binary search (eight dependent LDS round trips per pair: the
    // kernel was latency bound, 830 us per mini-batch): the last key of every 16-key bucket sits in a register (a thread serves ONE
    // feature), so the bucket is a count of 16 register compares; the bucket's 16 keys are four independent 16-byte reads and another
    // count.  Three pairs are in flight per thread; the query rows come through registers one group ahead.
    float spl[16];
#pragma unroll
    for (int k = 0; k < 16; k++) spl[k] = s_key[f][16 * k + 15];
    const float invK = 1.f / (float)K;
    const int pairs = a.q_div * P;
    constexpr int U = 3;
    const size_t row0 = (size_t)n * a.q_div;
    uint8_t *sm = save_m ? save_m + (((size_t)n * gridDim.y + blockIdx.y) * pairs) * SO_FC + f : nullptr;   // [episode][feature block][pair][16]: 16 lanes = 16 bytes
    float4 pv[U];
    auto load_p = [&](int base) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int pi = base + u * SO_PL;
            const int t = pi / P, i = pi - t * P;
            pv[u] = pi < pairs ? *(const float4 *)(a.p + (row0 + t) * a.p_rs + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_p(pl);
    for (int base = pl; base < pairs; base += U * SO_PL) {
        float c[U];
        int bk[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            c[u] = bias + w[0] * pv[u].x + w[1] * pv[u].y + w[2] * pv[u].z + w[3] * pv[u].w;
            int b = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) b += spl[k] < c[u] ? 1 : 0;
            bk[u] = b < 15 ? b : 15;          // (b = 16 needs a finite 256th key: K <= 255 keeps the last one +inf)
        }
        float4 kq[U][4];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) kq[u][q4] = *(const float4 *)&s_key[f][16 * bk[u] + 4 * q4];
        if (base + U * SO_PL < pairs) load_p(base + U * SO_PL);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int pi = base + u * SO_PL;
            if (pi >= pairs) continue;
            int m = 16 * bk[u];
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++)
                m += (kq[u][q4].x < c[u] ? 1 : 0) + (kq[u][q4].y < c[u] ? 1 : 0) + (kq[u][q4].z < c[u] ? 1 : 0) + (kq[u][q4].w < c[u] ? 1 : 0);
            const int t = pi / P, i = pi - t * P;
            out[((row0 + t) * P + i) * a.o_is + f0 + f] = ((float)m * c[u] - s_pre[f * KP + m]) * invK;
            if (sm) sm[(size_t)pi * SO_FC] = (uint8_t)m;
        }
    }
}

// partials: [gridDim.x = episodes][5][E] in the layout k_msg_agg_bwd_reduce sums (din = 4)
__global__ __launch_bounds__(SO_TPB) void k_msg_ones_sorted_bwd(SortedArgs a, const float *gout, const uint8_t *save_m, const float *qtab,
                                                               float *partials) {
    extern __shared__ __attribute__((aligned(16))) float so_smem[];
    const int tid = threadIdx.x, f = tid & (SO_FC - 1), pl = tid >> 4;
    const int n = blockIdx.x, f0 = blockIdx.y * SO_FC, K = a.K, P = a.P, KQ = 4 * (K + 1);
    float *s_qtab = so_smem;                       // [FC][4][K + 1]: qtab's (episode, feature block) slice [4][K + 1][FC], transposed on the way in
    float *s_red = so_smem + SO_FC * 4 * (K + 2);   // [PL][FC][5]
    {
        const float4 *src = (const float4 *)(qtab + ((size_t)n * gridDim.y + blockIdx.y) * KQ * SO_FC);
        for (int idx = tid; idx < SO_FC * KQ / 4; idx += SO_TPB) {   // 16-byte coalesced reads: four features of one (k, m) entry
            const float4 v = src[idx];
            const int km = idx >> 2, fq = (idx & 3) * 4;
            s_qtab[(fq + 0) * KQ + km] = v.x; s_qtab[(fq + 1) * KQ + km] = v.y; s_qtab[(fq + 2) * KQ + km] = v.z; s_qtab[(fq + 3) * KQ + km] = v.w;
        }
    }
    __syncthreads();
    const float *s_qt = s_qtab + f * KQ;
    const float invK = 1.f / (float)K;
    float gw[4] = {0.f, 0.f, 0.f, 0.f}, gb = 0.f;
    const int pairs = a.q_div * P;
    const uint8_t *sm = save_m + (((size_t)n * gridDim.y + blockIdx.y) * pairs) * SO_FC + f;    // the forward kernel's layout
    const size_t row0 = (size_t)n * a.q_div;
    constexpr int U = 4;   // pairs in flight per thread (every load of a pair is independent of the others)
    for (int base = pl; base < pairs; base += U * SO_PL) {
        float4 pv[U];
        float g[U];
        int m[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int pi = base + u * SO_PL;
            const int t = pi / P, i = pi - t * P;
            const bool ok = pi < pairs;
            pv[u] = ok ? *(const float4 *)(a.p + (row0 + t) * a.p_rs + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            g[u] = ok ? gout[((row0 + t) * P + i) * a.o_is + f0 + f] * invK : 0.f;
            m[u] = ok ? sm[(size_t)pi * SO_FC] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {     // ascending pair order per thread, as before: the sums round identically
            const float gm = g[u] * (float)m[u];
            gb += gm;
            gw[0] += gm * pv[u].x - g[u] * s_qt[m[u]];
            gw[1] += gm * pv[u].y - g[u] * s_qt[(K + 1) + m[u]];
            gw[2] += gm * pv[u].z - g[u] * s_qt[2 * (K + 1) + m[u]];
            gw[3] += gm * pv[u].w - g[u] * s_qt[3 * (K + 1) + m[u]];
        }
    }
    for (int k = 0; k < 4; k++) s_red[(pl * SO_FC + f) * 5 + k] = gw[k];
    s_red[(pl * SO_FC + f) * 5 + 4] = gb;
    __syncthreads();
    if (tid < SO_FC * 5) {
        const int ff = tid / 5, k = tid - ff * 5;
        float sum = 0.f;
        for (int c = 0; c < SO_PL; c++) sum += s_red[(c * SO_FC + ff) * 5 + k];   // fixed order: deterministic
        partials[((size_t)n * 5 + k) * a.E + f0 + ff] = sum;
    }
}

// ---- GAE ---------------------------------------------------------------------------------------------------
// The statistics of the advantage normaliser (mean, unbiased std over all N T P elements) are f64 sums through per-workgroup
// partials that one thread adds in index order: no atomics, so the same input gives the same bits every run (like k_ppo_loss).
// stats: [0] sum, [1] sum of squares / of squared deviations, [2] mean, [3] std, [4 ..] the partials (2 per workgroup).
constexpr int GAE_BLOCKS = 256;

// sum over the workgroup (256 threads) of two f64 values, in a fixed order; valid in thread 0
__device__ __forceinline__ void gae_block_sum(double &s, double &s2) {
    __shared__ double red[2][4];
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); s2 += __shfl_xor(s2, off); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ __launch_bounds__(256) void k_gae_scan(int N, int T, int P, const float *r, const float *v, const float *active, float gamma,
                                                  float lamda, float *adv, float *v_target, double *stats) {
    double s = 0.0, s2 = 0.0;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < N * P; idx += gridDim.x * 256) {
        const int n = idx / P, p = idx - n * P;
        float gae = 0.f;
        for (int t = T - 1; t >= 0; t--) {
            const size_t o = ((size_t)n * T + t) * P + p;
            const float vt = v[((size_t)n * (T + 1) + t) * P + p], vn = v[((size_t)n * (T + 1) + t + 1) * P + p];
            float delta = (r[o] + gamma * vn - vt) * active[o];
            gae = delta + gamma * lamda * gae;
            adv[o] = gae;
            v_target[o] = gae + vt;
            s += (double)gae;
            s2 += (double)gae * (double)gae;
        }
    }
    gae_block_sum(s, s2);
    if (threadIdx.x == 0) { stats[4 + 2 * blockIdx.x] = s; stats[5 + 2 * blockIdx.x] = s2; }
}

__global__ void k_gae_finalize(int64_t n, int nblk, double *stats) {
    double s = 0.0, s2 = 0.0;
    for (int b = 0; b < nblk; b++) { s += stats[4 + 2 * b]; s2 += stats[5 + 2 * b]; }
    const double mean = s / (double)n;
    const double var = (s2 - (double)n * mean * mean) / (double)(n - 1);  // unbiased, torch.std default (refined by k_gae_center)
    stats[0] = s;
    stats[1] = s2;
    stats[2] = mean;
    stats[3] = sqrt(var > 0.0 ? var : 0.0);
}

__global__ __launch_bounds__(256) void k_gae_center(int64_t n, const float *adv, double *stats) {
    // second pass for a numerically robust variance: sum (x - mean)^2
    const double mean = stats[2];
    double d = 0.0, unused = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double x = (double)adv[i] - mean;
        d += x * x;
    }
    gae_block_sum(d, unused);
    if (threadIdx.x == 0) stats[4 + 2 * blockIdx.x] = d;
}

__global__ void k_gae_std(int64_t n, int nblk, double *stats) {
    double d = 0.0;
    for (int b = 0; b < nblk; b++) d += stats[4 + 2 * b];
    stats[1] = d;
    stats[3] = sqrt(d / (double)(n - 1));
}

__global__ void k_gae_norm(int64_t n, float *adv, const float *active, const double *stats) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float mean = (float)stats[2], sd = (float)stats[3];
        adv[i] = (adv[i] - mean) / (sd + 1e-5f) * active[i];
    }
}

__global__ void k_advance_counter(uint64_t *ctr, uint64_t by) { *ctr += by; }

// ---- spectral normalisation of a small head (torch.nn.utils.spectral_norm's pre-forward hook; reference value head :485) ------
// n_iter power iterations  v = normalize(W^T u), u = normalize(W v)  in place, then sigma = u . (W v) and w_eff = W / sigma:
// the hook's ~14 tiny launches per forward (two matrix-vector products, norms, clamps, divisions, clones, a dot) as one
// workgroup.  normalize(x) = x / max(||x||_2, eps) as in F.normalize.  A <= SN_MAX_A rows, H <= SN_MAX_H columns.
constexpr int SN_MAX_A = 16, SN_MAX_H = 1024;
__global__ __launch_bounds__(256) void k_sn_power(int A, int H, const float *__restrict__ W, float *u, float *v, float eps, int n_iter,
                                                  float *__restrict__ w_eff) {
    __shared__ float s_u[SN_MAX_A], s_s[SN_MAX_A], s_v[SN_MAX_H], s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < A) s_u[tid] = u[tid];
    for (int k = tid; k < H; k += 256) s_v[k] = v[k];
    __syncthreads();
    for (int it = 0; it <= n_iter; it++) {
        if (it > 0) {  // v = normalize(W^T u)
            float part = 0.f;
            for (int k = tid; k < H; k += 256) {
                float t = 0.f;
                for (int a = 0; a < A; a++) t = __builtin_fmaf(W[(size_t)a * H + k], s_u[a], t);
                s_v[k] = t;
                part = __builtin_fmaf(t, t, part);
            }
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            if (lane == 0) s_red[wave] = part;
            __syncthreads();
            const float den = fmaxf(sqrtf((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])), eps);
            for (int k = tid; k < H; k += 256) s_v[k] = s_v[k] / den;
            __syncthreads();
        }
        // s = W v (needed for u and, with the final v, for sigma)
        for (int a = wave; a < A; a += 4) {
            float part = 0.f;
            for (int k = lane; k < H; k += 64) part = __builtin_fmaf(W[(size_t)a * H + k], s_v[k], part);
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            if (lane == 0) s_s[a] = part;
        }
        __syncthreads();
        if (it > 0) {  // u = normalize(s)
            if (tid == 0) {
                float q = 0.f;
                for (int a = 0; a < A; a++) q = __builtin_fmaf(s_s[a], s_s[a], q);
                const float den = fmaxf(sqrtf(q), eps);
                for (int a = 0; a < A; a++) s_u[a] = s_s[a] / den;
            }
            __syncthreads();
        }
    }
    float sigma = 0.f;
    for (int a = 0; a < A; a++) sigma = __builtin_fmaf(s_u[a], s_s[a], sigma);
    for (int i = tid; i < A * H; i += 256) w_eff[i] = W[i] / sigma;
    if (n_iter > 0) {
        if (tid < A) u[tid] = s_u[tid];
        for (int k = tid; k < H; k += 256) v[k] = s_v[k];
    }
}

// ---- Categorical sample / argmax ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t *o) {
    for (int i = 0; i < 10; i++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__global__ void k_categorical(int R, int A, const float *probs, uint64_t seed, uint64_t offset, const uint64_t *offset_dev, int greedy,
                              int32_t *action, float *logp) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    if (offset_dev) offset += *offset_dev;
    const float *p = probs + (size_t)r * A;
    float tot = 0.f;
    for (int k = 0; k < A; k++) tot += p[k];
    int a = 0;
    if (greedy) {
        float best = p[0];
        for (int k = 1; k < A; k++) if (p[k] > best) { best = p[k]; a = k; }
    } else {
        uint32_t o[4];
        const uint64_t ctr = offset + (uint64_t)r;
        philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        const float u = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f) * tot;  // (0, tot)
        float cum = 0.f;
        a = A - 1;
        for (int k = 0; k < A; k++) { cum += p[k]; if (u < cum) { a = k; break; } }
    }
    action[r] = a;
    if (logp) {
        // Categorical(probs=p).log_prob(a): log(clamp(p / sum p, eps, 1 - eps)), eps = FLT_EPSILON
        float pn = p[a] / tot;
        pn = fminf(fmaxf(pn, 1.1920929e-07f), 1.f - 1.1920929e-07f);
        logp[r] = logf(pn);
    }
}


// LDS hand-over between the lanes of ONE wave (no workgroup barrier): order the stores before the loads
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over each row of 16 lanes, result in all 16 (DPP only: quad butterflies, then the half-row and row mirrors)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);  // row_half_mirror: the other quad of the half
    v += dpp_f<0x140>(v);  // row_mirror: the other half
    return v;
}

// ---- output heads of the rollout tick ------------------------------------------------------------------------------------------
// y[r][a] = feat[r] . W[a] + b[a] for a head with A <= 16 outputs on H = 128 features.  Sixteen lanes share a row (lane i holds
// features 8 i .. 8 i + 7 and the matching weight columns in registers; the A dot products are folded with four xor-shuffles), a
// wave covers 64 rows in 16 such steps and parks the logits in LDS, then every lane finishes ONE row on its own:
//   SAMPLE: the actor -- softmax, Categorical sample and log-probability as in k_categorical: one launch instead of GEMM +
//           softmax + sample + counter update (the counter is advanced by the last workgroup out).
//   else  : the critic -- y written out (A = 1: the value), straight into the rollout's static storage.
constexpr int HEAD_MAX_A = 16, HEAD_H = 128;
template <bool SAMPLE, int AT>
__global__ __launch_bounds__(256) void k_head(int R, int A, const float *__restrict__ feat, const float *__restrict__ W, const float *__restrict__ b,
                                              float *__restrict__ y, uint64_t seed, uint64_t *counter, unsigned int *done, int greedy,
                                              int32_t *__restrict__ action, float *__restrict__ logp) {
    __shared__ float s_y[4][64][AT + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    const int i = lane & 15, g = lane >> 4;
    float w[AT][8];
#pragma unroll
    for (int a = 0; a < AT; a++)   // (A == AT: the host picks the instantiation)
#pragma unroll
        for (int k = 0; k < 8; k++) w[a][k] = W[a * HEAD_H + 8 * i + k];
    const uint64_t offset = SAMPLE ? *counter : 0ull;
    float (*sy)[AT + 1] = s_y[wave];
    for (int r0 = (blockIdx.x * 4 + wave) * 64; r0 < R; r0 += nw * 64) {
        float4 fall[16][2];  // all 64 rows' loads in flight before the first use (one memory latency per 64 rows, not sixteen)
#pragma unroll
        for (int st = 0; st < 16; st++) {
            const int r = r0 + 4 * st + g;
            fall[st][0] = fall[st][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < R) {
                fall[st][0] = *(const float4 *)(feat + (size_t)r * HEAD_H + 8 * i);
                fall[st][1] = *(const float4 *)(feat + (size_t)r * HEAD_H + 8 * i + 4);
            }
        }
#pragma unroll
        for (int st = 0; st < 16; st++) {
            const int row = 4 * st + g;
            const float4 fa = fall[st][0], fb = fall[st][1];
            float mine = 0.f;
#pragma unroll
            for (int a = 0; a < AT; a++) {
                float sum = fa.x * w[a][0];
                sum = __builtin_fmaf(fa.y, w[a][1], sum); sum = __builtin_fmaf(fa.z, w[a][2], sum); sum = __builtin_fmaf(fa.w, w[a][3], sum);
                sum = __builtin_fmaf(fb.x, w[a][4], sum); sum = __builtin_fmaf(fb.y, w[a][5], sum); sum = __builtin_fmaf(fb.z, w[a][6], sum);
                sum = __builtin_fmaf(fb.w, w[a][7], sum);
                sum = row16_sum(sum);
                mine = i == a ? sum : mine;   // lane a of the group keeps output a
            }
            if (i < AT) sy[row][i] = mine;
        }
        wave_fence();  // the 64 rows' logits are in LDS: lane l takes row l
        const int r = r0 + lane;
        if (r < R) {
            float v[AT];
#pragma unroll
            for (int a = 0; a < AT; a++) v[a] = sy[lane][a] + b[a];
            if (!SAMPLE) {
#pragma unroll
                for (int a = 0; a < AT; a++) y[(size_t)r * AT + a] = v[a];
            } else {
                // softmax over the A logits (max-subtracted, as torch.softmax), then k_categorical's sampling on the probabilities
                float mx = v[0];
#pragma unroll
                for (int a = 1; a < AT; a++) mx = fmaxf(mx, v[a]);
                float p[AT], den = 0.f;
#pragma unroll
                for (int a = 0; a < AT; a++) { p[a] = expf(v[a] - mx); den += p[a]; }
                float tot = 0.f;
#pragma unroll
                for (int a = 0; a < AT; a++) { p[a] = p[a] / den; tot += p[a]; }
                int act = 0;
                float pa = p[0];
                if (greedy) {
#pragma unroll
                    for (int a = 1; a < AT; a++) { const bool up = p[a] > pa; pa = up ? p[a] : pa; act = up ? a : act; }
                } else {
                    uint32_t o[4];
                    const uint64_t ctr = offset + (uint64_t)r;
                    philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
                    const float u = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f) * tot;  // (0, tot)
                    float cum = 0.f;
                    bool found = false;
                    act = AT - 1;
                    pa = p[AT - 1];
#pragma unroll
                    for (int a = 0; a < AT; a++) {
                        cum += p[a];
                        const bool hit = !found && u < cum;
                        act = hit ? a : act;
                        pa = hit ? p[a] : pa;
                        found = found || hit;
                    }
                }
                action[r] = act;
                float pn = pa / tot;
                pn = fminf(fmaxf(pn, 1.1920929e-07f), 1.f - 1.1920929e-07f);
                logp[r] = logf(pn);
            }
        }
        wave_fence();  // before the next 64 rows overwrite the logits
    }
    if (SAMPLE) {  // every workgroup has read the counter at its start; the last one to finish advances it and resets the ticket
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            if (atomicAdd(done, 1u) == gridDim.x - 1) {
                *counter = offset + (uint64_t)R;
                *done = 0u;
                __threadfence();
            }
        }
    }
}

// ---- GRU gate math (torch.nn.GRU cell; reference DHGN/mappo_parallel.py:397,424,434) ------------------------------
// gi = x W_ih^T + b_ih and gh = h W_hh^T come from MFMA GEMMs (rocBLAS/hipBLASLt fp32); everything between them and
// the next step's GEMM is fused here: bias, sigmoid/tanh, the state update and (for training) the saved gates.
//   r = s(gi_r + gh_r + bhh_r); z = s(gi_z + gh_z + bhh_z); hn = gh_n + bhh_n; n = tanh(gi_n + r * hn); h' = (1-z) n + z h
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// Gate math of the MFMA GRU kernels: the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each) instead of the
// libm sequences.  The gate phase is pure VALU time the matrix pipe waits for (3.5 k of 18 k cycles per tile in
// k_gru_cell); the results differ from expf / tanhf by < 3e-7 absolute, the size of the fp32 GEMM reordering noise.
__device__ __forceinline__ float sigmoid_hw(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_hw(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

__global__ void k_gru_gates_fwd(int B, int H, const float *gi, const float *gh, const float *bhh, const float *hprev, float *hout,
                                float *save /* [4][B][H]: r, z, n, hn or NULL */) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // over B*H/4
    const int H4 = H >> 2;
    if (idx >= B * H4) return;
    const int b = idx / H4, c = (idx - b * H4) << 2;
    const float4 gir = *(const float4 *)(gi + (size_t)b * 3 * H + c), giz = *(const float4 *)(gi + (size_t)b * 3 * H + H + c),
                 gin = *(const float4 *)(gi + (size_t)b * 3 * H + 2 * H + c);
    const float4 ghr = *(const float4 *)(gh + (size_t)b * 3 * H + c), ghz = *(const float4 *)(gh + (size_t)b * 3 * H + H + c),
                 ghn = *(const float4 *)(gh + (size_t)b * 3 * H + 2 * H + c);
    const float4 br = *(const float4 *)(bhh + c), bz = *(const float4 *)(bhh + H + c), bn = *(const float4 *)(bhh + 2 * H + c);
    const float4 hp = *(const float4 *)(hprev + (size_t)b * H + c);
    float4 r, z, n, hn, ho;
#define GRU_ONE(f)                                  \
    r.f = sigmoidf_(gir.f + ghr.f + br.f);          \
    z.f = sigmoidf_(giz.f + ghz.f + bz.f);          \
    hn.f = ghn.f + bn.f;                            \
    n.f = tanhf(gin.f + r.f * hn.f);                \
    ho.f = (1.f - z.f) * n.f + z.f * hp.f;
    GRU_ONE(x) GRU_ONE(y) GRU_ONE(z) GRU_ONE(w)
#undef GRU_ONE
    *(float4 *)(hout + (size_t)b * H + c) = ho;
    if (save) {
        const size_t BH = (size_t)B * H, o = (size_t)b * H + c;
        *(float4 *)(save + o) = r; *(float4 *)(save + BH + o) = z; *(float4 *)(save + 2 * BH + o) = n; *(float4 *)(save + 3 * BH + o) = hn;
    }
}

// dh = dout (may be NULL) + dh_carry ; outputs dgi [B][3H], dgh [B][3H], dh_direct [B][H] (= dh * z)
__global__ void k_gru_gates_bwd(int B, int H, const float *dout, const float *dcarry, const float *save, const float *hprev, float *dgi,
                                float *dgh, float *dhdirect) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int H4 = H >> 2;
    if (idx >= B * H4) return;
    const int b = idx / H4, c = (idx - b * H4) << 2;
    const size_t BH = (size_t)B * H, o = (size_t)b * H + c;
    const float4 r = *(const float4 *)(save + o), z = *(const float4 *)(save + BH + o), n = *(const float4 *)(save + 2 * BH + o),
                 hn = *(const float4 *)(save + 3 * BH + o);
    const float4 hp = *(const float4 *)(hprev + o);
    float4 dh = dcarry ? *(const float4 *)(dcarry + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (dout) { const float4 d2 = *(const float4 *)(dout + o); dh.x += d2.x; dh.y += d2.y; dh.z += d2.z; dh.w += d2.w; }
    float4 dr, dz, dn, dnr, dd;
#define GRU_ONE(f)                                           \
    {                                                        \
        const float dn_ = dh.f * (1.f - z.f) * (1.f - n.f * n.f); \
        const float dz_ = dh.f * (hp.f - n.f) * z.f * (1.f - z.f); \
        const float dr_ = dn_ * hn.f * r.f * (1.f - r.f);    \
        dr.f = dr_; dz.f = dz_; dn.f = dn_; dnr.f = dn_ * r.f; dd.f = dh.f * z.f; \
    }
    GRU_ONE(x) GRU_ONE(y) GRU_ONE(z) GRU_ONE(w)
#undef GRU_ONE
    float *gi = dgi + (size_t)b * 3 * H + c, *gh = dgh + (size_t)b * 3 * H + c;
    *(float4 *)(gi) = dr; *(float4 *)(gi + H) = dz; *(float4 *)(gi + 2 * H) = dn;
    *(float4 *)(gh) = dr; *(float4 *)(gh + H) = dz; *(float4 *)(gh + 2 * H) = dnr;
    *(float4 *)(dhdirect + o) = dd;
}

// ---- persistent GRU over a whole sequence (H = 128) ---------------------------------------------------------------
// The recurrence h_t = cell(gi_t, h_{t-1}) is latency bound when every step is a GEMM launch plus a gate launch
// (2 x 150 launches per layer, ~27 us per step at 3280 rows).  Here one workgroup (8 wavefronts) owns 16 batch rows for
// all T steps: W_hh (384 x 128 fp32 = 196 KB, more than LDS) lives in the workgroup's REGISTERS as v_mfma_f32_16x16x4_f32
// operands (96 VGPRs per lane), the h tile lives in LDS in operand order, the three gate tiles of a hidden unit land in the same
// lane, so the gate math needs no exchange and there is one barrier per step.  Kernels: k_gru_seq_fwd2 / k_gru_seq_bwd2 below.
constexpr int GRU_H = 128, GRU_RB = 16;

// gi_agents: row order of gi (and of dgi in the backward kernel).  0: time-major [t][b] like out.  P > 0: the rows of the
// encoder's output, (episode n, step t, agent p) with b = n P + p, i.e. row ((b / P) T + t) P + b % P -- the input projection
// and its gradients then run on the embedding as it lies in memory, and the two permuted copies of the (rows, 128) sequence
// tensors autograd would make around the GRU (reference _sequence_features, DHGN/mappo_parallel.py:426-437) do not exist.
__device__ __forceinline__ size_t gru_gi_row(int b, int t, int T, int B, int gi_agents) {
    return gi_agents ? ((size_t)(b / gi_agents) * T + t) * gi_agents + b % gi_agents : (size_t)t * B + b;
}

// ---- one GRU step for a large batch (the rollout: B = envs x agents rows, T = 1) ---------------------------------------
// torch.nn.GRU cell (I = H = 128) in ONE launch instead of two GEMMs plus a gate kernel: the workgroups are persistent
// (one per CU), wave w keeps the W_ih AND W_hh rows of hidden units 16w..16w+15 (r, z, n: 192 VGPRs) as MFMA B-operands and
// loops over 16-row tiles; the x and h tiles go through LDS in A-operand order (double buffered, the global loads of
// the next tile are in flight during the MFMAs), r and z accumulate the input and the recurrent product in the same
// accumulator, the gate math runs on the C layout with no exchange.
// The contraction index is permuted: MFMA step s takes k = 32 q + s from lane (i, q) (any bijection of 0..127 works as long
// as both operands use it).  A lane's 32 B-operand values of one weight row are then 128 CONTIGUOUS bytes, so the weights
// go global -> registers with eight 16-byte loads per row (a wave reads 16 rows = 8 KB back to back) -- no LDS staging and
// no barriers in the prologue, which used to be 15 of the launch's 83 us.
constexpr int GC_LD = 36;  // 32 contraction steps + 4 pad floats per (q-plane, row)
__device__ __forceinline__ int gc_idx(int row, int k) { return ((k >> 5) * GRU_RB + row) * GC_LD + (k & 31); }

// Several independent cells of one shape (the actor's and the critic's layer of the same depth) as ONE launch: blockIdx.y selects the
// cell, the persistent workgroups are divided between them.  Every workgroup pays the weight prologue (393 KB from L2) once per
// launch; two cells in one launch halve the prologues per tile and the launches per tick.
struct GruCellNets { mo_gru_cell_net n[MO_GRU_CELL_MAX_NETS]; };

__global__ __launch_bounds__(512) void k_gru_cell(int B, int nblk, GruCellNets nets) {
    const mo_gru_cell_net &net = nets.n[blockIdx.y];
    const float *__restrict__ x = net.x, *__restrict__ hprev = net.h_prev, *__restrict__ w_ih = net.w_ih, *__restrict__ w_hh = net.w_hh,
                *__restrict__ b_ih = net.b_ih, *__restrict__ b_hh = net.b_hh;
    float *__restrict__ hout = net.h_out;
    constexpr int TILE = 4 * GRU_RB * GC_LD;
    __shared__ __attribute__((aligned(16))) float xs[2][TILE], hs[2][TILE];
    __shared__ __attribute__((aligned(16))) float whn_s[8][TILE];  // W_hn B-operands of the 8 waves (64 KB), same layout as an A tile
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int c16 = l & 15, q = l >> 4;
    const int j = 16 * w + c16;
    // 160 weight registers per lane; the sixth slice (W_hn) is read from LDS every tile: 192 + accumulators + fragments do
    // not fit the 256 registers of a 512-thread workgroup (the round-1 kernel spilled 19 registers to scratch)
    float wir[32], wiz[32], win[32], whr[32], whz[32];
#define GRU_LOAD_W(dst, W, gate)                                                                   \
    {                                                                                              \
        const float4 *src = (const float4 *)((W) + (size_t)((gate) * GRU_H + j) * GRU_H + 32 * q); \
        _Pragma("unroll") for (int i = 0; i < 8; i++) {                                            \
            const float4 v = src[i];                                                               \
            dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;     \
        }                                                                                          \
    }
    GRU_LOAD_W(wir, w_ih, 0) GRU_LOAD_W(wiz, w_ih, 1) GRU_LOAD_W(win, w_ih, 2)
    GRU_LOAD_W(whr, w_hh, 0) GRU_LOAD_W(whz, w_hh, 1)
#undef GRU_LOAD_W
    float *whn_l = &whn_s[w][(q * GRU_RB + c16) * GC_LD];  // this lane's 32 values of W_hh[2H + j][32 q ..]
    {
        const float4 *src = (const float4 *)(w_hh + (size_t)(2 * GRU_H + j) * GRU_H + 32 * q);
#pragma unroll
        for (int i = 0; i < 8; i++) *(float4 *)(whn_l + 4 * i) = src[i];
    }
    const float br = b_ih[j] + b_hh[j], bz = b_ih[GRU_H + j] + b_hh[GRU_H + j], bin = b_ih[2 * GRU_H + j], bhn = b_hh[2 * GRU_H + j];
    // staging: thread e moves 4 consecutive k of one row (512 threads x float4 = one 16 x 128 tile), one 16-byte LDS store
    const int srow = tid >> 5, sk = (tid & 31) * 4;
    const int sidx = gc_idx(srow, sk);
    float4 px = make_float4(0.f, 0.f, 0.f, 0.f), ph = px;
    int blk = blockIdx.x;
#define GRU_FETCH(bk)                                                             \
    {                                                                             \
        const int row_ = (bk) * GRU_RB + srow;                                     \
        px = ph = make_float4(0.f, 0.f, 0.f, 0.f);                                 \
        if (row_ < B) {                                                           \
            px = *(const float4 *)(x + (size_t)row_ * GRU_H + sk);                 \
            ph = *(const float4 *)(hprev + (size_t)row_ * GRU_H + sk);             \
        }                                                                         \
    }
#define GRU_STAGE(buf) { *(float4 *)&xs[buf][sidx] = px; *(float4 *)&hs[buf][sidx] = ph; }
    if (blk < nblk) { GRU_FETCH(blk) GRU_STAGE(0) }
    lds_barrier();
    int cur = 0;
    for (; blk < nblk; blk += gridDim.x) {
        const int nxt = blk + gridDim.x;
        if (nxt < nblk) GRU_FETCH(nxt)  // in flight during the MFMAs below
        const float *xp = &xs[cur][(q * GRU_RB + c16) * GC_LD];
        const float *hp = &hs[cur][(q * GRU_RB + c16) * GC_LD];
        // r and z first (input and recurrent product in one accumulator each), then the two n products: the sigmoids of
        // r and z are VALU work the scheduler can run under the n-gate MFMAs
        f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, ain = ar, ahn = ar;
        // fragments one k-group ahead, pinned with scheduling barriers: left alone, the scheduler hoists all sixteen LDS reads
        // of a loop in front of it (64 more live registers) and spills weights
        f32x4 ax = *(const f32x4 *)xp, ah = *(const f32x4 *)hp;
#pragma unroll
        for (int k4 = 0; k4 < 8; k4++) {
            f32x4 nax = ax, nah = ah;
            if (k4 < 7) { nax = *(const f32x4 *)(xp + 4 * k4 + 4); nah = *(const f32x4 *)(hp + 4 * k4 + 4); }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[u], wir[4 * k4 + u], ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[u], wiz[4 * k4 + u], az, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[u], whr[4 * k4 + u], ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[u], whz[4 * k4 + u], az, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            ax = nax; ah = nah;
        }
        float rg[4], zg[4];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            rg[reg] = sigmoid_hw(ar[reg] + br);
            zg[reg] = sigmoid_hw(az[reg] + bz);
        }
        ax = *(const f32x4 *)xp; ah = *(const f32x4 *)hp;
        f32x4 bn = *(const f32x4 *)whn_l;
#pragma unroll
        for (int k4 = 0; k4 < 8; k4++) {
            f32x4 nax = ax, nah = ah, nbn = bn;
            if (k4 < 7) { nax = *(const f32x4 *)(xp + 4 * k4 + 4); nah = *(const f32x4 *)(hp + 4 * k4 + 4); nbn = *(const f32x4 *)(whn_l + 4 * k4 + 4); }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                ain = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[u], win[4 * k4 + u], ain, 0, 0, 0);
                ahn = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[u], bn[u], ahn, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            ax = nax; ah = nah; bn = nbn;
        }
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int row = 4 * q + reg;
            const float hpv = hs[cur][gc_idx(row, j)];
            const float r = rg[reg], z = zg[reg];
            const float n = tanh_hw(ain[reg] + bin + r * (ahn[reg] + bhn));
            if (blk * GRU_RB + row < B) hout[(size_t)(blk * GRU_RB + row) * GRU_H + j] = (1.f - z) * n + z * hpv;
        }
        if (nxt < nblk) GRU_STAGE(cur ^ 1)  // the other buffer: its last readers passed the barrier of the previous tile
        lds_barrier();
        cur ^= 1;
    }
#undef GRU_FETCH
#undef GRU_STAGE
}

// ---- the rollout's GRU step in fp32 ARITHMETIC ON THE bf16 MATRIX PIPE (k_gru_cell_sb) ------------------------------------------
// k_gru_cell is bound by v_mfma_f32_16x16x4_f32 (32 cycles per 16 x 16 x 4 step); the same silicon runs v_mfma_f32_16x16x32_bf16 in
// 16 cycles.  Every fp32 number is EXACTLY the sum of three bf16 numbers (24 significant bits = 3 x 8, same exponent range):
// x = x1 + x2 + x3, and a b = sum of the piece products ai bj, each exact in fp32.  Keeping the six with i + j <= 4 drops
// a2 b3 + a3 b2 + a3 b3 < 2^-23 |a b|: one fp32 rounding of the product -- the accumulation is fp32 either way.  Six bf16 MFMAs per 32
// contraction steps replace eight fp32 ones at half the cycles each: 2.67 x the matrix rate with fp32 inputs, fp32 outputs, and the
// same error against f64 as the fp32 kernel (tests/test_ops_gpu.py).  Only operands are split, nothing is stored in bf16.
// Domain: finite inputs up to bf16's largest finite value (3.39e38; fp32's is 3.40e38 -- beyond it the first piece rounds to infinity),
// which activations, weights and gradients of this model are ~30 orders of magnitude away from.
// Layout: the weights of BOTH projections as three pieces each are 590 KB -- more than a CU's registers --, so a 16-row tile is
// shared by TWO workgroups, each owning 64 hidden units (their r, z and n rows of W_ih and W_hh: 144 registers per lane as MFMA
// A-operands, split once in the prologue).  Wave (g, role): units 16 g .. 16 g + 15 of the half; role 0 accumulates r (input +
// recurrent product) and W_in x, role 1 z and W_hn h -- 72 MFMAs each per tile; the two roles of a unit group sit on the same SIMD,
// role 1 hands its two accumulators over through LDS and starts the next tile's products while role 0 does the gate math.  The x
// and h tiles are split while they are staged (one thread: 8 contraction steps of one row = one 16-byte LDS word per piece = one
// lane's B-operand) -- double buffered, one barrier per tile.  Both workgroups read whole rows of h_prev and write half rows of
// h_out: the state must NOT be updated in place (the rollout ping-pongs two buffers).
// Measured (tools/microbench/gru_cell_sb_lab.hip, per-wave stamps, 32 768 rows x 2 cells): 76 us against k_gru_cell's 127 us;
// per tile ~4 100 cycles of a SIMD at 1.9-2.0 GHz -- 2 300 of MFMA issue plus ~1 800 of vector work (splits, gate math with 24
// transcendentals per lane) that in practice does not hide under the MFMAs: moving the gate math between the roles, into the next
// tile's product phase, raising its priority or interleaving three accumulation chains each left the tile time where it was.
// test-only (include/mappo_ops_diag.h): the three pieces of every input, widened back to fp32
__global__ void k_sb_split_diag(int64_t n, const float *__restrict__ x, float *__restrict__ pieces) {
    const int64_t i = 2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const float x0 = x[i], x1 = i + 1 < n ? x[i + 1] : 0.f;
    uint32_t p[3];
    sb_split2(x0, x1, p[0], p[1], p[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        pieces[k * n + i] = __builtin_bit_cast(float, p[k] << 16);
        if (i + 1 < n) pieces[k * n + i + 1] = __builtin_bit_cast(float, p[k] & 0xffff0000u);
    }
}

#ifdef SBC_STAMP   // lab builds only (tools/microbench/gru_cell_sb_lab.hip): per-wave cycle sums of the tile loop's phases
__device__ unsigned long long sbc_stamps[8][8];
#define SBC_T(v) const unsigned long long v = __builtin_readcyclecounter();
#else
#define SBC_T(v)
#endif
constexpr int SBC_HLD = 33;                // float4 per row of the fp32 h tile (32 + 1 pad: the staging lanes are one row apart)
constexpr int SBC_TILE = 2 * 3 * 4 * 64;   // uint4 per staged tile: (x, h) x 3 pieces x 4 chunks of 32 steps x 64 lanes = 24 KB

__global__ __launch_bounds__(512) void k_gru_cell_sb(int B, int nblk, GruCellNets nets) {
    const mo_gru_cell_net &net = nets.n[blockIdx.y];
    const float *__restrict__ x = net.x, *__restrict__ hprev = net.h_prev, *__restrict__ w_ih = net.w_ih, *__restrict__ w_hh = net.w_hh;
    float *__restrict__ hout = net.h_out;
    __shared__ uint4 tile[2][SBC_TILE];              // B-operands: [(matrix, piece, chunk)][lane]
    __shared__ float4 hraw[2][GRU_RB * SBC_HLD];          // the h tile in fp32 (the state update needs h itself)
    __shared__ float4 xch[2][4][2][64];              // role-1 accumulators (z, W_hn h) of unit group g, per lane
    __shared__ float4 bias_s[4][16];                 // this half's b_r (input + recurrent), b_z, b_in, b_hn
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, i = l & 15, gq = l >> 4;
    const int half = blockIdx.x & 1, slot = blockIdx.x >> 1, nslots = gridDim.x >> 1;
    const int g = w & 3, role = w >> 2;
    const int u0 = 64 * half + 16 * g;               // first hidden unit of this wave's group
    // weights as A-operands: lane (i, gq) holds unit u0 + i, steps 32 c + 8 gq .. + 7 of chunk c
    uint4 wa[4][3], wb[4][3], wc[4][3];
    {
        const float *ra = (role ? w_ih + (size_t)(GRU_H + u0 + i) * GRU_H : w_ih + (size_t)(u0 + i) * GRU_H) + 8 * gq;          // W_iz | W_ir
        const float *rb = (role ? w_hh + (size_t)(GRU_H + u0 + i) * GRU_H : w_hh + (size_t)(u0 + i) * GRU_H) + 8 * gq;          // W_hz | W_hr
        const float *rc = (role ? w_hh + (size_t)(2 * GRU_H + u0 + i) * GRU_H : w_ih + (size_t)(2 * GRU_H + u0 + i) * GRU_H) + 8 * gq;   // W_hn | W_in
#pragma unroll
        for (int c = 0; c < 4; c++) {
            sb_split8(*(const float4 *)(ra + 32 * c), *(const float4 *)(ra + 32 * c + 4), wa[c]);
            sb_split8(*(const float4 *)(rb + 32 * c), *(const float4 *)(rb + 32 * c + 4), wb[c]);
            sb_split8(*(const float4 *)(rc + 32 * c), *(const float4 *)(rc + 32 * c + 4), wc[c]);
        }
    }
    if (tid < 64) {
        const int u = 64 * half + tid;
        float *bs = (float *)bias_s;
        bs[tid] = net.b_ih[u] + net.b_hh[u];
        bs[64 + tid] = net.b_ih[GRU_H + u] + net.b_hh[GRU_H + u];
        bs[128 + tid] = net.b_ih[2 * GRU_H + u];
        bs[192 + tid] = net.b_hh[2 * GRU_H + u];
    }
    // staging: thread -> (matrix m, 8 steps kk, row j); consecutive lanes take consecutive rows (one 16-byte LDS word each, no conflicts)
    const int sm = tid >> 8, skk = (tid >> 4) & 15, sj = tid & 15;
    const float *ssrc = (sm ? hprev : x) + 8 * skk;
    const int sdst = ((sm * 3) * 4 + (skk >> 2)) * 64 + (skk & 3) * 16 + sj;     // piece p adds p * 256
    float4 pf0, pf1;
#define SBC_FETCH(bk)                                                                  \
    {                                                                                  \
        const int row_ = (bk) * GRU_RB + sj;                                           \
        pf0 = pf1 = make_float4(0.f, 0.f, 0.f, 0.f);                                   \
        if (row_ < B) {                                                                \
            pf0 = *(const float4 *)(ssrc + (size_t)row_ * GRU_H);                      \
            pf1 = *(const float4 *)(ssrc + (size_t)row_ * GRU_H + 4);                  \
        }                                                                              \
    }
#define SBC_STAGE(buf)                                                                 \
    {                                                                                  \
        uint4 p_[3];                                                                   \
        sb_split8(pf0, pf1, p_);                                                       \
        tile[buf][sdst] = p_[0]; tile[buf][sdst + 256] = p_[1]; tile[buf][sdst + 512] = p_[2]; \
        if (sm) { hraw[buf][sj * SBC_HLD + 2 * skk] = pf0; hraw[buf][sj * SBC_HLD + 2 * skk + 1] = pf1; } \
    }
    int blk = slot;
    if (blk < nblk) { SBC_FETCH(blk) SBC_STAGE(0) }
    if (blk + nslots < nblk) SBC_FETCH(blk + nslots)
    lds_barrier();
    int cur = 0;
#ifdef SBC_STAMP
    unsigned long long s_mma = 0, s_stage = 0, s_bar = 0, s_gate = 0, s_tiles = 0;
    const unsigned long long c_begin = __builtin_readcyclecounter(), r_begin = __builtin_amdgcn_s_memrealtime();
#endif
    for (; blk < nblk; blk += nslots) {
        const int nxt = blk + nslots;
        const uint4 *tb = tile[cur] + l;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, lo0 = acc0, lo1 = acc0;
        SBC_T(t0)
        if (role == 0) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                uint4 bx[3], bh[3];
#pragma unroll
                for (int p = 0; p < 3; p++) { bx[p] = tb[(p * 4 + c) * 64]; bh[p] = tb[((3 + p) * 4 + c) * 64]; }
                sb_mma6_hl(wa[c], bx, acc0, lo0);
                sb_mma6_hl(wb[c], bh, acc0, lo0);
                sb_mma6_hl(wc[c], bx, acc1, lo1);
            }
            acc0 += lo0; acc1 += lo1;
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                uint4 bx[3], bh[3];
#pragma unroll
                for (int p = 0; p < 3; p++) { bx[p] = tb[(p * 4 + c) * 64]; bh[p] = tb[((3 + p) * 4 + c) * 64]; }
                sb_mma6_hl(wa[c], bx, acc0, lo0);
                sb_mma6_hl(wb[c], bh, acc0, lo0);
                sb_mma6_hl(wc[c], bh, acc1, lo1);
            }
            acc0 += lo0; acc1 += lo1;
            xch[cur][g][0][l] = (float4){acc0[0], acc0[1], acc0[2], acc0[3]};
            xch[cur][g][1][l] = (float4){acc1[0], acc1[1], acc1[2], acc1[3]};
        }
        // D layout: lane (i, gq), register q -> hidden unit u0 + 4 gq + q of batch row i
        const float4 hp = hraw[cur][i * SBC_HLD + (u0 >> 2) + gq];   // read before the barrier: the buffer is re-staged right after it
#ifdef SBC_STAMP
        asm volatile("s_nop 0" ::"v"(acc0), "v"(acc1));
#endif
        SBC_T(t1)
        if (nxt < nblk) SBC_STAGE(cur ^ 1)                        // tile `nxt` (fetched one tile ago) -> the other buffer
        if (nxt + nslots < nblk) SBC_FETCH(nxt + nslots)          // in flight during the next tile's products
        SBC_T(t2)
        lds_barrier();
        SBC_T(t3)
        if (role == 0) {
            const float4 zz = xch[cur][g][0][l], hn = xch[cur][g][1][l];
            const float4 br = bias_s[0][4 * g + gq], bz = bias_s[1][4 * g + gq], bin = bias_s[2][4 * g + gq], bhn = bias_s[3][4 * g + gq];
            float4 ho;
#define SBC_ONE(f, q_)                                                    \
            {                                                              \
                const float r = sigmoid_hw(acc0[q_] + br.f);               \
                const float z = sigmoid_hw(zz.f + bz.f);                   \
                const float n = tanh_hw(acc1[q_] + bin.f + r * (hn.f + bhn.f)); \
                ho.f = (1.f - z) * n + z * hp.f;                           \
            }
            SBC_ONE(x, 0) SBC_ONE(y, 1) SBC_ONE(z, 2) SBC_ONE(w, 3)
#undef SBC_ONE
            const int row = blk * GRU_RB + i;
            if (row < B) *(float4 *)(hout + (size_t)row * GRU_H + u0 + 4 * gq) = ho;
        }
#ifdef SBC_STAMP
        {
            SBC_T(t4)
            s_mma += t1 - t0; s_stage += t2 - t1; s_bar += t3 - t2; s_gate += t4 - t3; s_tiles++;
        }
#endif
        cur ^= 1;
    }
#ifdef SBC_STAMP
    if (blockIdx.x == 6 && blockIdx.y == 0 && l == 0) {
        unsigned long long *o = sbc_stamps[w];
        o[0] = s_mma; o[1] = s_stage; o[2] = s_bar; o[3] = s_gate; o[4] = s_tiles;
        o[5] = __builtin_readcyclecounter() - c_begin; o[6] = __builtin_amdgcn_s_memrealtime() - r_begin;
    }
#endif
#undef SBC_FETCH
#undef SBC_STAGE
}

// ---- the sequence kernels: TRANSPOSED tiles ------------------------------------------------------------------------------------
// A = weights, B = the h / dgh tile, so the MFMA result is gate^T: a lane owns FOUR CONSECUTIVE hidden units of ONE batch row
// (rounds 1-2 computed gate[row][unit] tiles: one unit of four rows per lane, every global access a scalar in its own 64-byte
// segment, 32 / 48 vector-memory instructions per lane and step).  gi / out / dout / dgi move as 16-byte accesses (3 + 1 + 4 in the
// forward step), the h tile is read and written with one ds_read_b128 / ds_write_b128 per lane, and the saved gates -- private to
// this kernel pair -- are stored in lane order (save[t][block][plane][thread] float4: 1 KB contiguous per wave instruction).  The
// contraction index is permuted so that a lane's four units are contiguous in the operand tile as well: MFMA step s of lane
// group q takes k = 32 q + s (forward; 96 q + s in the backward product over the 384 gate columns), k_gru_cell's bijection.
// Backward: dr and dz are no longer stored twice.  With time-major dgi rows (gi_agents == 0) the kernel writes dgi = (dr, dz, dn)
// and a separate dnr [T][B][H]; dW_hh = [dr dz | dnr]^T h_prev then takes its first 256 rows from dgi and the last 128 from dnr
// (-0.5 GB of stores per call, 859 -> 739 us).  With the encoder's row order (gi_agents > 0: dgi and h_prev rows do not line up)
// the full dgh = (dr, dz, dnr) is still written.  The gate-gradient tile is double buffered: one barrier per step instead of two.
// What bounds them (tools/microbench/gru_step_lab.hip, s_memtime stamps per wave): per step and SIMD the two waves
// issue 2 x 96 MFMAs = 6 144 pipe cycles; the older wave's gate math hides under the younger's MFMAs, the younger's (~860 cycles),
// the store issue (~500) and the first operand fetch after the barrier (~230) do not: 8 470 cycles per step with memory traffic,
// 7 660 without.  The instruction count of the memory phase and the barrier flavour do not move the time; bytes do (~80 us / GB).
constexpr int GRU2_LD = 36, GRU2_LD3 = 100;   // floats per (plane, row): 32 (96) operand slots + 4 pad, rows stay 16-byte aligned
__device__ __forceinline__ int gru2_hidx(int row, int k) { return ((k >> 5) * GRU_RB + row) * GRU2_LD + (k & 31); }

// Several independent layers of the same shape (the actor's and the critic's: different weights, same T and B) run as ONE launch:
// blockIdx.y selects the layer.  At the benchmark's mini-batch a layer is 205 workgroups; at a data-parallel rank's share (26-103
// workgroups) a layer alone leaves most CUs idle for T sequential steps, and the launch takes as long as at full size.
struct GruFwdNets { mo_gru_seq_net n[MO_GRU_MAX_NETS]; };
struct GruBwdNets { mo_gru_seq_bwd_net n[MO_GRU_MAX_NETS]; };

__global__ __launch_bounds__(512) void k_gru_seq_fwd2(int T, int Bmax, GruFwdNets nets, int gi_agents) {
    const mo_gru_seq_net &net = nets.n[blockIdx.y];
    const int B = net.B > 0 ? net.B : Bmax;                      // layers of one launch may differ in their number of sequences
    if ((int)blockIdx.x * GRU_RB >= B) return;                    // (uniform per workgroup, before any barrier)
    const float *__restrict__ gi = net.gi, *__restrict__ w_hh = net.w_hh, *__restrict__ b_hh = net.b_hh, *__restrict__ h0 = net.h0;
    float *__restrict__ out = net.out, *__restrict__ save = net.save;
    __shared__ __attribute__((aligned(16))) float hs[2][4 * GRU_RB * GRU2_LD];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int b0 = blockIdx.x * GRU_RB;
    const int c16 = l & 15, q = l >> 4;
    // A operand: lane (m = c16, k-group q) holds W[gate][16 w + c16][32 q + s], s = 0..31 -- 128 contiguous bytes per gate
    float wr[32], wz[32], wn[32];
#define GRU2_LOAD_W(dst, gate)                                                                              \
    {                                                                                                       \
        const float4 *src = (const float4 *)(w_hh + (size_t)((gate) * GRU_H + 16 * w + c16) * GRU_H + 32 * q); \
        _Pragma("unroll") for (int i = 0; i < 8; i++) {                                                     \
            const float4 v = src[i];                                                                        \
            dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;              \
        }                                                                                                   \
    }
    GRU2_LOAD_W(wr, 0) GRU2_LOAD_W(wz, 1) GRU2_LOAD_W(wn, 2)
#undef GRU2_LOAD_W
    // D^T tile: this lane owns batch row c16, hidden units u0 .. u0 + 3
    const int u0 = 16 * w + 4 * q, row = c16;
    const bool live = b0 + row < B;
    const float4 br = *(const float4 *)(b_hh + u0), bz = *(const float4 *)(b_hh + GRU_H + u0), bn = *(const float4 *)(b_hh + 2 * GRU_H + u0);
    {
        const int srow = tid >> 5, sk = (tid & 31) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b0 + srow < B) v = *(const float4 *)(h0 + (size_t)(b0 + srow) * GRU_H + sk);
        *(float4 *)&hs[0][gru2_hidx(srow, sk)] = v;
    }
    lds_barrier();
    const int hpos = gru2_hidx(row, u0);
    const size_t nblk = (size_t)(B + GRU_RB - 1) / GRU_RB;   // of THIS layer (the save area is laid out per layer)
    float4 *sv = save ? (float4 *)save + (size_t)blockIdx.x * 4 * 512 + tid : nullptr;
    int cur = 0;
    for (int t = 0; t < T; t++) {
        float4 gr = make_float4(0.f, 0.f, 0.f, 0.f), gz = gr, gn = gr;
        if (live) {
            const float *g = gi + gru_gi_row(b0 + row, t, T, B, gi_agents) * 3 * GRU_H + u0;
            gr = *(const float4 *)g; gz = *(const float4 *)(g + GRU_H); gn = *(const float4 *)(g + 2 * GRU_H);
        }
        const float *hp = &hs[cur][(q * GRU_RB + c16) * GRU2_LD];   // B operand: h[row c16][32 q + s]
        f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar;
#pragma unroll
        for (int k4 = 0; k4 < 8; k4++) {
            const f32x4 a = *(const f32x4 *)(hp + 4 * k4);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[4 * k4 + u], a[u], ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[4 * k4 + u], a[u], az, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[4 * k4 + u], a[u], an, 0, 0, 0);
            }
        }
        const float4 hprev = *(const float4 *)&hs[cur][hpos];
        float4 r, z, hn, n, hnew;
#define GRU2_ONE(f, i)                                   \
        r.f = sigmoid_hw(gr.f + ar[i] + br.f);           \
        z.f = sigmoid_hw(gz.f + az[i] + bz.f);           \
        hn.f = an[i] + bn.f;                             \
        n.f = tanh_hw(gn.f + r.f * hn.f);                \
        hnew.f = (1.f - z.f) * n.f + z.f * hprev.f;
        GRU2_ONE(x, 0) GRU2_ONE(y, 1) GRU2_ONE(z, 2) GRU2_ONE(w, 3)
#undef GRU2_ONE
        *(float4 *)&hs[cur ^ 1][hpos] = hnew;
        if (live) *(float4 *)(out + ((size_t)t * B + b0 + row) * GRU_H + u0) = hnew;
        if (sv) {   // lane order; rows past B are padding of the (opaque) save area
            float4 *s4 = sv + (size_t)t * nblk * 4 * 512;
            s4[0] = r; s4[512] = z; s4[1024] = n; s4[1536] = hn;
        }
        lds_barrier();
        cur ^= 1;
    }
}

__global__ __launch_bounds__(512) void k_gru_seq_bwd2(int T, int Bmax, GruBwdNets nets, int gi_agents) {
    const mo_gru_seq_bwd_net &net = nets.n[blockIdx.y];
    const int B = net.B > 0 ? net.B : Bmax;
    if ((int)blockIdx.x * GRU_RB >= B) return;
    const float *__restrict__ dout = net.dout, *__restrict__ save = net.save, *__restrict__ out = net.out, *__restrict__ h0 = net.h0,
                *__restrict__ w_hh = net.w_hh;
    float *__restrict__ dgi = net.dgi, *__restrict__ dgh = net.dgh, *__restrict__ dnr_out = net.dnr, *__restrict__ dh0 = net.dh0;
    float *__restrict__ bias_partials = net.db_ih ? (float *)net.workspace : nullptr;
    __shared__ __attribute__((aligned(16))) float gs[2][4 * GRU_RB * GRU2_LD3];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int b0 = blockIdx.x * GRU_RB;
    const int c16 = l & 15, q = l >> 4;
    // A operand = W_hh^T tile: lane (m = c16, k-group q) holds W_hh[96 q + s][16 w + c16], s = 0..95
    float wb[96];
#pragma unroll
    for (int s = 0; s < 96; s++) wb[s] = w_hh[(size_t)(96 * q + s) * GRU_H + 16 * w + c16];
    const int u0 = 16 * w + 4 * q, row = c16;
    const bool live = b0 + row < B;
    // where this lane's (dr, dz, dnr) quads go in the operand tile: gate column k0 = 128 g + u0 -> plane k0 / 96, slot k0 % 96
    int gpos[3];
#pragma unroll
    for (int g = 0; g < 3; g++) {
        const int k0 = GRU_H * g + u0;
        gpos[g] = ((k0 / 96) * GRU_RB + row) * GRU2_LD3 + k0 % 96;
    }
    const size_t nblk = (size_t)(B + GRU_RB - 1) / GRU_RB;   // of THIS layer (the save area is laid out per layer)
    const float4 *sv = (const float4 *)save + (size_t)blockIdx.x * 4 * 512 + tid;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sb_r = zero4, sb_z = zero4, sb_n = zero4, sb_nr = zero4;
    float4 dcarry = zero4;
    float4 pr, pz, pn, phn, php, pdo;   // one step ahead: none of them depends on the recurrence
    auto prefetch = [&](int t) {
        pr = pz = pn = phn = php = pdo = zero4;
        if (live) {
            const float4 *s4 = sv + (size_t)t * nblk * 4 * 512;
            pr = s4[0]; pz = s4[512]; pn = s4[1024]; phn = s4[1536];
            const size_t o = (size_t)(b0 + row) * GRU_H + u0;
            php = *(const float4 *)(t > 0 ? out + (size_t)(t - 1) * B * GRU_H + o : h0 + o);
            pdo = *(const float4 *)(dout + (size_t)t * B * GRU_H + o);
        }
    };
    prefetch(T - 1);
    int buf = 0;
    for (int t = T - 1; t >= 0; t--) {
        const float4 r = pr, z = pz, n = pn, hn = phn, hp = php, dO = pdo;
        if (t > 0) prefetch(t - 1);
        float4 dr, dz, dn, dnr, dhz;
#define GRU2_ONE(f)                                              \
        {                                                        \
            const float dh = dO.f + dcarry.f;                    \
            dn.f = dh * (1.f - z.f) * (1.f - n.f * n.f);         \
            dz.f = dh * (hp.f - n.f) * z.f * (1.f - z.f);        \
            dr.f = dn.f * hn.f * r.f * (1.f - r.f);              \
            dnr.f = dn.f * r.f;                                  \
            dhz.f = dh * z.f;                                    \
        }
        GRU2_ONE(x) GRU2_ONE(y) GRU2_ONE(z) GRU2_ONE(w)
#undef GRU2_ONE
        if (live) {
            float *g = dgi + gru_gi_row(b0 + row, t, T, B, gi_agents) * 3 * GRU_H + u0;
            *(float4 *)g = dr; *(float4 *)(g + GRU_H) = dz; *(float4 *)(g + 2 * GRU_H) = dn;
            const size_t tb = (size_t)t * B + b0 + row;
            if (dgh) {
                float *h = dgh + tb * 3 * GRU_H + u0;
                *(float4 *)h = dr; *(float4 *)(h + GRU_H) = dz; *(float4 *)(h + 2 * GRU_H) = dnr;
            } else {
                *(float4 *)(dnr_out + tb * GRU_H + u0) = dnr;
            }
#define GRU2_ACC(S, V) S.x += V.x; S.y += V.y; S.z += V.z; S.w += V.w;
            GRU2_ACC(sb_r, dr) GRU2_ACC(sb_z, dz) GRU2_ACC(sb_n, dn) GRU2_ACC(sb_nr, dnr)
#undef GRU2_ACC
        }
        float *gb = gs[buf];   // dead rows carry zeros (their loads were skipped)
        *(float4 *)(gb + gpos[0]) = dr; *(float4 *)(gb + gpos[1]) = dz; *(float4 *)(gb + gpos[2]) = dnr;
        lds_barrier();
        const float *gp = gb + (q * GRU_RB + c16) * GRU2_LD3;   // B operand: dgh[row c16][96 q + s]
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
#pragma unroll
        for (int k4 = 0; k4 < 24; k4 += 3) {
            const f32x4 x0 = *(const f32x4 *)(gp + 4 * k4), x1 = *(const f32x4 *)(gp + 4 * k4 + 4), x2 = *(const f32x4 *)(gp + 4 * k4 + 8);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[4 * k4 + u], x0[u], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[4 * k4 + 4 + u], x1[u], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[4 * k4 + 8 + u], x2[u], a2, 0, 0, 0);
            }
        }
        dcarry.x = dhz.x + (a0[0] + a1[0] + a2[0]); dcarry.y = dhz.y + (a0[1] + a1[1] + a2[1]);
        dcarry.z = dhz.z + (a0[2] + a1[2] + a2[2]); dcarry.w = dhz.w + (a0[3] + a1[3] + a2[3]);
        buf ^= 1;   // the other tile: its last readers passed this step's barrier before anyone writes it again
    }
    if (live) *(float4 *)(dh0 + (size_t)(b0 + row) * GRU_H + u0) = dcarry;
    // bias gradients: sums over this workgroup's 16 rows (the lanes of a 16-lane group) and all steps
    if (bias_partials) {
        float s[16] = {sb_r.x, sb_r.y, sb_r.z, sb_r.w, sb_z.x, sb_z.y, sb_z.z, sb_z.w, sb_n.x, sb_n.y, sb_n.z, sb_n.w, sb_nr.x, sb_nr.y, sb_nr.z, sb_nr.w};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            s[i] += __shfl_xor(s[i], 1); s[i] += __shfl_xor(s[i], 2); s[i] += __shfl_xor(s[i], 4); s[i] += __shfl_xor(s[i], 8);
        }
        if (c16 == 0) {
            float *bp = bias_partials + (size_t)blockIdx.x * 4 * GRU_H + u0;
#pragma unroll
            for (int g = 0; g < 4; g++) *(float4 *)(bp + g * GRU_H) = make_float4(s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]);
        }
    }
}

__global__ void k_gru_bias_reduce(int nblk, const float *partials, float *db_ih, float *db_hh) {
    const int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // 4 * H columns, one wave each
    const int lane = threadIdx.x & 63;
    if (idx >= 4 * GRU_H) return;
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += (double)partials[(size_t)b * 4 * GRU_H + idx];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
        const int g = idx / GRU_H, jj = idx - g * GRU_H;
        if (g == 0) { db_ih[jj] = (float)s; db_hh[jj] = (float)s; }
        else if (g == 1) { db_ih[GRU_H + jj] = (float)s; db_hh[GRU_H + jj] = (float)s; }
        else if (g == 2) db_ih[2 * GRU_H + jj] = (float)s;
        else db_hh[2 * GRU_H + jj] = (float)s;
    }
}

int check_msg(int R, int P, int K, int E, int din, int q_div, int adj_mode, const void *adj, const void *kvalid, const void *e) {
    if (R < 0 || P < 1 || P > MAX_P || K < 1 || E < 64 || E > 256 || (E & 63) || (din != 4 && din != 8) || q_div < 1) return MO_ERR_BAD_ARG;
    if ((adj_mode == MO_ADJ_TENSOR || adj_mode == MO_ADJ_BITS) && !adj) return MO_ERR_BAD_ARG;
    if (adj_mode == MO_ADJ_VALID && !kvalid) return MO_ERR_BAD_ARG;
    if (adj_mode < 0 || adj_mode > 3) return MO_ERR_BAD_ARG;
    if (din == 8 && !e) return MO_ERR_BAD_ARG;
    return 0;
}

constexpr int BWD_BLOCKS = 2048, FWD_BLOCKS = 8192;

bool msg_q_small(int K) { return 4 * K <= 64; }
bool msg_adj_small(int P, int K, int adj_mode) {
    if (adj_mode == MO_ADJ_TENSOR) return P * K <= 64;
    if (adj_mode == MO_ADJ_BITS) return P * MO_ADJ_ROW_WORDS(K) <= 128;
    return true;
}

MsgDims msg_dims(int R, int P, int K, int E, int din, int q_div, int adj_mode, int64_t p_rs, int64_t q_rs, int64_t e_rs, int64_t adj_rs,
                 int64_t o_is, int max_blocks, int *grid) {
    const int g0 = R < max_blocks ? (R > 0 ? R : 1) : max_blocks;
    const int rpb = (R + g0 - 1) / g0;
    *grid = (R + rpb - 1) / (rpb > 0 ? rpb : 1);
    if (*grid < 1) *grid = 1;
    return MsgDims{R, P, K, E, din, q_div, adj_mode, rpb > 0 ? rpb : 1, p_rs, q_rs, e_rs, adj_rs, o_is};
}

// ---- PPO policy / value loss of one mini-batch, forward and gradients in one pass (DHGN/mappo_parallel.py:692-706) -------
// Element-wise fp32 arithmetic in torch's op order (so the values match the op-by-op graph), masked sums in f64 through
// per-block partials (no atomics: deterministic).  The gradients w.r.t. logp_now / entropy / values_now are produced in the
// same pass with autograd's tie rules: min / max send half the gradient to each side of an exact tie (inside the clip
// range surr1 == surr2, and both halves reach the ratio), clamp passes the gradient on its closed range.
constexpr int PPO_BLOCKS = 256;
struct PpoElem { float la, lc, g_lp, g_ent, g_v; };
__device__ __forceinline__ PpoElem ppo_elem(float lp_now, float ent, float lp_old, float a, float act, float v_now, float v_old, float v_tgt, float inv,
                                            float eps, float ent_coef, int value_clip) {
    PpoElem o;
    const float ratio = expf(lp_now - lp_old);
    const float surr1 = ratio * a;
    const float rc = fminf(fmaxf(ratio, 1.f - eps), 1.f + eps);
    const float surr2 = rc * a;
    o.la = -fminf(surr1, surr2) - ent_coef * ent;
    const bool inside = ratio >= 1.f - eps && ratio <= 1.f + eps;
    const float w1 = surr1 < surr2 ? 1.f : (surr1 == surr2 ? 0.5f : 0.f);   // share of min() that flows to surr1
    const float w2 = surr2 < surr1 ? 1.f : (surr1 == surr2 ? 0.5f : 0.f);   // ... to surr2 (reaches the ratio inside the clip range)
    const float up = act * inv;                                               // d loss / d la
    o.g_lp = -up * (w1 + (inside ? w2 : 0.f)) * a * ratio;
    o.g_ent = -up * ent_coef;
    const float eo = v_now - v_tgt;
    float gv;
    if (value_clip) {
        const float d = v_now - v_old;
        const float dc = fminf(fmaxf(d, -eps), eps);
        const float ec = (dc + v_old) - v_tgt;
        const float qa = ec * ec, qb = eo * eo;
        o.lc = fmaxf(qa, qb);
        const float wa = qa > qb ? 1.f : (qa == qb ? 0.5f : 0.f), wb = qb > qa ? 1.f : (qa == qb ? 0.5f : 0.f);
        const bool din = d >= -eps && d <= eps;
        gv = wa * 2.f * ec * (din ? 1.f : 0.f) + wb * 2.f * eo;
    } else {
        o.lc = eo * eo;
        gv = 2.f * eo;
    }
    o.g_v = up * gv;
    return o;
}

__device__ __forceinline__ void ppo_block_sums(double sa, double sc, double *partials) {
    __shared__ double red[2][256];
    red[0][threadIdx.x] = sa; red[1][threadIdx.x] = sc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = red[0][0]; partials[2 * blockIdx.x + 1] = red[1][0]; }
}

__global__ __launch_bounds__(256) void k_ppo_loss(long n, const float *lp_now, const float *ent, const float *lp_old, const float *adv,
                                                  const float *active, const float *v_now, const float *v_old, const float *v_tgt,
                                                  const float *active_sum, float eps, float ent_coef, int value_clip, float *g_lp,
                                                  float *g_ent, float *g_v, double *partials) {
    const float inv = 1.f / active_sum[0];
    double sa = 0.0, sc = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float act = active[i];
        const PpoElem e = ppo_elem(lp_now[i], ent[i], lp_old[i], adv[i], act, v_now[i], value_clip ? v_old[i] : 0.f, v_tgt[i], inv, eps, ent_coef, value_clip);
        sa += (double)(e.la * act);
        sc += (double)(e.lc * act);
        g_lp[i] = e.g_lp; g_ent[i] = e.g_ent; g_v[i] = e.g_v;
    }
    ppo_block_sums(sa, sc, partials);
}

// The same loss from the policy's PROBABILITIES: torch.distributions.Categorical(prob) -- renormalisation, probs_to_logits' clamp to
// [eps, 1 - eps], log_prob's gather and entropy() (DHGN/mappo_parallel.py:451-456) -- evaluated in registers in torch's op order, and
// the gradient with respect to prob written straight out (autograd's rules: the clamp passes the gradient on its closed range).  One
// launch instead of Categorical's ~16 element-wise kernels over (rows, A) and their backward.  prob / its gradient and the values are
// read through 3-D views (index (i0, i1, i2) of the (mini-batch, T, P) rows): the heads' outputs are time-major.
struct PpoView { long d1, d2, s0, s1, s2; };
constexpr int PPO_MAX_A = 16;
__global__ __launch_bounds__(256) void k_ppo_loss_prob(long n, int A, const float *__restrict__ prob, PpoView pv, const float *__restrict__ action,
                                                       const float *lp_old, const float *adv, const float *active, const float *__restrict__ v_now, PpoView vv,
                                                       const float *v_old, const float *v_tgt, const float *active_sum, float eps, float ent_coef,
                                                       int value_clip, float *__restrict__ g_prob, float *__restrict__ g_v, double *partials) {
    const float inv = 1.f / active_sum[0];
    constexpr float P_EPS = 1.1920928955078125e-07f;      // torch.finfo(torch.float32).eps (clamp_probs)
    double sa = 0.0, sc = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long i2 = i % pv.d2, i01 = i / pv.d2, i1 = i01 % pv.d1, i0 = i01 / pv.d1;
        const long po = i0 * pv.s0 + i1 * pv.s1 + i2 * pv.s2;
        float p[PPO_MAX_A], l[PPO_MAX_A];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < PPO_MAX_A; k++) { p[k] = k < A ? prob[po + k] : 0.f; if (k < A) s += p[k]; }
        const int a_idx = (int)action[i];
        float lp = 0.f, plp = 0.f;
#pragma unroll
        for (int k = 0; k < PPO_MAX_A; k++)
            if (k < A) {
                p[k] = p[k] / s;                                              // Categorical.probs
                l[k] = logf(fminf(fmaxf(p[k], P_EPS), 1.f - P_EPS));          // probs_to_logits
                if (k == a_idx) lp = l[k];
                plp += l[k] * p[k];
            }
        const float act = active[i];
        const float vn = v_now[i0 * vv.s0 + i1 * vv.s1 + i2 * vv.s2];
        const PpoElem e = ppo_elem(lp, -plp, lp_old[i], adv[i], act, vn, value_clip ? v_old[i] : 0.f, v_tgt[i], inv, eps, ent_coef, value_clip);
        sa += (double)(e.la * act);
        sc += (double)(e.lc * act);
        g_v[i] = e.g_v;
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < PPO_MAX_A; k++)
            if (k < A) {
                const float c = fminf(fmaxf(p[k], P_EPS), 1.f - P_EPS);
                const bool pass = p[k] >= P_EPS && p[k] <= 1.f - P_EPS;
                const float gl = (k == a_idx ? e.g_lp : 0.f) - e.g_ent * p[k];   // d / d logits[k]: log_prob's gather, entropy's logits * probs
                l[k] = (pass ? gl / c : 0.f) - e.g_ent * l[k];                    // d / d probs[k]
                dot += l[k] * p[k];
            }
#pragma unroll
        for (int k = 0; k < PPO_MAX_A; k++)
            if (k < A) g_prob[po + k] = (l[k] - dot) / s;                        // through probs = prob / prob.sum(-1)
    }
    ppo_block_sums(sa, sc, partials);
}

__global__ void k_ppo_loss_finish(int nblk, const double *partials, const float *active_sum, float *losses) {
    if (threadIdx.x < 2) {
        double s = 0.0;
        for (int b = 0; b < nblk; b++) s += partials[2 * b + threadIdx.x];
        losses[threadIdx.x] = (float)s / active_sum[0];
    }
}

// ---- per-tick recording of the rollout into the replay buffer ------------------------------------------------------------
// One launch instead of ~10 strided tensor copies: workgroup n moves environment n's row of every item from the rollout's
// static tensors into slot [n, t] of the (N, T, ...) buffer tensors (dst rows are dst_row_stride bytes apart), converts the
// int32 actions to the buffer's float32 on the way, and adds the tick's raw team reward to the episode return.
struct RecordArgs {
    mo_record_item it[MO_RECORD_MAX_ITEMS];
    int n_items;
    const float *raw;  // [N][P] or null
    float *ret;        // [N]
    int P;
};

__global__ __launch_bounds__(256) void k_rollout_record(RecordArgs a) {
    const int n = blockIdx.x, tid = threadIdx.x;
    for (int k = 0; k < a.n_items; k++) {
        const mo_record_item &it = a.it[k];
        const char *src = (const char *)it.src + (size_t)n * it.row_bytes;
        char *dst = (char *)it.dst + (size_t)n * it.dst_row_stride;
        if (it.i32_to_f32) {
            for (int i = tid; i < (it.row_bytes >> 2); i += 256) ((float *)dst)[i] = (float)((const int32_t *)src)[i];
        } else if (((it.row_bytes | it.dst_row_stride) & 15) == 0 && ((((uintptr_t)it.src) | ((uintptr_t)it.dst)) & 15) == 0) {
            for (int i = tid; i < (it.row_bytes >> 4); i += 256) ((float4 *)dst)[i] = ((const float4 *)src)[i];
        } else {
            for (int i = tid; i < (it.row_bytes >> 2); i += 256) ((float *)dst)[i] = ((const float *)src)[i];
        }
    }
    if (a.raw && tid == 0) {
        float s = 0.f;
        for (int p = 0; p < a.P; p++) s += a.raw[(size_t)n * a.P + p];
        a.ret[n] += s;
    }
}

// ---- weight-gradient GEMM  C[M][N] = A^T B  (A [K][M], B [K][N] row-major, K ~ 5e5 rows, M, N <= 384) -------------
// The (rows x features) activations / output gradients of a Linear or GRU projection are reduced over every row of the
// minibatch: a tall-skinny "TN" GEMM whose whole output fits the accumulators of ONE workgroup.  Split-K: workgroup x
// owns a contiguous range of 4-row k-steps and the full (128 AM) x (128 BN) tile, 2 x 2 waves of (64 AM) x (64 BN) each,
// all accumulators in registers (AM * BN * 16 tiles of v_mfma_f32_16x16x4_f32 per wave).  Both operands are K-major, which
// is exactly the MFMA A/B lane order (lane = 16 * k + i): they go from global memory straight into the MFMA operand
// registers with 16-byte loads and never touch LDS -- lane (i, k) loads 4 consecutive features [4i, 4i+4) of row k, and
// feature 4i + t is declared row i of tile t (a permutation of the output rows undone when the partial is stored).
// Loads of the next chunk of k-steps are issued before the MFMAs of the current one (register double buffer).  The
// partial tiles go to `part` [S][M][N]; k_wgrad_reduce adds them in a fixed order (deterministic, no atomics).
typedef float v4f __attribute__((ext_vector_type(4)));

template <int AM, int BN, int U>
struct WgradRegs {
    float4 a[U][AM], b[U][BN];
};

template <int AM, int BN, int U>
__device__ __forceinline__ void wgrad_load(WgradRegs<AM, BN, U> &r, const float *pa, const float *pb, int64_t lda4, int64_t ldb4) {
#pragma unroll
    for (int u = 0; u < U; u++) {
#pragma unroll
        for (int a = 0; a < AM; a++) r.a[u][a] = *(const float4 *)(pa + u * lda4 + a * 64);
#pragma unroll
        for (int b = 0; b < BN; b++) r.b[u][b] = *(const float4 *)(pb + u * ldb4 + b * 64);
    }
}

template <int AM, int BN, int U>
__device__ __forceinline__ void wgrad_mma(const WgradRegs<AM, BN, U> &r, v4f (&acc)[AM][4][BN][4]) {
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int a = 0; a < AM; a++) {
            const float av[4] = {r.a[u][a].x, r.a[u][a].y, r.a[u][a].z, r.a[u][a].w};
#pragma unroll
            for (int b = 0; b < BN; b++) {
                const float bv[4] = {r.b[u][b].x, r.b[u][b].y, r.b[u][b].z, r.b[u][b].w};
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int v = 0; v < 4; v++) acc[a][t][b][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[v], acc[a][t][b][v], 0, 0, 0);
            }
        }
}

template <int AM, int BN>
__global__ __launch_bounds__(256) void k_wgrad(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb, int64_t K,
                                               int M, int N, float *__restrict__ part) {
    constexpr int U = 4;  // k-steps per pipeline stage (measured: 2 exposes load latency, 6+ lengthens the unpipelined ends)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 15, kk = lane >> 4;
    const int m0 = blockIdx.y * 128 * AM + (wave >> 1) * 64 * AM, n0 = blockIdx.z * 128 * BN + (wave & 1) * 64 * BN;
    const int64_t steps = K >> 2;  // full 4-row k-steps; rows 4*steps .. K-1 are the masked tail of the last workgroup
    const int64_t s_beg = steps * blockIdx.x / gridDim.x, s_end = steps * (blockIdx.x + 1) / gridDim.x;
    const int64_t lda4 = 4 * lda, ldb4 = 4 * ldb;
    const float *pa = A + (4 * s_beg + kk) * lda + m0 + 4 * i;
    const float *pb = B + (4 * s_beg + kk) * ldb + n0 + 4 * i;
    v4f acc[AM][4][BN][4];
#pragma unroll
    for (int a = 0; a < AM; a++)
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int b = 0; b < BN; b++)
#pragma unroll
                for (int v = 0; v < 4; v++) acc[a][t][b][v] = (v4f){0.f, 0.f, 0.f, 0.f};
    // Straight-line pipeline (no branch around a load: the s_waitcnt counts stay exact): pairs of U-step chunks, the
    // loads of the chunk after next are issued before the MFMAs of the current one; past the end the last chunk is
    // re-loaded (discarded) instead of branching.
    const int64_t n_s = s_end - s_beg, pairs = n_s / (2 * U);
    WgradRegs<AM, BN, U> r0, r1;
    if (pairs > 0) wgrad_load<AM, BN, U>(r0, pa, pb, lda4, ldb4);
    for (int64_t c = 0; c < pairs; c++) {
        wgrad_load<AM, BN, U>(r1, pa + U * lda4, pb + U * ldb4, lda4, ldb4);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (the scheduler would sink it to its use)
        wgrad_mma<AM, BN, U>(r0, acc);
        __builtin_amdgcn_sched_barrier(0);
        const int64_t adv = (c + 1 < pairs) ? 2 * U : U;
        pa += adv * lda4;
        pb += adv * ldb4;
        wgrad_load<AM, BN, U>(r0, pa, pb, lda4, ldb4);
        __builtin_amdgcn_sched_barrier(0);
        wgrad_mma<AM, BN, U>(r1, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    {   // the n_s % 2U left-over steps, then (last workgroup) the K % 4 tail rows, zero-filled
        pa = A + (4 * (s_beg + pairs * 2 * U) + kk) * lda + m0 + 4 * i;
        pb = B + (4 * (s_beg + pairs * 2 * U) + kk) * ldb + n0 + 4 * i;
        WgradRegs<AM, BN, 1> r;
        for (int64_t s = s_beg + pairs * 2 * U; s < s_end; s++) {
            wgrad_load<AM, BN, 1>(r, pa, pb, lda4, ldb4);
            wgrad_mma<AM, BN, 1>(r, acc);
            pa += lda4;
            pb += ldb4;
        }
        if (blockIdx.x == gridDim.x - 1 && (K & 3)) {
            const bool live = 4 * steps + kk < K;
#pragma unroll
            for (int a = 0; a < AM; a++) r.a[0][a] = live ? *(const float4 *)(pa + a * 64) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int b = 0; b < BN; b++) r.b[0][b] = live ? *(const float4 *)(pb + b * 64) : make_float4(0.f, 0.f, 0.f, 0.f);
            wgrad_mma<AM, BN, 1>(r, acc);
        }
    }
    // D tile (a,t | b,v): lane l, register q holds row 4 (l / 16) + q, column l % 16 of the tile, i.e. output row
    // m0 + 64 a + 16 (l / 16) + 4 q + t and column n0 + 64 b + 4 (l % 16) + v: the four v's are one 16-byte store
    float *po = part + ((size_t)blockIdx.x * M + m0 + 16 * kk) * N + n0 + 4 * i;
#pragma unroll
    for (int a = 0; a < AM; a++)
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int b = 0; b < BN; b++)
                    *(float4 *)(po + (size_t)(64 * a + 4 * q + t) * N + 64 * b) =
                        make_float4(acc[a][t][b][0][q], acc[a][t][b][1][q], acc[a][t][b][2][q], acc[a][t][b][3][q]);
}

// C[m][n] = (accumulate ? C[m][n] : 0) + sum_x part[x][m][n] in a fixed order (the same every run): wave q of a workgroup adds the
// partials x = q, q + 4, ... of 64 outputs (8 loads in flight per lane: one thread walking all 256 partials was a 18 us chain of
// dependent-latency loads, a quarter of the weight-gradient launch it follows at a data-parallel rank's batch), then (s0 + s1) + (s2 + s3)
__global__ __launch_bounds__(256) void k_wgrad_reduce(int S, int MN, const float *__restrict__ part, float *__restrict__ C, int accumulate) {
    __shared__ float sm[4][64];
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (idx < MN) {
#pragma unroll 8
        for (int x = q; x < S; x += 4) s += part[(size_t)x * MN + idx];
    }
    sm[q][lane] = s;
    __syncthreads();
    if (q == 0 && idx < MN) {
        const float t = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
        C[idx] = accumulate ? C[idx] + t : t;
    }
}

// ---- weight gradient  C[M][N] = A^T B in the split arithmetic (k_sb_wgrad; A [K][M], B [K][N], K ~ 5e5 rows) ---------------------
// One workgroup (8 waves, two per SIMD: one wave's splitting runs under the other's MFMAs) owns the WHOLE M x N output in its
// accumulators and a contiguous range of the rows (split-K, partials reduced in a fixed order afterwards).  Rows arrive in chunks
// of 16: a thread loads 4 consecutive rows x 4 features with 16-byte loads -- a wave reads 1 KB contiguous pieces of a row --,
// splits them and writes, per feature and piece, its 4 rows as half of a 16-byte LDS word that holds 8 consecutive rows of one
// feature: exactly the 8 contraction steps a lane feeds to v_mfma_f32_32x32x16_bf16 (lane (i, g): tile row/column i, steps
// 8 g .. 8 g + 7), so an operand is one ds_read_b128 and the transposition K-major -> feature-major costs nothing.  Features
// inside a 64-byte block are XOR-swizzled by (feature / 8) % 4 to spread the writes (lane stride 64 bytes) over the banks; readers
// of 32 consecutive features stay conflict-free.  The LDS image is double-buffered (2 x 48 KB): chunk c + 1 is split and written
// while chunk c is multiplied, one barrier per chunk; the raw rows of the next SB_DEPTH chunks are in flight in registers (128 KB
// per CU: one chunk ahead left the HBM latency exposed, 3.4 us under load against a 2 us matrix phase).
constexpr int SB_DEPTH = 4;

template <int MT, int NT>
struct SbWgCfg {
    static constexpr int M = 128 * MT, N = 128 * NT, COLS = M + N, FQ = COLS / 4;
    static constexpr int CR = 16, NO = 2;                // rows and row octets per chunk
    static constexpr int WGM = MT >= NT ? 4 : 2;         // wave grid WGM x WGN over the output, TM x TN tiles of 32 x 32 per wave
    static constexpr int WGN = 8 / WGM;
    static constexpr int TM = M / 32 / WGM, TN = N / 32 / WGN;
    static constexpr int UNITS = 2 * NO * FQ;            // (row octet, half, feature quad) load units per chunk: one per thread
    static constexpr int IMG = 3 * NO * COLS;            // uint4 per LDS image
    static constexpr int LDS_BYTES = 2 * IMG * 16;
};

// feature f's slot in the LDS image
__device__ __forceinline__ int sb_swz(int f) { return (f & ~3) | ((f & 3) ^ ((f >> 3) & 3)); }
// A may come in two column blocks (A2 != nullptr: columns M1 .. M - 1 from A2, row stride lda2): the GRU's dW_hh = [dr dz | dnr]^T h_prev
// takes (dr, dz) from dgi and dnr from its own tensor in ONE pass over h_prev.
template <int MT, int NT>
__global__ __launch_bounds__(512) void k_sb_wgrad(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb, int64_t K,
                                                  float *__restrict__ part, const float *__restrict__ A2, int64_t lda2, int M1) {
    using C = SbWgCfg<MT, NT>;
    extern __shared__ uint4 sb_lds[];                    // [buffer][piece][octet][feature slot] x 16 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;
    // loader role: thread -> (octet, half, feature quad)
    const bool on = tid < C::UNITS;
    const int rest = on ? tid / C::FQ : 0, col = 4 * (tid % C::FQ);
    const int row_in_chunk = 4 * rest;                   // = 8 octet + 4 half
    const bool in_a2 = A2 != nullptr && col >= M1 && col < C::M;
    const int64_t ld = in_a2 ? lda2 : (col < C::M ? lda : ldb);
    const float *src = (in_a2 ? A2 + (col - M1) : (col < C::M ? A + col : B + (col - C::M))) + row_in_chunk * ld;
    const int slot2 = 2 * ((rest >> 1) * C::COLS + col) + (rest & 1), sx = (col >> 3) & 3;   // in 8-byte units
    // multiplier role
    const int wm = wave / C::WGN, wn = wave % C::WGN;
    const int64_t chunks = K / C::CR;                    // full chunks; the K % 16 tail rows are the last workgroup's epilogue
    const int64_t c_beg = chunks * blockIdx.x / gridDim.x, c_end = chunks * (blockIdx.x + 1) / gridDim.x;

    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; a++)
#pragma unroll
        for (int b = 0; b < C::TN; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    float4 raw[SB_DEPTH][4];
    auto fetch = [&](float4 (&r)[4], int64_t c) {
        if (!on || c >= c_end) return;
        const float *p = src + c * C::CR * ld;
#pragma unroll
        for (int j = 0; j < 4; j++) r[j] = *(const float4 *)(p + j * ld);
    };
    // registers -> three bf16 pieces -> LDS image
    auto stage = [&](const float4 (&r)[4], uint4 *img) {
        if (!on) return;
        uint2 *img2 = (uint2 *)img;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            uint32_t p[3][2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const float x0 = t == 0 ? r[2 * q].x : t == 1 ? r[2 * q].y : t == 2 ? r[2 * q].z : r[2 * q].w;
                const float x1 = t == 0 ? r[2 * q + 1].x : t == 1 ? r[2 * q + 1].y : t == 2 ? r[2 * q + 1].z : r[2 * q + 1].w;
                sb_split2(x0, x1, p[0][q], p[1][q], p[2][q]);
            }
#pragma unroll
            for (int s = 0; s < 3; s++) img2[2 * s * C::NO * C::COLS + slot2 + 2 * (t ^ sx)] = make_uint2(p[s][0], p[s][1]);
        }
    };
    auto multiply = [&](const uint4 *buf) {
        const uint4 *img = buf + g * C::COLS;
        constexpr int PS = C::NO * C::COLS;              // piece stride
        if constexpr (C::TN <= C::TM) {                  // the narrower side's operands stay in registers across the other's tiles
            uint4 b[C::TN][3];
#pragma unroll
            for (int nt = 0; nt < C::TN; nt++) {
                const int f = sb_swz(C::M + 32 * (wn * C::TN + nt) + i);
#pragma unroll
                for (int s = 0; s < 3; s++) b[nt][s] = img[s * PS + f];
            }
#pragma unroll
            for (int mt = 0; mt < C::TM; mt++) {
                const int f = sb_swz(32 * (wm * C::TM + mt) + i);
                uint4 a[3];
#pragma unroll
                for (int s = 0; s < 3; s++) a[s] = img[s * PS + f];
#pragma unroll
                for (int nt = 0; nt < C::TN; nt++) acc[mt][nt] = sb_mma6_32(a, b[nt], acc[mt][nt]);
            }
        } else {
            uint4 a[C::TM][3];
#pragma unroll
            for (int mt = 0; mt < C::TM; mt++) {
                const int f = sb_swz(32 * (wm * C::TM + mt) + i);
#pragma unroll
                for (int s = 0; s < 3; s++) a[mt][s] = img[s * PS + f];
            }
#pragma unroll
            for (int nt = 0; nt < C::TN; nt++) {
                const int f = sb_swz(C::M + 32 * (wn * C::TN + nt) + i);
                uint4 b[3];
#pragma unroll
                for (int s = 0; s < 3; s++) b[s] = img[s * PS + f];
#pragma unroll
                for (int mt = 0; mt < C::TM; mt++) acc[mt][nt] = sb_mma6_32(a[mt], b, acc[mt][nt]);
            }
        }
    };
#pragma unroll
    for (int d = 0; d < SB_DEPTH; d++) fetch(raw[d], c_beg + d);
    if (c_beg < c_end) {
        stage(raw[0], sb_lds);
        fetch(raw[0], c_beg + SB_DEPTH);
    }
    lds_barrier();
    // invariant at the top of step c: image (c - c_beg) % 2 holds chunk c; raw[(c + k - c_beg) % DEPTH] holds chunk c + k, k = 1 .. DEPTH
    for (int64_t c = c_beg; c < c_end; c += SB_DEPTH) {
#pragma unroll
        for (int d = 0; d < SB_DEPTH; d++) {
            const int64_t cc = c + d;
            if (cc >= c_end) break;
            uint4 *cur = sb_lds + (d & 1) * C::IMG, *nxt = sb_lds + ((d + 1) & 1) * C::IMG;   // SB_DEPTH is even: parity of d = parity of cc - c_beg
            // the two waves of a SIMD (w and w + 4) run out of phase: one splits the next chunk while the other multiplies
            if (wave < 4) {
                if (cc + 1 < c_end) stage(raw[(d + 1) % SB_DEPTH], nxt);
                fetch(raw[(d + 1) % SB_DEPTH], cc + 1 + SB_DEPTH);
                multiply(cur);
            } else {
                multiply(cur);
                if (cc + 1 < c_end) stage(raw[(d + 1) % SB_DEPTH], nxt);
                fetch(raw[(d + 1) % SB_DEPTH], cc + 1 + SB_DEPTH);
            }
            lds_barrier();
        }
    }
    if (blockIdx.x == gridDim.x - 1 && (K % C::CR)) {     // the K % 16 tail rows, zero-filled
        float4 r[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            r[j] = (on && chunks * C::CR + row_in_chunk + j < K) ? *(const float4 *)(src + (chunks * C::CR + j) * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
        stage(r, sb_lds);
        lds_barrier();
        multiply(sb_lds);
    }
    // D tile (32 x 32): lane (i, g), register r -> row 8 (r / 4) + 4 g + r % 4 (the A operand's tile row: an M index), column i
    float *po = part + (size_t)blockIdx.x * C::M * C::N + (size_t)(32 * wm * C::TM + 4 * g) * C::N + 32 * wn * C::TN + i;
#pragma unroll
    for (int mt = 0; mt < C::TM; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int nt = 0; nt < C::TN; nt++) po[(size_t)(32 * mt + 8 * (r / 4) + (r % 4)) * C::N + 32 * nt] = acc[mt][nt][r];
}


// ---- neighbour mean of the fixed-depth recursive aggregation (DHGN.fcra, DHGN/mappo_parallel.py:204-233) ----------------------
// out[r][i][:] = act( sum_j abar_ij z[r][j][:] + bias ),  abar = adj / max(sum_j |adj|, 1e-12)  (F.normalize(adj, p=1, dim=-1))
// for the actor, abar = 1 / P for the critic (normalize(ones_like(adj))): the reference's torch.matmul(abar, hist) -- a batched
// P x P x E product per row, which the library runs as a strided-batched GEMM after the caller gathered the history slice into a
// contiguous copy.  Here: one pass, lane = feature, the P neighbour vectors of a row in registers, the P x P weights computed
// one per lane and read back with v_readlane; rows may be strided slices (n, t) of a (N, T + depth, P, E) history buffer.
// Both aggregates (actor from za, critic from zc) can come out of one launch (the rollout's paired tick).
struct NbrArgs {
    int R, P, E, T, relu;              // row r = (n, t) = (r / T, r % T)
    int64_t za_es, za_ts, zc_es, zc_ts;  // element strides of an episode / a step in za, zc
    int64_t adj_rs;                     // elements between adjacency rows
    int64_t out_ld;                     // elements between the output vectors of consecutive agents (E: dense)
};
// several hops in one launch (blockIdx.y): the FCRA hops of a rollout tick read stored history slots, not each other
struct NbrJobs { mo_nbr_job j[MO_NBR_MAX_JOBS]; };
template <int PT>
__global__ void k_nbr_mean(NbrArgs a, NbrJobs jobs, const float *__restrict__ adj) {
    const mo_nbr_job &job = jobs.j[blockIdx.y];
    const float *__restrict__ za = job.z_actor, *__restrict__ zc = job.z_critic, *__restrict__ bias = job.bias;
    float *__restrict__ out_a = job.out_actor, *__restrict__ out_c = job.out_critic;
    const int P = a.P, E = a.E, f = threadIdx.x, lane = threadIdx.x & 63;
    const float bf = bias ? bias[f] : 0.f;
    const float inv_p = 1.f / fmaxf((float)P, 1e-12f);
    for (int r = blockIdx.x; r < a.R; r += gridDim.x) {
        const int n = r / a.T, t = r - n * a.T;
        if (out_a) {
            const float *zr = za + (size_t)n * a.za_es + (size_t)t * a.za_ts;
            float z[PT];
#pragma unroll
            for (int j = 0; j < PT; j++) z[j] = j < P ? zr[(size_t)j * E + f] : 0.f;
            // weights: lane l = i P + j (P P <= 64) holds adj[i][j] / max(sum_j |adj[i][j]|, eps)
            float w = 0.f;
            if (P * P <= 64) {
                const float av = lane < P * P ? adj[(size_t)r * a.adj_rs + lane] : 0.f;
                const int li = lane / P < P ? lane / P : P - 1;
                float nrm = 0.f;
                for (int j = 0; j < P; j++) nrm += fabsf(__shfl(av, li * P + j));
                w = av / fmaxf(nrm, 1e-12f);
            }
#pragma unroll
            for (int i = 0; i < PT; i++)
                if (i < P) {
                    float acc = 0.f;
                    if (P * P <= 64) {
#pragma unroll
                        for (int j = 0; j < PT; j++)
                            if (j < P) acc = __builtin_fmaf(rl_f(w, i * P + j), z[j], acc);
                    } else {  // P > 8: weights through uniform addresses
                        const float *ar = adj + (size_t)r * a.adj_rs + (size_t)i * P;
                        float nrm = 0.f;
                        for (int j = 0; j < P; j++) nrm += fabsf(ar[j]);
                        nrm = fmaxf(nrm, 1e-12f);
#pragma unroll
                        for (int j = 0; j < PT; j++)
                            if (j < P) acc = __builtin_fmaf(ar[j] / nrm, z[j], acc);
                    }
                    acc += bf;
                    out_a[((size_t)r * P + i) * a.out_ld + f] = a.relu ? fmaxf(acc, 0.f) : acc;
                }
        }
        if (out_c) {
            const float *zr = zc + (size_t)n * a.zc_es + (size_t)t * a.zc_ts;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < PT; j++)
                if (j < P) acc = __builtin_fmaf(inv_p, zr[(size_t)j * E + f], acc);
            acc += bf;
            acc = a.relu ? fmaxf(acc, 0.f) : acc;
            for (int i = 0; i < P; i++) out_c[((size_t)r * P + i) * a.out_ld + f] = acc;
        }
    }
}

// E = 128, P <= 8: one WAVE per row (environment step), 16 bytes per lane -- lane (q, hf) owns the feature quad q of the output agents
// 4 hf .. 4 hf + 3; both halves read the P neighbour quads (the second read of an address is an L1 hit), the weights are wave-uniform
// (v_readlane) and selected per half.  Against the lane-per-feature form above (4-byte accesses, 3.1 / 3.7 TB/s actor / critic at the
// update's 61 500 rows): tools/nbr_probe.py.
__global__ __launch_bounds__(256) void k_nbr_mean4(NbrArgs a, NbrJobs jobs, const float *__restrict__ adj) {
    const mo_nbr_job &job = jobs.j[blockIdx.y];
    const float *__restrict__ za = job.z_actor, *__restrict__ zc = job.z_critic;
    float *__restrict__ out_a = job.out_actor, *__restrict__ out_c = job.out_critic;
    const int P = a.P, lane = threadIdx.x & 63, q = lane & 31, hf = lane >> 5;
    constexpr int E = 128;
    const float4 b4 = job.bias ? *(const float4 *)(job.bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv_p = 1.f / fmaxf((float)P, 1e-12f);
    const bool relu = a.relu != 0;
    auto put = [&](float *dst, float4 v) {
        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *(float4 *)dst = v;
    };
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < a.R; r += gridDim.x * 4) {
        const int n = r / a.T, t = r - n * a.T;
        if (out_a) {
            const float *zr = za + (size_t)n * a.za_es + (size_t)t * a.za_ts + 4 * q;
            float4 z[8];
#pragma unroll
            for (int j = 0; j < 8; j++) z[j] = j < P ? *(const float4 *)(zr + (size_t)j * E) : make_float4(0.f, 0.f, 0.f, 0.f);
            // lane l = i P + j holds adj[i][j] / max(sum_j |adj[i][j]|, eps)   (the same arithmetic as k_nbr_mean)
            const float av = lane < P * P ? adj[(size_t)r * a.adj_rs + lane] : 0.f;
            const int li = lane / P < P ? lane / P : P - 1;
            float nrm = 0.f;
            for (int j = 0; j < P; j++) nrm += fabsf(__shfl(av, li * P + j));
            const float w = av / fmaxf(nrm, 1e-12f);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (j < P) {
                        const float w_lo = rl_f(w, k * P + j), w_hi = 4 + k < P ? rl_f(w, (4 + k) * P + j) : 0.f;
                        const float wv = hf ? w_hi : w_lo;
                        acc.x = __builtin_fmaf(wv, z[j].x, acc.x); acc.y = __builtin_fmaf(wv, z[j].y, acc.y);
                        acc.z = __builtin_fmaf(wv, z[j].z, acc.z); acc.w = __builtin_fmaf(wv, z[j].w, acc.w);
                    }
                const int i = 4 * hf + k;
                if (i < P) put(out_a + ((size_t)r * P + i) * a.out_ld + 4 * q, acc);
            }
        }
        if (out_c) {
            const float *zr = zc + (size_t)n * a.zc_es + (size_t)t * a.zc_ts + 4 * q;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (j < P) {
                    const float4 v = *(const float4 *)(zr + (size_t)j * E);
                    acc.x = __builtin_fmaf(inv_p, v.x, acc.x); acc.y = __builtin_fmaf(inv_p, v.y, acc.y);
                    acc.z = __builtin_fmaf(inv_p, v.z, acc.z); acc.w = __builtin_fmaf(inv_p, v.w, acc.w);
                }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = 4 * hf + k;
                if (i < P) put(out_c + ((size_t)r * P + i) * a.out_ld + 4 * q, acc);
            }
        }
    }
}

// ---- ReLU backward + bias gradient in one pass ------------------------------------------------------------------------------
// gin = gout * [y > 0] (aten::threshold_backward on the saved output) and colsum[f] = sum_r gin[r][f] (the bias gradient of the
// Linear in front of the ReLU): autograd reads gin a second time for the sum (756 MB per mini-batch at DHGN's AGG layer);
// here it is added up while it is written.  F a multiple of 4, <= 1024; rows of F contiguous floats.
constexpr int RB_BLOCKS = 2048, RB_TPB = 256;
__global__ __launch_bounds__(RB_TPB) void k_relu_bwd_colsum(int64_t R, int F, const float *__restrict__ gout, int64_t ldg4,
                                                            const float *__restrict__ y, int64_t ldy4, float *__restrict__ gin,
                                                            float *__restrict__ part) {   // ldg4, ldy4: row strides of gout / y in float4
    __shared__ float4 s_sum[RB_TPB];
    const int f4 = F >> 2;                 // float4 columns per row
    const int cpb = RB_TPB / f4;           // rows a workgroup covers per step (host: f4 divides RB_TPB)
    const int col = threadIdx.x % f4, rr = threadIdx.x / f4;
    const int64_t rpb = (R + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * rpb, r1 = r0 + rpb < R ? r0 + rpb : R;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 *g4 = (const float4 *)gout, *y4 = (const float4 *)y;
    float4 *o4 = (float4 *)gin;
    int64_t r = r0 + rr;
    for (; r + 3 * cpb < r1; r += 4 * cpb) {  // four rows in flight per lane
        float4 g[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { g[u] = g4[(r + u * cpb) * ldg4 + col]; v[u] = y4[(r + u * cpb) * ldy4 + col]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float4 o;
            o.x = v[u].x > 0.f ? g[u].x : 0.f; o.y = v[u].y > 0.f ? g[u].y : 0.f; o.z = v[u].z > 0.f ? g[u].z : 0.f; o.w = v[u].w > 0.f ? g[u].w : 0.f;
            o4[(r + u * cpb) * f4 + col] = o;
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
    }
    for (; r < r1; r += cpb) {
        const float4 g = g4[r * ldg4 + col], v = y4[r * ldy4 + col];
        float4 o;
        o.x = v.x > 0.f ? g.x : 0.f; o.y = v.y > 0.f ? g.y : 0.f; o.z = v.z > 0.f ? g.z : 0.f; o.w = v.w > 0.f ? g.w : 0.f;
        o4[r * f4 + col] = o;
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    s_sum[threadIdx.x] = acc;
    __syncthreads();
    if (rr == 0) {
        for (int k = 1; k < cpb; k++) {
            const float4 t = s_sum[k * f4 + col];
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        ((float4 *)(part + (size_t)blockIdx.x * F))[col] = acc;
    }
}

// colsum[f] = sum over the partial blocks, f64, fixed order: 32 columns x 8 slices of the blocks per workgroup -- a wave instruction reads
// 128 contiguous bytes of two blocks (one wave per column walked the blocks with a 4 F-byte stride: 64 cache lines per instruction, 16 us
// behind every ReLU-backward pass)
__global__ __launch_bounds__(256) void k_colsum_reduce(int nblk, int F, const float *__restrict__ part, float *__restrict__ colsum) {
    __shared__ double red[8][32];
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5, f = blockIdx.x * 32 + c;
    double s = 0.0;
    if (f < F) {
#pragma unroll 4
        for (int b = sl; b < nblk; b += 8) s += (double)part[(size_t)b * F + f];
    }
    red[sl][c] = s;
    __syncthreads();
    if (sl == 0 && f < F) {
        double t = red[0][c];
#pragma unroll
        for (int k = 1; k < 8; k++) t += red[k][c];
        colsum[f] = (float)t;
    }
}

// ---- skinny weight gradient: C [NS][F] = S^T X over R rows, S [R][NS] with NS <= 16 -----------------------------------------
// The Linear layers with a handful of inputs or outputs (the K = 4 position part of DHGN's semantic layer, the action and value
// heads): the library runs these 128 x {1, 4, 9} x 492 000 reductions at 0.2-0.3 ms; they are one streaming pass over X
// (252 MB): lane = column of X, the row of S comes through uniform addresses, NS multiply-adds per element.  Optional column
// sums of X and of S (the bias gradients) from the same pass.  Per-workgroup partials, then k_skinny_reduce (fixed order).
constexpr int SK_MAX = 16, SK_BLOCKS = 2048, SK_U = 8;  // rows in flight per lane
template <int NS>
__global__ void k_wgrad_skinny(int64_t R, int F, const float *__restrict__ S, int64_t lds_, const float *__restrict__ X, int64_t ldx,
                               float *__restrict__ part) {
    const int f = threadIdx.x;
    const int64_t rpb = (R + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * rpb, r1 = r0 + rpb < R ? r0 + rpb : R;
    float acc[NS], cs[NS], cx = 0.f;
#pragma unroll
    for (int k = 0; k < NS; k++) { acc[k] = 0.f; cs[k] = 0.f; }
    int64_t r = r0;
    for (; r + SK_U <= r1; r += SK_U) {
        float xv[SK_U];
#pragma unroll
        for (int u = 0; u < SK_U; u++) xv[u] = X[(r + u) * ldx + f];
#pragma unroll
        for (int u = 0; u < SK_U; u++) {
            const float *__restrict__ sr = S + (r + u) * lds_;
            cx += xv[u];
#pragma unroll
            for (int k = 0; k < NS; k++) {
                const float sk = sr[k];
                acc[k] = __builtin_fmaf(sk, xv[u], acc[k]);
                cs[k] += sk;
            }
        }
    }
    for (; r < r1; r++) {
        const float x0 = X[r * ldx + f];
        const float *__restrict__ sr = S + r * lds_;
        cx += x0;
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const float sk = sr[k];
            acc[k] = __builtin_fmaf(sk, x0, acc[k]);
            cs[k] += sk;
        }
    }
    // partials [block][NS + 1][F] then [block][NS] (column sums of S, identical in every lane)
    float *dst = part + (size_t)blockIdx.x * ((NS + 1) * F + NS);
#pragma unroll
    for (int k = 0; k < NS; k++) dst[k * F + f] = acc[k];
    dst[NS * F + f] = cx;
    if (f == 0) {
#pragma unroll
        for (int k = 0; k < NS; k++) dst[(NS + 1) * F + k] = cs[k];
    }
}

// one wave per output element: C[k][f] (or C[f][k] when transposed), colsum_x[f], colsum_s[k]; f64 accumulation, fixed order
__global__ void k_skinny_reduce(int nblk, int NS, int F, const float *__restrict__ part, int transposed, float *__restrict__ C,
                                float *__restrict__ colsum_x, float *__restrict__ colsum_s) {
    const int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, per = (NS + 1) * F + NS;
    if (idx >= per) return;
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += (double)part[(size_t)b * per + idx];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane != 0) return;
    if (idx < NS * F) {
        const int k = idx / F, f = idx - k * F;
        C[transposed ? (size_t)f * NS + k : (size_t)k * F + f] = (float)s;
    } else if (idx < (NS + 1) * F) {
        if (colsum_x) colsum_x[idx - NS * F] = (float)s;
    } else if (colsum_s) {
        colsum_s[idx - (NS + 1) * F] = (float)s;
    }
}

int wgrad_split(int M, int N, int *am, int *bn) {  // tile shape and number of K-splits: one workgroup per CU
    *am = (M == 384 && N == 128) ? 3 : 1;
    *bn = (M == 128 && N == 384) ? 3 : 1;
    const int tiles = (M / (128 * *am)) * (N / (128 * *bn));
    int S = 256 / tiles;
    return S < 1 ? 1 : S;
}

int launch_msgw3(const mo_msg_rel *rel, int R, int P, int E, const float *p, int64_t p_rs, const int32_t *kvalid2, float *out, float *out_c,
                 int c_valid, int64_t out_stride, void *stream, bool pair, const float *Wp = nullptr, int64_t wp_rs = 0, const float *bp = nullptr,
                 float *pos_a = nullptr, float *pos_c = nullptr, int64_t pos_ld = 0, int c_rel2 = 1) {
    MsgDims d[3];
    int grid = 1;
    for (int r = 0; r < 3; r++) {
        const mo_msg_rel &m = rel[r];
        int rc = check_msg(R, P, m.K, E, m.din, m.q_div, m.adj_mode, m.adj, m.kvalid, m.e);
        if (rc) return rc;
        if ((m.q_rs & 3) || out_stride < 3 * E) return MO_ERR_BAD_ARG;
        if (r < 2 && m.adj_mode == MO_ADJ_VALID) return MO_ERR_BAD_ARG;  // only the obstacle relation carries kvalid
        d[r] = msg_dims(R, P, m.K, E, m.din, m.q_div, m.adj_mode, p_rs, m.q_rs, m.e_rs, m.adj_rs, out_stride, FWD_BLOCKS, &grid);
    }
    const bool s01 = msg_q_small(rel[0].K) && msg_q_small(rel[1].K) && msg_adj_small(P, rel[0].K, rel[0].adj_mode) &&
                     msg_adj_small(P, rel[1].K, rel[1].adj_mode);
    const bool as2 = msg_adj_small(P, rel[2].K, rel[2].adj_mode), ev2 = (E % 128) == 0;
    // Few rows (the rollout's tick: one row per environment): a row's agents are divided between two waves (PT = 4 of 8, or 8 of 16) --
    // twice the waves, half the dependent work per wave; the neighbour offsets d_j are then computed by both (+14 % arithmetic), which
    // is why launches with more rows keep one wave per row (measured at 4 096 rows: 77 us whole, 83 us halved; at 512 rows: see DESIGN 6).
    const bool halves = R <= 2048 && P > 4;
    const int pt = halves ? (P <= 8 ? 4 : 8) : (P <= 8 ? 8 : 16), gy = (P + pt - 1) / pt;
#define MSGW3_LAUNCH1(PT, S01, AS2, EV)                                                                                                     \
    hipLaunchKernelGGL((k_msgw3_fwd<PT, S01, AS2, EV>), dim3(grid, gy), dim3(E / EV), 0, (hipStream_t)stream, d[0], d[1], d[2], p, rel[0].q,   \
                       rel[0].e, rel[0].adj, rel[0].W, rel[0].b, rel[1].q, rel[1].adj, rel[1].W, rel[1].b, rel[2].q, rel[2].adj, kvalid2,      \
                       rel[2].W, rel[2].b, out, out_c, c_valid, c_rel2, Wp, wp_rs, bp, pos_a, pos_c, pos_ld ? pos_ld : (int64_t)E)
#define MSGW3_LAUNCH(PT, S01, AS2) do { if (ev2) MSGW3_LAUNCH1(PT, S01, AS2, 2); else MSGW3_LAUNCH1(PT, S01, AS2, 1); } while (0)
#define MSGW3_PT(PT) { if (s01 && as2) MSGW3_LAUNCH(PT, true, true); else if (s01) MSGW3_LAUNCH(PT, true, false); else if (as2) MSGW3_LAUNCH(PT, false, true); else MSGW3_LAUNCH(PT, false, false); }
    if (pt == 4) MSGW3_PT(4) else if (pt == 8) MSGW3_PT(8) else MSGW3_PT(16)
#undef MSGW3_PT
#undef MSGW3_LAUNCH
#undef MSGW3_LAUNCH1
    return (int)hipGetLastError();
}

}  // namespace

constexpr int SB_WGRAD_WGS = 256;  // one workgroup per CU (96 KB of LDS each)

static bool sb_wgrad_shape_ok(int M, int N) {
    return (M == 128 || M == 256 || M == 384) && (N == 128 || N == 256 || N == 384) && M + N <= 512 && M * N < 256 * 256;
}

template <int MT, int NT>
int launch_sb_wgrad(int64_t K, const float *A, int64_t lda, const float *B, int64_t ldb, float *Cm, int accumulate, float *part, hipStream_t st,
                    const float *A2 = nullptr, int64_t lda2 = 0, int M1 = 0) {
    using C = SbWgCfg<MT, NT>;
    static_assert(C::UNITS <= 512 && C::WGM * C::WGN == 8 && C::TM * C::WGM * 32 == C::M && C::TN * C::WGN * 32 == C::N && SB_DEPTH % 2 == 0, "tiling");
    static std::once_flag once;
    static hipError_t attr_rc = hipSuccess;
    std::call_once(once, [] { attr_rc = hipFuncSetAttribute((const void *)k_sb_wgrad<MT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES); });
    if (attr_rc != hipSuccess) return (int)attr_rc;
    hipLaunchKernelGGL((k_sb_wgrad<MT, NT>), dim3(SB_WGRAD_WGS), dim3(512), C::LDS_BYTES, st, A, lda, B, ldb, K, part, A2, lda2, M1);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((C::M * C::N + 63) / 64), dim3(256), 0, st, SB_WGRAD_WGS, C::M * C::N, (const float *)part, Cm, accumulate);
    return (int)hipGetLastError();
}

extern "C" {

int dhgn_msg_agg_fwd(int32_t R, int32_t P, int32_t K, int32_t E, int32_t din, const float *p, int64_t p_rs, const float *q,
                     int64_t q_rs, int32_t q_div, const float *e, int64_t e_rs, const void *adj, int64_t adj_rs, int32_t adj_mode,
                     const int32_t *kvalid, const float *W, const float *b, float *out, int64_t out_stride, void *stream) {
    int rc = check_msg(R, P, K, E, din, q_div, adj_mode, adj, kvalid, e);
    if (rc) return rc;
    if (R == 0) return 0;
    if ((q_rs & 3) || out_stride < E) return MO_ERR_BAD_ARG;
    int grid;
    const MsgDims d = msg_dims(R, P, K, E, din, q_div, adj_mode, p_rs, q_rs, e_rs, adj_rs, out_stride, FWD_BLOCKS, &grid);
    const bool qs = msg_q_small(K), as = msg_adj_small(P, K, adj_mode), ev2 = (E % 128) == 0;
#define MSGW_FWD1(PT, QS, AS, EV) hipLaunchKernelGGL((k_msgw_fwd<PT, QS, AS, EV>), dim3(grid), dim3(E / EV), 0, (hipStream_t)stream, d, p, q, e, adj, kvalid, W, b, out)
#define MSGW_FWD(PT, QS, AS) do { if (ev2) MSGW_FWD1(PT, QS, AS, 2); else MSGW_FWD1(PT, QS, AS, 1); } while (0)
#define MSGW_FWD_PT(PT) { if (qs && as) MSGW_FWD(PT, true, true); else if (qs) MSGW_FWD(PT, true, false); else if (as) MSGW_FWD(PT, false, true); else MSGW_FWD(PT, false, false); }
    if (P <= 8) MSGW_FWD_PT(8) else MSGW_FWD_PT(16)
#undef MSGW_FWD_PT
#undef MSGW_FWD
#undef MSGW_FWD1
    return (int)hipGetLastError();
}

int dhgn_msg_agg3_fwd(const mo_msg_rel *rel, int32_t R, int32_t P, int32_t E, const float *p, int64_t p_rs, float *out, int64_t out_stride,
                      void *stream) {
    if (!rel || !p || !out || R < 0) return MO_ERR_BAD_ARG;
    if (R == 0) return 0;
    return launch_msgw3(rel, R, P, E, p, p_rs, rel[2].kvalid, out, nullptr, 0, out_stride, stream, false);
}

int dhgn_msg_agg3_pair_fwd(const mo_msg_rel *rel, int32_t R, int32_t P, int32_t E, const float *p, int64_t p_rs, const int32_t *o_kvalid,
                           float *out_actor, float *out_critic, int64_t out_stride, void *stream) {
    if (!rel || !p || !out_actor || !out_critic || R < 0) return MO_ERR_BAD_ARG;
    if (R == 0) return 0;
    for (int r = 0; r < 3; r++)
        if (rel[r].adj_mode != MO_ADJ_TENSOR && rel[r].adj_mode != MO_ADJ_BITS) return MO_ERR_BAD_ARG;  // the actor's adjacency
    return launch_msgw3(rel, R, P, E, p, p_rs, o_kvalid, out_actor, out_critic, o_kvalid != nullptr, out_stride, stream, true);
}

int dhgn_msg_agg3_pair01_fwd(const mo_msg_rel *rel, int32_t R, int32_t P, int32_t E, const float *p, int64_t p_rs, float *out_actor,
                             float *out_critic, int64_t out_stride, void *stream) {
    if (!rel || !p || !out_actor || !out_critic || R < 0) return MO_ERR_BAD_ARG;
    if (R == 0) return 0;
    for (int r = 0; r < 3; r++)
        if (rel[r].adj_mode != MO_ADJ_TENSOR && rel[r].adj_mode != MO_ADJ_BITS) return MO_ERR_BAD_ARG;  // the actor's adjacency
    return launch_msgw3(rel, R, P, E, p, p_rs, nullptr, out_actor, out_critic, 0, out_stride, stream, true, nullptr, 0, nullptr, nullptr, nullptr, 0, 0);
}

int dhgn_msg_agg3_pair_pos_fwd(const mo_msg_rel *rel, int32_t R, int32_t P, int32_t E, const float *p, int64_t p_rs, const int32_t *o_kvalid,
                               float *out_actor, float *out_critic, int64_t out_stride, const float *Wp, int64_t wp_row_stride, const float *bp,
                               float *pos_actor, float *pos_critic, int64_t pos_stride, void *stream) {
    if (!rel || !p || !out_actor || !out_critic || R < 0 || !Wp || !bp || !pos_actor || wp_row_stride < 4 || pos_stride < E) return MO_ERR_BAD_ARG;
    if (R == 0) return 0;
    for (int r = 0; r < 3; r++)
        if (rel[r].adj_mode != MO_ADJ_TENSOR && rel[r].adj_mode != MO_ADJ_BITS) return MO_ERR_BAD_ARG;
    return launch_msgw3(rel, R, P, E, p, p_rs, o_kvalid, out_actor, out_critic, o_kvalid != nullptr, out_stride, stream, true, Wp, wp_row_stride, bp,
                        pos_actor, pos_critic, pos_stride);
}

int64_t dhgn_msg_agg_bwd_workspace(int32_t E, int32_t din) { return (int64_t)BWD_BLOCKS * (din + 1) * E * sizeof(float); }

static int msg_agg_bwd_launch(int32_t R, int32_t P, int32_t K, int32_t E, int32_t din, const float *p, int64_t p_rs, const float *q,
                              int64_t q_rs, int32_t q_div, const float *e, int64_t e_rs, const void *adj, int64_t adj_rs, int32_t adj_mode,
                              const int32_t *kvalid, const float *W, const float *b, const float *gout, const float *gout_c, int64_t gout_stride,
                              float *dW, float *db, void *workspace, void *stream) {
    int rc = check_msg(R, P, K, E, din, q_div, adj_mode, adj, kvalid, e);
    if (rc) return rc;
    if (!workspace || !gout || !dW || !db) return MO_ERR_BAD_ARG;
    if ((q_rs & 3) || gout_stride < E) return MO_ERR_BAD_ARG;
    if (gout_c && adj_mode != MO_ADJ_TENSOR) return MO_ERR_BAD_ARG;
    int grid;
    const MsgDims d = msg_dims(R, P, K, E, din, q_div, adj_mode, p_rs, q_rs, e_rs, adj_rs, gout_stride, BWD_BLOCKS, &grid);
    const bool qs = msg_q_small(K), as = msg_adj_small(P, K, adj_mode), ev2 = (E % 128) == 0;
#define MSGW_BWD2(PT, QS, AS, EV, PAIR) \
    hipLaunchKernelGGL((k_msgw_bwd<PT, QS, AS, EV, PAIR>), dim3(grid), dim3(E / EV), 0, (hipStream_t)stream, d, p, q, e, adj, kvalid, W, b, gout, gout_c, (float *)workspace)
#define MSGW_BWD1(PT, QS, AS, EV) do { if (gout_c) MSGW_BWD2(PT, QS, AS, EV, true); else MSGW_BWD2(PT, QS, AS, EV, false); } while (0)
#define MSGW_BWD(PT, QS, AS) do { if (ev2) MSGW_BWD1(PT, QS, AS, 2); else MSGW_BWD1(PT, QS, AS, 1); } while (0)
#define MSGW_BWD_PT(PT) { if (qs && as) MSGW_BWD(PT, true, true); else if (qs) MSGW_BWD(PT, true, false); else if (as) MSGW_BWD(PT, false, true); else MSGW_BWD(PT, false, false); }
    if (P <= 8) MSGW_BWD_PT(8) else MSGW_BWD_PT(16)
#undef MSGW_BWD_PT
#undef MSGW_BWD
#undef MSGW_BWD1
#undef MSGW_BWD2
    const int tot = (din + 1) * E;
    hipLaunchKernelGGL(k_msg_agg_bwd_reduce, dim3((tot + 3) / 4), dim3(256), 0, (hipStream_t)stream, grid, E, din,
                       (const float *)workspace, dW, db);
    return (int)hipGetLastError();
}

int dhgn_msg_agg_bwd(int32_t R, int32_t P, int32_t K, int32_t E, int32_t din, const float *p, int64_t p_rs, const float *q,
                     int64_t q_rs, int32_t q_div, const float *e, int64_t e_rs, const void *adj, int64_t adj_rs, int32_t adj_mode,
                     const int32_t *kvalid, const float *W, const float *b, const float *gout, int64_t gout_stride, float *dW, float *db,
                     void *workspace, void *stream) {
    return msg_agg_bwd_launch(R, P, K, E, din, p, p_rs, q, q_rs, q_div, e, e_rs, adj, adj_rs, adj_mode, kvalid, W, b, gout, nullptr, gout_stride, dW, db,
                              workspace, stream);
}

int dhgn_msg_agg_bwd_pair(int32_t R, int32_t P, int32_t K, int32_t E, int32_t din, const float *p, int64_t p_rs, const float *q,
                          int64_t q_rs, int32_t q_div, const float *e, int64_t e_rs, const float *adj, int64_t adj_rs, const float *W,
                          const float *b, const float *gout_actor, const float *gout_critic, int64_t gout_stride, float *dW, float *db,
                          void *workspace, void *stream) {
    if (!gout_critic) return MO_ERR_BAD_ARG;
    return msg_agg_bwd_launch(R, P, K, E, din, p, p_rs, q, q_rs, q_div, e, e_rs, adj, adj_rs, MO_ADJ_TENSOR, nullptr, W, b, gout_actor, gout_critic,
                              gout_stride, dW, db, workspace, stream);
}

int dhgn_msg_agg_ones_sorted_ok(int32_t R, int32_t P, int32_t K, int32_t E, int32_t din, int32_t q_div) {
    return R > 0 && P >= 1 && K >= 2 && K <= SO_MAXK && E >= SO_FC && (E % SO_FC) == 0 && din == 4 && q_div >= 1 && (R % q_div) == 0;
}

int64_t dhgn_msg_agg_ones_sorted_workspace(int32_t R, int32_t P, int32_t K, int32_t E, int32_t q_div, int64_t *m_bytes, int64_t *qtab_bytes,
                                           int64_t *partial_bytes) {
    const int64_t mb = (int64_t)R * P * E, qb = (int64_t)(R / q_div) * E * 4 * (K + 1) * sizeof(float), pb = (int64_t)(R / q_div) * 5 * E * sizeof(float);
    if (m_bytes) *m_bytes = mb;
    if (qtab_bytes) *qtab_bytes = qb;
    if (partial_bytes) *partial_bytes = pb;
    return mb + qb + pb;
}

int dhgn_msg_agg_ones_sorted_fwd(int32_t R, int32_t P, int32_t K, int32_t E, const float *p, int64_t p_rs, const float *q, int64_t q_rs,
                                 int32_t q_div, const float *W, const float *b, float *out, int64_t out_stride, uint8_t *save_m, float *qtab,
                                 void *stream) {
    if (!dhgn_msg_agg_ones_sorted_ok(R, P, K, E, 4, q_div) || !p || !q || !W || !b || !out) return MO_ERR_BAD_ARG;
    if ((q_rs & 3) || (p_rs & 3) || out_stride < E || ((uintptr_t)p & 15) || ((uintptr_t)q & 15)) return MO_ERR_BAD_ARG;
    SortedArgs a{R, P, K, E, q_div, p, q, W, b, p_rs, q_rs, out_stride};
    if (so_fwd_lds(K) > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_msg_ones_sorted_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)so_fwd_lds(K));
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_msg_ones_sorted_fwd, dim3(R / q_div, E / SO_FC), dim3(SO_TPB), so_fwd_lds(K), (hipStream_t)stream, a, out, save_m, qtab);
    return (int)hipGetLastError();
}

int dhgn_msg_agg_ones_sorted_bwd(int32_t R, int32_t P, int32_t K, int32_t E, const float *p, int64_t p_rs, int32_t q_div, const float *gout,
                                 int64_t gout_stride, const uint8_t *save_m, const float *qtab, float *dW, float *db, void *partials,
                                 void *stream) {
    if (!dhgn_msg_agg_ones_sorted_ok(R, P, K, E, 4, q_div) || !p || !gout || !save_m || !qtab || !dW || !db || !partials) return MO_ERR_BAD_ARG;
    if ((p_rs & 3) || gout_stride < E || ((uintptr_t)p & 15)) return MO_ERR_BAD_ARG;
    SortedArgs a{R, P, K, E, q_div, p, nullptr, nullptr, nullptr, p_rs, 0, gout_stride};
    const int nblk = R / q_div;
    if (so_bwd_lds(K) > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_msg_ones_sorted_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)so_bwd_lds(K));
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k_msg_ones_sorted_bwd, dim3(nblk, E / SO_FC), dim3(SO_TPB), so_bwd_lds(K), (hipStream_t)stream, a, gout, save_m, qtab, (float *)partials);
    hipLaunchKernelGGL(k_msg_agg_bwd_reduce, dim3((5 * E + 3) / 4), dim3(256), 0, (hipStream_t)stream, nblk, (int)E, 4, (const float *)partials, dW, db);
    return (int)hipGetLastError();
}

int64_t gae_advnorm_workspace(void) { return (int64_t)(4 + 2 * GAE_BLOCKS) * sizeof(double); }

int gae_advnorm(int32_t N, int32_t T, int32_t P, const float *r, const float *v, const float *active, float gamma, float lamda,
                int32_t use_adv_norm, float *adv, float *v_target, double *stats, void *stream) {
    if (N < 1 || T < 1 || P < 1 || !r || !v || !active || !adv || !v_target || !stats) return MO_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)N * T * P;
    const int64_t seq_blocks = ((int64_t)N * P + 255) / 256, el_blocks = (n + 255) / 256;
    const int g_scan = (int)(seq_blocks < GAE_BLOCKS ? seq_blocks : GAE_BLOCKS), g_el = (int)(el_blocks < GAE_BLOCKS ? el_blocks : GAE_BLOCKS);
    hipLaunchKernelGGL(k_gae_scan, dim3(g_scan), dim3(256), 0, s, N, T, P, r, v, active, gamma, lamda, adv, v_target, stats);
    hipLaunchKernelGGL(k_gae_finalize, dim3(1), dim3(1), 0, s, n, g_scan, stats);
    if (use_adv_norm) {
        hipLaunchKernelGGL(k_gae_center, dim3(g_el), dim3(256), 0, s, n, adv, stats);
        hipLaunchKernelGGL(k_gae_std, dim3(1), dim3(1), 0, s, n, g_el, stats);
        hipLaunchKernelGGL(k_gae_norm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, adv, active, stats);
    }
    return (int)hipGetLastError();
}

int categorical_sample(int32_t R, int32_t A, const float *probs, uint64_t seed, uint64_t offset, int32_t greedy, int32_t *action,
                       float *logp, void *stream) {
    if (R < 0 || A < 1 || !probs || !action) return MO_ERR_BAD_ARG;
    if (R == 0) return 0;
    hipLaunchKernelGGL(k_categorical, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, R, A, probs, seed, offset,
                       (const uint64_t *)nullptr, greedy, action, logp);
    return (int)hipGetLastError();
}

int categorical_sample_counter(int32_t R, int32_t A, const float *probs, uint64_t seed, uint64_t *counter, int32_t greedy, int32_t *action,
                               float *logp, void *stream) {
    if (R < 0 || A < 1 || !probs || !action || !counter) return MO_ERR_BAD_ARG;
    if (R == 0) return 0;
    hipLaunchKernelGGL(k_categorical, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, R, A, probs, seed, (uint64_t)0, counter, greedy,
                       action, logp);
    hipLaunchKernelGGL(k_advance_counter, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, (uint64_t)R);
    return (int)hipGetLastError();
}

int spectral_norm_weight(int32_t A, int32_t H, const float *W, float *u, float *v, float eps, int32_t n_power_iterations, float *w_eff,
                         void *stream) {
    if (A < 1 || A > SN_MAX_A || H < 1 || H > SN_MAX_H || !W || !u || !v || !w_eff || n_power_iterations < 0) return MO_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_sn_power, dim3(1), dim3(256), 0, (hipStream_t)stream, A, H, W, u, v, eps, n_power_iterations, w_eff);
    return (int)hipGetLastError();
}

int head_grid(int R) { const int g = (R + 255) / 256; return g < 1024 ? g : 1024; }  // 4 waves x 64 rows per workgroup step

int head_linear(int32_t R, int32_t A, int32_t H, const float *feat, const float *W, const float *b, float *y, void *stream) {
    if (R < 0 || A < 1 || A > HEAD_MAX_A || H != HEAD_H || !feat || !W || !b || !y || ((uintptr_t)feat & 15)) return MO_ERR_BAD_ARG;
    if (R == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
#define HEAD_LIN(AT) case AT: hipLaunchKernelGGL((k_head<false, AT>), dim3(head_grid(R)), dim3(256), 0, st, R, A, feat, W, b, y, (uint64_t)0, \
                                       (uint64_t *)nullptr, (unsigned int *)nullptr, 0, (int32_t *)nullptr, (float *)nullptr); break;
    switch (A) {
        HEAD_LIN(1) HEAD_LIN(2) HEAD_LIN(3) HEAD_LIN(4) HEAD_LIN(5) HEAD_LIN(6) HEAD_LIN(7) HEAD_LIN(8) HEAD_LIN(9) HEAD_LIN(10) HEAD_LIN(11)
        HEAD_LIN(12) HEAD_LIN(13) HEAD_LIN(14) HEAD_LIN(15) HEAD_LIN(16)
    }
#undef HEAD_LIN
    return (int)hipGetLastError();
}

int head_sample(int32_t R, int32_t A, int32_t H, const float *feat, const float *W, const float *b, uint64_t seed, uint64_t *counter,
                uint32_t *ticket, int32_t greedy, int32_t *action, float *logp, void *stream) {
    if (R < 0 || A < 1 || A > HEAD_MAX_A || H != HEAD_H || !feat || !W || !b || !counter || !ticket || !action || !logp || ((uintptr_t)feat & 15))
        return MO_ERR_BAD_ARG;
    if (R == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
#define HEAD_SMP(AT) case AT: hipLaunchKernelGGL((k_head<true, AT>), dim3(head_grid(R)), dim3(256), 0, st, R, A, feat, W, b, (float *)nullptr, seed, \
                                       counter, ticket, (int)greedy, action, logp); break;
    switch (A) {
        HEAD_SMP(1) HEAD_SMP(2) HEAD_SMP(3) HEAD_SMP(4) HEAD_SMP(5) HEAD_SMP(6) HEAD_SMP(7) HEAD_SMP(8) HEAD_SMP(9) HEAD_SMP(10) HEAD_SMP(11)
        HEAD_SMP(12) HEAD_SMP(13) HEAD_SMP(14) HEAD_SMP(15) HEAD_SMP(16)
    }
#undef HEAD_SMP
    return (int)hipGetLastError();
}

int gru_gates_fwd(int32_t B, int32_t H, const float *gi, const float *gh, const float *b_hh, const float *h_prev, float *h_out,
                  float *save, void *stream) {
    if (B < 1 || H < 4 || (H & 3) || !gi || !gh || !b_hh || !h_prev || !h_out) return MO_ERR_BAD_ARG;
    const int n = B * (H >> 2);
    hipLaunchKernelGGL(k_gru_gates_fwd, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, H, gi, gh, b_hh, h_prev, h_out, save);
    return (int)hipGetLastError();
}

int gru_gates_bwd(int32_t B, int32_t H, const float *dout, const float *dcarry, const float *save, const float *h_prev, float *dgi,
                  float *dgh, float *dh_direct, void *stream) {
    if (B < 1 || H < 4 || (H & 3) || !save || !h_prev || !dgi || !dgh || !dh_direct) return MO_ERR_BAD_ARG;
    const int n = B * (H >> 2);
    hipLaunchKernelGGL(k_gru_gates_bwd, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, H, dout, dcarry, save, h_prev, dgi, dgh,
                       dh_direct);
    return (int)hipGetLastError();
}

int gru_cell_fwd_multi(int32_t n_nets, const mo_gru_cell_net *nets, int32_t B, int32_t H, void *stream) {
    if (n_nets < 1 || n_nets > MO_GRU_CELL_MAX_NETS || !nets || B < 1 || H != GRU_H) return MO_ERR_BAD_ARG;
    GruCellNets a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < n_nets; k++) {
        const mo_gru_cell_net &m = nets[k];
        if (!m.x || !m.h_prev || !m.w_ih || !m.w_hh || !m.b_ih || !m.b_hh || !m.h_out) return MO_ERR_BAD_ARG;
        if (((uintptr_t)m.x & 15) || ((uintptr_t)m.h_prev & 15) || ((uintptr_t)m.w_ih & 15) || ((uintptr_t)m.w_hh & 15)) return MO_ERR_BAD_ARG;
        a.n[k] = m;
    }
    const int nblk = (B + GRU_RB - 1) / GRU_RB;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    int per = cus / n_nets;     // persistent workgroups per cell: one per CU in total
    if (per < 1) per = 1;
    const int grid = nblk < per ? nblk : per;
    hipLaunchKernelGGL(k_gru_cell, dim3(grid, n_nets), dim3(512), 0, (hipStream_t)stream, (int)B, nblk, a);
    return (int)hipGetLastError();
}

int gru_cell_split_fwd_multi(int32_t n_nets, const mo_gru_cell_net *nets, int32_t B, int32_t H, void *stream) {
    if (n_nets < 1 || n_nets > MO_GRU_CELL_MAX_NETS || !nets || B < 1 || H != GRU_H) return MO_ERR_BAD_ARG;
    GruCellNets a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < n_nets; k++) {
        const mo_gru_cell_net &m = nets[k];
        if (!m.x || !m.h_prev || !m.w_ih || !m.w_hh || !m.b_ih || !m.b_hh || !m.h_out) return MO_ERR_BAD_ARG;
        if (((uintptr_t)m.x & 15) || ((uintptr_t)m.h_prev & 15) || ((uintptr_t)m.w_ih & 15) || ((uintptr_t)m.w_hh & 15) || ((uintptr_t)m.h_out & 15))
            return MO_ERR_BAD_ARG;
        if (m.h_out == m.h_prev || m.h_out == m.x) return MO_ERR_BAD_ARG;   // two workgroups share a row tile: not in place
        a.n[k] = m;
    }
    const int nblk = (B + GRU_RB - 1) / GRU_RB;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    int pairs = cus / n_nets / 2;   // persistent workgroup PAIRS per cell (the two halves of the hidden units): one workgroup per CU in total
    if (pairs < 1) pairs = 1;
    if (pairs > nblk) pairs = nblk;
    hipLaunchKernelGGL(k_gru_cell_sb, dim3(2 * pairs, n_nets), dim3(512), 0, (hipStream_t)stream, (int)B, nblk, a);
    return (int)hipGetLastError();
}

int sb_split_diag(int64_t n, const float *x, float *pieces, void *stream) {
    if (n < 1 || !x || !pieces) return MO_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_sb_split_diag, dim3((unsigned)((n / 2 + 256) / 256)), dim3(256), 0, (hipStream_t)stream, n, x, pieces);
    return (int)hipGetLastError();
}

int gru_cell_fwd(int32_t B, int32_t H, const float *x, const float *h_prev, const float *w_ih, const float *w_hh, const float *b_ih,
                 const float *b_hh, float *h_out, void *stream) {
    const mo_gru_cell_net net{x, h_prev, w_ih, w_hh, b_ih, b_hh, h_out};
    return gru_cell_fwd_multi(1, &net, B, H, stream);
}

int64_t gru_seq_save_elems(int32_t T, int32_t B) { return (int64_t)T * ((B + GRU_RB - 1) / GRU_RB) * 4 * 512 * 4; }

int gru_seq_fwd_multi(int32_t n_nets, const mo_gru_seq_net *nets, int32_t T, int32_t B, int32_t H, int32_t gi_agents, void *stream) {
    if (n_nets < 1 || n_nets > MO_GRU_MAX_NETS || !nets || T < 1 || B < 1 || H != GRU_H || gi_agents < 0 || (gi_agents && B % gi_agents)) return MO_ERR_BAD_ARG;
    GruFwdNets a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < n_nets; k++) {
        const mo_gru_seq_net &m = nets[k];
        if (!m.gi || !m.w_hh || !m.b_hh || !m.h0 || !m.out) return MO_ERR_BAD_ARG;
        if ((((uintptr_t)m.gi | (uintptr_t)m.w_hh | (uintptr_t)m.b_hh | (uintptr_t)m.h0 | (uintptr_t)m.out | (uintptr_t)m.save) & 15)) return MO_ERR_BAD_ARG;
        if (m.B < 0 || m.B > B || (gi_agents && m.B % gi_agents)) return MO_ERR_BAD_ARG;
        a.n[k] = m;
    }
    const int nblk = (B + GRU_RB - 1) / GRU_RB;
    hipLaunchKernelGGL(k_gru_seq_fwd2, dim3(nblk, n_nets), dim3(512), 0, (hipStream_t)stream, T, B, a, (int)gi_agents);
    return (int)hipGetLastError();
}

int gru_seq_fwd(int32_t T, int32_t B, int32_t H, const float *gi, const float *w_hh, const float *b_hh, const float *h0, float *out,
                float *save, int32_t gi_agents, void *stream) {
    const mo_gru_seq_net net{gi, w_hh, b_hh, h0, out, save, 0, 0};
    return gru_seq_fwd_multi(1, &net, T, B, H, gi_agents, stream);
}

int64_t gru_seq_bwd_workspace(int32_t B) { return (int64_t)((B + GRU_RB - 1) / GRU_RB) * 4 * GRU_H * sizeof(float); }

int gru_seq_bwd_multi(int32_t n_nets, const mo_gru_seq_bwd_net *nets, int32_t T, int32_t B, int32_t H, int32_t gi_agents, void *stream) {
    if (n_nets < 1 || n_nets > MO_GRU_MAX_NETS || !nets || T < 1 || B < 1 || H != GRU_H || gi_agents < 0 || (gi_agents && B % gi_agents)) return MO_ERR_BAD_ARG;
    GruBwdNets a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < n_nets; k++) {
        const mo_gru_seq_bwd_net &m = nets[k];
        if (!m.dout || !m.save || !m.out || !m.h0 || !m.w_hh || !m.dgi || !m.dh0) return MO_ERR_BAD_ARG;
        if ((m.dgh == nullptr) == (m.dnr == nullptr)) return MO_ERR_BAD_ARG;        // exactly one of the two forms
        if ((m.db_ih || m.db_hh) && (!m.db_ih || !m.db_hh || !m.workspace)) return MO_ERR_BAD_ARG;
        if ((((uintptr_t)m.dout | (uintptr_t)m.save | (uintptr_t)m.out | (uintptr_t)m.h0 | (uintptr_t)m.dgi | (uintptr_t)m.dgh | (uintptr_t)m.dnr |
              (uintptr_t)m.dh0 | (uintptr_t)m.workspace) & 15)) return MO_ERR_BAD_ARG;
        if (m.B < 0 || m.B > B || (gi_agents && m.B % gi_agents)) return MO_ERR_BAD_ARG;
        a.n[k] = m;
    }
    const int nblk = (B + GRU_RB - 1) / GRU_RB;
    hipLaunchKernelGGL(k_gru_seq_bwd2, dim3(nblk, n_nets), dim3(512), 0, (hipStream_t)stream, T, B, a, (int)gi_agents);
    for (int k = 0; k < n_nets; k++)
        if (nets[k].db_ih) {
            const int nb = ((nets[k].B > 0 ? nets[k].B : B) + GRU_RB - 1) / GRU_RB;
            hipLaunchKernelGGL(k_gru_bias_reduce, dim3(4 * GRU_H / 4), dim3(256), 0, (hipStream_t)stream, nb, (const float *)nets[k].workspace, nets[k].db_ih,
                               nets[k].db_hh);
        }
    return (int)hipGetLastError();
}

int gru_seq_bwd(int32_t T, int32_t B, int32_t H, const float *dout, const float *save, const float *out, const float *h0, const float *w_hh,
                float *dgi, float *dgh, float *dnr, float *dh0, float *db_ih, float *db_hh, int32_t gi_agents, void *workspace, void *stream) {
    const mo_gru_seq_bwd_net net{dout, save, out, h0, w_hh, dgi, dgh, dnr, dh0, db_ih, db_hh, workspace, 0, 0};
    return gru_seq_bwd_multi(1, &net, T, B, H, gi_agents, stream);
}

int64_t wgrad_tn_workspace(int32_t M, int32_t N) {
    if (M < 128 || N < 128 || (M & 127) || (N & 127)) return -1;
    int am, bn;
    return (int64_t)wgrad_split(M, N, &am, &bn) * M * N * sizeof(float);
}

int wgrad_tn(int64_t K, int32_t M, int32_t N, const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int32_t accumulate,
             void *workspace, void *stream) {
    if (K < 1 || M < 128 || N < 128 || (M & 127) || (N & 127) || M > 1024 || N > 1024 || !A || !B || !C || !workspace) return MO_ERR_BAD_ARG;
    if (lda < M || ldb < N || (lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return MO_ERR_BAD_ARG;
    int am, bn;
    const int S = wgrad_split(M, N, &am, &bn);
    const dim3 grid(S, M / (128 * am), N / (128 * bn));
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
    if (am == 3) hipLaunchKernelGGL((k_wgrad<3, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, K, (int)M, (int)N, part);
    else if (bn == 3) hipLaunchKernelGGL((k_wgrad<1, 3>), grid, dim3(256), 0, st, A, lda, B, ldb, K, (int)M, (int)N, part);
    else hipLaunchKernelGGL((k_wgrad<1, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, K, (int)M, (int)N, part);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((M * N + 63) / 64), dim3(256), 0, st, S, M * N, (const float *)part, C, (int)accumulate);
    return (int)hipGetLastError();
}

int64_t wgrad_split_workspace(int32_t M, int32_t N) {
    if (!sb_wgrad_shape_ok(M, N)) return -1;
    return (int64_t)SB_WGRAD_WGS * M * N * sizeof(float);
}

int wgrad_split_tn(int64_t K, int32_t M, int32_t N, const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int32_t accumulate,
                   void *workspace, void *stream) {
    if (K < 1 || !sb_wgrad_shape_ok(M, N) || !A || !B || !C || !workspace) return MO_ERR_BAD_ARG;
    if (lda < M || ldb < N || (lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return MO_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
#define SB_WG(mt, nt) if (M == 128 * mt && N == 128 * nt) return launch_sb_wgrad<mt, nt>(K, A, lda, B, ldb, C, accumulate, part, st);
    SB_WG(1, 1) SB_WG(1, 2) SB_WG(2, 1) SB_WG(1, 3) SB_WG(3, 1)
#undef SB_WG
    return MO_ERR_BAD_ARG;
}

int wgrad_split_tn2(int64_t K, int32_t M1, int32_t M2, int32_t N, const float *A1, int64_t lda1, const float *A2, int64_t lda2, const float *B, int64_t ldb,
                    float *C, int32_t accumulate, void *workspace, void *stream) {
    const int M = M1 + M2;
    if (K < 1 || M1 < 4 || M2 < 4 || (M1 & 3) || (M2 & 3) || !sb_wgrad_shape_ok(M, N) || !A1 || !A2 || !B || !C || !workspace) return MO_ERR_BAD_ARG;
    if (lda1 < M1 || lda2 < M2 || ldb < N || (lda1 & 3) || (lda2 & 3) || (ldb & 3) || ((uintptr_t)A1 & 15) || ((uintptr_t)A2 & 15) || ((uintptr_t)B & 15))
        return MO_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
#define SB_WG(mt, nt) if (M == 128 * mt && N == 128 * nt) return launch_sb_wgrad<mt, nt>(K, A1, lda1, B, ldb, C, accumulate, part, st, A2, lda2, M1);
    SB_WG(1, 1) SB_WG(1, 2) SB_WG(2, 1) SB_WG(1, 3) SB_WG(3, 1)
#undef SB_WG
    return MO_ERR_BAD_ARG;
}

int fcra_neighbour_mean(int32_t R, int32_t P, int32_t E, int32_t T, const float *z_actor, int64_t za_episode_stride, int64_t za_step_stride,
                        const float *z_critic, int64_t zc_episode_stride, int64_t zc_step_stride, const float *adj, int64_t adj_row_stride,
                        const float *bias, int32_t relu, float *out_actor, float *out_critic, int64_t out_stride, void *stream) {
    const mo_nbr_job job{z_actor, z_critic, bias, out_actor, out_critic};
    return fcra_neighbour_mean_multi(1, &job, R, P, E, T, za_episode_stride, za_step_stride, zc_episode_stride, zc_step_stride, adj, adj_row_stride, relu,
                                     out_stride, stream);
}

int fcra_neighbour_mean_multi(int32_t n_jobs, const mo_nbr_job *jobs, int32_t R, int32_t P, int32_t E, int32_t T, int64_t za_episode_stride,
                              int64_t za_step_stride, int64_t zc_episode_stride, int64_t zc_step_stride, const float *adj, int64_t adj_row_stride,
                              int32_t relu, int64_t out_stride, void *stream) {
    if (n_jobs < 1 || n_jobs > MO_NBR_MAX_JOBS || !jobs) return MO_ERR_BAD_ARG;
    if (R < 0 || P < 1 || P > MAX_P || E < 64 || E > 1024 || (E & 63) || T < 1 || (R % T) || out_stride < E) return MO_ERR_BAD_ARG;
    NbrJobs a_jobs;
    memset(&a_jobs, 0, sizeof a_jobs);
    for (int k = 0; k < n_jobs; k++) {
        const mo_nbr_job &m = jobs[k];
        if ((m.out_actor && (!m.z_actor || !adj)) || (m.out_critic && !m.z_critic) || (!m.out_actor && !m.out_critic)) return MO_ERR_BAD_ARG;
        a_jobs.j[k] = m;
    }
    if (R == 0) return 0;
    NbrArgs a{R, P, E, T, relu, za_episode_stride, za_step_stride, zc_episode_stride, zc_step_stride, adj_row_stride, out_stride};
    if (E == 128 && P <= 8 && !(out_stride & 3) && !(za_episode_stride & 3) && !(za_step_stride & 3) && !(zc_episode_stride & 3) && !(zc_step_stride & 3)) {
        bool aligned = true;
        for (int k = 0; k < n_jobs; k++) {
            const mo_nbr_job &m = jobs[k];
            aligned = aligned && !(((uintptr_t)m.z_actor | (uintptr_t)m.z_critic | (uintptr_t)m.bias | (uintptr_t)m.out_actor | (uintptr_t)m.out_critic) & 15);
        }
        if (aligned) {
            const int per4 = 4096 / n_jobs, blocks = (R + 3) / 4;
            hipLaunchKernelGGL(k_nbr_mean4, dim3(blocks < per4 ? blocks : per4, n_jobs), dim3(256), 0, (hipStream_t)stream, a, a_jobs, adj);
            return (int)hipGetLastError();
        }
    }
    const int per = 16384 / n_jobs, grid = R < per ? R : per;
    if (P <= 8) hipLaunchKernelGGL(k_nbr_mean<8>, dim3(grid, n_jobs), dim3(E), 0, (hipStream_t)stream, a, a_jobs, adj);
    else hipLaunchKernelGGL(k_nbr_mean<16>, dim3(grid, n_jobs), dim3(E), 0, (hipStream_t)stream, a, a_jobs, adj);
    return (int)hipGetLastError();
}

int64_t relu_bwd_colsum_workspace(int32_t F) { return (int64_t)RB_BLOCKS * F * sizeof(float); }

int relu_bwd_colsum(int64_t R, int32_t F, const float *gout, int64_t g_row_stride, const float *y, int64_t y_row_stride, float *gin, float *colsum,
                    void *workspace, void *stream) {
    if (R < 1 || F < 4 || (F & 3) || F > 1024 || (RB_TPB % (F >> 2)) || !gout || !y || !gin || !colsum || !workspace) return MO_ERR_BAD_ARG;
    if (g_row_stride < F || y_row_stride < F || (g_row_stride & 3) || (y_row_stride & 3)) return MO_ERR_BAD_ARG;
    if ((((uintptr_t)gout | (uintptr_t)y | (uintptr_t)gin) & 15)) return MO_ERR_BAD_ARG;
    const int64_t rows_per_step = RB_TPB / (F >> 2);
    int64_t want = (R + rows_per_step - 1) / rows_per_step;
    const int grid = want < RB_BLOCKS ? (int)want : RB_BLOCKS;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_relu_bwd_colsum, dim3(grid), dim3(RB_TPB), 0, st, R, (int)F, gout, g_row_stride >> 2, y, y_row_stride >> 2, gin, (float *)workspace);
    hipLaunchKernelGGL(k_colsum_reduce, dim3((F + 31) / 32), dim3(256), 0, st, grid, (int)F, (const float *)workspace, colsum);
    return (int)hipGetLastError();
}

int64_t wgrad_skinny_workspace(int32_t NS, int32_t F) { return (int64_t)SK_BLOCKS * ((NS + 1) * F + NS) * sizeof(float); }

int wgrad_skinny(int64_t R, int32_t NS, int32_t F, const float *S, int64_t lds, const float *X, int64_t ldx, int32_t transposed, float *C,
                 float *colsum_x, float *colsum_s, void *workspace, void *stream) {
    if (R < 1 || NS < 1 || NS > SK_MAX || F < 64 || F > 1024 || (F & 63) || !S || !X || !C || !workspace || lds < NS || ldx < F) return MO_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int grid = R < SK_BLOCKS ? (int)R : SK_BLOCKS;
    float *part = (float *)workspace;
    switch (NS) {
#define SK_CASE(n) case n: hipLaunchKernelGGL((k_wgrad_skinny<n>), dim3(grid), dim3(F), 0, st, R, (int)F, S, lds, X, ldx, part); break;
        SK_CASE(1) SK_CASE(2) SK_CASE(3) SK_CASE(4) SK_CASE(5) SK_CASE(6) SK_CASE(7) SK_CASE(8) SK_CASE(9) SK_CASE(10) SK_CASE(11) SK_CASE(12)
        SK_CASE(13) SK_CASE(14) SK_CASE(15) SK_CASE(16)
#undef SK_CASE
    }
    const int per = (NS + 1) * F + NS;
    hipLaunchKernelGGL(k_skinny_reduce, dim3((per + 3) / 4), dim3(256), 0, st, grid, (int)NS, (int)F, (const float *)part, (int)transposed, C, colsum_x,
                       colsum_s);
    return (int)hipGetLastError();
}

int rollout_record(int32_t N, int32_t n_items, const mo_record_item *items, const float *raw, float *episode_return, int32_t P, void *stream) {
    if (N < 1 || n_items < 0 || n_items > MO_RECORD_MAX_ITEMS || (n_items && !items) || (raw && (!episode_return || P < 1))) return MO_ERR_BAD_ARG;
    RecordArgs a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < n_items; k++) {
        if (!items[k].src || !items[k].dst || items[k].row_bytes < 4 || (items[k].row_bytes & 3) || (items[k].dst_row_stride & 3) ||
            (((uintptr_t)items[k].src | (uintptr_t)items[k].dst) & 3))
            return MO_ERR_BAD_ARG;
        a.it[k] = items[k];
    }
    a.n_items = n_items;
    a.raw = raw;
    a.ret = episode_return;
    a.P = P;
    hipLaunchKernelGGL(k_rollout_record, dim3(N), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

int64_t ppo_loss_workspace(void) { return (int64_t)PPO_BLOCKS * 2 * sizeof(double); }

int ppo_loss_fwd_bwd(int64_t n, const float *logp_now, const float *entropy, const float *logp_old, const float *adv, const float *active,
                     const float *values_now, const float *values_old, const float *v_target, const float *active_sum, float epsilon,
                     float entropy_coef, int32_t use_value_clip, float *losses, float *grad_logp, float *grad_entropy, float *grad_values,
                     void *workspace, void *stream) {
    if (n < 1 || !logp_now || !entropy || !logp_old || !adv || !active || !values_now || !v_target || !active_sum || !losses || !grad_logp ||
        !grad_entropy || !grad_values || !workspace || (use_value_clip && !values_old))
        return MO_ERR_BAD_ARG;
    long blocks = (n + 255) / 256;
    if (blocks > PPO_BLOCKS) blocks = PPO_BLOCKS;
    hipLaunchKernelGGL(k_ppo_loss, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long)n, logp_now, entropy, logp_old, adv, active,
                       values_now, values_old, v_target, active_sum, epsilon, entropy_coef, (int)use_value_clip, grad_logp, grad_entropy,
                       grad_values, (double *)workspace);
    hipLaunchKernelGGL(k_ppo_loss_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, (int)blocks, (const double *)workspace, active_sum, losses);
    return (int)hipGetLastError();
}

int ppo_loss_prob_fwd_bwd(int64_t n, int32_t A, const float *prob, float *grad_prob, int64_t d1, int64_t d2, int64_t p_s0, int64_t p_s1, int64_t p_s2,
                          const float *action, const float *logp_old, const float *adv, const float *active, const float *values_now, int64_t v_s0,
                          int64_t v_s1, int64_t v_s2, const float *values_old, const float *v_target, const float *active_sum, float epsilon,
                          float entropy_coef, int32_t use_value_clip, float *losses, float *grad_values, void *workspace, void *stream) {
    if (n < 1 || A < 1 || A > PPO_MAX_A || d1 < 1 || d2 < 1 || (n % (d1 * d2)) || !prob || !grad_prob || !action || !logp_old || !adv || !active ||
        !values_now || !v_target || !active_sum || !losses || !grad_values || !workspace || (use_value_clip && !values_old))
        return MO_ERR_BAD_ARG;
    long blocks = (n + 255) / 256;
    if (blocks > PPO_BLOCKS) blocks = PPO_BLOCKS;
    const PpoView pv{d1, d2, p_s0, p_s1, p_s2}, vv{d1, d2, v_s0, v_s1, v_s2};
    hipLaunchKernelGGL(k_ppo_loss_prob, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long)n, (int)A, prob, pv, action, logp_old, adv, active,
                       values_now, vv, values_old, v_target, active_sum, epsilon, entropy_coef, (int)use_value_clip, grad_prob, grad_values,
                       (double *)workspace);
    hipLaunchKernelGGL(k_ppo_loss_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, (int)blocks, (const double *)workspace, active_sum, losses);
    return (int)hipGetLastError();
}

const char *mappo_ops_error_string(int code) {
    if (code == MO_ERR_BAD_ARG) return "mappo_ops: bad argument";
    return hipGetErrorString((hipError_t)code);
}

}  // extern "C"
