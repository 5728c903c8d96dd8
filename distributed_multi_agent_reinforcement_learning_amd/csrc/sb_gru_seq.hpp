// sb_gru_seq.hpp -- the persistent GRU recurrences of the update (torch.nn.GRU over T steps, H = 128; reference
// DHGN/mappo_parallel.py:397, :432-436) in fp32 arithmetic on the bf16 matrix pipe.  C ABI: include/mappo_ops.h
// gru_seq_split_fwd_multi / gru_seq_split_bwd_multi; same records, same save layout and same results to fp32 rounding as
// gru_seq_fwd_multi / gru_seq_bwd_multi (csrc/mappo_ops.hip k_gru_seq_fwd2 / k_gru_seq_bwd2), which stay the `runtime.matmul: fp32` route.
//
// k_gru_seq_fwd2 / bwd2 hold W_hh as v_mfma_f32_16x16x4_f32 operands and spend 6 144 matrix-pipe cycles per step and SIMD; as the update
// launches them (all mini-batches and both networks per launch) they are co-bound by that and by HBM (4 096 B per row and step forward,
// 5 120 B backward).  Here W_hh lives in the registers as three bf16 pieces (144 registers per lane, split once), the h tile (forward)
// / the gate-gradient tile (backward) is split by the lanes that produce it and written straight into the LDS image in B-operand
// order (8 bytes per piece and lane), and a step's product is 72 v_mfma_f32_16x16x32_bf16 per wave: 2 304 pipe cycles per step and
// SIMD.  What remains is the HBM traffic.  The large piece product and the five small ones accumulate separately (sb_mma6_hl).
#pragma once
#include <string.h>

#include <mutex>

#include "mappo_ops.h"
#include "sb_common.hpp"

constexpr int SBR_H = 128, SBR_RB = 16;

struct SbGruFwdNets { mo_gru_seq_net n[MO_GRU_MAX_NETS]; };
struct SbGruBwdNets { mo_gru_seq_bwd_net n[MO_GRU_MAX_NETS]; };

__device__ __forceinline__ float sbr_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float sbr_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
__device__ __forceinline__ size_t sbr_gi_row(int b, int t, int T, int B, int gi_agents) {
    return gi_agents ? ((size_t)(b / gi_agents) * T + t) * gi_agents + b % gi_agents : (size_t)t * B + b;
}

// four consecutive contraction steps (k0 .. k0 + 3, k0 a multiple of 4) of operand column `col`: split and stored as the matching half
// of the 16-byte words of a B-operand image [piece][chunk of 32 steps][lane (column, step octet)] with `chunks` chunks
__device__ __forceinline__ void sbr_put4(uint4 *img, int chunks, int k0, int col, const float4 &v) {
    uint32_t a[3], b[3];
    sb_split2(v.x, v.y, a[0], a[1], a[2]);
    sb_split2(v.z, v.w, b[0], b[1], b[2]);
    const int c = k0 >> 5, oct = (k0 & 31) >> 3, half = (k0 >> 2) & 1;
    uint2 *dst = (uint2 *)(img + c * 64 + oct * 16 + col) + half;
#pragma unroll
    for (int s = 0; s < 3; s++) dst[(size_t)s * chunks * 64 * 2] = make_uint2(a[s], b[s]);
}

// ---- forward: h_t = cell(gi_t, h_{t-1}) for 16 batch rows per workgroup, all T steps in one launch ---------------------------------
// Wave w owns hidden units 16 w .. 16 w + 15 of the three gates (A operands: W_hh rows, 3 x 4 chunks x 3 pieces); the result tile is
// gate^T: lane (row = l % 16, q = l / 16) owns units 16 w + 4 q .. + 3 of ONE batch row -- k_gru_seq_fwd2's layout, so gi / out move
// as 16-byte accesses and the saved gates keep that kernel's (lane-ordered) layout.  A lane's h values stay in its registers from
// step to step (the state update needs h itself); the image holds only the pieces.
__global__ __launch_bounds__(512) void k_gru_seq_fwd_sb(int T, int Bmax, SbGruFwdNets nets, int gi_agents) {
    const mo_gru_seq_net &net = nets.n[blockIdx.y];
    const int B = net.B > 0 ? net.B : Bmax;
    if ((int)blockIdx.x * SBR_RB >= B) return;
    const float *__restrict__ gi = net.gi, *__restrict__ w_hh = net.w_hh, *__restrict__ b_hh = net.b_hh, *__restrict__ h0 = net.h0;
    float *__restrict__ out = net.out, *__restrict__ save = net.save;
    __shared__ uint4 himg[2][3 * 4 * 64];                 // [buffer][piece][chunk][lane]: 2 x 12 KB
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, i = l & 15, gq = l >> 4;
    const int b0 = blockIdx.x * SBR_RB;
    uint4 wr[4][3], wz[4][3], wn[4][3];
    {
        const float *rr = w_hh + (size_t)(16 * w + i) * SBR_H + 8 * gq, *rz = rr + (size_t)SBR_H * SBR_H, *rn = rz + (size_t)SBR_H * SBR_H;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            sb_split8(*(const float4 *)(rr + 32 * c), *(const float4 *)(rr + 32 * c + 4), wr[c]);
            sb_split8(*(const float4 *)(rz + 32 * c), *(const float4 *)(rz + 32 * c + 4), wz[c]);
            sb_split8(*(const float4 *)(rn + 32 * c), *(const float4 *)(rn + 32 * c + 4), wn[c]);
        }
    }
    const int row = i, u0 = 16 * w + 4 * gq;
    const bool live = b0 + row < B;
    const float4 br = *(const float4 *)(b_hh + u0), bz = *(const float4 *)(b_hh + SBR_H + u0), bn = *(const float4 *)(b_hh + 2 * SBR_H + u0);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 hprev = live ? *(const float4 *)(h0 + (size_t)(b0 + row) * SBR_H + u0) : zero4;
    sbr_put4(himg[0], 4, u0, row, hprev);
    lds_barrier();
    const size_t nblk = (size_t)(B + SBR_RB - 1) / SBR_RB;   // of THIS layer (the save area is laid out per layer)
    float4 *sv = save ? (float4 *)save + (size_t)blockIdx.x * 4 * 512 + tid : nullptr;
    float4 pgr = zero4, pgz = zero4, pgn = zero4;
    auto prefetch = [&](int t) {
        if (live) {
            const float *g = gi + sbr_gi_row(b0 + row, t, T, B, gi_agents) * 3 * SBR_H + u0;
            pgr = *(const float4 *)g; pgz = *(const float4 *)(g + SBR_H); pgn = *(const float4 *)(g + 2 * SBR_H);
        }
    };
    prefetch(0);
    int cur = 0;
    for (int t = 0; t < T; t++) {
        const float4 gr = pgr, gz = pgz, gn = pgn;
        if (t + 1 < T) prefetch(t + 1);                   // no dependence on the recurrence: in flight during this step's products
        const uint4 *tb = himg[cur] + l;
        f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar, lr = ar, lz = ar, ln = ar;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            uint4 b[3];
#pragma unroll
            for (int p = 0; p < 3; p++) b[p] = tb[(p * 4 + c) * 64];
            // the three gates' products interleaved: consecutive instructions never share an accumulator
#define SBR_MMA(pi, pj, R_, Z_, N_)                                                                                                                   \
            R_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wr[c][pi]), __builtin_bit_cast(bf16x8, b[pj]), R_, 0, 0, 0);        \
            Z_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wz[c][pi]), __builtin_bit_cast(bf16x8, b[pj]), Z_, 0, 0, 0);        \
            N_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wn[c][pi]), __builtin_bit_cast(bf16x8, b[pj]), N_, 0, 0, 0);
            SBR_MMA(2, 0, lr, lz, ln) SBR_MMA(0, 2, lr, lz, ln) SBR_MMA(1, 1, lr, lz, ln) SBR_MMA(0, 0, ar, az, an) SBR_MMA(1, 0, lr, lz, ln)
            SBR_MMA(0, 1, lr, lz, ln)
#undef SBR_MMA
        }
        ar += lr; az += lz; an += ln;
        float4 r, z, hn, n, hnew;
#define SBR_ONE(f, q_)                                   \
        r.f = sbr_sigmoid(gr.f + ar[q_] + br.f);         \
        z.f = sbr_sigmoid(gz.f + az[q_] + bz.f);         \
        hn.f = an[q_] + bn.f;                            \
        n.f = sbr_tanh(gn.f + r.f * hn.f);               \
        hnew.f = (1.f - z.f) * n.f + z.f * hprev.f;
        SBR_ONE(x, 0) SBR_ONE(y, 1) SBR_ONE(z, 2) SBR_ONE(w, 3)
#undef SBR_ONE
        if (!live) hnew = zero4;                          // dead rows stay zero (their gi was never loaded)
        sbr_put4(himg[cur ^ 1], 4, u0, row, hnew);
        hprev = hnew;
        if (live) *(float4 *)(out + ((size_t)t * B + b0 + row) * SBR_H + u0) = hnew;
        if (sv) {   // lane order; rows past B are padding of the (opaque) save area
            float4 *s4 = sv + (size_t)t * nblk * 4 * 512;
            s4[0] = r; s4[512] = z; s4[1024] = n; s4[1536] = hn;
        }
        lds_barrier();
        cur ^= 1;
    }
}

inline int launch_gru_seq_fwd_sb(int n_nets, const mo_gru_seq_net *nets, int T, int B, int gi_agents, hipStream_t st) {
    SbGruFwdNets a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < n_nets; k++) a.n[k] = nets[k];
    const int nblk = (B + SBR_RB - 1) / SBR_RB;
    hipLaunchKernelGGL(k_gru_seq_fwd_sb, dim3(nblk, n_nets), dim3(512), 0, st, T, B, a, gi_agents);
    return (int)hipGetLastError();
}

// ---- backward: the reverse recurrence of the same 16 rows ------------------------------------------------------------------------
// dh_t = dout_t + dcarry;  (dr, dz, dn, dnr) from the saved gates;  dcarry' = dh z + [dr dz dnr] W_hh  (contraction over the 384 gate
// columns).  Wave w owns the 16 output units 16 w .. of dcarry (A operand: the W_hh^T tile as 12 chunks x 3 pieces); the lanes' gate
// gradients -- lane (row, q) holds units 16 w + 4 q .. + 3 of the three gates, k_gru_seq_bwd2's layout -- are split where they are
// produced and written into the B-operand image (36 KB, double-buffered: one barrier per step).  Five accumulators (two for the large
// piece product, three for the small ones) keep dependent matrix instructions four apart.  Outputs, row orders and the bias-gradient
// partials are k_gru_seq_bwd2's.
__global__ __launch_bounds__(512) void k_gru_seq_bwd_sb(int T, int Bmax, SbGruBwdNets nets, int gi_agents) {
    const mo_gru_seq_bwd_net &net = nets.n[blockIdx.y];
    const int B = net.B > 0 ? net.B : Bmax;
    if ((int)blockIdx.x * SBR_RB >= B) return;
    const float *__restrict__ dout = net.dout, *__restrict__ save = net.save, *__restrict__ out = net.out, *__restrict__ h0 = net.h0,
                *__restrict__ w_hh = net.w_hh;
    float *__restrict__ dgi = net.dgi, *__restrict__ dgh = net.dgh, *__restrict__ dnr_out = net.dnr, *__restrict__ dh0 = net.dh0;
    float *__restrict__ bias_partials = net.db_ih ? (float *)net.workspace : nullptr;
    extern __shared__ uint4 sbr_gimg[];                   // [buffer][piece][chunk 12][lane]: 2 x 36 KB
    constexpr int IMG = 3 * 12 * 64;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, i = l & 15, gq = l >> 4;
    const int b0 = blockIdx.x * SBR_RB;
    // A operand = W_hh^T: lane (unit i, octet gq) holds W_hh[32 c + 8 gq + s][16 w + i], s = 0 .. 7, for the 12 chunks c
    uint4 wt[12][3];
#pragma unroll
    for (int c = 0; c < 12; c++) {
        const float *col = w_hh + (size_t)(32 * c + 8 * gq) * SBR_H + 16 * w + i;
        const float4 u = make_float4(col[0], col[SBR_H], col[2 * SBR_H], col[3 * SBR_H]);
        const float4 v = make_float4(col[4 * SBR_H], col[5 * SBR_H], col[6 * SBR_H], col[7 * SBR_H]);
        sb_split8(u, v, wt[c]);
    }
    const int row = i, u0 = 16 * w + 4 * gq;
    const bool live = b0 + row < B;
    const size_t nblk = (size_t)(B + SBR_RB - 1) / SBR_RB;   // of THIS layer (the save area is laid out per layer)
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
            const size_t o = (size_t)(b0 + row) * SBR_H + u0;
            php = *(const float4 *)(t > 0 ? out + (size_t)(t - 1) * B * SBR_H + o : h0 + o);
            pdo = *(const float4 *)(dout + (size_t)t * B * SBR_H + o);
        }
    };
    prefetch(T - 1);
    int buf = 0;
    for (int t = T - 1; t >= 0; t--) {
        float4 dr, dz, dn, dnr, dhz;
        {
            const float4 r = pr, z = pz, n = pn, hn = phn, hp = php, dO = pdo;
#define SBR_ONE(f)                                               \
            {                                                    \
                const float dh = dO.f + dcarry.f;                \
                dn.f = dh * (1.f - z.f) * (1.f - n.f * n.f);     \
                dz.f = dh * (hp.f - n.f) * z.f * (1.f - z.f);    \
                dr.f = dn.f * hn.f * r.f * (1.f - r.f);          \
                dnr.f = dn.f * r.f;                              \
                dhz.f = dh * z.f;                                \
            }
            SBR_ONE(x) SBR_ONE(y) SBR_ONE(z) SBR_ONE(w)
#undef SBR_ONE
        }
        if (t > 0) prefetch(t - 1);
        uint4 *gb = sbr_gimg + buf * IMG;   // dead rows carry zeros (their loads were skipped)
        sbr_put4(gb, 12, u0, row, dr);
        sbr_put4(gb, 12, SBR_H + u0, row, dz);
        sbr_put4(gb, 12, 2 * SBR_H + u0, row, dnr);
        if (live) {
            float *g = dgi + sbr_gi_row(b0 + row, t, T, B, gi_agents) * 3 * SBR_H + u0;
            *(float4 *)g = dr; *(float4 *)(g + SBR_H) = dz; *(float4 *)(g + 2 * SBR_H) = dn;
            const size_t tb = (size_t)t * B + b0 + row;
            if (dgh) {
                float *h = dgh + tb * 3 * SBR_H + u0;
                *(float4 *)h = dr; *(float4 *)(h + SBR_H) = dz; *(float4 *)(h + 2 * SBR_H) = dnr;
            } else {
                *(float4 *)(dnr_out + tb * SBR_H + u0) = dnr;
            }
#define SBR_ACC(S, V) S.x += V.x; S.y += V.y; S.z += V.z; S.w += V.w;
            SBR_ACC(sb_r, dr) SBR_ACC(sb_z, dz) SBR_ACC(sb_n, dn) SBR_ACC(sb_nr, dnr)
#undef SBR_ACC
        }
        lds_barrier();
        const uint4 *tb = gb + l;
        f32x4 hi0 = {0.f, 0.f, 0.f, 0.f}, hi1 = hi0, lo0 = hi0, lo1 = hi0, lo2 = hi0;
#pragma unroll
        for (int c = 0; c < 12; c++) {
            uint4 b[3];
#pragma unroll
            for (int p = 0; p < 3; p++) b[p] = tb[(p * 12 + c) * 64];
#define SBR_MMA(pi, pj, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wt[c][pi]), __builtin_bit_cast(bf16x8, b[pj]), ACC, 0, 0, 0);
            if (c & 1) { SBR_MMA(2, 0, lo0) SBR_MMA(0, 2, lo1) SBR_MMA(1, 1, lo2) SBR_MMA(0, 0, hi1) SBR_MMA(1, 0, lo0) SBR_MMA(0, 1, lo1) }
            else       { SBR_MMA(2, 0, lo2) SBR_MMA(0, 2, lo0) SBR_MMA(1, 1, lo1) SBR_MMA(0, 0, hi0) SBR_MMA(1, 0, lo2) SBR_MMA(0, 1, lo0) }
#undef SBR_MMA
        }
        const f32x4 s = (hi0 + hi1) + ((lo0 + lo1) + lo2);
        dcarry.x = dhz.x + s[0]; dcarry.y = dhz.y + s[1]; dcarry.z = dhz.z + s[2]; dcarry.w = dhz.w + s[3];
        buf ^= 1;   // the other image: its last readers passed this step's barrier before anyone writes it again
    }
    if (live) *(float4 *)(dh0 + (size_t)(b0 + row) * SBR_H + u0) = dcarry;
    // bias gradients: sums over this workgroup's 16 rows (the lanes of a 16-lane group) and all steps
    if (bias_partials) {
        float s[16] = {sb_r.x, sb_r.y, sb_r.z, sb_r.w, sb_z.x, sb_z.y, sb_z.z, sb_z.w, sb_n.x, sb_n.y, sb_n.z, sb_n.w, sb_nr.x, sb_nr.y, sb_nr.z, sb_nr.w};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            s[k] += __shfl_xor(s[k], 1); s[k] += __shfl_xor(s[k], 2); s[k] += __shfl_xor(s[k], 4); s[k] += __shfl_xor(s[k], 8);
        }
        if (i == 0) {
            float *bp = bias_partials + (size_t)blockIdx.x * 4 * SBR_H + u0;
#pragma unroll
            for (int g = 0; g < 4; g++) *(float4 *)(bp + g * SBR_H) = make_float4(s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]);
        }
    }
}

// column sums of dgi / dgh from the per-workgroup partials [block][4][H] (r, z, n, nr), f64 accumulation, fixed order
__global__ void k_sbr_bias_reduce(int nblk, const float *partials, float *db_ih, float *db_hh) {
    const int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // 4 * H columns, one wave each
    const int lane = threadIdx.x & 63;
    if (idx >= 4 * SBR_H) return;
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += (double)partials[(size_t)b * 4 * SBR_H + idx];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
        const int g = idx / SBR_H, jj = idx - g * SBR_H;
        if (g == 0) { db_ih[jj] = (float)s; db_hh[jj] = (float)s; }
        else if (g == 1) { db_ih[SBR_H + jj] = (float)s; db_hh[SBR_H + jj] = (float)s; }
        else if (g == 2) db_ih[2 * SBR_H + jj] = (float)s;
        else db_hh[2 * SBR_H + jj] = (float)s;
    }
}

constexpr int SBR_BWD_LDS = 2 * 3 * 12 * 64 * 16;

inline int launch_gru_seq_bwd_sb(int n_nets, const mo_gru_seq_bwd_net *nets, int T, int B, int gi_agents, hipStream_t st) {
    static std::once_flag once;
    static hipError_t attr_rc = hipSuccess;
    std::call_once(once, [] { attr_rc = hipFuncSetAttribute((const void *)k_gru_seq_bwd_sb, hipFuncAttributeMaxDynamicSharedMemorySize, SBR_BWD_LDS); });
    if (attr_rc != hipSuccess) return (int)attr_rc;
    SbGruBwdNets a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < n_nets; k++) a.n[k] = nets[k];
    const int nblk = (B + SBR_RB - 1) / SBR_RB;
    hipLaunchKernelGGL(k_gru_seq_bwd_sb, dim3(nblk, n_nets), dim3(512), SBR_BWD_LDS, st, T, B, a, gi_agents);
    for (int k = 0; k < n_nets; k++)
        if (nets[k].db_ih) {
            const int nb = ((nets[k].B > 0 ? nets[k].B : B) + SBR_RB - 1) / SBR_RB;
            hipLaunchKernelGGL(k_sbr_bias_reduce, dim3(4 * SBR_H / 4), dim3(256), 0, st, nb, (const float *)nets[k].workspace, nets[k].db_ih, nets[k].db_hh);
        }
    return (int)hipGetLastError();
}
