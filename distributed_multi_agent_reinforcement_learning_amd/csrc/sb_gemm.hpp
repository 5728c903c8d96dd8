// sb_gemm.hpp -- Linear layers in fp32 arithmetic on the bf16 matrix pipe (C ABI: include/mappo_ops.h sb_gemm)
#pragma once
#include <mutex>

#include "sb_common.hpp"

#ifndef SBG_ABLATE
#define SBG_ABLATE 0   // lab only (tools/microbench/sb_gemm_lab.hip): 1 = no matrix instructions, 2 = no result stores, 4 = no operand loads
#endif

// ---- Y = act(X W^T + b [+ C]) for 128 (384) outputs, fp32 arithmetic on the bf16 matrix pipe (k_sb_gemm_n128) ------------------
// The rollout's Linear layers (DHGN AGG / semantic / FCRA layers, reference DHGN/mappo_parallel.py:148-233: 3e4-2e5 rows against a
// 128 x {128, 256, 384} weight; and the update's GRU input projection, 384 x 128) on the exact three-way bf16 split of k_gru_cell_sb.
// Persistent workgroups; wave w keeps output units 16 (w + 8 t) .. + 15 of W as A-operands (12 registers per 32 inputs and tile, split once); 32 rows per iteration stream through a
// double-buffered LDS image: a wave stages (chunk, half) blocks -- lane (gq, j) loads the 8 inputs 32 c + 8 gq .. of row j (16 rows
// x 128 contiguous bytes per instruction), splits them and writes one 16-byte word per piece, which IS lane (j, gq)'s B-operand.
// The result tile has a lane own four consecutive outputs of one row: bias, the optional addend (may be Y itself: beta = 1) and
// ReLU in registers, one 16-byte store.  X, C and Y may be column blocks of wider matrices (row strides).
// OPT (bit mask; the product launches 3): 1 = the addend rows of a tile are requested BEFORE its matrix phase (round 3 loaded them
// in the epilogue: one exposed HBM latency per 32 rows, 138 us against 47 us for the rollout's in-place semantic layer at 65 536
// rows); 2 = the two waves of a SIMD (w and w + 4) run out of phase -- one splits and stages the next tile while the other
// multiplies, as in k_sb_wgrad -- instead of all eight waves staging, then all multiplying;  4 = the ReLU-backward form of an input
// gradient (sb_gemm_masked): `addend` is not added but read as a MASK -- outputs of the first mask_tiles 128-column blocks are kept
// where it is positive, zeroed elsewhere -- and the column sums of the result (the bias gradient of the layer whose ReLU this is)
// leave the kernel as one row of partial sums per workgroup (fixed order: k_sb_colsum_reduce adds them deterministically);
// 8 = the kernel also writes the SIGN BITS of its result (bit k of byte j of a row: Y[row][8 j + k] > 0; two lanes' nibbles joined with
// one shuffle, ybits rows ldyb bytes apart): behind a ReLU epilogue these are relu'(Y), and the backward's masked input gradient of the
// layer that consumes Y (bit 16: with bit 4, `addend` is that byte matrix, lda its row stride in BYTES) reads 1 bit per element instead of
// the 4-byte value.
template <int KC, int NT, int OPT = 3>   // inputs / 32, outputs / 128
__global__ __launch_bounds__(512) void k_sb_gemm_n128(int64_t R, const float *__restrict__ X, int64_t ldx, const float *__restrict__ W, int64_t ldw,
                                                      const float *__restrict__ bias, const float *addend, int64_t lda, float *Y, int64_t ldy, int relu,
                                                      int mask_tiles, float *__restrict__ colsum_part, uint8_t *__restrict__ ybits = nullptr, int64_t ldyb = 0) {
    extern __shared__ uint4 sbg_tile[];                 // [buffer][piece][chunk][row half][lane]
    constexpr int IMG = 3 * KC * 2 * 64, UPW = KC / 4;  // uint4 per image; (chunk, half) blocks staged per wave and iteration
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, i = l & 15, gq = l >> 4;
    uint4 wg[NT][KC][3];                                // output tiles w, w + 8, .. (16 outputs each)
    float4 b4[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float *rw = W + (size_t)(16 * (w + 8 * t) + i) * ldw + 8 * gq;
#pragma unroll
        for (int c = 0; c < KC; c++) sb_split8(*(const float4 *)(rw + 32 * c), *(const float4 *)(rw + 32 * c + 4), wg[t][c]);
        b4[t] = !(OPT & 4) && bias ? *(const float4 *)(bias + 16 * (w + 8 * t) + 4 * gq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int64_t n_it = (R + 31) / 32;
    float4 pf[UPW][2];
    auto fetch = [&](int64_t it) {
#pragma unroll
        for (int n = 0; n < UPW; n++) {
            const int blk = w * UPW + n, c = blk >> 1, rt = blk & 1;      // this wave's n-th (chunk, half) block
            const int64_t row = it * 32 + rt * 16 + i;
            pf[n][0] = pf[n][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < R) {
                const float *src = X + row * ldx + 32 * c + 8 * gq;
                if constexpr (SBG_ABLATE & 4) {
                    pf[n][0] = pf[n][1] = make_float4((float)row * 1e-6f, 0.5f, -0.25f, (float)c);
                } else {
                    pf[n][0] = *(const float4 *)src;
                    pf[n][1] = *(const float4 *)(src + 4);
                }
            }
        }
    };
    auto stage = [&](uint4 *img) {
#pragma unroll
        for (int n = 0; n < UPW; n++) {
            const int blk = w * UPW + n;
            uint4 p_[3];
            sb_split8(pf[n][0], pf[n][1], p_);
#pragma unroll
            for (int p = 0; p < 3; p++) img[(p * KC * 2 + blk) * 64 + l] = p_[p];
        }
    };
    int64_t it = blockIdx.x;
    if (it < n_it) { fetch(it); stage(sbg_tile); }
    if (it + gridDim.x < n_it) fetch(it + gridDim.x);
    lds_barrier();
    int cur = 0;
    float4 cs[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) cs[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (; it < n_it; it += gridDim.x) {
        const uint4 *tb = sbg_tile + cur * IMG + l;
        f32x4 acc[NT][2];
        float4 a4[NT][2];
        if ((OPT & 1) && addend) {   // in flight during the matrix phase (OPT & 4: the mask; tiles beyond mask_tiles are loaded and ignored)
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const int64_t row = it * 32 + rt * 16 + i;
#pragma unroll
                for (int t = 0; t < NT; t++)
                    a4[t][rt] = row < R ? *(const float4 *)(addend + row * lda + 16 * (w + 8 * t) + 4 * gq) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        auto next_tile = [&]() {
            if (it + gridDim.x < n_it) stage(sbg_tile + (cur ^ 1) * IMG);          // the rows fetched one iteration ago -> the other image
            if (it + 2 * (int64_t)gridDim.x < n_it) fetch(it + 2 * (int64_t)gridDim.x);
        };
        if ((OPT & 2) && w < 4) next_tile();
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t][0] = acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the operands of chunk c + 1 are read while chunk c is multiplied (pinned with scheduling barriers: left alone, the compiler
        // issues a chunk's reads after the previous chunk's MFMAs and the matrix pipe drains behind every LDS round trip)
        // (the HBM-bound 128 x 128 variant is better off without: 106 against 114 us at 492 000 rows)
        constexpr bool AHEAD = KC * NT > 4;
        uint4 nb0[3], nb1[3];
        if constexpr (AHEAD) {
#pragma unroll
            for (int p = 0; p < 3; p++) { nb0[p] = tb[(p * KC * 2) * 64]; nb1[p] = tb[(p * KC * 2 + 1) * 64]; }
        }
#pragma unroll
        for (int c = 0; c < KC; c++) {
            uint4 b0[3], b1[3];
            if constexpr (AHEAD) {
#pragma unroll
                for (int p = 0; p < 3; p++) { b0[p] = nb0[p]; b1[p] = nb1[p]; }
                if (c + 1 < KC) {
#pragma unroll
                    for (int p = 0; p < 3; p++) { nb0[p] = tb[(p * KC * 2 + 2 * c + 2) * 64]; nb1[p] = tb[(p * KC * 2 + 2 * c + 3) * 64]; }
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int p = 0; p < 3; p++) { b0[p] = tb[(p * KC * 2 + 2 * c) * 64]; b1[p] = tb[(p * KC * 2 + 2 * c + 1) * 64]; }
            }
#define SBG_MMA(pi, pj)                                                                                                                             \
            _Pragma("unroll") for (int t = 0; t < NT; t++) {                                                                                          \
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wg[t][c][pi]), __builtin_bit_cast(bf16x8, b0[pj]), acc[t][0], 0, 0, 0); \
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wg[t][c][pi]), __builtin_bit_cast(bf16x8, b1[pj]), acc[t][1], 0, 0, 0); \
            }
            if constexpr (!(SBG_ABLATE & 1)) { SBG_MMA(2, 0) SBG_MMA(0, 2) SBG_MMA(1, 1) SBG_MMA(1, 0) SBG_MMA(0, 1) SBG_MMA(0, 0) }
            else {
#pragma unroll
                for (int t = 0; t < NT; t++) { acc[t][0][0] += __uint_as_float(b0[0].x ^ wg[t][c][0].x); acc[t][1][1] += __uint_as_float(b1[2].y ^ wg[t][c][2].w); }
            }
#undef SBG_MMA
            if constexpr (AHEAD) __builtin_amdgcn_sched_barrier(0);
        }
        if (!(OPT & 2) || w >= 4) next_tile();
        // D tile: lane (i, gq), register q -> output 16 (w + 8 t) + 4 gq + q of row i of the half
        unsigned mb[NT][2];
        if constexpr ((OPT & 4) && (OPT & 16)) {
            // relu' as bits: one byte per (tile, half) holds this lane's four (a nibble)
            __builtin_amdgcn_sched_barrier(0);
            const uint8_t *bits = (const uint8_t *)addend;
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const int64_t row = it * 32 + rt * 16 + i;
#pragma unroll
                for (int t = 0; t < NT; t++)
                    mb[t][rt] = row < R && t < mask_tiles ? (unsigned)bits[row * lda + 2 * (w + 8 * t) + (gq >> 1)] >> ((gq & 1) * 4) : 15u;
            }
        } else if constexpr ((OPT & 4) && !(OPT & 1)) {
            // the mask of the whole tile is requested at once, in registers the matrix phase has just released (the barrier keeps the
            // compiler from hoisting the loads into that phase, where they do not fit)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const int64_t row = it * 32 + rt * 16 + i;
#pragma unroll
                for (int t = 0; t < NT; t++)
                    a4[t][rt] = row < R && t < mask_tiles ? *(const float4 *)(addend + row * lda + 16 * (w + 8 * t) + 4 * gq) : make_float4(1.f, 1.f, 1.f, 1.f);
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
            const int64_t row = it * 32 + rt * 16 + i;
            if (row < R) {
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const int col = 16 * (w + 8 * t) + 4 * gq;
                    float4 v;
                    if constexpr (OPT & 4) {
                        v = make_float4(acc[t][rt][0], acc[t][rt][1], acc[t][rt][2], acc[t][rt][3]);
                        if constexpr (OPT & 16) {
                            const unsigned m = mb[t][rt];
                            v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f; v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
                        } else if (t < mask_tiles) {
                            v.x = a4[t][rt].x > 0.f ? v.x : 0.f; v.y = a4[t][rt].y > 0.f ? v.y : 0.f;
                            v.z = a4[t][rt].z > 0.f ? v.z : 0.f; v.w = a4[t][rt].w > 0.f ? v.w : 0.f;
                        }
                        cs[t].x += v.x; cs[t].y += v.y; cs[t].z += v.z; cs[t].w += v.w;
                    } else {
                        v = make_float4(acc[t][rt][0] + b4[t].x, acc[t][rt][1] + b4[t].y, acc[t][rt][2] + b4[t].z, acc[t][rt][3] + b4[t].w);
                        if (addend) {
                            if (!(OPT & 1)) a4[t][rt] = *(const float4 *)(addend + row * lda + col);
                            v.x += a4[t][rt].x; v.y += a4[t][rt].y; v.z += a4[t][rt].z; v.w += a4[t][rt].w;
                        }
                        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        if constexpr (OPT & 8) {   // lanes (i, gq) and (i, gq ^ 1) hold the two nibbles of a byte (same row: both are active here)
                            const unsigned nib = (unsigned)(v.x > 0.f) | ((unsigned)(v.y > 0.f) << 1) | ((unsigned)(v.z > 0.f) << 2) | ((unsigned)(v.w > 0.f) << 3);
                            const unsigned other = (unsigned)__shfl_xor((int)nib, 16);
                            if (!(gq & 1)) ybits[row * ldyb + 2 * (w + 8 * t) + (gq >> 1)] = (uint8_t)(nib | (other << 4));
                        }
                    }
                    if constexpr (SBG_ABLATE & 2) { if (v.x == 12345.678f) *(float4 *)(Y + row * ldy + col) = v; }
                    else *(float4 *)(Y + row * ldy + col) = v;
                }
            }
        }
        lds_barrier();
        cur ^= 1;
    }
    if constexpr (OPT & 4) {   // this workgroup's column sums: the 16 row lanes of a column group fold, lane i = 0 stores
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                cs[t].x += __shfl_xor(cs[t].x, m); cs[t].y += __shfl_xor(cs[t].y, m);
                cs[t].z += __shfl_xor(cs[t].z, m); cs[t].w += __shfl_xor(cs[t].w, m);
            }
            if (i == 0) *(float4 *)(colsum_part + (size_t)blockIdx.x * (128 * NT) + 16 * (w + 8 * t) + 4 * gq) = cs[t];
        }
    }
}

// out[f] = sum over the workgroups' partial rows, in order (one thread per column, four independent chains)
__global__ void k_sb_colsum_reduce(int n_parts, int F, const float *__restrict__ parts, float *__restrict__ out) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int p = 0;
    for (; p + 4 <= n_parts; p += 4) {
        a0 += parts[(size_t)p * F + f]; a1 += parts[(size_t)(p + 1) * F + f];
        a2 += parts[(size_t)(p + 2) * F + f]; a3 += parts[(size_t)(p + 3) * F + f];
    }
    for (; p < n_parts; p++) a0 += parts[(size_t)p * F + f];
    out[f] = (a0 + a1) + (a2 + a3);
}

static int g_sbg_wgs_per_cu = 0;   // lab override (tools/microbench/sb_gemm_lab.hip); 0 = the default below

template <int KC, int NT, int OPT>
int launch_sb_gemm(int64_t R, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, const float *addend, int64_t lda,
                   float *Y, int64_t ldy, int relu, hipStream_t st, int mask_tiles = 0, float *colsum_part = nullptr, float *colsum = nullptr,
                   uint8_t *ybits = nullptr, int64_t ldyb = 0) {
    constexpr int lds = 2 * 3 * KC * 2 * 64 * 16;
    static std::once_flag once;   // the evaluator's thread may launch concurrently with the trainer's
    static hipError_t attr_rc = hipSuccess;
    std::call_once(once, [] { attr_rc = hipFuncSetAttribute((const void *)k_sb_gemm_n128<KC, NT, OPT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
    if (attr_rc != hipSuccess) return (int)attr_rc;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int64_t n_it = (R + 31) / 32;
    // persistent workgroups per CU: the 128 x 128 variant (114-119 registers) fits two, whose phases interleave -- from 131 072 rows up
    // that pays (492 000 rows: 133 -> 121 us plain, 177 -> 152 us in place; 1 476 000: 361 -> 341 / 567 -> 490); below, one
    const int wpc = g_sbg_wgs_per_cu ? g_sbg_wgs_per_cu : (KC * NT == 4 && R >= 131072 ? 2 : 1);
    const int grid = n_it < (int64_t)cus * wpc ? (int)n_it : cus * wpc;
    hipLaunchKernelGGL((k_sb_gemm_n128<KC, NT, OPT>), dim3(grid), dim3(512), lds, st, R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, mask_tiles,
                       colsum_part, ybits, ldyb);
    if constexpr (OPT & 4) hipLaunchKernelGGL(k_sb_colsum_reduce, dim3((128 * NT + 127) / 128), dim3(128), 0, st, grid, 128 * NT, colsum_part, colsum);
    return (int)hipGetLastError();
}

// Which OPT per shape, plain / with an addend (tools/microbench/sb_gemm_lab.hip, 492 000 rows, us; profiles/r04_sb_gemm_lab.txt):
//   128 <- 384: 366 / 416 (OPT 0) -> 298 / 391 (OPT 2);   128 <- 256: 253 / 315 -> 213 (OPT 2) / 277 (OPT 3);   128 <- 128: 123 (OPT 0) /
//   210 -> 180 (OPT 1);   256 <- 128: 263 / 363 -> 223 (OPT 2) / 311 (OPT 1);   384 <- 128: 363 / 516 -> 310 (OPT 2) / 516 (OPT 0).
// The early addend request costs registers the 384-input and 384-output variants do not have (it spills there).
// the masked (ReLU-backward) forms: see tools/microbench/sb_gemm_lab.hip
// (492 000 rows, us, OPT 4 / 5 / 6 / 7: 256 <- 128: 329 / 316 / 307 / 327;  384 <- 128: 552 / 560 / 573 / 574;  128 <- 384: 450 / 467 / 423 / 488)
constexpr int SBG_MASK_OPT_256 = 6, SBG_MASK_OPT_384 = 4, SBG_MASK_OPT_K384 = 6;

template <int KC, int NT>
int launch_sb_gemm_best(int64_t R, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, const float *addend, int64_t lda,
                        float *Y, int64_t ldy, int relu, hipStream_t st, uint8_t *ybits = nullptr, int64_t ldyb = 0) {
    constexpr int PLAIN = (KC == 4 && NT == 1) ? 0 : 2;
    constexpr int ADD = KC == 12 ? 2 : (KC == 8 ? 3 : (NT == 3 ? 0 : 1));
    if constexpr (NT == 1 && KC <= 8) if (ybits) {      // + the sign bits of Y (sb_gemm_signs): the same schedule, the bytes written beside the result
        if (addend) return launch_sb_gemm<KC, NT, ADD | 8>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st, 0, nullptr, nullptr, ybits, ldyb);
        if constexpr (KC == 4 && NT == 1) if (R >= 786432) return launch_sb_gemm<4, 1, 10>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st, 0, nullptr, nullptr, ybits, ldyb);
        return launch_sb_gemm<KC, NT, PLAIN | 8>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st, 0, nullptr, nullptr, ybits, ldyb);
    }
    if (addend) return launch_sb_gemm<KC, NT, ADD>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st);
    // 128 x 128 beyond the Infinity Cache (the update's AGG layer over three relations, 1 476 000 rows): the phase shift wins there
    // (405 -> 341 us; at 492 000 rows 121 against 124)
    if constexpr (KC == 4 && NT == 1) if (R >= 786432) return launch_sb_gemm<4, 1, 2>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st);
    return launch_sb_gemm<KC, NT, PLAIN>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st);
}
