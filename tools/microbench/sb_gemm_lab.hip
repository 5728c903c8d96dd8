// sb_gemm_lab.hip -- lab of the split-bf16 Linear GEMM Y = act(X W^T + b [+ C]) with 128 outputs and 256 / 384 inputs (the product
// kernels live in csrc/sb_gemm.hpp; this file times them beside ablated copies and checks the results against f64 on the host).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I distributed_multi_agent_reinforcement_learning_amd/csrc -I include \
//         tools/microbench/sb_gemm_lab.hip -o tools/microbench/sb_gemm_lab
//   ./tools/microbench/sb_gemm_lab [rows ...]
// Results: profiles/r04_sb_gemm_lab.txt.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mappo_ops.h"
#include "sb_gemm.hpp"

// ---- a variant that did NOT pay: 128 outputs on 256 / 384 inputs with the contraction SPLIT OVER TWO WAVE GROUPS (k_sb_gemm_ks2) ----
// (393 us against k_sb_gemm_n128's 366 / 298 us at 492 000 rows x 384 inputs; its ablations say what bounds all these kernels: matrix
// phase alone 182 us, loads + stores + staging alone 204 us -- they do not overlap fully)
// k_sb_gemm_n128 gives every wave 16 outputs and the whole contraction: all eight waves read the whole staged X image (590 KB of
// ds_read_b128 per 32 rows at 384 inputs, 72 reads feeding 144 MFMAs per wave).  Here wave (og, ks) owns 32 outputs (two 16-output
// tiles) and HALF of the inputs: the same 144 weight registers, half the operand reads (each feeds two tiles), and the partial sums
// of the upper half cross to the lower half's waves through LDS -- written into the image that has just been consumed (the part the
// READING waves re-stage themselves, so program order protects it): two barriers per 32 rows, no extra LDS.  (One accumulator per
// tile: with the weights at 144 registers there is no room for sb_mma6_hl's second set; tests/test_split_bf16_gpu.py.)  ABL (lab builds, tools/microbench/sb_gemm_lab.hip): 1 no MFMAs, 2 no split arithmetic, 4 no global
// loads, 8 no stores.
template <int KC, int ABL = 0>   // inputs / 32 (8 or 12)
__global__ __launch_bounds__(512) void k_sb_gemm_ks2(int64_t R, const float *__restrict__ X, int64_t ldx, const float *__restrict__ W, int64_t ldw,
                                                     const float *__restrict__ bias, const float *addend, int64_t lda, float *Y, int64_t ldy, int relu) {
    extern __shared__ uint4 sbk_tile[];                 // [buffer][piece][(chunk, row half)][lane]
    constexpr int IMG = 3 * KC * 2 * 64, UPW = KC / 4, KH = KC / 2;
    static_assert(KC % 4 == 0 && 16 <= 4 * UPW * 3, "the partial sums must fit the blocks the lower wave group stages");
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, i = l & 15, gq = l >> 4;
    const int ks = w >> 2, og = w & 3;
    uint4 wg[2][KH][3];                                 // outputs 32 og + 16 t + i, inputs 32 (ks KH + c) + 8 gq ..
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const float *rw = W + (size_t)(32 * og + 16 * t + i) * ldw + 32 * ks * KH + 8 * gq;
#pragma unroll
        for (int c = 0; c < KH; c++) sb_split8(*(const float4 *)(rw + 32 * c), *(const float4 *)(rw + 32 * c + 4), wg[t][c]);
    }
    const int64_t n_it = (R + 31) / 32;
    float4 pf[UPW][2];
    auto fetch = [&](int64_t it) {
#pragma unroll
        for (int n = 0; n < UPW; n++) {
            const int blk = w * UPW + n, c = blk >> 1, rt = blk & 1;      // this wave's n-th (chunk, half) block
            const int64_t row = it * 32 + rt * 16 + i;
            pf[n][0] = pf[n][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(ABL & 4) && row < R) {
                const float *src = X + row * ldx + 32 * c + 8 * gq;
                pf[n][0] = *(const float4 *)src;
                pf[n][1] = *(const float4 *)(src + 4);
            }
        }
    };
    auto stage = [&](uint4 *img) {
#pragma unroll
        for (int n = 0; n < UPW; n++) {
            const int blk = w * UPW + n;
            uint4 p_[3];
            if constexpr (ABL & 2) {
                p_[0] = __builtin_bit_cast(uint4, pf[n][0]); p_[1] = __builtin_bit_cast(uint4, pf[n][1]); p_[2] = p_[0];
            } else {
                sb_split8(pf[n][0], pf[n][1], p_);
            }
#pragma unroll
            for (int p = 0; p < 3; p++) img[(p * KC * 2 + blk) * 64 + l] = p_[p];
        }
    };
    // where partial sum q (one f32x4 per lane) of output group og lives inside a consumed image: 1 KB blocks staged by waves 0 .. 3
    auto xch = [&](uint4 *img, int q) -> float4 * {
        const int k = og * 4 + q;
        return (float4 *)(img + ((k / KC) * KC * 2 + (k % KC)) * 64 + l);
    };
    int64_t it = blockIdx.x;
    if (it < n_it) { fetch(it); stage(sbk_tile); }
    if (it + gridDim.x < n_it) fetch(it + gridDim.x);
    lds_barrier();
    int cur = 0;
    for (; it < n_it; it += gridDim.x) {
        uint4 *img = sbk_tile + cur * IMG;
        const uint4 *tb = img + l;
        f32x4 hi[2][2];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int rt = 0; rt < 2; rt++) hi[t][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the first row half's operands of chunk c + 1 are read while chunk c is multiplied, the second half's while the first half
        // is (a full look-ahead of both halves needs 24 more registers than the 144 of the weights leave at 384 inputs)
        uint4 nb0[3];
#pragma unroll
        for (int p = 0; p < 3; p++) nb0[p] = tb[(p * KC * 2 + 2 * ks * KH) * 64];
#pragma unroll
        for (int c = 0; c < KH; c++) {
            const int cc = ks * KH + c;
            uint4 b0[3], b1[3];
#pragma unroll
            for (int p = 0; p < 3; p++) { b0[p] = nb0[p]; b1[p] = tb[(p * KC * 2 + 2 * cc + 1) * 64]; }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(ABL & 1)) {
#pragma unroll
                for (int t = 0; t < 2; t++) hi[t][0] = sb_mma6(wg[t][c], b0, hi[t][0]);
                if (c + 1 < KH) {
#pragma unroll
                    for (int p = 0; p < 3; p++) nb0[p] = tb[(p * KC * 2 + 2 * cc + 2) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 2; t++) hi[t][1] = sb_mma6(wg[t][c], b1, hi[t][1]);
            } else {
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    hi[t][0][0] += __builtin_bit_cast(float, b0[0].x ^ b0[1].y ^ b0[2].z ^ wg[t][c][0].x);
                    hi[t][1][0] += __builtin_bit_cast(float, b1[0].x ^ b1[1].y ^ b1[2].z ^ wg[t][c][1].x);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (it + gridDim.x < n_it) stage(sbk_tile + (cur ^ 1) * IMG);          // the rows fetched one iteration ago -> the other image
        if (it + 2 * (int64_t)gridDim.x < n_it) fetch(it + 2 * (int64_t)gridDim.x);
        lds_barrier();                                   // A: this image is consumed, the other one is staged
        float4 a4[2][2];
        if (ks == 1) {
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int rt = 0; rt < 2; rt++) *xch(img, 2 * t + rt) = (float4){hi[t][rt][0], hi[t][rt][1], hi[t][rt][2], hi[t][rt][3]};
        } else if (addend) {   // the addend rows (ks = 0 waves own the epilogue): in flight across the second barrier
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const int64_t row = it * 32 + rt * 16 + i;
#pragma unroll
                for (int t = 0; t < 2; t++)
                    a4[rt][t] = row < R ? *(const float4 *)(addend + row * lda + 32 * og + 16 * t + 4 * gq) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        lds_barrier();                                   // B: the upper half's partial sums are in the consumed image
        if (ks == 0) {
            // D tile: lane (i, gq), register q -> output 32 og + 16 t + 4 gq + q of row i of the half
            float4 b4[2];   // (re-read per tile: 8 registers the matrix phase needs more)
#pragma unroll
            for (int t = 0; t < 2; t++) b4[t] = bias ? *(const float4 *)(bias + 32 * og + 16 * t + 4 * gq) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                const int64_t row = it * 32 + rt * 16 + i;
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const float4 o4 = *xch(img, 2 * t + rt);
                    const int col = 32 * og + 16 * t + 4 * gq;
                    float4 v = make_float4(hi[t][rt][0] + o4.x + b4[t].x, hi[t][rt][1] + o4.y + b4[t].y, hi[t][rt][2] + o4.z + b4[t].z,
                                           hi[t][rt][3] + o4.w + b4[t].w);
                    if (addend) { v.x += a4[rt][t].x; v.y += a4[rt][t].y; v.z += a4[rt][t].z; v.w += a4[rt][t].w; }
                    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (!(ABL & 8) && row < R) *(float4 *)(Y + row * ldy + col) = v;
                }
            }
        }
        cur ^= 1;
    }
}

template <int KC, int ABL = 0>
int launch_sb_gemm_ks2(int64_t R, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, const float *addend, int64_t lda,
                       float *Y, int64_t ldy, int relu, hipStream_t st) {
    constexpr int lds = 2 * 3 * KC * 2 * 64 * 16;
    static std::once_flag once;
    static hipError_t attr_rc = hipSuccess;
    std::call_once(once, [] { attr_rc = hipFuncSetAttribute((const void *)k_sb_gemm_ks2<KC, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
    if (attr_rc != hipSuccess) return (int)attr_rc;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int64_t n_it = (R + 31) / 32;
    const int grid = n_it < cus ? (int)n_it : cus;
    hipLaunchKernelGGL((k_sb_gemm_ks2<KC, ABL>), dim3(grid), dim3(512), lds, st, R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu);
    return (int)hipGetLastError();
}

// ---- the variants under test ------------------------------------------------------------------------------------------------------
#define SBG_LAB_VARIANTS 9
static const char *sbg_lab_name(int v) {
    static const char *n[] = {"n128 OPT=0 (round 3)", "n128 OPT=1 addend ahead", "n128 OPT=2 phase shift", "n128 OPT=3 both", "k_sb_gemm_ks2", "ks2, no MFMAs",
                              "ks2, no global loads", "ks2, no stores", "ks2, no loads, no stores"};
    return n[v];
}
static bool sbg_lab_checks(int v) { return v <= 4; }
static int sbg_lab_launch(int v, int64_t R, int N, int K, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *b, const float *add,
                          int64_t lda, float *Y, int64_t ldy, int relu, hipStream_t st) {
#define LAB_K(KC)                                                                                                   \
    switch (v) {                                                                                                    \
        case 0: return launch_sb_gemm<KC, 1, 0>(R, X, ldx, W, ldw, b, add, lda, Y, ldy, relu, st);                  \
        case 1: return launch_sb_gemm<KC, 1, 1>(R, X, ldx, W, ldw, b, add, lda, Y, ldy, relu, st);                  \
        case 2: return launch_sb_gemm<KC, 1, 2>(R, X, ldx, W, ldw, b, add, lda, Y, ldy, relu, st);                  \
        case 3: return launch_sb_gemm<KC, 1, 3>(R, X, ldx, W, ldw, b, add, lda, Y, ldy, relu, st);                  \
        case 4: return launch_sb_gemm_ks2<KC, 0>(R, X, ldx, W, ldw, b, add, lda, Y, ldy, relu, st);                 \
        case 5: return launch_sb_gemm_ks2<KC, 1>(R, X, ldx, W, ldw, b, add, lda, Y, ldy, relu, st);                 \
        case 6: return launch_sb_gemm_ks2<KC, 4>(R, X, ldx, W, ldw, b, add, lda, Y, ldy, relu, st);                 \
        case 7: return launch_sb_gemm_ks2<KC, 8>(R, X, ldx, W, ldw, b, add, lda, Y, ldy, relu, st);                 \
        case 8: return launch_sb_gemm_ks2<KC, 12>(R, X, ldx, W, ldw, b, add, lda, Y, ldy, relu, st);                \
    }
    if (N == 128 && K == 384) { LAB_K(12) }
    if (N == 128 && K == 256) { LAB_K(8) }
#undef LAB_K
#define LAB_N(NT)                                                                                                   \
    switch (v) {                                                                                                    \
        case 0: return launch_sb_gemm<4, NT, 0>(R, X, ldx, W, ldw, b, add, lda, Y, ldy, relu, st);                  \
        case 1: return launch_sb_gemm<4, NT, 1>(R, X, ldx, W, ldw, b, add, lda, Y, ldy, relu, st);                  \
        case 2: return launch_sb_gemm<4, NT, 2>(R, X, ldx, W, ldw, b, add, lda, Y, ldy, relu, st);                  \
        case 3: return launch_sb_gemm<4, NT, 3>(R, X, ldx, W, ldw, b, add, lda, Y, ldy, relu, st);                  \
    }
    if (K == 128 && N == 128) { LAB_N(1) }
    if (K == 128 && N == 256) { LAB_N(2) }
    if (K == 128 && N == 384) { LAB_N(3) }
#undef LAB_N
    return MO_ERR_BAD_ARG;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename F>
static float time_us(F f, int reps = 20) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / reps;
}

int main(int argc, char **argv) {
    std::vector<int64_t> sizes;
    for (int i = 1; i < argc; i++) sizes.push_back(atoll(argv[i]));
    if (sizes.empty()) sizes = {65536, 196608, 492000};
    if (getenv("SBG_WPC")) g_sbg_wgs_per_cu = atoi(getenv("SBG_WPC"));   // persistent workgroups per CU (register budget permitting)
    const int shapes[][2] = {{128, 384}, {128, 256}, {128, 128}, {256, 128}, {384, 128}};
    for (auto &nk : shapes) {
        const int N = nk[0], K = nk[1];
        for (int64_t R : sizes) {
            std::vector<float> hx((size_t)R * K), hw((size_t)N * K), hb(N), hy((size_t)R * N);
            srand(7);
            for (auto &v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
            for (auto &v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
            for (auto &v : hb) v = (rand() / (float)RAND_MAX - 0.5f);
            float *X, *W, *B, *Y, *C;
            CK(hipMalloc(&X, hx.size() * 4)); CK(hipMalloc(&W, hw.size() * 4)); CK(hipMalloc(&B, N * 4)); CK(hipMalloc(&Y, hy.size() * 4));
            CK(hipMalloc(&C, hy.size() * 4));
            CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(B, hb.data(), N * 4, hipMemcpyHostToDevice));
            CK(hipMemset(C, 0, hy.size() * 4));
            auto check = [&](const char *what) {
                CK(hipMemcpy(hy.data(), Y, hy.size() * 4, hipMemcpyDeviceToHost));
                double worst = 0.0;
                for (int s = 0; s < 97; s++) {
                    const int64_t r = (s * 7919 + (s == 96 ? R - 1 : 0)) % R;
                    for (int n = 0; n < N; n++) {
                        double ref = hb[n];
                        for (int k = 0; k < K; k++) ref += (double)hx[(size_t)r * K + k] * (double)hw[(size_t)n * K + k];
                        ref = ref > 0.0 ? ref : 0.0;
                        worst = fmax(worst, fabs(ref - (double)hy[(size_t)r * N + n]));
                    }
                }
                printf("    %-28s max |err| vs f64 on 97 rows: %.3g\n", what, worst);
            };
            const double bytes = (double)R * (K + N) * 4.0, flop = 2.0 * R * K * N;
            // the OPT variants execute the same arithmetic in the same order: their results must be bit-identical
            {
                std::vector<float> y0(hy.size()), y1(hy.size());
                for (int mode = 0; mode < 2; mode++) {        // plain (bias + ReLU), then in place Y += X W^T
                    for (int variant = 0; variant < 4; variant++) {
                        for (size_t q = 0; q < hy.size(); q++) hy[q] = (float)((q * 2654435761u) % 1000) * 1e-3f;
                        CK(hipMemcpy(Y, hy.data(), hy.size() * 4, hipMemcpyHostToDevice));
                        int rc = mode == 0 ? sbg_lab_launch(variant, R, N, K, X, K, W, K, B, nullptr, 0, Y, N, 1, 0)
                                           : sbg_lab_launch(variant, R, N, K, X, K, W, K, nullptr, Y, N, Y, N, 0, 0);
                        if (rc) continue;
                        CK(hipDeviceSynchronize());
                        CK(hipMemcpy((variant == 0 ? y0 : y1).data(), Y, hy.size() * 4, hipMemcpyDeviceToHost));
                        if (variant > 0) {
                            size_t bad = 0, first = 0;
                            for (size_t q = 0; q < hy.size(); q++) if (memcmp(&y0[q], &y1[q], 4)) { if (!bad) first = q; bad++; }
                            printf("N=%d K=%d rows=%lld %s OPT=%d vs OPT=0: %zu differing elements%s\n", N, K, (long long)R, mode ? "in place" : "plain", variant, bad,
                                   bad ? " <-- HAZARD" : "");
                            if (bad) printf("    first at row %zu col %zu: %.9g vs %.9g\n", first / N, first % N, y0[first], y1[first]);
                        }
                    }
                }
            }
            for (int variant = 0; variant < SBG_LAB_VARIANTS; variant++) {
                CK(hipMemset(Y, 0, hy.size() * 4));
                int rc = sbg_lab_launch(variant, R, N, K, X, K, W, K, B, nullptr, 0, Y, N, 1, 0);
                if (rc == MO_ERR_BAD_ARG) continue;
                CK(hipDeviceSynchronize());
                if (rc) { printf("variant %d: launch error %d\n", variant, rc); continue; }
                const float us = time_us([&] { sbg_lab_launch(variant, R, N, K, X, K, W, K, B, nullptr, 0, Y, N, 1, 0); });
                if (sbg_lab_checks(variant)) check(sbg_lab_name(variant));
                // beta = 1 in place (the rollout's semantic layer): Y += X W^T, no bias, no ReLU
                const float us_add = time_us([&] { sbg_lab_launch(variant, R, N, K, X, K, W, K, nullptr, Y, N, Y, N, 0, 0); });
                printf("N=%d K=%d rows=%lld  %-26s %8.1f us  %5.2f TB/s %6.1f TF | in place += : %8.1f us %5.2f TB/s\n", N, K, (long long)R, sbg_lab_name(variant), us,
                       bytes / us * 1e-6, flop / us * 1e-6, us_add, (bytes + (double)R * N * 4.0) / us_add * 1e-6);
            }
            // the ReLU-backward form (OPT bit 4): Y = (X W^T) * (M > 0) on the first mask_cols columns, column sums from per-workgroup partials
            if ((N == 256 && K == 128) || (N == 384 && K == 128) || (N == 128 && K == 384)) {
                std::vector<float> hm(hy.size());
                for (size_t q = 0; q < hm.size(); q++) hm[q] = ((q * 2246822519u) >> 7) % 3 ? (float)(q % 17 + 1) : 0.f;   // a third of the units inactive
                CK(hipMemcpy(C, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
                float *part, *cs;
                CK(hipMalloc(&part, 1024 * N * 4)); CK(hipMalloc(&cs, N * 4));
                const int mcols = N == 256 ? 128 : N;          // first FCRA hop: only the aggregate half is behind a ReLU
                std::vector<float> y0(hy.size());
                for (int opt = 4; opt < 8; opt++) {
                    auto run = [&]() -> int {
#define LAB_M(KC, NT)                                                                                                                  \
                        switch (opt) {                                                                                                \
                            case 4: return launch_sb_gemm<KC, NT, 4>(R, X, K, W, K, nullptr, C, N, Y, N, 0, 0, mcols / 128, part, cs); \
                            case 5: return launch_sb_gemm<KC, NT, 5>(R, X, K, W, K, nullptr, C, N, Y, N, 0, 0, mcols / 128, part, cs); \
                            case 6: return launch_sb_gemm<KC, NT, 6>(R, X, K, W, K, nullptr, C, N, Y, N, 0, 0, mcols / 128, part, cs); \
                            case 7: return launch_sb_gemm<KC, NT, 7>(R, X, K, W, K, nullptr, C, N, Y, N, 0, 0, mcols / 128, part, cs); \
                        }
                        if (N == 256) { LAB_M(4, 2) }
                        if (N == 384) { LAB_M(4, 3) }
                        if (N == 128) { LAB_M(12, 1) }
#undef LAB_M
                        return -1;
                    };
                    CK(hipMemset(Y, 0, hy.size() * 4));
                    if (run()) { printf("masked OPT=%d: launch error\n", opt); continue; }
                    CK(hipDeviceSynchronize());
                    const float us = time_us(run);
                    CK(hipMemcpy(hy.data(), Y, hy.size() * 4, hipMemcpyDeviceToHost));
                    std::vector<float> hcs(N);
                    CK(hipMemcpy(hcs.data(), cs, N * 4, hipMemcpyDeviceToHost));
                    double worst = 0.0, worst_cs = 0.0, cs_scale = 0.0;
                    for (int s2 = 0; s2 < 97; s2++) {
                        const int64_t r = (s2 * 7919 + (s2 == 96 ? R - 1 : 0)) % R;
                        for (int n = 0; n < N; n++) {
                            double ref = 0.0;
                            for (int k = 0; k < K; k++) ref += (double)hx[(size_t)r * K + k] * (double)hw[(size_t)n * K + k];
                            if (n < mcols && !(hm[(size_t)r * N + n] > 0.f)) ref = 0.0;
                            worst = fmax(worst, fabs(ref - (double)hy[(size_t)r * N + n]));
                        }
                    }
                    std::vector<double> col(N, 0.0), cabs(N, 0.0);
                    for (int64_t r = 0; r < R; r++)
                        for (int n = 0; n < N; n++) { col[n] += hy[(size_t)r * N + n]; cabs[n] += fabs(hy[(size_t)r * N + n]); }
                    for (int n = 0; n < N; n++) { worst_cs = fmax(worst_cs, fabs(col[n] - (double)hcs[n])); cs_scale = fmax(cs_scale, cabs[n]); }
                    size_t bad = 0;
                    if (opt == 4) y0 = hy; else for (size_t q = 0; q < hy.size(); q++) bad += memcmp(&y0[q], &hy[q], 4) != 0;
                    printf("N=%d K=%d rows=%lld  masked OPT=%d (mask on %d columns) %8.1f us  %5.2f TB/s | max |err| vs f64 %.3g, column sums off by %.3g of %.3g (sum |.|), "
                           "%zu elements differ from OPT=4\n", N, K, (long long)R, opt, mcols, us, ((double)R * (K + 2 * N) * 4.0) / us * 1e-6, worst, worst_cs, cs_scale, bad);
                }
                CK(hipFree(part)); CK(hipFree(cs));
            }
            // the ReLU layers' variant that also writes relu'(Y) as sign bits (OPT bit 8) beside the plain kernel
            if (N == 128 && (K == 128 || K == 256)) {
                uint8_t *bits;
                CK(hipMalloc(&bits, (size_t)R * 16));
                auto plain = [&] { return K == 128 ? (R >= 786432 ? launch_sb_gemm<4, 1, 2>(R, X, K, W, K, B, nullptr, 0, Y, N, 1, 0) : launch_sb_gemm<4, 1, 0>(R, X, K, W, K, B, nullptr, 0, Y, N, 1, 0))
                                                   : launch_sb_gemm<8, 1, 2>(R, X, K, W, K, B, nullptr, 0, Y, N, 1, 0); };
                auto signs = [&] { return K == 128 ? (R >= 786432 ? launch_sb_gemm<4, 1, 10>(R, X, K, W, K, B, nullptr, 0, Y, N, 1, 0, 0, nullptr, nullptr, bits, 16)
                                                                   : launch_sb_gemm<4, 1, 8>(R, X, K, W, K, B, nullptr, 0, Y, N, 1, 0, 0, nullptr, nullptr, bits, 16))
                                                   : launch_sb_gemm<8, 1, 10>(R, X, K, W, K, B, nullptr, 0, Y, N, 1, 0, 0, nullptr, nullptr, bits, 16); };
                const float us_p = time_us(plain), us_s = time_us(signs);
                printf("N=%d K=%d rows=%lld  relu layer: plain %8.1f us, with sign bits of the result %8.1f us\n", N, K, (long long)R, us_p, us_s);
                CK(hipFree(bits));
            }
            // in place accumulate (beta = 1) through the product entry point
            CK(hipFree(X)); CK(hipFree(W)); CK(hipFree(B)); CK(hipFree(Y)); CK(hipFree(C));
        }
    }
    return 0;
}
