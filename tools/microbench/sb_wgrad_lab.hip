// sb_wgrad_lab.hip -- phase / clock lab of the weight-gradient GEMM in fp32 arithmetic on the bf16 matrix pipe (the product kernel is
// k_sb_wgrad in csrc/mappo_ops.hip, same structure; this copy carries the probes).  Results (profiles/r03_split_bf16_lab.txt): as
// accurate as the fp32-MFMA kernel (7.8e-7 against 5.9e-7 of max |C| for 492 000 rows; the BLAS library's fp32 GEMM: 6.7e-6), 414 us
// against 482 us alone.  The first A/B inside the training iteration showed no gain (319.3 against 317.9 ms) because this file's
// partial reduction (one thread per output walking 256 partials) cost the 40 us the kernel had won; with the product's four-wave
// reduction the gain is 5.5 ms per iteration.  The clock figures stand: with HBM, VALU, LDS and the matrix pipe all busy the shader
// clock falls to 1.6 GHz (2.2 GHz for the matrix phase alone, 2.28 GHz on half the CUs): the chip is power-limited.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DSB_PROBE=n] [-DSB_WGS=n] [-DSB_CLOCK] tools/microbench/sb_wgrad_lab.hip -o tools/microbench/sb_wgrad_lab_n
//   SB_PROBE: 1 no split arithmetic, 2 no LDS image, 3 no matrix phase, 4 no loads and no staging, 5 matrix pipe only
//
// Every fp32 operand is split EXACTLY into three bf16 pieces (x = x1 + x2 + x3) when it is staged, and a product of two operands
// is the six piece products with i + j <= 4 on v_mfma_f32_16x16x32_bf16 (fp32 accumulate): 6 x 16 cycles per 32 contraction
// steps per 16 x 16 tile against 8 x 32 cycles of v_mfma_f32_16x16x4_f32.  What bounds these kernels afterwards is HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SB_ERR_BAD_ARG (-2)

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

// (lo, hi) -> two bf16 in one dword, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
    const v2f x = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf2));
}
__device__ __forceinline__ float bf_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// x = p1 + p2 + p3 exactly, for two values at once (they become two consecutive contraction steps of one operand dword):
// p1 = bf16(x); x - p1 has <= 16 significant bits and is exact in fp32; p2 = bf16(x - p1); the rest has <= 8 bits and IS a bf16.
__device__ __forceinline__ void split3(float x0, float x1, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    p1 = pk_bf16(x0, x1);
    const float r0 = x0 - bf_lo(p1), r1 = x1 - bf_hi(p1);
    p2 = pk_bf16(r0, r1);
    p3 = pk_bf16(r0 - bf_lo(p2), r1 - bf_hi(p2));
}

// the six piece products of one 32 x 32 x 16 tile step, smallest first
__device__ __forceinline__ v16f mma6(const uint4 (&a)[3], const uint4 (&b)[3], v16f c) {
#define SB_MMA(i, j) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[i]), __builtin_bit_cast(bf8, b[j]), c, 0, 0, 0);
    SB_MMA(2, 0) SB_MMA(0, 2) SB_MMA(1, 1) SB_MMA(1, 0) SB_MMA(0, 1) SB_MMA(0, 0)
#undef SB_MMA
    return c;
}

// waves exchange data through LDS only: wait for this wave's LDS operations, not for its global loads in flight
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- weight gradient  C[M][N] = A^T B  (A [K][M], B [K][N], K ~ 5e5 rows) ------------------------------------------------
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
struct WgCfg {
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
#ifndef SB_PROBE
#define SB_PROBE 0
#endif

#ifdef SB_CLOCK
__device__ uint64_t sb_clock[2];
#endif
template <int MT, int NT>
__global__ __launch_bounds__(512) void k_sb_wgrad(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb, int64_t K,
                                                  float *__restrict__ part) {
    using C = WgCfg<MT, NT>;
    extern __shared__ uint4 sb_lds[];                    // [buffer][piece][octet][feature slot] x 16 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, g = lane >> 5;
    // loader role: thread -> (octet, half, feature quad)
    const bool on = tid < C::UNITS;
    const int rest = on ? tid / C::FQ : 0, col = 4 * (tid % C::FQ);
    const int row_in_chunk = 4 * rest;                   // = 8 octet + 4 half
    const float *src = (col < C::M ? A + col : B + (col - C::M)) + row_in_chunk * (col < C::M ? lda : ldb);
    const int64_t ld = col < C::M ? lda : ldb;
    const int slot2 = 2 * ((rest >> 1) * C::COLS + col) + (rest & 1), sx = (col >> 3) & 3;   // in 8-byte units
    // multiplier role
    const int wm = wave / C::WGN, wn = wave % C::WGN;
    const int64_t chunks = K / C::CR;                    // full chunks; the K % 16 tail rows are the last workgroup's epilogue
    const int64_t c_beg = chunks * blockIdx.x / gridDim.x, c_end = chunks * (blockIdx.x + 1) / gridDim.x;

#ifdef SB_CLOCK
    const uint64_t clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    v16f acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; a++)
#pragma unroll
        for (int b = 0; b < C::TN; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    float4 raw[SB_DEPTH][4];
    auto fetch = [&](float4 (&r)[4], int64_t c) {
        if (!on || c >= c_end) return;
        if constexpr (SB_PROBE == 4 || SB_PROBE == 5) {
#pragma unroll
            for (int j = 0; j < 4; j++) r[j] = make_float4(1.f, 2.f, 3.f, 4.f);
            return;
        }
        const float *p = src + c * C::CR * ld;
#pragma unroll
        for (int j = 0; j < 4; j++) r[j] = *(const float4 *)(p + j * ld);
    };
    // registers -> three bf16 pieces -> LDS image
    auto stage = [&](const float4 (&r)[4], uint4 *img) {
        if constexpr (SB_PROBE == 2 || SB_PROBE == 4 || SB_PROBE == 5) {
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" ::"v"(r[j].x), "v"(r[j].y), "v"(r[j].z), "v"(r[j].w));
            return;
        }
        if (!on) return;
        uint2 *img2 = (uint2 *)img;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            uint32_t p[3][2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const float x0 = t == 0 ? r[2 * q].x : t == 1 ? r[2 * q].y : t == 2 ? r[2 * q].z : r[2 * q].w;
                const float x1 = t == 0 ? r[2 * q + 1].x : t == 1 ? r[2 * q + 1].y : t == 2 ? r[2 * q + 1].z : r[2 * q + 1].w;
                if constexpr (SB_PROBE == 1) {
                    p[0][q] = __builtin_bit_cast(uint32_t, x0), p[1][q] = __builtin_bit_cast(uint32_t, x1), p[2][q] = p[0][q] ^ p[1][q];
                } else {
                    split3(x0, x1, p[0][q], p[1][q], p[2][q]);
                }
            }
#pragma unroll
            for (int s = 0; s < 3; s++) img2[2 * s * C::NO * C::COLS + slot2 + 2 * (t ^ sx)] = make_uint2(p[s][0], p[s][1]);
        }
    };
    auto multiply = [&](const uint4 *buf) {
        if constexpr (SB_PROBE == 3) return;
        if constexpr (SB_PROBE == 5) {                   // matrix pipe only: no LDS reads
            uint4 z[3] = {make_uint4(tid, 1, 2, 3), make_uint4(4, tid, 6, 7), make_uint4(8, 9, tid, 11)};
#pragma unroll
            for (int mt = 0; mt < C::TM; mt++)
#pragma unroll
                for (int nt = 0; nt < C::TN; nt++) acc[mt][nt] = mma6(z, z, acc[mt][nt]);
            return;
        }
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
                for (int nt = 0; nt < C::TN; nt++) acc[mt][nt] = mma6(a, b[nt], acc[mt][nt]);
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
                for (int mt = 0; mt < C::TM; mt++) acc[mt][nt] = mma6(a[mt], b, acc[mt][nt]);
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
#ifdef SB_CLOCK
    if (tid == 0 && blockIdx.x == 7) { sb_clock[0] = __builtin_readcyclecounter() - clk0; sb_clock[1] = __builtin_amdgcn_s_memrealtime() - rt0; }
#endif
    // D tile (32 x 32): lane (i, g), register r -> row 8 (r / 4) + 4 g + r % 4 (the A operand's tile row: an M index), column i
    float *po = part + (size_t)blockIdx.x * C::M * C::N + (size_t)(32 * wm * C::TM + 4 * g) * C::N + 32 * wn * C::TN + i;
#pragma unroll
    for (int mt = 0; mt < C::TM; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int nt = 0; nt < C::TN; nt++) po[(size_t)(32 * mt + 8 * (r / 4) + (r % 4)) * C::N + 32 * nt] = acc[mt][nt][r];
}

// C[m][n] = (accumulate ? C[m][n] : 0) + sum_x part[x][m][n], x ascending: the same order every run
__global__ __launch_bounds__(256) void k_sb_reduce(int S, int MN, const float *__restrict__ part, float *__restrict__ Cm, int accumulate) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= MN) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int x = 0;
    for (; x + 4 <= S; x += 4) {
        s0 += part[(size_t)x * MN + idx];
        s1 += part[(size_t)(x + 1) * MN + idx];
        s2 += part[(size_t)(x + 2) * MN + idx];
        s3 += part[(size_t)(x + 3) * MN + idx];
    }
    for (; x < S; x++) s0 += part[(size_t)x * MN + idx];
    const float s = (s0 + s1) + (s2 + s3);
    Cm[idx] = accumulate ? Cm[idx] + s : s;
}

#ifndef SB_WGS
#define SB_WGS 256
#endif
constexpr int SB_WGRAD_WGS = SB_WGS;  // one workgroup per CU (96 KB of LDS each)

bool wgrad_shape_ok(int M, int N) {
    return (M == 128 || M == 256 || M == 384) && (N == 128 || N == 256 || N == 384) && M + N <= 512 && M * N < 256 * 256;
}

template <int MT, int NT>
int launch_wgrad(int64_t K, const float *A, int64_t lda, const float *B, int64_t ldb, float *Cm, int accumulate, float *part, hipStream_t st) {
    using C = WgCfg<MT, NT>;
    static_assert(C::UNITS <= 512 && C::WGM * C::WGN == 8 && C::TM * C::WGM * 32 == C::M && C::TN * C::WGN * 32 == C::N && SB_DEPTH % 2 == 0, "tiling");
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void *)k_sb_wgrad<MT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL((k_sb_wgrad<MT, NT>), dim3(SB_WGRAD_WGS), dim3(512), C::LDS_BYTES, st, A, lda, B, ldb, K, part);
    hipLaunchKernelGGL(k_sb_reduce, dim3((C::M * C::N + 255) / 256), dim3(256), 0, st, SB_WGRAD_WGS, C::M * C::N, (const float *)part, Cm,
                       accumulate);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int64_t sb_wgrad_workspace(int32_t M, int32_t N) {
    if (!wgrad_shape_ok(M, N)) return -1;
    return (int64_t)SB_WGRAD_WGS * M * N * sizeof(float);
}

int sb_wgrad_tn(int64_t K, int32_t M, int32_t N, const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int32_t accumulate,
                void *workspace, void *stream) {
    if (K < 1 || !wgrad_shape_ok(M, N) || !A || !B || !C || !workspace) return SB_ERR_BAD_ARG;
    if (lda < M || ldb < N || (lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return SB_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
#define SB_WG(mt, nt) if (M == 128 * mt && N == 128 * nt) return launch_wgrad<mt, nt>(K, A, lda, B, ldb, C, accumulate, part, st);
    SB_WG(1, 1) SB_WG(1, 2) SB_WG(2, 1) SB_WG(1, 3) SB_WG(3, 1)
#undef SB_WG
    return SB_ERR_BAD_ARG;
}

}  // extern "C"

// ---- harness ------------------------------------------------------------------------------------------------------------
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv) {
    const int64_t K = argc > 1 ? atoll(argv[1]) : 492000;
    const int shapes[][2] = {{384, 128}, {128, 384}, {128, 128}, {128, 256}};
    for (auto &sh : shapes) {
        const int M = sh[0], N = sh[1];
        float *A, *B, *C;
        void *ws;
        hipMalloc(&A, K * M * 4);
        hipMalloc(&B, K * N * 4);
        hipMalloc(&C, M * N * 4);
        hipMalloc(&ws, sb_wgrad_workspace(M, N));
        std::vector<float> h(K * 384);
        for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
        hipMemcpy(A, h.data(), K * M * 4, hipMemcpyHostToDevice);
        hipMemcpy(B, h.data(), K * N * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int w = 0; w < 3; w++) sb_wgrad_tn(K, M, N, A, M, B, N, C, 0, ws, 0);
        hipEventRecord(e0, 0);
        const int n = 20;
        for (int w = 0; w < n; w++) sb_wgrad_tn(K, M, N, A, M, B, N, C, 0, ws, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
#ifdef SB_CLOCK
        uint64_t hc[2];
        hipMemcpyFromSymbol(hc, HIP_SYMBOL(sb_clock), 16);
        printf("   shader clock %.0f MHz over %.1f us   ", hc[0] / (hc[1] / 100.0), hc[1] / 100.0);
#endif
        printf("probe %d wgs %d  %3d x %3d  K %lld: %7.1f us per call (kernel + reduce)  %6.0f GB/s\n", SB_PROBE, SB_WGS, M, N, (long long)K, ms / n * 1e3,
               K * (M + N) * 4.0 / (ms / n * 1e-3) / 1e9);
        hipFree(A); hipFree(B); hipFree(C); hipFree(ws);
    }
    return 0;
}
