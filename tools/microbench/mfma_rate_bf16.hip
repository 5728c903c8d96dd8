// Issue rate of the bf16 MFMAs the split kernels use, per SIMD, with one and two waves per SIMD and 1..6 accumulators per wave:
// cycles per instruction of v_mfma_f32_16x16x32_bf16 (4 passes) and v_mfma_f32_32x32x16_bf16 (8 passes) on every CU at once.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/mfma_rate_bf16.hip -o tools/microbench/mfma_rate_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool BIG>
__global__ __launch_bounds__(512) void k_rate(int iters, float *out, unsigned long long *cyc) {
    const int l = threadIdx.x;
    uint4 a[6], b[6];
    for (int k = 0; k < 6; k++) { a[k] = make_uint4(l + k, 2 * l, 3, 4 + k); b[k] = make_uint4(7 * l, k, l, 9); }
    f32x4 c4[6];
    f32x16 c16[6];
    for (int k = 0; k < 6; k++) { c4[k] = (f32x4){0, 0, 0, 0}; for (int r = 0; r < 16; r++) c16[k][r] = 0.f; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 6; u++)
#pragma unroll
            for (int k = 0; k < NACC; k++) {
                if constexpr (BIG) c16[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(u + k) % 6]), __builtin_bit_cast(bf16x8, b[u]), c16[k], 0, 0, 0);
                else c4[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(u + k) % 6]), __builtin_bit_cast(bf16x8, b[u]), c4[k], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int k = 0; k < NACC; k++) { s += c4[k][0] + c16[k][0]; }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0 && blockIdx.x == 3) cyc[0] = t1 - t0;
}

template <int NACC, bool BIG>
void run(int threads, const char *what) {
    float *out; unsigned long long *cyc, h;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k_rate<NACC, BIG>), dim3(256), dim3(threads), 0, 0, iters, out, cyc);
    hipLaunchKernelGGL((k_rate<NACC, BIG>), dim3(256), dim3(threads), 0, 0, iters, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per_wave = (double)h / (iters * 6.0 * NACC);
    const int waves_per_simd = threads / 256;
    printf("%-14s %d accumulators, %d wave(s) per SIMD: %6.1f cycles per MFMA per wave = %6.1f per SIMD\n", what, NACC, waves_per_simd, per_wave, per_wave / waves_per_simd);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<1, false>(256, "16x16x32"); run<2, false>(256, "16x16x32"); run<3, false>(256, "16x16x32"); run<6, false>(256, "16x16x32");
    run<1, false>(512, "16x16x32"); run<2, false>(512, "16x16x32"); run<3, false>(512, "16x16x32"); run<6, false>(512, "16x16x32");
    run<1, true>(256, "32x32x16"); run<2, true>(256, "32x32x16"); run<6, true>(256, "32x32x16");
    run<1, true>(512, "32x32x16"); run<2, true>(512, "32x32x16"); run<6, true>(512, "32x32x16");
    return 0;
}
