// Issue rate of v_mfma_f32_16x16x4_f32 for one / two waves per SIMD and 1..16 independent accumulators
// (hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate).  Prints cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(float *out, long long *cyc, int iters) {
    v4f acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    float a = (float)threadIdx.x, b = 1.0f + (float)(threadIdx.x & 3);
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int threads) {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 256 * sizeof(long long));
    const int iters = 4096 / NACC;
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const int waves_per_simd = threads / 256;
    printf("waves/SIMD %d, %2d accumulators: %.1f cycles per MFMA per wave, %.1f per SIMD\n", waves_per_simd, NACC,
           (double)h[0] / (iters * NACC), (double)h[0] / (iters * NACC) / waves_per_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int th : {256, 512}) { run<1>(th); run<2>(th); run<4>(th); run<8>(th); run<16>(th); }
    return 0;
}
