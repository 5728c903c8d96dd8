// v_mfma_f32_16x16x4_f32 issue rate with the operand patterns of the GRU kernels: B operands from 96 distinct VGPRs (weights in
// registers), A operands from LDS (ds_read_b128), 3-4 accumulators.  Prints cycles per MFMA for one wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, long long *cyc, const float *w, int iters) {
    __shared__ __attribute__((aligned(16))) float hs[4 * 16 * 36];
    float wr[32], wz[32], wn[32];
    for (int i = 0; i < 32; i++) { wr[i] = w[threadIdx.x + 64 * i]; wz[i] = w[threadIdx.x + 64 * i + 7]; wn[i] = w[threadIdx.x + 64 * i + 13]; }
    for (int i = threadIdx.x; i < 4 * 16 * 36; i += blockDim.x) hs[i] = (float)i * 1e-3f;
    __syncthreads();
    const int l = threadIdx.x & 63, c16 = l & 15, q = l >> 4;
    const float *hp = &hs[(q * 16 + c16) * 36];
    v4f ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar;
    float areg = (float)l;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k4 = 0; k4 < 8; k4++) {
            v4f a;
            if (MODE & 1) a = *(const v4f *)(hp + 4 * k4);          // A from LDS
            else a = (v4f){areg, areg, areg, areg};                   // A constant register
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (MODE & 2) {                                          // B from 96 distinct registers
                    ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], wr[4 * k4 + u], ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], wz[4 * k4 + u], az, 0, 0, 0);
                    an = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], wn[4 * k4 + u], an, 0, 0, 0);
                } else {
                    ar = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], wr[0], ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], wz[0], az, 0, 0, 0);
                    an = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], wn[0], an, 0, 0, 0);
                }
            }
        }
        if (MODE & 4) __syncthreads();                                   // one barrier per 96 MFMAs, as per GRU step
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = ar[0] + az[1] + an[2] + wr[5] + wz[9] + wn[31];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(int threads, const float *w) {
    float *out; long long *cyc;
    (void)hipMalloc(&out, 256 * 1024 * sizeof(float)); (void)hipMalloc(&cyc, 256 * sizeof(long long));
    const int iters = 64;
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, w, iters);
    (void)hipDeviceSynchronize();
    long long h[256]; (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("threads %4d  A %-8s B %-12s barrier %d : %.1f cycles per MFMA per wave\n", threads, (MODE & 1) ? "LDS" : "reg", (MODE & 2) ? "96 regs" : "1 reg",
           (MODE & 4) ? 1 : 0, (double)h[0] / (iters * 96.0));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    float *w; (void)hipMalloc(&w, 1 << 20); (void)hipMemset(w, 0, 1 << 20);
    for (int th : {256, 512}) { run<0>(th, w); run<1>(th, w); run<2>(th, w); run<3>(th, w); run<7>(th, w); }
    return 0;
}
