// In-kernel clock under a sustained fp32-MFMA load shaped like a persistent GRU step (MI355X_MICROARCH.md, DVFS give-back item 6):
// clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, stamped around a loop of v_mfma_f32_16x16x4_f32 on random operands, 8 waves
// per workgroup, one workgroup per CU, after ~2 s of back-to-back launches.  Also prints cycles per MFMA per SIMD.
// The 157.3 TFLOP/s fp32 peak assumes 2.4 GHz; the MFMA floor of a kernel is set by the clock the chip holds under ITS load.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool BARRIER>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *stamps, const float *w, int iters) {
    __shared__ __attribute__((aligned(16))) float hs[4 * 16 * 36];
    float wr[32], wz[32], wn[32];
    for (int i = 0; i < 32; i++) { wr[i] = w[threadIdx.x + 512 * i]; wz[i] = w[threadIdx.x + 512 * i + 7]; wn[i] = w[threadIdx.x + 512 * i + 13]; }
    for (int i = threadIdx.x; i < 4 * 16 * 36; i += blockDim.x) hs[i] = w[i + 31];
    __syncthreads();
    const int l = threadIdx.x & 63, c16 = l & 15, q = l >> 4;
    const float *hp = &hs[(q * 16 + c16) * 36];
    v4f ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k4 = 0; k4 < 8; k4++) {
            const v4f a = *(const v4f *)(hp + 4 * k4);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[4 * k4 + u], a[u], ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[4 * k4 + u], a[u], az, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[4 * k4 + u], a[u], an, 0, 0, 0);
            }
        }
        if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = ar[0] + az[1] + an[2];
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <bool BARRIER>
void run(const float *w, int grid) {
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float)); (void)hipMalloc(&st, 2 * 256 * sizeof(unsigned long long));
    const int iters = 2000;   // 2000 x 96 MFMAs per wave: ~5 ms per launch
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 400; rep++) {   // ~2 s of back-to-back launches, the last one is read
        if (rep == 399) (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<BARRIER>, dim3(grid), dim3(512), 0, 0, out, st, w, iters);
        if (rep == 399) (void)hipEventRecord(e1, 0);
    }
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * grid);
    (void)hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> ghz(grid), cpm(grid);
    for (int b = 0; b < grid; b++) { ghz[b] = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1; cpm[b] = (double)h[2 * b] / (iters * 96.0 * 2.0); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cpm.begin(), cpm.end());
    const double flops = 2.0 * 16 * 16 * 4 * 96.0 * iters * 8 * grid;
    printf("grid %3d barrier %d: in-kernel clock median %.3f GHz (min %.3f max %.3f), %.1f shader cycles per MFMA per SIMD (2 waves), launch %.3f ms = %.1f TFLOP/s "
           "(%.1f at a full chip)\n", grid, (int)BARRIER, ghz[grid / 2], ghz[0], ghz[grid - 1], cpm[grid / 2], ms, flops / ms / 1e9, flops / ms / 1e9 * 256.0 / grid);
    (void)hipFree(out); (void)hipFree(st);
}

int main() {
    float *w; (void)hipMalloc(&w, 1 << 20);
    std::vector<float> hw(1 << 18);
    srand(1);
    for (auto &v : hw) v = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMemcpy(w, hw.data(), hw.size() * sizeof(float), hipMemcpyHostToDevice);
    run<false>(w, 256); run<true>(w, 256); run<true>(w, 205);
    return 0;
}
