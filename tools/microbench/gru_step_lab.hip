// Phase timing of one step of the persistent GRU forward kernel (csrc/mappo_ops.hip k_gru_seq_fwd2, same code, plus s_memtime stamps):
// per wave of workgroup 0: [0] step top, [1] operand fragments landed / first MFMA issued, [2] last MFMA result consumed (start of the gate
// math), [3] gate math done (before the stores), [4] stores issued (before the barrier), [5] after the barrier.  Diagnostic build only.
// Variants: 5 = the stores of a step issued at the top of the NEXT step, behind the barrier; 0 = as shipped; 1 = no global stores of out/save; 2 = no gi loads; 3 = neither; 4 = as shipped but the gate math replaced by moves
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int GRU_H = 128, GRU_RB = 16, GRU2_LD = 36, NST = 6;
__device__ __forceinline__ float sigmoid_hw(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_hw(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
__device__ __forceinline__ int gru2_hidx(int row, int k) { return ((k >> 5) * GRU_RB + row) * GRU2_LD + (k & 31); }
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#define STAMP(i) if (stamp_on) { st[i] = __builtin_amdgcn_s_memtime(); }

template <int V>
__global__ __launch_bounds__(512) void k_fwd(int T, int B, const float *__restrict__ gi, const float *__restrict__ w_hh, const float *__restrict__ b_hh,
                                             const float *__restrict__ h0, float *__restrict__ out, float *__restrict__ save, unsigned long long *stamps, int t_probe,
                                             unsigned long long *wg_clk) {
    const unsigned long long kc0 = __builtin_amdgcn_s_memtime(), kr0 = __builtin_amdgcn_s_memrealtime();
    __shared__ __attribute__((aligned(16))) float hs[2][4 * GRU_RB * GRU2_LD];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int b0 = blockIdx.x * GRU_RB;
    const int c16 = l & 15, q = l >> 4;
    float wr[32], wz[32], wn[32];
#define LOADW(dst, gate) { const float4 *src = (const float4 *)(w_hh + (size_t)((gate) * GRU_H + 16 * w + c16) * GRU_H + 32 * q); \
    _Pragma("unroll") for (int i = 0; i < 8; i++) { const float4 v = src[i]; dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w; } }
    LOADW(wr, 0) LOADW(wz, 1) LOADW(wn, 2)
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
    const size_t nblk = gridDim.x;
    float4 *sv = (float4 *)save + (size_t)blockIdx.x * 4 * 512 + tid;
    int cur = 0;
    unsigned long long st[NST];
    float4 d_r, d_z, d_n, d_hn, d_h;   // variant 5: last step's results, stored one step late
    d_r = d_z = d_n = d_hn = d_h = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < T; t++) {
        const bool stamp_on = (t == t_probe) && blockIdx.x == 0;
        STAMP(0)
        float4 gr = make_float4(0.1f, 0.2f, 0.3f, 0.4f), gz = gr, gn = gr;
        if (live && !(V & 2)) {
            const float *g = gi + ((size_t)t * B + b0 + row) * 3 * GRU_H + u0;
            gr = *(const float4 *)g; gz = *(const float4 *)(g + GRU_H); gn = *(const float4 *)(g + 2 * GRU_H);
        }
        const float *hp = &hs[cur][(q * GRU_RB + c16) * GRU2_LD];
        f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar;
#pragma unroll
        for (int k4 = 0; k4 < 8; k4++) {
            const f32x4 a = *(const f32x4 *)(hp + 4 * k4);
            if (V == 5 && k4 == 1 && t > 0 && live) {   // after the first group of MFMAs is in the pipe
                *(float4 *)(out + ((size_t)(t - 1) * B + b0 + row) * GRU_H + u0) = d_h;
                float4 *s4 = sv + (size_t)(t - 1) * nblk * 4 * 512;
                s4[0] = d_r; s4[512] = d_z; s4[1024] = d_n; s4[1536] = d_hn;
            }
            if (k4 == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); STAMP(1) }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[4 * k4 + u], a[u], ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_16x16x4f32(wz[4 * k4 + u], a[u], az, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[4 * k4 + u], a[u], an, 0, 0, 0);
            }
        }
        const float4 hprev = *(const float4 *)&hs[cur][hpos];
        float keep = ar[0] + az[1] + an[2];   // forces the accumulators
        asm volatile("" : "+v"(keep));
        STAMP(2)
        float4 r, z, hn, n, hnew;
        if (V == 4) {
            r = make_float4(ar[0], ar[1], ar[2], ar[3]); z = make_float4(az[0], az[1], az[2], az[3]); hn = make_float4(an[0], an[1], an[2], an[3]); n = gr;
            hnew = make_float4(0.5f * hprev.x + 1e-3f * r.x, 0.5f * hprev.y + 1e-3f * z.y, 0.5f * hprev.z + 1e-3f * hn.z, 0.5f * hprev.w + gn.x * 1e-3f + gz.x * 1e-3f);
        } else {
#define ONE(f, i) r.f = sigmoid_hw(gr.f + ar[i] + br.f); z.f = sigmoid_hw(gz.f + az[i] + bz.f); hn.f = an[i] + bn.f; n.f = tanh_hw(gn.f + r.f * hn.f); \
        hnew.f = (1.f - z.f) * n.f + z.f * hprev.f;
            ONE(x, 0) ONE(y, 1) ONE(z, 2) ONE(w, 3)
        }
        hnew.x += keep * 0.f;
        *(float4 *)&hs[cur ^ 1][hpos] = hnew;
        asm volatile("" ::: "memory");
        STAMP(3)
        if (V == 5) { d_r = r; d_z = z; d_n = n; d_hn = hn; d_h = hnew; }
        else if (live && !(V & 1)) {
            *(float4 *)(out + ((size_t)t * B + b0 + row) * GRU_H + u0) = hnew;
            float4 *s4 = sv + (size_t)t * nblk * 4 * 512;
            s4[0] = r; s4[512] = z; s4[1024] = n; s4[1536] = hn;
        }
        STAMP(4)
        lds_barrier();
        STAMP(5)
        if (stamp_on && l == 0) for (int i = 0; i < NST; i++) stamps[w * NST + i] = st[i];
        cur ^= 1;
    }
    if ((V & 1) && V != 5 && live) *(float4 *)(out + (size_t)(b0 + row) * GRU_H + u0) = *(const float4 *)&hs[cur][hpos];
    if (V == 5 && live) {
        *(float4 *)(out + ((size_t)(T - 1) * B + b0 + row) * GRU_H + u0) = d_h;
        float4 *s4 = sv + (size_t)(T - 1) * nblk * 4 * 512;
        s4[0] = d_r; s4[512] = d_z; s4[1024] = d_n; s4[1536] = d_hn;
    }
    if (tid == 0) { wg_clk[3 * blockIdx.x] = __builtin_amdgcn_s_memtime() - kc0; wg_clk[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - kr0; wg_clk[3 * blockIdx.x + 2] = kr0; }
}

template <int V>
void run(int T, int B, const float *gi, const float *w, const float *b, const float *h0, float *out, float *save, unsigned long long *stamps) {
    const int nblk = (B + 15) / 16;
    static unsigned long long *wg_clk = nullptr;
    if (!wg_clk) (void)hipMalloc(&wg_clk, 3 * 256 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_fwd<V>, dim3(nblk), dim3(512), 0, 0, T, B, gi, w, b, h0, out, save, stamps, -1, wg_clk);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k_fwd<V>, dim3(nblk), dim3(512), 0, 0, T, B, gi, w, b, h0, out, save, stamps, -1, wg_clk);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("variant %d: %.1f us per launch, %.3f us per step\n", V, ms * 100.f, ms * 100.f / T);
    {
        std::vector<unsigned long long> h(3 * nblk);
        (void)hipMemcpy(h.data(), wg_clk, h.size() * 8, hipMemcpyDeviceToHost);
        double cmin = 1e30, cmax = 0, rmin = 1e30, rmax = 0, smin = 1e30, smax = 0;
        for (int i = 0; i < nblk; i++) {
            cmin = std::min(cmin, (double)h[3 * i]); cmax = std::max(cmax, (double)h[3 * i]);
            rmin = std::min(rmin, (double)h[3 * i + 1]); rmax = std::max(rmax, (double)h[3 * i + 1]);
            smin = std::min(smin, (double)h[3 * i + 2]); smax = std::max(smax, (double)h[3 * i + 2]);
        }
        printf("  per workgroup: %.0f .. %.0f shader cycles, %.1f .. %.1f us wall (s_memrealtime), start skew %.1f us; clock of the slowest = %.2f GHz\n",
               cmin, cmax, rmin * 0.01, rmax * 0.01, (smax - smin) * 0.01, cmax / rmax * 0.1);
    }
    for (int tp : {60, 61}) {
        hipLaunchKernelGGL(k_fwd<V>, dim3(nblk), dim3(512), 0, 0, T, B, gi, w, b, h0, out, save, stamps, tp, wg_clk);
        (void)hipDeviceSynchronize();
        unsigned long long h[8 * NST]; (void)hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost);
        unsigned long long base = h[0];
        for (int wv = 0; wv < 8; wv++) if (h[wv * NST] < base) base = h[wv * NST];
        printf("  step %d (cycles from the earliest wave's step top): top | frag | mfma-done | gates-done | stores-issued | barrier-passed\n", tp);
        for (int wv = 0; wv < 8; wv++) {
            printf("    w%d:", wv);
            for (int i = 0; i < NST; i++) printf(" %6llu", h[wv * NST + i] - base);
            printf("\n");
        }
    }
}

int main(int argc, char **argv) {
    const int T = 150, B = 3280, nblk = (B + 15) / 16;
    float *gi, *w, *b, *h0, *out, *save; unsigned long long *stamps;
    (void)hipMalloc(&gi, (size_t)T * B * 384 * 4); (void)hipMalloc(&w, 384 * 128 * 4); (void)hipMalloc(&b, 384 * 4); (void)hipMalloc(&h0, (size_t)B * 128 * 4);
    (void)hipMalloc(&out, (size_t)T * B * 128 * 4); (void)hipMalloc(&save, (size_t)T * nblk * 4 * 512 * 16); (void)hipMalloc(&stamps, 8 * NST * 8);
    std::vector<float> hg((size_t)T * B * 384);
    srand(1);
    for (auto &v : hg) v = ((float)rand() / 2147483647.f - 0.5f) * 2.f;
    (void)hipMemcpy(gi, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < 384 * 128; i++) hg[i] *= 0.08f;
    (void)hipMemcpy(w, hg.data(), 384 * 128 * 4, hipMemcpyHostToDevice);
    (void)hipMemset(b, 0, 384 * 4); (void)hipMemset(h0, 0, (size_t)B * 128 * 4);
    run<0>(T, B, gi, w, b, h0, out, save, stamps);
    run<1>(T, B, gi, w, b, h0, out, save, stamps);
    run<2>(T, B, gi, w, b, h0, out, save, stamps);
    run<3>(T, B, gi, w, b, h0, out, save, stamps);
    run<4>(T, B, gi, w, b, h0, out, save, stamps);
    run<5>(T, B, gi, w, b, h0, out, save, stamps);
    return 0;
}
