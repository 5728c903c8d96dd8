// How do the matrix instructions round their fp32 accumulator?  One wave, C = +-1.0 (or 2^23), ONE non-zero product p per output:
// D = C + p for p = f x 2^-24 (f = 0.25 .. 1.99 ulp-halves of 1.0).  Round-to-nearest-even gives 1 + 2^-23 from f > 1 (and for f = 1.5);
// truncation keeps 1.0 until f >= 2.  Also: k products of 0.6 x 2^-24 each in one instruction (is the 32-term sum formed before it
// meets C, or are the terms truncated one by one against C's exponent?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/mfma_round_probe.hip -o tools/microbench/mfma_round_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ uint32_t bf16_bits(float x) { uint32_t u = __builtin_bit_cast(uint32_t, x); return u >> 16; }   // exact for the values used here

// kind 0: v_mfma_f32_16x16x32_bf16, 1: v_mfma_f32_32x32x16_bf16, 2: v_mfma_f32_16x16x4_f32
__global__ void k_probe(int kind, float c, float a, float b, int nterms, int signs, float *out) {
    const int l = threadIdx.x;
    // A row i, steps 8 g .. 8 g + 7 (16x16x32: i = l % 16, g = l / 16): put `a` into the first `nterms` steps of every row
    uint32_t aw[4] = {0, 0, 0, 0}, bw[4] = {0, 0, 0, 0};
    const int g = kind == 1 ? l / 32 : l / 16;
    for (int s = 0; s < 8; s++) {
        const int step = 8 * g + s;
        if (step < nterms) {
            const float as = (signs == 1 || (signs == 2 && (step & 1))) ? -a : a;
            aw[s / 2] |= bf16_bits(as) << (16 * (s & 1));
            bw[s / 2] |= bf16_bits(b) << (16 * (s & 1));
        }
    }
    const uint4 A = make_uint4(aw[0], aw[1], aw[2], aw[3]), B = make_uint4(bw[0], bw[1], bw[2], bw[3]);
    float r = 0.f;
    if (kind == 0) {
        f32x4 acc = {c, c, c, c};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc, 0, 0, 0);
        r = acc[0];
    } else if (kind == 1) {
        f32x16 acc;
        for (int i = 0; i < 16; i++) acc[i] = c;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc, 0, 0, 0);
        r = acc[0];
    } else {
        f32x4 acc = {c, c, c, c};
        const int k = l / 16;   // 16x16x4: lane (i, k) holds step k
        const float as = (signs == 1 || (signs == 2 && (k & 1))) ? -a : a;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k < nterms ? as : 0.f, k < nterms ? b : 0.f, acc, 0, 0, 0);
        r = acc[0];
    }
    if (l == 0) out[0] = r;
}

static float run(int kind, float c, float a, float b, int nterms, int signs = 0) {
    float *d, h;
    (void)hipMalloc(&d, 4);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, kind, c, a, b, nterms, signs, d);
    (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return h;
}

int main() {
    const char *names[3] = {"v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x4_f32 "};
    const float ulp = 1.1920929e-07f;   // 2^-23
    for (int kind = 0; kind < 3; kind++) {
        printf("%s  D = C + a*b, C = 1.0; result in ulps of 1.0 above C\n", names[kind]);
        const float fs[] = {0.25f, 0.5f, 0.75f, 1.0f, 1.25f, 1.5f, 1.75f, 1.9921875f, 2.0f, 2.5f, 3.0f};
        for (float f : fs) {
            // p = f * 2^-24 = (f) * (2^-24): both factors are bf16-exact (f has <= 8 significant bits)
            const float up = run(kind, 1.0f, f, 5.9604645e-08f, 1), dn = run(kind, -1.0f, f, -5.9604645e-08f, 1);
            const float mixed = run(kind, 1.0f, f, -5.9604645e-08f, 1);
            printf("  p = %-9g x 2^-24:  C=+1,p>0 -> +%g ulp   C=-1,p<0 -> %g ulp   C=+1,p<0 -> %g ulp(below)\n", f, (up - 1.0f) / ulp, (dn + 1.0f) / ulp,
                   (mixed - 1.0f) / (ulp / 2));
        }
        const int maxk = kind == 2 ? 4 : (kind == 0 ? 32 : 16);
        for (int k = 1; k <= maxk; k *= 2) {
            const float v = run(kind, 1.0f, 0.59765625f, 5.9604645e-08f, k);   // k terms of 0.598 x 2^-24: exact sum = 0.598 k x 2^-24
            printf("  %2d terms of 0.598 x 2^-24 (exact sum %.3f ulp): +%g ulp\n", k, 0.59765625 * k / 2, (v - 1.0f) / ulp);
        }
        // direction of the truncation: C = 1.5 (ulp 2^-23 on both sides), maxk terms of +-0.299 ulp
        const float pos = run(kind, 1.5f, 0.59765625f, 5.9604645e-08f, maxk, 0), neg = run(kind, 1.5f, 0.59765625f, 5.9604645e-08f, maxk, 1),
                    alt = run(kind, 1.5f, 0.59765625f, 5.9604645e-08f, maxk, 2), cneg = run(kind, -1.5f, 0.59765625f, 5.9604645e-08f, maxk, 0);
        printf("  C = 1.5, %d terms of 0.299 ulp (exact %.2f ulp): all positive %+g, all negative %+g, alternating %+g; C = -1.5, all positive %+g\n", maxk,
               0.29882812 * maxk, (pos - 1.5f) / ulp, (neg - 1.5f) / ulp, (alt - 1.5f) / ulp, (cneg + 1.5f) / ulp);
        // C = 0: are the products then summed exactly?  32 terms, one of 1.0 and the rest 0.299 x 2^-23 each
        const float z = run(kind, 0.0f, 0.59765625f, 5.9604645e-08f, maxk, 0);
        printf("  C = 0, %d terms of 0.598 x 2^-24: %.9g (exact %.9g)\n", maxk, z, 0.59765625 * 5.9604645e-08 * maxk);
    }
    return 0;
}
