// sb_common.hpp -- building blocks of the split-bf16 kernels (fp32 arithmetic on the bf16 matrix pipe) shared by csrc/mappo_ops.hip
// and csrc/mappo_split.hip (and the lab builds under tools/microbench/).
//
// Every fp32 number is EXACTLY the sum of three bf16 numbers (24 significant bits = 3 x 8, same exponent range): x = x1 + x2 + x3, and
// a b = sum of the piece products ai bj, each exact in fp32.  Keeping the six with i + j <= 4 drops a2 b3 + a3 b2 + a3 b3 < 2^-25 |a b|;
// the accumulation is fp32 either way.  Six bf16 MFMAs per 32 contraction steps replace eight fp32 ones at half the cycles each.
// Domain (tests/test_split_bf16_gpu.py): exact for 2^-110 <= |x| <= 0x7F7F7FFF (3.3895e38); below 2^-110 the low pieces fall under
// bf16's subnormal grid and the sum is x to within 2^-134; from 0x7F7F8000 up the first piece rounds to infinity.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Workgroup barrier for kernels whose waves exchange data through LDS only.  __syncthreads() is a workgroup-scope release +
// acquire around s_barrier, and the release makes every wave wait for ALL its outstanding memory operations (s_waitcnt vmcnt(0)):
// in the persistent GRU kernels that drained 40-80 KB of freshly issued global stores per step at ~10 B/clk/CU before any wave
// could start the next step's MFMAs (1.7 of 4.6 us per step).  Here a wave waits for its own LDS operations only; global loads
// and stores stay in flight across the barrier and complete under the next matrix phase.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ uint32_t sb_pk(float lo, float hi) {   // two bf16 in one dword, round to nearest even (v_cvt_pk_bf16_f32)
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// x = p1 + p2 + p3 exactly: p1 = bf16(x); x - p1 has <= 16 significant bits and is exact in fp32; p2 = bf16(x - p1); the rest IS a bf16
__device__ __forceinline__ void sb_split2(float x0, float x1, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    p1 = sb_pk(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, p1 << 16), r1 = x1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
    p2 = sb_pk(r0, r1);
    p3 = sb_pk(r0 - __builtin_bit_cast(float, p2 << 16), r1 - __builtin_bit_cast(float, p2 & 0xffff0000u));
}
// 8 consecutive contraction steps -> the three operand words of one lane
__device__ __forceinline__ void sb_split8(const float4 &u, const float4 &v, uint4 (&p)[3]) {
    sb_split2(u.x, u.y, p[0].x, p[1].x, p[2].x);
    sb_split2(u.z, u.w, p[0].y, p[1].y, p[2].y);
    sb_split2(v.x, v.y, p[0].z, p[1].z, p[2].z);
    sb_split2(v.z, v.w, p[0].w, p[1].w, p[2].w);
}
// the six piece products of one 16 x 16 x 32 step, smallest first
__device__ __forceinline__ f32x4 sb_mma6(const uint4 (&a)[3], const uint4 (&b)[3], f32x4 c) {
#define SB_MMA(i, j) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), c, 0, 0, 0);
    SB_MMA(2, 0) SB_MMA(0, 2) SB_MMA(1, 1) SB_MMA(1, 0) SB_MMA(0, 1) SB_MMA(0, 0)
#undef SB_MMA
    return c;
}

// The same six products with the LARGE one (a1 b1) and the five small ones in separate accumulators.  The bf16 matrix instruction
// aligns its 32 products to the largest exponent among them and the accumulator and truncates each with two guard bits
// (tools/microbench/mfma_round_probe.hip; the fp32 instruction is a chain of round-to-nearest FMAs): a small piece product added to a
// large accumulator loses up to a quarter ulp of the ACCUMULATOR.  In `lo` the small products meet an accumulator 2^-8 of the size,
// `hi` takes one sixth of the accumulations; hi + lo is one round-to-nearest add at the end.
__device__ __forceinline__ void sb_mma6_hl(const uint4 (&a)[3], const uint4 (&b)[3], f32x4 &hi, f32x4 &lo) {
#define SB_MMA(i, j, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), c, 0, 0, 0);
    SB_MMA(2, 0, lo) SB_MMA(0, 2, lo) SB_MMA(1, 1, lo) SB_MMA(0, 0, hi) SB_MMA(1, 0, lo) SB_MMA(0, 1, lo)
#undef SB_MMA
}

// the six piece products of one 32 x 32 x 16 tile step, smallest first
__device__ __forceinline__ f32x16 sb_mma6_32(const uint4 (&a)[3], const uint4 (&b)[3], f32x16 c) {
#define SB_MMA(i, j) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), c, 0, 0, 0);
    SB_MMA(2, 0) SB_MMA(0, 2) SB_MMA(1, 1) SB_MMA(1, 0) SB_MMA(0, 1) SB_MMA(0, 0)
#undef SB_MMA
    return c;
}

