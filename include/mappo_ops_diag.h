/*
 * mappo_ops_diag.h -- TEST-ONLY entry points of libmappo_ops.so (not part of the drop-in boundary of mappo_ops.h).
 * They expose, for the parity tests, the exact-arithmetic building block the split-bf16 kernels rely on.
 */
#ifndef MAPPO_OPS_DIAG_H
#define MAPPO_OPS_DIAG_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The three-way bf16 split of the split-bf16 kernels (csrc/mappo_ops.hip sb_split2: p1 = bf16(x), p2 = bf16(x - p1),
 * p3 = bf16(x - p1 - p2), round to nearest even) applied to x [n]; pieces [3][n] receives the pieces widened back to fp32.
 * The kernels rely on p1 + p2 + p3 == x EXACTLY for every finite |x| < 2^128 - 2^119 (the largest fp32 whose first piece does
 * not round to infinity; tests/test_ops_gpu.py test_split_bf16_pieces_sum_exactly). */
int sb_split_diag(int64_t n, const float *x, float *pieces, void *stream);

#ifdef __cplusplus
}
#endif
#endif
