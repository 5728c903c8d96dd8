/*
 * mappo_gemm.h -- C ABI of the Linear-layer GEMM with a strided output and a fused epilogue (hipBLASLt fp32 MFMA underneath).
 *
 * Replaces, on the MAPPO hot path, torch's `F.linear` / `torch.cat(...)` + `nn.Linear` + `ReLU` sequences of DHGN.fcra
 * (reference DHGN/mappo_parallel.py:204-233: `FCRA_layers[k](cat([agg, h]))`, `ReLU`) where PyTorch's own GEMM bindings cannot
 * express the call: a bias + ReLU epilogue whose OUTPUT is a column block of a wider matrix (row stride ldd > N), so that the
 * [agg | h] operand of the next layer is written in place by its two producers and never concatenated.  Row-major throughout:
 *     D[m][n] = act( sum_k A[m][k] W[n][k] + bias[n] + C[m][n] ),   m < M, n < N, k < K
 * A: M x K, rows lda apart (lda >= K);  W: N x K, rows ldw apart (an nn.Linear weight or a column slice of one);  bias: N or NULL;
 * C: M x N addend, rows ldc apart, or NULL;  D: M x N, rows ldd apart (D may alias C);  act = ReLU when relu != 0.
 * All DEVICE pointers, fp32, 16-byte aligned; workspace: device scratch of workspace_bytes (>= mo_gemm_workspace_bytes()).
 * The launch is enqueued on `stream`; returns 0 or a non-zero code (MO_GEMM_ERR_* / hipblasStatus_t + 1000), never throws.
 * State: one hipBLASLt handle and a cache of the selected algorithm per problem shape (process-wide, mutex-protected).
 */
#ifndef MAPPO_GEMM_H
#define MAPPO_GEMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MO_GEMM_ERR_BAD_ARG 20001
#define MO_GEMM_ERR_NO_ALGO 20002

int64_t mo_gemm_workspace_bytes(void);
int mo_gemm_nt(int64_t M, int32_t N, int32_t K, const float *A, int64_t lda, const float *W, int64_t ldw, const float *bias, const float *C,
               int64_t ldc, int32_t relu, float *D, int64_t ldd, void *workspace, int64_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
