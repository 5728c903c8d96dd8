// mappo_split.hip -- the split-bf16 GEMM (and GRU sequence) kernels of libmappo_ops.so: a second translation unit so that a change
// to these kernels does not recompile csrc/mappo_ops.hip.  C ABI: include/mappo_ops.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "mappo_ops.h"
#include "sb_gemm.hpp"
#include "sb_gru_seq.hpp"

extern "C" {

static int sb_gemm_any(int64_t R, int32_t N, int32_t K, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, int32_t relu,
                       const float *addend, int64_t lda, float *Y, int64_t ldy, uint8_t *ybits, int64_t ldyb, void *stream) {
    if (R < 0 || !X || !W || !Y || ldx < K || ldw < K || ldy < N || (addend && lda < N)) return MO_ERR_BAD_ARG;
    if ((ldx & 3) || (ldw & 3) || (ldy & 3) || (lda & 3) || ((uintptr_t)X & 15) || ((uintptr_t)W & 15) || ((uintptr_t)Y & 15) || ((uintptr_t)addend & 15) ||
        ((uintptr_t)bias & 15))
        return MO_ERR_BAD_ARG;
    if (R == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (N == 128 && K == 128) return launch_sb_gemm_best<4, 1>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st, ybits, ldyb);
    if (N == 128 && K == 256) return launch_sb_gemm_best<8, 1>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st, ybits, ldyb);
    if (ybits) return MO_ERR_BAD_ARG;      // sign bits: the ReLU layers' shapes (128 outputs from 128 / 256 inputs)
    if (N == 128 && K == 384) return launch_sb_gemm_best<12, 1>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st);
    if (N == 256 && K == 128) return launch_sb_gemm_best<4, 2>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st);
    if (N == 384 && K == 128) return launch_sb_gemm_best<4, 3>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st);
    return MO_ERR_BAD_ARG;
}

int sb_gemm(int64_t R, int32_t N, int32_t K, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, int32_t relu,
            const float *addend, int64_t lda, float *Y, int64_t ldy, void *stream) {
    return sb_gemm_any(R, N, K, X, ldx, W, ldw, bias, relu, addend, lda, Y, ldy, nullptr, 0, stream);
}

int sb_gemm_signs(int64_t R, int32_t N, int32_t K, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, int32_t relu,
                  const float *addend, int64_t lda, float *Y, int64_t ldy, uint8_t *y_sign_bits, int64_t ld_bytes, void *stream) {
    if (!y_sign_bits || ld_bytes < N / 8) return MO_ERR_BAD_ARG;
    return sb_gemm_any(R, N, K, X, ldx, W, ldw, bias, relu, addend, lda, Y, ldy, y_sign_bits, ld_bytes, stream);
}

int64_t sb_gemm_masked_workspace(int32_t N) { return N > 0 ? (int64_t)1024 * N * 4 : 0; }   // one row of partial sums per workgroup (<= CUs)

int sb_gemm_masked(int64_t R, int32_t N, int32_t K, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *M, int64_t ldm,
                   int32_t mask_cols, float *Y, int64_t ldy, float *colsum, void *workspace, void *stream) {
    if (R < 0 || !X || !W || !M || !Y || !colsum || !workspace || ldx < K || ldw < K || ldy < N || ldm < N) return MO_ERR_BAD_ARG;
    if (mask_cols < 0 || mask_cols > N || (mask_cols & 127)) return MO_ERR_BAD_ARG;
    if ((ldx & 3) || (ldw & 3) || (ldy & 3) || (ldm & 3) || ((uintptr_t)X & 15) || ((uintptr_t)W & 15) || ((uintptr_t)Y & 15) || ((uintptr_t)M & 15) ||
        ((uintptr_t)colsum & 15) || ((uintptr_t)workspace & 15))
        return MO_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (R == 0) return (int)hipMemsetAsync(colsum, 0, (size_t)N * 4, st);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    if (cus > 512) return MO_ERR_BAD_ARG;          // (the workspace holds 1024 rows: up to two workgroups per CU)
    float *part = (float *)workspace;
    const int mt = mask_cols / 128;
    if (N == 256 && K == 128) return launch_sb_gemm<4, 2, SBG_MASK_OPT_256>(R, X, ldx, W, ldw, nullptr, M, ldm, Y, ldy, 0, st, mt, part, colsum);
    if (N == 384 && K == 128) return launch_sb_gemm<4, 3, SBG_MASK_OPT_384>(R, X, ldx, W, ldw, nullptr, M, ldm, Y, ldy, 0, st, mt, part, colsum);
    if (N == 128 && K == 384) return launch_sb_gemm<12, 1, SBG_MASK_OPT_K384>(R, X, ldx, W, ldw, nullptr, M, ldm, Y, ldy, 0, st, mt, part, colsum);
    return MO_ERR_BAD_ARG;
}

int sb_gemm_masked_bits(int64_t R, int32_t N, int32_t K, const float *X, int64_t ldx, const float *W, int64_t ldw, const uint8_t *sign_bits,
                        int64_t ld_bytes, int32_t mask_cols, float *Y, int64_t ldy, float *colsum, void *workspace, void *stream) {
    if (R < 0 || !X || !W || !sign_bits || !Y || !colsum || !workspace || ldx < K || ldw < K || ldy < N || ld_bytes < N / 8) return MO_ERR_BAD_ARG;
    if (mask_cols < 0 || mask_cols > N || (mask_cols & 127)) return MO_ERR_BAD_ARG;
    if ((ldx & 3) || (ldw & 3) || (ldy & 3) || ((uintptr_t)X & 15) || ((uintptr_t)W & 15) || ((uintptr_t)Y & 15) || ((uintptr_t)colsum & 15) ||
        ((uintptr_t)workspace & 15))
        return MO_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (R == 0) return (int)hipMemsetAsync(colsum, 0, (size_t)N * 4, st);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    if (cus > 512) return MO_ERR_BAD_ARG;
    float *part = (float *)workspace;
    const float *m = (const float *)sign_bits;     // (the kernel reads it as bytes: OPT bit 16)
    const int mt = mask_cols / 128;
    if (N == 256 && K == 128) return launch_sb_gemm<4, 2, SBG_MASK_OPT_256 | 16>(R, X, ldx, W, ldw, nullptr, m, ld_bytes, Y, ldy, 0, st, mt, part, colsum);
    if (N == 384 && K == 128) return launch_sb_gemm<4, 3, SBG_MASK_OPT_384 | 16>(R, X, ldx, W, ldw, nullptr, m, ld_bytes, Y, ldy, 0, st, mt, part, colsum);
    if (N == 128 && K == 384) return launch_sb_gemm<12, 1, SBG_MASK_OPT_K384 | 16>(R, X, ldx, W, ldw, nullptr, m, ld_bytes, Y, ldy, 0, st, mt, part, colsum);
    return MO_ERR_BAD_ARG;
}

int sb_gemm_n128(int64_t R, int32_t K, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, int32_t relu, const float *addend,
                 int64_t lda, float *Y, int64_t ldy, void *stream) {
    return sb_gemm(R, 128, K, X, ldx, W, ldw, bias, relu, addend, lda, Y, ldy, stream);
}

int gru_seq_split_fwd_multi(int32_t n_nets, const mo_gru_seq_net *nets, int32_t T, int32_t B, int32_t H, int32_t gi_agents, void *stream) {
    if (n_nets < 1 || n_nets > MO_GRU_MAX_NETS || !nets || T < 1 || B < 1 || H != SBR_H || gi_agents < 0 || (gi_agents && B % gi_agents)) return MO_ERR_BAD_ARG;
    for (int k = 0; k < n_nets; k++) {
        const mo_gru_seq_net &m = nets[k];
        if (!m.gi || !m.w_hh || !m.b_hh || !m.h0 || !m.out) return MO_ERR_BAD_ARG;
        if ((((uintptr_t)m.gi | (uintptr_t)m.w_hh | (uintptr_t)m.b_hh | (uintptr_t)m.h0 | (uintptr_t)m.out | (uintptr_t)m.save) & 15)) return MO_ERR_BAD_ARG;
        if (m.B < 0 || m.B > B || (gi_agents && m.B % gi_agents)) return MO_ERR_BAD_ARG;
    }
    return launch_gru_seq_fwd_sb(n_nets, nets, T, B, gi_agents, (hipStream_t)stream);
}

int gru_seq_split_bwd_multi(int32_t n_nets, const mo_gru_seq_bwd_net *nets, int32_t T, int32_t B, int32_t H, int32_t gi_agents, void *stream) {
    if (n_nets < 1 || n_nets > MO_GRU_MAX_NETS || !nets || T < 1 || B < 1 || H != SBR_H || gi_agents < 0 || (gi_agents && B % gi_agents)) return MO_ERR_BAD_ARG;
    for (int k = 0; k < n_nets; k++) {
        const mo_gru_seq_bwd_net &m = nets[k];
        if (!m.dout || !m.save || !m.out || !m.h0 || !m.w_hh || !m.dgi || !m.dh0) return MO_ERR_BAD_ARG;
        if ((m.dgh == nullptr) == (m.dnr == nullptr)) return MO_ERR_BAD_ARG;        // exactly one of the two forms
        if ((m.db_ih || m.db_hh) && (!m.db_ih || !m.db_hh || !m.workspace)) return MO_ERR_BAD_ARG;
        if ((((uintptr_t)m.dout | (uintptr_t)m.save | (uintptr_t)m.out | (uintptr_t)m.h0 | (uintptr_t)m.dgi | (uintptr_t)m.dgh | (uintptr_t)m.dnr |
              (uintptr_t)m.dh0 | (uintptr_t)m.workspace) & 15)) return MO_ERR_BAD_ARG;
        if (m.B < 0 || m.B > B || (gi_agents && m.B % gi_agents)) return MO_ERR_BAD_ARG;
    }
    return launch_gru_seq_bwd_sb(n_nets, nets, T, B, gi_agents, (hipStream_t)stream);
}

}  // extern "C"
