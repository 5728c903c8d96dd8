// mappo_split.hip -- the split-bf16 GEMM (and GRU sequence) kernels of libmappo_ops.so: a second translation unit so that a change
// to these kernels does not recompile csrc/mappo_ops.hip.  C ABI: include/mappo_ops.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "mappo_ops.h"
#include "sb_gemm.hpp"

extern "C" {

int sb_gemm(int64_t R, int32_t N, int32_t K, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, int32_t relu,
            const float *addend, int64_t lda, float *Y, int64_t ldy, void *stream) {
    if (R < 0 || !X || !W || !Y || ldx < K || ldw < K || ldy < N || (addend && lda < N)) return MO_ERR_BAD_ARG;
    if ((ldx & 3) || (ldw & 3) || (ldy & 3) || (lda & 3) || ((uintptr_t)X & 15) || ((uintptr_t)W & 15) || ((uintptr_t)Y & 15) || ((uintptr_t)addend & 15) ||
        ((uintptr_t)bias & 15))
        return MO_ERR_BAD_ARG;
    if (R == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (N == 128 && K == 128) return launch_sb_gemm_best<4, 1>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st);
    if (N == 128 && K == 256) return launch_sb_gemm_best<8, 1>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st);
    if (N == 128 && K == 384) return launch_sb_gemm_best<12, 1>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st);
    if (N == 256 && K == 128) return launch_sb_gemm_best<4, 2>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st);
    if (N == 384 && K == 128) return launch_sb_gemm_best<4, 3>(R, X, ldx, W, ldw, bias, addend, lda, Y, ldy, relu, st);
    return MO_ERR_BAD_ARG;
}

int sb_gemm_n128(int64_t R, int32_t K, const float *X, int64_t ldx, const float *W, int64_t ldw, const float *bias, int32_t relu, const float *addend,
                 int64_t lda, float *Y, int64_t ldy, void *stream) {
    return sb_gemm(R, 128, K, X, ldx, W, ldw, bias, relu, addend, lda, Y, ldy, stream);
}

}  // extern "C"
