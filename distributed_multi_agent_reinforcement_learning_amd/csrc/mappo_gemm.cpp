// mappo_gemm.cpp -- include/mappo_gemm.h on hipBLASLt (fp32 in, fp32 accumulate: v_mfma_f32_*_f32 kernels of the library).
//
// hipBLASLt is column-major.  The row-major product D (M x N, ldd) = A (M x K, lda) . W^T (W: N x K, ldw) is the column-major
// product D^T (N x M, ldd) = op(W) . A^T with W read as a K x N column-major matrix (ld = ldw, transposed) and A as a K x M
// column-major matrix (ld = lda, not transposed); the bias vector runs along the N rows of D^T, which is what the library's
// BIAS epilogues add.  An addend C (beta = 1) and the output D take any leading dimension -- the reason this file exists:
// torch's addmm epilogue path (at::_addmm_activation) only writes contiguous results.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <tuple>

#include "mappo_gemm.h"

namespace {

constexpr int64_t WORKSPACE_BYTES = 32ll << 20;

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr, d = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws = 0;
};

using Key = std::tuple<int64_t, int32_t, int32_t, int64_t, int64_t, int64_t, int64_t, int, int>;

std::mutex g_mu;
hipblasLtHandle_t g_handle = nullptr;
std::map<Key, Plan> g_plans;

int status_code(hipblasStatus_t s) { return s == HIPBLAS_STATUS_SUCCESS ? 0 : 1000 + (int)s; }

int make_plan(const Key &key, int64_t M, int32_t N, int32_t K, int64_t lda, int64_t ldw, int64_t ldc, int64_t ldd, int epi, Plan *out) {
    Plan p;
    hipblasStatus_t s = hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F);
    if (s != HIPBLAS_STATUS_SUCCESS) return status_code(s);
    const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof ta);
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof tb);
    const hipblasLtEpilogue_t e = (hipblasLtEpilogue_t)epi;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &e, sizeof e);
    if (epi == HIPBLASLT_EPILOGUE_BIAS || epi == HIPBLASLT_EPILOGUE_RELU_BIAS) {
        const hipDataType bt = HIP_R_32F;
        hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof bt);
    }
    if ((s = hipblasLtMatrixLayoutCreate(&p.a, HIP_R_32F, (uint64_t)K, (uint64_t)N, ldw)) != HIPBLAS_STATUS_SUCCESS) return status_code(s);
    if ((s = hipblasLtMatrixLayoutCreate(&p.b, HIP_R_32F, (uint64_t)K, (uint64_t)M, lda)) != HIPBLAS_STATUS_SUCCESS) return status_code(s);
    if ((s = hipblasLtMatrixLayoutCreate(&p.c, HIP_R_32F, (uint64_t)N, (uint64_t)M, ldc)) != HIPBLAS_STATUS_SUCCESS) return status_code(s);
    if ((s = hipblasLtMatrixLayoutCreate(&p.d, HIP_R_32F, (uint64_t)N, (uint64_t)M, ldd)) != HIPBLAS_STATUS_SUCCESS) return status_code(s);
    hipblasLtMatmulPreference_t pref;
    if ((s = hipblasLtMatmulPreferenceCreate(&pref)) != HIPBLAS_STATUS_SUCCESS) return status_code(s);
    const uint64_t ws = (uint64_t)WORKSPACE_BYTES;
    hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof ws);
    hipblasLtMatmulHeuristicResult_t res[4];
    int found = 0;
    s = hipblasLtMatmulAlgoGetHeuristic(g_handle, p.desc, p.a, p.b, p.c, p.d, pref, 4, res, &found);
    hipblasLtMatmulPreferenceDestroy(pref);
    if (s != HIPBLAS_STATUS_SUCCESS) return status_code(s);
    if (found < 1) return MO_GEMM_ERR_NO_ALGO;
    p.algo = res[0].algo;
    p.ws = res[0].workspaceSize;
    *out = p;
    g_plans[key] = p;
    return 0;
}

}  // namespace

extern "C" {

int64_t mo_gemm_workspace_bytes(void) { return WORKSPACE_BYTES; }

int mo_gemm_nt(int64_t M, int32_t N, int32_t K, const float *A, int64_t lda, const float *W, int64_t ldw, const float *bias, const float *C,
               int64_t ldc, int32_t relu, float *D, int64_t ldd, void *workspace, int64_t workspace_bytes, void *stream) {
    if (M < 1 || N < 1 || K < 1 || !A || !W || !D || lda < K || ldw < K || ldd < N || (C && ldc < N)) return MO_GEMM_ERR_BAD_ARG;
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)D | (uintptr_t)C | (uintptr_t)bias) & 15)) return MO_GEMM_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < WORKSPACE_BYTES) return MO_GEMM_ERR_BAD_ARG;
    const int epi = bias ? (relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS) : (relu ? HIPBLASLT_EPILOGUE_RELU : HIPBLASLT_EPILOGUE_DEFAULT);
    if (!C) ldc = ldd;
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_handle) {
        hipblasStatus_t s = hipblasLtCreate(&g_handle);
        if (s != HIPBLAS_STATUS_SUCCESS) return status_code(s);
    }
    const Key key{M, N, K, lda, ldw, ldc, ldd, epi, C ? 1 : 0};
    Plan plan;
    auto it = g_plans.find(key);
    if (it != g_plans.end()) plan = it->second;
    else {
        int rc = make_plan(key, M, N, K, lda, ldw, ldc, ldd, epi, &plan);
        if (rc) return rc;
    }
    // the bias pointer is per call (the descriptor is cached per shape): set it under the lock, right before the launch
    hipblasLtMatmulDescSetAttribute(plan.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof bias);
    const float alpha = 1.f, beta = C ? 1.f : 0.f;
    hipblasStatus_t s = hipblasLtMatmul(g_handle, plan.desc, &alpha, W, plan.a, A, plan.b, &beta, C ? C : D, plan.c, D, plan.d, &plan.algo, workspace,
                                        (size_t)workspace_bytes, (hipStream_t)stream);
    return status_code(s);
}

}  // extern "C"
