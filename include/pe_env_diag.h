/*
 * pe_env_diag.h -- TEST-ONLY entry points of libpe_env.so (not part of the drop-in boundary of pe_env.h).
 * They expose, for the parity tests, the exact-arithmetic building blocks the tick kernel relies on.
 */
#ifndef PE_ENV_DIAG_H
#define PE_ENV_DIAG_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Device f64 primitives against the host (tests/test_env_gpu.py): out[5][n] = norm2(a,b) = sqrt(fma(b,b,a*a)), a / b,
 * round-half-even(a), a / c0 and b / c1 through the kernel's three-flop division by a host-known constant. */
int pe_diag_norm2(int32_t n, const double *a, const double *b, double *out, double c0, double c1, void *stream);
/* Host only (no GPU): the comparison threshold the tick kernel uses in place of sqrt(x) <= r (strict: sqrt(x) < r) ... */
double pe_diag_sq_threshold(double r, int32_t strict);
/* ... and the reciprocal it divides by a constant with (0 = the plain IEEE division is used). */
double pe_diag_div_reciprocal(double b);

#ifdef __cplusplus
}
#endif
#endif
