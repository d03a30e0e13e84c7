/*
 * CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see faer_oracle_impl.h).
 *
 * Plain-C restatement of faer 0.24.4's CPU algorithms for the hot path
 * (GEMM, triangular product, TRSM, LLT / LDLT, partial-pivot LU, Householder QR)
 * and for the SURVEY.md section 8f rows built so far (LU with full pivoting, QR
 * with column pivoting), instantiated for f64 and f32.  Built by oracle/Makefile into
 * oracle/libfaer_oracle.so and loaded through ctypes by oracle/oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (faer-rs_amd/) never links or calls it.
 *
 * Parity: UNPINNED at bit level (reference GEMM lives in unvendored crates,
 * Rust toolchain absent => reference cannot be run here); pinned at tolerance
 * level against the reference's own known-answer tests (qr/mod.rs:116-191,
 * reductions/norm_l2.rs:216-218), its matmul doctests, its property tests and
 * LAPACK (residuals; geqp3's permutation for the col-pivot QR) -- tests/test_oracle.py.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define T double
#define FN(x) x##_f64
#define FMA(a, b, c) fma((a), (b), (c))
#define SQRT(x) sqrt(x)
#define FABS(x) fabs(x)
#define HYPOT(a, b) hypot((a), (b))
#define TMIN DBL_MIN
#define TMAX (1.0 / DBL_MIN) /* faer-traits/src/lib.rs:2519-2521: max_positive = MIN_POSITIVE.recip() */
#define TEPS DBL_EPSILON
#include "faer_oracle_impl.h"
#undef T
#undef FN
#undef FMA
#undef SQRT
#undef FABS
#undef HYPOT
#undef TMIN
#undef TMAX
#undef TEPS

#define T float
#define FN(x) x##_f32
#define FMA(a, b, c) fmaf((a), (b), (c))
#define SQRT(x) sqrtf(x)
#define FABS(x) fabsf(x)
#define HYPOT(a, b) hypotf((a), (b))
#define TMIN FLT_MIN
#define TMAX (1.0f / FLT_MIN)
#define TEPS FLT_EPSILON
#include "faer_oracle_impl.h"
