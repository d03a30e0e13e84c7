/* A C client of libfaer_hip.so written against the REFERENCE's own header, faer-ffi/faer.h (which is not in this
 * repository: pass -I<reference>/faer-ffi; tests/test_cabi_client.py does, and skips when the reference is absent).
 *
 * 1. every repr(C) struct / enum of the boundary has the same size, field offsets and enumerator values in
 *    include/faer_hip.h as in faer.h (_Static_assert: checked at compile time);
 * 2. the program links: every `libfaer_v0_23_*` prototype it names, taken from faer.h, is exported;
 * 3. without arguments it runs the host-only entry points (params, scratch sizes, global par) -- no GPU needed;
 *    with `compute` it factors and solves small systems on the GPU through the faer.h prototypes alone and checks
 *    the residuals (tests -m gpu, when the binary was built where the reference is available).
 * This is the compiled stand-in for SURVEY.md section 8 rows a23 / a31 (an unchanged caller of the boundary). */
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "faer.h" /* the reference's header */
#define FAER_HIP_NO_FFI_PROTOTYPES
#include "faer_hip.h" /* ours: types + faer_hip_* runtime functions */

#define SAME_SIZE(A, B) _Static_assert(sizeof(A) == sizeof(B) && _Alignof(A) == _Alignof(B), "size/alignment of " #A " != " #B)
#define SAME_FIELD(A, B, F) _Static_assert(offsetof(A, F) == offsetof(B, F) && sizeof(((A *) 0)->F) == sizeof(((B *) 0)->F), "field " #F " of " #A " / " #B)

SAME_SIZE(FaerV0_24_MatRef, FaerMatRef);
SAME_FIELD(FaerV0_24_MatRef, FaerMatRef, ptr);
SAME_FIELD(FaerV0_24_MatRef, FaerMatRef, nrows);
SAME_FIELD(FaerV0_24_MatRef, FaerMatRef, ncols);
SAME_FIELD(FaerV0_24_MatRef, FaerMatRef, row_stride);
SAME_FIELD(FaerV0_24_MatRef, FaerMatRef, col_stride);
SAME_SIZE(FaerV0_24_MatMut, FaerMatMut);
SAME_FIELD(FaerV0_24_MatMut, FaerMatMut, ptr);
SAME_FIELD(FaerV0_24_MatMut, FaerMatMut, nrows);
SAME_FIELD(FaerV0_24_MatMut, FaerMatMut, ncols);
SAME_FIELD(FaerV0_24_MatMut, FaerMatMut, row_stride);
SAME_FIELD(FaerV0_24_MatMut, FaerMatMut, col_stride);
SAME_SIZE(FaerV0_24_VecRef, FaerVecRef);
SAME_FIELD(FaerV0_24_VecRef, FaerVecRef, ptr);
SAME_FIELD(FaerV0_24_VecRef, FaerVecRef, len);
SAME_FIELD(FaerV0_24_VecRef, FaerVecRef, stride);
SAME_SIZE(FaerV0_24_VecMut, FaerVecMut);
SAME_SIZE(FaerV0_24_SliceRef, FaerSliceRef);
SAME_FIELD(FaerV0_24_SliceRef, FaerSliceRef, ptr);
SAME_FIELD(FaerV0_24_SliceRef, FaerSliceRef, len);
SAME_SIZE(FaerV0_24_SliceMut, FaerSliceMut);
SAME_FIELD(FaerV0_24_SliceMut, FaerSliceMut, ptr);
SAME_FIELD(FaerV0_24_SliceMut, FaerSliceMut, len);
SAME_SIZE(FaerV0_24_Par, FaerPar);
SAME_FIELD(FaerV0_24_Par, FaerPar, tag);
SAME_FIELD(FaerV0_24_Par, FaerPar, nthreads);
SAME_SIZE(FaerV0_24_Layout, FaerLayout);
SAME_FIELD(FaerV0_24_Layout, FaerLayout, len_bytes);
SAME_FIELD(FaerV0_24_Layout, FaerLayout, align_bytes);
SAME_SIZE(FaerV0_24_MemAlloc, FaerMemAlloc);
SAME_FIELD(FaerV0_24_MemAlloc, FaerMemAlloc, ptr);
SAME_FIELD(FaerV0_24_MemAlloc, FaerMemAlloc, len_bytes);
SAME_SIZE(FaerV0_24_LltStatus, FaerLltStatus);
SAME_FIELD(FaerV0_24_LltStatus, FaerLltStatus, tag);
SAME_FIELD(FaerV0_24_LltStatus, FaerLltStatus, ok);
SAME_FIELD(FaerV0_24_LltStatus, FaerLltStatus, non_positive_pivot);
SAME_SIZE(FaerV0_24_LdltStatus, FaerLdltStatus);
SAME_FIELD(FaerV0_24_LdltStatus, FaerLdltStatus, tag);
SAME_FIELD(FaerV0_24_LdltStatus, FaerLdltStatus, ok);
SAME_FIELD(FaerV0_24_LdltStatus, FaerLdltStatus, zero_pivot);
SAME_SIZE(FaerV0_24_PartialPivLuStatus, FaerPartialPivLuStatus);
SAME_FIELD(FaerV0_24_PartialPivLuStatus, FaerPartialPivLuStatus, ok);
SAME_SIZE(FaerV0_24_FullPivLuStatus, FaerFullPivLuStatus);
SAME_FIELD(FaerV0_24_FullPivLuStatus, FaerFullPivLuStatus, ok);
SAME_SIZE(FaerV0_24_QrStatus, FaerQrStatus);
SAME_FIELD(FaerV0_24_QrStatus, FaerQrStatus, ok);
SAME_SIZE(FaerV0_24_ColPivQrStatus, FaerColPivQrStatus);
SAME_FIELD(FaerV0_24_ColPivQrStatus, FaerColPivQrStatus, ok);
SAME_SIZE(FaerV0_24_LltParams, FaerLltParams);
SAME_FIELD(FaerV0_24_LltParams, FaerLltParams, recursion_threshold);
SAME_FIELD(FaerV0_24_LltParams, FaerLltParams, block_size);
SAME_SIZE(FaerV0_24_LdltParams, FaerLdltParams);
SAME_SIZE(FaerV0_24_PartialPivLuParams, FaerPartialPivLuParams);
SAME_FIELD(FaerV0_24_PartialPivLuParams, FaerPartialPivLuParams, recursion_threshold);
SAME_FIELD(FaerV0_24_PartialPivLuParams, FaerPartialPivLuParams, block_size);
SAME_FIELD(FaerV0_24_PartialPivLuParams, FaerPartialPivLuParams, par_threshold);
SAME_SIZE(FaerV0_24_FullPivLuParams, FaerFullPivLuParams);
SAME_SIZE(FaerV0_24_QrParams, FaerQrParams);
SAME_FIELD(FaerV0_24_QrParams, FaerQrParams, blocking_threshold);
SAME_FIELD(FaerV0_24_QrParams, FaerQrParams, par_threshold);
SAME_SIZE(FaerV0_24_ColPivQrParams, FaerColPivQrParams);
SAME_SIZE(FaerV0_24_LltRegularization, FaerLltRegularization);
SAME_FIELD(FaerV0_24_LltRegularization, FaerLltRegularization, dynamic_regularization_delta);
SAME_FIELD(FaerV0_24_LltRegularization, FaerLltRegularization, dynamic_regularization_epsilon);
SAME_SIZE(FaerV0_24_LdltRegularization, FaerLdltRegularization);
SAME_FIELD(FaerV0_24_LdltRegularization, FaerLdltRegularization, dynamic_regularization_signs);
_Static_assert((int) FaerV0_24_Accum_Replace == (int) FaerAccum_Replace && (int) FaerV0_24_Accum_Add == (int) FaerAccum_Add, "Accum");
_Static_assert((int) FaerV0_24_Conj_No == (int) FaerConj_No && (int) FaerV0_24_Conj_Yes == (int) FaerConj_Yes, "Conj");
_Static_assert((int) FaerV0_24_ParTag_Seq == (int) FaerParTag_Seq && (int) FaerV0_24_ParTag_Rayon == (int) FaerParTag_Rayon, "ParTag");
_Static_assert((int) FaerV0_24_Block_Rectangular == (int) FaerBlock_Rectangular && (int) FaerV0_24_Block_TriangularLower == (int) FaerBlock_TriangularLower &&
		       (int) FaerV0_24_Block_UnitTriangularUpper == (int) FaerBlock_UnitTriangularUpper,
	       "Block");
_Static_assert((int) FaerV0_24_LltStatus_Ok == (int) FaerLltStatus_Ok && (int) FaerV0_24_LltStatus_NonPositivePivot == (int) FaerLltStatus_NonPositivePivot &&
		       (int) FaerV0_24_LltStatus_Unknown == (int) FaerLltStatus_Unknown,
	       "LltStatus tags");

static int fails = 0;
#define CHECK(cond)                                                                                                    \
	do {                                                                                                           \
		if (!(cond)) {                                                                                         \
			fprintf(stderr, "ref_client: FAILED %s (line %d)\n", #cond, __LINE__);                          \
			++fails;                                                                                       \
		}                                                                                                      \
	} while (0)

static double frand(void)
{
	static unsigned long long s = 88172645463325252ull;
	s ^= s << 13;
	s ^= s >> 7;
	s ^= s << 17;
	return (double) (s >> 11) / 9007199254740992.0 - 0.5;
}

static void host_only(void)
{
	FaerV0_24_Par seq = {FaerV0_24_ParTag_Seq, 0};
	/* params getters: the reference's Auto<T> defaults (cholesky/ldlt/factor.rs:705-714, lu/.../factor.rs:212-222) */
	FaerV0_24_LltParams lp = libfaer_v0_23_LltParams_f64();
	CHECK(lp.recursion_threshold == 64 && lp.block_size == 128);
	FaerV0_24_PartialPivLuParams up = libfaer_v0_23_PartialPivLuParams_f64();
	CHECK(up.recursion_threshold == 16 && up.block_size == 64);
	FaerV0_24_QrParams qp = libfaer_v0_23_QrParams_f32();
	CHECK(qp.blocking_threshold == 48 * 48);
	/* scratch conventions (linalg/mod.rs:38-54): llt dim scalars, lu min(dim, block) indices, qr bs x ncols scalars */
	FaerV0_24_Layout l = libfaer_v0_23_llt_factor_in_place_scratch_f64(1000, seq, lp);
	CHECK(l.len_bytes >= 1000 * sizeof(double) && l.align_bytes >= sizeof(double));
	l = libfaer_v0_23_partial_piv_lu_factor_in_place_scratch_u64_f64(500, 400, seq, up);
	CHECK(l.len_bytes >= 64 * sizeof(size_t));
	size_t bs = libfaer_v0_23_qr_recommended_block_size_f64(1000, 300);
	CHECK(bs >= 1 && bs <= 300);
	l = libfaer_v0_23_qr_factor_in_place_scratch_f64(1000, 300, bs, seq, libfaer_v0_23_QrParams_f64());
	CHECK(l.len_bytes >= bs * 300 * sizeof(double));
	/* global parallelism is stored and returned (lib.rs:2523-2543) */
	FaerV0_24_Par old = libfaer_v0_23_get_global_par();
	FaerV0_24_Par ray = {FaerV0_24_ParTag_Rayon, 7};
	libfaer_v0_23_set_global_par(ray);
	FaerV0_24_Par got = libfaer_v0_23_get_global_par();
	CHECK(got.tag == FaerV0_24_ParTag_Rayon && got.nthreads == 7);
	libfaer_v0_23_set_global_par(old);
	CHECK(faer_hip_version() != NULL);
}

/* column-major n x n helpers on the host; the library stages host operands itself (include/faer_hip.h, "Memory") */
static double resid_llt(const double *a, const double *l, size_t n)
{
	double worst = 0, amax = 0;
	for (size_t j = 0; j < n; ++j)
		for (size_t i = j; i < n; ++i) {
			double s = 0;
			for (size_t k = 0; k <= j; ++k)
				s += l[i + k * n] * l[j + k * n];
			worst = fmax(worst, fabs(s - a[i + j * n]));
			amax = fmax(amax, fabs(a[i + j * n]));
		}
	return worst / amax;
}

static void compute(void)
{
	const size_t n = 300, k = 5;
	FaerV0_24_Par seq = {FaerV0_24_ParTag_Seq, 0};
	FaerV0_24_MemAlloc nomem = {NULL, 0};
	double *g = malloc(n * n * sizeof(double)), *a = malloc(n * n * sizeof(double)), *l = malloc(n * n * sizeof(double));
	double *b = malloc(n * k * sizeof(double)), *x = malloc(n * k * sizeof(double));
	for (size_t i = 0; i < n * n; ++i)
		g[i] = frand();
	for (size_t i = 0; i < n * k; ++i)
		b[i] = frand();
	/* A = G G^T + n I through the library's own matmul (la::matmul::matmul, lib.rs:855) */
	const double one = 1.0;
	FaerV0_24_MatRef G = {g, n, n, 1, (ptrdiff_t) n}, Gt = {g, n, n, (ptrdiff_t) n, 1};
	FaerV0_24_MatMut A = {a, n, n, 1, (ptrdiff_t) n};
	libfaer_v0_23_matmul_f64(A, FaerV0_24_Accum_Replace, G, Gt, (const FaerV0_24_Scalar *) &one, seq);
	for (size_t i = 0; i < n; ++i)
		a[i + i * n] += (double) n;
	/* ---- llt: factor, solve */
	memcpy(l, a, n * n * sizeof(double));
	FaerV0_24_MatMut L = {l, n, n, 1, (ptrdiff_t) n};
	FaerV0_24_LltRegularization noreg = {NULL, NULL};
	FaerV0_24_LltStatus st = libfaer_v0_23_llt_factor_in_place_f64(L, noreg, seq, nomem, libfaer_v0_23_LltParams_f64());
	CHECK(st.tag == FaerV0_24_LltStatus_Ok && st.ok.dynamic_regularization_count == 0);
	CHECK(resid_llt(a, l, n) < 1e-13);
	memcpy(x, b, n * k * sizeof(double));
	FaerV0_24_MatRef Lr = {l, n, n, 1, (ptrdiff_t) n};
	FaerV0_24_MatMut X = {x, n, k, 1, (ptrdiff_t) n};
	libfaer_v0_23_llt_solve_in_place_f64(Lr, FaerV0_24_Conj_No, X, seq, nomem);
	double worst = 0;
	for (size_t c = 0; c < k; ++c)
		for (size_t i = 0; i < n; ++i) {
			double s = -b[i + c * n];
			for (size_t j = 0; j < n; ++j)
				s += a[i + j * n] * x[j + c * n];
			worst = fmax(worst, fabs(s));
		}
	CHECK(worst < 1e-10);
	/* a non positive pivot comes back BY VALUE with its index (LltStatus::NonPositivePivot) */
	memcpy(l, a, n * n * sizeof(double));
	l[17 + 17 * n] = -1.0;
	st = libfaer_v0_23_llt_factor_in_place_f64(L, noreg, seq, nomem, libfaer_v0_23_LltParams_f64());
	CHECK(st.tag == FaerV0_24_LltStatus_NonPositivePivot && st.non_positive_pivot.index == 17);
	/* ---- partial pivot LU of G: perm arrays are element counts (lib.rs:251-258), factor + solve */
	memcpy(l, g, n * n * sizeof(double));
	size_t *pf = malloc(n * sizeof(size_t)), *pb = malloc(n * sizeof(size_t));
	FaerV0_24_SliceMut PF = {pf, n}, PB = {pb, n};
	FaerV0_24_PartialPivLuStatus lst =
		libfaer_v0_23_partial_piv_lu_factor_in_place_u64_f64(L, PF, PB, seq, nomem, libfaer_v0_23_PartialPivLuParams_f64());
	CHECK(lst.tag == FaerV0_24_PartialPivLuStatus_Ok);
	for (size_t i = 0; i < n; ++i)
		CHECK(pf[i] < n && pb[pf[i]] == i);
	memcpy(x, b, n * k * sizeof(double));
	FaerV0_24_SliceRef PFr = {pf, n}, PBr = {pb, n};
	libfaer_v0_23_partial_piv_lu_solve_in_place_u64_f64(Lr, Lr, FaerV0_24_Conj_No, PFr, PBr, X, seq, nomem);
	worst = 0;
	for (size_t c = 0; c < k; ++c)
		for (size_t i = 0; i < n; ++i) {
			double s = -b[i + c * n];
			for (size_t j = 0; j < n; ++j)
				s += g[i + j * n] * x[j + c * n];
			worst = fmax(worst, fabs(s));
		}
	CHECK(worst < 1e-9);
	/* ---- QR of the leading n x 40 block, rank */
	const size_t qn = 40;
	size_t bs = libfaer_v0_23_qr_recommended_block_size_f64(n, qn);
	double *h = calloc(bs * qn, sizeof(double));
	memcpy(l, g, n * qn * sizeof(double));
	FaerV0_24_MatMut Q = {l, n, qn, 1, (ptrdiff_t) n}, H = {h, bs, qn, 1, (ptrdiff_t) bs};
	FaerV0_24_QrStatus qst = libfaer_v0_23_qr_factor_in_place_f64(Q, H, seq, nomem, libfaer_v0_23_QrParams_f64());
	CHECK(qst.tag == FaerV0_24_QrStatus_Ok && qst.ok.rank == qn);
	/* |R_jj| = norm of the j-th column of the running residual: here only the first, ||G e_0|| */
	double c0 = 0;
	for (size_t i = 0; i < n; ++i)
		c0 += g[i] * g[i];
	CHECK(fabs(fabs(l[0]) - sqrt(c0)) < 1e-12 * sqrt(c0));
	free(g), free(a), free(l), free(b), free(x), free(pf), free(pb), free(h);
}

int main(int argc, char **argv)
{
	host_only();
	if (argc > 1 && strcmp(argv[1], "compute") == 0) {
		if (faer_hip_device_count() < 1) {
			fprintf(stderr, "ref_client: no gfx950 device\n");
			return 2;
		}
		compute();
	}
	if (fails == 0)
		printf("ref_client ok (%s)\n", argc > 1 ? argv[1] : "host-only");
	return fails == 0 ? 0 : 1;
}
