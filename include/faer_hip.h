/*
 * faer_hip.h -- C ABI of libfaer_hip.so, the MI355X (gfx950) dense backend for faer.
 *
 * Two boundaries are exported:
 *
 *  1. INNER boundary: faer_hip_gemm() has the argument list of
 *     private_gemm_x86::gemm, the third-party kernel faer calls at
 *       faer/src/linalg/matmul/mod.rs:1373-1411            (DstKind::Full)
 *       faer/src/linalg/matmul/triangular.rs:641-680       (DstKind::Lower)
 *       faer/src/linalg/matmul/internal/mod.rs:143-201     (row/col idx + diag)
 *     minus the x86 `InstrSet` argument.  A 3-site #[cfg(feature = "hip")]
 *     patch in faer routes every O(n^3) flop of every decomposition here
 *     (INTEGRATION.md).
 *
 *  2. OUTER boundary: the f32/f64 subset of faer-ffi's C ABI
 *     (faer-ffi/src/lib.rs, generated header faer-ffi/faer.h) with identical
 *     repr(C) struct layouts and identical symbol names
 *     (libfaer_v0_23_<fn>_<dtype>, index-typed ones libfaer_v0_23_<fn>_<u32|u64>_<dtype>),
 *     so a C/C++ client of faer-ffi links against libfaer_hip.so unchanged and
 *     gets whole factorizations executed on the GPU.
 *
 * Pointers: every matrix / slice pointer may be HOST memory (a faer::Mat) or
 * DEVICE memory (hipMalloc / a torch tensor).  The library asks the HIP runtime
 * (hipPointerGetAttributes) and stages host operands through device buffers;
 * device operands are used in place with no copy.  Strides are in ELEMENTS and
 * may be negative or non-unit (faer/src/mat/mod.rs:7-13).
 *
 * Errors: like faer (which panics across extern "C" on precondition
 * violations, faer-ffi/src/lib.rs) a violated precondition or a HIP runtime
 * failure prints a message to stderr and abort()s; numerical failure is
 * reported by value through the *Status unions.  There is NO CPU fallback: if
 * no gfx950 device is usable every compute entry point aborts.
 */
#ifndef FAER_HIP_H
#define FAER_HIP_H

#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FAER_HIP_API __attribute__((visibility("default")))

/* ---- repr(C) structs: field-for-field faer-ffi/src/lib.rs:12-113 (faer.h:203-258) ---- */
typedef struct FaerMatRef { const void *ptr; size_t nrows; size_t ncols; ptrdiff_t row_stride; ptrdiff_t col_stride; } FaerMatRef;
typedef struct FaerMatMut { void *ptr; size_t nrows; size_t ncols; ptrdiff_t row_stride; ptrdiff_t col_stride; } FaerMatMut;
typedef struct FaerVecRef { const void *ptr; size_t len; ptrdiff_t stride; } FaerVecRef;
typedef struct FaerVecMut { void *ptr; size_t len; ptrdiff_t stride; } FaerVecMut;
/* len is an ELEMENT count (faer-ffi/src/lib.rs:251-258) */
typedef struct FaerSliceRef { const void *ptr; size_t len; } FaerSliceRef;
typedef struct FaerSliceMut { void *ptr; size_t len; } FaerSliceMut;

typedef enum FaerAccum { FaerAccum_Replace = 0, FaerAccum_Add = 1 } FaerAccum;          /* lib.rs:59-64 */
typedef enum FaerConj { FaerConj_No = 0, FaerConj_Yes = 1 } FaerConj;                   /* lib.rs:65-70 */
typedef enum FaerParTag { FaerParTag_Seq = 0, FaerParTag_Rayon = 1 } FaerParTag;        /* lib.rs:71-76 */
typedef struct FaerPar { FaerParTag tag; size_t nthreads; } FaerPar;                    /* lib.rs:77-82 */
typedef enum FaerBlock {                                                                 /* lib.rs:83-93 */
	FaerBlock_Rectangular = 0,
	FaerBlock_TriangularLower = 1,
	FaerBlock_TriangularUpper = 2,
	FaerBlock_StrictTriangularLower = 3,
	FaerBlock_StrictTriangularUpper = 4,
	FaerBlock_UnitTriangularLower = 5,
	FaerBlock_UnitTriangularUpper = 6
} FaerBlock;
typedef struct FaerLayout { size_t len_bytes; size_t align_bytes; } FaerLayout;          /* lib.rs:94-99 */
typedef struct FaerMemAlloc { void *ptr; size_t len_bytes; } FaerMemAlloc;               /* lib.rs:108-113 */

/* status unions: faer-ffi/src/lib.rs:552-595 + cerr! (:372-407), faer.h:383-469 */
typedef enum FaerLltStatus_Tag { FaerLltStatus_Ok = 0, FaerLltStatus_NonPositivePivot = 1, FaerLltStatus_Unknown = 2 } FaerLltStatus_Tag;
typedef struct FaerLltStatus {
	FaerLltStatus_Tag tag;
	union {
		struct { size_t dynamic_regularization_count; } ok;
		struct { size_t index; } non_positive_pivot;
	};
} FaerLltStatus;
typedef enum FaerPartialPivLuStatus_Tag { FaerPartialPivLuStatus_Ok = 0, FaerPartialPivLuStatus_Unknown = 1 } FaerPartialPivLuStatus_Tag;
typedef struct FaerPartialPivLuStatus {
	FaerPartialPivLuStatus_Tag tag;
	union { struct { size_t transposition_count; } ok; };
} FaerPartialPivLuStatus;
typedef enum FaerQrStatus_Tag { FaerQrStatus_Ok = 0, FaerQrStatus_Unknown = 1 } FaerQrStatus_Tag;
typedef struct FaerQrStatus {
	FaerQrStatus_Tag tag;
	union { struct { size_t rank; } ok; };
} FaerQrStatus;

/* params: faer-ffi/src/lib.rs:650-690 (cparams!) */
typedef struct FaerLltParams { size_t recursion_threshold; size_t block_size; } FaerLltParams;
typedef struct FaerLdltParams { size_t recursion_threshold; size_t block_size; } FaerLdltParams; /* lib.rs:660-664 */
/* lib.rs:572-575, faer.h:346-366: Ok{dynamic_regularization_count} | ZeroPivot{index} | Unknown */
typedef enum FaerLdltStatus_Tag { FaerLdltStatus_Ok = 0, FaerLdltStatus_ZeroPivot = 1, FaerLdltStatus_Unknown = 2 } FaerLdltStatus_Tag;
typedef struct FaerLdltStatus {
	FaerLdltStatus_Tag tag;
	union {
		struct { size_t dynamic_regularization_count; } ok;
		struct { size_t index; } zero_pivot;
	};
} FaerLdltStatus;
typedef struct FaerPartialPivLuParams { size_t recursion_threshold; size_t block_size; size_t par_threshold; } FaerPartialPivLuParams;
typedef struct FaerFullPivLuParams { size_t par_threshold; } FaerFullPivLuParams;
typedef struct FaerColPivQrParams { size_t blocking_threshold; size_t par_threshold; } FaerColPivQrParams;
typedef enum FaerColPivQrStatus_Tag { FaerColPivQrStatus_Ok = 0, FaerColPivQrStatus_Unknown = 1 } FaerColPivQrStatus_Tag;
typedef struct FaerColPivQrStatus {
	FaerColPivQrStatus_Tag tag;
	union {
		struct { size_t transposition_count; } ok;
	};
} FaerColPivQrStatus;
typedef enum FaerFullPivLuStatus_Tag { FaerFullPivLuStatus_Ok = 0, FaerFullPivLuStatus_Unknown = 1 } FaerFullPivLuStatus_Tag;
typedef struct FaerFullPivLuStatus {
	FaerFullPivLuStatus_Tag tag;
	union {
		struct { size_t transposition_count; } ok;
	};
} FaerFullPivLuStatus;
typedef struct FaerQrParams { size_t blocking_threshold; size_t par_threshold; } FaerQrParams;
/* faer-ffi/src/lib.rs:796-801; pointers to a real scalar of the matrix dtype (HOST memory), NULL == 0 */
typedef struct FaerLltRegularization { const void *dynamic_regularization_delta; const void *dynamic_regularization_epsilon; } FaerLltRegularization;
/* lib.rs:820-828: signs is a slice of i8 (HOST memory, `dim` entries) or a null ptr */
typedef struct FaerLdltRegularization { const void *dynamic_regularization_delta; const void *dynamic_regularization_epsilon; FaerSliceMut dynamic_regularization_signs; } FaerLdltRegularization;

/* ---------------------------------------------------------------------------------------------
 * 1. INNER boundary -- replaces private_gemm_x86::gemm (call sites above).
 * --------------------------------------------------------------------------------------------- */
typedef enum FaerHipDType { FaerHipDType_F32 = 0, FaerHipDType_F64 = 1, FaerHipDType_C32 = 2, FaerHipDType_C64 = 3 } FaerHipDType;
typedef enum FaerHipIType { FaerHipIType_U32 = 0, FaerHipIType_U64 = 1 } FaerHipIType;
typedef enum FaerHipDstKind { FaerHipDstKind_Full = 0, FaerHipDstKind_Lower = 1, FaerHipDstKind_Upper = 2 } FaerHipDstKind;

/* dst[row_idx[i], col_idx[j]] (or dst[i,j] when the index arrays are NULL), restricted to the
 * Full / Lower (i>=j) / Upper (i<=j) part, <- [dst +] alpha * lhs * diag(diag) * rhs.
 * Accum_Replace never reads dst.  F32/F64 only (C32/C64 abort: out of the north-star scope).
 * n_threads is accepted for signature compatibility and ignored.  Thread safe. */
FAER_HIP_API void faer_hip_gemm(FaerHipDType dtype, FaerHipIType itype, size_t m, size_t n, size_t k,
                                void *dst, ptrdiff_t dst_rs, ptrdiff_t dst_cs,
                                const void *row_idx, const void *col_idx,
                                FaerHipDstKind dst_kind, FaerAccum accum,
                                const void *lhs, ptrdiff_t lhs_rs, ptrdiff_t lhs_cs, bool conj_lhs,
                                const void *diag, ptrdiff_t diag_stride,
                                const void *rhs, ptrdiff_t rhs_rs, ptrdiff_t rhs_cs, bool conj_rhs,
                                const void *alpha, size_t n_threads);

/* ---------------------------------------------------------------------------------------------
 * 2. OUTER boundary -- faer-ffi symbol-compatible entry points (f32 / f64).
 *    Each comment cites the Rust function in faer-ffi/src/lib.rs it replaces.
 * --------------------------------------------------------------------------------------------- */
/* A client that includes the reference's own faer-ffi/faer.h for these prototypes (tests/cabi_client/) defines
 * FAER_HIP_NO_FFI_PROTOTYPES before including this header and gets the types and the faer_hip_* functions only
 * (two declarations of one symbol with layout-identical but differently named struct types do not mix in C). */
#ifndef FAER_HIP_NO_FFI_PROTOTYPES
/* lib.rs:855-871  la::matmul::matmul */
FAER_HIP_API void libfaer_v0_23_matmul_f64(FaerMatMut C, FaerAccum accum, FaerMatRef A, FaerMatRef B, const void *alpha, FaerPar par);
FAER_HIP_API void libfaer_v0_23_matmul_f32(FaerMatMut C, FaerAccum accum, FaerMatRef A, FaerMatRef B, const void *alpha, FaerPar par);
/* lib.rs:872-895  la::matmul::triangular::matmul */
FAER_HIP_API void libfaer_v0_23_matmul_triangular_f64(FaerMatMut C, FaerBlock C_block, FaerAccum accum, FaerMatRef A, FaerBlock A_block, FaerMatRef B, FaerBlock B_block, const void *alpha, FaerPar par);
FAER_HIP_API void libfaer_v0_23_matmul_triangular_f32(FaerMatMut C, FaerBlock C_block, FaerAccum accum, FaerMatRef A, FaerBlock A_block, FaerMatRef B, FaerBlock B_block, const void *alpha, FaerPar par);
/* lib.rs:896-937  la::triangular_solve::solve_{,unit_}{lower,upper}_triangular_in_place_with_conj */
FAER_HIP_API void libfaer_v0_23_solve_triangular_lower_in_place_f64(FaerMatRef L, FaerConj L_conj, FaerMatMut rhs, FaerPar par);
FAER_HIP_API void libfaer_v0_23_solve_triangular_lower_in_place_f32(FaerMatRef L, FaerConj L_conj, FaerMatMut rhs, FaerPar par);
FAER_HIP_API void libfaer_v0_23_solve_triangular_upper_in_place_f64(FaerMatRef U, FaerConj U_conj, FaerMatMut rhs, FaerPar par);
FAER_HIP_API void libfaer_v0_23_solve_triangular_upper_in_place_f32(FaerMatRef U, FaerConj U_conj, FaerMatMut rhs, FaerPar par);
FAER_HIP_API void libfaer_v0_23_solve_unit_triangular_lower_in_place_f64(FaerMatRef L, FaerConj L_conj, FaerMatMut rhs, FaerPar par);
FAER_HIP_API void libfaer_v0_23_solve_unit_triangular_lower_in_place_f32(FaerMatRef L, FaerConj L_conj, FaerMatMut rhs, FaerPar par);
FAER_HIP_API void libfaer_v0_23_solve_unit_triangular_upper_in_place_f64(FaerMatRef U, FaerConj U_conj, FaerMatMut rhs, FaerPar par);
FAER_HIP_API void libfaer_v0_23_solve_unit_triangular_upper_in_place_f32(FaerMatRef U, FaerConj U_conj, FaerMatMut rhs, FaerPar par);

/* lib.rs:650-655 (cparams! getter)  Auto<T> for LltParams: {64, 128} (cholesky/ldlt/factor.rs:705-714) */
FAER_HIP_API FaerLltParams libfaer_v0_23_LltParams_f64(void);
FAER_HIP_API FaerLltParams libfaer_v0_23_LltParams_f32(void);
/* lib.rs:984-995  cholesky_in_place_scratch (cholesky/llt/factor.rs:58-66): dim scalars */
FAER_HIP_API FaerLayout libfaer_v0_23_llt_factor_in_place_scratch_f64(size_t dim, FaerPar par, FaerLltParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_llt_factor_in_place_scratch_f32(size_t dim, FaerPar par, FaerLltParams params);
/* lib.rs:996-1011  cholesky_in_place (cholesky/llt/factor.rs:67-97) */
FAER_HIP_API FaerLltStatus libfaer_v0_23_llt_factor_in_place_f64(FaerMatMut A, FaerLltRegularization regularization, FaerPar par, FaerMemAlloc mem, FaerLltParams params);
FAER_HIP_API FaerLltStatus libfaer_v0_23_llt_factor_in_place_f32(FaerMatMut A, FaerLltRegularization regularization, FaerPar par, FaerMemAlloc mem, FaerLltParams params);
/* lib.rs:1012-1040  llt::solve::solve_in_place_with_conj (cholesky/llt/solve.rs:12-35) */
FAER_HIP_API FaerLayout libfaer_v0_23_llt_solve_in_place_scratch_f64(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API FaerLayout libfaer_v0_23_llt_solve_in_place_scratch_f32(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_llt_solve_in_place_f64(FaerMatRef L, FaerConj A_conj, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API void libfaer_v0_23_llt_solve_in_place_f32(FaerMatRef L, FaerConj A_conj, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);

/* lib.rs:679-685  Auto<T> for PartialPivLuParams: {16, 64, 128*128} (lu/partial_pivoting/factor.rs:212-222) */
FAER_HIP_API FaerPartialPivLuParams libfaer_v0_23_PartialPivLuParams_f64(void);
FAER_HIP_API FaerPartialPivLuParams libfaer_v0_23_PartialPivLuParams_f32(void);
/* lib.rs:1952-1965  lu_in_place_scratch (lu/partial_pivoting/factor.rs:224-233): min(dim, block_size) indices.
 * NB the ffi passes (dim, block_size) as (nrows, ncols). */
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_factor_in_place_scratch_u32_f64(size_t dim, size_t block_size, FaerPar par, FaerPartialPivLuParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_factor_in_place_scratch_u64_f64(size_t dim, size_t block_size, FaerPar par, FaerPartialPivLuParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_factor_in_place_scratch_u32_f32(size_t dim, size_t block_size, FaerPar par, FaerPartialPivLuParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_factor_in_place_scratch_u64_f32(size_t dim, size_t block_size, FaerPar par, FaerPartialPivLuParams params);
/* lib.rs:1966-1984  lu_in_place (lu/partial_pivoting/factor.rs:234-295); perm_fwd[i] = source row of row i of P*A */
FAER_HIP_API FaerPartialPivLuStatus libfaer_v0_23_partial_piv_lu_factor_in_place_u32_f64(FaerMatMut A, FaerSliceMut perm_fwd, FaerSliceMut perm_bwd, FaerPar par, FaerMemAlloc mem, FaerPartialPivLuParams params);
FAER_HIP_API FaerPartialPivLuStatus libfaer_v0_23_partial_piv_lu_factor_in_place_u64_f64(FaerMatMut A, FaerSliceMut perm_fwd, FaerSliceMut perm_bwd, FaerPar par, FaerMemAlloc mem, FaerPartialPivLuParams params);
FAER_HIP_API FaerPartialPivLuStatus libfaer_v0_23_partial_piv_lu_factor_in_place_u32_f32(FaerMatMut A, FaerSliceMut perm_fwd, FaerSliceMut perm_bwd, FaerPar par, FaerMemAlloc mem, FaerPartialPivLuParams params);
FAER_HIP_API FaerPartialPivLuStatus libfaer_v0_23_partial_piv_lu_factor_in_place_u64_f32(FaerMatMut A, FaerSliceMut perm_fwd, FaerSliceMut perm_bwd, FaerPar par, FaerMemAlloc mem, FaerPartialPivLuParams params);

/* lib.rs:667-671  Auto<T> for QrParams: {48*48, 192*256} (qr/no_pivoting/factor.rs:127-136) */
FAER_HIP_API FaerQrParams libfaer_v0_23_QrParams_f64(void);
FAER_HIP_API FaerQrParams libfaer_v0_23_QrParams_f32(void);
/* lib.rs:1520-1527  recommended_block_size (qr/no_pivoting/factor.rs:91-116) */
FAER_HIP_API size_t libfaer_v0_23_qr_recommended_block_size_f64(size_t nrows, size_t ncols);
FAER_HIP_API size_t libfaer_v0_23_qr_recommended_block_size_f32(size_t nrows, size_t ncols);
/* lib.rs:1528-1543  qr_in_place_scratch (qr/no_pivoting/factor.rs:305-316): block_size x ncols scalars */
FAER_HIP_API FaerLayout libfaer_v0_23_qr_factor_in_place_scratch_f64(size_t nrows, size_t ncols, size_t block_size, FaerPar par, FaerQrParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_qr_factor_in_place_scratch_f32(size_t nrows, size_t ncols, size_t block_size, FaerPar par, FaerQrParams params);
/* lib.rs:1544-1559  qr_in_place (qr/no_pivoting/factor.rs:258-301); Q_coeff is block_size x min(m,n) */
FAER_HIP_API FaerQrStatus libfaer_v0_23_qr_factor_in_place_f64(FaerMatMut A, FaerMatMut Q_coeff, FaerPar par, FaerMemAlloc mem, FaerQrParams params);
FAER_HIP_API FaerQrStatus libfaer_v0_23_qr_factor_in_place_f32(FaerMatMut A, FaerMatMut Q_coeff, FaerPar par, FaerMemAlloc mem, FaerQrParams params);
/* lib.rs:1423-1470  apply_block_householder_sequence_{,transpose_}on_the_left_in_place_with_conj (householder.rs:724-808) */
FAER_HIP_API FaerLayout libfaer_v0_23_apply_householder_on_the_left_scratch_f64(size_t dim, size_t block_size, size_t rhs_ncols);
FAER_HIP_API FaerLayout libfaer_v0_23_apply_householder_on_the_left_scratch_f32(size_t dim, size_t block_size, size_t rhs_ncols);
FAER_HIP_API void libfaer_v0_23_apply_householder_on_the_left_f64(FaerMatRef householder_basis, FaerMatRef householder_factor, FaerConj householder_conj, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API void libfaer_v0_23_apply_householder_on_the_left_f32(FaerMatRef householder_basis, FaerMatRef householder_factor, FaerConj householder_conj, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_apply_householder_transpose_on_the_left_scratch_f64(size_t dim, size_t block_size, size_t rhs_ncols);
FAER_HIP_API FaerLayout libfaer_v0_23_apply_householder_transpose_on_the_left_scratch_f32(size_t dim, size_t block_size, size_t rhs_ncols);
FAER_HIP_API void libfaer_v0_23_apply_householder_transpose_on_the_left_f64(FaerMatRef householder_basis, FaerMatRef householder_factor, FaerConj householder_conj, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API void libfaer_v0_23_apply_householder_transpose_on_the_left_f32(FaerMatRef householder_basis, FaerMatRef householder_factor, FaerConj householder_conj, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);


/* lib.rs:1985-2056  lu::partial_pivoting::solve::{solve_in_place, solve_transpose_in_place}_with_conj
 * (lu/partial_pivoting/solve.rs:20-80): rhs <- A^-1 rhs = U^-1 L^-1 P rhs, resp. rhs <- A^-T rhs.
 * perm slices are HOST memory with `dim` entries (element count, see SURVEY.md section 8b caveat iii). */
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_solve_in_place_scratch_u32_f64(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_solve_in_place_u32_f64(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_solve_transpose_in_place_scratch_u32_f64(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_solve_transpose_in_place_u32_f64(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_solve_in_place_scratch_u32_f32(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_solve_in_place_u32_f32(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_solve_transpose_in_place_scratch_u32_f32(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_solve_transpose_in_place_u32_f32(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_solve_in_place_scratch_u64_f64(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_solve_in_place_u64_f64(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_solve_transpose_in_place_scratch_u64_f64(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_solve_transpose_in_place_u64_f64(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_solve_in_place_scratch_u64_f32(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_solve_in_place_u64_f32(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_solve_transpose_in_place_scratch_u64_f32(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_solve_transpose_in_place_u64_f32(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
/* lib.rs:1560-1660  qr::no_pivoting::solve::{solve_in_place, solve_transpose_in_place, solve_lstsq_in_place}_with_conj
 * (qr/no_pivoting/solve.rs:38-175): rhs <- R^-1 Q^H rhs (square or least squares: the solution is the top
 * ncols rows of rhs), resp. rhs <- Q R^-T rhs. */
FAER_HIP_API FaerLayout libfaer_v0_23_qr_solve_in_place_scratch_f64(size_t dim, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_qr_solve_in_place_f64(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_qr_solve_transpose_in_place_scratch_f64(size_t dim, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_qr_solve_transpose_in_place_f64(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_qr_solve_lstsq_in_place_scratch_f64(size_t nrows, size_t ncols, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_qr_solve_lstsq_in_place_f64(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_qr_solve_in_place_scratch_f32(size_t dim, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_qr_solve_in_place_f32(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_qr_solve_transpose_in_place_scratch_f32(size_t dim, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_qr_solve_transpose_in_place_f32(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_qr_solve_lstsq_in_place_scratch_f32(size_t nrows, size_t ncols, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_qr_solve_lstsq_in_place_f32(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);


/* ---- next scope row: L D L^T without pivoting ---------------------------------
 * lib.rs:1189-1217  ldlt::factor::cholesky_in_place (cholesky/ldlt/factor.rs:742-800): unit lower L strictly below
 * the diagonal of A, D on it; lib.rs:1218-1250  ldlt::solve::solve_in_place_with_conj (cholesky/ldlt/solve.rs:12-50) */
FAER_HIP_API FaerLdltParams libfaer_v0_23_LdltParams_f64(void);
FAER_HIP_API FaerLdltParams libfaer_v0_23_LdltParams_f32(void);
FAER_HIP_API FaerLayout libfaer_v0_23_ldlt_factor_in_place_scratch_f64(size_t dim, FaerPar par, FaerLdltParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_ldlt_factor_in_place_scratch_f32(size_t dim, FaerPar par, FaerLdltParams params);
FAER_HIP_API FaerLdltStatus libfaer_v0_23_ldlt_factor_in_place_f64(FaerMatMut A, FaerLdltRegularization regularization, FaerPar par, FaerMemAlloc mem, FaerLdltParams params);
FAER_HIP_API FaerLdltStatus libfaer_v0_23_ldlt_factor_in_place_f32(FaerMatMut A, FaerLdltRegularization regularization, FaerPar par, FaerMemAlloc mem, FaerLdltParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_ldlt_solve_in_place_scratch_f64(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API FaerLayout libfaer_v0_23_ldlt_solve_in_place_scratch_f32(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_ldlt_solve_in_place_f64(FaerMatRef L, FaerVecRef D, FaerConj A_conj, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API void libfaer_v0_23_ldlt_solve_in_place_f32(FaerMatRef L, FaerVecRef D, FaerConj A_conj, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);

/* lib.rs:2523-2543  get/set_global_parallelism (faer/src/lib.rs:1107-1150).  Stored and returned only:
 * the GPU backend has no host thread pool. */
FAER_HIP_API FaerPar libfaer_v0_23_get_global_par(void);
FAER_HIP_API void libfaer_v0_23_set_global_par(FaerPar par);

/* ---------------------------------------------------------------------------------------------
 * 2b. Triangular inverse, reconstruct / inverse of the factorizations, reflectors applied on the right
 *     (faer-ffi/src/lib.rs:938-983, :1039-1075, :1247-1288, :1661-1720, :2071-2124, :1471-1518; f32/f64 subset).
 *     Same argument lists as the reference; the `mem` scratch is accepted and ignored (device scratch is
 *     pooled inside the library).  Triangular outputs leave the other triangle of `out` untouched.
 * --------------------------------------------------------------------------------------------- */
FAER_HIP_API void libfaer_v0_23_inverse_triangular_lower_in_place_f64(FaerMatMut T_inv, FaerMatRef T, FaerPar par);
FAER_HIP_API void libfaer_v0_23_inverse_triangular_upper_in_place_f64(FaerMatMut T_inv, FaerMatRef T, FaerPar par);
FAER_HIP_API void libfaer_v0_23_inverse_unit_triangular_lower_in_place_f64(FaerMatMut T_inv, FaerMatRef T, FaerPar par);
FAER_HIP_API void libfaer_v0_23_inverse_unit_triangular_upper_in_place_f64(FaerMatMut T_inv, FaerMatRef T, FaerPar par);
FAER_HIP_API FaerLayout libfaer_v0_23_llt_reconstruct_scratch_f64(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_llt_reconstruct_f64(FaerMatMut A, FaerMatRef L, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_llt_inverse_scratch_f64(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_llt_inverse_f64(FaerMatMut A_inv, FaerMatRef L, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_ldlt_reconstruct_scratch_f64(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_ldlt_reconstruct_f64(FaerMatMut A, FaerMatRef L, FaerVecRef D, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_ldlt_inverse_scratch_f64(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_ldlt_inverse_f64(FaerMatMut A_inv, FaerMatRef L, FaerVecRef D, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_reconstruct_scratch_u32_f64(size_t nrows, size_t ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_reconstruct_u32_f64(FaerMatMut A, FaerMatRef L, FaerMatRef U, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_inverse_scratch_u32_f64(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_inverse_u32_f64(FaerMatMut A_inv, FaerMatRef L, FaerMatRef U, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_reconstruct_scratch_u64_f64(size_t nrows, size_t ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_reconstruct_u64_f64(FaerMatMut A, FaerMatRef L, FaerMatRef U, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_inverse_scratch_u64_f64(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_inverse_u64_f64(FaerMatMut A_inv, FaerMatRef L, FaerMatRef U, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_qr_reconstruct_scratch_f64(size_t nrows, size_t ncols, size_t block_size, FaerPar par);
FAER_HIP_API void libfaer_v0_23_qr_reconstruct_f64(FaerMatMut A, FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_qr_inverse_scratch_f64(size_t dim, size_t block_size, FaerPar par);
FAER_HIP_API void libfaer_v0_23_qr_inverse_f64(FaerMatMut A_inv, FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_apply_householder_on_the_right_scratch_f64(size_t dim, size_t block_size, size_t rhs_nrows);
FAER_HIP_API void libfaer_v0_23_apply_householder_on_the_right_f64(FaerMatRef householder_basis, FaerMatRef householder_factor, FaerConj conj, FaerMatMut matrix, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_apply_householder_transpose_on_the_right_scratch_f64(size_t dim, size_t block_size, size_t rhs_nrows);
FAER_HIP_API void libfaer_v0_23_apply_householder_transpose_on_the_right_f64(FaerMatRef householder_basis, FaerMatRef householder_factor, FaerConj conj, FaerMatMut matrix, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API void libfaer_v0_23_inverse_triangular_lower_in_place_f32(FaerMatMut T_inv, FaerMatRef T, FaerPar par);
FAER_HIP_API void libfaer_v0_23_inverse_triangular_upper_in_place_f32(FaerMatMut T_inv, FaerMatRef T, FaerPar par);
FAER_HIP_API void libfaer_v0_23_inverse_unit_triangular_lower_in_place_f32(FaerMatMut T_inv, FaerMatRef T, FaerPar par);
FAER_HIP_API void libfaer_v0_23_inverse_unit_triangular_upper_in_place_f32(FaerMatMut T_inv, FaerMatRef T, FaerPar par);
FAER_HIP_API FaerLayout libfaer_v0_23_llt_reconstruct_scratch_f32(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_llt_reconstruct_f32(FaerMatMut A, FaerMatRef L, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_llt_inverse_scratch_f32(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_llt_inverse_f32(FaerMatMut A_inv, FaerMatRef L, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_ldlt_reconstruct_scratch_f32(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_ldlt_reconstruct_f32(FaerMatMut A, FaerMatRef L, FaerVecRef D, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_ldlt_inverse_scratch_f32(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_ldlt_inverse_f32(FaerMatMut A_inv, FaerMatRef L, FaerVecRef D, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_reconstruct_scratch_u32_f32(size_t nrows, size_t ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_reconstruct_u32_f32(FaerMatMut A, FaerMatRef L, FaerMatRef U, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_inverse_scratch_u32_f32(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_inverse_u32_f32(FaerMatMut A_inv, FaerMatRef L, FaerMatRef U, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_reconstruct_scratch_u64_f32(size_t nrows, size_t ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_reconstruct_u64_f32(FaerMatMut A, FaerMatRef L, FaerMatRef U, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_partial_piv_lu_inverse_scratch_u64_f32(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_partial_piv_lu_inverse_u64_f32(FaerMatMut A_inv, FaerMatRef L, FaerMatRef U, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_qr_reconstruct_scratch_f32(size_t nrows, size_t ncols, size_t block_size, FaerPar par);
FAER_HIP_API void libfaer_v0_23_qr_reconstruct_f32(FaerMatMut A, FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_qr_inverse_scratch_f32(size_t dim, size_t block_size, FaerPar par);
FAER_HIP_API void libfaer_v0_23_qr_inverse_f32(FaerMatMut A_inv, FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_apply_householder_on_the_right_scratch_f32(size_t dim, size_t block_size, size_t rhs_nrows);
FAER_HIP_API void libfaer_v0_23_apply_householder_on_the_right_f32(FaerMatRef householder_basis, FaerMatRef householder_factor, FaerConj conj, FaerMatMut matrix, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_apply_householder_transpose_on_the_right_scratch_f32(size_t dim, size_t block_size, size_t rhs_nrows);
FAER_HIP_API void libfaer_v0_23_apply_householder_transpose_on_the_right_f32(FaerMatRef householder_basis, FaerMatRef householder_factor, FaerConj conj, FaerMatMut matrix, FaerPar par, FaerMemAlloc mem);

/* ---------------------------------------------------------------------------------------------
 * 2c. LU with full pivoting (faer-ffi/src/lib.rs:2125-2260; lu/full_pivoting/factor.rs, solve.rs): a level-2,
 *     HBM-bound algorithm -- every step is one fused "rank-1 update + search of the next pivot" pass (csrc/fplu.hip).
 *     Permutation slices are HOST memory (nrows / ncols entries).
 * --------------------------------------------------------------------------------------------- */
FAER_HIP_API FaerFullPivLuParams libfaer_v0_23_FullPivLuParams_f64(void);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_factor_in_place_scratch_u32_f64(size_t dim, size_t block_size, FaerPar par, FaerFullPivLuParams params);
FAER_HIP_API FaerFullPivLuStatus libfaer_v0_23_full_piv_lu_factor_in_place_u32_f64(FaerMatMut A, FaerSliceMut row_perm_fwd, FaerSliceMut row_perm_bwd, FaerSliceMut col_perm_fwd, FaerSliceMut col_perm_bwd, FaerPar par, FaerMemAlloc mem, FaerFullPivLuParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_solve_in_place_scratch_u32_f64(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_solve_in_place_u32_f64(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_solve_transpose_in_place_scratch_u32_f64(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_solve_transpose_in_place_u32_f64(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_factor_in_place_scratch_u64_f64(size_t dim, size_t block_size, FaerPar par, FaerFullPivLuParams params);
FAER_HIP_API FaerFullPivLuStatus libfaer_v0_23_full_piv_lu_factor_in_place_u64_f64(FaerMatMut A, FaerSliceMut row_perm_fwd, FaerSliceMut row_perm_bwd, FaerSliceMut col_perm_fwd, FaerSliceMut col_perm_bwd, FaerPar par, FaerMemAlloc mem, FaerFullPivLuParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_solve_in_place_scratch_u64_f64(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_solve_in_place_u64_f64(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_solve_transpose_in_place_scratch_u64_f64(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_solve_transpose_in_place_u64_f64(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerFullPivLuParams libfaer_v0_23_FullPivLuParams_f32(void);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_factor_in_place_scratch_u32_f32(size_t dim, size_t block_size, FaerPar par, FaerFullPivLuParams params);
FAER_HIP_API FaerFullPivLuStatus libfaer_v0_23_full_piv_lu_factor_in_place_u32_f32(FaerMatMut A, FaerSliceMut row_perm_fwd, FaerSliceMut row_perm_bwd, FaerSliceMut col_perm_fwd, FaerSliceMut col_perm_bwd, FaerPar par, FaerMemAlloc mem, FaerFullPivLuParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_solve_in_place_scratch_u32_f32(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_solve_in_place_u32_f32(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_solve_transpose_in_place_scratch_u32_f32(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_solve_transpose_in_place_u32_f32(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_factor_in_place_scratch_u64_f32(size_t dim, size_t block_size, FaerPar par, FaerFullPivLuParams params);
FAER_HIP_API FaerFullPivLuStatus libfaer_v0_23_full_piv_lu_factor_in_place_u64_f32(FaerMatMut A, FaerSliceMut row_perm_fwd, FaerSliceMut row_perm_bwd, FaerSliceMut col_perm_fwd, FaerSliceMut col_perm_bwd, FaerPar par, FaerMemAlloc mem, FaerFullPivLuParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_solve_in_place_scratch_u64_f32(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_solve_in_place_u64_f32(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_solve_transpose_in_place_scratch_u64_f32(size_t dim, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_solve_transpose_in_place_u64_f32(FaerMatRef L, FaerMatRef U, FaerConj A_conj, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);

/* ---------------------------------------------------------------------------------------------
 * 2d. QR with column pivoting (faer-ffi/src/lib.rs:1721-1876; qr/col_pivoting/factor.rs, solve.rs): HBM-bound;
 *     delayed rank-1 updates fused with the next step's dot products, down-dated column norms with the reference's
 *     recomputation rule (csrc/qr.hip, colpiv_qr_dev).  Permutation slices are HOST memory (ncols entries).
 * --------------------------------------------------------------------------------------------- */
FAER_HIP_API FaerColPivQrParams libfaer_v0_23_ColPivQrParams_f64(void);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_factor_in_place_scratch_u32_f64(size_t nrows, size_t ncols, size_t block_size, FaerPar par, FaerColPivQrParams params);
FAER_HIP_API FaerColPivQrStatus libfaer_v0_23_colpiv_qr_factor_in_place_u32_f64(FaerMatMut A, FaerMatMut Q_coeff, FaerSliceMut perm_fwd, FaerSliceMut perm_bwd, FaerPar par, FaerMemAlloc mem, FaerColPivQrParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_solve_in_place_scratch_u32_f64(size_t dim, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_solve_in_place_u32_f64(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_solve_transpose_in_place_scratch_u32_f64(size_t dim, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_solve_transpose_in_place_u32_f64(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_solve_lstsq_in_place_scratch_u32_f64(size_t nrows, size_t ncols, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_solve_lstsq_in_place_u32_f64(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_factor_in_place_scratch_u64_f64(size_t nrows, size_t ncols, size_t block_size, FaerPar par, FaerColPivQrParams params);
FAER_HIP_API FaerColPivQrStatus libfaer_v0_23_colpiv_qr_factor_in_place_u64_f64(FaerMatMut A, FaerMatMut Q_coeff, FaerSliceMut perm_fwd, FaerSliceMut perm_bwd, FaerPar par, FaerMemAlloc mem, FaerColPivQrParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_solve_in_place_scratch_u64_f64(size_t dim, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_solve_in_place_u64_f64(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_solve_transpose_in_place_scratch_u64_f64(size_t dim, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_solve_transpose_in_place_u64_f64(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_solve_lstsq_in_place_scratch_u64_f64(size_t nrows, size_t ncols, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_solve_lstsq_in_place_u64_f64(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerColPivQrParams libfaer_v0_23_ColPivQrParams_f32(void);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_factor_in_place_scratch_u32_f32(size_t nrows, size_t ncols, size_t block_size, FaerPar par, FaerColPivQrParams params);
FAER_HIP_API FaerColPivQrStatus libfaer_v0_23_colpiv_qr_factor_in_place_u32_f32(FaerMatMut A, FaerMatMut Q_coeff, FaerSliceMut perm_fwd, FaerSliceMut perm_bwd, FaerPar par, FaerMemAlloc mem, FaerColPivQrParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_solve_in_place_scratch_u32_f32(size_t dim, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_solve_in_place_u32_f32(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_solve_transpose_in_place_scratch_u32_f32(size_t dim, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_solve_transpose_in_place_u32_f32(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_solve_lstsq_in_place_scratch_u32_f32(size_t nrows, size_t ncols, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_solve_lstsq_in_place_u32_f32(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_factor_in_place_scratch_u64_f32(size_t nrows, size_t ncols, size_t block_size, FaerPar par, FaerColPivQrParams params);
FAER_HIP_API FaerColPivQrStatus libfaer_v0_23_colpiv_qr_factor_in_place_u64_f32(FaerMatMut A, FaerMatMut Q_coeff, FaerSliceMut perm_fwd, FaerSliceMut perm_bwd, FaerPar par, FaerMemAlloc mem, FaerColPivQrParams params);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_solve_in_place_scratch_u64_f32(size_t dim, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_solve_in_place_u64_f32(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_solve_transpose_in_place_scratch_u64_f32(size_t dim, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_solve_transpose_in_place_u64_f32(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_solve_lstsq_in_place_scratch_u64_f32(size_t nrows, size_t ncols, size_t block_size, size_t rhs_ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_solve_lstsq_in_place_u64_f32(FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerConj A_conj, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerMatMut rhs, FaerPar par, FaerMemAlloc mem);

/* reconstruct / inverse of the pivoted factorizations (faer-ffi/src/lib.rs:1877-1950, :2246-2330) */
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_reconstruct_scratch_u32_f64(size_t nrows, size_t ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_reconstruct_u32_f64(FaerMatMut A, FaerMatRef L, FaerMatRef U, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_inverse_scratch_u32_f64(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_inverse_u32_f64(FaerMatMut A_inv, FaerMatRef L, FaerMatRef U, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_reconstruct_scratch_u32_f64(size_t nrows, size_t ncols, size_t block_size, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_reconstruct_u32_f64(FaerMatMut A, FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_inverse_scratch_u32_f64(size_t dim, size_t block_size, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_inverse_u32_f64(FaerMatMut A_inv, FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_reconstruct_scratch_u64_f64(size_t nrows, size_t ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_reconstruct_u64_f64(FaerMatMut A, FaerMatRef L, FaerMatRef U, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_inverse_scratch_u64_f64(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_inverse_u64_f64(FaerMatMut A_inv, FaerMatRef L, FaerMatRef U, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_reconstruct_scratch_u64_f64(size_t nrows, size_t ncols, size_t block_size, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_reconstruct_u64_f64(FaerMatMut A, FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_inverse_scratch_u64_f64(size_t dim, size_t block_size, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_inverse_u64_f64(FaerMatMut A_inv, FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_reconstruct_scratch_u32_f32(size_t nrows, size_t ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_reconstruct_u32_f32(FaerMatMut A, FaerMatRef L, FaerMatRef U, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_inverse_scratch_u32_f32(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_inverse_u32_f32(FaerMatMut A_inv, FaerMatRef L, FaerMatRef U, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_reconstruct_scratch_u32_f32(size_t nrows, size_t ncols, size_t block_size, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_reconstruct_u32_f32(FaerMatMut A, FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_inverse_scratch_u32_f32(size_t dim, size_t block_size, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_inverse_u32_f32(FaerMatMut A_inv, FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_reconstruct_scratch_u64_f32(size_t nrows, size_t ncols, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_reconstruct_u64_f32(FaerMatMut A, FaerMatRef L, FaerMatRef U, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_full_piv_lu_inverse_scratch_u64_f32(size_t dim, FaerPar par);
FAER_HIP_API void libfaer_v0_23_full_piv_lu_inverse_u64_f32(FaerMatMut A_inv, FaerMatRef L, FaerMatRef U, FaerSliceRef row_perm_fwd, FaerSliceRef row_perm_bwd, FaerSliceRef col_perm_fwd, FaerSliceRef col_perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_reconstruct_scratch_u64_f32(size_t nrows, size_t ncols, size_t block_size, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_reconstruct_u64_f32(FaerMatMut A, FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
FAER_HIP_API FaerLayout libfaer_v0_23_colpiv_qr_inverse_scratch_u64_f32(size_t dim, size_t block_size, FaerPar par);
FAER_HIP_API void libfaer_v0_23_colpiv_qr_inverse_u64_f32(FaerMatMut A_inv, FaerMatRef Q_basis, FaerMatRef Q_coeff, FaerMatRef R, FaerSliceRef perm_fwd, FaerSliceRef perm_bwd, FaerPar par, FaerMemAlloc mem);
#endif /* FAER_HIP_NO_FFI_PROTOTYPES */

/* ---------------------------------------------------------------------------------------------
 * 3. Runtime control (new: the reference has no device).
 * --------------------------------------------------------------------------------------------- */
/* Library version / build target string, e.g. "faer_hip 0.1 gfx950". Never NULL. */
FAER_HIP_API const char *faer_hip_version(void);
/* Number of usable gfx950 devices (0 when none: compute entry points will abort). */
FAER_HIP_API int faer_hip_device_count(void);
/* Binds the calling thread to `device` (hipSetDevice). */
FAER_HIP_API void faer_hip_set_device(int device);
/* All work of the calling thread is enqueued on `hip_stream` (a hipStream_t; NULL = the null stream).
 * Lets a host framework (e.g. torch's current stream) order our kernels with its own. */
FAER_HIP_API void faer_hip_set_stream(void *hip_stream);
FAER_HIP_API void *faer_hip_get_stream(void);
/* Blocks until all work enqueued on the calling thread's stream is done. */
FAER_HIP_API void faer_hip_synchronize(void);
/* Releases the calling thread's internal look-ahead streams and events (they are re-created on demand).
 * Call before unloading the library or at process exit. */
FAER_HIP_API void faer_hip_shutdown(void);
/* Device memory helpers for C clients without another allocator. */
FAER_HIP_API void *faer_hip_malloc(size_t bytes);
FAER_HIP_API void faer_hip_free(void *ptr);
FAER_HIP_API void faer_hip_memcpy_h2d(void *dst_device, const void *src_host, size_t bytes);
FAER_HIP_API void faer_hip_memcpy_d2h(void *dst_host, const void *src_device, size_t bytes);
/* Tuning knob used by bench.py / tests: selects the GEMM tile variant (0 = auto). */
FAER_HIP_API void faer_hip_set_gemm_variant(int variant);
/* Debugging aid.  `which`: 0 = the caller's stream, 1 / 2 = the internal bulk / panel look-ahead stream; writes
 * {XCC id, HW_ID} of the CU each of `nblocks` probe workgroups ran on (2 * nblocks words of host memory). */
FAER_HIP_API void faer_hip_debug_stream_xcc(int which, int nblocks, unsigned *out_host);
/* Host-side planning logic of the drivers, callable without a GPU (unit tests): the look-ahead panel starts of the
 * blocked Cholesky (last entry = start of the sequential tail; returns the number of entries) and the leaf width
 * the cooperative LU panel kernel picks for `nrows` rows when `resident_workgroups` workgroups fit the device. */
FAER_HIP_API size_t faer_hip_debug_llt_plan(size_t n, size_t tail_rows, size_t nb2, size_t *starts, size_t cap);
FAER_HIP_API int faer_hip_debug_lu_leaf_width(size_t nrows, FaerHipDType dtype, int resident_workgroups);
/* tests: run every leaf of the partial-pivot LU on the non-cooperative path (the fallback for panels taller than the
 * cooperative kernel can keep resident and for the rerun after an exchange timeout) */
FAER_HIP_API void faer_hip_debug_lu_force_general(int on);
/* tests / A-B measurements: 1 = the look-ahead LU and Cholesky drivers lend the panel stream's idle compute units to their big
 * trailing products (helper launches that pull tiles from per-XCD counters).  Default 0: measured without gain on MI355X
 * (profiles/r06_exp_lend.txt); the factors do not depend on it. */
FAER_HIP_API void faer_hip_debug_lend_cus(int on);
/* tests: switch-over points of the look-ahead LU driver, in rows below the panel (0 = the tuned default): 256-column staged steps
 * below `nb2_from`, pipelined bulk-bound steps from `pipe_from` on, look-ahead at all from `la_min_cols` columns.  Lets a test drive
 * every phase of the driver and the transitions between them at N = 2-6 k; pivots and factors do not depend on the plan. */
FAER_HIP_API void faer_hip_debug_lu_plan(size_t nb2_from, size_t pipe_from, size_t la_min_cols);
/* tests: how many columns the one-pass tall-skinny QR path (csrc/tsqr.hip) completed in the calling thread's last
 * qr_factor_in_place (== ncols: the whole factorization; fewer: a panel was rejected and the classic path finished; -1: the
 * path was not applicable) */
FAER_HIP_API long faer_hip_debug_qr_one_pass_columns(void);
/* tests / A-B measurements: 0 = the one-pass QR path applies a panel and forms the next panel's Gram products in separate launches (the
 * round 3-5 schedule), 1 (default) = in one pass per panel with the next panel's kernel beside its second half (csrc/tsqr.hip), 2 = the
 * same without the raw copy of the panel (what matrices of more than 4.19 M rows run: V = P M as a launch of its own behind U2), 3 = the
 * plain schedule (Gram, panel, y, update per panel) on the streaming kernels of the fp64 instantiation, for fp32 data with 16-byte aligned
 * columns (5e5 x 256: 1.80 ms against 1.68 fused -- its panel kernels are not hidden; ahead of the fused schedule below ~50000 rows). */
FAER_HIP_API void faer_hip_debug_qr_fused(int on);
/* tests / A-B measurements: 0 = fp64 matrices never take the one-pass tall-skinny QR path (the classic path of rounds 1-6 runs), 1 (default) =
 * they take it under the same shape rule as fp32 (rows >= 1024, rows >= 3 cols, cols <= 512, unit row stride; columns that are not 16-byte aligned run scalar-access variants of the kernels). */
FAER_HIP_API void faer_hip_debug_qr_one_pass_f64(int on);
/* tests / A-B measurements: 0 = the classic QR path (square / wide matrices, rejected panels) factors its panels by the recursion down to
 * the 8-column cooperative leaf as in rounds 1-6, 1 (default) = a panel of up to 64 columns with at least 256 rows (and 4 rows per column) takes the one-pass
 * panel of csrc/tsqr.hip (and the recursion only if that refuses it). */
FAER_HIP_API void faer_hip_debug_qr_panels_one_pass(int on);
/* tests / A-B measurements: the shape rule of the whole-matrix one-pass QR path -- at least `min_rows` rows and `min_rows_per_column` rows per
 * column (0, 0 = the defaults). */
FAER_HIP_API void faer_hip_debug_qr_one_pass_shape_rule(long min_rows, long min_rows_per_column);
/* Full-pivot LU: 1 = the in-place path (two launches per step) instead of the one-launch-per-step path between two scratch copies
 * (default 0; identical factors and permutations, tests/test_gpu_factor.py). */
FAER_HIP_API void faer_hip_debug_fplu_inplace(int on);
/* tests: 1 = the single-workgroup vector kernels of the tridiagonal / bidiagonal / Hessenberg reductions run their memory-resident bodies
 * at every size (default: from 4096 remaining rows down they keep their columns in registers); results must not depend on it. */
FAER_HIP_API void faer_hip_debug_level2_force_memory_bodies(int on);
/* host logic of the distributed LU: may a step factor its look-ahead panel of `panel_rows` rows on the CU-masked panel stream? */
FAER_HIP_API int faer_hip_debug_dist_two_streams_ok(size_t panel_rows, FaerHipDType dtype, int panel_cus, int all_cus);
/* Instrumented builds (make -C csrc timing): prints and resets the in-kernel phase counters; a no-op otherwise. */
FAER_HIP_API void faer_hip_debug_dump_timing(void);
/* The internal CU-masked streams themselves (1 = bulk, 2 = panel), for microbenchmarks via faer_hip_set_stream. */
FAER_HIP_API void *faer_hip_debug_internal_stream(int which);
/* Measures `iters` back-to-back launches of the dense GEMM kernel on the calling thread's stream with
 * hipEvents and returns the average milliseconds per launch (operands must be device memory).
 * bench.py uses it for the `roofline` object. */
FAER_HIP_API double faer_hip_time_gemm_ms(FaerHipDType dtype, size_t m, size_t n, size_t k, void *dst, ptrdiff_t dst_cs,
                                          const void *lhs, ptrdiff_t lhs_cs, const void *rhs, ptrdiff_t rhs_cs, int iters);
/* Raw fp64/fp32 MFMA issue-rate probe: every wave of a full-chip grid runs `iters` dependent-free
 * v_mfma 16x16x4 instructions from registers; returns achieved TFLOP/s.  The measured ceiling that
 * roofline fractions are quoted against next to the datasheet peak. */
FAER_HIP_API double faer_hip_mfma_peak_tflops(FaerHipDType dtype, int iters);

/* Kernel-class profile of the calling thread's library calls (bench.py's per-workload `roofline` objects): between
 * faer_hip_prof_begin and faer_hip_prof_end every launch of a dominant kernel class is bracketed by two timing events on
 * the stream it runs on.  prof_end synchronises the device and fills out[3 * cls + {0, 1, 2}] = {milliseconds inside the
 * class's launches, launches, units} for cls = 0 big-tile MFMA products (units: flop), 1 LU panel kernel (columns),
 * 2 one-pass QR update (algorithmic bytes), 3 one-pass QR Gram (bytes), 4 one-pass QR panel kernel (launches),
 * 5 Cholesky leaf (columns).  Measurement aid: the events cost a few microseconds per launch, do not time a profiled call. */
#define FAER_HIP_PROF_CLASSES 6
FAER_HIP_API void faer_hip_prof_begin(void);
FAER_HIP_API void faer_hip_prof_end(double *out_3_x_classes);
/* The same, plus one record of 8 doubles per profiled launch (at most `cap` records; returns how many): {class, ms, units,
 * d0, d1, d2, d3 + 65536 * stream, start in ms after the first recorded launch}.  Class 0 (the MFMA products): d = m, n, k of the
 * product and 0 for a Full, 1 + tri_skip for a Lower destination; stream 0 = the caller's, 1 = bulk, 2 = panel, 3 = side.
 * tools/gpu_update_in_situ.py replays every product of a factorization alone on the same stream from these records. */
FAER_HIP_API size_t faer_hip_prof_end_spans(double *out_3_x_classes, double *spans_8_per_launch, size_t cap);
/* Idle-chip hand-off latency between two resident workgroups on different XCDs, microseconds per one-way hop (tagged
 * 8-byte granule, write-through store -> polling load).  The latency-bound kernels of the LU / Cholesky chains scale
 * with it: bench.py prints it so that a line from a slow box of a pool is recognisable.  < 0: the probe timed out. */
FAER_HIP_API double faer_hip_xwg_hop_us(int iters);

/* Partial-pivot LU on a GPU shared with other work.  The cooperative panel kernel exchanges pivots between resident
 * workgroups; the library arranges their residency itself (one workgroup per compute unit of the stream's CU set, taller
 * panels on a non-cooperative kernel), so on a GPU of its own the exchange cannot stall.  If OTHER work holds compute units
 * for longer than the bounded spin (~0.2 s) the call returns PartialPivLuStatus::Unknown and leaves a partially factored
 * matrix -- for matrices above 512 MiB.  Up to that size the library keeps a copy of its own (< 1 % of the call), restores
 * A after a timeout and finishes on the non-cooperative path, so callers of the reference's FFI surface need nothing.
 * For larger matrices a caller that wants the factorization completed even then lends a copy of the input first:
 * `device_copy` = device memory holding the nrows x ncols matrix (elements of `elem_bytes` = 4 or 8 bytes) column major
 * with leading dimension nrows, valid until the calling thread's NEXT partial_piv_lu call returns.  That call consumes
 * the loan at entry whatever its own shape; a loan made for another shape or scalar type is a precondition violation
 * (abort), never a read with the wrong extent.  After a timeout the call restores A from the copy and factors it on the
 * non-cooperative path (identical pivots, slower). */
FAER_HIP_API void faer_hip_partial_piv_lu_lend_copy(const void *device_copy, size_t nrows, size_t ncols, int elem_bytes);

/* ---------------------------------------------------------------------------------------------
 * 4. Multi-GPU: 1-D block-column partition, one process per GPU.
 *    The caller owns the transport: `bcast` is invoked with device buffers and must broadcast
 *    `bytes` bytes from rank `root` to every rank on the calling thread's stream
 *    (torch.distributed.broadcast over RCCL in bench.py; gloo in the CPU tests).
 * --------------------------------------------------------------------------------------------- */
typedef void (*FaerHipBcastFn)(void *user, void *device_buf, size_t bytes, int root);
/* Optional asynchronous pair: `ibcast` starts the broadcast of `bytes` bytes at `device_buf` from `root` (ordered
 * after the work already enqueued on the calling thread's stream), `wait` makes that stream wait for the broadcast
 * started with the same `slot` (0 .. FAER_HIP_COMM_SLOTS - 1; at most one broadcast per slot is in flight: the LU uses
 * slots 0 / 1, the Cholesky ships a panel in up to four row chunks and alternates two panels, slots 0 .. 7).  `wait` may be
 * called MORE THAN ONCE for the same slot between two `ibcast`s of it (the Cholesky waits for a chunk on every internal
 * stream that reads it) and must then order the calling thread's stream behind the same transfer again.  With both NULL
 * the drivers fall back to the blocking `bcast` (same results, no overlap of transfers with compute). */
#define FAER_HIP_COMM_SLOTS 8
typedef void (*FaerHipIbcastFn)(void *user, void *device_buf, size_t bytes, int root, int slot);
typedef void (*FaerHipWaitFn)(void *user, int slot);
typedef struct FaerHipComm {
	int rank; int world_size; FaerHipBcastFn bcast; void *user; FaerHipIbcastFn ibcast; FaerHipWaitFn wait;
} FaerHipComm;

/* Built-in transport: RCCL broadcasts on a dedicated stream, ordered against the calling thread's stream with events
 * (csrc/rccl_transport.hip).  librccl is dlopen-ed on first use.  Rank 0 obtains the 128-byte ncclUniqueId with
 * faer_hip_rccl_unique_id (0 = ok) and the application ships it to the other ranks; every rank then calls
 * faer_hip_rccl_create (collective) and passes faer_hip_rccl_comm(handle) to the faer_hip_dist_* entry points. */
FAER_HIP_API int faer_hip_rccl_unique_id(void *out_128_bytes);
FAER_HIP_API void *faer_hip_rccl_create(const void *unique_id_128_bytes, int rank, int world_size);
FAER_HIP_API FaerHipComm faer_hip_rccl_comm(void *handle);
FAER_HIP_API void faer_hip_rccl_destroy(void *handle);

/* Number of block columns of width `nb` owned by `rank` out of n columns distributed block-cyclically. */
/* Loop-back transport (TEST infrastructure): the ranks of a distributed factorization as threads of ONE process on ONE GPU, data moved by
 * device-to-device copies; same contract as the RCCL transport (its calls only need the calling thread's current stream), so the stream
 * schedule a rank runs over the built-in transport can be exercised with several ranks on a one-GPU box (tests/test_gpu_dist_threads.py).
 * faer_hip_loopback_group_create once, faer_hip_loopback_rank_create on each rank's thread, faer_hip_loopback_comm(rank_handle) -> comm. */
FAER_HIP_API void *faer_hip_loopback_group_create(int world_size);
FAER_HIP_API void *faer_hip_loopback_rank_create(void *group, int rank);
FAER_HIP_API FaerHipComm faer_hip_loopback_comm(void *rank_handle);
FAER_HIP_API void faer_hip_loopback_stats(void *rank_handle, double *out2);
FAER_HIP_API void faer_hip_loopback_rank_destroy(void *rank_handle);
FAER_HIP_API void faer_hip_loopback_group_destroy(void *group);
FAER_HIP_API size_t faer_hip_dist_local_ncols(size_t n, size_t nb, int rank, int world_size);
/* measurement aids (bench.py --gpus N --workload lu|llt): the calling thread's last faer_hip_dist_* factorization --
 * out3 = {device ms of the whole call, device ms of the panel factorizations this rank owned, their number};
 * faer_hip_rccl_stats: out4 = {ranks the communicator reports (ncclCommCount), broadcasts, bytes, device ms inside
 * ncclBroadcast} since the previous call */
FAER_HIP_API void faer_hip_dist_last_stats(double *out3);
FAER_HIP_API void faer_hip_rccl_stats(void *handle, double *out4);
/* Scalars of device scratch the distributed LU needs: all pivots + two broadcast buffers ({pivots, packed panel}). */
FAER_HIP_API size_t faer_hip_dist_panel_ws_scalars(size_t nrows, size_t nb, FaerHipDType dtype);
/* Distributed partial-pivot LU of an m x n matrix whose block columns (width nb) are dealt block-cyclically
 * over comm.world_size ranks; A_local holds this rank's columns (m x local_ncols, device memory, col-major).
 * perm_fwd / perm_bwd (m entries, u64, HOST memory) are filled on every rank.  `panel_ws` is device scratch of
 * faer_hip_dist_panel_ws_scalars(m, nb) scalars.  Per block column: the owner factors its m_k x nb panel with
 * the single-GPU panel code (same pivoting rule, lu/partial_pivoting/factor.rs:19-187), ONE broadcast ships
 * {pivots, panel}, every rank applies the interchanges, the unit-lower solve and the trailing update to its own
 * columns; look-ahead: the owner of the next block column updates and factors it first and its broadcast
 * overlaps the remaining updates (csrc/dist_lu.h). */
FAER_HIP_API FaerPartialPivLuStatus faer_hip_dist_partial_piv_lu_f64(FaerMatMut A_local, size_t n_global, size_t nb, FaerSliceMut perm_fwd, FaerSliceMut perm_bwd, FaerHipComm comm, void *panel_ws);
FAER_HIP_API FaerPartialPivLuStatus faer_hip_dist_partial_piv_lu_f32(FaerMatMut A_local, size_t n_global, size_t nb, FaerSliceMut perm_fwd, FaerSliceMut perm_bwd, FaerHipComm comm, void *panel_ws);


/* Distributed Cholesky (lower) of an n x n matrix, same partition and transport (csrc/dist_llt.h): A_local holds
 * this rank's block columns at full height (n x local_ncols, device memory, col-major; only the lower triangle of
 * the global matrix is referenced or written).  Per block column the owner factors the diagonal block and solves
 * the rows below (cholesky/ldlt/factor.rs:407-433), ONE broadcast ships that column panel, every rank applies
 * lower(A11) -= L10 L10^T (:436-446) to the block columns it owns, with the same look-ahead as the LU.
 * `panel_ws`: device scratch of faer_hip_dist_llt_ws_scalars(n, nb) scalars.  The status is identical on all ranks
 * (NonPositivePivot carries the smallest failing global index). */
FAER_HIP_API size_t faer_hip_dist_llt_ws_scalars(size_t n, size_t nb, FaerHipDType dtype);
FAER_HIP_API FaerLltStatus faer_hip_dist_llt_f64(FaerMatMut A_local, size_t n_global, size_t nb, FaerLltRegularization regularization, FaerHipComm comm, void *panel_ws);
FAER_HIP_API FaerLltStatus faer_hip_dist_llt_f32(FaerMatMut A_local, size_t n_global, size_t nb, FaerLltRegularization regularization, FaerHipComm comm, void *panel_ws);

/* ------------------------------------------------------------------------------------------------
 * Reduction to condensed form, first member: tridiagonalization of a self-adjoint
 * matrix, faer::linalg::evd::tridiag::tridiag_in_place (faer/src/linalg/evd/tridiag.rs:274; the FFI of the reference
 * reaches it only inside self_adjoint_evd, so this entry point is an extension with the Rust function's argument
 * meaning).  A: n x n, only the lower triangle is read or written.  On return its diagonal and subdiagonal hold the
 * tridiagonal T with A = Q T Q^H, the essential parts of the Householder reflectors sit below the subdiagonal and
 * `householder` (block_size x (n - 1), block_size >= 1) holds the block Householder factors of
 * A.submatrix(1, 0, n - 1, n - 1) -- what apply_block_householder_sequence_* consumes (tridiag.rs:516-533, :561-585).
 * Level-2, HBM-bound like the reference (csrc/qr.hip, "Tridiagonalization").  Host or device operands. */
FAER_HIP_API void faer_hip_tridiag_in_place_f64(FaerMatMut A, FaerMatMut householder);
FAER_HIP_API void faer_hip_tridiag_in_place_f32(FaerMatMut A, FaerMatMut householder);

/* Second member: bidiagonalization, faer::linalg::svd::bidiag::bidiag_in_place (faer/src/linalg/svd/bidiag.rs:47).  On
 * return the diagonal and superdiagonal of A hold the upper bidiagonal B with A = U B V^H; the left reflectors sit below
 * the diagonal with their block factors in H_left (bl x min(m, n)), the right reflectors right of the superdiagonal with
 * their block factors in H_right (br x (min(m, n) - 1)) -- the layout apply_block_householder_sequence_* consumes
 * (bidiag.rs:216-254, :404-428).  The reference's SVD only calls it with nrows >= ncols; a wide matrix is processed the
 * way the reference processes it (min(m, n) columns, the last row left normalised without a right reflector).
 * Level-2, HBM-bound like the reference (csrc/qr.hip, "Bidiagonalization").  Host or device operands. */
FAER_HIP_API void faer_hip_bidiag_in_place_f64(FaerMatMut A, FaerMatMut H_left, FaerMatMut H_right);
FAER_HIP_API void faer_hip_bidiag_in_place_f32(FaerMatMut A, FaerMatMut H_left, FaerMatMut H_right);

/* Third member: reduction to upper Hessenberg form, faer::linalg::evd::hessenberg::hessenberg_in_place
 * (faer/src/linalg/evd/hessenberg.rs:549).  A: n x n.  On return the part of A on and above the subdiagonal holds H with
 * A = Q H Q^H, the essential parts of the reflectors sit below the subdiagonal, `householder` (block_size x (n - 1)) holds
 * the block Householder factors of A.submatrix(1, 0, n - 1, n - 1) (hessenberg.rs:382-406, :758-783).  The reference picks
 * an unblocked variant below n = 256 and a blocked one above; both compute the same reflectors in different orders of
 * operations and this entry point agrees with either up to rounding (csrc/qr.hip, "Hessenberg reduction").  Level-2,
 * HBM-bound.  Host or device operands. */
FAER_HIP_API void faer_hip_hessenberg_in_place_f64(FaerMatMut A, FaerMatMut householder);
FAER_HIP_API void faer_hip_hessenberg_in_place_f32(FaerMatMut A, FaerMatMut householder);

#ifdef __cplusplus
}
#endif
#endif /* FAER_HIP_H */
