// In-place lower Cholesky (LLT) for gfx950.
//
// Replaces faer/src/linalg/cholesky/llt/factor.rs:67-97 and the engine behind it,
// cholesky/ldlt/factor.rs:7-498.
//
// The reference is right-looking with 128-column steps (factor.rs:392).  A rank-128 fp64 update is only
// ~16 flop/byte -- at the MI355X machine balance -- so the GPU driver recurses by HALVES instead:
//     A00 = L00 L00^T (recurse) ; A10 <- A10 L00^-T (TRSM) ; A11 -= A10 A10^T (MFMA SYRK, K = n/2) ; recurse A11
// which performs the same arithmetic per entry but moves O(n^2 log n) bytes instead of O(n^3 / 128).
//
// Leaf (n <= 128): ONE 512-thread workgroup, the block resident in LDS (lds_blocks.h), blocked right-looking
// in four 32-column steps:
//   * panel step: the 32 x 32 diagonal block and the rows below it are factored by up to three wavefronts
//     WITHOUT any synchronisation inside the step: every wavefront keeps the diagonal block's rows in lanes
//     0-31 (redundantly) and 32 of the rows below in lanes 32-63, one matrix row per lane in 32 registers; the
//     multipliers l_kj reach the other lanes through v_readlane.  The per-column dependency chain
//     (sqrt -> reciprocal -> scale -> update of the next diagonal entry) is the only serial part;
//   * trailing step: A22 -= L21 L21^T on the MFMA pipe straight out of LDS (K = 32, 16 x 16 tiles over 8 waves);
//   * semantics of the reference's base kernel (factor.rs:122-174): d = a_jj (optionally regularised and
//     counted), fail with the GLOBAL column index if !(d > 0), every entry of column j -- the diagonal one
//     included -- is multiplied by the reciprocal 1 / sqrt(d).
// The leaf also inverts its own L_kk in place in LDS (lds_tri_inv_inplace) and stores the inverse in the
// factorization's workspace: the TRSMs of all enclosing recursion levels then run as MFMA GEMMs against those
// inverses (trsm_lower_pre_dev) with no further triangular kernels.
// Failure is reported through a device status word (first failing GLOBAL column index + 1); later
// kernels see it and become no-ops, so the host synchronises exactly once per factorization.
#include "common.h"
#include "lds_blocks.h"

namespace fh {

constexpr int POTRF_NB = LDS_NB;
constexpr int POTRF_PB = 32; // panel width inside the leaf

// sq = sqrt(d) and inv = 1 / sq for a wave-uniform d > 0: v_rsq + the coupled Newton iteration that the
// compiler's own sqrt expansion uses (two residual corrections => sq is the correctly rounded root in all but
// rare half-ulp cases), and the reciprocal from the same iteration's h ~ 1 / (2 sq) instead of a separate
// 12-instruction division: this chain is the serial part of the leaf (128 dependent columns).
// Outside a safe exponent range (and for d <= 0 / NaN) it falls back to the library sqrt and division.
static __device__ __forceinline__ void sqrt_and_recip(double d, double &sq, double &inv)
{
	if (d > 1e-280 && d < 1e280) {
		const double y = __builtin_amdgcn_rsq(d);
		double g = d * y, h = 0.5 * y;
		double r = __builtin_fma(-h, g, 0.5);
		g = __builtin_fma(g, r, g);
		h = __builtin_fma(h, r, h);
		double e = __builtin_fma(-g, g, d);
		g = __builtin_fma(e, h, g);
		e = __builtin_fma(-g, g, d);
		g = __builtin_fma(e, h, g);
		r = __builtin_fma(-h, g, 0.5);
		h = __builtin_fma(h, r, h);
		r = __builtin_fma(-h, g, 0.5);
		h = __builtin_fma(h, r, h);
		sq = g;
		inv = h + h;
	} else {
		sq = sqrt(d);
		inv = 1.0 / sq;
	}
}
static __device__ __forceinline__ void sqrt_and_recip(float d, float &sq, float &inv)
{
	sq = sqrtf(d);
	inv = 1.0f / sq;
}

template <typename T>
__global__ __launch_bounds__(LDS_NT) void potrf_leaf_kernel(T *A, idx_t rs, idx_t cs, int n, int regularize, T eps, T delta,
							   int *status, int offset, T *Winv)
{
	__shared__ T S[LDS_NB * LDS_LDP];
	__shared__ int s_fail;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	if (status[0] != 0)
		return; // an earlier block already failed
	if (tid == 0)
		s_fail = 0;
	lds_load_lower<T>(S, A, rs, cs, n);
	__syncthreads();

	const int np = (n + POTRF_PB - 1) / POTRF_PB * POTRF_PB; // identity padded
	int count = 0;
	bool failed = false;
	for (int j0 = 0; j0 < np; j0 += POTRF_PB) {
		// ---- panel step: rows j0 .. np-1 of columns j0 .. j0+31
		const int below = np - j0 - POTRF_PB; // rows under the diagonal block (multiple of 32)
		const int nw = below > 0 ? below / 32 : 1;
		if (wave < nw) {
			const bool diag_lane = lane < 32;
			const int row = diag_lane ? j0 + lane : j0 + POTRF_PB + wave * 32 + (lane - 32);
			const bool valid = row < np;
			const int rr = valid ? row : j0;
			T a[POTRF_PB];
#pragma unroll
			for (int c = 0; c < POTRF_PB; ++c)
				a[c] = S[(j0 + c) * LDS_LDP + rr];
			int fail_col = 0;
#pragma unroll
			for (int j = 0; j < POTRF_PB; ++j) {
				if (fail_col == 0) { // wave uniform
					T d = lane_bcast(a[j], j); // a_jj lives in lane j (diagonal rows)
					if (regularize && d <= eps) { // cholesky/ldlt/factor.rs:122-131 (llt: sign == +1)
						d = delta;
						if (j0 + j < n)
							++count;
					}
					T sq, inv;
					sqrt_and_recip(d, sq, inv);
					if (!(d > (T) 0) || sq == (T) 0 || !isfinite(sq)) {
						fail_col = j0 + j + 1;
					} else {
						const T lj = a[j] * inv; // column j, diagonal entry included (factor.rs:160-174)
						// column j is final: park it in the block image (every panel wave writes the same
						// diagonal-block values) and fetch the multipliers l_kj back as broadcast reads --
						// LDS operations of one wavefront execute in order
						T *colj = S + (j0 + j) * LDS_LDP;
						if (valid && (!diag_lane || lane >= j))
							colj[row] = lj;
						__builtin_amdgcn_wave_barrier();
#pragma unroll
						for (int k = j + 1; k < POTRF_PB; ++k)
							a[k] = __builtin_fma(-lj, colj[j0 + k], a[k]); // a_ik -= l_ij l_kj
					}
				}
			}
			if (fail_col != 0 && tid == 0)
				s_fail = fail_col;
		}
		__syncthreads();
		if (s_fail != 0) {
			failed = true;
			break;
		}
		// ---- trailing step: A22(lower) -= L21 L21^T, L21 = rows j0+32 .. np-1 of the panel (K = 32)
		if (below > 0) {
			const int nt = below / 16;
			const int ntiles = nt * (nt + 1) / 2;
			const int l15 = lane & 15, lhi = lane >> 4;
			const int t0 = j0 + POTRF_PB;
			for (int t = wave; t < ntiles; t += LDS_NW) {
				// t enumerates (ti >= tj) row by row
				int ti = (int) ((sqrtf(8.0f * (float) t + 1.0f) - 1.0f) * 0.5f);
				while ((ti + 1) * (ti + 2) / 2 <= t)
					++ti;
				while (ti * (ti + 1) / 2 > t)
					--ti;
				const int tj = t - ti * (ti + 1) / 2;
				// D[i][j] = sum_k L[t0 + 16 ti + i][k] L[t0 + 16 tj + j][k]: A-type reads for both operands
				typename Mfma<T>::acc_t acc = (typename Mfma<T>::acc_t) (T) 0;
				const T *pa = S + (j0 + lhi) * LDS_LDP + t0 + ti * 16 + l15;
				const T *pb = S + (j0 + lhi) * LDS_LDP + t0 + tj * 16 + l15;
#pragma unroll
				for (int kk = 0; kk < POTRF_PB; kk += 4)
					acc = Mfma<T>::run(pa[kk * LDS_LDP], pb[kk * LDS_LDP], acc);
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const int gi = t0 + ti * 16 + Mfma<T>::row(r, lhi), gj = t0 + tj * 16 + l15;
					if (gi >= gj)
						S[gj * LDS_LDP + gi] -= acc[r];
				}
			}
		}
		__syncthreads();
	}
	// ---- write back the lower triangle (also after a failure: the columns before the failing one are final)
	lds_store_block<T>(S, A, rs, cs, n, true);
	if (failed) {
		if (tid == 0)
			atomicCAS(status, 0, offset + s_fail);
		return;
	}
	if (tid == 0 && count > 0)
		atomicAdd(status + 1, count);
	if (Winv) {
		__syncthreads();
		lds_tri_inv_inplace<T>(S, 0);
		lds_store_block<T>(S, Winv, 1, LDS_NB, LDS_NB, false);
	}
}

template <typename T>
static void potrf_rec(MatV<T> A, int regularize, T eps, T delta, int *status, idx_t offset, T *Wbase, bool need_inv)
{
	const idx_t n = A.nrows;
	if (n == 0)
		return;
	if (n <= POTRF_NB) {
		T *W = need_inv ? Wbase + (size_t) (offset / POTRF_NB) * POTRF_NB * POTRF_NB : nullptr;
		hipLaunchKernelGGL(potrf_leaf_kernel<T>, dim3(1), dim3(LDS_NT), 0, ctx().stream, A.p, A.rs, A.cs, (int) n, regularize,
				   eps, delta, status, (int) offset, W);
		FH_HIP(hipGetLastError());
		return;
	}
	const idx_t h = ((n / 2 + POTRF_NB - 1) / POTRF_NB) * POTRF_NB;
	MatV<T> A00 = A.sub(0, 0, h, h), A10 = A.sub(h, 0, n - h, h), A11 = A.sub(h, h, n - h, n - h);
	potrf_rec<T>(A00, regularize, eps, delta, status, offset, Wbase, true);
	// A10 <- A10 L00^-T, expressed like the reference (cholesky/ldlt/factor.rs:422-426) as L00 \ A10^T
	trsm_lower_pre_dev<T>(A00.c(), A10.t(), Wbase + (size_t) (offset / POTRF_NB) * POTRF_NB * POTRF_NB);
	// lower(A11) -= A10 A10^T  (cholesky/ldlt/factor.rs:436-446 -> triangular.rs:602 DstKind::Lower)
	gemm_dev<T>(A11, DST_LOWER, true, A10.c(), A10.t().c(), (T) -1);
	potrf_rec<T>(A11, regularize, eps, delta, status, offset + h, Wbase, need_inv);
}

template <typename T> long potrf_lower_dev(MatV<T> A, T reg_delta, T reg_eps)
{
	FH_CHECK(A.nrows == A.ncols, "potrf: matrix must be square");
	FH_CHECK(A.nrows < (1L << 30), "potrf: matrix too large");
	if (A.nrows == 0)
		return 0;
	const idx_t n = A.nrows;
	Scratch st(64);
	int *status = st.as<int>();
	FH_HIP(hipMemsetAsync(status, 0, 64, ctx().stream));
	// one 128 x 128 inverse per diagonal block (only the blocks that some TRSM will use are filled)
	const idx_t nblk = (n + POTRF_NB - 1) / POTRF_NB;
	Scratch winv(n > POTRF_NB ? (size_t) nblk * POTRF_NB * POTRF_NB * sizeof(T) : 256);
	const int regularize = (reg_delta > (T) 0 && reg_eps > (T) 0) ? 1 : 0; // cholesky/llt/factor.rs:85-86
	potrf_rec<T>(A, regularize, reg_eps, reg_delta, status, 0, winv.as<T>(), false);
	int h[2] = {0, 0};
	FH_HIP(hipMemcpyAsync(h, status, sizeof(h), hipMemcpyDeviceToHost, ctx().stream));
	ctx().sync();
	if (h[0] != 0)
		return -(long) h[0]; // -(index + 1)
	return (long) h[1];
}

template long potrf_lower_dev<double>(MatV<double>, double, double);
template long potrf_lower_dev<float>(MatV<float>, float, float);

} // namespace fh
