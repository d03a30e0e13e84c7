// In-place lower Cholesky (LLT) for gfx950.
//
// Replaces faer/src/linalg/cholesky/llt/factor.rs:67-97 and the engine behind it,
// cholesky/ldlt/factor.rs:7-498.
//
// The reference is right-looking with 128-column steps (factor.rs:392).  A rank-128 fp64 update is only
// ~16 flop/byte -- at the MI355X machine balance -- so the GPU drivers use larger steps with the same arithmetic per
// entry: recursion by HALVES below 2048 columns (potrf_rec), 1024-column steps above (potrf_lookahead: right-looking
// panels in 128-column blocks, trailing updates with K = 1024, look-ahead on two CU-masked streams for large n).
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
// The leaf also writes the PACKED IMAGE of its L_kk (trsm_pack.h) into the factorization's workspace: the panel
// solves of all enclosing levels substitute against it (trsm_lower_pre_dev: substitution inside the 128-blocks like
// the reference's triangular_solve.rs, MFMA products off the diagonal) without a separate packing launch.
// Failure is reported through a device status word (first failing GLOBAL column index + 1); later
// kernels see it and become no-ops, so the host synchronises exactly once per factorization.
#include "common.h"
#include "lds_blocks.h"
#include "trsm_pack.h"

namespace fh {

constexpr int POTRF_NB = LDS_NB;
constexpr int POTRF_PB = 32; // panel width inside the leaf

// inv = 1 / sqrt(d) for a wave-uniform d > 0 on the shortest dependent chain: v_rsq_f64, one coupled Newton
// step for (g ~ sqrt d, h ~ 1 / (2 sqrt d)) and one more step for h: 7 dependent operations instead of the
// ~25 of sqrt() followed by a division.  This chain is the serial part of the leaf (128 dependent columns);
// the result is within an ulp or two of the reference's (1 / sqrt(d)) (cholesky/ldlt/factor.rs:160-163), well
// inside the parity tolerance.  Outside a safe exponent range it falls back to the library sqrt and division.
// Returns false for the reference's failure cases (!(d > 0), sqrt not finite or zero).
static __device__ __forceinline__ bool recip_sqrt(double d, double &inv)
{
	if (d > 1e-280 && d < 1e280) {
		const double y = __builtin_amdgcn_rsq(d);
		double g = d * y, h = 0.5 * y;
		double r = fh_fma(-h, g, 0.5);
		g = fh_fma(g, r, g);
		h = fh_fma(h, r, h);
		r = fh_fma(-h, g, 0.5);
		h = fh_fma(h, r, h);
		inv = h + h;
		return true;
	}
	const double sq = sqrt(d);
	inv = 1.0 / sq;
	return d > 0.0 && sq != 0.0 && isfinite(sq);
}
static __device__ __forceinline__ bool recip_sqrt(float d, float &inv)
{
	const float sq = sqrtf(d);
	inv = 1.0f / sq;
	return d > 0.0f && sq != 0.0f && isfinite(sq);
}

#ifdef FH_LEAF_TIMING
#define FH_LT(i)                                                                                                         \
	do {                                                                                                             \
		const long long now_ = (long long) __builtin_readcyclecounter();                                         \
		tacc[i] += now_ - tlast;                                                                                 \
		tlast = now_;                                                                                            \
	} while (0)
__device__ unsigned long long g_leaf_timing[8];
#else
#define FH_LT(i)                                                                                                         \
	do {                                                                                                             \
	} while (0)
#endif

// LDLT == true: the same leaf for the unit-lower L D L^T factorization (cholesky/ldlt/factor.rs with is_llt ==
// false): the pivot d_j is kept (no root), regularised according to its expected sign, a zero / non finite pivot
// is the error, the column is divided by d_j, the updates are weighted by d_j, D goes to `Dout` and -- like the
// reference's cholesky_in_place (:791-798) -- onto the diagonal of A.
template <typename T, bool LDLT>
__global__ __launch_bounds__(LDS_NT) void potrf_leaf_kernel(T *A, idx_t rs, idx_t cs, int n, int regularize, T eps, T delta,
							   int *status, int offset, T *Winv, const signed char *signs, T *Dout)
{
	__shared__ T S[LDS_NB * LDS_LDP];
	__shared__ T s_d[LDS_NB]; // LDLT: the pivots of this block
	__shared__ int s_fail;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	if (status[0] != 0)
		return; // an earlier block already failed
	if (tid == 0)
		s_fail = 0;
#ifdef FH_LEAF_TIMING
	long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	long long tlast = (long long) __builtin_readcyclecounter();
#endif
	lds_load_lower<T>(S, A, rs, cs, n);
	__syncthreads();
	FH_LT(0);

	const int np = (n + POTRF_PB - 1) / POTRF_PB * POTRF_PB; // identity padded
	int count = 0;
	bool failed = false;
	for (int j0 = 0; j0 < np; j0 += POTRF_PB) {
		// ---- panel step: rows j0 .. np-1 of columns j0 .. j0+31
		const int below = np - j0 - POTRF_PB; // rows under the diagonal block (multiple of 32)
		const int nw = below > 0 ? below / 32 : 1;
		// every panel wave first LOADS its rows (the diagonal block's rows are loaded by all of them), and only after
		// a workgroup barrier do the waves start writing finished columns back into the block image: without it
		// a wave that runs ahead overwrites diagonal-block entries that a slower wave has not loaded yet (seen as
		// rare spurious failures once other kernels shared the chip and skewed the waves)
		const bool diag_lane = lane < 32;
		const int row = diag_lane ? j0 + lane : j0 + POTRF_PB + wave * 32 + (lane - 32);
		const bool valid = row < np;
		T a[POTRF_PB];
		if (wave < nw) {
			const int rr = valid ? row : j0;
#pragma unroll
			for (int c = 0; c < POTRF_PB; ++c)
				a[c] = S[(j0 + c) * LDS_LDP + rr];
		}
		__syncthreads();
		if (wave < nw) {
			int fail_col = 0;
			// the reciprocal root of column j is computed one iteration ahead, right after the diagonal entry
			// a_jj has received its last update through the register path, so that the long latencies
			// (rsq chain, LDS round trip of the multipliers) overlap instead of adding up
			T inv, dj = (T) 1; // dj: the pivot of the current column (LDLT)
			bool ok;
			// pivot of column jc from its updated diagonal entry d: regularisation (cholesky/ldlt/factor.rs:122-144;
			// llt: sign == +1, only that correction is counted), then 1/sqrt(d) (LLT) or 1/d (LDLT)
			auto pivot = [&](T d, int jc, T &inv_out, T &d_out) -> bool {
				if (regularize) {
					if constexpr (LDLT) {
						const int sign = (signs && jc < n) ? (int) signs[offset + jc] : 0;
						const bool small_or_negative = d <= eps, minus_small_or_positive = d >= -eps;
						if (sign == 1 && small_or_negative) {
							d = delta;
							if (jc < n)
								++count;
						} else if (sign == -1 && minus_small_or_positive) {
							d = -delta;
						} else if (small_or_negative && minus_small_or_positive) {
							d = d < (T) 0 ? -delta : delta;
						}
					} else if (d <= eps) {
						d = delta;
						if (jc < n)
							++count;
					}
				}
				d_out = d;
				if constexpr (LDLT) {
					inv_out = (T) 1 / d;
					return d != (T) 0 && isfinite(d);
				} else {
					return recip_sqrt(d, inv_out);
				}
			};
			ok = pivot(lane_bcast(a[0], 0), j0, inv, dj);
#pragma unroll
			for (int j = 0; j < POTRF_PB; ++j) {
				if (fail_col == 0) { // wave uniform
					if (!ok) {
						fail_col = j0 + j + 1;
						if (LDLT && wave == 0 && lane == 0)
							s_d[j0 + j] = dj; // the failing pivot still goes to D (factor.rs:343-347, :791-798)
					} else {
						const T lj = a[j] * inv; // column j, diagonal entry included (factor.rs:160-174)
						const T wj = LDLT ? dj : (T) 1; // weight of column j in the updates: a_ik -= l_ij d_j l_kj
						// column j is final: park it in the block image (every panel wave writes the same
						// diagonal-block values); the multipliers l_kj, k >= j + 2, come back as broadcast reads
						// (LDS operations of one wavefront execute in order)
						T *colj = S + (j0 + j) * LDS_LDP;
						if (valid && (!diag_lane || lane >= j))
							colj[row] = lj;
						if (LDLT && wave == 0 && lane == 0)
							s_d[j0 + j] = dj;
						__builtin_amdgcn_wave_barrier();
						T mult[POTRF_PB];
#pragma unroll
						for (int k = j + 2; k < POTRF_PB; ++k)
							mult[k] = colj[j0 + k] * wj;
						if (j + 1 < POTRF_PB) {
							// critical path: next diagonal entry through v_readlane, then its pivot
							a[j + 1] = fh_fma(-lj, lane_bcast(lj, j + 1) * wj, a[j + 1]);
							ok = pivot(lane_bcast(a[j + 1], j + 1), j0 + j + 1, inv, dj);
						}
#pragma unroll
						for (int k = j + 2; k < POTRF_PB; ++k)
							a[k] = fh_fma(-lj, mult[k], a[k]); // a_ik -= l_ij (d_j) l_kj
					}
				}
			}
			if (fail_col != 0 && tid == 0)
				s_fail = fail_col;
		}
		__syncthreads();
		FH_LT(1);
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
					acc = Mfma<T>::run(pa[kk * LDS_LDP], LDLT ? pb[kk * LDS_LDP] * s_d[j0 + kk + lhi] : pb[kk * LDS_LDP], acc);
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const int gi = t0 + ti * 16 + Mfma<T>::row(r, lhi), gj = t0 + tj * 16 + l15;
					if (gi >= gj)
						S[gj * LDS_LDP + gi] -= acc[r];
				}
			}
		}
		__syncthreads();
		FH_LT(2);
	}
	// ---- write back the lower triangle (also after a failure: the columns before the failing one are final)
	if constexpr (LDLT) {
		// D for the columns 0 .. index (cholesky/ldlt/factor.rs:791-798), both on the diagonal of A and in Dout
		const int init = failed ? s_fail : n;
		__syncthreads();
		if (tid < n && tid < init) {
			S[tid * LDS_LDP + tid] = s_d[tid];
			Dout[offset + tid] = s_d[tid];
		}
		__syncthreads();
	}
	lds_store_block<T>(S, A, rs, cs, n, true);
	FH_LT(3);
	if (failed) {
		if (tid == 0)
			atomicCAS(status, 0, offset + s_fail);
		return;
	}
	if (tid == 0 && count > 0)
		atomicAdd(status + 1, count);
	if (Winv) {
		// packed image of L_kk for the substitution leaf of the panel solves (unit diagonal for LDLT)
		typedef TriPack<T> P;
		for (int e = tid; e < TP_NT * P::DG_SZ; e += LDS_NT) // alignment holes of the diagonal tiles
			Winv[P::OFF_DG + e] = (T) 0;
		__syncthreads();
		for (int e = tid; e < LDS_NB * LDS_NB; e += LDS_NT) {
			const int i = e % LDS_NB, j = e / LDS_NB;
			if (j > i)
				continue;
			const T v = S[j * LDS_LDP + i];
			bool neg;
			const int ps = P::pos(i, j, neg);
			const T val = i == j ? ((LDLT || i >= n) ? (T) 1 : (T) 1 / v) : v; // the diagonal enters as its reciprocal
			Winv[ps] = neg ? -val : val;
		}
		FH_LT(4);
	}
#ifdef FH_LEAF_TIMING
	if (tid == 0)
		for (int i = 0; i < 8; ++i)
			atomicAdd(&g_leaf_timing[i], (unsigned long long) tacc[i]);
#endif
}

template <typename T>
static void potrf_rec(MatV<T> A, int regularize, T eps, T delta, int *status, idx_t offset, T *Wbase, bool need_inv)
{
	const idx_t n = A.nrows;
	if (n == 0)
		return;
	if (n <= POTRF_NB) {
		T *W = need_inv ? Wbase + (size_t) (offset / POTRF_NB) * TriPack<T>::SIZE : nullptr;
		ProfScope prof(5, (double) n);
		hipLaunchKernelGGL((potrf_leaf_kernel<T, false>), dim3(1), dim3(LDS_NT), 0, ctx().stream, A.p, A.rs, A.cs, (int) n, regularize,
				   eps, delta, status, (int) offset, W, (const signed char *) nullptr, (T *) nullptr);
		FH_HIP(hipGetLastError());
		return;
	}
	const idx_t h = ((n / 2 + POTRF_NB - 1) / POTRF_NB) * POTRF_NB;
	MatV<T> A00 = A.sub(0, 0, h, h), A10 = A.sub(h, 0, n - h, h), A11 = A.sub(h, h, n - h, n - h);
	potrf_rec<T>(A00, regularize, eps, delta, status, offset, Wbase, true);
	// A10 <- A10 L00^-T, expressed like the reference (cholesky/ldlt/factor.rs:422-426) as L00 \ A10^T
	trsm_lower_pre_dev<T>(A00.c(), A10.t(), Wbase + (size_t) (offset / POTRF_NB) * TriPack<T>::SIZE);
	// lower(A11) -= A10 A10^T  (cholesky/ldlt/factor.rs:436-446 -> triangular.rs:602 DstKind::Lower)
	gemm_dev<T>(A11, DST_LOWER, true, A10.c(), A10.t().c(), (T) -1);
	potrf_rec<T>(A11, regularize, eps, delta, status, offset + h, Wbase, need_inv);
}

// Factorization of a tall panel P (R x w, R >= w; the top w x w block is the diagonal block) in 128-column blocks -- three
// launches per block on ONE dependent chain, right-looking like the reference's own sweep (cholesky/ldlt/factor.rs:367-498
// with its 128-column step, :392):
//     leaf on the diagonal block                               (also yields the packed image W_j of L_jj)
//     rows below      <-  rows below * L_jj^-T                 (substitution leaf against W_j, trsm.hip)
//     panel right of block j  -=  P[c1:, j] P[c1:w, j]^T       (lower trapezoid, ONE GEMM with K = 128)
// instead of the ~38 launches of the recursion above for 8 blocks.  Used where the diagonal-block chain is the critical
// path (look-ahead panel stream, the sequential tail, the distributed driver's panels).
// Rounds 2-4 ran this LEFT-looking (block column j -= P[c0:, 0:c0] P[c0:c0+128, 0:c0]^T before its leaf, K = c0): on the 32
// reserved CUs those launches were 15-59 us (a K = 896 product on a 128-column output has 4-16 workgroups), the K = 128
// trapezoids are 12-25 us.  Measured (profiles/r05_exp_llt_driver.txt): N = 16384 34.7-35.0 -> 34.0-34.1 ms, N = 8192 10.4-11.5
// -> 9.6-9.7 ms; with the cheaper diagonal chains the look-ahead steps beat the sequential tail down to 1024 rows: 33.2 ms.
template <typename T, typename AfterBlock>
static void potrf_panel_flat_hook(MatV<T> P, int regularize, T eps, T delta, int *status, idx_t offset, T *Wbase, idx_t wblk0, AfterBlock after_block)
{
	// the inverse of the 128-block starting at global column offset + c0 goes to slot (offset + c0) / 128 - wblk0 of Wbase
	const idx_t R = P.nrows, w = P.ncols;
	for (idx_t c0 = 0; c0 < w; c0 += POTRF_NB) {
		const idx_t nb = POTRF_NB < w - c0 ? POTRF_NB : w - c0;
		T *W = Wbase + (size_t) ((offset + c0) / POTRF_NB - wblk0) * TriPack<T>::SIZE;
		MatV<T> D = P.sub(c0, c0, nb, nb);
		{
			ProfScope prof(5, (double) nb);
			hipLaunchKernelGGL((potrf_leaf_kernel<T, false>), dim3(1), dim3(LDS_NT), 0, ctx().stream, D.p, D.rs, D.cs, (int) nb, regularize, eps,
					   delta, status, (int) (offset + c0), W, (const signed char *) nullptr, (T *) nullptr);
		}
		FH_HIP(hipGetLastError());
		if (R > c0 + nb) // rows below <- rows below * L_kk^-T: substitution leaf, lanes along the rows of the panel
			trsm_lower_pre_dev<T>(D.c(), P.sub(c0 + nb, c0, R - c0 - nb, nb).t(), W);
		after_block(c0, nb, W); // (block column c0 of L is final, the packed image of its diagonal block is in W)
		if (c0 + nb < w) {
			const idx_t c1 = c0 + nb;
			gemm_dev<T>(P.sub(c1, c1, R - c1, w - c1), DST_LOWER, true, P.sub(c1, c0, R - c1, nb).c(), P.sub(c1, c0, w - c1, nb).t().c(), (T) -1);
		}
	}
}
template <typename T>
static void potrf_panel_flat(MatV<T> P, int regularize, T eps, T delta, int *status, idx_t offset, T *Wbase, idx_t wblk0 = 0)
{
	potrf_panel_flat_hook<T>(P, regularize, eps, delta, status, offset, Wbase, wblk0, [](idx_t, idx_t, T *) {});
}

// Right-looking driver with look-ahead for large matrices: steps of LA_NB columns,
//     [panel stream]  D_k = chol(A_kk)                         (recursive driver above; latency bound, few CUs)
//     [bulk stream]   P_k = A_{>k,k} L_kk^-T                   (in place: substitution leaves + MFMA products)
//     [bulk stream]   A_{k+1,k+1} -= P_k[0] P_k[0]^T           -> releases D_{k+1} on the panel stream
//     [bulk stream]   rest of the trailing matrix -= P_k P_k^T (K = LA_NB: compute bound)
// The diagonal-block factorizations -- a chain of ~130 small dependent launches each -- run concurrently with
// the trailing update of the previous step on CUs reserved for them (Ctx::lookahead_streams), instead of
// leaving 255 CUs idle.  Same arithmetic per entry as the reference's right-looking sweep
// (cholesky/ldlt/factor.rs:367-498) with a larger step.
constexpr idx_t LA_NB = 1024;
#ifndef LLT_SIDE_RMIN
#define LLT_SIDE_RMIN 8192 // rows of the remaining lower square from which the panel solve runs on the side stream
#endif

// Step plan of the blocked driver (pure host logic, unit tested without a GPU through faer_hip_debug_llt_plan):
// starts of the look-ahead panels; the last entry is where the sequential tail takes over.  The first step is LA_NB
// wide, the following ones nb2 while at least 2 * nb2 rows remain behind them, LA_NB again towards the end; steps
// stop once no more than `tail_rows` rows remain (or the next panel would reach the end of the matrix).
std::vector<idx_t> llt_plan(idx_t n, idx_t tail_rows, idx_t nb2)
{
	// width of the first look-ahead step: the whole chip waits for the first diagonal block, so it is ONE 128-block (a
	// leaf, ~55 us) instead of a 1024-wide one (~1.05 ms): 41.1 -> 40.1 ms at N = 16384
	const idx_t first = POTRF_NB;
	const idx_t second = 512; // (measured: 0 / 256 / 512: 33.54-33.56 / 33.34 / 33.21-33.30 ms at N = 16384)
	std::vector<idx_t> J;
	J.push_back(0);
	while (true) {
		const idx_t j0 = J.back();
		idx_t w = J.size() == 1 ? LA_NB : nb2;
		if (n - j0 - w < 2 * w)
			w = LA_NB; // narrow steps again towards the end
		if (J.size() == 1 && first > 0 && first < w)
			w = first; // (the whole chip waits for the first diagonal block)
		// the SECOND step: its diagonal chain + panel solve have only the K = `first` update of the whole matrix to hide
		// behind (0.73 ms at N = 16384; a 1024-wide step needs 1.1 + 0.6 ms: the bulk stream idled 1.06 ms) -- LLT_SECOND wide
		if (J.size() == 2 && first > 0 && second > 0 && second < w && n - j0 - second >= 2 * w)
			w = second;
		if (!(n - j0 > tail_rows && j0 + w < n))
			break;
		J.push_back(j0 + w);
	}
	return J;
}

template <typename T>
static void potrf_lookahead(MatV<T> A, int regularize, T eps, T delta, int *status, T *Wbase, hipStream_t caller)
{
	Ctx &c = ctx();
	const idx_t n = A.nrows;
	// Once the remaining matrix is small the chain "diagonal block -> panel solve -> next diagonal block" is longer
	// than the trailing update it is meant to hide behind: the tail is factored on the caller's stream, whole chip,
	// one tall panel (potrf_panel_flat) + ONE trailing update per step.
	// Look-ahead pays from ~10k rows upwards (measured, N = 8192: 13.3 ms sequential against 13.8 ms); below that and
	// for the last `tail_rows` rows of a large matrix the steps run back to back on the caller's stream.
	// (round 5, right-looking diagonal chains: look-ahead from 8192 rows on -- 9.5 ms either way there, 6144: 6.2 sequential against
	// 6.5 -- and down to the last 1024 rows: tails of 4096 / 3072 / 2048 / 1024 rows 34.1 / 33.6-33.7 / 33.4 / 33.2 ms at N = 16384)
	const idx_t tail_rows = getenv("FAER_HIP_LLT_TAIL") ? atol(getenv("FAER_HIP_LLT_TAIL")) : (n < 8 * LA_NB ? n : LA_NB);
	// Step widths of the look-ahead part: the FIRST step is LA_NB wide (its diagonal block is factored with the rest
	// of the chip idle), the following ones LA_NB2 (wider steps: K = LA_NB2 trailing updates run closer to the dense
	// rate and there are fewer launch boundaries per factorization) while at least 2 * LA_NB2 rows remain.
	const idx_t nb2 = LA_NB; // (wider later steps measured no gain in round 2)
	const std::vector<idx_t> J = llt_plan(n, tail_rows, nb2); // look-ahead steps: panel columns [J[k], J[k + 1])
	idx_t ks = (idx_t) J.size() - 1; // look-ahead steps
	if (ks > 0 && !c.lookahead_streams())
		ks = 0;
	const idx_t tail0 = ks > 0 ? J[(size_t) ks] : 0;
	if (ks > 0) {
		c.reset_events();
		hipEvent_t e0 = c.next_event();
		FH_HIP(hipEventRecord(e0, caller));
		stream_wait(c.la_bulk, e0);
		stream_wait(c.la_panel, e0);
		hipEvent_t ev_diag; // D_k factored (its packed 128-blocks are in Wbase)
		c.qr_side_streams();
		hipStream_t side = c.qr_side[0];
		// Round 5: the late steps.  Once the trailing products are shorter than the chains (fewer than LLT_SIDE_RMIN rows below the
		// next panel) a step was the SUM of two chains: the diagonal block D_{k+1} on the panel stream (24 launches, 0.97 ms), then
		// the solve of all of P_{k+1} on the bulk stream (15 launches, 0.41 ms), then the product of D_{k+2}: 1.48 ms per 1024
		// columns with the trailing product hidden beside the first chain (kernel trace, profiles/r05_exp_llt_driver.txt).  But
		// D_{k+2} needs only the TOP rows X0_{k+1} of P_{k+1} (the next diagonal block's rows), and those can be solved block column
		// by block column right behind the leaves of D_{k+1}: a FOLLOWER on the side stream -- per 128-block one small product and one
		// substitution leaf on w2 rows, behind an event of the leaf that produced the block's packed image -- finishes one block
		// behind the diagonal chain.  The solve of the rows below X0 then runs on the bulk stream BESIDE the next diagonal chain.
		// For the follower to start early the bulk stream brings X0's rows up to date in a launch of their own (and the diagonal
		// block two steps ahead, which shares those rows) before the rest of the product.
		// Measured (profiles/r05_exp_llt_driver.txt): a late step 1.47 -> 1.36 ms, N = 16384 35.1-35.4 -> 34.9-35.0 ms -- less than the
		// 0.4 ms per step the chains promise: the follower cannot start before the bulk stream has solved the rows that update X0
		// (0.6 ms into the step), its small products run ~5 x slower beside the trailing product, and with the extra launches the
		// bulk stream's own chain (solve 0.41 + four products) is now as long as the panel stream's.
		// (re-swept with the right-looking chains: side solve from 4096 / 6144 / 8192 / 10240 rows 33.7 / 33.0-33.2 / 33.1-33.6 / 33.6-34.0 ms,
		// next diagonal block's product on the panel stream from 4096 / 8192 / 12288 rows 33.2 / 33.1-33.6 / 33.6-33.9: both left at 8192)
		const idx_t side_rmin = LLT_SIDE_RMIN, dpanel_rmin = 8192;
		const bool x_follow = true;
		auto rows_below = [&](idx_t kk) { return n - J[(size_t) kk + 1]; };
		auto solved_on_side = [&](idx_t kk) { return kk >= 1 && rows_below(kk) >= side_rmin; }; // (decided in step kk - 1)
		auto follow = [&](idx_t kk) { return x_follow && kk + 1 < ks && !solved_on_side(kk); };
		const bool lend = (g_lend_cus.load() & 1) != 0;
		hipEvent_t ev_x0 = nullptr;     // X0_k solved by the follower (implies D_k factored)
		hipEvent_t ev_x0upd = nullptr;  // the rows of X0_{k+1} are up to date with panel k (bulk stream)
		// D_kk on the current (panel) stream, with the follower for X0_kk if that panel is solved that way
		auto factor_diag = [&](idx_t kk, hipEvent_t x0upd) {
			const idx_t jj0 = J[(size_t) kk], jj1 = J[(size_t) kk + 1], ww = jj1 - jj0;
			MatV<T> D = A.sub(jj0, jj0, ww, ww);
			if (!follow(kk)) {
				potrf_panel_flat<T>(D, regularize, eps, delta, status, jj0, Wbase);
			} else {
				const idx_t ww1 = J[(size_t) kk + 2] - jj1;
				MatV<T> X0f = A.sub(jj1, jj0, ww1, ww);
				bool first = true;
				potrf_panel_flat_hook<T>(D, regularize, eps, delta, status, jj0, Wbase, 0, [&](idx_t c0, idx_t nb, T *W) {
					hipEvent_t el = c.next_event();
					FH_HIP(hipEventRecord(el, c.la_panel));
					StreamScope ss(side);
					if (first && x0upd)
						stream_wait(side, x0upd);
					first = false;
					stream_wait(side, el);
					// RIGHT-looking on these rows: solve block column c0, then take it out of the block columns right of it
					// (K = 128 products: ~15 us each; the left-looking form with K = c0 on w1 x 128 outputs took ~100 us per block
					// and the follower finished 0.34 ms behind the diagonal chain -- kernel trace r5v29)
					trsm_lower_pre_dev<T>(A.sub(jj0 + c0, jj0 + c0, nb, nb).c(), X0f.sub(0, c0, ww1, nb).t(), W);
					const idx_t c1 = c0 + nb;
					if (c1 < ww)
						gemm_dev<T>(X0f.sub(0, c1, ww1, ww - c1), DST_FULL, true, X0f.sub(0, c0, ww1, nb).c(), A.sub(jj0 + c1, jj0 + c0, ww - c1, nb).t().c(), (T) -1);
				});
				ev_x0 = c.next_event();
				FH_HIP(hipEventRecord(ev_x0, side));
			}
			ev_diag = c.next_event();
			FH_HIP(hipEventRecord(ev_diag, c.la_panel));
		};
		{
			StreamScope sc(c.la_panel);
			factor_diag(0, nullptr);
		}
		// trailing size from which the update of the next diagonal block runs on the panel stream (it has slack to
		// spare while the trailing matrix is large, and the bulk stream then issues fewer launches per step)
		// Round 4: the panel solve of step k + 1 (a dependent chain of ~15 small launches, ~0.4 ms whatever the number of rows:
		// 6.4 of the bulk stream's 34 ms, profiles/r03_llt_timeline.txt) no longer sits between two trailing updates on the bulk
		// stream.  Update k brings block column k + 1 up to date FIRST (its own launch), and while the rest of update k -- the
		// lower square right of it, which neither reads nor writes that block column -- runs on the bulk stream, a plain side
		// stream solves P_{k+1} = A_{>k+1,k+1} L_{k+1,k+1}^-T as soon as the panel stream has factored the diagonal block.
		// Its small kernels find their slots among the product's workgroups (the two launches are independent); towards the
		// end, where the rest of the update is shorter than the chain, the chain is exposed as before.
		hipEvent_t ev_solved = nullptr; // P_k solved (recorded on the stream that did it)
		// Lending the panel stream's idle CUs to the big product of a step (as in getrf.hip, getrf_lookahead): the diagonal chain of
		// step k + 1 takes ~1 ms, the trailing product beside it up to 4 ms.  The product's tiles are handed out through per-XCD
		// counters; a helper launch of the same product queued on the panel stream behind the diagonal chain takes tiles on the
		// reserved CUs while more than a margin remain; the next step's bulk work waits for both.
		// MEASURED, OFF BY DEFAULT (faer_hip_debug_lend_cus; profiles/r06_exp_lend.txt): the helped products end 0.3-0.45 ms earlier
		// per step, and the factorization takes as long as before -- the reserved CUs were not idle in effect: the side stream's
		// solve chain ran on them, and among the product's workgroups its kernels take 110 instead of 69 us; the chain "block column
		// up to date -> D_{k+1} -> solve of P_{k+1}" is as long as the product it hides behind (3.5-3.7 ms in the early steps).
		constexpr int LLT_TICKETS = 32;
		Scratch tickb((size_t) LLT_TICKETS * 8 * sizeof(int));
		{
			StreamScope sb(c.la_bulk);
			FH_HIP(hipMemsetAsync(tickb.p, 0, (size_t) LLT_TICKETS * 8 * sizeof(int), c.la_bulk));
		}
		int tick_used = 0;
		struct HelpJob {
			bool on = false;
			MatV<T> C;
			MatV<const T> X;
			GemmExtra<T> ex;
			hipEvent_t ev_in = nullptr;
		} help;
		hipEvent_t ev_help_prev = nullptr;
		// Cd(lower) -= Xd Xd^T on the bulk stream (tri_skip in exb); helped when a diagonal chain follows on the panel stream
		auto big_lower = [&](MatV<T> Cd, MatV<const T> Xd, GemmExtra<T> exb, bool helped) {
			const idx_t tr = (Cd.nrows + 127) / 128, sk = exb.tri_skip / 128;
			const idx_t tiles = tr * (tr + 1) / 2 - sk * (sk + 1) / 2;
			if (helped && lend && tiles >= 2048 && tick_used < LLT_TICKETS) {
				exb.ticket = tickb.as<int>() + 8 * (tick_used++);
				help.on = true;
				help.C = Cd;
				help.X = Xd;
				help.ex = exb;
				help.ex.helper_wgs = (int) (tiles / 8 + 64);
				help.ex.helper_margin = 2 * (c.ncu > 0 ? c.ncu - c.la_panel_cus : 224) / 8 * 2;
				help.ev_in = c.next_event();
				FH_HIP(hipEventRecord(help.ev_in, c.la_bulk));
			}
			gemm_dev<T>(Cd, DST_LOWER, true, Xd, Xd.t(), (T) -1, &exb);
		};
		for (idx_t k = 0; k < ks; ++k) {
			const idx_t j0 = J[(size_t) k], j1 = J[(size_t) k + 1], w = j1 - j0; // panel columns [j0, j1)
			const idx_t r = n - j1;						       // rows below
			const bool last = k + 1 == ks;					       // the tail driver takes over after this step
			const idx_t w1 = last ? 0 : J[(size_t) k + 2] - j1;		       // width of the next look-ahead panel
			MatV<T> Pk = A.sub(j1, j0, r, w);
			MatV<const T> X = Pk.c(), X0 = Pk.sub(0, 0, w1, w).c();
			const bool fol_k = follow(k), fol_n = !last && follow(k + 1);
			const bool d_on_panel = !last && dpanel_rmin > 0 && r >= dpanel_rmin && !fol_k;
			// (the side-stream solve pays only while the rest of the update is much longer than the chain it hides)
			const bool side_solve = !last && r - w1 >= side_rmin;
			hipEvent_t ev_upd, ev_col = nullptr;
			{
				StreamScope sc(c.la_bulk);
				if (ev_help_prev) {
					stream_wait(c.la_bulk, ev_help_prev); // (the helper's tiles of the previous big product)
					ev_help_prev = nullptr;
				}
				if (ev_solved) {
					stream_wait(c.la_bulk, ev_solved);
				} else if (fol_k) {
					// X0_k came from the follower: the next diagonal block at once (-> the panel stream), then the rows below X0_k
					stream_wait(c.la_bulk, ev_x0);
					gemm_dev<T>(A.sub(j1, j1, w1, w1), DST_LOWER, true, X0, X0.t(), (T) -1);
					ev_col = c.next_event();
					FH_HIP(hipEventRecord(ev_col, c.la_bulk));
					if (r > w1)
						trsm_lower_pre_dev<T>(A.sub(j0, j0, w, w).c(), Pk.sub(w1, 0, r - w1, w).t(), Wbase + (size_t) (j0 / POTRF_NB) * TriPack<T>::SIZE);
				} else {
					stream_wait(c.la_bulk, ev_diag);
					// P_k <- P_k L_kk^-T in place (cholesky/ldlt/factor.rs:422-426): the reference's TRSM recursion on the
					// 128-blocks of L_kk -- substitution leaves against the packed diagonal blocks the panel stream
					// left in Wbase, MFMA products in between
					trsm_lower_pre_dev<T>(A.sub(j0, j0, w, w).c(), Pk.t(), Wbase + (size_t) (j0 / POTRF_NB) * TriPack<T>::SIZE);
				}
				ev_upd = c.next_event(); // P_k is solved
				FH_HIP(hipEventRecord(ev_upd, c.la_bulk));
				if (last) {
					gemm_dev<T>(A.sub(j1, j1, r, r), DST_LOWER, true, X, X.t(), (T) -1);
				} else if (!side_solve) {
					if (!d_on_panel && !fol_k) { // next diagonal block first
						gemm_dev<T>(A.sub(j1, j1, w1, w1), DST_LOWER, true, X0, X0.t(), (T) -1);
						ev_col = c.next_event();
						FH_HIP(hipEventRecord(ev_col, c.la_bulk));
					}
					GemmExtra<T> ex;
					ex.tri_skip = w1;
					ev_x0upd = nullptr;
					if (fol_n) {
						// the rows of X0_{k+1} (the follower of the next diagonal block waits for them) and, in the same rows, the
						// diagonal block two steps ahead; the merged product below then skips these rows as well
						// (ONE launch: the rows [w1, w1 + w2) of the lower triangle of the leading (w1 + w2)^2 block)
						const idx_t w2 = J[(size_t) k + 3] - J[(size_t) k + 2];
						MatV<const T> X01 = Pk.sub(0, 0, w1 + w2, w).c();
						GemmExtra<T> exb;
						exb.tri_skip = w1;
						gemm_dev<T>(A.sub(j1, j1, w1 + w2, w1 + w2), DST_LOWER, true, X01, X01.t(), (T) -1, &exb);
						ev_x0upd = c.next_event();
						FH_HIP(hipEventRecord(ev_x0upd, c.la_bulk));
						ex.tri_skip = w1 + w2;
					}
					// block column k+1 below its diagonal block + the remaining lower square in ONE launch: the lower
					// triangle of the whole trailing matrix minus its leading rows
					if (ex.tri_skip < r)
						big_lower(A.sub(j1, j1, r, r), X, ex, true);
				} else {
					ev_x0upd = nullptr;
					if (!d_on_panel && !fol_k) // next diagonal block first (a follower step has done it above)
						gemm_dev<T>(A.sub(j1, j1, w1, w1), DST_LOWER, true, X0, X0.t(), (T) -1);
					// block column k + 1 below its diagonal block
					gemm_dev<T>(A.sub(j1 + w1, j1, r - w1, w1), DST_FULL, true, Pk.sub(w1, 0, r - w1, w).c(), X0.t(), (T) -1);
					ev_col = c.next_event();
					FH_HIP(hipEventRecord(ev_col, c.la_bulk));
					// the remaining lower square
					MatV<const T> X2 = Pk.sub(w1, 0, r - w1, w).c();
					big_lower(A.sub(j1 + w1, j1 + w1, r - w1, r - w1), X2, GemmExtra<T>(), true);
				}
			}
			ev_solved = nullptr;
			if (!last) {
				{
					StreamScope sc(c.la_panel);
					stream_wait(c.la_panel, d_on_panel ? ev_upd : ev_col);
					if (d_on_panel)
						gemm_dev<T>(A.sub(j1, j1, w1, w1), DST_LOWER, true, X0, X0.t(), (T) -1);
					// (the follower of D_{k+1} must not start before X0_{k+1}'s rows are up to date with panel k: in the steps that
					// solve on the side stream the whole block column is updated before ev_col, in the late steps ev_x0upd says so)
					factor_diag(k + 1, fol_n ? (ev_x0upd ? ev_x0upd : ev_col) : nullptr);
					if (help.on) { // the panel stream's CUs join the step's big product until its tiles run low
						stream_wait(c.la_panel, help.ev_in);
						gemm_dev<T>(help.C, DST_LOWER, true, help.X, help.X.t(), (T) -1, &help.ex);
						ev_help_prev = c.next_event();
						FH_HIP(hipEventRecord(ev_help_prev, c.la_panel));
						help.on = false;
					}
				}
				if (side_solve) {
					StreamScope sc(side);
					stream_wait(side, ev_col);
					stream_wait(side, ev_diag);
					trsm_lower_pre_dev<T>(A.sub(j1, j1, w1, w1).c(), A.sub(j1 + w1, j1, r - w1, w1).t(), Wbase + (size_t) (j1 / POTRF_NB) * TriPack<T>::SIZE);
					ev_solved = c.next_event();
					FH_HIP(hipEventRecord(ev_solved, side));
				}
			}
		}
		if (ev_solved) // (cannot happen: the last look-ahead step starts no solve; kept for symmetry)
			stream_wait(caller, ev_solved);
		// rejoin the caller's stream
		hipEvent_t eb = c.next_event(), ep = c.next_event();
		FH_HIP(hipEventRecord(eb, c.la_bulk));
		FH_HIP(hipEventRecord(ep, c.la_panel));
		stream_wait(caller, eb);
		stream_wait(caller, ep);
	}
	// ---- tail (everything, if the matrix is small): sequential, whole chip
	// (steps of 128 / 256 / 512 / 2048 columns here: 34.2-34.3 / 34.3 / 34.4-34.5 ms at N = 16384 against 34.0-34.1; N = 8192: 10.4 / 9.7 / 9.6 / 9.8 against 9.65)
	for (idx_t j0 = tail0; j0 < n; j0 += LA_NB) {
		const idx_t w = LA_NB < n - j0 ? LA_NB : n - j0, R = n - j0;
		potrf_panel_flat<T>(A.sub(j0, j0, R, w), regularize, eps, delta, status, j0, Wbase);
		if (R > w) {
			MatV<const T> P2 = A.sub(j0 + w, j0, R - w, w).c();
			gemm_dev<T>(A.sub(j0 + w, j0 + w, R - w, R - w), DST_LOWER, true, P2, P2.t(), (T) -1);
		}
	}
}

// Tall panel entry point for the distributed driver (dist_llt.h): Cholesky of the top square block of P and the
// solve of the rows below it, right-looking in 128-column blocks (potrf_panel_flat).  status: 2 device ints,
// [0] = first failing global index + 1 (kept if already set), [1] += regularisation count; no synchronisation.
template <typename T> void potrf_panel_dev(MatV<T> P, T reg_delta, T reg_eps, int *status_dev, idx_t offset)
{
	FH_CHECK(P.nrows >= P.ncols, "potrf_panel: the panel must be tall");
	if (P.ncols == 0)
		return;
	const idx_t nblk = (P.ncols + POTRF_NB - 1) / POTRF_NB;
	Scratch winv((size_t) nblk * TriPack<T>::BYTES);
	const int regularize = (reg_delta > (T) 0 && reg_eps > (T) 0) ? 1 : 0;
	potrf_panel_flat<T>(P, regularize, reg_eps, reg_delta, status_dev, offset, winv.as<T>(), offset / POTRF_NB);
}
template void potrf_panel_dev<double>(MatV<double>, double, double, int *, idx_t);
template void potrf_panel_dev<float>(MatV<float>, float, float, int *, idx_t);

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
	Scratch winv(n > POTRF_NB ? (size_t) nblk * TriPack<T>::BYTES : 256);
	const int regularize = (reg_delta > (T) 0 && reg_eps > (T) 0) ? 1 : 0; // cholesky/llt/factor.rs:85-86
	// FAER_HIP_LLT_LA_MIN / FAER_HIP_LLT_TAIL: thresholds of the blocked driver (tests lower them to reach every
	// code path at small sizes)
	const idx_t la_min = getenv("FAER_HIP_LLT_LA_MIN") ? atol(getenv("FAER_HIP_LLT_LA_MIN")) : 2 * LA_NB;
	if (n >= la_min && n > LA_NB)
		potrf_lookahead<T>(A, regularize, reg_eps, reg_delta, status, winv.as<T>(), ctx().stream);
	else
		potrf_rec<T>(A, regularize, reg_eps, reg_delta, status, 0, winv.as<T>(), false);
	int h[2] = {0, 0};
	FH_HIP(hipMemcpyAsync(h, status, sizeof(h), hipMemcpyDeviceToHost, ctx().stream));
	ctx().sync();
	ctx().quiesce();
#ifdef FH_LEAF_TIMING
	{
		unsigned long long d[8];
		FH_HIP(hipMemcpyFromSymbol(d, HIP_SYMBOL(g_leaf_timing), sizeof(d)));
		fprintf(stderr, "leaf timing (cycles, thread 0, cumulative): load %llu | panel %llu | syrk %llu | store %llu | inverse %llu | store W %llu\n",
			d[0], d[1], d[2], d[3], d[4], d[5]);
	}
#endif
	if (h[0] != 0)
		return -(long) h[0]; // -(index + 1)
	return (long) h[1];
}

// ------------------------------------------------------------------------------------------------
// L D L^T (unit lower L, diagonal D; no pivoting) -- cholesky/ldlt/factor.rs:367-498 with is_llt == false,
// SURVEY.md section 8f item 1.  Same recursion by halves and the same leaf; the panel solve is the unit-lower
// TRSM against the leaf inverses, then A10 is scaled by 1/D0 (:447-455) and the trailing update is the
// diagonally weighted product of the spicy_matmul surface (:456-470): lower(A11) -= L10 D0 L10^T.
// ------------------------------------------------------------------------------------------------
template <typename T> __global__ void scale_cols_recip_kernel(T *X, idx_t rs, idx_t cs, idx_t m, idx_t n, const T *__restrict__ d)
{
	const idx_t total = m * n;
	for (idx_t e = (idx_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (idx_t) gridDim.x * blockDim.x) {
		const idx_t i = e % m, k = e / m;
		X[i * rs + k * cs] *= (T) 1 / d[k];
	}
}

template <typename T>
static void sytrf_rec(MatV<T> A, int regularize, T eps, T delta, int *status, idx_t offset, T *Wbase, bool need_inv, const signed char *signs,
		      T *Dv)
{
	const idx_t n = A.nrows;
	if (n == 0)
		return;
	if (n <= POTRF_NB) {
		T *W = need_inv ? Wbase + (size_t) (offset / POTRF_NB) * TriPack<T>::SIZE : nullptr;
		hipLaunchKernelGGL((potrf_leaf_kernel<T, true>), dim3(1), dim3(LDS_NT), 0, ctx().stream, A.p, A.rs, A.cs, (int) n, regularize, eps,
				   delta, status, (int) offset, W, signs, Dv);
		FH_HIP(hipGetLastError());
		return;
	}
	const idx_t h = ((n / 2 + POTRF_NB - 1) / POTRF_NB) * POTRF_NB;
	MatV<T> A00 = A.sub(0, 0, h, h), A10 = A.sub(h, 0, n - h, h), A11 = A.sub(h, h, n - h, n - h);
	sytrf_rec<T>(A00, regularize, eps, delta, status, offset, Wbase, true, signs, Dv);
	// A10 <- A10 L00^-T (unit lower; the leaf inverses were taken of the unit triangles) = L10 D0
	trsm_lower_pre_dev<T>(A00.c(), A10.t(), Wbase + (size_t) (offset / POTRF_NB) * TriPack<T>::SIZE);
	// A10 <- L10 = A10 D0^-1
	{
		const idx_t total = (n - h) * h;
		idx_t blocks = (total + 255) / 256;
		if (blocks > 65536)
			blocks = 65536;
		hipLaunchKernelGGL(scale_cols_recip_kernel<T>, dim3((unsigned) blocks), dim3(256), 0, ctx().stream, A10.p, A10.rs, A10.cs, n - h, h,
				   Dv + offset);
		FH_HIP(hipGetLastError());
	}
	// lower(A11) -= L10 D0 L10^T
	GemmExtra<T> ex;
	ex.diag = Dv + offset;
	ex.diag_stride = 1;
	gemm_dev<T>(A11, DST_LOWER, true, A10.c(), A10.t().c(), (T) -1, &ex);
	sytrf_rec<T>(A11, regularize, eps, delta, status, offset + h, Wbase, need_inv, signs, Dv);
}

// returns >= 0: regularization count, < 0: -(index + 1) of the zero pivot.  `signs_host`: n int8 or NULL.
template <typename T> long sytrf_lower_dev(MatV<T> A, T reg_delta, T reg_eps, const signed char *signs_host)
{
	FH_CHECK(A.nrows == A.ncols, "ldlt: matrix must be square");
	FH_CHECK(A.nrows < (1L << 30), "ldlt: matrix too large");
	if (A.nrows == 0)
		return 0;
	const idx_t n = A.nrows;
	Scratch st(64);
	int *status = st.as<int>();
	FH_HIP(hipMemsetAsync(status, 0, 64, ctx().stream));
	const idx_t nblk = (n + POTRF_NB - 1) / POTRF_NB;
	Scratch winv(n > POTRF_NB ? (size_t) nblk * TriPack<T>::BYTES : 256);
	Scratch dv((size_t) n * sizeof(T)), sg((size_t) n + 256);
	const int regularize = (reg_delta > (T) 0 && reg_eps > (T) 0) ? 1 : 0; // cholesky/ldlt/factor.rs:766-767
	const signed char *signs = nullptr;
	if (signs_host && regularize) {
		FH_HIP(hipMemcpyAsync(sg.p, signs_host, (size_t) n, hipMemcpyHostToDevice, ctx().stream));
		signs = sg.as<signed char>();
	}
	sytrf_rec<T>(A, regularize, reg_eps, reg_delta, status, 0, winv.as<T>(), false, signs, dv.as<T>());
	int h[2] = {0, 0};
	FH_HIP(hipMemcpyAsync(h, status, sizeof(h), hipMemcpyDeviceToHost, ctx().stream));
	ctx().sync();
	ctx().quiesce();
	if (h[0] != 0)
		return -(long) h[0];
	return (long) h[1];
}

template long sytrf_lower_dev<double>(MatV<double>, double, double, const signed char *);
template long sytrf_lower_dev<float>(MatV<float>, float, float, const signed char *);
template long potrf_lower_dev<double>(MatV<double>, double, double);
template long potrf_lower_dev<float>(MatV<float>, float, float);

} // namespace fh
