// Householder QR without pivoting for gfx950.
//
// Replaces faer/src/linalg/qr/no_pivoting/factor.rs:11-301 and the pieces of
// faer/src/linalg/householder.rs it uses (make_householder_imp :59-107, upgrade_householder_factor :132-272,
// apply_block_householder_* :370-808) -- SURVEY.md section 8a rows a24-a30.
//
// Output convention (householder.rs:1-33): R in the upper triangle, the tails of the reflectors v_j
// (v_jj = 1 implicit) below it, and Q_coeff = one upper triangular T per block of `block_size` columns
// with T_jj = tau_j = |v_j|^2 / 2 and T_ij = v_i^H v_j (i < j), so that H_0 .. H_{b-1} = I - V T^-1 V^H.
//
// Two paths, both on the GPU:
//  * FAST (full column rank, the case of every benchmark shape): the reference's recursion on the block
//    size (factor.rs:137-256), with every level-3 step on the MFMA GEMM (split-K for the K = nrows inner
//    products) and an 8-column cooperative leaf: the panel's row chunks stay resident in LDS across the 8
//    column steps, each step needs ONE device-wide reduction (the dot products x^H a_c of the current column
//    with all remaining panel columns are summed in the same pass that would compute its norm, so the
//    reflector, its tau and the update all follow from one all-reduce), plus one more for the 8x8 T block.
//    The rank test of the reference (factor.rs:52-82) is evaluated on the fly; the first column that fails
//    it raises a flag and the factorization is redone from a saved copy by
//  * GENERAL (rank revealing): a literal restatement of qr_in_place_unblocked (factor.rs:11-86) with the
//    running `row` kept in device memory (no host round trip per column), followed by the T blocks
//    T = striu(V^H V) + diag(tau) per block of accepted reflectors.
#include <atomic>
#include <climits>
#include <limits>

#include "common.h"
#include "xwg.h"

namespace fh {

// ------------------------------------------------------------------------------------------------
// block reflector application (householder.rs:370-620, generic path)
// ------------------------------------------------------------------------------------------------
// M <- (I - V T^-H V^H) M  (forward)   or   (I - V T^-1 V^H) M  (!forward); V unit lower trapezoidal m x b
template <typename T> static void apply_block_householder_dev(MatV<const T> V, MatV<const T> Tf, MatV<T> M, bool forward)
{
	const idx_t m = V.nrows, b = V.ncols, k = M.ncols;
	if (b == 0 || k == 0 || m == 0)
		return;
	FH_CHECK(Tf.nrows == b && Tf.ncols == b && M.nrows == m && m >= b, "apply_block_householder: shape mismatch");
	Scratch tmpb((size_t) b * (size_t) k * sizeof(T));
	MatV<T> tmp{tmpb.as<T>(), b, k, 1, b};
	MatV<const T> Vtop = V.sub(0, 0, b, b), Vbot = V.sub(b, 0, m - b, b);
	MatV<T> Mtop = M.sub(0, 0, b, k), Mbot = M.sub(b, 0, m - b, k);
	// tmp = V_top^H M_top + V_bot^H M_bot   (householder.rs:541-563)
	matmul_triangular_dev<T>(tmp, 0, false, Vtop.t(), 6 /*unit upper*/, Mtop.c(), 0, (T) 1);
	if (m > b) {
		GemmExtra<T> big;
		big.prefer_big_tiles = true;
		gemm_dev<T>(tmp, DST_FULL, true, Vbot.t(), Mbot.c(), (T) 1, &big);
	}
	// tmp <- T^-H tmp or T^-1 tmp          (householder.rs:564-578)
	if (forward)
		trsm_lower_dev<T>(Tf.t(), false, tmp);
	else
		trsm_upper_dev<T>(Tf, false, tmp);
	// M -= V tmp                            (householder.rs:579-601)
	matmul_triangular_dev<T>(Mtop, 0, true, Vtop, 5 /*unit lower*/, tmp.c(), 0, (T) -1);
	if (m > b)
		gemm_dev<T>(Mbot, DST_FULL, true, Vbot, tmp.c(), (T) -1);
}

// householder.rs:724-808
template <typename T>
void apply_householder_sequence_left_dev(MatV<const T> V, MatV<const T> H, MatV<T> M, bool transpose)
{
	const idx_t m = V.nrows, n = V.ncols;
	const idx_t size = m < n ? m : n;
	const idx_t block_size = H.nrows;
	FH_CHECK(block_size > 0 && H.ncols == size && M.nrows == m, "apply_householder_sequence: shape mismatch");
	if (transpose) {
		for (idx_t j = 0; j < size;) {
			const idx_t bs = block_size < size - j ? block_size : size - j;
			apply_block_householder_dev<T>(V.sub(j, j, m - j, bs), H.sub(0, j, bs, bs), M.sub(j, 0, m - j, M.ncols),
						       true);
			j += bs;
		}
	} else {
		idx_t j = size;
		idx_t bs = size % block_size;
		if (bs == 0)
			bs = block_size;
		while (j > 0) {
			const idx_t jp = j - bs;
			bs = block_size;
			apply_block_householder_dev<T>(V.sub(jp, jp, m - jp, j - jp), H.sub(0, jp, j - jp, j - jp),
						       M.sub(jp, 0, m - jp, M.ncols), false);
			j = jp;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// shared device helpers
// ------------------------------------------------------------------------------------------------
template <typename T> struct Lim;
template <> struct Lim<double> {
	static constexpr double eps = 2.220446049250313e-16, minpos = 2.2250738585072014e-308;
};
template <> struct Lim<float> {
	static constexpr float eps = 1.1920929e-07f, minpos = 1.17549435e-38f;
};

static __device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1)
		v += __shfl_xor(v, off, 64);
	return v;
}

// ------------------------------------------------------------------------------------------------
// FAST path leaf: cooperative 8-column Householder panel
// ------------------------------------------------------------------------------------------------
constexpr int QR_PW = 8;
constexpr int QR_GMAX = 512;
constexpr int QR_NPAIR = QR_PW * (QR_PW - 1) / 2; // 28
constexpr int QR_SLOT = 32;			  // doubles per workgroup slot (>= QR_NPAIR, >= QR_PW)

template <typename T> struct QrPanelArgs {
	T *P;	     // panel view, row 0 = diagonal row of its first column
	idx_t rs, cs;
	int m, w;
	int R;
	const T *above; // A[0, abs_col0] (rows above the panel), same strides
	int row_abs;	// number of rows above the panel
	T *Tb;		// w x w block of Q_coeff (upper)
	idx_t trs, tcs;
	double *slots; // [2][G][QR_SLOT]
	double *head;  // [2][QR_PW + 1]: row j of the panel (cols j..w) and |above|^2
	xwg_u64 *flags; // [G] per-workgroup epoch flags (xwg.h)
	xwg_u64 *gran;	// [2][G][2 * QR_PW] tagged granules: the QR_PW column sums of every workgroup (column steps)
	xwg_u64 *gran_head; // [2][2 * (QR_PW + 1)]: row j of the panel and |above|^2, published by workgroup 0
	xwg_u64 epoch_base;
	int *status; // [2] exchange timeout, [3] rank deficiency detected
};

// sums `vals[0..cnt)` over the workgroup into s_red[0..cnt) (every thread then reads s_red)
template <int CNT> static __device__ __forceinline__ void block_sum(double (&vals)[CNT], double *s_part, double *s_red)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
	for (int c = 0; c < CNT; ++c) {
		const double s = wave_sum(vals[c]);
		if (lane == 0)
			s_part[wave * CNT + c] = s;
	}
	__syncthreads();
	if (tid < CNT)
		s_red[tid] = s_part[tid] + s_part[CNT + tid] + s_part[2 * CNT + tid] + s_part[3 * CNT + tid];
	__syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Register-resident panel kernel: every thread keeps QR2_RPT whole panel rows (8 columns each) in registers and
// the 8 column steps are unrolled at compile time, so the dot products x^H a_c, the scaling and the rank-1 update
// are register FMAs (in the first, LDS-resident version of this kernel the LDS loops were ~3/4 of its time,
// exactly as in the LU panel: profiles/r01_lu_panel_phase_timing.txt).
// ------------------------------------------------------------------------------------------------
constexpr int QR2_NT = 512;
template <typename T> static constexpr int qr2_rpt() { return sizeof(T) == 8 ? 4 : 8; }

template <int CNT> static __device__ __forceinline__ void block_sum_nt(double (&vals)[CNT], double *s_part, double *s_red)
{
	// like block_sum, for QR2_NT threads: wave sums, then CNT threads add the 8 wave partials
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
	for (int c = 0; c < CNT; ++c) {
		const double sv = wave_sum(vals[c]);
		if (lane == 0)
			s_part[wave * CNT + c] = sv;
	}
	__syncthreads();
	if (tid < CNT) {
		double t = 0.0;
#pragma unroll
		for (int k = 0; k < QR2_NT / 64; ++k)
			t += s_part[k * CNT + tid];
		s_red[tid] = t;
	}
	__syncthreads();
}

template <typename T> struct Qr2Shared {
	double part[(QR2_NT / 64) * QR_SLOT], red[QR_SLOT], S[QR_SLOT];
	double tau[QR_PW], head[QR_PW + 1];
	T top[QR_PW][QR_PW]; // rows 0..7 of the panel (chunk 0): top[r][c]
	int flag;
};

// one column step; false: leave (timeout or rank deficiency recorded in `why`)
template <typename T, int RPT, int J>
static __device__ __forceinline__ bool qr2_step(const QrPanelArgs<T> &a, T (&x)[RPT][QR_PW], Qr2Shared<T> &sh, int r0, int G, int &bar,
						int &why)
{
	const int tid = threadIdx.x;
	const int g = blockIdx.x;
	const int w = a.w;
	const int q = bar & 1;
	// ---- partial sums s_c = sum_{r > J} x_r a_rc, c = J .. w-1 (c == J gives |tail|^2)
	double acc[QR_PW];
#pragma unroll
	for (int c = 0; c < QR_PW; ++c)
		acc[c] = 0.0;
	double ab[1] = {0.0};
#pragma unroll
	for (int i = 0; i < RPT; ++i) {
		const int gr = r0 + tid + i * QR2_NT;
		if (gr > J && gr < a.m) {
			const double xv = (double) x[i][J];
#pragma unroll
			for (int c = J; c < QR_PW; ++c)
				if (c < w)
					acc[c] += xv * (double) x[i][c];
		}
		if (gr < J) { // rows of the panel above the diagonal (chunk 0 only): part of |above|^2
			const double v = (double) x[i][J];
			ab[0] += v * v;
		}
	}
	block_sum_nt<QR_PW>(acc, sh.part, sh.red);
	// row J of the panel lives in thread J of chunk 0
	if (g == 0 && tid == J) {
#pragma unroll
		for (int c = 0; c < QR_PW; ++c)
			sh.top[J][c] = x[0][c];
	}
	if (g == 0) {
		// rows above the panel (outside it)
		for (int i = tid; i < a.row_abs; i += QR2_NT) {
			const double v = (double) a.above[(idx_t) i * a.rs + (idx_t) J * a.cs];
			ab[0] += v * v;
		}
		block_sum_nt<1>(ab, sh.part, sh.S); // sh.S[0] = |above|^2 (also publishes sh.top through its barriers)
	}
	if (G > 1) {
		// All-reduce of the QR_PW column sums with data-tagged granules (xwg.h, recipe R2): {tag, 32 payload bits}
		// per 8-byte write-through store, no store drain, no flag, and the consumers fetch sums AND row j in the
		// same round: one store latency + one load latency per column instead of drain + flag + two dependent
		// reads.  Every thread owns the pairs pe = tid + QR2_NT * i of the G x QR_PW table, i.e. always column
		// c = tid % QR_PW: fixed summation order (i ascending, lanes by xor 8 / 16 / 32, waves ascending).
		const unsigned tag = (unsigned) (a.epoch_base + (xwg_u64) (bar + 1));
		if (tid < QR_PW || (g == 0 && tid <= 2 * QR_PW)) {
			const double v = tid < QR_PW ? sh.red[tid] : tid < 2 * QR_PW ? (tid - QR_PW < w ? (double) sh.top[J][tid - QR_PW] : 0.0) : sh.S[0];
			xwg_u64 *dst = tid < QR_PW ? a.gran + ((size_t) q * G + g) * (2 * QR_PW) + 2 * tid
					       : a.gran_head + (size_t) q * 2 * (QR_PW + 1) + 2 * (tid - QR_PW);
			const xwg_u64 vb = (xwg_u64) __double_as_longlong(v);
			xwg_store_gran(dst, tag, (unsigned) (vb >> 32));
			xwg_store_gran(dst + 1, tag, (unsigned) vb);
		}
		constexpr int MAXI = QR_GMAX * QR_PW / QR2_NT;
		const int np = G * QR_PW;
		xwg_u64 hi[MAXI], lo[MAXI], hh = 0, hl = 0;
		const xwg_u64 *gq = a.gran + (size_t) q * G * (2 * QR_PW);
		const xwg_u64 *hq = a.gran_head + (size_t) q * 2 * (QR_PW + 1) + 2 * (tid <= QR_PW ? tid : 0);
		int ok = 0;
		for (int spin = 0; spin < (1 << 21); ++spin) {
			bool all = true;
#pragma unroll
			for (int i = 0; i < MAXI; ++i) {
				const int pe = tid + QR2_NT * i;
				if (pe < np) {
					hi[i] = xwg_load_gran(gq + 2 * pe);
					lo[i] = xwg_load_gran(gq + 2 * pe + 1);
					all = all && (unsigned) (hi[i] >> 32) == tag && (unsigned) (lo[i] >> 32) == tag;
				}
			}
			if (tid <= QR_PW) {
				hh = xwg_load_gran(hq);
				hl = xwg_load_gran(hq + 1);
				all = all && (unsigned) (hh >> 32) == tag && (unsigned) (hl >> 32) == tag;
			}
			if (__all(all)) {
				ok = 1;
				break;
			}
			__builtin_amdgcn_s_sleep(1);
		}
		if (!__syncthreads_and(ok)) {
			why = 2;
			return false;
		}
		++bar;
		double mine = 0.0;
#pragma unroll
		for (int i = 0; i < MAXI; ++i)
			if (tid + QR2_NT * i < np)
				mine += __longlong_as_double((long long) (((hi[i] & 0xffffffffull) << 32) | (lo[i] & 0xffffffffull)));
		mine += __shfl_xor(mine, 8, 64);
		mine += __shfl_xor(mine, 16, 64);
		mine += __shfl_xor(mine, 32, 64);
		if ((tid & 63) < QR_PW)
			sh.part[(tid >> 6) * QR_PW + (tid & 63)] = mine;
		__syncthreads();
		if (tid < QR_PW) {
			double t = 0.0;
#pragma unroll
			for (int k = 0; k < QR2_NT / 64; ++k)
				t += sh.part[k * QR_PW + tid];
			sh.S[tid] = t;
		}
		if (tid <= QR_PW)
			sh.head[tid] = __longlong_as_double((long long) (((hh & 0xffffffffull) << 32) | (hl & 0xffffffffull)));
	} else {
		__syncthreads(); // sh.S[0] (|above|^2) read below before sh.S is overwritten
		const double above2 = sh.S[0];
		__syncthreads();
		if (tid < QR_PW) {
			sh.S[tid] = sh.red[tid];
			sh.head[tid] = tid < w ? (double) sh.top[J][tid] : 0.0;
		}
		if (tid == 0)
			sh.head[QR_PW] = above2;
	}
	__syncthreads();
	// ---- reflector (householder.rs:59-107), evaluated identically by every thread
	T head = (T) sh.head[J];
	const double above2 = sh.head[QR_PW];
	const T tail_norm = (T) sqrt(sh.S[J]);
	T head_norm = fabs(head);
	if (head_norm < Lim<T>::minpos) {
		head = (T) 0;
		head_norm = (T) 0;
	}
	if (tail_norm < Lim<T>::minpos) {
		// householder.rs:70-77 + factor.rs:59-63: tau = inf, nothing is scaled or updated; the column
		// is accepted iff its head is non zero (e.g. the last column of a square matrix)
		if (!(head_norm > (T) 0)) {
			why = 3;
			return false;
		}
		if (tid == 0)
			sh.tau[J] = (double) std::numeric_limits<T>::infinity();
		__syncthreads();
		return true;
	}
	const T norm = (T) hypot((double) head_norm, (double) tail_norm);
	const T sign = head_norm != (T) 0 ? head * ((T) 1 / head_norm) : (T) 1;
	const T signed_norm = sign * norm;
	const T hinv = (T) 1 / (head + signed_norm);
	const T tn = tail_norm * fabs(hinv);
	const T tau = (T) 0.5 * ((T) 1 + tn * tn);
	// rank test (factor.rs:52-82)
	const T full_norm = (T) hypot((double) norm, sqrt(above2));
	const T threshold = Lim<T>::eps * (T) ((double) (a.m - J) * 16.0) * full_norm;
	const T tau_inv = (T) 1 / tau;
	if (tau_inv < Lim<T>::minpos || !(norm > threshold)) {
		why = 3;
		return false;
	}
	if (tid == 0)
		sh.tau[J] = (double) tau;
	// ---- update: v = x * hinv ; k_c = -(a_jc + v^H a_c) / tau ; a_c += k_c v   (factor.rs:65-80)
	T kc[QR_PW];
#pragma unroll
	for (int c = 0; c < QR_PW; ++c)
		kc[c] = (c > J && c < w) ? -(((T) sh.head[c] + hinv * (T) sh.S[c]) * tau_inv) : (T) 0;
#pragma unroll
	for (int i = 0; i < RPT; ++i) {
		const int gr = r0 + tid + i * QR2_NT;
		if (gr > J && gr < a.m) {
			const T v = x[i][J] * hinv;
			x[i][J] = v;
#pragma unroll
			for (int c = J + 1; c < QR_PW; ++c)
				if (c < w)
					x[i][c] += kc[c] * v;
		}
	}
	if (g == 0 && tid == J) { // row J itself
		x[0][J] = -signed_norm;
#pragma unroll
		for (int c = J + 1; c < QR_PW; ++c)
			if (c < w)
				x[0][c] += kc[c];
	}
	__syncthreads();
	return true;
}

template <typename T, int RPT, int J> struct Qr2Steps {
	static __device__ __forceinline__ bool run(const QrPanelArgs<T> &a, T (&x)[RPT][QR_PW], Qr2Shared<T> &sh, int r0, int G, int &bar,
						   int &why, int steps)
	{
		if constexpr (J < QR_PW) {
			if (J >= steps)
				return true;
			if (!qr2_step<T, RPT, J>(a, x, sh, r0, G, bar, why))
				return false;
			return Qr2Steps<T, RPT, J + 1>::run(a, x, sh, r0, G, bar, why, steps);
		} else {
			return true;
		}
	}
};

template <typename T, int RPT> __global__ __launch_bounds__(QR2_NT) void qr_panel2_kernel(const QrPanelArgs<T> a)
{
	__shared__ Qr2Shared<T> sh;
	constexpr int R = QR2_NT * RPT;
	const int tid = threadIdx.x;
	const int g = blockIdx.x, G = gridDim.x;
	const int r0 = g * R;
	const int w = a.w;
	if (a.status[3] != 0)
		return; // an earlier panel found a rank deficiency: the whole factorization is being abandoned
	T x[RPT][QR_PW];
#pragma unroll
	for (int i = 0; i < RPT; ++i) {
		const int gr = r0 + tid + i * QR2_NT;
#pragma unroll
		for (int c = 0; c < QR_PW; ++c) {
			const bool in = gr < a.m && c < w;
			const T v = a.P[in ? (idx_t) gr * a.rs + (idx_t) c * a.cs : (idx_t) 0];
			x[i][c] = in ? v : (T) 0;
		}
	}
	const int steps = min(w, a.m);
	int bar = 0, why = 0;
	if (!Qr2Steps<T, RPT, 0>::run(a, x, sh, r0, G, bar, why, steps)) {
		if (tid == 0)
			atomicExch(a.status + why, 1);
		return;
	}
	// ---- T block: T_ij = v_i[j] + sum_{r > j} v_ri v_rj  (i < j) ; T_jj = tau_j
	{
		const int q = bar & 1;
		double acc2[QR_NPAIR];
#pragma unroll
		for (int p = 0; p < QR_NPAIR; ++p)
			acc2[p] = 0.0;
#pragma unroll
		for (int i = 0; i < RPT; ++i) {
			const int gr = r0 + tid + i * QR2_NT;
			if (gr < a.m) {
				int p = 0;
#pragma unroll
				for (int jj = 1; jj < QR_PW; ++jj)
#pragma unroll
					for (int ii = 0; ii < jj; ++ii, ++p)
						if (gr > jj && jj < w)
							acc2[p] += (double) x[i][ii] * (double) x[i][jj];
			}
		}
		block_sum_nt<QR_NPAIR>(acc2, sh.part, sh.red);
		// rows 0..7 of chunk 0 (V's top block) for the v_i[j] terms
		if (g == 0 && tid < QR_PW) {
#pragma unroll
			for (int c = 0; c < QR_PW; ++c)
				sh.top[tid][c] = x[0][c];
		}
		if (G > 1) {
			if (tid < QR_NPAIR)
				xwg_store(a.slots + ((size_t) q * G + g) * QR_SLOT + tid, sh.red[tid]);
			if (tid < 64)
				xwg_publish(a.flags, g, a.epoch_base + (xwg_u64) (bar + 1), tid == 0);
			if (!xwg_wait_all(a.flags, G, a.epoch_base + (xwg_u64) (bar + 1), &sh.flag)) {
				if (tid == 0)
					atomicExch(a.status + 2, 1);
				return;
			}
			double tot[QR_NPAIR];
#pragma unroll
			for (int p = 0; p < QR_NPAIR; ++p)
				tot[p] = 0.0;
			if (g == 0) {
				for (int t = tid; t < G; t += QR2_NT)
#pragma unroll
					for (int p = 0; p < QR_NPAIR; ++p)
						tot[p] += xwg_load(a.slots + ((size_t) q * G + t) * QR_SLOT + p);
				block_sum_nt<QR_NPAIR>(tot, sh.part, sh.red);
			}
		}
		__syncthreads();
		if (g == 0 && tid == 0) {
			int p = 0;
			for (int jj = 0; jj < w; ++jj)
				a.Tb[(idx_t) jj * a.trs + (idx_t) jj * a.tcs] = (T) sh.tau[jj];
			for (int jj = 1; jj < QR_PW; ++jj)
				for (int ii = 0; ii < jj; ++ii, ++p)
					if (jj < w)
						a.Tb[(idx_t) ii * a.trs + (idx_t) jj * a.tcs] = (T) ((double) sh.top[jj][ii] + sh.red[p]);
		}
	}
#pragma unroll
	for (int i = 0; i < RPT; ++i) {
		const int gr = r0 + tid + i * QR2_NT;
#pragma unroll
		for (int c = 0; c < QR_PW; ++c)
			if (gr < a.m && c < w)
				a.P[(idx_t) gr * a.rs + (idx_t) c * a.cs] = x[i][c];
	}
}

template <typename T> struct QrWork {
	double *slots, *head;
	xwg_u64 *flags, *gran, *gran_head;
	xwg_u64 epoch_base;
	int *status;
	const T *a_top; // A[0, 0]
	idx_t rs, cs;
};

template <typename T> static constexpr int qr_rmax() { return sizeof(T) == 8 ? 2240 : 4480; }

template <typename T> static void qr_leaf(MatV<T> P, MatV<T> Tb, idx_t row_abs, idx_t col_abs, QrWork<T> &wk)
{
	constexpr int R = QR2_NT * qr2_rpt<T>();
	const idx_t m = P.nrows;
	const int w = (int) P.ncols;
	int G = (int) ((m + R - 1) / R);
	if (G < 1)
		G = 1;
	FH_CHECK(G <= QR_GMAX, "qr: panel too tall for the cooperative kernel");
	QrPanelArgs<T> a;
	a.P = P.p;
	a.rs = P.rs;
	a.cs = P.cs;
	a.m = (int) m;
	a.w = w;
	a.R = R;
	a.above = wk.a_top + col_abs * wk.cs;
	a.row_abs = (int) row_abs;
	a.Tb = Tb.p;
	a.trs = Tb.rs;
	a.tcs = Tb.cs;
	a.slots = wk.slots;
	a.head = wk.head;
	a.flags = wk.flags;
	a.gran = wk.gran;
	a.gran_head = wk.gran_head;
	a.epoch_base = wk.epoch_base;
	a.status = wk.status;
	hipLaunchKernelGGL((qr_panel2_kernel<T, qr2_rpt<T>()>), dim3(G), dim3(QR2_NT), 0, ctx().stream, a);
	FH_HIP(hipGetLastError());
	if (G > 1) {
		const int steps = w < (int) m ? w : (int) m;
		wk.epoch_base += (xwg_u64) (steps + 1);
	}
}

// tsqr.hip: the one-pass path (whole tall matrices: geqrf_dev; single panels of this recursion: qr_rec)
bool tsqr_applicable(idx_t m, idx_t n, idx_t rs, idx_t cs, idx_t bs);
idx_t tsqr_factor(MatV<float> A, MatV<float> H, float *taus, int *reason, idx_t top = 0);
bool tsqr_applicable64(idx_t m, idx_t n, idx_t rs, idx_t cs, idx_t bs, const void *p);
idx_t tsqr_factor64(MatV<double> A, MatV<double> H, double *taus, int *reason, idx_t top = 0);
bool tsqr_panel_applicable(idx_t m, idx_t w, idx_t rs, idx_t cs, const void *p, int elem);
static inline idx_t tsqr_run(MatV<float> A, MatV<float> H, float *taus, int *reason, idx_t top = 0) { return tsqr_factor(A, H, taus, reason, top); }
static inline idx_t tsqr_run(MatV<double> A, MatV<double> H, double *taus, int *reason, idx_t top = 0) { return tsqr_factor64(A, H, taus, reason, top); }

// P: rows from the diagonal row of its first column; Tb: w x w block of Q_coeff
template <typename T> static void qr_rec(MatV<T> P, MatV<T> Tb, idx_t row_abs, idx_t col_abs, QrWork<T> &wk)
{
	const idx_t m = P.nrows, w = P.ncols;
	if (w == 0 || m == 0)
		return;
	if (w <= QR_PW) {
		qr_leaf<T>(P, Tb, row_abs, col_abs, wk);
		return;
	}
	// A panel of up to 64 columns that is tall enough takes the one-pass panel (tsqr.hip): Gram matrix, ONE small kernel, V = P M --
	// instead of 8 cooperative leaves (one all-reduce per column) and the level-3 steps between them.  A panel it refuses (ill
	// conditioned, a column failing the reference's rank test, ...) is untouched and goes down the recursion as before.
	const bool onepass_tall = P.nrows >= 256;
	bool first_done = false; // a 128-column node whose first panel the one-pass path completed (and applied to the second) before it stopped
	if ((w <= 64 || w == 128) && tsqr_panel_applicable(m, w, P.rs, P.cs, P.p, (int) sizeof(T))) {
		Scratch taus((size_t) w * sizeof(T));
		int reason = 0;
		// (rows above the panel in the parent count in the rank test: row_abs of them)
		const idx_t done = tsqr_run(P, MatV<T>{Tb.p, w, w, Tb.rs, Tb.cs}, taus.as<T>(), &reason, row_abs);
		if (done == w)
			return;
		first_done = w == 128 && done == 64; // T11 is complete (one panel), the second panel is untouched but updated
	}
	idx_t w1 = ((w / 2 + QR_PW - 1) / QR_PW) * QR_PW;
	if (w1 >= w)
		w1 = w - QR_PW;
	if (onepass_tall && w > 64) {
		// split at a multiple of 64 so that both halves end in whole one-pass panels where they can
		w1 = ((w / 2 + 63) / 64) * 64;
		if (w1 >= w)
			w1 = w - 64;
	}
	if (first_done)
		w1 = 64;
	const idx_t w2 = w - w1;
	MatV<T> V1 = P.sub(0, 0, m, w1), B = P.sub(0, w1, m, w2);
	MatV<T> T11 = Tb.sub(0, 0, w1, w1), T12 = Tb.sub(0, w1, w1, w2), T22 = Tb.sub(w1, w1, w2, w2);
	if (!first_done) {
		qr_rec<T>(V1, T11, row_abs, col_abs, wk);
		apply_block_householder_dev<T>(V1.c(), T11.c(), B, true); // factor.rs:241-249
	}
	if (m > w1) {
		MatV<T> P2 = P.sub(w1, w1, m - w1, w2);
		qr_rec<T>(P2, T22, row_abs + w1, col_abs + w1, wk);
		// T12 = V1[w1:, :]^H V2 (householder.rs:249-267): V2 unit lower trapezoidal (m - w1) x w2
		const idx_t mt = m - w1;
		const idx_t top = mt < w2 ? mt : w2;
		MatV<const T> V1m = P.sub(w1, 0, top, w1).c(), V2top = P.sub(w1, w1, top, w2).c();
		// rectangular (w1 x top) times unit-lower-trapezoidal top block: treat the top x w2 block as
		// [unit lower | 0]; when mt < w2 only the leading mt x mt part is triangular
		if (top == w2) {
			matmul_triangular_dev<T>(T12, 0, false, V1m.t(), 0, V2top, 5, (T) 1);
			if (mt > w2)
				gemm_dev<T>(T12, DST_FULL, true, P.sub(w1 + w2, 0, mt - w2, w1).c().t(),
					    P.sub(w1 + w2, w1, mt - w2, w2).c(), (T) 1);
		} else {
			FH_CHECK(false, "qr: internal: wide panel in the fast path");
		}
	}
}

// ------------------------------------------------------------------------------------------------
// GENERAL path: qr_in_place_unblocked (factor.rs:11-86) with `row` kept on the device
// ------------------------------------------------------------------------------------------------
struct GqState {
	int row;     // accepted reflectors so far
	int lim;     // min(size, m)
	int mode;    // 1: the scaled tail is written this column
	int apply;   // 1: update the remaining columns
	int accept;  // 1: row += 1 afterwards
	int active;  // 0 once row reached lim
	double acc[6]; // scaled sums: above {sml, med, big}, tail {sml, med, big}
	double hinv, tau_inv;
};

template <typename T> struct GqArgs {
	T *A;
	idx_t rs, cs;
	int m, n, col;
	GqState *st;
	double *dots; // n entries
	double *kvec; // n entries
	T *taus;      // size entries
	int top;      // rows of a parent matrix ABOVE A(0, :) that belong to the columns as well (they count in the rank test)
};

template <typename T> static __device__ __forceinline__ double scale_sml() { return sqrt((double) Lim<T>::minpos); }
template <typename T> static __device__ __forceinline__ double scale_big() { return sqrt(1.0 / (double) Lim<T>::minpos); }

// reductions/norm_l2.rs:6-45: three accumulators of (x*sml)^2, x^2, (x*big)^2, in the scalar type
template <typename T> __global__ void gq_norms_kernel(const GqArgs<T> a)
{
	__shared__ double s_part[4 * 6], s_red[6];
	const int row = a.st->row;
	if (row >= a.st->lim)
		return;
	const T sml = (T) scale_sml<T>(), big = (T) scale_big<T>();
	T acc[6] = {0, 0, 0, 0, 0, 0};
	// (i < 0: the column's entries in the parent's rows above this submatrix -- factor.rs:26,52-58 takes the norm of ALL rows above)
	for (int i = -a.top + (int) (blockIdx.x * blockDim.x + threadIdx.x); i < a.m; i += gridDim.x * blockDim.x) {
		if (i == row)
			continue;
		const T x = a.A[(idx_t) i * a.rs + (idx_t) a.col * a.cs];
		const int o = i < row ? 0 : 3;
		acc[o + 0] += (x * sml) * (x * sml);
		acc[o + 1] += x * x;
		acc[o + 2] += (x * big) * (x * big);
	}
	double accd[6];
	for (int k = 0; k < 6; ++k)
		accd[k] = (double) acc[k];
	block_sum<6>(accd, s_part, s_red);
	if (threadIdx.x < 6)
		atomicAdd(&a.st->acc[threadIdx.x], s_red[threadIdx.x]);
}

template <typename T> static __device__ T norm_from3(const double *acc)
{
	// reductions/norm_l2.rs:173-184
	const T sml = (T) scale_sml<T>(), big = (T) scale_big<T>();
	const T a0 = (T) acc[0], a1 = (T) acc[1], a2 = (T) acc[2];
	if (a0 >= (T) 1)
		return sqrt(a0) * big;
	if (a1 >= (T) 1)
		return sqrt(a1);
	return sqrt(a2) * sml;
}

template <typename T> __global__ void gq_house_kernel(const GqArgs<T> a)
{
	GqState *st = a.st;
	if (threadIdx.x != 0 || blockIdx.x != 0)
		return;
	const int row = st->row;
	st->mode = 0;
	st->apply = 0;
	st->accept = 0;
	st->active = row < st->lim ? 1 : 0;
	if (!st->active)
		return;
	const T norm_above = norm_from3<T>(st->acc);
	const T tail_norm = norm_from3<T>(st->acc + 3);
	for (int k = 0; k < 6; ++k)
		st->acc[k] = 0.0;
	T *hp = a.A + (idx_t) row * a.rs + (idx_t) a.col * a.cs;
	T head = *hp;
	T head_norm = fabs(head);
	T tau, inorm;
	if (head_norm < Lim<T>::minpos) { // householder.rs:66-69
		head = (T) 0;
		head_norm = (T) 0;
		*hp = head;
	}
	if (tail_norm < Lim<T>::minpos) { // householder.rs:70-77
		tau = std::numeric_limits<T>::infinity();
		inorm = head_norm;
	} else {
		const T norm = (T) hypot((double) head_norm, (double) tail_norm);
		const T sign = head_norm != (T) 0 ? head * ((T) 1 / head_norm) : (T) 1;
		const T signed_norm = sign * norm;
		const T hinv = (T) 1 / (head + signed_norm);
		*hp = -signed_norm;
		const T tn = tail_norm * fabs(hinv);
		tau = (T) 0.5 * ((T) 1 + tn * tn);
		inorm = norm;
		st->hinv = (double) hinv;
		st->mode = 1;
	}
	const T full = (T) hypot((double) inorm, (double) norm_above);
	const T threshold = Lim<T>::eps * (T) ((double) (a.m - row) * 16.0) * full;
	const T tau_inv = (T) 1 / tau;
	a.taus[row] = tau; // H[row] = tau (factor.rs:57)
	st->tau_inv = (double) tau_inv;
	if (tau_inv < Lim<T>::minpos) {
		if (inorm > (T) 0)
			st->accept = 1;
	} else if (inorm > threshold) {
		st->apply = 1;
		st->accept = 1;
	}
}

// v = tail * hinv written into column `row` (in place when row == col); when row != col the first
// min(len, col - row) entries of the original tail are zeroed (factor.rs:39-50)
template <typename T> __global__ void gq_scale_kernel(const GqArgs<T> a)
{
	const GqState *st = a.st;
	if (!st->active)
		return;
	const int row = st->row, col = a.col;
	const T hinv = (T) st->hinv;
	for (int i = row + 1 + blockIdx.x * blockDim.x + threadIdx.x; i < a.m; i += gridDim.x * blockDim.x) {
		T *src = a.A + (idx_t) i * a.rs + (idx_t) col * a.cs;
		if (st->mode == 1)
			a.A[(idx_t) i * a.rs + (idx_t) row * a.cs] = *src * hinv;
		if (row != col && i - (row + 1) < col - row)
			*src = (T) 0;
	}
}

template <typename T> __global__ void gq_dots_kernel(const GqArgs<T> a)
{
	__shared__ double s_part[4], s_red[1];
	const GqState *st = a.st;
	if (!st->active || !st->apply)
		return;
	const int row = st->row;
	const int c = a.col + 1 + blockIdx.y;
	double acc[1] = {0.0};
	for (int i = row + 1 + blockIdx.x * blockDim.x + threadIdx.x; i < a.m; i += gridDim.x * blockDim.x)
		acc[0] += (double) a.A[(idx_t) i * a.rs + (idx_t) row * a.cs] * (double) a.A[(idx_t) i * a.rs + (idx_t) c * a.cs];
	block_sum<1>(acc, s_part, s_red);
	if (threadIdx.x == 0)
		atomicAdd(&a.dots[c], s_red[0]);
}

template <typename T> __global__ void gq_heads_kernel(const GqArgs<T> a)
{
	const GqState *st = a.st;
	const int c = a.col + 1 + blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= a.n)
		return;
	if (st->active && st->apply) {
		T *hp = a.A + (idx_t) st->row * a.rs + (idx_t) c * a.cs;
		const T dot = *hp + (T) a.dots[c];
		const T k = -(dot * (T) st->tau_inv);
		*hp += k;
		a.kvec[c] = (double) k;
	}
	a.dots[c] = 0.0;
}

template <typename T> __global__ void gq_update_kernel(const GqArgs<T> a)
{
	const GqState *st = a.st;
	if (!st->active || !st->apply)
		return;
	const int row = st->row;
	const int c = a.col + 1 + blockIdx.y;
	const T k = (T) a.kvec[c];
	for (int i = row + 1 + blockIdx.x * blockDim.x + threadIdx.x; i < a.m; i += gridDim.x * blockDim.x)
		a.A[(idx_t) i * a.rs + (idx_t) c * a.cs] += k * a.A[(idx_t) i * a.rs + (idx_t) row * a.cs];
}

__global__ void gq_advance_kernel(GqState *st)
{
	if (st->active && st->accept)
		st->row += 1;
}

template <typename T> static long qr_general(MatV<T> A, T *taus_dev, idx_t top = 0)
{
	const idx_t m = A.nrows, n = A.ncols;
	const idx_t size = m < n ? m : n;
	hipStream_t s = ctx().stream;
	Scratch stb(sizeof(GqState)), dotsb((size_t) n * 8 + 8), kb((size_t) n * 8 + 8);
	FH_HIP(hipMemsetAsync(stb.p, 0, sizeof(GqState), s));
	FH_HIP(hipMemsetAsync(dotsb.p, 0, (size_t) n * 8 + 8, s));
	GqState init;
	memset(&init, 0, sizeof(init));
	init.lim = (int) size;
	FH_HIP(hipMemcpyAsync(stb.p, &init, sizeof(init), hipMemcpyHostToDevice, s));
	FH_HIP(hipStreamSynchronize(s)); // `init` lives on the stack
	GqArgs<T> a;
	a.A = A.p;
	a.rs = A.rs;
	a.cs = A.cs;
	a.m = (int) m;
	a.n = (int) n;
	a.st = stb.as<GqState>();
	a.dots = dotsb.as<double>();
	a.kvec = kb.as<double>();
	a.taus = taus_dev;
	a.top = (int) top;
	int rb = (int) ((m + 255) / 256);
	if (rb > 1024)
		rb = 1024;
	if (rb < 1)
		rb = 1;
	for (idx_t col = 0; col < n; ++col) {
		a.col = (int) col;
		const int rem = (int) (n - col - 1);
		hipLaunchKernelGGL(gq_norms_kernel<T>, dim3(rb), dim3(256), 0, s, a);
		hipLaunchKernelGGL(gq_house_kernel<T>, dim3(1), dim3(64), 0, s, a);
		hipLaunchKernelGGL(gq_scale_kernel<T>, dim3(rb), dim3(256), 0, s, a);
		if (rem > 0) {
			for (int c0 = 0; c0 < rem; c0 += 32768) {
				GqArgs<T> b = a;
				b.col = (int) col + c0; // columns col+1+c0 ...
				const int nc = rem - c0 < 32768 ? rem - c0 : 32768;
				// dots/update index columns relative to b.col; `row`/v come from the state
				hipLaunchKernelGGL(gq_dots_kernel<T>, dim3(rb, nc), dim3(256), 0, s, b);
			}
			hipLaunchKernelGGL(gq_heads_kernel<T>, dim3((rem + 255) / 256), dim3(256), 0, s, a);
			for (int c0 = 0; c0 < rem; c0 += 32768) {
				GqArgs<T> b = a;
				b.col = (int) col + c0;
				const int nc = rem - c0 < 32768 ? rem - c0 : 32768;
				hipLaunchKernelGGL(gq_update_kernel<T>, dim3(rb, nc), dim3(256), 0, s, b);
			}
		}
		hipLaunchKernelGGL(gq_advance_kernel, dim3(1), dim3(1), 0, s, a.st);
	}
	FH_HIP(hipGetLastError());
	GqState fin;
	FH_HIP(hipMemcpyAsync(&fin, stb.p, sizeof(fin), hipMemcpyDeviceToHost, s));
	FH_HIP(hipStreamSynchronize(s));
	return fin.row;
}

// Q_coeff post-processing.  mode 0: write T_jj = taus[j] for j < rank (general path);
// always: Q_coeff[:, rank..] = 0 and the +inf diagonal of factor.rs:287-299.
template <typename T>
__global__ void qr_finalize_kernel(T *H, idx_t hrs, idx_t hcs, int bs, int size, int rank, const T *taus, int write_tau)
{
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= size)
		return;
	if (j < rank) {
		if (write_tau)
			H[(idx_t) (j % bs) * hrs + (idx_t) j * hcs] = taus[j];
		return;
	}
	for (int i = 0; i < bs; ++i)
		H[(idx_t) i * hrs + (idx_t) j * hcs] = (T) 0;
	// block containing j starts at col0 = j / bs * bs; the diagonal entries from max(rank, col0) on are +inf
	const int col0 = j / bs * bs;
	if (col0 >= rank / bs * bs) {
		const int start = rank > col0 ? rank : col0;
		if (j >= start)
			H[(idx_t) (j - col0) * hrs + (idx_t) j * hcs] = std::numeric_limits<T>::infinity();
	}
}

// T blocks over the first `rank` reflectors from their taus: striu(V^H V) (householder.rs:185-209), diagonal = tau;
// the columns from `rank` on get the zero / +inf pattern of factor.rs:287-299
// striu(V_b^H V_b) of EVERY block of reflectors in one launch (one workgroup per block): the blocks are independent, and
// one block is a 32 x 32 .. 64 x 64 Gram matrix over up to m rows -- as separate split-K GEMM launches they were 127 x
// 118 us = 15 ms of every N = 4096 reduction to condensed form.  Rows in chunks through LDS with the unit-lower structure
// applied on the way in (zeros above the diagonal, 1 on it), sums in fp64 in ascending row order.
// (round 6: a second instantiation for blocks of up to 128 reflectors -- the recommended block size from 2048^2 entries on -- with 512 threads:
// the column-pivot QR at N = 4096 spent 9 of its 134 ms in 126 per-block GEMM launches)
constexpr int TB_ROWS = 32;
template <typename T, int TB_MAXW, int TB_NT>
__global__ __launch_bounds__(TB_NT) void qr_tblock_gram_kernel(const T *V, idx_t vrs, idx_t vcs, int m, int rank, int bs, T *H, idx_t hrs, idx_t hcs)
{
	__shared__ T L[TB_ROWS][TB_MAXW + 1];
	const int tid = threadIdx.x;
	const int c0 = blockIdx.x * bs;
	const int wb = min(bs, rank - c0);
	const int npairs = wb * (wb - 1) / 2;
	constexpr int MAXP = (TB_MAXW * (TB_MAXW - 1) / 2 + TB_NT - 1) / TB_NT;
	double acc[MAXP];
	int pi[MAXP], pj[MAXP];
#pragma unroll
	for (int q = 0; q < MAXP; ++q) {
		acc[q] = 0.0;
		const int p = tid + q * TB_NT;
		// pair p -> (i < j), enumerated column by column: j = 1: (0,1); j = 2: (0,2), (1,2); ...
		int j = 1;
		if (p < npairs) {
			j = (int) ((1.0f + sqrtf(1.0f + 8.0f * (float) p)) * 0.5f);
			while (j * (j - 1) / 2 > p)
				--j;
			while ((j + 1) * j / 2 <= p)
				++j;
		}
		pj[q] = j;
		pi[q] = p < npairs ? p - j * (j - 1) / 2 : 0;
	}
	const int rows = m - c0; // rows of this block's reflectors
	for (int r0 = 0; r0 < rows; r0 += TB_ROWS) {
		__syncthreads();
		for (int e = tid; e < TB_ROWS * wb; e += TB_NT) {
			const int c = e / TB_ROWS, rr = e - c * TB_ROWS, r = r0 + rr; // lanes along the rows (unit stride for column-major V)
			T v = (T) 0;
			if (r < rows)
				v = r < c ? (T) 0 : (r == c ? (T) 1 : V[(idx_t) (c0 + r) * vrs + (idx_t) (c0 + c) * vcs]);
			L[rr][c] = v;
		}
		__syncthreads();
#pragma unroll
		for (int q = 0; q < MAXP; ++q) {
			if (tid + q * TB_NT < npairs) {
				double t = acc[q];
#pragma unroll 8
				for (int rr = 0; rr < TB_ROWS; ++rr)
					t += (double) L[rr][pi[q]] * (double) L[rr][pj[q]];
				acc[q] = t;
			}
		}
	}
#pragma unroll
	for (int q = 0; q < MAXP; ++q)
		if (tid + q * TB_NT < npairs)
			H[(idx_t) pi[q] * hrs + (idx_t) (c0 + pj[q]) * hcs] = (T) acc[q];
}

template <typename T> static void qr_t_blocks_from_taus(MatV<T> A, MatV<T> H, idx_t rank, const T *taus)
{
	const idx_t m = A.nrows, n = A.ncols, bs = H.nrows;
	const idx_t size = m < n ? m : n;
	const bool batched = true; // (one split-K GEMM per block was 15 ms of every N = 4096 reduction: DESIGN.md 3.8)
	if (batched && bs <= 128 && rank > 0) {
		if (bs > 64)
			hipLaunchKernelGGL((qr_tblock_gram_kernel<T, 128, 512>), dim3((unsigned) ((rank + bs - 1) / bs)), dim3(512), 0, ctx().stream, A.p, A.rs, A.cs,
					   (int) m, (int) rank, (int) bs, H.p, H.rs, H.cs);
		else if (bs > 1)
			hipLaunchKernelGGL((qr_tblock_gram_kernel<T, 64, 256>), dim3((unsigned) ((rank + bs - 1) / bs)), dim3(256), 0, ctx().stream, A.p, A.rs, A.cs,
					   (int) m, (int) rank, (int) bs, H.p, H.rs, H.cs);
		FH_HIP(hipGetLastError());
	} else {
		for (idx_t c0 = 0; c0 < rank; c0 += bs) {
			const idx_t wb = bs < rank - c0 ? bs : rank - c0;
			MatV<T> Tb = H.sub(0, c0, wb, wb);
			MatV<const T> Vtop = A.sub(c0, c0, wb, wb).c();
			matmul_triangular_dev<T>(Tb, 6, false, Vtop.t(), 6, Vtop, 5, (T) 1);
			if (m - c0 > wb) {
				MatV<const T> Vbot = A.sub(c0 + wb, c0, m - c0 - wb, wb).c();
				GemmExtra<T> ex;
				ex.dst_strict = true;
				gemm_dev<T>(Tb, DST_UPPER, true, Vbot.t(), Vbot, (T) 1, &ex);
			}
		}
	}
	hipLaunchKernelGGL(qr_finalize_kernel<T>, dim3((unsigned) ((size + 255) / 256)), dim3(256), 0, ctx().stream, H.p, H.rs, H.cs, (int) bs,
			   (int) size, (int) rank, taus, 1);
	FH_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// QR with column pivoting -- faer/src/linalg/qr/col_pivoting/factor.rs:107-395.
// A level-2, HBM-bound algorithm like the reference's: per step the remaining column of largest (down-dated) norm is
// swapped in, its reflector is made, and the trailing rank-1 update is DELAYED by one step and fused with the dot
// products of the next one (update_mat_and_dot_simd, :7-105) unless the best down-dated norm fell below
// sqrt(eps) x the best norm at the last recomputation (:178-203: apply at once, recompute all norms).
// Five small launches per step, no host synchronisation inside the loop; one workgroup per column in the passes
// over the trailing matrix (lanes along the rows), the pivot search over the n norms by one workgroup.
// ------------------------------------------------------------------------------------------------
struct CpState {
	double best_threshold, scale_fwd, scale_bwd;
	double l, tau_inv;
	int delayed, best_col, n_trans, flush; // flush: this step applies the pending update at once and recomputes the norms (:178-203)
	int timeout, pad;		       // the flush blocks of a step did not report (GPU shared with other work)
};

template <typename T> struct CpArgs {
	T *A;
	idx_t rs, cs;
	int m, n, size, k, delayed_ok;
	T *norm, *dot, *taus;
	int *perm;
	xwg_u64 *flags; // per flush block: the step it has finished (cp_step_kernel)
	CpState *st;
};

// sum over the workgroup (256 threads) of `cnt` doubles -> s_red (all threads may read it afterwards)
template <int CNT> static __device__ __forceinline__ void cp_block_sum(double (&v)[CNT], double *s_part, double *s_red)
{
	block_sum<CNT>(v, s_part, s_red);
}

// norm_l2 (reductions/norm_l2.rs) of rows r0.. of column j by one workgroup of 256 threads
template <typename T> static __device__ T cp_col_norm(const CpArgs<T> &a, int r0, int j, double *s_part, double *s_red)
{
	const T sml = (T) scale_sml<T>(), big = (T) scale_big<T>();
	T acc[3] = {0, 0, 0};
	for (int i = r0 + threadIdx.x; i < a.m; i += 256) {
		const T x = a.A[(idx_t) i * a.rs + (idx_t) j * a.cs];
		acc[0] += (x * sml) * (x * sml);
		acc[1] += x * x;
		acc[2] += (x * big) * (x * big);
	}
	double accd[3] = {(double) acc[0], (double) acc[1], (double) acc[2]};
	cp_block_sum<3>(accd, s_part, s_red);
	const T r = norm_from3<T>(s_red);
	__syncthreads();
	return r;
}

template <typename T> __global__ __launch_bounds__(256) void cp_norms_kernel(const CpArgs<T> a)
{
	__shared__ double s_part[4 * 3], s_red[3];
	const int j = blockIdx.x;
	const T v = cp_col_norm<T>(a, 0, j, s_part, s_red);
	if (threadIdx.x == 0)
		a.norm[j] = v;
}

// first maximum (strict '>') of norm[lo .. n)
// Loads / stores that pass the caches (xwg.h): for data another workgroup of the SAME launch has written or will read
static __device__ __forceinline__ double cp_ld(const double *p) { return xwg_load(p); }
static __device__ __forceinline__ float cp_ld(const float *p)
{
	return __int_as_float(__hip_atomic_load(reinterpret_cast<const int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
static __device__ __forceinline__ void cp_st(double *p, double v) { xwg_store(p, v); }
static __device__ __forceinline__ void cp_st(float *p, float v)
{
	__hip_atomic_store(reinterpret_cast<int *>(p), __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T, bool COH = false> static __device__ void cp_argmax(const T *norm, int lo, int n, T &best, int &col, double *s_v, int *s_c)
{
	T bv = (T) 0;
	int bc = lo;
	for (int j = lo + threadIdx.x; j < n; j += blockDim.x) {
		const T v = COH ? cp_ld(norm + j) : norm[j];
		if (v > bv) { // ascending j per thread: the first maximum of the thread's subsequence
			bv = v;
			bc = j;
		}
	}
	double v = (double) bv;
	int cidx = bc;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		const double ov = __shfl_xor(v, off, 64);
		const int oc = __shfl_xor(cidx, off, 64);
		if (ov > v || (ov == v && ov > 0.0 && oc < cidx)) {
			v = ov;
			cidx = oc;
		}
	}
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0) {
		s_v[wave] = v;
		s_c[wave] = cidx;
	}
	__syncthreads();
	v = s_v[0];
	cidx = s_c[0];
	for (int w = 1; w < (int) blockDim.x / 64; ++w)
		if (s_v[w] > v || (s_v[w] == v && s_v[w] > 0.0 && s_c[w] < cidx)) {
			v = s_v[w];
			cidx = s_c[w];
		}
	__syncthreads();
	best = (T) v;
	col = v > 0.0 ? cidx : lo;
}

// factor.rs:142-160: scale by the reciprocal of the largest column norm
template <typename T> __global__ __launch_bounds__(1024) void cp_init_kernel(const CpArgs<T> a)
{
	__shared__ double s_v[16];
	__shared__ int s_c[16];
	T best;
	int col;
	cp_argmax<T>(a.norm, 0, a.n, best, col, s_v, s_c);
	const T scale_bwd = (T) 1 / best;
	for (int j = threadIdx.x; j < a.n; j += 1024) {
		a.norm[j] = a.norm[j] * scale_bwd;
		a.dot[j] = (T) 0;
		a.perm[j] = j;
	}
	if (threadIdx.x == 0) {
		a.st->scale_fwd = (double) best;
		a.st->scale_bwd = (double) scale_bwd;
		a.st->best_threshold = (double) ((best * scale_bwd) * (T) sqrt((double) Lim<T>::eps));
		a.st->n_trans = 0;
	}
}

// A *= scale (all of it with `upper` == 0, the upper triangle with the diagonal otherwise)
template <typename T> __global__ void cp_scale_kernel(const CpArgs<T> a, int upper)
{
	const T sc = (T) (upper ? a.st->scale_fwd : a.st->scale_bwd);
	const idx_t total = (idx_t) a.m * a.n;
	for (idx_t e = (idx_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (idx_t) gridDim.x * blockDim.x) {
		const idx_t i = e % a.m, j = e / a.m;
		if (!upper || i <= j)
			a.A[i * a.rs + j * a.cs] *= sc;
	}
}

// factor.rs:204-252: column swap, the pending update of column k, its reflector (householder.rs:59-107)
// COH: the columns and norms may have been rewritten by the flush blocks of the SAME launch (cp_step_kernel): every read of them
// passes the caches
template <typename T, bool COH = false> static __device__ __forceinline__ void cp_house_body(const CpArgs<T> &a, const int bc, const int delayed)
{
	auto ld = [&](const T *q) -> T { return COH ? cp_ld(q) : *q; };

	__shared__ double s_part[16 * 3], s_red[3];
	const int tid = threadIdx.x, k = a.k;
	if (bc != k) {
		for (int i = tid; i < a.m; i += 1024) {
			T *p = a.A + (idx_t) i * a.rs + (idx_t) k * a.cs, *q = a.A + (idx_t) i * a.rs + (idx_t) bc * a.cs;
			const T x = ld(p), y = ld(q);
			*p = y;
			*q = x;
		}
		if (tid == 0) {
			const int tp = a.perm[k];
			a.perm[k] = a.perm[bc];
			a.perm[bc] = tp;
			const T td = a.dot[k], tn = ld(a.norm + k);
			a.dot[k] = a.dot[bc];
			a.dot[bc] = td;
			a.norm[k] = ld(a.norm + bc);
			a.norm[bc] = tn;
			a.st->n_trans += 1;
		}
	}
	__syncthreads();
	const T l = delayed ? a.A[(idx_t) k * a.rs + (idx_t) (k - 1) * a.cs] : (T) 0;
	const T r = a.dot[k];
	__syncthreads();
	// pending update of column k and the scaled sums of its tail in one pass
	const T sml = (T) scale_sml<T>(), big = (T) scale_big<T>();
	T acc[3] = {0, 0, 0};
	for (int i = k + 1 + tid; i < a.m; i += 1024) {
		T *p = a.A + (idx_t) i * a.rs + (idx_t) k * a.cs;
		T x = ld(p);
		if (delayed) {
			x += r * a.A[(idx_t) i * a.rs + (idx_t) (k - 1) * a.cs];
			*p = x;
		}
		acc[0] += (x * sml) * (x * sml);
		acc[1] += x * x;
		acc[2] += (x * big) * (x * big);
	}
	double accd[3] = {(double) acc[0], (double) acc[1], (double) acc[2]};
	{ // 16 waves
		const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			const double sv = wave_sum(accd[c]);
			if (lane == 0)
				s_part[wave * 3 + c] = sv;
		}
		__syncthreads();
		if (tid < 3) {
			double t = 0.0;
			for (int w = 0; w < 16; ++w)
				t += s_part[w * 3 + tid];
			s_red[tid] = t;
		}
		__syncthreads();
	}
	const T tail_norm = norm_from3<T>(s_red);
	T *hp = a.A + (idx_t) k * a.rs + (idx_t) k * a.cs;
	T head = ld(hp);
	if (delayed)
		head += l * r;
	T head_norm = fabs(head);
	if (head_norm < Lim<T>::minpos) {
		head = (T) 0;
		head_norm = (T) 0;
	}
	T tau, hinv = (T) 0;
	bool scale_tail = false;
	if (tail_norm < Lim<T>::minpos) {
		tau = std::numeric_limits<T>::infinity();
	} else {
		const T norm = (T) hypot((double) head_norm, (double) tail_norm);
		const T sign = head_norm != (T) 0 ? head * ((T) 1 / head_norm) : (T) 1;
		const T signed_norm = sign * norm;
		hinv = (T) 1 / (head + signed_norm);
		head = -signed_norm;
		const T tn = tail_norm * fabs(hinv);
		tau = (T) 0.5 * ((T) 1 + tn * tn);
		scale_tail = true;
	}
	__syncthreads();
	if (scale_tail)
		for (int i = k + 1 + tid; i < a.m; i += 1024)
		{
				T *pp = a.A + (idx_t) i * a.rs + (idx_t) k * a.cs;
				*pp = ld(pp) * hinv;
			}
	if (tid == 0) {
		*hp = head;
		a.taus[k] = tau;
		a.st->tau_inv = (double) ((T) 1 / tau);
		a.st->l = (double) l;
	}
	if (k + 1 == a.size && delayed) // factor.rs:253-262
		for (int j = k + 1 + tid; j < a.n; j += 1024)
			a.A[(idx_t) k * a.rs + (idx_t) j * a.cs] += l * a.dot[j];
}

// cp_house_body with the two swapped columns and column k - 1 in registers (m <= 4 x 1024 rows): one round trip to memory after the pivot
// is known instead of three (swap, pending update, scaling), every entry stored once.  Same arithmetic, expression by expression.
template <typename T, bool COH> static __device__ __forceinline__ void cp_house_body_reg(const CpArgs<T> &a, const int bc, const int delayed)
{
	auto ld = [&](const T *q) -> T { return COH ? cp_ld(q) : *q; };
	constexpr int E = 4;
	__shared__ double s_part[16 * 3], s_red[3];
	__shared__ T s_head;
	const int tid = threadIdx.x, k = a.k, m = a.m;
	const bool sw = bc != k;
	T ck[E], cb[E], c1[E];
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const int i = tid + e * 1024, ic = i < m ? i : m - 1;
		ck[e] = ld(a.A + (idx_t) ic * a.rs + (idx_t) k * a.cs);
		cb[e] = sw ? ld(a.A + (idx_t) ic * a.rs + (idx_t) bc * a.cs) : ck[e];
		c1[e] = delayed ? a.A[(idx_t) ic * a.rs + (idx_t) (k - 1) * a.cs] : (T) 0;
	}
	const T l = delayed ? a.A[(idx_t) k * a.rs + (idx_t) (k - 1) * a.cs] : (T) 0;
	const T r = a.dot[sw ? bc : k];
	T tn_k = (T) 0, tn_b = (T) 0;
	if (tid == 0 && sw) {
		tn_k = ld(a.norm + k);
		tn_b = ld(a.norm + bc);
	}
	__syncthreads(); // every read of dot / norm / perm is done
	if (tid == 0 && sw) {
		const int tp = a.perm[k];
		a.perm[k] = a.perm[bc];
		a.perm[bc] = tp;
		const T td = a.dot[k];
		a.dot[k] = r;
		a.dot[bc] = td;
		a.norm[k] = tn_b;
		a.norm[bc] = tn_k;
		a.st->n_trans += 1;
	}
	// pending update of column k and the scaled sums of its tail
	const T sml = (T) scale_sml<T>(), big = (T) scale_big<T>();
	T acc[3] = {0, 0, 0};
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const int i = tid + e * 1024;
		if (i < m && i >= k + 1) {
			T x = cb[e];
			if (delayed)
				x += r * c1[e];
			cb[e] = x;
			acc[0] += (x * sml) * (x * sml);
			acc[1] += x * x;
			acc[2] += (x * big) * (x * big);
		}
		if (i == k)
			s_head = cb[e];
	}
	double accd[3] = {(double) acc[0], (double) acc[1], (double) acc[2]};
	{ // 16 waves
		const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			const double sv = wave_sum(accd[c]);
			if (lane == 0)
				s_part[wave * 3 + c] = sv;
		}
		__syncthreads();
		if (tid < 3) {
			double t = 0.0;
			for (int w = 0; w < 16; ++w)
				t += s_part[w * 3 + tid];
			s_red[tid] = t;
		}
		__syncthreads();
	}
	const T tail_norm = norm_from3<T>(s_red);
	T head = s_head;
	if (delayed)
		head += l * r;
	T head_norm = fabs(head);
	if (head_norm < Lim<T>::minpos) {
		head = (T) 0;
		head_norm = (T) 0;
	}
	T tau, hinv = (T) 0;
	bool scale_tail = false;
	if (tail_norm < Lim<T>::minpos) {
		tau = std::numeric_limits<T>::infinity();
	} else {
		const T norm = (T) hypot((double) head_norm, (double) tail_norm);
		const T sign = head_norm != (T) 0 ? head * ((T) 1 / head_norm) : (T) 1;
		const T signed_norm = sign * norm;
		hinv = (T) 1 / (head + signed_norm);
		head = -signed_norm;
		const T tn = tail_norm * fabs(hinv);
		tau = (T) 0.5 * ((T) 1 + tn * tn);
		scale_tail = true;
	}
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const int i = tid + e * 1024;
		if (i < m) {
			if (sw)
				a.A[(idx_t) i * a.rs + (idx_t) bc * a.cs] = ck[e];
			T *pk = a.A + (idx_t) i * a.rs + (idx_t) k * a.cs;
			if (i > k) {
				if (sw || delayed || scale_tail)
					*pk = scale_tail ? cb[e] * hinv : cb[e];
			} else if (i == k) {
				*pk = head;
			} else if (sw) {
				*pk = cb[e];
			}
		}
	}
	if (tid == 0) {
		a.taus[k] = tau;
		a.st->tau_inv = (double) ((T) 1 / tau);
		a.st->l = (double) l;
	}
	if (k + 1 == a.size && delayed) // factor.rs:253-262
		for (int j = k + 1 + tid; j < a.n; j += 1024)
			a.A[(idx_t) k * a.rs + (idx_t) j * a.cs] += l * a.dot[j];
}
template <typename T, bool COH> static __device__ __forceinline__ void cp_house(const CpArgs<T> &a, const int bc, const int delayed)
{
	if (a.m <= 4 * 1024)
		cp_house_body_reg<T, COH>(a, bc, delayed);
	else
		cp_house_body<T, COH>(a, bc, delayed);
}

// Round 6: ONE launch per step.  Block 0 (factor.rs:163-177): the best remaining column by the down-dated norms and the decision "delayed
// update or recompute", published to the other blocks of the launch (one flag word: 2 (k + 1) + recompute).  Common case: the column swap
// and the reflector follow in block 0 and the other blocks leave as soon as they see the flag.  Recompute (:178-203, k > 0): blocks 1 ..
// apply the pending update to the trailing columns, A11 += A10[:, k-1] dot[k:], and recompute their norms (a column per block and turn; the
// first 256 threads work, with the strides and the sum order of the 256-thread kernel this replaces), write-through, then raise their flag;
// block 0 waits for the flags, picks the pivot from the fresh norms and makes the reflector -- reading what the other blocks wrote past its
// caches.  Rounds 1-5: select, flush, select2, house = four launches, two of them returning at once on almost every step (3-5 us each).
constexpr int CP_NFL = 15; // helper blocks
template <typename T> __global__ __launch_bounds__(1024) void cp_step_kernel(const CpArgs<T> a)
{
	__shared__ double s_v[16];
	__shared__ int s_c[16];
	__shared__ double s_part[16 * 3], s_red[3];
	__shared__ int s_flag;
	const int tid = threadIdx.x, k = a.k;
	xwg_u64 *dflag = a.flags + CP_NFL;
	if (blockIdx.x > 0) {
		// ---- helper block: wait for block 0's decision
		if (tid == 0) {
			int dec = -1;
			for (int spin = 0; spin < (1 << 21); ++spin) {
				const xwg_u64 v = __hip_atomic_load(dflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if ((v >> 1) == (xwg_u64) (k + 1)) {
					dec = (int) (v & 1);
					break;
				}
				__builtin_amdgcn_s_sleep(2);
			}
			s_flag = dec;
		}
		__syncthreads();
		if (s_flag != 1)
			return; // (no recomputation -- or block 0 never spoke: it reports the time-out itself)
		const T sml = (T) scale_sml<T>(), big = (T) scale_big<T>();
		for (int j = k + (int) blockIdx.x - 1; j < a.n; j += (int) gridDim.x - 1) {
			const T d = a.dot[j];
			T acc[3] = {0, 0, 0};
			if (tid < 256)
				for (int i = k + tid; i < a.m; i += 256) {
					T *p = a.A + (idx_t) i * a.rs + (idx_t) j * a.cs;
					const T x = fh_fma(a.A[(idx_t) i * a.rs + (idx_t) (k - 1) * a.cs], d, *p);
					cp_st(p, x);
					acc[0] += (x * sml) * (x * sml);
					acc[1] += x * x;
					acc[2] += (x * big) * (x * big);
				}
			const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const double sv = wave_sum((double) acc[c]);
				if (lane == 0)
					s_part[wave * 3 + c] = sv;
			}
			__syncthreads();
			if (tid < 3)
				s_red[tid] = s_part[tid] + s_part[3 + tid] + s_part[6 + tid] + s_part[9 + tid]; // (the four working wavefronts, in their order)
			__syncthreads();
			if (tid == 0)
				cp_st(a.norm + j, norm_from3<T>(s_red));
			__syncthreads();
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (tid == 0)
			__hip_atomic_store(a.flags + (blockIdx.x - 1), (xwg_u64) (k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return;
	}
	// ---- block 0
	T best;
	int col;
	cp_argmax<T>(a.norm, k, a.n, best, col, s_v, s_c);
	const int delayed = (a.delayed_ok && k > 0 && (double) best >= a.st->best_threshold) ? 1 : 0;
	const int flush = k > 0 && !delayed;
	__syncthreads(); // (everyone has read the threshold)
	if (tid == 0) {
		if (gridDim.x > 1)
			__hip_atomic_store(dflag, ((xwg_u64) (k + 1) << 1) | (xwg_u64) flush, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		a.st->delayed = delayed;
		a.st->best_col = col;
		a.st->flush = flush;
	}
	if (!flush) {
		cp_house<T, false>(a, col, delayed);
		return;
	}
	// (a time-out cannot be repaired here -- the columns are half rewritten by then --: it is reported through the status word)
	if (!xwg_wait_all(a.flags, (int) gridDim.x - 1, (xwg_u64) (k + 1), &s_flag)) {
		if (tid == 0)
			a.st->timeout = 1;
		return;
	}
	cp_argmax<T, true>(a.norm, k, a.n, best, col, s_v, s_c);
	if (tid == 0) {
		a.st->best_col = col;
		a.st->best_threshold = (double) (best * (T) sqrt((double) Lim<T>::eps));
	}
	cp_house<T, true>(a, col, 0);
}

// factor.rs:266-301 / update_mat_and_dot_simd (:60-98): one workgroup per trailing column
template <typename T> __global__ __launch_bounds__(256) void cp_update_kernel(const CpArgs<T> a)
{
	__shared__ double s_part[4], s_red[1];
	const int k = a.k, j = k + 1 + blockIdx.x, tid = threadIdx.x;
	const int delayed = a.st->delayed;
	const T b0 = a.dot[j];
	T acc = (T) 0;
	for (int i = k + 1 + tid; i < a.m; i += 256) {
		T *p = a.A + (idx_t) i * a.rs + (idx_t) j * a.cs;
		T dst = *p;
		if (delayed) {
			dst = fh_fma(a.A[(idx_t) i * a.rs + (idx_t) (k - 1) * a.cs], b0, dst);
			*p = dst;
		}
		acc = fh_fma(a.A[(idx_t) i * a.rs + (idx_t) k * a.cs], dst, acc);
	}
	double accd[1] = {(double) acc};
	cp_block_sum<1>(accd, s_part, s_red);
	if (tid == 0) {
		const T tau_inv = (T) a.st->tau_inv, l = (T) a.st->l;
		T *up = a.A + (idx_t) k * a.rs + (idx_t) j * a.cs;
		T u;
		if (delayed) {
			const T tmp = *up + l * b0;
			const T d0 = (tmp + (T) s_red[0]) * (-tau_inv);
			u = tmp + d0;
			a.dot[j] = d0;
		} else {
			const T d = -((*up + (T) s_red[0]) * tau_inv);
			u = *up + d;
			a.dot[j] = d;
		}
		*up = u;
		const T nj = a.norm[j];
		a.norm[j] = sqrt(nj * nj - u * u);
	}
}

// A: m x n, H: block_size x min(m, n); col_perm / col_perm_inv: HOST arrays of n entries.  Returns the transposition count.
template <typename T> long colpiv_qr_dev(MatV<T> A, MatV<T> H, idx_t *col_perm, idx_t *col_perm_inv)
{
	const idx_t m = A.nrows, n = A.ncols;
	const idx_t size = m < n ? m : n;
	FH_CHECK(H.nrows > 0 && H.ncols == size, "colpiv_qr: Q_coeff must be block_size x min(nrows, ncols)");
	FH_CHECK(m < (1L << 30) && n < (1L << 30), "colpiv_qr: matrix too large");
	for (idx_t j = 0; j < n; ++j)
		col_perm[j] = col_perm_inv[j] = j;
	if (size == 0)
		return 0;
	hipStream_t s = ctx().stream;
	Scratch nb((size_t) (2 * n + size) * sizeof(T) + 256), pb((size_t) n * sizeof(int) + 256), stb(sizeof(CpState)), flb((size_t) (CP_NFL + 1) * sizeof(xwg_u64));
	CpArgs<T> a;
	a.flags = flb.as<xwg_u64>();
	FH_HIP(hipMemsetAsync(flb.p, 0, (size_t) (CP_NFL + 1) * sizeof(xwg_u64), s));
	a.A = A.p;
	a.rs = A.rs;
	a.cs = A.cs;
	a.m = (int) m;
	a.n = (int) n;
	a.size = (int) size;
	a.k = 0;
	a.delayed_ok = A.rs == 1 ? 1 : 0; // the reference's SIMD path needs column-major storage (factor.rs:174-176)
	a.norm = nb.as<T>();
	a.dot = a.norm + n;
	a.taus = a.dot + n;
	a.perm = pb.as<int>();
	a.st = stb.as<CpState>();
	FH_HIP(hipMemsetAsync(stb.p, 0, sizeof(CpState), s));
	hipLaunchKernelGGL(cp_norms_kernel<T>, dim3((unsigned) n), dim3(256), 0, s, a);
	hipLaunchKernelGGL(cp_init_kernel<T>, dim3(1), dim3(1024), 0, s, a);
	hipLaunchKernelGGL(cp_scale_kernel<T>, dim3(1024), dim3(256), 0, s, a, 0);
	for (idx_t k = 0; k < size; ++k) {
		a.k = (int) k;
		hipLaunchKernelGGL(cp_step_kernel<T>, dim3((unsigned) (k > 0 ? 1 + (n - k < CP_NFL ? n - k : CP_NFL) : 1)), dim3(1024), 0, s, a);
		if (k + 1 < size)
			hipLaunchKernelGGL(cp_update_kernel<T>, dim3((unsigned) (n - k - 1)), dim3(256), 0, s, a);
	}
	hipLaunchKernelGGL(cp_scale_kernel<T>, dim3(1024), dim3(256), 0, s, a, 1);
	FH_HIP(hipGetLastError());
	qr_t_blocks_from_taus<T>(A, H, size, a.taus);
	std::vector<int> hp((size_t) n);
	CpState fin;
	FH_HIP(hipMemcpyAsync(hp.data(), a.perm, (size_t) n * sizeof(int), hipMemcpyDeviceToHost, s));
	FH_HIP(hipMemcpyAsync(&fin, stb.p, sizeof(fin), hipMemcpyDeviceToHost, s));
	FH_HIP(hipStreamSynchronize(s));
	FH_CHECK(!fin.timeout, "colpiv_qr: the blocks that recompute the column norms did not report in time (GPU shared with other work)");
	for (idx_t j = 0; j < n; ++j) {
		FH_CHECK(hp[(size_t) j] >= 0 && hp[(size_t) j] < n, "colpiv_qr: corrupt permutation");
		col_perm[j] = hp[(size_t) j];
	}
	for (idx_t j = 0; j < n; ++j)
		col_perm_inv[col_perm[j]] = j;
	return fin.n_trans;
}
template long colpiv_qr_dev<double>(MatV<double>, MatV<double>, idx_t *, idx_t *);
template long colpiv_qr_dev<float>(MatV<float>, MatV<float>, idx_t *, idx_t *);

// ------------------------------------------------------------------------------------------------
// Tridiagonalization of a self-adjoint matrix -- faer/src/linalg/evd/tridiag.rs:274-535.
// A level-2, HBM-bound algorithm like the reference's: per column ONE pass over the remaining lower triangle that
// applies the symmetric rank-2 update of the previous reflector and multiplies the updated matrix by the new one
// (tridiag_fused_op, :36-272), between two short vector phases.  Two launches per column, no host synchronisation:
//   td_step_kernel(k)   block 0: finishes y of step k-1 (:484-511), brings column k up to date (:300-318), makes its reflector
//                       (:330-336, householder.rs:59-107), updates column k+1 (:348-359), w <- y; blocks 1 ..: add the shares of the
//                       previous pass in a fixed order -> ysum (one block per 64 entries) while block 0 loads its columns, and hand
//                       them over inside the launch (write-through stores + one flag per block, xwg.h: no fence)
//   td_fused_kernel(k)  one workgroup per 128 x 64 tile of the lower triangle of A22 = A[k+2.., k+2..], lanes along the rows:
//                       A22 -= u w^H + w u^H written back (every entry is read and written ONCE) and the tile's share of both halves
//                       of sym(A22) x -- the row sums tril(A22) x and the column sums striu(A22^H) x
// History (N = 4096 fp64): rounds 2-5 read the triangle twice (a column pass that wrote back and a 16-row pass, each sum complete in
// one wavefront / workgroup): 185 ms; cut into uniform pieces with partial sums added by the step kernel: 150 ms (the chip had waited
// for the longest workgroup); one pass over tiles, the sums inside that pass behind a ticket (write-through + atomic, no fence): 140 ms
// -- the tail "drain, ticket, reload, add" is four memory round trips in every launch; the sums as a launch of their own: 116 ms; loads
// without branches, DPP column sums, a step kernel that holds its three columns in registers: 95 ms; the sums as helper blocks of the
// step launch: 88 ms.  Running the vector phase in the last workgroup behind a release fence measured 2.4x slower in round 2 (an
// agent-scope fence writes the L2 back).
// ------------------------------------------------------------------------------------------------
// tests: 1 = the vector kernels of the three reductions run their memory-resident bodies at every size (the bodies that keep their columns
// in registers take over from 4096 remaining rows down)
static std::atomic<int> g_l2_force_mem{0};
void level2_debug_force_memory_bodies(int on) { g_l2_force_mem.store(on); }

struct TdState {
	double tau_inv;
};
template <typename T> struct TdArgs {
	T *A;
	idx_t rs, cs;
	int n, k, force_mem;
	T *y, *w, *taus;
	double *ysum;	       // sym(A22) x of the fused pass, complete
	double *rpart, *cpart; // shares of the tiles: row sums rpart[J * n + i] (column block J), column sums cpart[I * n + j] (row block I)
	xwg_u64 *flags;	       // per index block: the step whose sums are complete (td_sum_block / td_wait_sums)
	TdState *st;
};
constexpr int TD_NT = 1024; // td_step_kernel
constexpr int TD_UNR = 4;   // independent loads in flight per thread in the matrix passes (bidiagonalization, Hessenberg)
constexpr int TD_PW = 16;   // rows per workgroup of their row passes (128-byte segments per column)
constexpr int TF_TR = 128;  // rows of a tile of the fused pass
constexpr int TF_TC = 64;   // columns of a tile = entries of an index block
constexpr int TF_NT = 256;  // its threads: wavefront w owns 16 columns of the tile, a lane two rows

// sums CNT doubles over the 1024 threads; every thread may read s_red afterwards
template <int CNT> static __device__ __forceinline__ void td_block_sum(double (&v)[CNT], double *s_part, double *s_red)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
	for (int c = 0; c < CNT; ++c) {
		const double sv = wave_sum(v[c]);
		if (lane == 0)
			s_part[wave * CNT + c] = sv;
	}
	__syncthreads();
	if (tid < CNT) {
		double t = 0.0;
		for (int wv = 0; wv < TD_NT / 64; ++wv)
			t += s_part[wv * CNT + tid];
		s_red[tid] = t;
	}
	__syncthreads();
}

// Tiles of the lower triangle of an r x r matrix, TF_TR rows x TF_TC columns: tile (I, J) exists for J <= 2 I + 1 and J < ncb; row block I has
// td_row_tiles(I) of them, column block J lives in the row blocks J / 2 .. nbr - 1.
static __device__ __forceinline__ int td_row_tiles(int I, int ncb) { return min(2 * I + 2, ncb); }

// Index block b of TF_TC entries of the fused pass of step kk (A22 = A[kk+2.., kk+2..]): the shares of its row tiles and of its column's
// tiles, added in a fixed order -> ysum, by the 1024 threads of the calling workgroup; stored write-through (xwg.h) because the reader is
// another workgroup of the SAME launch: the blocks 1 .. of td_step_kernel(k) add the shares of pass k - 1 while block 0 loads its columns,
// then raise their flag; block 0 waits for the flags and reads the sums past its caches.  (A launch of its own for the sums -- round 6's
// first version -- cost 4.7 us per column: a launch and a memory round trip that nothing overlapped.)
constexpr int TS_NT = 1024, TS_NS = TS_NT / TF_TC; // 16 slices of the list of shares per entry
template <typename T> static __device__ __forceinline__ void td_sum_block(const TdArgs<T> &a, const int kk, const int b)
{
	__shared__ double s_q[TS_NS][TF_TC];
	const int tid = threadIdx.x;
	const int base = kk + 2, r = a.n - base;
	const int nbr = (r + TF_TR - 1) / TF_TR, ncb = (r + TF_TC - 1) / TF_TC;
	const int Ib = b >> 1;
	const int nrp = td_row_tiles(Ib, ncb), tot = nrp + (nbr - Ib);
	const int e = tid & 63, qq = tid >> 6, i = min(b * TF_TC + e, r - 1);
	const int per = (tot + TS_NS - 1) / TS_NS, p0 = qq * per;
	double sacc = 0.0;
	for (int pb = 0; pb < per; pb += 4) {
		double v[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int p = p0 + pb + u;
			const bool in = pb + u < per && p < tot;
			const int pc = in ? p : 0;
			const double *src = pc < nrp ? a.rpart + (size_t) pc * a.n : a.cpart + (size_t) (Ib + pc - nrp) * a.n;
			v[u] = src[base + i];
			if (!in)
				v[u] = 0.0;
		}
		sacc += (v[0] + v[1]) + (v[2] + v[3]);
	}
	__syncthreads(); // (s_q of a previous call has been read)
	s_q[qq][e] = sacc;
	__syncthreads();
	if (tid < TF_TC && b * TF_TC + tid < r) {
		double t = 0.0;
#pragma unroll
		for (int q = 0; q < TS_NS; ++q)
			t += s_q[q][tid];
		xwg_store(a.ysum + base + b * TF_TC + tid, t);
	}
}
// block 0 of td_step_kernel(k), k > 0: the sums of pass k - 1 are complete (flags of the helper blocks; if they do not come -- the
// launch shares the GPU and the helpers are not resident -- block 0 adds the shares itself)
template <typename T> static __device__ __forceinline__ void td_wait_sums(const TdArgs<T> &a, const int k)
{
	__shared__ int s_flag;
	const int nsb = (a.n - (k + 1) + TF_TC - 1) / TF_TC;
	if (nsb <= 0)
		return;
	if (!xwg_wait_all(a.flags, nsb, (xwg_u64) k, &s_flag)) {
		for (int b = 0; b < nsb; ++b)
			td_sum_block<T>(a, k - 1, b);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
	}
}

template <typename T> static __device__ __forceinline__ void td_step_body(const TdArgs<T> &a, const int k)
{
	__shared__ double s_part[(TD_NT / 64) * 3], s_red[3];
	const int tid = threadIdx.x, n = a.n;
	auto at = [&](int i, int j) -> T & { return a.A[(idx_t) i * a.rs + (idx_t) j * a.cs]; };
	T nacc[3] = {0, 0, 0}; // scaled sums of the tail of column k (reductions/norm_l2.rs:6-45)
	const T sml = (T) scale_sml<T>(), big = (T) scale_big<T>();
	if (k > 0) {
		// ---- y of step k - 1 (:484-511): x = the reflector in column k-1 (rows k+1..), ysum = sym(A22) x from the fused pass
		const T tau_inv = (T) a.st->tau_inv;
		td_wait_sums<T>(a, k);
		double d[2] = {0.0, 0.0};
		for (int i = k + 1 + tid; i < n; i += TD_NT) {
			const T aik = at(i, k), xi = at(i, k - 1);
			T yv = tau_inv * (T) xwg_load(a.ysum + i);
			yv += aik * tau_inv;
			a.y[i] = yv;
			d[0] += (double) aik * (double) xi;
			d[1] += (double) xi * (double) yv;
		}
		td_block_sum<2>(d, s_part, s_red);
		T y1 = (at(k, k) + (T) s_red[0]) * tau_inv;
		const T b = ((y1 + (T) s_red[1]) * (T) 0.5) * tau_inv;
		y1 -= b;
		__syncthreads(); // at(k, k) was read by everyone
		// ---- y -= b x, then column k receives the rest of the rank-2 update (:300-318), norm of its tail on the way
		for (int i = k + 1 + tid; i < n; i += TD_NT) {
			const T xi = at(i, k - 1);
			const T yi = a.y[i] - b * xi;
			a.y[i] = yi;
			const T v = at(i, k) - (y1 * xi + yi);
			at(i, k) = v;
			if (i >= k + 2) {
				nacc[0] += (v * sml) * (v * sml);
				nacc[1] += v * v;
				nacc[2] += (v * big) * (v * big);
			}
		}
		if (tid == 0) {
			a.y[k] = y1;
			at(k, k) -= y1 + y1;
		}
	} else {
		for (int i = 2 + tid; i < n; i += TD_NT) {
			const T v = at(i, 0);
			nacc[0] += (v * sml) * (v * sml);
			nacc[1] += v * v;
			nacc[2] += (v * big) * (v * big);
		}
	}
	if (k + 1 >= n)
		return;
	// ---- reflector of column k below the diagonal (:330-336)
	double accd[3] = {(double) nacc[0], (double) nacc[1], (double) nacc[2]};
	td_block_sum<3>(accd, s_part, s_red); // (its barriers also publish the column written above)
	const T tail_norm = norm_from3<T>(s_red);
	T head = at(k + 1, k);
	T head_norm = fabs(head);
	if (head_norm < Lim<T>::minpos) {
		head = (T) 0;
		head_norm = (T) 0;
	}
	T tau, hinv = (T) 0;
	bool scale_tail = false;
	if (tail_norm < Lim<T>::minpos) {
		tau = std::numeric_limits<T>::infinity();
	} else {
		const T norm = (T) hypot((double) head_norm, (double) tail_norm);
		const T sign = head_norm != (T) 0 ? head * ((T) 1 / head_norm) : (T) 1;
		const T signed_norm = sign * norm;
		hinv = (T) 1 / (head + signed_norm);
		head = -signed_norm;
		const T tn = tail_norm * fabs(hinv);
		tau = (T) 0.5 * ((T) 1 + tn * tn);
		scale_tail = true;
	}
	__syncthreads(); // everyone has read the old head
	const T u1 = k > 0 ? at(k + 1, k - 1) : (T) 0, y1n = k > 0 ? a.y[k + 1] : (T) 0;
	for (int i = k + 2 + tid; i < n; i += TD_NT) {
		if (scale_tail)
			at(i, k) *= hinv;
		if (k > 0) { // :348-359
			const T yi = a.y[i];
			at(i, k + 1) -= at(i, k - 1) * y1n + yi * u1;
			a.w[i] = yi;
		}
	}
	if (tid == 0) {
		at(k + 1, k) = head;
		a.taus[k] = tau;
		a.st->tau_inv = (double) ((T) 1 / tau);
		if (k > 0)
			at(k + 1, k + 1) -= u1 * y1n + y1n * u1;
	}
}

// The same step with every entry of the three columns it touches held in registers (at most TD_E rows per thread: n - k - 1 <= TD_E TD_NT):
// all loads are issued at the start -- ONE round trip to memory instead of one per loop -- and every entry is stored once.  Same arithmetic,
// expression by expression, as td_step_body.
constexpr int TD_E = 4;
template <typename T> static __device__ __forceinline__ void td_step_body_reg(const TdArgs<T> &a, const int k)
{
	__shared__ double s_part[(TD_NT / 64) * 3], s_red[3];
	__shared__ T s_bc[3];
	const int tid = threadIdx.x, n = a.n;
	auto at = [&](int i, int j) -> T & { return a.A[(idx_t) i * a.rs + (idx_t) j * a.cs]; };
	const T sml = (T) scale_sml<T>(), big = (T) scale_big<T>();
	const bool upd = k > 0, more = k + 1 < n;
	// rows i = k + 1 + tid + e TD_NT: column k (aik), the previous reflector (xi), column k + 1 (ak1), the product of the fused pass (ys)
	T aik[TD_E], xi[TD_E], ak1[TD_E], yi[TD_E];
	double ys[TD_E];
#pragma unroll
	for (int e = 0; e < TD_E; ++e) {
		const int i = k + 1 + tid + e * TD_NT;
		const bool in = i < n;
		const int ic = in ? i : n - 1;
		aik[e] = at(ic, k);
		xi[e] = upd ? at(ic, k - 1) : (T) 0;
		ys[e] = 0.0;
		ak1[e] = (upd && more) ? at(ic, k + 1) : (T) 0;
		yi[e] = (T) 0;
	}
	const T akk = at(k, k), tau_inv = (T) a.st->tau_inv;
	if (upd) { // (the loads above are in flight while the helper blocks finish the sums)
		td_wait_sums<T>(a, k);
#pragma unroll
		for (int e = 0; e < TD_E; ++e) {
			const int i = k + 1 + tid + e * TD_NT;
			ys[e] = xwg_load(a.ysum + (i < n ? i : n - 1));
		}
	}
	T y1 = (T) 0;
	T nacc[3] = {0, 0, 0}; // scaled sums of the tail of column k (reductions/norm_l2.rs:6-45)
	if (upd) {
		// ---- y of step k - 1 (:484-511)
		double d[2] = {0.0, 0.0};
#pragma unroll
		for (int e = 0; e < TD_E; ++e)
			if (k + 1 + tid + e * TD_NT < n) {
				T yv = tau_inv * (T) ys[e];
				yv += aik[e] * tau_inv;
				yi[e] = yv;
				d[0] += (double) aik[e] * (double) xi[e];
				d[1] += (double) xi[e] * (double) yv;
			}
		td_block_sum<2>(d, s_part, s_red);
		y1 = (akk + (T) s_red[0]) * tau_inv;
		const T b = ((y1 + (T) s_red[1]) * (T) 0.5) * tau_inv;
		y1 -= b;
		// ---- y -= b x, then column k receives the rest of the rank-2 update (:300-318), norm of its tail on the way
#pragma unroll
		for (int e = 0; e < TD_E; ++e) {
			const int i = k + 1 + tid + e * TD_NT;
			if (i < n) {
				yi[e] -= b * xi[e];
				aik[e] -= y1 * xi[e] + yi[e];
			}
		}
	}
#pragma unroll
	for (int e = 0; e < TD_E; ++e) {
		const int i = k + 1 + tid + e * TD_NT;
		if (i < n && i >= k + 2) {
			const T v = aik[e];
			nacc[0] += (v * sml) * (v * sml);
			nacc[1] += v * v;
			nacc[2] += (v * big) * (v * big);
		}
	}
	if (tid == 0 && upd)
		at(k, k) = akk - (y1 + y1);
	if (!more)
		return;
	if (tid == 0) { // row k + 1: the head of the column, y and u of the update of column k + 1
		s_bc[0] = aik[0];
		s_bc[1] = yi[0];
		s_bc[2] = xi[0];
	}
	// ---- reflector of column k below the diagonal (:330-336)
	double accd[3] = {(double) nacc[0], (double) nacc[1], (double) nacc[2]};
	td_block_sum<3>(accd, s_part, s_red);
	const T tail_norm = norm_from3<T>(s_red);
	T head = s_bc[0];
	T head_norm = fabs(head);
	if (head_norm < Lim<T>::minpos) {
		head = (T) 0;
		head_norm = (T) 0;
	}
	T tau, hinv = (T) 0;
	bool scale_tail = false;
	if (tail_norm < Lim<T>::minpos) {
		tau = std::numeric_limits<T>::infinity();
	} else {
		const T norm = (T) hypot((double) head_norm, (double) tail_norm);
		const T sign = head_norm != (T) 0 ? head * ((T) 1 / head_norm) : (T) 1;
		const T signed_norm = sign * norm;
		hinv = (T) 1 / (head + signed_norm);
		head = -signed_norm;
		const T tn = tail_norm * fabs(hinv);
		tau = (T) 0.5 * ((T) 1 + tn * tn);
		scale_tail = true;
	}
	const T u1 = upd ? s_bc[2] : (T) 0, y1n = upd ? s_bc[1] : (T) 0;
#pragma unroll
	for (int e = 0; e < TD_E; ++e) {
		const int i = k + 1 + tid + e * TD_NT;
		if (i < n && i >= k + 2) {
			if (scale_tail || upd)
				at(i, k) = scale_tail ? aik[e] * hinv : aik[e];
			if (upd) { // :348-359
				at(i, k + 1) = ak1[e] - (xi[e] * y1n + yi[e] * u1);
				a.w[i] = yi[e];
			}
		}
	}
	if (tid == 0) {
		at(k + 1, k) = head;
		a.taus[k] = tau;
		a.st->tau_inv = (double) ((T) 1 / tau);
		if (upd)
			at(k + 1, k + 1) = ak1[0] - (u1 * y1n + y1n * u1);
	}
}

template <typename T> __global__ __launch_bounds__(TD_NT) void td_step_kernel(const TdArgs<T> a)
{
	if (blockIdx.x > 0) { // helper block: the sums of index block blockIdx.x - 1 of pass k - 1
		td_sum_block<T>(a, a.k - 1, (int) blockIdx.x - 1);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (threadIdx.x == 0)
			__hip_atomic_store(a.flags + (blockIdx.x - 1), (xwg_u64) a.k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		return;
	}
	if (a.n - a.k - 1 <= TD_E * TD_NT && !a.force_mem)
		td_step_body_reg<T>(a, a.k);
	else
		td_step_body<T>(a, a.k);
}

// value of lane `l` (compile-time after unrolling) for the whole wavefront
static __device__ __forceinline__ double td_lane(double v, int l)
{
	const long long b = __double_as_longlong(v);
	const int lo = __builtin_amdgcn_readlane((int) b, l), hi = __builtin_amdgcn_readlane((int) (b >> 32), l);
	return __longlong_as_double(((long long) hi << 32) | (unsigned) lo);
}
static __device__ __forceinline__ float td_lane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// sum over the wavefront, uniform: butterflies inside the rows of 16 lanes on the DPP path (xor 1, xor 2, half mirror, mirror), then
// the four row sums in a fixed order (the __shfl_xor form goes through the LDS crossbar: 12 ds_bpermute per sum)
template <int CTRL> static __device__ __forceinline__ double td_dpp(double v)
{
	const long long b = __double_as_longlong(v);
	const int lo = __builtin_amdgcn_update_dpp(0, (int) b, CTRL, 0xf, 0xf, true), hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), CTRL, 0xf, 0xf, true);
	return __longlong_as_double(((long long) hi << 32) | (unsigned) lo);
}
static __device__ __forceinline__ double td_wave_sum(double v)
{
	v += td_dpp<0xB1>(v);  // quad_perm [1, 0, 3, 2]
	v += td_dpp<0x4E>(v);  // quad_perm [2, 3, 0, 1]
	v += td_dpp<0x141>(v); // row_half_mirror
	v += td_dpp<0x140>(v); // row_mirror
	return ((td_lane(v, 0) + td_lane(v, 16)) + td_lane(v, 32)) + td_lane(v, 48);
}


// Tile (I, J) of A22 (header of this section).  Lane l of wavefront w: rows 128 I + l and + 64, columns 64 J + 16 w .. + 15; all 32 loads
// of a thread are issued before the first use.
template <typename T, bool upd> __global__ __launch_bounds__(TF_NT) void td_fused_kernel(const TdArgs<T> a)
{
	constexpr int CW = TF_TC / (TF_NT / 64); // 16 columns per wavefront
	__shared__ double s_row[TF_NT / 64][TF_TR];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, k = a.k;
	const int base = k + 2, r = a.n - base; // A22 = A[base.., base..], r x r
	const int nbr = (r + TF_TR - 1) / TF_TR, ncb = (r + TF_TC - 1) / TF_TC;
	const int t = blockIdx.x; // tile (I, J): t in [I (I + 1), (I + 1)(I + 2))
	int I = (int) ((sqrtf(4.0f * (float) t + 1.0f) - 1.0f) * 0.5f);
	while (I * (I + 1) > t)
		--I;
	while ((I + 1) * (I + 2) <= t)
		++I;
	const int J = t - I * (I + 1);
	if (J >= ncb)
		return;
	const T *u = a.A + (idx_t) base * a.rs + (idx_t) (upd ? k - 1 : 0) * a.cs; // u[i * rs]
	const T *x = a.A + (idx_t) base * a.rs + (idx_t) k * a.cs;
	const T *w = a.w + base;
	T *A22 = a.A + (idx_t) base * a.rs + (idx_t) base * a.cs;
	const int i0 = I * TF_TR, j0 = J * TF_TC + CW * wv;
	int gi[2];
	bool vr[2];
	T xi[2], ui[2], wi[2];
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		gi[h] = i0 + lane + 64 * h;
		vr[h] = gi[h] < r;
		gi[h] = min(gi[h], r - 1); // (loads below are unconditional: a row past the end reads the last row and takes part in nothing)
		const idx_t o = (idx_t) gi[h] * a.rs;
		xi[h] = x[o];
		ui[h] = upd ? u[o] : (T) 0;
		wi[h] = upd ? w[gi[h]] : (T) 0;
	}
	T xjl, ujl, wjl; // column values of the wavefront's 16 columns, one per lane
	{
		const int gj = min(j0 + (lane & (CW - 1)), r - 1);
		const idx_t o = (idx_t) gj * a.rs;
		xjl = x[o];
		ujl = upd ? u[o] : (T) 0;
		wjl = upd ? w[gj] : (T) 0;
	}
	T v[2][CW];
#pragma unroll
	for (int c = 0; c < CW; ++c)
#pragma unroll
		for (int h = 0; h < 2; ++h) // an entry above the diagonal reads the diagonal entry of its row instead (and is not used)
			v[h][c] = A22[(idx_t) gi[h] * a.rs + (idx_t) min(j0 + c, gi[h]) * a.cs];
	double racc[2] = {0.0, 0.0};
#pragma unroll
	for (int c = 0; c < CW; ++c) {
		const int gj = j0 + c;
		const T xj = td_lane(xjl, c), uj = td_lane(ujl, c), wj = td_lane(wjl, c);
		double cs_ = 0.0;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const bool in = vr[h] && gj <= gi[h];
			T tv = v[h][c];
			if (upd) {
				tv = fh_fma(-ui[h], wj, tv);
				tv = fh_fma(-wi[h], uj, tv);
				if (in)
					A22[(idx_t) gi[h] * a.rs + (idx_t) gj * a.cs] = tv;
			}
			racc[h] += in ? (double) tv * (double) xj : 0.0;
			cs_ += (in && gj < gi[h]) ? (double) tv * (double) xi[h] : 0.0; // the diagonal entry is not part of the strictly-upper product
		}
		const double sv = td_wave_sum(cs_);
		if (lane == 0 && gj < r)
			a.cpart[(size_t) I * a.n + base + gj] = sv;
	}
	s_row[wv][lane] = racc[0];
	s_row[wv][lane + 64] = racc[1];
	__syncthreads();
	if (tid < TF_TR && i0 + tid < r)
		a.rpart[(size_t) J * a.n + base + i0 + tid] = ((s_row[0][tid] + s_row[1][tid]) + s_row[2][tid]) + s_row[3][tid];
}

// A: n x n (lower triangle used), H: block_size x (n - 1)
template <typename T> void tridiag_dev(MatV<T> A, MatV<T> H)
{
	const idx_t n = A.nrows;
	FH_CHECK(A.ncols == n, "tridiag: the matrix must be square");
	FH_CHECK(H.nrows > 0 && H.ncols == (n > 0 ? n - 1 : 0), "tridiag: householder must be block_size x (n - 1)");
	FH_CHECK(n < (1L << 30), "tridiag: matrix too large");
	if (n <= 1)
		return;
	hipStream_t s = ctx().stream;
	const idx_t nbr = (n + TF_TR - 1) / TF_TR, ncb = (n + TF_TC - 1) / TF_TC;
	Scratch vb((size_t) (4 * n) * sizeof(T) + 256), cb((size_t) (1 + ncb + nbr) * (size_t) n * sizeof(double)), stb(sizeof(TdState)),
		cntb((size_t) (ncb + 1) * sizeof(xwg_u64));
	TdArgs<T> a;
	a.force_mem = g_l2_force_mem.load();
	a.A = A.p;
	a.rs = A.rs;
	a.cs = A.cs;
	a.n = (int) n;
	a.y = vb.as<T>();
	a.w = a.y + n;
	a.taus = a.w + 2 * n;
	a.ysum = cb.as<double>();
	a.rpart = a.ysum + n;
	a.cpart = a.rpart + (size_t) ncb * (size_t) n;
	a.flags = cntb.as<xwg_u64>();
	a.st = stb.as<TdState>();
	FH_HIP(hipMemsetAsync(vb.p, 0, (size_t) (4 * n) * sizeof(T), s));
	FH_HIP(hipMemsetAsync(stb.p, 0, sizeof(TdState), s));
	FH_HIP(hipMemsetAsync(cntb.p, 0, (size_t) (ncb + 1) * sizeof(xwg_u64), s));
	for (idx_t k = 0; k < n; ++k) {
		a.k = (int) k;
		// block 0: the step; blocks 1 ..: the sums of the previous pass (k > 0), one per index block of TF_TC entries
		const unsigned nsb = k > 0 ? (unsigned) ((n - k - 1 + TF_TC - 1) / TF_TC) : 0u;
		hipLaunchKernelGGL(td_step_kernel<T>, dim3(1 + nsb), dim3(TD_NT), 0, s, a);
		const idx_t r = n - k - 2;
		if (r > 0) {
			const idx_t rb = (r + TF_TR - 1) / TF_TR;
			if (k > 0)
				hipLaunchKernelGGL((td_fused_kernel<T, true>), dim3((unsigned) (rb * (rb + 1))), dim3(TF_NT), 0, s, a);
			else
				hipLaunchKernelGGL((td_fused_kernel<T, false>), dim3((unsigned) (rb * (rb + 1))), dim3(TF_NT), 0, s, a);
		}
	}
	FH_HIP(hipGetLastError());
	// block Householder factors of A.submatrix(1, 0, n - 1, n - 1) (:516-533)
	qr_t_blocks_from_taus<T>(A.sub(1, 0, n - 1, n - 1), H, n - 1, a.taus);
	FH_HIP(hipStreamSynchronize(s)); // the scratch vectors above are released on return
}
template void tridiag_dev<double>(MatV<double>, MatV<double>);
template void tridiag_dev<float>(MatV<float>, MatV<float>);

// ------------------------------------------------------------------------------------------------
// Bidiagonalization -- faer/src/linalg/svd/bidiag.rs:47-255.
// The reference's unblocked level-2 algorithm: per column k (i) column k and row k receive the rest of the previous
// step's rank-2 update (:80-98), (ii) the left reflector of column k (:99-102), (iii) ONE pass over A22 that applies
// A22 -= up y2 + z2 vp and forms y2 = u^H A22 (bidiag_fused_op, :257-301), (iv) y2, row k and its norm (:156-164),
// (v) z2 = A22 A12^H (:165-172), (vi) the right reflector of the normalised row and the correction of z2 (:176-213).
// Four launches per column, no host synchronisation in the loop:
//   bd_pre_kernel(k)   block 0: (vi)'s correction of z for step k-1, then (i) and (ii); blocks 1 ..: the sums of the row pass of step k-1
//   bd_col_kernel(k)   256 x 32 tiles of A22, lanes along the rows, 32 loads in flight per thread: (iii) written back and the tile's
//                      share of y2 = u^H A22
//   bd_mid_kernel(k)   block 0: (iv), a copy of the normalised row for (v), then the reflector part of (vi); blocks 1 ..: the sums of
//                      the column pass
//   bd_row_kernel(k)   128 x 128 tiles of A22 (read only): the tile's share of (v)
// The shares of the tiles are added in a fixed order (bd_sum_block) and handed to block 0 inside the launch (xwg.h).  Rounds 2-5: one
// wavefront per column / one workgroup per 16 rows with complete sums (302 ms at N = 4096); tiles with sums as launches of their own:
// 272; pre / mid with their rows and columns in registers: 230; the sums as helper blocks: 210.
// Algorithmic bytes: A22 read + written once and read once more per column, sum_k 3 (m-k-1)(n-k-1) sizeof(T).
// ------------------------------------------------------------------------------------------------
struct BdState {
	double tl_inv;			  // left reflector of the current step
	double tr_inv, beta, hinv, b, norm; // right reflector of the current step (consumed by the next bd_pre_kernel)
	int hinv_inf, pad;
};
template <typename T> struct BdArgs {
	T *A;
	idx_t rs, cs;
	int m, n, size, k, force_mem;
	T *y, *z, *vrow, *taul, *taur;
	double *ysum, *zsum;   // u^H A22 and A22 v of the two passes, complete (fixed-order sums of the shares below)
	double *ypart, *zpart; // shares of the tiles: ypart[row block * n + j], zpart[column block * m + i]
	xwg_u64 *yflag, *zflag; // per helper block of bd_mid_kernel / bd_pre_kernel: the launch whose sums are complete
	int nh;		       // helper blocks of this launch (0: the sums were prepared otherwise)
	BdState *st;
};

// make_householder_imp (householder.rs:59-107) from the head and the scaled sums of the tail; returns tau, sets
// head <- beta, hinv (0 if the tail is negligible: nothing is scaled), inf_flag
template <typename T> static __device__ __forceinline__ T bd_householder(T &head, T tail_norm, T &hinv, bool &negligible)
{
	T head_norm = fabs(head);
	if (head_norm < Lim<T>::minpos) {
		head = (T) 0;
		head_norm = (T) 0;
	}
	negligible = tail_norm < Lim<T>::minpos;
	hinv = (T) 0;
	if (negligible)
		return std::numeric_limits<T>::infinity();
	const T norm = (T) hypot((double) head_norm, (double) tail_norm);
	const T sign = head_norm != (T) 0 ? head * ((T) 1 / head_norm) : (T) 1;
	const T signed_norm = sign * norm;
	hinv = (T) 1 / (head + signed_norm);
	head = -signed_norm;
	const T tn = tail_norm * fabs(hinv);
	return (T) 0.5 * ((T) 1 + tn * tn);
}

constexpr int BC_TR = 256, BC_TC = 32, BC_NT = 256; // tiles of bd_col_kernel
constexpr int BR_TR = 128, BR_TC = 128, BR_NT = 256; // tiles of bd_row_kernel
// Helper block h of a single-workgroup launch: out[off + e] = the sum of `np` shares part[p * stride + off + e], e in [1024 h, 1024 h + 1024)
// and < len, added in the order of p; stored write-through and flagged, because the reader is block 0 of the SAME launch (xwg.h; the
// tridiagonalization's td_sum_block has the reasoning).  All 1024 threads of the calling workgroup.
static __device__ __forceinline__ void bd_sum_block(const double *part, int np, size_t stride, int off, int len, double *out, int h)
{
	const int e = h * TD_NT + (int) threadIdx.x;
	if (e >= len)
		return;
	const double *src = part + off + e;
	double s0 = 0.0;
	int p = 0;
	for (; p + 8 <= np; p += 8) {
		double v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u)
			v[u] = src[(size_t) (p + u) * stride];
#pragma unroll
		for (int u = 0; u < 8; ++u)
			s0 += v[u];
	}
	for (; p < np; ++p)
		s0 += src[(size_t) p * stride];
	xwg_store(out + off + e, s0);
}
// which = 0: z sums of row pass k - 1 for bd_pre_kernel(k) (rows k .., column blocks of BR_TC); 1: y sums of column pass k for
// bd_mid_kernel(k) (columns k + 1 .., row blocks of BC_TR)
template <typename T> static __device__ __forceinline__ void bd_helper(const BdArgs<T> &a, int which, int h)
{
	if (which == 0)
		bd_sum_block(a.zpart, (a.n - a.k + BR_TC - 1) / BR_TC, (size_t) a.m, a.k, a.m - a.k, a.zsum, h);
	else
		bd_sum_block(a.ypart, (a.m - a.k - 1 + BC_TR - 1) / BC_TR, (size_t) a.n, a.k + 1, a.n - a.k - 1, a.ysum, h);
}
template <typename T> static __device__ __forceinline__ void bd_helper_block(const BdArgs<T> &a, int which)
{
	bd_helper<T>(a, which, (int) blockIdx.x - 1);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0)
		__hip_atomic_store((which == 0 ? a.zflag : a.yflag) + (blockIdx.x - 1), (xwg_u64) (a.k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// block 0: the sums are complete (or it adds the shares itself if the helpers' flags do not come)
template <typename T> static __device__ __forceinline__ void bd_wait_sums(const BdArgs<T> &a, int which)
{
	__shared__ int s_flag;
	if (a.nh <= 0)
		return;
	if (!xwg_wait_all(which == 0 ? a.zflag : a.yflag, a.nh, (xwg_u64) (a.k + 1), &s_flag)) {
		for (int h = 0; h < a.nh; ++h)
			bd_helper<T>(a, which, h);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
	}
}

template <typename T> static __device__ __forceinline__ void bd_pre_body(const BdArgs<T> &a)
{
	__shared__ double s_part[(TD_NT / 64) * 3], s_red[3];
	const int tid = threadIdx.x, k = a.k, m = a.m, n = a.n;
	auto at = [&](int i, int j) -> T & { return a.A[(idx_t) i * a.rs + (idx_t) j * a.cs]; };
	const T sml = (T) scale_sml<T>(), big = (T) scale_big<T>();
	T nacc[3] = {0, 0, 0};
	if (k > 0) {
		// ---- z of step k-1 (:186-213): zsum = A22 A12^H of that step, u = its left reflector (column k-1), A22_a = column k
		const T beta = (T) a.st->beta, hinv = (T) a.st->hinv, b = (T) a.st->b, tr_inv = (T) a.st->tr_inv;
		const bool inf = a.st->hinv_inf != 0;
		auto fix = [&](T zs, T a22a, T u) -> T {
			T w;
			if (!inf) {
				w = zs - a22a * beta;
				w = w * hinv;
				w = w - u * b;
			} else {
				w = a22a - u * b;
			}
			return w * tr_inv;
		};
		bd_wait_sums<T>(a, 0);
		const T up0 = at(k, k - 1), y1 = a.y[k];
		const T z1 = fix((T) xwg_load(a.zsum + k), at(k, k), up0);
		__syncthreads(); // everyone has read a_kk
		// ---- (i): the rest of the previous rank-2 update on column k, row k and a_kk (:80-98)
		if (tid == 0) {
			a.z[k] = z1;
			at(k, k) -= up0 * y1 + z1;
		}
		for (int i = k + 1 + tid; i < m; i += TD_NT) {
			const T u = at(i, k - 1), old = at(i, k);
			const T zf = fix((T) xwg_load(a.zsum + i), old, u);
			a.z[i] = zf;
			const T v = old - (u * y1 + zf);
			at(i, k) = v;
			nacc[0] += (v * sml) * (v * sml);
			nacc[1] += v * v;
			nacc[2] += (v * big) * (v * big);
		}
		for (int j = k + 1 + tid; j < n; j += TD_NT)
			at(k, j) -= up0 * a.y[j] + z1 * at(k - 1, j);
	} else {
		for (int i = 1 + tid; i < m; i += TD_NT) {
			const T v = at(i, 0);
			nacc[0] += (v * sml) * (v * sml);
			nacc[1] += v * v;
			nacc[2] += (v * big) * (v * big);
		}
	}
	// ---- (ii) left reflector of column k (:99-102)
	double accd[3] = {(double) nacc[0], (double) nacc[1], (double) nacc[2]};
	td_block_sum<3>(accd, s_part, s_red);
	const T tail_norm = norm_from3<T>(s_red);
	T head = at(k, k), hinv;
	bool negligible;
	const T tau = bd_householder<T>(head, tail_norm, hinv, negligible);
	__syncthreads(); // everyone has read the old head
	if (!negligible)
		for (int i = k + 1 + tid; i < m; i += TD_NT)
			at(i, k) *= hinv;
	if (tid == 0) {
		at(k, k) = head;
		a.taul[k] = tau;
		a.st->tl_inv = (double) ((T) 1 / tau);
	}
}

template <typename T> static __device__ __forceinline__ void bd_mid_body(const BdArgs<T> &a);

// bd_pre_body with every entry it touches in registers (at most TD_E per thread and direction): all loads -- the strided ones of rows k - 1
// and k among them -- are issued at the start, every entry is stored once.  Same arithmetic, expression by expression.
template <typename T> static __device__ __forceinline__ void bd_pre_body_reg(const BdArgs<T> &a)
{
	__shared__ double s_part[(TD_NT / 64) * 3], s_red[3];
	const int tid = threadIdx.x, k = a.k, m = a.m, n = a.n;
	auto at = [&](int i, int j) -> T & { return a.A[(idx_t) i * a.rs + (idx_t) j * a.cs]; };
	const T sml = (T) scale_sml<T>(), big = (T) scale_big<T>();
	const bool upd = k > 0;
	T cu[TD_E], cold[TD_E], czs[TD_E], rk[TD_E], rkm[TD_E], ry[TD_E];
#pragma unroll
	for (int e = 0; e < TD_E; ++e) {
		const int i = k + 1 + tid + e * TD_NT, ic = i < m ? i : m - 1;
		cold[e] = at(ic, k);
		cu[e] = upd ? at(ic, k - 1) : (T) 0;
		czs[e] = (T) 0;
		const int j = k + 1 + tid + e * TD_NT, jc = j < n ? j : n - 1;
		rk[e] = upd ? at(k, jc) : (T) 0;
		rkm[e] = upd ? at(k - 1, jc) : (T) 0;
		ry[e] = upd ? a.y[jc] : (T) 0;
	}
	T akk = at(k, k);
	T nacc[3] = {0, 0, 0};
	if (upd) {
		const T beta = (T) a.st->beta, hinv = (T) a.st->hinv, b = (T) a.st->b, tr_inv = (T) a.st->tr_inv;
		const bool inf = a.st->hinv_inf != 0;
		auto fix = [&](T zs, T a22a, T u) -> T {
			T w;
			if (!inf) {
				w = zs - a22a * beta;
				w = w * hinv;
				w = w - u * b;
			} else {
				w = a22a - u * b;
			}
			return w * tr_inv;
		};
		const T up0 = at(k, k - 1), y1 = a.y[k];
		bd_wait_sums<T>(a, 0); // (the loads above are in flight while the helper blocks finish the sums)
#pragma unroll
		for (int e = 0; e < TD_E; ++e) {
			const int i = k + 1 + tid + e * TD_NT;
			czs[e] = (T) xwg_load(a.zsum + (i < m ? i : m - 1));
		}
		const T z1 = fix((T) xwg_load(a.zsum + k), akk, up0);
		akk -= up0 * y1 + z1;
		if (tid == 0)
			a.z[k] = z1; // (a_kk itself is stored once, below, as the reflector's beta: nobody may see an intermediate value)
#pragma unroll
		for (int e = 0; e < TD_E; ++e) {
			const int i = k + 1 + tid + e * TD_NT;
			if (i < m) {
				const T zf = fix(czs[e], cold[e], cu[e]);
				a.z[i] = zf;
				cold[e] = cold[e] - (cu[e] * y1 + zf);
			}
			const int j = k + 1 + tid + e * TD_NT;
			if (j < n)
				at(k, j) = rk[e] - (up0 * ry[e] + z1 * rkm[e]);
		}
	}
#pragma unroll
	for (int e = 0; e < TD_E; ++e)
		if (k + 1 + tid + e * TD_NT < m) {
			const T v = cold[e];
			nacc[0] += (v * sml) * (v * sml);
			nacc[1] += v * v;
			nacc[2] += (v * big) * (v * big);
		}
	// ---- (ii) left reflector of column k (:99-102)
	double accd[3] = {(double) nacc[0], (double) nacc[1], (double) nacc[2]};
	td_block_sum<3>(accd, s_part, s_red);
	const T tail_norm = norm_from3<T>(s_red);
	T head = akk, hinv;
	bool negligible;
	const T tau = bd_householder<T>(head, tail_norm, hinv, negligible);
#pragma unroll
	for (int e = 0; e < TD_E; ++e) {
		const int i = k + 1 + tid + e * TD_NT;
		if (i < m && (upd || !negligible))
			at(i, k) = negligible ? cold[e] : cold[e] * hinv;
	}
	if (tid == 0) {
		at(k, k) = head;
		a.taul[k] = tau;
		a.st->tl_inv = (double) ((T) 1 / tau);
	}
}

template <typename T> __global__ __launch_bounds__(TD_NT) void bd_pre_kernel(const BdArgs<T> a)
{
	if (blockIdx.x > 0) {
		bd_helper_block<T>(a, 0);
		return;
	}
	if (a.m - a.k - 1 <= TD_E * TD_NT && a.n - a.k - 1 <= TD_E * TD_NT && !a.force_mem)
		bd_pre_body_reg<T>(a);
	else
		bd_pre_body<T>(a);
}

// bd_mid_body with row k in registers: ONE strided read and one strided write of the row instead of three each.
template <typename T> static __device__ __forceinline__ void bd_mid_body_reg(const BdArgs<T> &a)
{
	__shared__ double s_part[(TD_NT / 64) * 3], s_red[3];
	__shared__ T s_bc[1];
	const int tid = threadIdx.x, k = a.k, n = a.n;
	auto row = [&](int j) -> T & { return a.A[(idx_t) k * a.rs + (idx_t) j * a.cs]; };
	const T sml = (T) scale_sml<T>(), big = (T) scale_big<T>();
	const T tl_inv = (T) a.st->tl_inv;
	T v[TD_E], yv[TD_E];
#pragma unroll
	for (int e = 0; e < TD_E; ++e) {
		const int j = k + 1 + tid + e * TD_NT, jc = j < n ? j : n - 1;
		v[e] = row(jc);
	}
	bd_wait_sums<T>(a, 1); // (the strided loads of the row are in flight meanwhile)
#pragma unroll
	for (int e = 0; e < TD_E; ++e) {
		const int j = k + 1 + tid + e * TD_NT;
		yv[e] = (T) xwg_load(a.ysum + (j < n ? j : n - 1));
	}
	// ---- (iv) y2 = (y2 + A12) / tau_l, A12 -= y2, norm of A12 (:156-164)
	T nacc[3] = {0, 0, 0};
#pragma unroll
	for (int e = 0; e < TD_E; ++e) {
		const int j = k + 1 + tid + e * TD_NT;
		if (j < n) {
			yv[e] = (yv[e] + v[e]) * tl_inv;
			a.y[j] = yv[e];
			v[e] = v[e] - yv[e];
			nacc[0] += (v[e] * sml) * (v[e] * sml);
			nacc[1] += v[e] * v[e];
			nacc[2] += (v[e] * big) * (v[e] * big);
		}
	}
	double accd[3] = {(double) nacc[0], (double) nacc[1], (double) nacc[2]};
	td_block_sum<3>(accd, s_part, s_red);
	const T norm = norm_from3<T>(s_red);
	const T norm_inv = (T) 1 / norm;
	T tacc[3] = {0, 0, 0};
#pragma unroll
	for (int e = 0; e < TD_E; ++e) {
		const int j = k + 1 + tid + e * TD_NT;
		if (j < n) {
			if (norm != (T) 0)
				v[e] *= norm_inv;
			a.vrow[j] = v[e]; // (v) multiplies by the normalised row as it is BEFORE the right reflector touches it
			if (j >= k + 2) {
				tacc[0] += (v[e] * sml) * (v[e] * sml);
				tacc[1] += v[e] * v[e];
				tacc[2] += (v[e] * big) * (v[e] * big);
			}
		}
	}
	if (k + 1 >= a.size) {
#pragma unroll
		for (int e = 0; e < TD_E; ++e)
			if (k + 1 + tid + e * TD_NT < n)
				row(k + 1 + tid + e * TD_NT) = v[e];
		return;
	}
	if (tid == 0)
		s_bc[0] = v[0]; // the head of the row (j = k + 1)
	// ---- (vi) right reflector of the normalised row (:176-185) and b (:186-193)
	double tad[3] = {(double) tacc[0], (double) tacc[1], (double) tacc[2]};
	td_block_sum<3>(tad, s_part, s_red);
	const T tail_norm = norm_from3<T>(s_red);
	T head = s_bc[0], hinv;
	bool negligible;
	const T tau = bd_householder<T>(head, tail_norm, hinv, negligible);
	double d[1] = {0.0};
#pragma unroll
	for (int e = 0; e < TD_E; ++e) {
		const int j = k + 1 + tid + e * TD_NT;
		if (j < n && j >= k + 2) {
			if (!negligible)
				v[e] *= hinv;
			row(j) = v[e];
			d[0] += (double) yv[e] * (double) v[e];
		}
	}
	td_block_sum<1>(d, s_part, s_red);
	if (tid == 0) {
		const T b = yv[0] + (T) s_red[0];
		row(k + 1) = head * norm; // beta, rescaled (:183-184)
		a.taur[k] = tau;
		a.st->tr_inv = (double) ((T) 1 / tau);
		a.st->beta = (double) head;
		a.st->hinv = (double) hinv;
		a.st->hinv_inf = negligible ? 1 : 0;
		a.st->b = (double) b;
		a.st->norm = (double) norm;
	}
}

template <typename T> __global__ __launch_bounds__(TD_NT) void bd_mid_kernel(const BdArgs<T> a)
{
	if (blockIdx.x > 0) {
		bd_helper_block<T>(a, 1);
		return;
	}
	if (a.n - a.k - 1 <= TD_E * TD_NT && !a.force_mem)
		bd_mid_body_reg<T>(a);
	else
		bd_mid_body<T>(a);
}

// (iii): tile of BC_TR rows x BC_TC columns of A22 = A[k+1.., k+1..]; wavefront w owns 8 columns, a lane four rows of each (32 loads in
// flight per thread): A22 -= up y2 + z2 vp written back, and the tile's share of y2 = u^H A22 -> ypart[row block][column].
template <typename T, bool upd> __global__ __launch_bounds__(BC_NT) void bd_col_kernel(const BdArgs<T> a)
{
	constexpr int CW = BC_TC / (BC_NT / 64), RH = BC_TR / 64; // 8 columns per wavefront, 4 rows per lane
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, k = a.k;
	const int base = k + 1, rr = a.m - base, cc = a.n - base;
	const int ncb = (cc + BC_TC - 1) / BC_TC;
	const int I = blockIdx.x / ncb, J = blockIdx.x - I * ncb;
	const int i0 = I * BC_TR, j0 = J * BC_TC + CW * wv;
	T *A22 = a.A + (idx_t) base * a.rs + (idx_t) base * a.cs;
	const T *ucol = a.A + (idx_t) base * a.rs + (idx_t) k * a.cs;		    // u_i (the left reflector, rows base..)
	const T *upcol = a.A + (idx_t) base * a.rs + (idx_t) (upd ? k - 1 : 0) * a.cs; // up_i
	const T *vprow = a.A + (idx_t) (upd ? k - 1 : 0) * a.rs + (idx_t) base * a.cs; // vp_j (row k - 1)
	int gi[RH];
	bool vr[RH];
	T ui[RH], upi[RH], zi[RH];
#pragma unroll
	for (int h = 0; h < RH; ++h) {
		gi[h] = i0 + lane + 64 * h;
		vr[h] = gi[h] < rr;
		gi[h] = min(gi[h], rr - 1);
		const idx_t o = (idx_t) gi[h] * a.rs;
		ui[h] = ucol[o];
		upi[h] = upd ? upcol[o] : (T) 0;
		zi[h] = upd ? a.z[base + gi[h]] : (T) 0;
	}
	T yjl = (T) 0, vpjl = (T) 0; // column values of the wavefront's columns, one per lane
	if (upd) {
		const int gj = min(j0 + (lane & (CW - 1)), cc - 1);
		yjl = a.y[base + gj];
		vpjl = vprow[(idx_t) gj * a.cs];
	}
	T v[RH][CW];
#pragma unroll
	for (int c = 0; c < CW; ++c)
#pragma unroll
		for (int h = 0; h < RH; ++h)
			v[h][c] = A22[(idx_t) gi[h] * a.rs + (idx_t) min(j0 + c, cc - 1) * a.cs];
#pragma unroll
	for (int c = 0; c < CW; ++c) {
		const int gj = j0 + c;
		const T yj = td_lane(yjl, c), vpj = td_lane(vpjl, c);
		double cs_ = 0.0;
#pragma unroll
		for (int h = 0; h < RH; ++h) {
			const bool in = vr[h] && gj < cc;
			T tv = v[h][c];
			if (upd) {
				tv = fh_fma(-upi[h], yj, tv); // A22 -= up y2 (:292)
				tv = fh_fma(-zi[h], vpj, tv); // A22 -= z2 vp (:293)
				if (in)
					A22[(idx_t) gi[h] * a.rs + (idx_t) gj * a.cs] = tv;
			}
			cs_ += in ? (double) ui[h] * (double) tv : 0.0; // y2 = u^H A22 (:294-300)
		}
		const double sv = td_wave_sum(cs_);
		if (lane == 0 && gj < cc)
			a.ypart[(size_t) I * a.n + base + gj] = sv;
	}
}

template <typename T> static __device__ __forceinline__ void bd_mid_body(const BdArgs<T> &a)
{
	__shared__ double s_part[(TD_NT / 64) * 3], s_red[3];
	const int tid = threadIdx.x, k = a.k, n = a.n;
	auto row = [&](int j) -> T & { return a.A[(idx_t) k * a.rs + (idx_t) j * a.cs]; };
	const T sml = (T) scale_sml<T>(), big = (T) scale_big<T>();
	const T tl_inv = (T) a.st->tl_inv;
	// ---- (iv) y2 = (y2 + A12) / tau_l, A12 -= y2, norm of A12 (:156-164)
	T nacc[3] = {0, 0, 0};
	bd_wait_sums<T>(a, 1);
	for (int j = k + 1 + tid; j < n; j += TD_NT) {
		const T yv = ((T) xwg_load(a.ysum + j) + row(j)) * tl_inv;
		a.y[j] = yv;
		const T v = row(j) - yv;
		row(j) = v;
		nacc[0] += (v * sml) * (v * sml);
		nacc[1] += v * v;
		nacc[2] += (v * big) * (v * big);
	}
	double accd[3] = {(double) nacc[0], (double) nacc[1], (double) nacc[2]};
	td_block_sum<3>(accd, s_part, s_red);
	const T norm = norm_from3<T>(s_red);
	const T norm_inv = (T) 1 / norm;
	T tacc[3] = {0, 0, 0};
	for (int j = k + 1 + tid; j < n; j += TD_NT) {
		T v = row(j);
		if (norm != (T) 0) {
			v *= norm_inv;
			row(j) = v;
		}
		a.vrow[j] = v; // (v) multiplies by the normalised row as it is BEFORE the right reflector touches it
		if (j >= k + 2) {
			tacc[0] += (v * sml) * (v * sml);
			tacc[1] += v * v;
			tacc[2] += (v * big) * (v * big);
		}
	}
	if (k + 1 >= a.size)
		return;
	// ---- (vi) right reflector of the normalised row (:176-185) and b (:186-193)
	double tad[3] = {(double) tacc[0], (double) tacc[1], (double) tacc[2]};
	td_block_sum<3>(tad, s_part, s_red); // (its barriers publish the row written above)
	const T tail_norm = norm_from3<T>(s_red);
	T head = row(k + 1), hinv;
	bool negligible;
	const T tau = bd_householder<T>(head, tail_norm, hinv, negligible);
	__syncthreads(); // everyone has read the old head
	double d[1] = {0.0};
	for (int j = k + 2 + tid; j < n; j += TD_NT) {
		T v = row(j);
		if (!negligible) {
			v *= hinv;
			row(j) = v;
		}
		d[0] += (double) a.y[j] * (double) v;
	}
	td_block_sum<1>(d, s_part, s_red);
	if (tid == 0) {
		const T b = a.y[k + 1] + (T) s_red[0];
		row(k + 1) = head * norm; // beta, rescaled (:183-184)
		a.taur[k] = tau;
		a.st->tr_inv = (double) ((T) 1 / tau);
		a.st->beta = (double) head;
		a.st->hinv = (double) hinv;
		a.st->hinv_inf = negligible ? 1 : 0;
		a.st->b = (double) b;
		a.st->norm = (double) norm;
	}
}

// (v): z2 = A22 A12^H with the normalised row (vrow), read only: tile of BR_TR rows x BR_TC columns; wavefront w owns 32 columns, a lane two
// rows of each (64 loads in two batches); the tile's share of the row sums -> zpart[column block][row].
template <typename T> __global__ __launch_bounds__(BR_NT) void bd_row_kernel(const BdArgs<T> a)
{
	constexpr int CW = BR_TC / (BR_NT / 64); // 32 columns per wavefront
	__shared__ double s_row[BR_NT / 64][BR_TR];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, k = a.k;
	const int base = k + 1, rr = a.m - base, cc = a.n - base;
	const int ncb = (cc + BR_TC - 1) / BR_TC;
	const int I = blockIdx.x / ncb, J = blockIdx.x - I * ncb;
	const int i0 = I * BR_TR, j0 = J * BR_TC + CW * wv;
	const T *A22 = a.A + (idx_t) base * a.rs + (idx_t) base * a.cs;
	int gi[2];
#pragma unroll
	for (int h = 0; h < 2; ++h)
		gi[h] = min(i0 + lane + 64 * h, rr - 1);
	double racc[2] = {0.0, 0.0};
#pragma unroll
	for (int cb = 0; cb < CW; cb += 16) {
		const int gjl = min(j0 + cb + (lane & 15), cc - 1);
		const T xl = (j0 + cb + (lane & 15) < cc) ? a.vrow[base + gjl] : (T) 0; // (columns past the end contribute nothing)
		T v[2][16];
#pragma unroll
		for (int c = 0; c < 16; ++c)
#pragma unroll
			for (int h = 0; h < 2; ++h)
				v[h][c] = A22[(idx_t) gi[h] * a.rs + (idx_t) min(j0 + cb + c, cc - 1) * a.cs];
#pragma unroll
		for (int c = 0; c < 16; ++c) {
			const T xj = td_lane(xl, c);
#pragma unroll
			for (int h = 0; h < 2; ++h)
				racc[h] += (double) v[h][c] * (double) xj;
		}
	}
	s_row[wv][lane] = racc[0];
	s_row[wv][lane + 64] = racc[1];
	__syncthreads();
	if (tid < BR_TR && i0 + tid < rr)
		a.zpart[(size_t) J * a.m + base + i0 + tid] = ((s_row[0][tid] + s_row[1][tid]) + s_row[2][tid]) + s_row[3][tid];
}

// A: m x n; Hl: bl x min(m, n), Hr: br x (min(m, n) - 1)
template <typename T> void bidiag_dev(MatV<T> A, MatV<T> Hl, MatV<T> Hr)
{
	const idx_t m = A.nrows, n = A.ncols;
	// (m < n runs like the reference does -- svd/bidiag.rs loops over min(m, n) columns and leaves the last row of a wide
	// matrix normalised, without its right reflector; its SVD only ever passes tall matrices)
	const idx_t size = m < n ? m : n;
	FH_CHECK(Hl.ncols == size && Hr.ncols == (size > 0 ? size - 1 : 0), "bidiag: householder factors must have n and n - 1 columns");
	FH_CHECK((Hl.nrows > 0 || size == 0) && (Hr.nrows > 0 || size <= 1), "bidiag: householder factors need at least one row");
	FH_CHECK(m < (1L << 30) && n < (1L << 30), "bidiag: matrix too large");
	if (size == 0)
		return;
	hipStream_t s = ctx().stream;
	const idx_t nrb = (m + BC_TR - 1) / BC_TR, ncb = (n + BR_TC - 1) / BR_TC;
	const idx_t nhy = (n + TD_NT - 1) / TD_NT, nhz = (m + TD_NT - 1) / TD_NT; // helper blocks of bd_mid_kernel / bd_pre_kernel at most
	Scratch vb((size_t) (4 * n + m) * sizeof(T) + 256), stb(sizeof(BdState)), pb((size_t) (nrb * n + ncb * m + n + m) * sizeof(double)),
		fb((size_t) (nhy + nhz) * sizeof(xwg_u64));
	BdArgs<T> a;
	a.force_mem = g_l2_force_mem.load();
	a.A = A.p;
	a.rs = A.rs;
	a.cs = A.cs;
	a.m = (int) m;
	a.n = (int) n;
	a.size = (int) size;
	a.y = vb.as<T>();
	a.vrow = a.y + n;
	a.z = a.vrow + n;
	a.taul = a.z + m;
	a.taur = a.taul + n;
	a.ypart = pb.as<double>();
	a.zpart = a.ypart + (size_t) nrb * (size_t) n;
	a.ysum = a.zpart + (size_t) ncb * (size_t) m;
	a.zsum = a.ysum + n;
	a.yflag = fb.as<xwg_u64>();
	a.zflag = a.yflag + nhy;
	a.st = stb.as<BdState>();
	FH_HIP(hipMemsetAsync(vb.p, 0, (size_t) (4 * n + m) * sizeof(T), s));
	FH_HIP(hipMemsetAsync(a.ysum, 0, (size_t) (n + m) * sizeof(double), s));
	FH_HIP(hipMemsetAsync(fb.p, 0, (size_t) (nhy + nhz) * sizeof(xwg_u64), s));
	FH_HIP(hipMemsetAsync(stb.p, 0, sizeof(BdState), s));
	bool have_z = false; // a row pass has left shares for the next bd_pre_kernel
	for (idx_t k = 0; k < size; ++k) {
		a.k = (int) k;
		const idx_t rr = m - k - 1, cc = n - k - 1;
		// block 0: the step; blocks 1 ..: the sums of the previous row pass (1024 entries each)
		a.nh = have_z ? (int) ((m - k + TD_NT - 1) / TD_NT) : 0;
		hipLaunchKernelGGL(bd_pre_kernel<T>, dim3((unsigned) (1 + a.nh)), dim3(TD_NT), 0, s, a);
		have_z = false;
		if (cc > 0) {
			a.nh = 0;
			if (rr > 0) {
				const unsigned rb = (unsigned) ((rr + BC_TR - 1) / BC_TR), cb = (unsigned) ((cc + BC_TC - 1) / BC_TC);
				if (k > 0)
					hipLaunchKernelGGL((bd_col_kernel<T, true>), dim3(rb * cb), dim3(BC_NT), 0, s, a);
				else
					hipLaunchKernelGGL((bd_col_kernel<T, false>), dim3(rb * cb), dim3(BC_NT), 0, s, a);
				a.nh = (int) ((cc + TD_NT - 1) / TD_NT);
			} else {
				FH_HIP(hipMemsetAsync(a.ysum + k + 1, 0, (size_t) cc * sizeof(double), s)); // (no row below: y2 = 0)
			}
			hipLaunchKernelGGL(bd_mid_kernel<T>, dim3((unsigned) (1 + a.nh)), dim3(TD_NT), 0, s, a);
			if (k + 1 < size && rr > 0) {
				const unsigned rb = (unsigned) ((rr + BR_TR - 1) / BR_TR), cb = (unsigned) ((cc + BR_TC - 1) / BR_TC);
				hipLaunchKernelGGL(bd_row_kernel<T>, dim3(rb * cb), dim3(BR_NT), 0, s, a);
				have_z = true;
			}
		}
	}
	FH_HIP(hipGetLastError());
	// block Householder factors (:216-254): the left ones are in the QR layout, the right ones in its transpose
	qr_t_blocks_from_taus<T>(A, Hl, size, a.taul);
	if (size > 1)
		qr_t_blocks_from_taus<T>(A.sub(0, 1, size - 1, n - 1).t(), Hr, size - 1, a.taur);
	FH_HIP(hipStreamSynchronize(s)); // the scratch vectors above are released on return
}
template void bidiag_dev<double>(MatV<double>, MatV<double>, MatV<double>);
template void bidiag_dev<float>(MatV<float>, MatV<float>, MatV<float>);

// ------------------------------------------------------------------------------------------------
// Hessenberg reduction -- faer/src/linalg/evd/hessenberg.rs:230-408 (hessenberg_rearranged_unblocked; SURVEY.md section
// 8f item 4).  The reference switches to a blocked variant (hessenberg_gqvdg_blocked, :568-736) for n >= 256: the same
// reflectors of the same columns in another order of operations; this path runs the level-2 variant at every size (its
// passes are HBM streams here, not cache-blocked loops) and agrees with either up to rounding.
// Per column k: (i) row k, column k and a_kk receive the rest of the previous two-sided update (:266-281), (ii) the
// reflector of column k below the subdiagonal, head = 1 while the step runs (:294-305), (iii) ONE pass over A22 applying
// A22 -= u2 y2 + z2 u2^H and forming x^H A22 and A22 x (hessenberg_fused_op, :149-193), (iv) y2, z2 (:342-357), (v) the
// reflector from the right on row k and the rows above it (:358-378).  Three launches per column (round 6):
//   hs_pre_kernel(k)    block 0: (iv) of step k-1, restores its beta, (i), (ii); blocks 1 ..: w of step k-1 from the shares of its
//                       top-rows pass, and the application dwp = w / tau it leaves pending
//   hs_fused_kernel(k)  128 x 64 tiles of A22: the update written back and the tile's shares of BOTH x^H A22 and A22 x (same x)
//   hs_top_kernel(k)    128 x 64 tiles of the rows 0 .. k: the pending application of step k-1 written back, the shares of this
//                       step's w; extra blocks add the shares of hs_fused_kernel -> ysum, zsum
// Traffic per column: A22 and the k+1 rows above read and written ONCE (the reference's count -- and rounds 2-5 -- read A22 twice and the
// rows above twice).  N = 4096 fp64: 257.6 ms (round 5) -> 219 (one fused pass over A22) -> 165 (deferred application on the rows above).
// ------------------------------------------------------------------------------------------------
struct HsState {
	double tau_inv, beta;
};
template <typename T> struct HsArgs {
	T *A;
	idx_t rs, cs;
	int n, k, force_mem;
	T *y, *z, *ysum, *zsum, *taus;
	double *ypart, *zpart; // shares of the tiles of the fused pass: ypart[row block * n + j], zpart[column block * n + i]
	double *wpart;	       // shares of the tiles of the top-rows pass: wpart[column block * n + i]
	T *dwp;		       // per row i <= k - 1: w_i / tau of step k - 1, the right-side application that is still pending (0: none)
	HsState *st;
};

template <typename T> static __device__ __forceinline__ void hs_pre_body(const HsArgs<T> &a)
{
	__shared__ double s_part[(TD_NT / 64) * 3], s_red[3];
	const int tid = threadIdx.x, k = a.k, n = a.n;
	auto at = [&](int i, int j) -> T & { return a.A[(idx_t) i * a.rs + (idx_t) j * a.cs]; };
	const T sml = (T) scale_sml<T>(), big = (T) scale_big<T>();
	T nacc[3] = {0, 0, 0};
	if (k > 0) {
		// ---- (iv) of step k-1 (:342-357): x = [1; A[k+1.., k-1]] (its head is still 1 in memory), zsum = A22 x, ysum = x^H A22
		const T tau_inv = (T) a.st->tau_inv;
		double d[1] = {0.0};
		for (int i = k + tid; i < n; i += TD_NT)
			d[0] += (double) at(i, k - 1) * (double) a.zsum[i];
		td_block_sum<1>(d, s_part, s_red);
		const T b = ((T) s_red[0] * (T) 0.5) * tau_inv;
		const T x0 = at(k, k - 1); // == 1
		const T y1 = (a.ysum[k] - b * x0) * tau_inv, z1 = (a.zsum[k] - b * x0) * tau_inv;
		__syncthreads(); // everyone has read the head of the previous reflector
		// ---- (i) (:266-281) fused with the rest of (iv)
		if (tid == 0) {
			at(k, k - 1) = (T) a.st->beta; // (:379) the previous reflector's head goes back to beta
			a.y[k] = y1;
			a.z[k] = z1;
			at(k, k) -= y1 + z1;
		}
		for (int i = k + 1 + tid; i < n; i += TD_NT) {
			const T u = at(i, k - 1);
			const T yi = (a.ysum[i] - b * u) * tau_inv, zi = (a.zsum[i] - b * u) * tau_inv;
			a.y[i] = yi;
			a.z[i] = zi;
			at(k, i) -= yi + z1 * u;	    // row k: A12 -= y2 + z1 u2^H
			const T v = at(i, k) - (u * y1 + zi); // column k: A21 -= u2 y1 + z2
			at(i, k) = v;
			if (i >= k + 2) {
				nacc[0] += (v * sml) * (v * sml);
				nacc[1] += v * v;
				nacc[2] += (v * big) * (v * big);
			}
		}
	} else {
		for (int i = 2 + tid; i < n; i += TD_NT) {
			const T v = at(i, 0);
			nacc[0] += (v * sml) * (v * sml);
			nacc[1] += v * v;
			nacc[2] += (v * big) * (v * big);
		}
	}
	if (k + 1 >= n)
		return;
	// ---- (ii) reflector of column k below the subdiagonal (:294-305)
	double accd[3] = {(double) nacc[0], (double) nacc[1], (double) nacc[2]};
	td_block_sum<3>(accd, s_part, s_red);
	const T tail_norm = norm_from3<T>(s_red);
	T head = at(k + 1, k), hinv;
	bool negligible;
	const T tau = bd_householder<T>(head, tail_norm, hinv, negligible);
	__syncthreads(); // everyone has read the old head
	if (!negligible)
		for (int i = k + 2 + tid; i < n; i += TD_NT)
			at(i, k) *= hinv;
	if (tid == 0) {
		at(k + 1, k) = (T) 1; // head of x while the step runs; beta comes back in the next hs_pre_kernel
		a.taus[k] = tau;
		a.st->tau_inv = (double) ((T) 1 / tau);
		a.st->beta = (double) head;
	}
}

// hs_pre_body with column k - 1, column k, row k and the two products in registers (n - k <= TD_E TD_NT): one round trip to memory, every
// entry stored once.  Same arithmetic, expression by expression.
template <typename T> static __device__ __forceinline__ void hs_pre_body_reg(const HsArgs<T> &a)
{
	__shared__ double s_part[(TD_NT / 64) * 3], s_red[3];
	__shared__ T s_bc[1];
	const int tid = threadIdx.x, k = a.k, n = a.n;
	auto at = [&](int i, int j) -> T & { return a.A[(idx_t) i * a.rs + (idx_t) j * a.cs]; };
	const T sml = (T) scale_sml<T>(), big = (T) scale_big<T>();
	const bool upd = k > 0;
	// rows / columns i = k + tid + e TD_NT
	T u[TD_E], ck[TD_E], rk[TD_E], ys[TD_E], zs[TD_E];
#pragma unroll
	for (int e = 0; e < TD_E; ++e) {
		const int i = k + tid + e * TD_NT, ic = i < n ? i : n - 1;
		ck[e] = at(ic, k);
		u[e] = upd ? at(ic, k - 1) : (T) 0;
		rk[e] = upd ? at(k, ic) : (T) 0;
		ys[e] = upd ? a.ysum[ic] : (T) 0;
		zs[e] = upd ? a.zsum[ic] : (T) 0;
	}
	T nacc[3] = {0, 0, 0};
	if (upd) {
		// ---- (iv) of step k-1 (:342-357)
		const T tau_inv = (T) a.st->tau_inv, x0 = at(k, k - 1), ysk = a.ysum[k], zsk = a.zsum[k], beta = (T) a.st->beta;
		double d[1] = {0.0};
#pragma unroll
		for (int e = 0; e < TD_E; ++e)
			if (k + tid + e * TD_NT < n)
				d[0] += (double) u[e] * (double) zs[e];
		td_block_sum<1>(d, s_part, s_red);
		const T b = ((T) s_red[0] * (T) 0.5) * tau_inv;
		const T y1 = (ysk - b * x0) * tau_inv, z1 = (zsk - b * x0) * tau_inv;
		// ---- (i) (:266-281) fused with the rest of (iv)
#pragma unroll
		for (int e = 0; e < TD_E; ++e) {
			const int i = k + tid + e * TD_NT;
			if (i < n && i >= k + 1) {
				const T yi = (ys[e] - b * u[e]) * tau_inv, zi = (zs[e] - b * u[e]) * tau_inv;
				a.y[i] = yi;
				a.z[i] = zi;
				at(k, i) = rk[e] - (yi + z1 * u[e]);   // row k: A12 -= y2 + z1 u2^H
				ck[e] = ck[e] - (u[e] * y1 + zi);       // column k: A21 -= u2 y1 + z2
			}
		}
		if (tid == 0) {
			at(k, k - 1) = beta; // (:379) the previous reflector's head goes back to beta
			a.y[k] = y1;
			a.z[k] = z1;
			at(k, k) = ck[0] - (y1 + z1);
		}
	}
#pragma unroll
	for (int e = 0; e < TD_E; ++e) {
		const int i = k + tid + e * TD_NT;
		if (i < n && i >= k + 2) {
			const T v = ck[e];
			nacc[0] += (v * sml) * (v * sml);
			nacc[1] += v * v;
			nacc[2] += (v * big) * (v * big);
		}
	}
	if (k + 1 >= n)
		return;
	if (tid == 1)
		s_bc[0] = ck[0]; // row k + 1: the head of the column
	// ---- (ii) reflector of column k below the subdiagonal (:294-305)
	double accd[3] = {(double) nacc[0], (double) nacc[1], (double) nacc[2]};
	td_block_sum<3>(accd, s_part, s_red);
	const T tail_norm = norm_from3<T>(s_red);
	T head = s_bc[0], hinv;
	bool negligible;
	const T tau = bd_householder<T>(head, tail_norm, hinv, negligible);
#pragma unroll
	for (int e = 0; e < TD_E; ++e) {
		const int i = k + tid + e * TD_NT;
		if (i < n && i >= k + 2 && (upd || !negligible))
			at(i, k) = negligible ? ck[e] : ck[e] * hinv;
	}
	if (tid == 0) {
		at(k + 1, k) = (T) 1; // head of x while the step runs; beta comes back in the next hs_pre_kernel
		a.taus[k] = tau;
		a.st->tau_inv = (double) ((T) 1 / tau);
		a.st->beta = (double) head;
	}
}

template <typename T> __global__ __launch_bounds__(TD_NT) void hs_pre_kernel(const HsArgs<T> a)
{
	if (blockIdx.x > 0) {
		// helper block: rows 1024 (blockIdx.x - 1) ..: w of step k - 1 = the shares of its top-rows pass in a fixed order, then the
		// pending application dwp = w / tau_{k-1} for hs_top_kernel(k).  Independent of block 0 (which overwrites st->tau_inv: taken from taus)
		const int k = a.k, i = ((int) blockIdx.x - 1) * TD_NT + (int) threadIdx.x;
		if (i >= k)
			return;
		const int np = (a.n - (k - 1) + TF_TC - 1) / TF_TC;
		const double *src = a.wpart + i;
		double s0 = 0.0;
		int p = 0;
		for (; p + 8 <= np; p += 8) {
			double v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u)
				v[u] = src[(size_t) (p + u) * a.n];
#pragma unroll
			for (int u = 0; u < 8; ++u)
				s0 += v[u];
		}
		for (; p < np; ++p)
			s0 += src[(size_t) p * a.n];
		a.dwp[i] = (T) s0 * ((T) 1 / a.taus[k - 1]);
		return;
	}
	if (a.n - a.k <= TD_E * TD_NT && !a.force_mem)
		hs_pre_body_reg<T>(a);
	else
		hs_pre_body<T>(a);
}

// Round 6: both products of a step use the SAME x, so ONE pass over A22 = A[k+1.., k+1..] applies the two-sided update of the previous step
// and forms the tile's shares of l_out = x^H A22 (column sums) and r_out = A22 x (row sums): 128 x 64 tiles as in the tridiagonalization
// (lane = two rows, wavefront = 16 columns, 32 loads in flight per thread); the shares are added in a fixed order by extra blocks of hs_top_kernel.
// The rows above (0 .. k, which receive the reflector from the right once their sums are complete) stay with hs_rowpass_kernel.
template <typename T, bool upd> __global__ __launch_bounds__(TF_NT) void hs_fused_kernel(const HsArgs<T> a)
{
	constexpr int CW = TF_TC / (TF_NT / 64); // 16 columns per wavefront
	__shared__ double s_row[TF_NT / 64][TF_TR];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, k = a.k;
	const int base = k + 1, r = a.n - base;
	const int ncb = (r + TF_TC - 1) / TF_TC;
	const int I = blockIdx.x / ncb, J = blockIdx.x - I * ncb;
	const int i0 = I * TF_TR, j0 = J * TF_TC + CW * wv;
	T *A22 = a.A + (idx_t) base * a.rs + (idx_t) base * a.cs;
	const T *xcol = a.A + (idx_t) base * a.rs + (idx_t) k * a.cs;		    // x (head = 1 in memory)
	const T *ucol = a.A + (idx_t) base * a.rs + (idx_t) (upd ? k - 1 : 0) * a.cs; // u2: the previous reflector's tail
	int gi[2];
	bool vr[2];
	T xi[2], ui[2], zi[2];
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		gi[h] = i0 + lane + 64 * h;
		vr[h] = gi[h] < r;
		gi[h] = min(gi[h], r - 1);
		const idx_t o = (idx_t) gi[h] * a.rs;
		xi[h] = xcol[o];
		ui[h] = upd ? ucol[o] : (T) 0;
		zi[h] = upd ? a.z[base + gi[h]] : (T) 0;
	}
	T xjl, yjl = (T) 0, ujl = (T) 0;
	{
		const int gj = min(j0 + (lane & (CW - 1)), r - 1);
		xjl = xcol[(idx_t) gj * a.rs];
		if (upd) {
			yjl = a.y[base + gj];
			ujl = ucol[(idx_t) gj * a.rs];
		}
	}
	T v[2][CW];
#pragma unroll
	for (int c = 0; c < CW; ++c)
#pragma unroll
		for (int h = 0; h < 2; ++h)
			v[h][c] = A22[(idx_t) gi[h] * a.rs + (idx_t) min(j0 + c, r - 1) * a.cs];
	double racc[2] = {0.0, 0.0};
#pragma unroll
	for (int c = 0; c < CW; ++c) {
		const int gj = j0 + c;
		const T xj = td_lane(xjl, c), yj = td_lane(yjl, c), uj = td_lane(ujl, c);
		double cs_ = 0.0;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const bool in = vr[h] && gj < r;
			T tv = v[h][c];
			if (upd) {
				tv = fh_fma(-ui[h], yj, tv); // A22 -= u2 y2      (:160-167)
				tv = fh_fma(-zi[h], uj, tv); // A22 -= z2 u2^H    (:168-175)
				if (in)
					A22[(idx_t) gi[h] * a.rs + (idx_t) gj * a.cs] = tv;
			}
			cs_ += in ? (double) xi[h] * (double) tv : 0.0;	 // l_out = x^H A22 (:184-191)
			racc[h] += in ? (double) tv * (double) xj : 0.0; // r_out = A22 x   (:176-183)
		}
		const double sv = td_wave_sum(cs_);
		if (lane == 0 && gj < r)
			a.ypart[(size_t) I * a.n + base + gj] = sv;
	}
	s_row[wv][lane] = racc[0];
	s_row[wv][lane + 64] = racc[1];
	__syncthreads();
	if (tid < TF_TR && i0 + tid < r)
		a.zpart[(size_t) J * a.n + base + i0 + tid] = ((s_row[0][tid] + s_row[1][tid]) + s_row[2][tid]) + s_row[3][tid];
}

// Rows 0 .. k (hs_fused_kernel has the rows below): their sums with x are the w of the right-side application (:358-378), A[0..k, k+1..] -=
// (w / tau) x^H.  Rounds 2-5 gave 16 rows to a workgroup that summed them and then applied the reflector (every row read twice per step).
// Round 6: the application is DEFERRED by one step and rides on the next step's pass -- tile (rows 0 .. k) x (columns k .. n-1): entries
// first receive the pending application of step k - 1 (dwp_i x_{k-1,j}; column k only that: it is final afterwards), are written back, and
// contribute to the new sums with x_k; the shares of the tiles are added by helper blocks of the next hs_pre_kernel, which also turn them
// into the next dwp.  One read and one write per entry and step; a last call behind the loop (k = n - 1) applies what is still pending.
// The blocks behind the tiles add the shares of hs_fused_kernel in a fixed order -> ysum, zsum (one launch less per column).
template <typename T> __global__ __launch_bounds__(TF_NT) void hs_top_kernel(const HsArgs<T> a)
{
	constexpr int CW = TF_TC / (TF_NT / 64); // 16 columns per wavefront
	__shared__ double s_row[TF_NT / 64][TF_TR];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, k = a.k, n = a.n;
	const int nrows = k + 1, ncols = n - k; // rows 0 .. k, columns k .. n - 1
	const int ncb = (ncols + TF_TC - 1) / TF_TC, nrb = (nrows + TF_TR - 1) / TF_TR;
	if ((int) blockIdx.x >= nrb * ncb) {
		const int r = n - (k + 1);
		const int e = ((int) blockIdx.x - nrb * ncb) * TF_NT + tid;
		if (e >= 2 * r)
			return;
		const bool isz = e >= r;
		const int np = isz ? (r + TF_TC - 1) / TF_TC : (r + TF_TR - 1) / TF_TR;
		const double *src = (isz ? a.zpart : a.ypart) + (k + 1) + (isz ? e - r : e);
		double s0 = 0.0;
		int p = 0;
		for (; p + 8 <= np; p += 8) {
			double v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u)
				v[u] = src[(size_t) (p + u) * n];
#pragma unroll
			for (int u = 0; u < 8; ++u)
				s0 += v[u];
		}
		for (; p < np; ++p)
			s0 += src[(size_t) p * n];
		(isz ? a.zsum : a.ysum)[(k + 1) + (isz ? e - r : e)] = (T) s0;
		return;
	}
	const bool upd = k > 0;
	const int I = blockIdx.x / ncb, J = blockIdx.x - I * ncb;
	const int i0 = I * TF_TR, j0 = k + J * TF_TC + CW * wv; // (absolute column)
	int gi[2];
	bool vr[2];
	T dwi[2];
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		gi[h] = i0 + lane + 64 * h;
		vr[h] = gi[h] < nrows;
		gi[h] = min(gi[h], nrows - 1);
		dwi[h] = upd ? a.dwp[gi[h]] : (T) 0;
	}
	T xpl, xnl; // per lane: the pending reflector x_{k-1} and the new one x_k at this wavefront's columns
	{
		const int gj = min(j0 + (lane & (CW - 1)), n - 1);
		xpl = !upd ? (T) 0 : (gj == k ? (T) 1 : a.A[(idx_t) gj * a.rs + (idx_t) (k - 1) * a.cs]);
		xnl = gj >= k + 1 ? a.A[(idx_t) gj * a.rs + (idx_t) k * a.cs] : (T) 0; // (head = 1 in memory while the step runs)
	}
	T v[2][CW];
#pragma unroll
	for (int c = 0; c < CW; ++c)
#pragma unroll
		for (int h = 0; h < 2; ++h)
			v[h][c] = a.A[(idx_t) gi[h] * a.rs + (idx_t) min(j0 + c, n - 1) * a.cs];
	double racc[2] = {0.0, 0.0};
#pragma unroll
	for (int c = 0; c < CW; ++c) {
		const int gj = j0 + c;
		const T xp = td_lane(xpl, c), xn = td_lane(xnl, c);
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const bool in = vr[h] && gj < n;
			T tv = v[h][c];
			if (upd) {
				tv = tv - dwi[h] * xp;
				if (in)
					a.A[(idx_t) gi[h] * a.rs + (idx_t) gj * a.cs] = tv;
			}
			racc[h] += in ? (double) tv * (double) xn : 0.0;
		}
	}
	s_row[wv][lane] = racc[0];
	s_row[wv][lane + 64] = racc[1];
	__syncthreads();
	if (tid < TF_TR && i0 + tid < nrows)
		a.wpart[(size_t) J * n + i0 + tid] = ((s_row[0][tid] + s_row[1][tid]) + s_row[2][tid]) + s_row[3][tid];
}

// A: n x n, H: block_size x (n - 1)
template <typename T> void hessenberg_dev(MatV<T> A, MatV<T> H)
{
	const idx_t n = A.nrows;
	FH_CHECK(A.ncols == n, "hessenberg: the matrix must be square");
	FH_CHECK(H.ncols == (n > 0 ? n - 1 : 0), "hessenberg: householder must be block_size x (n - 1)");
	FH_CHECK(n < (1L << 30), "hessenberg: matrix too large");
	if (n <= 1)
		return;
	FH_CHECK(H.nrows > 0, "hessenberg: householder needs at least one row");
	hipStream_t s = ctx().stream;
	const idx_t nrb = (n + TF_TR - 1) / TF_TR, ncb = (n + TF_TC - 1) / TF_TC;
	Scratch vb((size_t) (6 * n) * sizeof(T) + 256), stb(sizeof(HsState)), pb((size_t) (nrb + 2 * ncb) * (size_t) n * sizeof(double));
	HsArgs<T> a;
	a.force_mem = g_l2_force_mem.load();
	a.ypart = pb.as<double>();
	a.zpart = a.ypart + (size_t) nrb * (size_t) n;
	a.wpart = a.zpart + (size_t) ncb * (size_t) n;
	a.A = A.p;
	a.rs = A.rs;
	a.cs = A.cs;
	a.n = (int) n;
	a.y = vb.as<T>();
	a.z = a.y + n;
	a.ysum = a.z + n;
	a.zsum = a.ysum + n;
	a.taus = a.zsum + n;
	a.dwp = a.taus + n;
	a.st = stb.as<HsState>();
	FH_HIP(hipMemsetAsync(vb.p, 0, (size_t) (6 * n) * sizeof(T), s));
	FH_HIP(hipMemsetAsync(stb.p, 0, sizeof(HsState), s));
	auto launch_top = [&](idx_t k) {
		const idx_t r = n - k - 1;
		const unsigned tiles = (unsigned) (((k + 1 + TF_TR - 1) / TF_TR) * ((n - k + TF_TC - 1) / TF_TC));
		hipLaunchKernelGGL(hs_top_kernel<T>, dim3(tiles + (unsigned) ((2 * r + TF_NT - 1) / TF_NT)), dim3(TF_NT), 0, s, a);
	};
	for (idx_t k = 0; k < n; ++k) {
		a.k = (int) k;
		// block 0: the step; blocks 1 ..: w of step k - 1 and the pending application it leaves (1024 rows each)
		hipLaunchKernelGGL(hs_pre_kernel<T>, dim3((unsigned) (1 + (k + TD_NT - 1) / TD_NT)), dim3(TD_NT), 0, s, a);
		const idx_t r = n - k - 1;
		if (r > 0) {
			const unsigned rb = (unsigned) ((r + TF_TR - 1) / TF_TR), cb = (unsigned) ((r + TF_TC - 1) / TF_TC);
			if (k > 0)
				hipLaunchKernelGGL((hs_fused_kernel<T, true>), dim3(rb * cb), dim3(TF_NT), 0, s, a);
			else
				hipLaunchKernelGGL((hs_fused_kernel<T, false>), dim3(rb * cb), dim3(TF_NT), 0, s, a);
			launch_top(k);
		}
	}
	// the application of the last step (k = n - 2) is still pending on column n - 1
	a.k = (int) (n - 1);
	launch_top(n - 1);
	FH_HIP(hipGetLastError());
	qr_t_blocks_from_taus<T>(A.sub(1, 0, n - 1, n - 1), H, n - 1, a.taus); // (:382-406)
	FH_HIP(hipStreamSynchronize(s)); // the scratch vectors above are released on return
}
template void hessenberg_dev<double>(MatV<double>, MatV<double>);
template void hessenberg_dev<float>(MatV<float>, MatV<float>);

// Backup copy of A for the fast path, fused with the range guard of the fp64 fast path: the cooperative leaf
// accumulates PLAIN squares and dot products in fp64 (exact for fp32 data, whose squares cannot leave the fp64
// range), whereas the reference's norm_l2 keeps three differently scaled accumulators
// (reductions/norm_l2.rs:6-45,173-184) and stays accurate for |x| ~ 1e+-250.  fp64 data with a non-zero entry
// outside [1e-120, 1e120] (or a non-finite one) therefore raises the "abandon the fast path" flag and the
// factorization runs on the general path, which restates norm_l2 literally.
template <typename T>
__global__ void copy_guard_kernel(T *d, idx_t drs, idx_t dcs, const T *s, idx_t srs, idx_t scs, idx_t M, idx_t N, int *flag)
{
	const idx_t total = M * N;
	bool bad = false;
	for (idx_t e = (idx_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (idx_t) gridDim.x * blockDim.x) {
		const idx_t i = e % M, j = e / M;
		const T v = s[i * srs + j * scs];
		d[i * drs + j * dcs] = v;
		if constexpr (sizeof(T) == 8) {
			const double av = fabs((double) v);
			bad = bad || (av != 0.0 && !(av >= 1e-120 && av <= 1e120));
		}
	}
	if (__any(bad) && (threadIdx.x & 63) == 0)
		atomicExch(flag, 1);
}

// `off` > 0: A is the trailing part parent(off:, off:) of a matrix whose first `off` columns are finished reflectors (the
// one-pass path stopped there).  The reference's rank test looks at a column's entries in ALL rows above the current one
// (factor.rs:26,52-58), so the `off` parent rows above A count: both paths below read them through negative row offsets.
template <typename T> static long geqrf_classic(MatV<T> A, MatV<T> H, idx_t blocking_threshold, idx_t off = 0)
{
	(void) blocking_threshold; // the GPU recursion always blocks; leaves are 8 columns wide
	const idx_t m = A.nrows, n = A.ncols;
	const idx_t size = m < n ? m : n;
	const idx_t bs = H.nrows;
	FH_CHECK(bs > 0 && H.ncols == size, "qr: Q_coeff must be block_size x min(nrows, ncols)");
	FH_CHECK(m < (1L << 30) && n < (1L << 30), "qr: matrix too large");
	if (size == 0)
		return 0;
	hipStream_t s = ctx().stream;
	Scratch misc(256);
	FH_HIP(hipMemsetAsync(misc.p, 0, 256, s));
	int *status = misc.as<int>() + 8;
	long rank = -1;

	// The reference's rank test (factor.rs:52-58) accepts column 0 only if norm > eps * 16 * m * norm, i.e. it
	// rejects EVERY column once 16 * eps * nrows >= 1 (fp32: nrows >= 524288).  That outcome (rank 0) is
	// reproduced by the general path; the fast path would only discover it one launch later.
	const bool ref_rejects_all = (double) Lim<T>::eps * 16.0 * (double) m >= 1.0;
	// the cooperative leaf needs all its workgroups resident: one 512-thread workgroup per CU at its register footprint
	hipDeviceProp_t prop;
	FH_HIP(hipGetDeviceProperties(&prop, ctx().device));
	const idx_t gcap = prop.multiProcessorCount < QR_GMAX ? prop.multiProcessorCount : QR_GMAX;
	const bool fast_ok = m <= (idx_t) QR2_NT * qr2_rpt<T>() * gcap && !ref_rejects_all;
	Scratch backup(fast_ok ? (size_t) m * (size_t) n * sizeof(T) : 256);
	MatV<T> Bk{backup.as<T>(), m, n, 1, m};
	if (fast_ok) {
		{
			const idx_t total = m * n;
			idx_t blocks = (total + 255) / 256;
			if (blocks > 65536)
				blocks = 65536;
			MatV<T> Ad = A, Bd = Bk;
			auto ab = [](idx_t x) { return x < 0 ? -x : x; };
			if (ab(Ad.cs) < ab(Ad.rs)) { // fast index along the smaller stride
				Ad = Ad.t();
				Bd = Bd.t();
			}
			hipLaunchKernelGGL(copy_guard_kernel<T>, dim3((unsigned) blocks), dim3(256), 0, s, Bd.p, Bd.rs, Bd.cs, Ad.p, Ad.rs, Ad.cs,
					   Ad.nrows, Ad.ncols, status + 3);
			FH_HIP(hipGetLastError());
		}
		Scratch slots((size_t) 2 * QR_GMAX * QR_SLOT * sizeof(double)), head((size_t) 2 * (QR_PW + 1) * sizeof(double));
		QrWork<T> wk;
		wk.slots = slots.as<double>();
		wk.head = head.as<double>();
		Scratch flagb((size_t) QR_GMAX * sizeof(xwg_u64));
		FH_HIP(hipMemsetAsync(flagb.p, 0, (size_t) QR_GMAX * sizeof(xwg_u64), s));
		wk.flags = flagb.as<xwg_u64>();
		const size_t gran_n = (size_t) 2 * QR_GMAX * 2 * QR_PW + (size_t) 2 * 2 * (QR_PW + 1);
		Scratch granb(gran_n * sizeof(xwg_u64));
		FH_HIP(hipMemsetAsync(granb.p, 0, gran_n * sizeof(xwg_u64), s));
		wk.gran = granb.as<xwg_u64>();
		wk.gran_head = wk.gran + (size_t) 2 * QR_GMAX * 2 * QR_PW;
		wk.epoch_base = 0;
		wk.status = status;
		wk.a_top = A.p - off * A.rs - off * A.cs; // origin of the parent
		wk.rs = A.rs;
		wk.cs = A.cs;
		for (idx_t c0 = 0; c0 < size; c0 += bs) {
			const idx_t wb = bs < size - c0 ? bs : size - c0;
			MatV<T> P = A.sub(c0, c0, m - c0, wb);
			MatV<T> Tb = H.sub(0, c0, wb, wb);
			qr_rec<T>(P, Tb, off + c0, off + c0, wk);
			if (c0 + wb < n) // factor.rs:241-249: apply Q_k^H to everything on the right
				apply_block_householder_dev<T>(P.c(), Tb.c(), A.sub(c0, c0 + wb, m - c0, n - c0 - wb), true);
		}
		int st[4];
		FH_HIP(hipMemcpyAsync(st, status, sizeof(st), hipMemcpyDeviceToHost, s));
		FH_HIP(hipStreamSynchronize(s));
		if (st[2] != 0) // the cooperative leaf did not get all its workgroups resident in time (GPU shared with other work):
			fprintf(stderr, "faer_hip: qr: the cross-workgroup exchange of the panel kernel timed out; redoing on the general path\n");
		if (st[2] == 0 && st[3] == 0)
			rank = (long) size;
		else
			copy_dev<T>(A, Bk.c()); // rank deficient (or timed out): redo from the saved copy on the general path
	}
	if (rank < 0) {
		Scratch taus((size_t) size * sizeof(T));
		rank = qr_general<T>(A, taus.as<T>(), off);
		qr_t_blocks_from_taus<T>(A, H, rank, taus.as<T>());
		FH_HIP(hipStreamSynchronize(s)); // taus scratch is released on return
	}
	return rank;
}

// tsqr.hip: the one-pass path for tall fp32 matrices
// (fp64: fp64 Gram sums -- well-conditioned panels only, the rest goes to the classic path)
static inline bool tsqr_ok(const MatV<float> &A, idx_t bs) { return tsqr_applicable(A.nrows, A.ncols, A.rs, A.cs, bs); }
static inline bool tsqr_ok(const MatV<double> &A, idx_t bs) { return tsqr_applicable64(A.nrows, A.ncols, A.rs, A.cs, bs, A.p); }

template <typename T> __global__ void qr_taus_from_blocks_kernel(const T *H, idx_t hrs, idx_t hcs, int bs, int count, T *taus)
{
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < count)
		taus[j] = H[(idx_t) (j % bs) * hrs + (idx_t) j * hcs];
}

// Tall matrices take the one-pass path (tsqr.hip; fp64 since the end of round 6).  It stops in front of the first 64-column panel it cannot
// take (ill conditioned, a column failing the reference's rank test, ...) with every earlier reflector applied to
// everything on its right -- the state qr_in_place_blocked (factor.rs:137-256) is in at that column -- so the classic
// path simply factors the remaining submatrix and the T blocks are rebuilt from V and the taus.
static thread_local long g_qr_one_pass_columns = -1; // of this thread's last QR: columns the one-pass path completed, -1 = not taken
long qr_last_one_pass_columns() { return g_qr_one_pass_columns; }

template <typename T> long geqrf_dev(MatV<T> A, MatV<T> H, idx_t blocking_threshold)
{
	g_qr_one_pass_columns = -1;
	{
		const idx_t m = A.nrows, n = A.ncols, bs = H.nrows;
		const idx_t size = m < n ? m : n;
		const bool ref_rejects_all = (double) Lim<T>::eps * 16.0 * (double) m >= 1.0;
		// A block size of Q_coeff that neither divides a 64-column panel nor is a multiple of it (faer recommends 48 for many moderately
		// tall shapes): the path runs with 64-column blocks of its own and the caller's blocks are rebuilt from V and the taus (one
		// batched Gram launch over V: one workgroup per block walks all rows -- 8192 x 64 with blocks of 48: 1.2 ms of rebuild against
		// 0.44 ms for the whole classic factorization -- so only up to 3072 rows: tools/gpu_qr_shape_rule.py).
		const bool bs_direct = bs > 0 && (bs % 64 == 0 || 64 % bs == 0);
		if (size > 0 && bs > 0 && H.ncols == size && !ref_rejects_all && (bs_direct || m <= 3072) && tsqr_ok(A, bs_direct ? bs : 64)) {
			Scratch taus((size_t) size * sizeof(T));
			Scratch hown(bs_direct ? 256 : (size_t) 64 * (size_t) size * sizeof(T));
			int reason = 0;
			const idx_t done = tsqr_run(A, bs_direct ? H : MatV<T>{hown.as<T>(), 64, size, 1, 64}, taus.as<T>(), &reason);
			g_qr_one_pass_columns = (long) done;
			if (done == size) {
				if (!bs_direct) {
					qr_t_blocks_from_taus<T>(A, H, size, taus.as<T>());
					FH_HIP(hipStreamSynchronize(ctx().stream)); // scratch is released on return
				}
				return (long) size;
			}
			static const bool verbose = getenv("FAER_HIP_VERBOSE") != nullptr;
			if (verbose)
				fprintf(stderr, "faer_hip: qr: one-pass path stopped at column %ld (reason %d); classic path for the rest\n", (long) done, reason);
			MatV<T> B = A.sub(done, done, m - done, n - done);
			const idx_t size2 = size - done;
			Scratch h2((size_t) bs * (size_t) size2 * sizeof(T));
			MatV<T> H2{h2.as<T>(), bs, size2, 1, bs};
			fill_dev<T>(H2, DST_FULL, (T) 0);
			const long r2 = geqrf_classic<T>(B, H2, blocking_threshold, done);
			if (r2 > 0) {
				hipLaunchKernelGGL(qr_taus_from_blocks_kernel<T>, dim3((unsigned) ((r2 + 255) / 256)), dim3(256), 0, ctx().stream, H2.p, H2.rs, H2.cs,
						   (int) bs, (int) r2, taus.as<T>() + done);
				FH_HIP(hipGetLastError());
			}
			const long rank = (long) done + r2;
			qr_t_blocks_from_taus<T>(A, H, rank, taus.as<T>());
			FH_HIP(hipStreamSynchronize(ctx().stream)); // scratch is released on return
			return rank;
		}
	}
	return geqrf_classic<T>(A, H, blocking_threshold);
}

template long geqrf_dev<double>(MatV<double>, MatV<double>, idx_t);
template long geqrf_dev<float>(MatV<float>, MatV<float>, idx_t);
template void apply_householder_sequence_left_dev<double>(MatV<const double>, MatV<const double>, MatV<double>, bool);
template void apply_householder_sequence_left_dev<float>(MatV<const float>, MatV<const float>, MatV<float>, bool);

} // namespace fh
