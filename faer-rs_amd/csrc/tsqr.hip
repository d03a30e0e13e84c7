// One-pass tall-skinny Householder QR for gfx950 (fp32 data): faer's (V, T, R) -- qr/no_pivoting/factor.rs:137-256,
// householder.rs:21-23,59-107,132-272 -- without a cross-workgroup reduction per column.
//
// The classic path (qr.hip) follows the reference's recursion: every column of a panel costs one device-wide
// all-reduce and every level of the recursion a handful of dependent launches; a 5e5 x 256 matrix is streamed
// dozens of times and nothing is bound by HBM or by the matrix cores.  Here the matrix is processed in 64-column
// panels with a constant number of launches per panel, every trailing column read and written once per panel:
//
//   gram    G = P^T P (fp64 matrix cores: products of fp32 data are exact in fp64) and C = P^T X (fp32 matrix cores,
//           per-workgroup partial sums added in fp64 in a fixed order) over the rows from the panel's diagonal down;
//           P = the 64 panel columns, X = the columns right of it.  Row chunks go through LDS; the row order inside
//           a chunk is irrelevant for a Gram sum, so operands are read back as 16-byte quads of consecutive rows.
//   panel   (one workgroup, fp64) R~ = chol(G)^T; Householder reconstruction on the top block A1 (Ballard, Demmel,
//           Grigori, Jacquelin, Nguyen, Solomonik: "Reconstructing Householder vectors from TSQR"): Q1~ = A1 R~^-1,
//           sign-choosing LU  I - Q1~ S = V1 U  (s_j = -sign of the pivot candidate: exactly the reference's
//           beta = -sign(x0) |x|, householder.rs:82-101), R = S R~, M = -(U R)^-1, T = V1^T U^-1
//           (= striu(V^T V) + diag(tau), the reference's factor with H = I - V T^-1 V^T).
//           Below the top block the reflectors are V = P M: one more product, no reduction.
//   y       D = R^-T C are the new top rows of X (= Q^T X), Y = R^-1 (V1 U)^-1 (D - X_top); below the top block the
//           block reflector applied to X is X - P Y (the raw panel again, not V).
//   update  X <- X - P Y and V = P M in ONE pass: rows are independent, so a wavefront maps ANY 32 rows to the 32
//           columns of a v_mfma_f32_32x32x2 tile -- each lane loads and stores 16-byte quads of its own rows straight
//           from / to HBM, no LDS on the streamed side.
//
// Accuracy: R comes from an fp64 Cholesky factor of an exactly accumulated Gram matrix: relative error
// ~ cond(panel)^2 * 2^-53, i.e. below fp32 rounding for cond(panel) < ~1e4 (measured on the CPU prototype
// tools/proto_tsqr.py: closer to an fp64 Householder QR than the fp32 Householder QR is).  V = P M and D = R^-T C
// amplify fp32 rounding by cond(panel), so the panel kernel REFUSES a panel whose Frobenius condition estimate exceeds
// TQ_COND_MAX, whose updated column is (numerically) zero below the diagonal (the reference's tau = +inf case) or whose
// column fails the reference's rank test (factor.rs:52-64, evaluated from R): nothing of that panel has been written
// at that point, every earlier reflector has been applied to everything right of it, and geqrf_dev continues with the
// classic path on the remaining submatrix.  fp64 input never comes here (it would need a wider Gram accumulator).
#include "common.h"

namespace fh {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int TQ_PW = 64;   // panel width
constexpr int TQ_LP = 68;   // LDS pitch (floats) of one staged column of a 64-row chunk: 16-byte aligned, quad slots rotate
constexpr int TQ_TS = 192;  // widest strip of trailing columns per launch (LDS: 256 staged columns)
constexpr int TQ_NB = 512;  // workgroups (= partial sums) of a Gram launch: two per CU
constexpr int TQ_DP = 65;   // pitch of the fp64 64 x 64 matrices in LDS
constexpr double TQ_COND_MAX = 64.0 * 512.0; // |R|_F |R^-1|_F (>= 64 for any panel): cond_2(panel) below ~512
constexpr double TQ_TAIL_MIN = 1e-9;	     // 1 - |head| / |column| below this: the tail is numerically zero
enum { TQ_OK = 0, TQ_FAIL_CHOL = 1, TQ_FAIL_TAIL = 2, TQ_FAIL_RANK = 3, TQ_FAIL_COND = 4, TQ_FAIL_RANGE = 5 };

// ------------------------------------------------------------------------------------------------
// gram
// ------------------------------------------------------------------------------------------------
struct TqGramArgs {
	const float *P; // A[r0, c0]
	const float *X; // A[r0, cx]
	long ld;
	int rows, w, t, tp; // rows from r0 down, panel width, trailing columns of this launch, t rounded up to 32
	int nchunks;
	int want_g, want_sq;
	double *Gp; // [grid][64 * 64]
	float *Cp;  // [grid][64 * tp]
	float *Sp;  // [grid][256] per-column sums of squares of the staged columns (range guard of the first launch)
	const int *stat;
};

template <bool VEC> __global__ __launch_bounds__(256, 2) void tq_gram_kernel(const TqGramArgs a)
{
	__shared__ float sm[(TQ_PW + TQ_TS) * TQ_LP];
	if (a.stat[0])
		return;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int quad = tid & 15, cg = tid >> 4;
	const int ncol = TQ_PW + a.tp;
	const int ntile = 2 * (a.tp >> 5);
	f64x4 gacc[4];
	f32x16 cacc[3];
#pragma unroll
	for (int i = 0; i < 4; ++i)
		gacc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
	for (int i = 0; i < 3; ++i)
#pragma unroll
		for (int r = 0; r < 16; ++r)
			cacc[i][r] = 0.f;
	float sq[16];
#pragma unroll
	for (int i = 0; i < 16; ++i)
		sq[i] = 0.f;
	f32x4 st[16];
	auto load_chunk = [&](int ch) {
		const int rbase = ch * 64 + quad * 4;
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			f32x4 v = {0.f, 0.f, 0.f, 0.f};
			const int c = i * 16 + cg;
			if (i * 16 < ncol) {
				const bool isp = c < TQ_PW;
				const int cc = isp ? c : c - TQ_PW;
				const bool colok = isp ? cc < a.w : cc < a.t;
				if (colok && rbase < a.rows) {
					const float *src = (isp ? a.P : a.X) + (long) cc * a.ld + rbase;
					if (VEC && rbase + 3 < a.rows) {
						v = *reinterpret_cast<const f32x4 *>(src);
					} else {
#pragma unroll
						for (int e = 0; e < 4; ++e)
							if (rbase + e < a.rows)
								v[e] = src[e];
					}
				}
			}
			st[i] = v;
		}
	};
	int ch = blockIdx.x;
	if (ch < a.nchunks)
		load_chunk(ch);
	for (; ch < a.nchunks; ch += gridDim.x) {
		__syncthreads(); // the previous chunk has been consumed
#pragma unroll
		for (int i = 0; i < 16; ++i)
			if (i * 16 < ncol) {
				*reinterpret_cast<f32x4 *>(&sm[(i * 16 + cg) * TQ_LP + quad * 4]) = st[i];
				if (a.want_sq)
					sq[i] += st[i][0] * st[i][0] + st[i][1] * st[i][1] + st[i][2] * st[i][2] + st[i][3] * st[i][3];
			}
		__syncthreads();
		if (ch + (int) gridDim.x < a.nchunks)
			load_chunk(ch + gridDim.x); // in flight during the products
		if (a.want_g) {
			// wave wv: the 16 x 64 block row wv of G; lane: A[i = l & 15][k = l >> 4], four k-slices per 16-byte read
#pragma unroll
			for (int s = 0; s < 4; ++s) {
				const int roff = 16 * s + 4 * (lane >> 4);
				const f32x4 av = *reinterpret_cast<const f32x4 *>(&sm[(16 * wv + (lane & 15)) * TQ_LP + roff]);
#pragma unroll
				for (int ib = 0; ib < 4; ++ib) {
					const f32x4 bv = *reinterpret_cast<const f32x4 *>(&sm[(16 * ib + (lane & 15)) * TQ_LP + roff]);
#pragma unroll
					for (int q = 0; q < 4; ++q)
						gacc[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64((double) av[q], (double) bv[q], gacc[ib], 0, 0, 0);
				}
			}
		}
		if (ntile > 0) {
			// 32 x 32 tiles of C: tile tl = wv + 4 u: panel half ia = tl & 1 = wv & 1, trailing 32-block ib = tl >> 1
			const int ia = wv & 1;
#pragma unroll
			for (int s = 0; s < 8; ++s) {
				const int roff = 8 * s + 4 * (lane >> 5);
				const f32x4 av = *reinterpret_cast<const f32x4 *>(&sm[(32 * ia + (lane & 31)) * TQ_LP + roff]);
#pragma unroll
				for (int u = 0; u < 3; ++u) {
					const int tl = wv + 4 * u;
					if (tl < ntile) {
						const int ib = tl >> 1;
						const f32x4 bv =
							*reinterpret_cast<const f32x4 *>(&sm[(TQ_PW + 32 * ib + (lane & 31)) * TQ_LP + roff]);
#pragma unroll
						for (int q = 0; q < 4; ++q)
							cacc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], cacc[u], 0, 0, 0);
					}
				}
			}
		}
	}
	const long blk = blockIdx.x;
	if (a.want_g) {
		// f64 16x16x4 result map: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
		for (int ib = 0; ib < 4; ++ib)
#pragma unroll
			for (int r = 0; r < 4; ++r)
				a.Gp[blk * 4096 + (16 * wv + (lane >> 4) + 4 * r) * 64 + 16 * ib + (lane & 15)] = gacc[ib][r];
	}
#pragma unroll
	for (int u = 0; u < 3; ++u) {
		const int tl = wv + 4 * u;
		if (tl < ntile) {
			const int ia = tl & 1, ib = tl >> 1;
			// f32 32x32x2 result map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
				a.Cp[blk * 64 * a.tp + (long) (32 * ia + i) * a.tp + 32 * ib + (lane & 31)] = cacc[u][r];
			}
		}
	}
	if (a.want_sq) {
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			float v = sq[i];
			v += __shfl_xor(v, 1);
			v += __shfl_xor(v, 2);
			v += __shfl_xor(v, 4);
			v += __shfl_xor(v, 8);
			if (quad == 0)
				a.Sp[blk * 256 + i * 16 + cg] = v;
		}
	}
}

// fixed-order sums of the per-workgroup partials: G (4096 entries), C (64 x tp -> C[a * ldc + coff + b]), column squares
__global__ __launch_bounds__(256) void tq_reduce_kernel(const double *Gp, const float *Cp, const float *Sp, int nb, int tp, int want_g,
							 int want_sq, double *G, double *C, int ldc, int coff, double *S, const int *stat)
{
	if (stat[0])
		return;
	const int e = blockIdx.x * 256 + threadIdx.x;
	const int ng = want_g ? 4096 : 0, nc = 64 * tp, ns = want_sq ? 256 : 0;
	if (e < ng) {
		double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
		int b = 0;
		for (; b + 4 <= nb; b += 4) {
			s0 += Gp[(long) b * 4096 + e];
			s1 += Gp[(long) (b + 1) * 4096 + e];
			s2 += Gp[(long) (b + 2) * 4096 + e];
			s3 += Gp[(long) (b + 3) * 4096 + e];
		}
		for (; b < nb; ++b)
			s0 += Gp[(long) b * 4096 + e];
		G[e] = (s0 + s1) + (s2 + s3);
	} else if (e < ng + nc) {
		const int idx = e - ng;
		double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
		int b = 0;
		for (; b + 4 <= nb; b += 4) {
			s0 += (double) Cp[(long) b * nc + idx];
			s1 += (double) Cp[(long) (b + 1) * nc + idx];
			s2 += (double) Cp[(long) (b + 2) * nc + idx];
			s3 += (double) Cp[(long) (b + 3) * nc + idx];
		}
		for (; b < nb; ++b)
			s0 += (double) Cp[(long) b * nc + idx];
		const int arow = idx / tp, bcol = idx - arow * tp;
		C[(long) arow * ldc + coff + bcol] = (s0 + s1) + (s2 + s3);
	} else if (e < ng + nc + ns) {
		const int idx = e - ng - nc;
		double s = 0;
		for (int b = 0; b < nb; ++b)
			s += (double) Sp[(long) b * 256 + idx];
		S[idx] = s;
	}
}

// ------------------------------------------------------------------------------------------------
// panel: everything 64 x 64, fp64, one workgroup
// ------------------------------------------------------------------------------------------------
struct TqPanelArgs {
	float *A;
	long ld;
	int m, r0, c0, w, n;
	const double *G;   // reduced Gram matrix, row major 64 x 64 (w x w valid)
	const double *S;   // column squares of the first launch (checked when check_range != 0): [0, 64) panel, [64, ..) trailing
	int check_range, range_cols; // number of staged trailing columns covered by S
	double *abv;	   // per global column: sum of squares of the R entries above the current block row
	double *N1, *N2;   // out: R^-T, R^-1 U^-1 V1^-1 (row major 64 x 64)
	float *Mn;	   // out: M = -(U R)^-1, row major 64 x 64
	float *H;
	long hrs, hcs;
	int bs;
	float *taus;
	int *stat;
};

__global__ __launch_bounds__(256) void tq_panel_kernel(const TqPanelArgs a)
{
	__shared__ double Gs[64 * TQ_DP]; // G -> L (lower Cholesky factor, R~ = L^T) -> M
	__shared__ double Ws[64 * TQ_DP]; // A1 -> Q1~ -> [V1 strictly lower | U upper]
	__shared__ double Ri[64 * TQ_DP]; // R~^-1 (upper)
	__shared__ double UL[64 * TQ_DP]; // U^-1 (upper incl. diagonal) | V1^-1 (strictly lower, unit diagonal implied)
	__shared__ double sgn[64];
	__shared__ double dg[64];
	__shared__ int s_fail;
	if (a.stat[0])
		return;
	const int tid = threadIdx.x, w = a.w;
	const int ti = tid & 63, tg = tid >> 6;
	if (tid == 0)
		s_fail = 0;
	for (int e = tid; e < 4096; e += 256) {
		const int i = e >> 6, j = e & 63;
		double g = (i == j) ? 1.0 : 0.0, x = 0.0;
		if (i < w && j < w) {
			g = a.G[e];
			x = (double) a.A[(long) (a.c0 + j) * a.ld + a.r0 + i];
		}
		Gs[i * TQ_DP + j] = g;
		Ws[i * TQ_DP + j] = x;
		Ri[i * TQ_DP + j] = 0.0;
		UL[i * TQ_DP + j] = 0.0;
	}
	__syncthreads();
	if (a.check_range) {
		// fp32 products of the C sums underflow / overflow for columns far from unit scale: rms outside [1e-12, 1e12]
		const double lo = 1e-24 * (double) (a.m - a.r0), hi = 1e24 * (double) (a.m - a.r0);
		bool bad = false;
		for (int c = tid; c < 64 + a.range_cols; c += 256) {
			if (c < 64 && c >= w)
				continue;
			const double s = a.S[c];
			bad = bad || !(s >= lo && s <= hi);
		}
		if (bad)
			s_fail = TQ_FAIL_RANGE;
		__syncthreads();
		if (s_fail) {
			if (tid == 0) {
				a.stat[1] = a.c0;
				a.stat[2] = s_fail;
				a.stat[0] = 1;
			}
			return;
		}
	}
	// ---- Cholesky, right looking, in place in the lower triangle; the roots of the pivots go to dg[] first
	for (int j = 0; j < 64; ++j) {
		__syncthreads(); // the update of step j - 1 is complete
		const double d = Gs[j * TQ_DP + j];
		if (!(d > 0.0) || !(d < 1e300)) {
			if (tid == 0) {
				a.stat[1] = a.c0;
				a.stat[2] = TQ_FAIL_CHOL;
				a.stat[0] = 1;
			}
			return; // uniform: every thread read the same pivot
		}
		const double rd = sqrt(d);
		if (tid == 0)
			dg[j] = rd;
		if (tid > j && tid < 64)
			Gs[tid * TQ_DP + j] /= rd;
		__syncthreads();
		if (ti > j) {
			const double lij = Gs[ti * TQ_DP + j];
			for (int c = j + 1 + tg; c <= ti; c += 4)
				Gs[ti * TQ_DP + c] -= lij * Gs[c * TQ_DP + j];
		}
	}
	__syncthreads();
	if (tid < 64)
		Gs[tid * TQ_DP + tid] = dg[tid];
	__syncthreads();
	// ---- Ri = R~^-1 (R~[i][l] = Gs[l][i]); thread c solves R~ x = e_c, uniform loops
	if (tid < 64) {
		const int c = tid;
		for (int i = 63; i >= 0; --i) {
			double acc = (i == c) ? 1.0 : 0.0;
			for (int l = i + 1; l < 64; ++l)
				acc -= Gs[l * TQ_DP + i] * Ri[l * TQ_DP + c];
			Ri[i * TQ_DP + c] = acc / Gs[i * TQ_DP + i];
		}
	}
	__syncthreads();
	// ---- Q1~ = A1 Ri
	{
		double wv[16];
#pragma unroll
		for (int u = 0; u < 16; ++u) {
			const int j = tg + 4 * u;
			double acc = 0.0;
			for (int l = 0; l <= j; ++l)
				acc += Ws[ti * TQ_DP + l] * Ri[l * TQ_DP + j];
			wv[u] = acc;
		}
		__syncthreads();
#pragma unroll
		for (int u = 0; u < 16; ++u)
			Ws[ti * TQ_DP + tg + 4 * u] = wv[u];
	}
	__syncthreads();
	// ---- sign-choosing LU of I - Q1~ S on the linear part (column j of I - W S is e_j - s_j W_j)
	for (int j = 0; j < 64; ++j) {
		__syncthreads(); // the update of step j - 1 is complete
		const double alpha = Ws[j * TQ_DP + j];
		const double sj = alpha >= 0.0 ? -1.0 : 1.0;
		const double piv = 1.0 + fabs(alpha);
		if (j < w && !(1.0 - fabs(alpha) >= TQ_TAIL_MIN)) {
			if (tid == 0) {
				a.stat[1] = a.c0;
				a.stat[2] = TQ_FAIL_TAIL;
				a.stat[0] = 1;
			}
			return; // uniform
		}
		if (tid == 0)
			sgn[j] = sj;
		if (tid > j && tid < 64)
			Ws[tid * TQ_DP + j] = -sj * Ws[tid * TQ_DP + j] / piv;
		__syncthreads();
		if (ti > j) {
			const double lij = Ws[ti * TQ_DP + j];
			for (int c = j + 1 + tg; c < 64; c += 4)
				Ws[ti * TQ_DP + c] -= lij * Ws[j * TQ_DP + c];
		}
	}
	__syncthreads();
	for (int e = tid; e < 4096; e += 256) {
		const int i = e >> 6, j = e & 63;
		if (j >= i)
			Ws[i * TQ_DP + j] = (i == j ? 1.0 : 0.0) - sgn[j] * Ws[i * TQ_DP + j];
	}
	__syncthreads();
	// ---- wave 0: U^-1 (upper part of UL); wave 1: V1^-1 (strictly lower part of UL); wave 2: the reference's rank test;
	//      wave 3: condition estimate.  Uniform loops, predicated reads: the two substitutions never touch each other's half.
	if (tg == 0) {
		const int c = ti;
		for (int i = 63; i >= 0; --i) {
			double acc = (i == c) ? 1.0 : 0.0;
			for (int l = i + 1; l < 64; ++l) {
				const double x = l <= c ? UL[l * TQ_DP + c] : 0.0;
				acc -= Ws[i * TQ_DP + l] * x;
			}
			if (i <= c)
				UL[i * TQ_DP + c] = acc / Ws[i * TQ_DP + i];
		}
	} else if (tg == 1) {
		// column c of V1^-1: x_c = 1 (implied), x_i = -V1[i][c] - sum_{c < l < i} V1[i][l] x_l
		const int c = ti;
		for (int i = 1; i < 64; ++i) {
			double acc = -Ws[i * TQ_DP + c];
			for (int l = 1; l < i; ++l) {
				const double x = l > c ? UL[l * TQ_DP + c] : 0.0;
				acc -= Ws[i * TQ_DP + l] * x;
			}
			if (i > c)
				UL[i * TQ_DP + c] = acc;
		}
	} else if (tg == 2) {
		// factor.rs:52-64: |R_jj| > eps * 16 * (m - row) * hypot(|R_jj|, |R[0 .. j, j]|)
		const int j = ti;
		if (j < w) {
			double ab = a.abv[a.c0 + j];
			for (int l = 0; l < j; ++l)
				ab += Gs[j * TQ_DP + l] * Gs[j * TQ_DP + l];
			const double rjj = Gs[j * TQ_DP + j];
			const double full = sqrt(rjj * rjj + ab);
			const double thr = (double) 1.1920928955078125e-07f * 16.0 * (double) (a.m - a.c0 - j) * full;
			if (!(rjj > thr))
				atomicMax(&s_fail, (int) TQ_FAIL_RANK);
		}
	} else {
		double f1 = 0.0, f2 = 0.0;
		for (int l = 0; l < 64; ++l) {
			f1 += l <= ti ? Gs[ti * TQ_DP + l] * Gs[ti * TQ_DP + l] : 0.0;
			f2 += Ri[l * TQ_DP + ti] * Ri[l * TQ_DP + ti];
		}
		for (int o = 32; o > 0; o >>= 1) {
			f1 += __shfl_xor(f1, o);
			f2 += __shfl_xor(f2, o);
		}
		if (!(sqrt(f1 * f2) <= TQ_COND_MAX))
			atomicMax(&s_fail, (int) TQ_FAIL_COND);
	}
	__syncthreads();
	if (s_fail) {
		if (tid == 0) {
			a.stat[1] = a.c0;
			a.stat[2] = s_fail;
			a.stat[0] = 1;
		}
		return;
	}
	// ---- outputs that read L: the top block of A (R = S R~ on and above the diagonal, V1 below), N1 = R^-T
	for (int e = tid; e < 4096; e += 256) {
		const int i = e >> 6, j = e & 63;
		if (i < w && j < w)
			a.A[(long) (a.c0 + j) * a.ld + a.r0 + i] = (float) (i <= j ? sgn[i] * Gs[j * TQ_DP + i] : Ws[i * TQ_DP + j]);
		// R^-1 = Ri S  =>  N1[i][l] = R^-1[l][i] = Ri[l][i] * s_i
		a.N1[e] = Ri[j * TQ_DP + i] * sgn[i];
	}
	// ---- M = -Ri S U^-1 (upper) into registers, then into Gs
	{
		double mv[16];
#pragma unroll
		for (int u = 0; u < 16; ++u) {
			const int j = tg + 4 * u, k = ti;
			double acc = 0.0;
			for (int l = k; l <= j; ++l)
				acc += Ri[k * TQ_DP + l] * sgn[l] * UL[l * TQ_DP + j];
			mv[u] = k <= j ? -acc : 0.0;
		}
		__syncthreads(); // all reads of L are done
#pragma unroll
		for (int u = 0; u < 16; ++u) {
			const int j = tg + 4 * u;
			Gs[ti * TQ_DP + j] = mv[u];
			a.Mn[ti * 64 + j] = (float) mv[u];
		}
	}
	__syncthreads();
	// ---- N2 = -M V1^-1,  T = triu(V1^T U^-1)
	for (int u = 0; u < 16; ++u) {
		const int j = tg + 4 * u, k = ti;
		// V1^-1[l][j]: 1 on the diagonal, UL[l][j] for l > j, 0 above
		double acc = Gs[k * TQ_DP + j];
		for (int l = j + 1; l < 64; ++l)
			acc += Gs[k * TQ_DP + l] * UL[l * TQ_DP + j];
		a.N2[k * 64 + j] = -acc;
		if (k <= j && j < w) {
			double tt = UL[k * TQ_DP + j]; // l = k term: V1[k][k] = 1
			for (int l = k + 1; l <= j; ++l)
				tt += Ws[l * TQ_DP + k] * UL[l * TQ_DP + j];
			const int gi = a.c0 + k, gj = a.c0 + j;
			if (gi / a.bs == gj / a.bs)
				a.H[(long) (gi % a.bs) * a.hrs + (long) gj * a.hcs] = (float) tt;
			if (k == j)
				a.taus[gj] = (float) tt;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// y: 16 trailing columns per workgroup
// ------------------------------------------------------------------------------------------------
struct TqYArgs {
	float *A;
	long ld;
	int r0, cx, w, t;
	const double *C; // reduced, row major 64 x ldc
	int ldc;
	const double *N1, *N2;
	double *abv;
	float *Yn; // out: -Y, row major 64 x typ
	int typ;
	const int *stat;
};

__global__ __launch_bounds__(256) void tq_y_kernel(const TqYArgs a)
{
	__shared__ double n1[64 * TQ_DP], n2[64 * TQ_DP];
	__shared__ double v[64 * 17];
	if (a.stat[0])
		return;
	const int tid = threadIdx.x;
	for (int e = tid; e < 4096; e += 256) {
		n1[(e >> 6) * TQ_DP + (e & 63)] = a.N1[e];
		n2[(e >> 6) * TQ_DP + (e & 63)] = a.N2[e];
	}
	const int bl = tid & 15, ig = tid >> 4; // column, row group: rows ig, ig + 16, ig + 32, ig + 48
	const int b = blockIdx.x * 16 + bl;
	const bool colok = b < a.t;
	for (int u = 0; u < 4; ++u) {
		const int i = ig + 16 * u;
		v[i * 17 + bl] = colok ? a.C[(long) i * a.ldc + b] : 0.0;
	}
	__syncthreads();
	double d[4], e4[4];
#pragma unroll
	for (int u = 0; u < 4; ++u) {
		const int i = ig + 16 * u;
		double acc = 0.0;
		for (int l = 0; l <= i; ++l)
			acc += n1[i * TQ_DP + l] * v[l * 17 + bl];
		d[u] = acc;
		float *xp = a.A + (long) (a.cx + b) * a.ld + a.r0 + i;
		double xt = 0.0;
		if (colok && i < a.w) {
			xt = (double) *xp;
			*xp = (float) acc;
		}
		e4[u] = acc - xt;
	}
	__syncthreads();
#pragma unroll
	for (int u = 0; u < 4; ++u)
		v[(ig + 16 * u) * 17 + bl] = e4[u];
	// column norms of the R rows just produced (rank test of the later panels)
	double ss = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
	__shared__ double sred[256];
	sred[tid] = ss;
	__syncthreads();
	if (ig == 0 && colok) {
		double s = 0.0;
		for (int g = 0; g < 16; ++g)
			s += sred[g * 16 + bl];
		a.abv[a.cx + b] += s;
	}
#pragma unroll
	for (int u = 0; u < 4; ++u) {
		const int k = ig + 16 * u;
		double acc = 0.0;
		for (int l = 0; l < 64; ++l)
			acc += n2[k * TQ_DP + l] * v[l * 17 + bl];
		if (b < a.typ)
			a.Yn[(long) k * a.typ + b] = colok ? (float) -acc : 0.f;
	}
}

// ------------------------------------------------------------------------------------------------
// update: X <- X - P Y (columns [coff, coff + ts) of the trailing block) and, if do_v, V = P M over the panel
// ------------------------------------------------------------------------------------------------
struct TqUpdArgs {
	float *P; // A[r1, c0], r1 = first row below the top block
	float *X; // A[r1, cx + coff]
	long ld;
	int rows, w, ts; // rows from r1 down; strip width (<= 192)
	const float *Yn;
	int typ, coff;
	const float *Mn;
	int do_v;
	int nrb; // 128-row blocks
	const int *stat;
};

template <bool VEC> static __device__ __forceinline__ f32x4 tq_ld4(const float *col, int lam, int rows)
{
	f32x4 v = {0.f, 0.f, 0.f, 0.f};
	if (VEC) {
		const int r = 4 * lam;
		if (r + 3 < rows) {
			v = *reinterpret_cast<const f32x4 *>(col + r);
		} else {
#pragma unroll
			for (int e = 0; e < 4; ++e)
				if (r + e < rows)
					v[e] = col[r + e];
		}
	} else {
#pragma unroll
		for (int q = 0; q < 4; ++q)
			if (32 * q + lam < rows)
				v[q] = col[32 * q + lam];
	}
	return v;
}
template <bool VEC> static __device__ __forceinline__ void tq_st4(float *col, int lam, int rows, f32x4 v)
{
	if (VEC) {
		const int r = 4 * lam;
		if (r + 3 < rows) {
			*reinterpret_cast<f32x4 *>(col + r) = v;
		} else {
#pragma unroll
			for (int e = 0; e < 4; ++e)
				if (r + e < rows)
					col[r + e] = v[e];
		}
	} else {
#pragma unroll
		for (int q = 0; q < 4; ++q)
			if (32 * q + lam < rows)
				col[32 * q + lam] = v[q];
	}
}

// One wavefront owns 128 rows: its 128 x 64 block of the panel is loaded ONCE into registers (32 quads per lane: lane
// l & 31 holds rows 4 (l & 31) .. + 3 of column 2 i + (l >> 5) -- row slot q of the four interleaved 32-row tiles), then
// 32-column strips of X stream through the accumulators: out(rows, j) = in(rows, j) + sum_k P(rows, k) Ys[k][j], and last
// the two strips of V = P M, stored over the panel rows the wave has in registers.
template <bool VEC> __global__ __launch_bounds__(256, 1) void tq_update_kernel(const TqUpdArgs a)
{
	constexpr int PITCH = TQ_TS + 64;
	__shared__ float Ys[64 * PITCH];
	if (a.stat[0])
		return;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	for (int e = tid; e < 64 * PITCH; e += 256) {
		const int k = e / PITCH, c = e - k * PITCH;
		float v = 0.f;
		if (c < a.ts)
			v = a.Yn[(long) k * a.typ + a.coff + c];
		else if (c >= TQ_TS && a.do_v)
			v = a.Mn[k * 64 + (c - TQ_TS)];
		Ys[e] = v;
	}
	__syncthreads();
	const int lam = lane & 31, h = lane >> 5;
	const int nx = (a.ts + 31) >> 5;	   // 32-column strips of X
	const int nv = a.do_v ? (a.w + 31) >> 5 : 0; // strips of V
	for (int rb = blockIdx.x * 4 + wv; rb < a.nrb; rb += gridDim.x * 4) {
		const int rows = a.rows - rb * 128; // valid rows from the block's first row (may exceed 128)
		const float *Pb = a.P + (long) rb * 128;
		float *Xb = a.X + (long) rb * 128;
		f32x4 pr[32];
#pragma unroll
		for (int i = 0; i < 32; ++i) {
			const int kc = 2 * i + h;
			pr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
			if (kc < a.w)
				pr[i] = tq_ld4<VEC>(Pb + (long) kc * a.ld, lam, rows);
		}
		// one wavefront per SIMD (512 registers): the next strip of X is fetched while the matrix cores work on this one
		f32x4 xn[16];
#pragma unroll
		for (int r = 0; r < 16; ++r) {
			const int j = (r & 3) + 8 * (r >> 2) + 4 * h;
			xn[r] = f32x4{0.f, 0.f, 0.f, 0.f};
			if (nx > 0 && j < a.ts)
				xn[r] = tq_ld4<VEC>(Xb + (long) j * a.ld, lam, rows);
		}
#pragma unroll 1
		for (int st = 0; st < nx + nv; ++st) {
			const bool isv = st >= nx;
			const int sc = isv ? st - nx : st;
			const int lcol = (isv ? TQ_TS : 0) + 32 * sc;
			const int nvalid = (isv ? a.w : a.ts) - 32 * sc;
			float *dst = (isv ? a.P + (long) rb * 128 : Xb) + (long) (32 * sc) * a.ld;
			f32x16 acc[4];
#pragma unroll
			for (int r = 0; r < 16; ++r)
#pragma unroll
				for (int q = 0; q < 4; ++q)
					acc[q][r] = isv ? 0.f : xn[r][q];
			if (st + 1 < nx) {
				const float *nxt = Xb + (long) (32 * (st + 1)) * a.ld;
				const int nval = a.ts - 32 * (st + 1);
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					const int j = (r & 3) + 8 * (r >> 2) + 4 * h;
					xn[r] = f32x4{0.f, 0.f, 0.f, 0.f};
					if (j < nval)
						xn[r] = tq_ld4<VEC>(nxt + (long) j * a.ld, lam, rows);
				}
			}
#pragma unroll
			for (int i = 0; i < 32; ++i) {
				const float a0 = Ys[(2 * i + h) * PITCH + lcol + lam];
#pragma unroll
				for (int q = 0; q < 4; ++q)
					acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, pr[i][q], acc[q], 0, 0, 0);
			}
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				const int j = (r & 3) + 8 * (r >> 2) + 4 * h;
				if (j < nvalid)
					tq_st4<VEC>(dst + (long) j * a.ld, lam, rows, f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]});
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------
// cross-panel blocks of T inside one block of Q_coeff: T[k-panel cols, l-panel cols] = V_k^T V_l, k < l
// C (reduced) = V_l^T [V_first .. V_{l-1}] over the rows below panel l's top block; the top block's rows are added here
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tq_tcross_kernel(const float *A, long ld, int m, int cfirst, int cl, int wl, const double *C, int ldc,
							 float *H, long hrs, long hcs, int bs, const int *stat)
{
	if (stat[0])
		return;
	const int e = blockIdx.x * 256 + threadIdx.x;
	const int nk = cl - cfirst; // columns of the earlier panels
	if (e >= nk * wl)
		return;
	const int i = e % nk, j = e / nk; // i: column cfirst + i of an earlier panel, j: column cl + j of panel l
	double acc = C[(long) j * ldc + i];
	// rows cl .. cl + wl - 1: V_l's top block is unit lower triangular
	for (int r = j; r < wl && cl + r < m; ++r) {
		const double vl = r == j ? 1.0 : (double) A[(long) (cl + j) * ld + cl + r];
		acc += (double) A[(long) (cfirst + i) * ld + cl + r] * vl;
	}
	const int gi = cfirst + i, gj = cl + j;
	H[(long) (gi % bs) * hrs + (long) gj * hcs] = (float) acc;
}

__global__ void tq_zero_kernel(double *p, int n)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i < n)
		p[i] = 0.0;
}

// ------------------------------------------------------------------------------------------------
// driver
// ------------------------------------------------------------------------------------------------
static void tq_gram(const float *P, const float *X, long ld, int rows, int w, int t, bool want_g, bool want_sq, bool vec, double *Gp, float *Cp,
		    float *Sp, double *G, double *C, int ldc, int coff, double *S, const int *stat)
{
	hipStream_t s = ctx().stream;
	TqGramArgs g;
	g.P = P;
	g.X = X;
	g.ld = ld;
	g.rows = rows;
	g.w = w;
	g.t = t;
	g.tp = (t + 31) & ~31;
	g.nchunks = (rows + 63) / 64;
	g.want_g = want_g;
	g.want_sq = want_sq;
	g.Gp = Gp;
	g.Cp = Cp;
	g.Sp = Sp;
	g.stat = stat;
	const int nb = g.nchunks < TQ_NB ? g.nchunks : TQ_NB;
	if (nb <= 0)
		return;
	if (vec)
		hipLaunchKernelGGL(tq_gram_kernel<true>, dim3(nb), dim3(256), 0, s, g);
	else
		hipLaunchKernelGGL(tq_gram_kernel<false>, dim3(nb), dim3(256), 0, s, g);
	const int total = (want_g ? 4096 : 0) + 64 * g.tp + (want_sq ? 256 : 0);
	hipLaunchKernelGGL(tq_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, Gp, Cp, Sp, nb, g.tp, (int) want_g, (int) want_sq, G, C, ldc,
			   coff, S, stat);
	FH_HIP(hipGetLastError());
}

bool tsqr_applicable(idx_t m, idx_t n, idx_t rs, idx_t cs, idx_t bs)
{
	static const bool off = getenv("FAER_HIP_QR_TSQR") && atoi(getenv("FAER_HIP_QR_TSQR")) == 0; // A/B switch
	if (off)
		return false;
	if (rs != 1 || cs < m || n < 1 || n > 512 || m < 16384 || m < 8 * n || m >= (1L << 30))
		return false;
	// T blocks are written per 64-column panel: a block of Q_coeff is either a whole number of panels or divides one
	return bs % TQ_PW == 0 || TQ_PW % bs == 0;
}

// Factors the leading panels of A (m x n fp32, column major) on the one-pass path.  Returns the number of COLUMNS
// completed (a multiple of 64, or n); the state is then that of the reference algorithm after those columns: R and
// V in place, the T blocks in H, taus[j] = T_jj, every reflector applied to all columns on the right.
// `reason` reports why it stopped early (TQ_FAIL_*).
idx_t tsqr_factor(MatV<float> A, MatV<float> H, float *taus, int *reason)
{
	const idx_t m = A.nrows, n = A.ncols, ld = A.cs, bs = H.nrows;
	hipStream_t s = ctx().stream;
	const bool vec = (ld % 4 == 0) && ((uintptr_t) A.p % 16 == 0);
	const int npan = (int) ((n + TQ_PW - 1) / TQ_PW);
	const int tmax = (int) n; // trailing width bound
	const int ldc = (tmax + 63) & ~63;
	const int typ = ldc;
	Scratch gp((size_t) TQ_NB * 4096 * 8), cp((size_t) TQ_NB * 64 * TQ_TS * 4), sp((size_t) TQ_NB * 256 * 4);
	Scratch small((size_t) (4096 * 3 + 64 * ldc + 512 + 2 * n + 64) * 8 + (size_t) (4096 + 64 * typ) * 4 + 256);
	double *G = small.as<double>();
	double *N1 = G + 4096, *N2 = N1 + 4096, *C = N2 + 4096, *S = C + (size_t) 64 * ldc, *abv = S + 512;
	float *Mn = reinterpret_cast<float *>(abv + 2 * n + 64);
	float *Yn = Mn + 4096;
	int *stat = reinterpret_cast<int *>(Yn + (size_t) 64 * typ);
	FH_HIP(hipMemsetAsync(stat, 0, 64, s));
	FH_HIP(hipMemsetAsync(abv, 0, (size_t) (2 * n + 64) * 8, s));
	auto launch_gram = [&](int c0, int w, bool first) {
		// G of panel [c0, c0 + w) and C against everything right of it, rows from c0 down, strips of <= 192 columns
		const int t = (int) n - c0 - w;
		const float *P = A.p + (long) c0 * ld + c0;
		const int rows = (int) (m - c0);
		if (t == 0) {
			tq_gram(P, P, ld, rows, w, 0, true, first, vec, gp.as<double>(), cp.as<float>(), sp.as<float>(), G, C, ldc, 0, S, stat);
			return;
		}
		for (int off = 0; off < t; off += TQ_TS) {
			const int ts = t - off < TQ_TS ? t - off : TQ_TS;
			// the range guard covers the first strip only (n <= 256); wider matrices check the rest per panel through G
			tq_gram(P, A.p + (long) (c0 + w + off) * ld + c0, ld, rows, w, ts, off == 0, first && off == 0, vec, gp.as<double>(),
				cp.as<float>(), sp.as<float>(), G, C, ldc, off, S, stat);
		}
	};
	launch_gram(0, (int) (n < TQ_PW ? n : TQ_PW), true);
	for (int k = 0; k < npan; ++k) {
		const int c0 = k * TQ_PW;
		const int w = (int) (n - c0 < TQ_PW ? n - c0 : TQ_PW);
		const int t = (int) n - c0 - w;
		TqPanelArgs pa;
		pa.A = A.p;
		pa.ld = ld;
		pa.m = (int) m;
		pa.r0 = c0;
		pa.c0 = c0;
		pa.w = w;
		pa.n = (int) n;
		pa.G = G;
		pa.S = S;
		pa.check_range = k == 0;
		pa.range_cols = t < TQ_TS ? t : TQ_TS;
		pa.abv = abv;
		pa.N1 = N1;
		pa.N2 = N2;
		pa.Mn = Mn;
		pa.H = H.p;
		pa.hrs = H.rs;
		pa.hcs = H.cs;
		pa.bs = (int) bs;
		pa.taus = taus;
		pa.stat = stat;
		hipLaunchKernelGGL(tq_panel_kernel, dim3(1), dim3(256), 0, s, pa);
		if (t > 0) {
			TqYArgs ya;
			ya.A = A.p;
			ya.ld = ld;
			ya.r0 = c0;
			ya.cx = c0 + w;
			ya.w = w;
			ya.t = t;
			ya.C = C;
			ya.ldc = ldc;
			ya.N1 = N1;
			ya.N2 = N2;
			ya.abv = abv;
			ya.Yn = Yn;
			ya.typ = typ;
			ya.stat = stat;
			hipLaunchKernelGGL(tq_y_kernel, dim3((t + 15) / 16), dim3(256), 0, s, ya);
		}
		const int r1 = c0 + w;
		const int rows = (int) (m - r1);
		if (rows > 0) {
			TqUpdArgs ua;
			ua.ld = ld;
			ua.rows = rows;
			ua.w = w;
			ua.Yn = Yn;
			ua.typ = typ;
			ua.Mn = Mn;
			ua.nrb = (rows + 127) / 128;
			ua.stat = stat;
			ua.P = A.p + (long) c0 * ld + r1;
			int nwg = (ua.nrb + 3) / 4;
			if (nwg > TQ_NB)
				nwg = TQ_NB;
			const int nstrip = (t + TQ_TS - 1) / TQ_TS;
			for (int st = 0; st < nstrip || (st == 0 && nstrip == 0); ++st) {
				ua.coff = st * TQ_TS;
				ua.ts = nstrip == 0 ? 0 : (t - ua.coff < TQ_TS ? t - ua.coff : TQ_TS);
				ua.X = A.p + (long) (c0 + w + ua.coff) * ld + r1;
				ua.do_v = nstrip <= 1; // V overwrites P: only when no other launch still reads the panel
				if (vec && r1 % 4 == 0)
					hipLaunchKernelGGL(tq_update_kernel<true>, dim3(nwg), dim3(256), 0, s, ua);
				else
					hipLaunchKernelGGL(tq_update_kernel<false>, dim3(nwg), dim3(256), 0, s, ua);
			}
			if (nstrip > 1) {
				ua.coff = 0;
				ua.ts = 0;
				ua.X = ua.P;
				ua.do_v = 1;
				if (vec && r1 % 4 == 0)
					hipLaunchKernelGGL(tq_update_kernel<true>, dim3(nwg), dim3(256), 0, s, ua);
				else
					hipLaunchKernelGGL(tq_update_kernel<false>, dim3(nwg), dim3(256), 0, s, ua);
			}
		}
		if (t > 0) {
			const int wn = t < TQ_PW ? t : TQ_PW;
			launch_gram(c0 + w, wn, false);
		}
		FH_HIP(hipGetLastError());
	}
	// cross-panel blocks of T (blocks of Q_coeff wider than one panel)
	if (bs > TQ_PW) {
		for (int l = 1; l < npan; ++l) {
			const int cl = l * TQ_PW;
			const int cfirst = (int) ((cl / bs) * bs);
			if (cfirst == cl)
				continue; // panel l starts a block of Q_coeff
			const int wl = (int) (n - cl < TQ_PW ? n - cl : TQ_PW);
			const int nk = cl - cfirst;
			const int r1 = cl + wl;
			const int rows = (int) (m - r1);
			for (int off = 0; off < nk; off += TQ_TS) {
				const int ts = nk - off < TQ_TS ? nk - off : TQ_TS;
				if (rows > 0)
					tq_gram(A.p + (long) cl * ld + r1, A.p + (long) (cfirst + off) * ld + r1, ld, rows, wl, ts, false, false, vec && r1 % 4 == 0,
						gp.as<double>(), cp.as<float>(), sp.as<float>(), G, C, ldc, off, S, stat);
				else
					hipLaunchKernelGGL(tq_zero_kernel, dim3((64 * ldc + 255) / 256), dim3(256), 0, s, C, 64 * ldc);
			}
			hipLaunchKernelGGL(tq_tcross_kernel, dim3((nk * wl + 255) / 256), dim3(256), 0, s, A.p, ld, (int) m, cfirst, cl, wl, C, ldc, H.p,
					   H.rs, H.cs, (int) bs, stat);
		}
		FH_HIP(hipGetLastError());
	}
	int st[4];
	FH_HIP(hipMemcpyAsync(st, stat, sizeof(st), hipMemcpyDeviceToHost, s));
	FH_HIP(hipStreamSynchronize(s));
	*reason = st[0] ? st[2] : TQ_OK;
	return st[0] ? (idx_t) st[1] : n;
}

} // namespace fh
