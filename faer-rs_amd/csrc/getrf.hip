// Partial-pivot LU  P A = L U  for gfx950.
//
// Replaces faer/src/linalg/lu/partial_pivoting/factor.rs:19-295.
//
// Driver: the reference's recursion (factor.rs:68-187) -- factor the left half, unit-lower TRSM on the
// top-right block, MFMA GEMM on the bottom-right block, factor the right half, then apply the row
// transpositions to the columns outside the half -- with the same split rule, so all O(n^3) work is a few
// large GEMMs.  The pivot rule is the reference's (factor.rs:35-43): first row of strictly largest
// |a_ij|, an all-zero column keeps the diagonal.
//
// Leaf (<= 32 columns, any height): ONE cooperative launch, "wavefront-level pivot reduction":
//   * the m x w panel is split in row chunks, one workgroup per chunk, each chunk RESIDENT IN LDS for
//     the whole leaf (read from HBM once, written once);
//   * per column: DPP/shuffle arg-max inside each wave, LDS across the 4 waves, then ONE all-to-all
//     exchange per column through L2: every workgroup publishes its candidate {|a|, row, the candidate's
//     whole panel row} and workgroup 0 publishes the diagonal row; after a single device-scope barrier every
//     workgroup picks the winner itself, patches the two swapped rows from the published copies and
//     performs scale + rank-1 update on its own rows.  No second synchronisation per column;
//   * the barrier follows the gfx950 recipe (MI355X_MICROARCH.md, "barrier-counter"): plain stores,
//     lane-0 agent release + s_waitcnt, relaxed monotonic counter, relaxed polling with s_sleep,
//     lane-0 agent acquire.  Every spin is bounded; a timeout raises a device error instead of hanging.
// Row interchanges of the outside columns are NOT applied one transposition at a time: the transposition
// list is composed into a net row permutation in parallel (each row traces its source backwards through
// the list) and applied as one gather.
#include <climits>

#include "common.h"

namespace fh {

constexpr int LU_W = 32; // leaf width

struct Cand {
	double v; // |a| (kept in double for both dtypes)
	int r;	  // global row, INT_MAX == none
};
static __device__ __forceinline__ bool better(double av, int ar, double bv, int br)
{
	return av > bv || (av == bv && ar < br);
}
static __device__ __forceinline__ void wave_argmax(double &v, int &r)
{
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		const double ov = __shfl_xor(v, off, 64);
		const int orow = __shfl_xor(r, off, 64);
		if (better(ov, orow, v, r)) {
			v = ov;
			r = orow;
		}
	}
}

// Device-scope barrier among the gridDim.x resident workgroups of a launch.  Returns false on timeout.
static __device__ bool grid_barrier(unsigned long long *cnt, unsigned long long target, int *s_flag)
{
	__syncthreads();
	if (threadIdx.x == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__hip_atomic_fetch_add(cnt, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		int ok = 0;
		for (int spin = 0; spin < (1 << 22); ++spin) {
			if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {
				ok = 1;
				break;
			}
			__builtin_amdgcn_s_sleep(1);
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		*s_flag = ok;
	}
	__syncthreads();
	return *s_flag != 0;
}

template <typename T> struct PanelArgs {
	T *P;
	idx_t rs, cs;
	int m, w;   // panel shape
	int R;	    // rows per workgroup
	int *piv;   // piv[j] = row_base + pivot row (absolute index in the top-level matrix)
	int row_base;
	double *slot_val; // [2][G][1 + LU_W]  (candidate |a| then the candidate's panel row)
	int *slot_row;	  // [2][G]
	double *diag_row; // [2][LU_W]
	unsigned long long *counter;
	unsigned long long counter_base;
	int *status; // status[2] = barrier timeout flag
};

template <typename T, int RMAX> __global__ __launch_bounds__(256) void getrf_panel_kernel(const PanelArgs<T> a)
{
	__shared__ T Ps[LU_W * RMAX]; // Ps[c * RMAX + r]
	__shared__ double s_v[4];
	__shared__ int s_r[4];
	__shared__ T s_piv[LU_W], s_diag[LU_W];
	__shared__ int s_p, s_gw, s_flag;

	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int g = blockIdx.x, G = gridDim.x;
	const int r0 = g * a.R;
	const int nr = min(a.R, a.m - r0);
	const int w = a.w;

	for (int c = 0; c < w; ++c)
		for (int r = tid; r < nr; r += 256)
			Ps[c * RMAX + r] = a.P[(idx_t) (r0 + r) * a.rs + (idx_t) c * a.cs];
	__syncthreads();

	const int steps = min(w, a.m);
	bool failed = false;
	for (int j = 0; j < steps; ++j) {
		const int q = j & 1;
		// ---- 1. local arg-max of |a(:, j)| over owned rows >= j (first strictly largest)
		double bv = 0.0;
		int br = INT_MAX;
		for (int r = tid; r < nr; r += 256) {
			const int gr = r0 + r;
			if (gr >= j) {
				const double av = fabs((double) Ps[j * RMAX + r]);
				if (av > bv) {
					bv = av;
					br = gr;
				}
			}
		}
		wave_argmax(bv, br);
		if (lane == 0) {
			s_v[wave] = bv;
			s_r[wave] = br;
		}
		__syncthreads();
		if (tid == 0) {
			double v = s_v[0];
			int r = s_r[0];
			for (int k = 1; k < 4; ++k)
				if (better(s_v[k], s_r[k], v, r)) {
					v = s_v[k];
					r = s_r[k];
				}
			if (!(v > 0.0))
				r = INT_MAX; // zero / NaN-only column chunk: no candidate
			s_p = r;
			s_v[0] = v;
		}
		__syncthreads();
		int p; // global pivot row
		if (G == 1) {
			p = s_p == INT_MAX ? j : s_p;
			if (tid < w) {
				s_piv[tid] = Ps[tid * RMAX + p];
				s_diag[tid] = Ps[tid * RMAX + j];
			}
			__syncthreads();
		} else {
			// ---- 2. publish candidate (+ its panel row) and, from workgroup 0, the diagonal row
			const int cand = s_p;
			double *sv = a.slot_val + ((size_t) q * G + g) * (1 + LU_W);
			if (tid == 0) {
				a.slot_row[q * G + g] = cand;
				sv[0] = s_v[0];
			}
			if (tid < w && cand != INT_MAX)
				sv[1 + tid] = (double) Ps[tid * RMAX + (cand - r0)];
			if (g == 0 && tid < w)
				a.diag_row[q * LU_W + tid] = (double) Ps[tid * RMAX + j]; // row j < 32 <= R lives in chunk 0
			// ---- 3. one device-scope barrier per column
			if (!grid_barrier(a.counter, a.counter_base + (unsigned long long) G * (j + 1), &s_flag)) {
				failed = true;
				break;
			}
			// ---- 4. every workgroup picks the winner itself
			double v = 0.0;
			int r = INT_MAX, gw = 0;
			for (int t = tid; t < G; t += 256) {
				const int rr = a.slot_row[q * G + t];
				const double vv = a.slot_val[((size_t) q * G + t) * (1 + LU_W)];
				if (rr != INT_MAX && better(vv, rr, v, r)) {
					v = vv;
					r = rr;
				}
			}
			wave_argmax(v, r);
			if (lane == 0) {
				s_v[wave] = v;
				s_r[wave] = r;
			}
			__syncthreads();
			if (tid == 0) {
				for (int k = 1; k < 4; ++k)
					if (better(s_v[k], s_r[k], v, r)) {
						v = s_v[k];
						r = s_r[k];
					}
				s_p = r == INT_MAX ? j : r;
				s_gw = r == INT_MAX ? 0 : r / a.R;
			}
			__syncthreads();
			p = s_p;
			gw = s_gw;
			if (tid < w) {
				const T d = (T) a.diag_row[q * LU_W + tid];
				s_diag[tid] = d;
				s_piv[tid] = p == j ? d : (T) a.slot_val[((size_t) q * G + gw) * (1 + LU_W) + 1 + tid];
			}
			__syncthreads();
		}
		// ---- 5. swap rows j <-> p inside the chunks that own them (from the published copies)
		if (p != j) {
			if (p >= r0 && p < r0 + nr && tid < w)
				Ps[tid * RMAX + (p - r0)] = s_diag[tid];
			if (g == 0 && tid < w)
				Ps[tid * RMAX + j] = s_piv[tid];
		}
		if (g == 0 && tid == 0)
			a.piv[j] = a.row_base + p;
		__syncthreads();
		// ---- 6. scale by the reciprocal pivot and rank-1 update of the owned rows below the diagonal
		//         (factor.rs:50-64; rank_update_imp: dst = fma(l_i, -u_c, dst))
		const T inv = (T) 1 / s_piv[j];
		for (int r = tid; r < nr; r += 256) {
			if (r0 + r > j) {
				const T l = Ps[j * RMAX + r] * inv;
				Ps[j * RMAX + r] = l;
				for (int c = j + 1; c < w; ++c)
					Ps[c * RMAX + r] = __builtin_fma(l, -s_piv[c], Ps[c * RMAX + r]);
			}
		}
		__syncthreads();
	}
	if (failed) {
		if (tid == 0)
			atomicExch(a.status + 2, 1);
		return;
	}
	for (int c = 0; c < w; ++c)
		for (int r = tid; r < nr; r += 256)
			a.P[(idx_t) (r0 + r) * a.rs + (idx_t) c * a.cs] = Ps[c * RMAX + r];
}

// ------------------------------------------------------------------------------------------------
// row interchanges as one gather
// ------------------------------------------------------------------------------------------------
// src[d] = row (relative to the block) whose content ends at row d after applying the transpositions
// (j <-> piv[j] - row_base), j = 0 .. nt-1, in order: trace d backwards through the list.
__global__ void compose_perm_kernel(const int *__restrict__ piv, int nt, int row_base, int nrows, int *src)
{
	const int d = blockIdx.x * blockDim.x + threadIdx.x;
	if (d >= nrows)
		return;
	int pos = d;
	for (int j = nt - 1; j >= 0; --j) {
		const int pj = piv[j] - row_base;
		if (pos == j)
			pos = pj;
		else if (pos == pj)
			pos = j;
	}
	src[d] = pos;
}
template <typename T>
__global__ void gather_rows_kernel(const T *B, idx_t rs, idx_t cs, int nrows, int ncols, const int *__restrict__ src, T *tmp)
{
	const int d = blockIdx.x * blockDim.x + threadIdx.x;
	const int c = blockIdx.y;
	if (d >= nrows)
		return;
	const int s = src[d];
	if (s != d)
		tmp[(size_t) c * nrows + d] = B[(idx_t) s * rs + (idx_t) c * cs];
}
template <typename T>
__global__ void scatter_rows_kernel(T *B, idx_t rs, idx_t cs, int nrows, int ncols, const int *__restrict__ src, const T *tmp)
{
	const int d = blockIdx.x * blockDim.x + threadIdx.x;
	const int c = blockIdx.y;
	if (d >= nrows)
		return;
	if (src[d] != d)
		B[(idx_t) d * rs + (idx_t) c * cs] = tmp[(size_t) c * nrows + d];
}

// Applies the transpositions (j <-> piv[j] - row_base), j < nt, to all columns of B (B's row 0 is the
// row the first transposition refers to).
template <typename T> static void laswp_dev(MatV<T> B, const int *piv, int nt, int row_base)
{
	if (B.nrows == 0 || B.ncols == 0 || nt == 0)
		return;
	const int nrows = (int) B.nrows;
	hipStream_t s = ctx().stream;
	Scratch srcb((size_t) nrows * sizeof(int));
	int *src = srcb.as<int>();
	hipLaunchKernelGGL(compose_perm_kernel, dim3((nrows + 255) / 256), dim3(256), 0, s, piv, nt, row_base, nrows, src);
	const idx_t chunk = 32768; // columns per pass (grid.y limit 65535, bounds the temp buffer too)
	const idx_t maxc = B.ncols < chunk ? B.ncols : chunk;
	Scratch tmpb((size_t) nrows * (size_t) maxc * sizeof(T));
	for (idx_t c0 = 0; c0 < B.ncols; c0 += chunk) {
		const idx_t nc = B.ncols - c0 < chunk ? B.ncols - c0 : chunk;
		MatV<T> S = B.sub(0, c0, B.nrows, nc);
		dim3 grid((nrows + 255) / 256, (unsigned) nc);
		hipLaunchKernelGGL(gather_rows_kernel<T>, grid, dim3(256), 0, s, S.p, S.rs, S.cs, nrows, (int) nc, src,
				   tmpb.as<T>());
		hipLaunchKernelGGL(scatter_rows_kernel<T>, grid, dim3(256), 0, s, S.p, S.rs, S.cs, nrows, (int) nc, src,
				   tmpb.as<T>());
	}
	FH_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// driver
// ------------------------------------------------------------------------------------------------
template <typename T> struct LuWork {
	int *piv;	    // device, min(m, n) entries, absolute rows
	double *slot_val;   // [2][GMAX][1 + LU_W]
	int *slot_row;	    // [2][GMAX]
	double *diag_row;   // [2][LU_W]
	unsigned long long *counter;
	unsigned long long counter_base;
	int *status;
};
constexpr int LU_GMAX = 224;

template <typename T> static void getrf_leaf(MatV<T> P, int col0, int row_base, LuWork<T> &wk)
{
	constexpr int RMAX = sizeof(T) == 8 ? 448 : 896;
	const idx_t m = P.nrows;
	const int w = (int) P.ncols;
	FH_CHECK(w <= LU_W, "getrf leaf: panel too wide");
	FH_CHECK(m <= (idx_t) RMAX * LU_GMAX, "partial_piv_lu: more rows than the cooperative panel kernel supports");
	int G = (int) ((m + RMAX - 1) / RMAX);
	if (G < 1)
		G = 1;
	int R = (int) ((m + G - 1) / G);
	R = (R + 63) / 64 * 64;
	if (R > RMAX)
		R = RMAX;
	if (R < 64)
		R = 64;
	G = (int) ((m + R - 1) / R);
	PanelArgs<T> a;
	a.P = P.p;
	a.rs = P.rs;
	a.cs = P.cs;
	a.m = (int) m;
	a.w = w;
	a.R = R;
	a.piv = wk.piv + col0;
	a.row_base = row_base;
	a.slot_val = wk.slot_val;
	a.slot_row = wk.slot_row;
	a.diag_row = wk.diag_row;
	a.counter = wk.counter;
	a.counter_base = wk.counter_base;
	a.status = wk.status;
	hipLaunchKernelGGL((getrf_panel_kernel<T, RMAX>), dim3(G), dim3(256), 0, ctx().stream, a);
	FH_HIP(hipGetLastError());
	const int steps = w < (int) m ? w : (int) m;
	if (G > 1)
		wk.counter_base += (unsigned long long) G * steps;
}

static idx_t next_pow2(idx_t n)
{
	idx_t p = 1;
	while (p < n)
		p <<= 1;
	return p;
}

// P: the panel (all remaining rows x n columns); col0/row_base: absolute position of P(0,0).
template <typename T> static void getrf_rec(MatV<T> P, int col0, int row_base, LuWork<T> &wk)
{
	const idx_t m = P.nrows, n = P.ncols; // n <= m
	if (n == 0)
		return;
	if (n <= LU_W) {
		getrf_leaf<T>(P, col0, row_base, wk);
		return;
	}
	// factor.rs:84-86 split rule
	const idx_t half = n / 2;
	idx_t pw = next_pow2(half);
	if (pw > 16)
		pw = 16;
	const idx_t bs = (half + pw - 1) / pw * pw;
	MatV<T> left = P.sub(0, 0, m, bs), right = P.sub(0, bs, m, n - bs);
	getrf_rec<T>(left, col0, row_base, wk);
	// bring the right half up to date: swaps, A01 <- L00^-1 A01, A11 -= A10 A01 (factor.rs:98-117)
	laswp_dev<T>(right, wk.piv + col0, (int) bs, row_base);
	MatV<T> A00 = P.sub(0, 0, bs, bs), A01 = P.sub(0, bs, bs, n - bs), A10 = P.sub(bs, 0, m - bs, bs),
		A11 = P.sub(bs, bs, m - bs, n - bs);
	trsm_lower_dev<T>(A00.c(), true, A01);
	gemm_dev<T>(A11, DST_FULL, true, A10.c(), A01.c(), (T) -1);
	getrf_rec<T>(A11, col0 + (int) bs, row_base + (int) bs, wk);
	// the right half's transpositions act on the rows below bs of the left half (factor.rs:127-185)
	laswp_dev<T>(A10, wk.piv + col0 + bs, (int) (n - bs < m - bs ? n - bs : m - bs), row_base + (int) bs);
}

template <typename T> long getrf_dev(MatV<T> A, idx_t *perm, idx_t *perm_inv)
{
	const idx_t m = A.nrows, n = A.ncols;
	FH_CHECK(m < (1L << 30) && n < (1L << 30), "partial_piv_lu: matrix too large");
	for (idx_t i = 0; i < m; ++i)
		perm[i] = i;
	const idx_t size = m < n ? m : n;
	long n_trans = 0;
	if (size > 0) {
		Scratch pivb((size_t) size * sizeof(int));
		Scratch slotv((size_t) 2 * LU_GMAX * (1 + LU_W) * sizeof(double));
		Scratch slotr((size_t) 2 * LU_GMAX * sizeof(int));
		Scratch diag((size_t) 2 * LU_W * sizeof(double));
		Scratch misc(256);
		LuWork<T> wk;
		wk.piv = pivb.as<int>();
		wk.slot_val = slotv.as<double>();
		wk.slot_row = slotr.as<int>();
		wk.diag_row = diag.as<double>();
		wk.counter = misc.as<unsigned long long>();
		wk.counter_base = 0;
		wk.status = misc.as<int>() + 8;
		FH_HIP(hipMemsetAsync(misc.p, 0, 256, ctx().stream));

		getrf_rec<T>(A.sub(0, 0, m, size), 0, 0, wk);
		if (m < n) { // factor.rs:278-285 (+ the swaps of the columns right of the square part)
			MatV<T> right = A.sub(0, size, m, n - size);
			laswp_dev<T>(right, wk.piv, (int) size, 0);
			trsm_lower_dev<T>(A.sub(0, 0, size, size).c(), true, right);
		}
		std::vector<int> piv((size_t) size);
		int st[4] = {0, 0, 0, 0};
		FH_HIP(hipMemcpyAsync(piv.data(), wk.piv, (size_t) size * sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
		FH_HIP(hipMemcpyAsync(st, wk.status, sizeof(st), hipMemcpyDeviceToHost, ctx().stream));
		ctx().sync();
		FH_CHECK(st[2] == 0, "partial_piv_lu: device barrier timed out in the panel kernel");
		// factor.rs:274-277: perm = identity with the transpositions applied in order
		for (idx_t j = 0; j < size; ++j) {
			const idx_t p = piv[(size_t) j];
			FH_CHECK(p >= j && p < m, "partial_piv_lu: corrupt pivot index");
			if (p != j) {
				std::swap(perm[j], perm[p]);
				++n_trans;
			}
		}
	}
	for (idx_t i = 0; i < m; ++i)
		perm_inv[perm[i]] = i;
	return n_trans;
}

template long getrf_dev<double>(MatV<double>, idx_t *, idx_t *);
template long getrf_dev<float>(MatV<float>, idx_t *, idx_t *);

} // namespace fh
