// LU with full pivoting for gfx950 -- faer/src/linalg/lu/full_pivoting/factor.rs:255-525.
//
// A level-2 algorithm: every step is one pass over the trailing matrix (rank-1 update) that also finds the next
// pivot, so it is an HBM stream, not MFMA work.  Two launches per step, no host synchronisation inside the loop:
//   pivot kernel  (one workgroup): combines the per-workgroup candidates of the previous pass in a fixed order,
//                 records the transpositions, swaps row k / column k with the pivot's, scales the pivot column
//                 (the pivot row on the transposed view) by the reciprocal pivot (factor.rs:333-362);
//   update kernel (grid over the trailing matrix, lanes along the unit stride): a_ij <- fma(-l_i, u_j, a_ij) and the
//                 running arg-max of |a_ij| in the same pass (:363-426, the reference's
//                 rank_one_update_and_best_in_matrix).
// Pivot rule = best_in_matrix_fallback (:255-273): column-major scan with a strict '>' on |a|, i.e. the first
// maximum in (column, row) order; a best score below the smallest positive normal ends the elimination (:324-332).
// Like the reference (:474-497) the driver works on the view whose row stride is the smaller one.
// Algorithmic bytes: sum_k 2 (m-k)(n-k) sizeof(T) ~ (2/3) n^3 sizeof(T) for a square matrix; roofline = HBM.
#include <limits>

#include "common.h"

namespace fh {

struct FpBest {
	double score, val; // |a|, a
	int row, col;
};

static __device__ __forceinline__ bool fp_better(const FpBest &a, const FpBest &b)
{
	return a.score > b.score || (a.score == b.score && a.score > 0.0 && (a.col < b.col || (a.col == b.col && a.row < b.row)));
}

static __device__ __forceinline__ FpBest fp_shfl_xor(const FpBest &v, int off)
{
	FpBest o;
	o.score = __shfl_xor(v.score, off, 64);
	o.val = __shfl_xor(v.val, off, 64);
	o.row = __shfl_xor(v.row, off, 64);
	o.col = __shfl_xor(v.col, off, 64);
	return o;
}

// workgroup-wide best (256 threads) -> valid in thread 0
static __device__ __forceinline__ FpBest fp_block_best(FpBest v, FpBest *s_part)
{
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		const FpBest o = fp_shfl_xor(v, off);
		if (fp_better(o, v))
			v = o;
	}
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0)
		s_part[wave] = v;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < (int) blockDim.x / 64; ++w)
			if (fp_better(s_part[w], v))
				v = s_part[w];
	}
	return v;
}

constexpr int FP_ROWS = 1024; // rows per workgroup (256 threads x 4)
constexpr int FP_COLS = 8;	  // columns per workgroup

// rows r0 .. m-1, columns c0 .. n-1 of V: optional rank-1 update with column / row k, candidate per workgroup
template <typename T>
__global__ __launch_bounds__(256) void fplu_update_kernel(T *V, idx_t rs, idx_t cs, int m, int n, int r0, int c0, int k, int do_update,
							   FpBest *partials, const int *done)
{
	__shared__ FpBest s_part[4];
	if (*done)
		return;
	const int tid = threadIdx.x;
	const int ib = r0 + blockIdx.x * FP_ROWS, jb = c0 + blockIdx.y * FP_COLS;
	T l[4];
#pragma unroll
	for (int r = 0; r < 4; ++r) {
		const int i = ib + tid + 256 * r;
		l[r] = (do_update && i < m) ? V[(idx_t) i * rs + (idx_t) k * cs] : (T) 0;
	}
	FpBest best{0.0, 0.0, 0, 0};
	// all FP_COLS x 4 elements of the thread are loaded before the first store: stores to V would otherwise pin every
	// later load behind them (same base pointer) and the pass would run one memory round trip per column
	T u[FP_COLS], v[FP_COLS][4];
#pragma unroll
	for (int jj = 0; jj < FP_COLS; ++jj) {
		const int j = min(jb + jj, n - 1);
		u[jj] = do_update ? V[(idx_t) k * rs + (idx_t) j * cs] : (T) 0;
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int i = min(ib + tid + 256 * r, m - 1); // clamped: unconditional loads
			v[jj][r] = V[(idx_t) i * rs + (idx_t) j * cs];
		}
	}
#pragma unroll
	for (int jj = 0; jj < FP_COLS; ++jj) {
		const int j = jb + jj;
		if (j >= n)
			break; // uniform
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int i = ib + tid + 256 * r;
			if (i < m) {
				T x = v[jj][r];
				if (do_update) {
					x = fh_fma(-l[r], u[jj], x);
					V[(idx_t) i * rs + (idx_t) j * cs] = x;
				}
				const FpBest c{fabs((double) x), (double) x, i, j};
				if (fp_better(c, best))
					best = c;
			}
		}
	}
	best = fp_block_best(best, s_part);
	if (tid == 0)
		partials[(size_t) blockIdx.y * gridDim.x + blockIdx.x] = best;
}

// Round 6: the pivot step is a grid over the longer of the two dimensions.  EVERY workgroup combines the candidates of the last pass in the
// same fixed order (a few thousand 24-byte records: cheaper than a launch of its own), workgroup 0 records the transpositions; then thread e
// of the grid swaps the pair (k, e) / (mr, e) of the two rows and the pair (e, k) / (e, mc) of the two columns and scales the pivot column
// (the pivot row on the transposed view) by the reciprocal pivot (factor.rs:333-362); the four entries where the swapped rows meet the
// swapped columns belong to thread 0 alone, so every entry is read and written by exactly one thread.  As ONE workgroup (rounds 1-5) the
// two swaps -- the row swap walks 2 n entries a leading dimension apart -- took 23 us per step whatever the size of the trailing matrix:
// more than the update pass.
template <typename T>
__global__ __launch_bounds__(256) void fplu_swap_kernel(T *V, idx_t rs, idx_t cs, int m, int n, int k, int size, int transpose,
							 const FpBest *__restrict__ partials, int nparts, int *rt, int *ct, int *done)
{
	__shared__ FpBest s_part[4];
	__shared__ FpBest s_best;
	if (*done)
		return;
	const int tid = threadIdx.x;
	FpBest best{0.0, 0.0, 0, 0};
	for (int p = tid; p < nparts; p += 256)
		if (fp_better(partials[p], best))
			best = partials[p];
	best = fp_block_best(best, s_part);
	if (tid == 0)
		s_best = best;
	__syncthreads();
	best = s_best;
	if (best.score < (double) std::numeric_limits<T>::min()) { // factor.rs:324-332
		if (blockIdx.x == 0) {
			for (int i = k + tid; i < size; i += 256) {
				rt[i] = i;
				ct[i] = i;
			}
		}
		// (every workgroup of THIS launch has read *done == 0 or will see 1 and return: both leave the matrix alone from here on)
		if (blockIdx.x == 0 && tid == 0)
			*done = 1;
		return;
	}
	const int e = blockIdx.x * 256 + tid;
	const int mr = best.row, mc = best.col;
	const T inv = (T) 1 / (T) best.val;
	auto at = [&](int i, int j) -> T & { return V[(idx_t) i * rs + (idx_t) j * cs]; };
	if (e == 0) {
		rt[k] = mr;
		ct[k] = mc;
		// new (i, j) = old (pr(i), pc(j)) on {k, mr} x {k, mc}, pr = (k mr), pc = (k mc)
		const int R[2] = {k, mr}, C[2] = {k, mc};
		T o[2][2];
#pragma unroll
		for (int x = 0; x < 2; ++x)
#pragma unroll
			for (int y = 0; y < 2; ++y)
				o[x][y] = at(R[x], C[y]);
#pragma unroll
		for (int x = 0; x < 2; ++x)
#pragma unroll
			for (int y = 0; y < 2; ++y) {
				if ((x == 1 && mr == k) || (y == 1 && mc == k))
					continue;
				const int i = R[x], j = C[y];
				T v = o[1 - x][1 - y];
				if (transpose ? (i == k && j > k) : (j == k && i > k))
					v *= inv;
				at(i, j) = v;
			}
	}
	const bool crosses_r = e == k || e == mr, crosses_c = e == k || e == mc;
	// ---- row pair in column e (not a swapped column), scaled on the transposed view
	if (e < n && !crosses_c) {
		T a = at(k, e), b = mr != k ? at(mr, e) : a;
		T nk = mr != k ? b : a;
		if (transpose && e > k)
			nk *= inv;
		if (mr != k)
			at(mr, e) = a;
		if (mr != k || (transpose && e > k))
			at(k, e) = nk;
	}
	// ---- column pair in row e (not a swapped row), scaled on the plain view
	if (e < m && !crosses_r) {
		T a = at(e, k), b = mc != k ? at(e, mc) : a;
		T nk = mc != k ? b : a;
		if (!transpose && e > k)
			nk *= inv;
		if (mc != k)
			at(e, mc) = a;
		if (mc != k || (!transpose && e > k))
			at(e, k) = nk;
	}
}

// A: m x n device view; the four permutation arrays are HOST arrays (m, m, n, n entries).  Returns the transposition count.
template <typename T> long full_piv_lu_dev(MatV<T> A, idx_t *row_perm, idx_t *row_perm_inv, idx_t *col_perm, idx_t *col_perm_inv)
{
	const idx_t M = A.nrows, N = A.ncols;
	FH_CHECK(M < (1L << 30) && N < (1L << 30), "full_piv_lu: matrix too large");
	const idx_t size = M < N ? M : N;
	for (idx_t i = 0; i < M; ++i)
		row_perm[i] = i;
	for (idx_t j = 0; j < N; ++j)
		col_perm[j] = j;
	long n_trans = 0;
	if (size > 0) {
		auto iabs = [](idx_t v) { return v < 0 ? -v : v; };
		const bool transpose = !(iabs(A.rs) < iabs(A.cs)); // factor.rs:474-497
		MatV<T> V = transpose ? A.t() : A;
		const int m = (int) V.nrows, n = (int) V.ncols;
		hipStream_t s = ctx().stream;
		auto grid_of = [&](int rows, int cols) { return dim3((unsigned) ((rows + FP_ROWS - 1) / FP_ROWS), (unsigned) ((cols + FP_COLS - 1) / FP_COLS)); };
		const dim3 g0 = grid_of(m, n);
		Scratch partb((size_t) g0.x * g0.y * sizeof(FpBest)), transb((size_t) (2 * size + 4) * sizeof(int));
		int *rt = transb.as<int>(), *ct = rt + size, *done = ct + size;
		FH_HIP(hipMemsetAsync(done, 0, 4 * sizeof(int), s));
		hipLaunchKernelGGL(fplu_update_kernel<T>, g0, dim3(256), 0, s, V.p, V.rs, V.cs, m, n, 0, 0, 0, 0, partb.as<FpBest>(), done);
		int nparts = (int) (g0.x * g0.y);
		const unsigned swap_grid = (unsigned) (((m > n ? m : n) + 255) / 256);
		for (idx_t k = 0; k < size; ++k) {
			hipLaunchKernelGGL(fplu_swap_kernel<T>, dim3(swap_grid), dim3(256), 0, s, V.p, V.rs, V.cs, m, n, (int) k, (int) size, transpose ? 1 : 0,
					   partb.as<const FpBest>(), nparts, rt, ct, done);
			if (k + 1 == size)
				break;
			const dim3 g = grid_of(m - (int) k - 1, n - (int) k - 1);
			hipLaunchKernelGGL(fplu_update_kernel<T>, g, dim3(256), 0, s, V.p, V.rs, V.cs, m, n, (int) k + 1, (int) k + 1, (int) k, 1,
					   partb.as<FpBest>(), done);
			nparts = (int) (g.x * g.y);
		}
		FH_HIP(hipGetLastError());
		std::vector<int> h((size_t) 2 * size);
		FH_HIP(hipMemcpyAsync(h.data(), rt, (size_t) 2 * size * sizeof(int), hipMemcpyDeviceToHost, s));
		ctx().sync();
		const int *hrt = h.data(), *hct = h.data() + size;
		for (idx_t k = 0; k < size; ++k)
			n_trans += (hrt[k] != k) + (hct[k] != k);
		// on the transposed view rows and columns trade places (factor.rs:486-496)
		const int *row_t = transpose ? hct : hrt, *col_t = transpose ? hrt : hct;
		for (idx_t i = 0; i < size; ++i) {
			FH_CHECK(row_t[i] >= 0 && row_t[i] < M && col_t[i] >= 0 && col_t[i] < N, "full_piv_lu: corrupt transposition");
			std::swap(row_perm[i], row_perm[row_t[i]]);
			std::swap(col_perm[i], col_perm[col_t[i]]);
		}
	}
	for (idx_t i = 0; i < M; ++i)
		row_perm_inv[row_perm[i]] = i;
	for (idx_t j = 0; j < N; ++j)
		col_perm_inv[col_perm[j]] = j;
	return n_trans;
}

template long full_piv_lu_dev<double>(MatV<double>, idx_t *, idx_t *, idx_t *, idx_t *);
template long full_piv_lu_dev<float>(MatV<float>, idx_t *, idx_t *, idx_t *, idx_t *);

} // namespace fh
