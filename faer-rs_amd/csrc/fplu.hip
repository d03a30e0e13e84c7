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
#include <atomic>
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
	// (the flag is looked at AFTER the loads of the pass are in flight -- they are clamped and always in bounds: a return in
	// front of them cost one more dependent round trip per launch)
	const int stop = *done;
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
		const T uj = V[(idx_t) k * rs + (idx_t) j * cs];
		u[jj] = do_update ? uj : (T) 0;
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int i = min(ib + tid + 256 * r, m - 1); // clamped: unconditional loads
			v[jj][r] = V[(idx_t) i * rs + (idx_t) j * cs];
		}
	}
	// (pins the loads in front of the branch: the compiler sinks them behind it otherwise)
#pragma unroll
	for (int jj = 0; jj < FP_COLS; ++jj)
#pragma unroll
		for (int r = 0; r < 4; ++r)
			asm volatile("" : "+v"(v[jj][r]));
	if (stop)
		return;
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
	const int stop = *done;
	const int tid = threadIdx.x;
	FpBest best{0.0, 0.0, 0, 0};
	// batches of 8 independent records per thread (a 4096 x 4096 pass leaves 2048): one round trip, not one per record
	for (int p0 = tid; p0 < nparts; p0 += 256 * 8) {
		FpBest c[8];
#pragma unroll
		for (int u = 0; u < 8; ++u)
			c[u] = partials[min(p0 + 256 * u, nparts - 1)];
#pragma unroll
		for (int u = 0; u < 8; ++u)
			if (p0 + 256 * u < nparts && fp_better(c[u], best))
				best = c[u];
	}
	asm volatile("" : "+v"(best.score), "+v"(best.val), "+v"(best.row), "+v"(best.col));
	if (stop)
		return;
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

// ------------------------------------------------------------------------------------------------
// Round 6, second version: ONE launch per step, out of place between two scratch copies of the matrix.
// In place, the interchange of step k (rows k / mr, columns k / mc of the WHOLE trailing matrix) has to be finished before any
// workgroup of the update pass reads its multipliers, hence a launch of its own in front of every pass (fplu_swap_kernel:
// 7.2 us of 26.6 per step at N = 4096).  Reading the trailing matrix from `cur` and writing the updated one to `nxt`, a tile
// applies the interchange while it loads -- new (i, j) = old (pr(i), pc(j)), pr = (k mr), pc = (k mc) -- and nothing it
// reads is written by anybody in the same launch:
//   * every workgroup combines the candidates of the last pass (same fixed order as before);
//   * tile workgroups (1024 x 8 as before; 1024 x 16 for the wide steps measured slower: 108 against 101 ms): their entries are loaded from the uninterchanged positions BESIDE the candidates
//     (one round trip), the few entries in row mr / column mc and the multipliers l_i = old(pr(i), mc), u_j = old(mr, pc(j))
//     (scaled like factor.rs:333-362) in a second one; a'_ij = fma(-l_i, u_j, a_ij) -> nxt, candidate -> partials;
//   * edge workgroups write row k and column k of the factors to the caller's matrix O, workgroup 0 the pivot and the
//     transpositions; the interchanges of the finished parts (columns < k of L, rows < k of U) are done in O by further
//     workgroups of the same launch.  O is never read as `cur`.
// Same arithmetic per entry and the same pivot rule as the in-place path (kept for matrices whose two copies would not be
// worth their memory, and for A/B tests: faer_hip_debug_fplu_inplace).  Traffic per step is unchanged: (m - k)(n - k) entries
// read and written once -- but two copies are twice the footprint: at N = 4096 (2 x 128 MiB) the first steps run SLOWER than in
// place (56 against 45 + 7 us: the in-place matrix stays in the 256 MiB last-level cache, the pair does not), the steps
// from ~1500 columns on faster; N = 4096 fp64 in all 104.4 -> 101.1 ms.
// ------------------------------------------------------------------------------------------------
template <typename T> struct FpStep {
	const T *cur; // m x n, column major, leading dimension ld (rows / columns >= k are live)
	T *nxt;
	idx_t ld;
	T *O; // the caller's matrix (the view V): factors
	idx_t ors, ocs;
	int m, n, k, size, transpose;
	const FpBest *pin; // candidates of the last pass
	int npin;
	FpBest *pout;
	int *rt, *ct, *done;
	int tiles_x, tiles_y; // tiles over rows k+1 .. m-1 / columns k+1 .. n-1
	int edge_c, edge_r;   // workgroups for column k / row k of the factors
	int sw;		      // workgroups for the interchanges inside the finished parts (256 indices < k each)
};

template <typename T, int FC> __global__ __launch_bounds__(256) void fplu_step_kernel(FpStep<T> a)
{
	__shared__ FpBest s_part[4];
	__shared__ FpBest s_best;
	const int stop = *a.done;
	const int tid = threadIdx.x, k = a.k, m = a.m, n = a.n;
	const idx_t ld = a.ld;
	const int ntile = a.tiles_x * a.tiles_y;
	const int wg = blockIdx.x; // 0: pivot + transpositions; 1 .. ntile: tiles; then column edge, row edge, interchanges
	const bool is_tile = wg >= 1 && wg <= ntile;
	const int bx = is_tile ? (wg - 1) % a.tiles_x : 0, by = is_tile ? (wg - 1) / a.tiles_x : 0;
	const int ib = k + 1 + bx * FP_ROWS, jb = k + 1 + by * FC;
	T v[FC][4];
	if (is_tile) {
#pragma unroll
		for (int jj = 0; jj < FC; ++jj) {
			const int j = min(jb + jj, n - 1);
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int i = min(ib + tid + 256 * r, m - 1);
				v[jj][r] = a.cur[(idx_t) i + (idx_t) j * ld];
			}
		}
	}
	FpBest best{0.0, 0.0, 0, 0};
	for (int p0 = tid; p0 < a.npin; p0 += 256 * 8) {
		FpBest c[8];
#pragma unroll
		for (int u = 0; u < 8; ++u)
			c[u] = a.pin[min(p0 + 256 * u, a.npin - 1)];
#pragma unroll
		for (int u = 0; u < 8; ++u)
			if (p0 + 256 * u < a.npin && fp_better(c[u], best))
				best = c[u];
	}
	// (pins both batches of loads in front of the branch and the reduction: the compiler sinks them to their first use otherwise)
	asm volatile("" : "+v"(best.score), "+v"(best.val), "+v"(best.row), "+v"(best.col));
	if (is_tile) {
#pragma unroll
		for (int jj = 0; jj < FC; ++jj)
#pragma unroll
			for (int r = 0; r < 4; ++r)
				asm volatile("" : "+v"(v[jj][r]));
	}
	if (stop)
		return;
	best = fp_block_best(best, s_part);
	if (tid == 0)
		s_best = best;
	__syncthreads();
	best = s_best;
	if (best.score < (double) std::numeric_limits<T>::min()) { // factor.rs:324-332: the elimination ends here
		if (wg == 0) {
			for (int i = k + tid; i < a.size; i += 256) {
				a.rt[i] = i;
				a.ct[i] = i;
			}
			if (tid == 0)
				*a.done = k + 1; // (the live block, rows / columns >= k of `cur`, goes back to O in fplu_finish_kernel)
		}
		return;
	}
	const int mr = best.row, mc = best.col;
	const T inv = (T) 1 / (T) best.val;
	const T lsc = a.transpose ? (T) 1 : inv, usc = a.transpose ? inv : (T) 1;
	auto pr = [&](int i) { return i == mr ? k : i; }; // (only called with i > k)
	auto pc = [&](int j) { return j == mc ? k : j; };
	auto Oat = [&](int i, int j) -> T & { return a.O[(idx_t) i * a.ors + (idx_t) j * a.ocs]; };
	if (is_tile) {
		T l[4], u[FC];
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int i = min(ib + tid + 256 * r, m - 1);
			l[r] = a.cur[(idx_t) pr(i) + (idx_t) mc * ld];
		}
#pragma unroll
		for (int jj = 0; jj < FC; ++jj) {
			const int j = min(jb + jj, n - 1);
			u[jj] = a.cur[(idx_t) mr + (idx_t) pc(j) * ld];
		}
		// the entries of the interchanged row / column come from somewhere else
		const bool col_hit = mc >= jb && mc < jb + FC;
		const bool row_hit = mr >= ib && mr < ib + FP_ROWS;
		if (col_hit || row_hit) {
#pragma unroll
			for (int jj = 0; jj < FC; ++jj) {
				const int j = min(jb + jj, n - 1);
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const int i = min(ib + tid + 256 * r, m - 1);
					if (i == mr || j == mc)
						v[jj][r] = a.cur[(idx_t) pr(i) + (idx_t) pc(j) * ld];
				}
			}
		}
#pragma unroll
		for (int r = 0; r < 4; ++r)
			l[r] *= lsc;
#pragma unroll
		for (int jj = 0; jj < FC; ++jj)
			u[jj] *= usc;
		FpBest nb{0.0, 0.0, 0, 0};
#pragma unroll
		for (int jj = 0; jj < FC; ++jj) {
			const int j = jb + jj;
			if (j >= n)
				break; // uniform
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int i = ib + tid + 256 * r;
				if (i < m) {
					const T x = fh_fma(-l[r], u[jj], v[jj][r]);
					a.nxt[(idx_t) i + (idx_t) j * ld] = x;
					const FpBest c{fabs((double) x), (double) x, i, j};
					if (fp_better(c, nb))
						nb = c;
				}
			}
		}
		__syncthreads(); // (s_part is reused)
		nb = fp_block_best(nb, s_part);
		if (tid == 0)
			a.pout[wg - 1] = nb;
		return;
	}
	int w = wg - 1 - ntile;
	if (wg == 0) {
		if (tid == 0) {
			a.rt[k] = mr;
			a.ct[k] = mc;
			Oat(k, k) = (T) best.val;
		}
		return;
	}
	if (w < a.edge_c) { // column k of the factors, rows k+1 ..
		const int i = k + 1 + w * 256 + tid;
		if (i < m)
			Oat(i, k) = a.cur[(idx_t) pr(i) + (idx_t) mc * ld] * lsc;
		return;
	}
	w -= a.edge_c;
	if (w < a.edge_r) { // row k of the factors, columns k+1 ..
		const int j = k + 1 + w * 256 + tid;
		if (j < n)
			Oat(k, j) = a.cur[(idx_t) mr + (idx_t) pc(j) * ld] * usc;
		return;
	}
	w -= a.edge_r;
	{ // interchanges inside the finished parts: rows k / mr in the columns e < k, columns k / mc in the rows e < k
		const int e = w * 256 + tid;
		if (e < k) {
			if (mr != k) {
				const T x = Oat(k, e), y = Oat(mr, e);
				Oat(k, e) = y;
				Oat(mr, e) = x;
			}
			if (mc != k) {
				const T x = Oat(e, k), y = Oat(e, mc);
				Oat(e, k) = y;
				Oat(e, mc) = x;
			}
		}
	}
}

// first pass: copy of V into the scratch matrix + candidates (grid = tiles over the whole matrix)
template <typename T>
__global__ __launch_bounds__(256) void fplu_first_kernel(const T *V, idx_t rs, idx_t cs, T *S, idx_t ld, int m, int n, FpBest *pout)
{
	__shared__ FpBest s_part[4];
	const int tid = threadIdx.x;
	const int ib = blockIdx.x * FP_ROWS, jb = blockIdx.y * FP_COLS;
	T v[FP_COLS][4];
#pragma unroll
	for (int jj = 0; jj < FP_COLS; ++jj) {
		const int j = min(jb + jj, n - 1);
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int i = min(ib + tid + 256 * r, m - 1);
			v[jj][r] = V[(idx_t) i * rs + (idx_t) j * cs];
		}
	}
	FpBest nb{0.0, 0.0, 0, 0};
#pragma unroll
	for (int jj = 0; jj < FP_COLS; ++jj) {
		const int j = jb + jj;
		if (j >= n)
			break;
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int i = ib + tid + 256 * r;
			if (i < m) {
				const T x = v[jj][r];
				S[(idx_t) i + (idx_t) j * ld] = x;
				const FpBest c{fabs((double) x), (double) x, i, j};
				if (fp_better(c, nb))
					nb = c;
			}
		}
	}
	nb = fp_block_best(nb, s_part);
	if (tid == 0)
		pout[(size_t) blockIdx.y * gridDim.x + blockIdx.x] = nb;
}

// the elimination ended early at step kd = *done - 1: rows / columns >= kd of the scratch matrix that was `cur` then are the rest of A
template <typename T>
__global__ __launch_bounds__(256) void fplu_finish_kernel(const T *S0, const T *S1, idx_t ld, T *O, idx_t ors, idx_t ocs, int m, int n, const int *done)
{
	const int d = *done;
	if (d == 0)
		return;
	const int kd = d - 1;
	const T *S = (kd & 1) ? S1 : S0;
	const idx_t rows = m - kd, total = rows * (idx_t) (n - kd);
	for (idx_t e = (idx_t) blockIdx.x * 256 + threadIdx.x; e < total; e += (idx_t) gridDim.x * 256) {
		const idx_t i = kd + e % rows, j = kd + e / rows;
		O[i * ors + j * ocs] = S[i + j * ld];
	}
}

static std::atomic<int> g_fplu_inplace{0};
void fplu_debug_inplace(int on) { g_fplu_inplace.store(on); }

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
		Scratch partb((size_t) 2 * g0.x * g0.y * sizeof(FpBest)), transb((size_t) (2 * size + 4) * sizeof(int));
		int *rt = transb.as<int>(), *ct = rt + size, *done = ct + size;
		FH_HIP(hipMemsetAsync(done, 0, 4 * sizeof(int), s));
		// two scratch copies of the matrix for the one-launch-per-step path: up to 2 x 4 GiB (N = 23170 in fp64), else in place
		const size_t copy_bytes = (size_t) m * (size_t) n * sizeof(T);
		if (!g_fplu_inplace.load() && copy_bytes <= ((size_t) 4 << 30)) {
			Scratch sb(2 * copy_bytes);
			const idx_t ld = m;
			T *S[2] = {sb.as<T>(), sb.as<T>() + (size_t) m * n};
			FpBest *P[2] = {partb.as<FpBest>(), partb.as<FpBest>() + (size_t) g0.x * g0.y};
			hipLaunchKernelGGL(fplu_first_kernel<T>, g0, dim3(256), 0, s, V.p, V.rs, V.cs, S[0], ld, m, n, P[0]);
			int nparts = (int) (g0.x * g0.y);
			for (idx_t k = 0; k < size; ++k) {
				FpStep<T> a;
				a.cur = S[k & 1];
				a.nxt = S[(k + 1) & 1];
				a.ld = ld;
				a.O = V.p;
				a.ors = V.rs;
				a.ocs = V.cs;
				a.m = m;
				a.n = n;
				a.k = (int) k;
				a.size = (int) size;
				a.transpose = transpose ? 1 : 0;
				a.pin = P[k & 1];
				a.npin = nparts;
				a.pout = P[(k + 1) & 1];
				a.rt = rt;
				a.ct = ct;
				a.done = done;
				const int tr = m - (int) k - 1, tc = n - (int) k - 1;
				a.tiles_x = tr > 0 && tc > 0 ? (tr + FP_ROWS - 1) / FP_ROWS : 0;
				a.tiles_y = tr > 0 && tc > 0 ? (tc + FP_COLS - 1) / FP_COLS : 0;
				a.edge_c = (tr + 255) / 256;
				a.edge_r = (tc + 255) / 256;
				a.sw = ((int) k + 255) / 256;
				const unsigned grid = 1u + (unsigned) (a.tiles_x * a.tiles_y + a.edge_c + a.edge_r + a.sw);
				hipLaunchKernelGGL((fplu_step_kernel<T, FP_COLS>), dim3(grid), dim3(256), 0, s, a);
				nparts = a.tiles_x * a.tiles_y;
			}
			hipLaunchKernelGGL(fplu_finish_kernel<T>, dim3(1024), dim3(256), 0, s, (const T *) S[0], (const T *) S[1], ld, V.p, V.rs, V.cs, m, n,
					   (const int *) done);
			FH_HIP(hipGetLastError());
			std::vector<int> h((size_t) 2 * size);
			FH_HIP(hipMemcpyAsync(h.data(), rt, (size_t) 2 * size * sizeof(int), hipMemcpyDeviceToHost, s));
			ctx().sync(); // (also: the scratch copies are released behind their last reader)
			const int *hrt = h.data(), *hct = h.data() + size;
			for (idx_t k = 0; k < size; ++k)
				n_trans += (hrt[k] != k) + (hct[k] != k);
			const int *row_t = transpose ? hct : hrt, *col_t = transpose ? hrt : hct;
			for (idx_t i = 0; i < size; ++i) {
				FH_CHECK(row_t[i] >= 0 && row_t[i] < M && col_t[i] >= 0 && col_t[i] < N, "full_piv_lu: corrupt transposition");
				std::swap(row_perm[i], row_perm[row_t[i]]);
				std::swap(col_perm[i], col_perm[col_t[i]]);
			}
			for (idx_t i = 0; i < M; ++i)
				row_perm_inv[row_perm[i]] = i;
			for (idx_t j = 0; j < N; ++j)
				col_perm_inv[col_perm[j]] = j;
			return n_trans;
		}
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
