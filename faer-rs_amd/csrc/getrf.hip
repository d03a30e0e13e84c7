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
// Leaf (<= 64 columns): ONE cooperative launch, "wavefront-level pivot reduction" (getrf_panel2_kernel below):
//   * the m x w panel is split in row chunks, one 512-thread workgroup per chunk, every thread keeping whole panel
//     rows in REGISTERS for the whole leaf (read from HBM once, written once); the leaf shape (64 x 1 ... 8 x 8
//     columns x rows per thread) is chosen so that all workgroups are resident;
//   * per column: DPP arg-max inside each wave, LDS across the 8 waves, then ONE all-to-all exchange per column:
//     every workgroup publishes its candidate {|a|, row, the candidate's whole panel row} and workgroup 0 the
//     diagonal row; every workgroup picks the winner itself, patches the two swapped rows from the published
//     copies and performs scale + rank-1 update on its own rows;
//   * the exchange follows the gfx950 recipe for in-launch hand-offs (xwg.h; cdna_hip_programming.md Guideline 16
//     R2): data-tagged 8-byte granules written with write-through (sc1) stores and read back with sc1 loads -- no
//     flags, no fences, no store drain; two memory round trips per column.  Every spin is bounded; a timeout
//     raises a device error instead of hanging.
// Row interchanges of the outside columns are NOT applied one transposition at a time: the transposition list is
// composed into a net row permutation in parallel (each row traces its source backwards through the list) and
// applied as one gather.  Large matrices run the recursion with look-ahead on two CU-masked streams
// (getrf_lookahead).
#include <climits>

#include <atomic>

#include "common.h"
#include <optional>
#include <functional>
#include "lds_blocks.h"
#include "mfma.h"
#include "xwg.h"
#include "lu_wpanel.h"
#include "lu_small_leaf.h"

namespace fh {

constexpr int LU_W = 64; // leaf width of the recursion

// ------------------------------------------------------------------------------------------------
// Register-resident cooperative panel kernel.  Phase timing of its LDS-resident predecessor
// (profiles/r01_lu_panel_phase_timing.txt) showed that only ~1.6 us of ~6.4 us per column was the
// cross-workgroup hand-off; the rest was LDS round trips of the column scan, the winner selection and the
// rank-1 update.  Here every thread keeps RPT whole panel rows in registers (W = 32 columns each):
//   * the column loop is unrolled at compile time (column index = template constant), so the update of the
//     trailing columns is 31 - j register FMAs per row against the pivot row read once from LDS;
//   * the local arg-max comes straight out of registers; wave reductions + one LDS hop across the 8 waves;
//   * wave 0 alone runs the exchange: publish -> poll -> read the G candidate records -> pick the winner ->
//     fetch its row and the diagonal row -> hand them to the workgroup through LDS (3 __syncthreads per column);
//   * 1024 rows per workgroup => at most 16 workgroups for a 16384-row panel (fewer flags, fewer records).
// ------------------------------------------------------------------------------------------------
constexpr int LU2_NT = 512;	  // threads per workgroup
constexpr int LU2_GMAX = 2048;	  // workgroups per panel (rows / (LU2_NT * RPT))
constexpr int LU_WMAX = 64;	  // widest leaf
constexpr int LU2_GSLOT = 4 + 2 * LU_WMAX; // granules per producer slot: {row}, {|a| hi}, {|a| lo}, pad, W x {hi, lo}

template <typename T> struct Panel2Args {
	T *P;
	idx_t rs, cs;
	int m, w;
	int *piv; // piv[j] = row_base + pivot row
	int row_base;
	xwg_u64 *gran;	    // [2][G][LU2_GSLOT] tagged granules
	xwg_u64 *gran_diag; // [2][2 * LU_WMAX] row J as published by workgroup 0
	xwg_u64 epoch_base;
	int *status;
};

template <typename T, int W> struct Panel2Shared {
	double wv[LU2_NT / 64];
	int wr[LU2_NT / 64];
	T cand[W]; // this workgroup's candidate row
	T drow[W]; // row j (workgroup 0)
	T piv[W];  // winning pivot row
	T diag[W]; // row j as published
	int p;	   // winning row (global)
	int flag;
};

static __device__ __forceinline__ double gran_pair_to_double(xwg_u64 h, xwg_u64 l)
{
	return __longlong_as_double((long long) (((h & 0xffffffffull) << 32) | (l & 0xffffffffull)));
}

// One column step.  The panel lives in registers in ROTATED order: after every group of 8 columns each row is
// rotated left by 8 positions, so the column being eliminated always sits at a compile-time position JJ in
// 0..7 while the group index `grp` is a run-time value.  That keeps the code to 8 unrolled step bodies for
// any leaf width (a fully unrolled 64-column kernel takes tens of minutes to compile), finished columns stay
// in the registers (positions >= lim = W - 8 grp) and keep taking part in the row interchanges.
// Returns false on exchange timeout.  W <= 64 so that wave 0 has one lane per panel column.
template <typename T, int W, int RPT, int JJ>
static __device__ __forceinline__ bool panel2_step(const Panel2Args<T> &a, T (&x)[RPT][W], Panel2Shared<T, W> &sh, int r0, int G, int grp)
{
	constexpr int R = LU2_NT * RPT;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int g = blockIdx.x;
	const int J = grp * 8 + JJ; // global panel column
	const int lim = W - grp * 8; // positions < lim hold unfinished columns
	const int q = J & 1;
	// ---- 1. local arg-max of |a(:, J)| over owned rows >= J (first strictly largest, factor.rs:35-43)
	double bv = 0.0;
	int br = INT_MAX;
#pragma unroll
	for (int i = 0; i < RPT; ++i) {
		const int gr = r0 + tid + i * LU2_NT;
		const double av = fabs((double) x[i][JJ]);
		if (gr >= J && gr < a.m && av > bv) {
			bv = av;
			br = gr;
		}
	}
	wave_argmax2(bv, br);
	if (lane == 0) {
		sh.wv[wave] = bv;
		sh.wr[wave] = br;
	}
	__syncthreads();
	bv = sh.wv[0];
	br = sh.wr[0];
#pragma unroll
	for (int k = 1; k < LU2_NT / 64; ++k)
		if (better(sh.wv[k], sh.wr[k], bv, br)) {
			bv = sh.wv[k];
			br = sh.wr[k];
		}
	if (!(bv > 0.0))
		br = INT_MAX; // zero / NaN-only chunk: no candidate
	// Data-tagged granules (xwg.h, recipe R2): every 8-byte word carries {epoch tag, 32 payload bits} and is written by
	// ONE write-through store, so it needs neither a store drain nor a flag: a consumer that reads the expected tag has
	// the data.  The HEADER of this workgroup's record (candidate value and row) goes out at once: the other workgroups
	// only need the headers to pick the winner, so that round trip runs while the candidate row is still being parked
	// and published (only the winner's row is fetched, one round trip later): LU N = 16384 124.7 -> 120.9 ms.
	// Going further -- updating the NEXT column first, publishing its header in the middle of this step and finishing
	// the rank-1 update while it travels (in-panel look-ahead) -- was built and measured 11 % SLOWER: the restructured
	// step needs ~30 registers more than the 256 a 512-thread workgroup leaves per lane and spills to scratch
	// (163 vs 146 ms in one visit on a slow box).
	const unsigned tag = (unsigned) (a.epoch_base + (xwg_u64) (J + 1));
	xwg_u64 *sg = a.gran + ((size_t) q * G + g) * LU2_GSLOT;
	if (G > 1 && tid == 0) {
		const xwg_u64 vb = (xwg_u64) __double_as_longlong(bv);
		xwg_store_gran(sg + 0, tag, (unsigned) br);
		xwg_store_gran(sg + 1, tag, (unsigned) (vb >> 32));
		xwg_store_gran(sg + 2, tag, (unsigned) vb);
	}
	// The WAVE that owns the candidate row (and, in workgroup 0, the one that owns row J) parks it in LDS and publishes it
	// ITSELF: lane c of that wave stores register position c.  Wave 0 does not wait for it -- there is no workgroup barrier
	// between the arg-max and the exchange any more -- it goes straight to polling the record headers, and only the
	// winner's row is fetched one round trip later, long after its owner's stores.  (G == 1: the rows are read from LDS
	// behind the barrier below.)
#pragma unroll
	for (int i = 0; i < RPT; ++i) {
		const int gr = r0 + tid + i * LU2_NT;
		const bool mine_c = gr == br, mine_d = gr == J;
		if (__any(mine_c)) { // wave uniform
			if (mine_c) {
#pragma unroll
				for (int c = 0; c < W; ++c)
					sh.cand[c] = x[i][c];
			}
			__builtin_amdgcn_wave_barrier();
			if (G > 1 && lane < W) {
				const xwg_u64 cb = (xwg_u64) __double_as_longlong((double) sh.cand[lane]);
				xwg_store_gran(sg + 4 + 2 * lane, tag, (unsigned) (cb >> 32));
				xwg_store_gran(sg + 5 + 2 * lane, tag, (unsigned) cb);
			}
		}
		if (__any(mine_d)) { // (only in the workgroup that holds row J)
			if (mine_d) {
#pragma unroll
				for (int c = 0; c < W; ++c)
					sh.drow[c] = x[i][c];
			}
			__builtin_amdgcn_wave_barrier();
			if (G > 1 && lane < W) {
				const xwg_u64 db = (xwg_u64) __double_as_longlong((double) sh.drow[lane]);
				xwg_u64 *dg = a.gran_diag + (size_t) q * 2 * LU_WMAX + 2 * lane;
				xwg_store_gran(dg, tag, (unsigned) (db >> 32));
				xwg_store_gran(dg + 1, tag, (unsigned) db);
			}
		}
	}
	if (G == 1)
		__syncthreads();
	// ---- 2. wave 0: exchange and winner selection
	if (G > 1) {
		if (wave == 0) {
			// Two round trips per column (record headers, then the winner's row) instead of four (drain, flag, records,
			// row).  Lane c handles register position c.
			// ---- records of all producers (lane t reads producer t, strided for G > 64)
			int ok = 0;
			double v = 0.0;
			int r = INT_MAX;
			for (int spin = 0; spin < (1 << 21); ++spin) {
				bool all = true;
				v = 0.0;
				r = INT_MAX;
				for (int t = lane; t < G; t += 64) {
					const xwg_u64 *rec = a.gran + ((size_t) q * G + t) * LU2_GSLOT;
					const xwg_u64 g0 = xwg_load_gran(rec), g1 = xwg_load_gran(rec + 1), g2 = xwg_load_gran(rec + 2);
					all = all && (unsigned) (g0 >> 32) == tag && (unsigned) (g1 >> 32) == tag && (unsigned) (g2 >> 32) == tag;
					const int rr = (int) (unsigned) g0;
					const double vv = gran_pair_to_double(g1, g2);
					if (rr != INT_MAX && better(vv, rr, v, r)) {
						v = vv;
						r = rr;
					}
				}
				if (__all(all)) {
					ok = 1;
					break;
				}
				__builtin_amdgcn_s_sleep(1);
			}
			wave_argmax2(v, r);
			const int p = r == INT_MAX ? J : r;
			const int gw = r == INT_MAX ? 0 : r / R;
			// ---- the winner's row and the diagonal row
			if (ok) {
				const bool want = lane < W, want_piv = want && p != J;
				const xwg_u64 *ps = a.gran + ((size_t) q * G + gw) * LU2_GSLOT + 4 + 2 * (lane & (W - 1));
				const xwg_u64 *ds = a.gran_diag + (size_t) q * 2 * LU_WMAX + 2 * (lane & (W - 1));
				xwg_u64 ph = 0, pl = 0, dh = 0, dl = 0;
				ok = 0;
				for (int spin = 0; spin < (1 << 21); ++spin) {
					bool got = true;
					if (want) {
						dh = xwg_load_gran(ds);
						dl = xwg_load_gran(ds + 1);
						got = (unsigned) (dh >> 32) == tag && (unsigned) (dl >> 32) == tag;
					}
					if (want_piv) {
						ph = xwg_load_gran(ps);
						pl = xwg_load_gran(ps + 1);
						got = got && (unsigned) (ph >> 32) == tag && (unsigned) (pl >> 32) == tag;
					}
					if (__all(got)) {
						ok = 1;
						break;
					}
					__builtin_amdgcn_s_sleep(1);
				}
				if (want) {
					const T dval = (T) gran_pair_to_double(dh, dl);
					sh.diag[lane] = dval;
					sh.piv[lane] = want_piv ? (T) gran_pair_to_double(ph, pl) : dval; // p == J: row J is the pivot row
				}
			}
			if (lane == 0) {
				sh.p = p;
				sh.flag = ok;
			}
		}
	} else {
		if (tid < W) {
			const int p = br == INT_MAX ? J : br;
			sh.diag[tid] = sh.drow[tid];
			sh.piv[tid] = p == J ? sh.drow[tid] : sh.cand[tid];
			if (tid == 0) {
				sh.p = p;
				sh.flag = 1;
			}
		}
	}
	__syncthreads();
	if (!sh.flag)
		return false;
	// ---- 3. swap rows J <-> p (all W positions, finished columns included), scale by the reciprocal pivot,
	//         rank-1 update of the unfinished columns (factor.rs:45-64)
	const int p = sh.p;
	if (g == 0 && tid == 0)
		a.piv[J] = a.row_base + p;
	const T inv = (T) 1 / sh.piv[JJ];
	T l[RPT];
#pragma unroll
	for (int i = 0; i < RPT; ++i) {
		const int gr = r0 + tid + i * LU2_NT;
		if (p != J) {
			// wave-uniform guards: exactly one lane of the whole grid owns each of the two rows, and without
			// the guards the compiler speculates the 2 W LDS loads of every thread into registers
			if (__any(gr == p)) {
				if (gr == p) {
#pragma unroll
					for (int c = 0; c < W; ++c)
						x[i][c] = sh.diag[c];
				}
			}
			if (__any(gr == J)) {
				if (gr == J) {
#pragma unroll
					for (int c = 0; c < W; ++c)
						x[i][c] = sh.piv[c];
				}
			}
		}
		l[i] = (T) 0;
		if (gr > J) {
			l[i] = x[i][JJ] * inv;
			x[i][JJ] = l[i];
		}
	}
	// positions in blocks of 8: a block takes part while it still holds unfinished columns (wave-uniform test);
	// rows with gr <= J carry l == 0 but must stay bitwise untouched (l * u could be NaN for an infinite u)
	// The pivot row reaches the FMAs through the scalar unit: lane c of every wave reads position c ONCE from LDS,
	// the multiplier of position p is v_readlane(.., p) (wave uniform, an SGPR operand of the FMA).  Read per thread
	// from LDS instead, the 64 broadcast reads per thread and column keep the CU's one LDS pipe busy for ~3000
	// cycles per column (8 waves x 64 reads); the readlanes run on the four SIMDs in parallel.
	const T myu = sh.piv[lane & (W - 1)];
#pragma unroll
	for (int cb = 0; cb < W / 8; ++cb) {
		if (cb * 8 < lim) {
			T u[8];
#pragma unroll
			for (int k = 0; k < 8; ++k)
				u[k] = lane_bcast(myu, cb * 8 + k);
#pragma unroll
			for (int i = 0; i < RPT; ++i) {
				const int gr = r0 + tid + i * LU2_NT;
				if (gr > J) {
#pragma unroll
					for (int k = 0; k < 8; ++k)
						if (cb * 8 + k > JJ)
							x[i][cb * 8 + k] = fh_fma(l[i], -u[k], x[i][cb * 8 + k]); // rank_update_imp: fma(l_i, -u_c, dst)
				}
			}
		}
	}
	return true;
}

template <typename T, int W, int RPT, int JJ> struct Panel2Group {
	// the 8 column steps of group `grp`; stops at `steps`; returns 1 = group complete, 0 = stopped early, -1 = timeout
	static __device__ __forceinline__ int run(const Panel2Args<T> &a, T (&x)[RPT][W], Panel2Shared<T, W> &sh, int r0, int G, int grp,
						  int steps)
	{
		if constexpr (JJ < 8) {
			if (grp * 8 + JJ >= steps)
				return 0;
			if (!panel2_step<T, W, RPT, JJ>(a, x, sh, r0, G, grp))
				return -1;
			return Panel2Group<T, W, RPT, JJ + 1>::run(a, x, sh, r0, G, grp, steps);
		} else {
			return 1;
		}
	}
};

template <typename T, int W, int RPT> __global__ __launch_bounds__(LU2_NT) void getrf_panel2_kernel(const Panel2Args<T> a)
{
	static_assert((W & (W - 1)) == 0 && W % 8 == 0 && W <= 64, "leaf width");
	__shared__ Panel2Shared<T, W> sh;
	constexpr int R = LU2_NT * RPT;
	const int tid = threadIdx.x;
	const int g = blockIdx.x, G = gridDim.x;
	const int r0 = g * R;
	const int w = a.w;
	T x[RPT][W];
#pragma unroll
	for (int i = 0; i < RPT; ++i) {
		const int gr = r0 + tid + i * LU2_NT;
#pragma unroll
		for (int c = 0; c < W; ++c) {
			const bool in = gr < a.m && c < w;
			const T v = a.P[in ? (idx_t) gr * a.rs + (idx_t) c * a.cs : (idx_t) 0];
			x[i][c] = in ? v : (T) 0;
		}
	}
	const int steps = min(w, a.m);
	int rot = 0; // positions the rows have been rotated by so far
	for (int grp = 0; grp * 8 < steps; ++grp) {
		const int st = Panel2Group<T, W, RPT, 0>::run(a, x, sh, r0, G, grp, steps);
		if (st < 0) {
			if (tid == 0)
				atomicExch(a.status + 2, 1);
			return;
		}
		if (st == 0)
			break;
		// rotate every row left by 8: the finished columns go to the tail
#pragma unroll
		for (int i = 0; i < RPT; ++i) {
			T t8[8];
#pragma unroll
			for (int k = 0; k < 8; ++k)
				t8[k] = x[i][k];
#pragma unroll
			for (int c = 0; c + 8 < W; ++c)
				x[i][c] = x[i][c + 8];
#pragma unroll
			for (int k = 0; k < 8; ++k)
				x[i][W - 8 + k] = t8[k];
		}
		rot += 8;
	}
#pragma unroll
	for (int i = 0; i < RPT; ++i) {
		const int gr = r0 + tid + i * LU2_NT;
#pragma unroll
		for (int c = 0; c < W; ++c) {
			const int gc = (c + rot) & (W - 1); // global panel column of register position c
			if (gr < a.m && gc < w)
				a.P[(idx_t) gr * a.rs + (idx_t) gc * a.cs] = x[i][c];
		}
	}
}

constexpr int LU2_NSLOT = 3; // epochs of granule slots the exchange workspace holds

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

// One-launch variant for short transposition lists (nt <= LASWP_SMALL_NT: every node of the recursion below
// 1024 columns, i.e. all but a handful of the calls): each workgroup owns LASWP_CC columns, rebuilds the net
// permutation of the <= 2 nt affected rows itself (parallel backward trace through the list held in LDS),
// gathers their sources into LDS and scatters them back -- no global temporary, no separate compose pass.
constexpr int LASWP_SMALL_NT = 512;
constexpr int LASWP_CC = 8;
constexpr int LUN_W_HOST = 64; // (= LUN_W, defined with the node kernel below)

// `top_out` (optional, nt x ncols, column major with pitch nt): receives a copy of the rows 0 .. nt-1 AFTER the interchanges
// (the fused node update below reads the block it overwrites from there).
static __device__ __forceinline__ double lu_readlane(double v, int l)
{
	const long long b = __double_as_longlong(v);
	const int lo = __builtin_amdgcn_readlane((int) b, l), hi = __builtin_amdgcn_readlane((int) (b >> 32), l);
	return __longlong_as_double(((long long) hi << 32) | (unsigned) lo);
}
static __device__ __forceinline__ float lu_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// `L00` (optional, with top_out and nt == 64): the unit lower 64 x 64 block of the leaf whose interchanges these are (column stride lcs).
// The workgroup then SOLVES its columns of the interchanged top block, U = L00^-1 top (factor.rs:104-109: forward substitution, the
// same fma per row and column in the same order as lu_node64_kernel's), and writes U into rows 0 .. 63 of B and into top_out: the node
// launch behind it finds the solved block and only updates (round 6; every workgroup of the node launch used to repeat the solve of
// its column group: a third of that launch on the panel chain).
template <typename T>
__global__ __launch_bounds__(256) void laswp_small_kernel(T *B, idx_t rs, idx_t cs, int nrows, int ncols,
							  const int *__restrict__ piv, int nt, int row_base, T *top_out, const T *__restrict__ L00, idx_t lcs)
{
	__shared__ int s_piv[LASWP_SMALL_NT];
	__shared__ int s_dst[2 * LASWP_SMALL_NT], s_src[2 * LASWP_SMALL_NT];
	__shared__ T tmp[LASWP_CC * 2 * LASWP_SMALL_NT];
	const int tid = threadIdx.x;
	const bool solve = L00 != nullptr && top_out != nullptr; // (host: nt == 64)
	// L00 into the part of tmp the gather does not use (nt = 64: 8 x 128 entries of it), row major with pitch 65 -- issued first, so that
	// its round trip overlaps the list building below
	T *s_l = tmp + LASWP_CC * 2 * 64;
	if (solve) {
		T lv[16];
#pragma unroll
		for (int q = 0; q < 16; ++q) {
			const int idx = tid + 256 * q;
			lv[q] = L00[(idx & 63) + (idx_t) (idx >> 6) * lcs];
		}
#pragma unroll
		for (int q = 0; q < 16; ++q) {
			const int idx = tid + 256 * q;
			s_l[(idx & 63) * 65 + (idx >> 6)] = lv[q];
		}
	}
	for (int j = tid; j < nt; j += 256)
		s_piv[j] = piv[j] - row_base;
	__syncthreads();
	// affected destinations: rows 0 .. nt-1 and the pivot rows >= nt (duplicates are harmless: same source)
	const int ne = 2 * nt;
	for (int e = tid; e < ne; e += 256) {
		int d = e < nt ? e : s_piv[e - nt];
		int pos = -1;
		if (e < nt || d >= nt) {
			pos = d;
			for (int j = nt - 1; j >= 0; --j) {
				const int pj = s_piv[j];
				pos = pos == j ? pj : (pos == pj ? j : pos);
			}
			if (pos == d)
				pos = -1; // row stays where it is
		}
		s_dst[e] = d;
		s_src[e] = pos;
	}
	__syncthreads();
	const int c0 = blockIdx.x * LASWP_CC;
	const int nc = min(LASWP_CC, ncols - c0);
	// (solve: this thread's two rows of the top block as they are now -- used where the row stays; in flight during the gather)
	T keep0 = (T) 0, keep1 = (T) 0;
	if (solve && (tid >> 5) < nc) {
		keep0 = B[(idx_t) (tid & 31) * rs + (idx_t) (c0 + (tid >> 5)) * cs];
		keep1 = B[(idx_t) ((tid & 31) + 32) * rs + (idx_t) (c0 + (tid >> 5)) * cs];
	}
	// gather in batches of 8 independent loads per thread (one memory round trip per batch, not per element)
	for (int idx0 = tid; idx0 < nc * ne; idx0 += 256 * 8) {
		T v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int idx = idx0 + u * 256;
			const bool in = idx < nc * ne;
			const int c = in ? idx / ne : 0, e = in ? idx - c * ne : 0;
			const int sr = in ? s_src[e] : -1;
			v[u] = B[sr >= 0 ? (idx_t) sr * rs + (idx_t) (c0 + c) * cs : (idx_t) 0]; // unconditional load, clamped address
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int idx = idx0 + u * 256;
			if (idx < nc * ne)
				tmp[idx] = v[u];
		}
	}
	__syncthreads();
	for (int idx = tid; idx < nc * ne; idx += 256) {
		const int c = idx / ne, e = idx - c * ne;
		if (s_src[e] >= 0 && !(solve && s_dst[e] < nt)) // (rows 0 .. 63 receive U below)
			B[(idx_t) s_dst[e] * rs + (idx_t) (c0 + c) * cs] = tmp[idx];
	}
	if (solve) {
		// thread: column c = tid >> 5 of the workgroup's eight, rows h and h + 32; row k's value is read from the lane that owns it
		// (v_readlane per half wavefront: a column lives in one half)
		const int c = tid >> 5, h = tid & 31, lane = tid & 63;
		const bool vc = c < nc;
		T u0 = (T) 0, u1 = (T) 0;
		if (vc) {
			u0 = s_src[h] >= 0 ? tmp[c * ne + h] : keep0;
			u1 = s_src[h + 32] >= 0 ? tmp[c * ne + h + 32] : keep1;
		}
#pragma unroll
		for (int k = 0; k < 64; ++k) {
			const T src = k < 32 ? u0 : u1;
			const T lo = lu_readlane(src, k & 31), hi = lu_readlane(src, 32 + (k & 31));
			const T uk = lane < 32 ? lo : hi;
			if (k < 31) // (rows h <= k of the first half are final)
				u0 = h > k ? fh_fma(-s_l[h * 65 + k], uk, u0) : u0;
			u1 = h + 32 > k ? fh_fma(-s_l[(h + 32) * 65 + k], uk, u1) : u1;
		}
		if (vc) {
			B[(idx_t) h * rs + (idx_t) (c0 + c) * cs] = u0;
			B[(idx_t) (h + 32) * rs + (idx_t) (c0 + c) * cs] = u1;
			top_out[(size_t) (c0 + c) * 64 + h] = u0;
			top_out[(size_t) (c0 + c) * 64 + h + 32] = u1;
		}
		return;
	}
	if (top_out) {
		for (int idx = tid; idx < nc * nt; idx += 256) {
			const int c = idx / nt, d = idx - c * nt; // (entry e = d of the lists is row d)
			top_out[(size_t) (c0 + c) * nt + d] = s_src[d] >= 0 ? tmp[c * ne + d] : B[(idx_t) d * rs + (idx_t) (c0 + c) * cs];
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Fused update of the right sibling of a 64-column leaf (the 128-column nodes of the recursion, factor.rs:98-117):
//     A01 <- L00^-1 A01  (64 x nr, unit lower L00),   A11 -= A10 A01  ((m - 64) x nr, K = 64)
// in ONE launch instead of substitution leaf + product (28 + 45-65 us on the panel stream's 32 CUs, each bound by its own
// latency): every workgroup solves the 64 x 64 system itself from the copy of the interchanged top block the interchange
// launch left in `top` (so nobody reads what workgroup 0 overwrites with A01), then updates its own 256 rows on the
// matrix cores straight from / to memory.
//   * solve: thread = column c of the block, wavefront w holds the rows 16 w .. 16 w + 15 in registers; four 16-row steps:
//     the owner substitutes inside its diagonal block (multipliers are LDS broadcasts), writes its rows of U to LDS, the
//     wavefronts below eliminate against them;
//   * product: D' = (A10 A01)^T per 16 x 16 tile (mfma.h: operand a = U[k][c], b = -L[r][k]), so that 16 consecutive lanes
//     hold 16 consecutive rows of a column: 128-byte accesses in the column-major panel.
// ------------------------------------------------------------------------------------------------
constexpr int LUN_W = 64;   // left width = rows of the triangular system
constexpr int LU_FLAT_MAXW = 512; // widest flat panel / widest run of column groups one node launch serves: sizes LuWork::ttop
constexpr int LUN_LP = 65;  // LDS pitch of L00 (row major: broadcast reads)
constexpr int LUN_UP = 80;  // LDS pitch of U (k major; pitch = 16 mod 32 keeps the four k groups of an operand read on disjoint banks)
constexpr int LUN_ROWS = 256; // rows of A11 per workgroup
// the node kernel addresses its 64-column blocks as base + 32-bit byte offset: rows + 64 columns must stay below 4 GB
template <typename T> static bool lu_node_offsets_ok(idx_t m, idx_t cs)
{
	return m > 0 && cs > 0 && m < (1L << 30) && (double) sizeof(T) * ((double) m + 64.0 * (double) cs) < 4294967296.0;
}

#ifdef FH_LU_TIMING
// timing build: s_memtime ticks of workgroup (0, 0), wavefront 0 per phase of the node kernel, summed over launches (lu_dump_timing)
__device__ unsigned long long g_node_phase[8];
#define LUN_TICK(i)                                                                                                                      \
	do {                                                                                                                             \
		if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {                                                                    \
			const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                    \
			atomicAdd(&g_node_phase[i], now_ - tick_);                                                                       \
			tick_ = now_;                                                                                                    \
		}                                                                                                                        \
	} while (0)
#else
#define LUN_TICK(i)
#endif
template <typename T>
__global__ __launch_bounds__(256, 2) void lu_node64_kernel(T *P, idx_t cs, int m, int nr, const T *__restrict__ top, int solved)
{
	typedef typename Mfma<T>::acc_t acc_t;
	__shared__ T Ls[LUN_W * LUN_LP];
	__shared__ T Us[LUN_W * LUN_UP];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int l15 = lane & 15, lhi = lane >> 4;
	// blockIdx.y: group of 64 right-hand columns (the flat panel driver updates ALL columns right of a leaf with one launch; the
	// groups are independent: each workgroup solves for its own group and updates its 256 rows of it)
	const int cg = blockIdx.y;
	nr = min(LUN_W, nr - cg * LUN_W);
	top += (size_t) cg * LUN_W * LUN_W;
	T *B = P + (idx_t) (LUN_W + cg * LUN_W) * cs; // the right block
#ifdef FH_LU_TIMING
	unsigned long long tick_ = __builtin_amdgcn_s_memtime();
#endif
	// ---- A11 -= A10 A01 runs on 16-row tiles: chunks of LUN_ROWS rows (several per workgroup when the launch has many of them:
	// the solve below is paid once per workgroup), wavefront w takes 64 rows of a chunk, 16 at a time.  Every tile is a dependent
	// "32 loads -> 64 matrix-core steps -> 16 stores", and a wavefront runs its tiles one after the other: with the loads issued
	// when the tile starts, each tile waited a full memory round trip (alone on the panel stream 4-5 us per tile for < 1 us of
	// arithmetic, profiles/r05_exp_lu_driver.txt item 8).  So the loads run ONE TILE AHEAD in a second set of registers -- raw
	// values, clamped addresses, no arithmetic on them before the tile is due -- and the first tile's loads are issued before the
	// triangular solve, which neither reads nor writes those rows.
	const int nchunks = (m - LUN_W + LUN_ROWS - 1) / LUN_ROWS;
	auto tile_r0 = [&](int idx) -> int { // first row of this wavefront's idx-th tile, -1 behind the last one (wave uniform)
		const int chunk = (int) blockIdx.x + (idx >> 2) * (int) gridDim.x;
		if (chunk >= nchunks)
			return -1;
		const int r0 = LUN_W + chunk * LUN_ROWS + wave * 64 + 16 * (idx & 3);
		return r0 < m ? r0 : -1;
	};
	// lanes outside the matrix (rows >= m, columns >= nr) load a valid entry instead and never store: every output entry depends on
	// its own row of L, its own column of U and its own old value only
	// (addresses as uniform base + 32-bit byte offset per lane -- lu_node_offsets_ok on the host: 64-bit addresses for the 2 x 32 loads in
	// flight cost more registers than two workgroups per CU leave)
	const char *Pc = reinterpret_cast<const char *>(P), *Bc = reinterpret_cast<const char *>(B);
	const unsigned csb = (unsigned) cs * (unsigned) sizeof(T);
	auto tile_load = [&](int r0, T (&braw)[16], acc_t (&acc)[4]) {
		const unsigned rb = (unsigned) min(r0 + l15, m - 1) * (unsigned) sizeof(T);
#pragma unroll
		for (int j = 0; j < 16; ++j)
			braw[j] = *reinterpret_cast<const T *>(Pc + (rb + (unsigned) (4 * j + lhi) * csb)); // L[r][4 j + lhi]: b operand of step k0 = 4 j (negated when used)
#pragma unroll
		for (int ct = 0; ct < 4; ++ct)
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int c = 16 * ct + Mfma<T>::row(q, lhi);
				acc[ct][q] = *reinterpret_cast<const T *>(Bc + (rb + (unsigned) (c < nr ? c : 0) * csb));
			}
	};
	auto tile_finish = [&](int r0, T (&braw)[16], acc_t (&acc)[4]) {
		const int r = r0 + l15;
		// (the U operands one k step ahead, and nothing moves across a step: left to itself the scheduler hoists all 64 LDS reads
		// of a tile, which with two tiles' loads in registers no longer fits two workgroups per CU)
		T a_cur[4], a_nxt[4];
#pragma unroll
		for (int ct = 0; ct < 4; ++ct)
			a_cur[ct] = Us[lhi * LUN_UP + 16 * ct + l15];
#pragma unroll
		for (int j = 0; j < 16; ++j) {
			if (j + 1 < 16) {
#pragma unroll
				for (int ct = 0; ct < 4; ++ct)
					a_nxt[ct] = Us[(4 * (j + 1) + lhi) * LUN_UP + 16 * ct + l15];
			}
			const T bneg = -braw[j];
#pragma unroll
			for (int ct = 0; ct < 4; ++ct)
				acc[ct] = Mfma<T>::run(a_cur[ct], bneg, acc[ct]);
#pragma unroll
			for (int ct = 0; ct < 4; ++ct)
				a_cur[ct] = a_nxt[ct];
			__builtin_amdgcn_sched_barrier(0);
		}
#pragma unroll
		for (int ct = 0; ct < 4; ++ct)
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int c = 16 * ct + Mfma<T>::row(q, lhi);
				if (r < m && c < nr)
					*reinterpret_cast<T *>(const_cast<char *>(Bc) + ((unsigned) r * (unsigned) sizeof(T) + (unsigned) c * csb)) = acc[ct][q];
			}
	};
	T b_a[16], b_b[16];
	acc_t acc_a[4], acc_b[4];
	int r_a = tile_r0(0), r_b = -1;
	if (r_a >= 0)
		tile_load(r_a, b_a, acc_a);
	if (solved) {
		// the interchange launch has solved the block (laswp_small_kernel with L00): `top` holds U = A01, already stored
#pragma unroll
		for (int i = 0; i < 16; ++i)
			Us[(16 * wave + i) * LUN_UP + lane] = lane < nr ? top[(size_t) lane * LUN_W + 16 * wave + i] : (T) 0;
		__syncthreads();
		LUN_TICK(0);
		LUN_TICK(1);
		LUN_TICK(2);
	} else {
	// ---- L00 (strictly lower part is used) and this thread's 16 rows of column c = lane of the interchanged top block
#pragma unroll
	for (int j = 0; j < LUN_W / 4; ++j) {
		const int k = wave + 4 * j;
		Ls[lane * LUN_LP + k] = P[lane + (idx_t) k * cs];
	}
	T u[16];
#pragma unroll
	for (int i = 0; i < 16; ++i)
		u[i] = lane < nr ? top[(size_t) lane * LUN_W + 16 * wave + i] : (T) 0;
	__syncthreads();
	LUN_TICK(0);
	// ---- U = L00^-1 top (factor.rs:104-109 -> triangular_solve.rs: forward substitution, unit diagonal)
#pragma unroll
	for (int kb = 0; kb < 4; ++kb) {
		if (wave == kb) {
#pragma unroll
			for (int k = 0; k < 16; ++k) {
#pragma unroll
				for (int i = k + 1; i < 16; ++i)
					u[i] = fh_fma(-Ls[(16 * kb + i) * LUN_LP + 16 * kb + k], u[k], u[i]);
			}
#pragma unroll
			for (int i = 0; i < 16; ++i)
				Us[(16 * kb + i) * LUN_UP + lane] = u[i];
		}
		__syncthreads();
		if (wave > kb) {
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				const T uk = Us[(16 * kb + k) * LUN_UP + lane];
#pragma unroll
				for (int i = 0; i < 16; ++i)
					u[i] = fh_fma(-Ls[(16 * wave + i) * LUN_LP + 16 * kb + k], uk, u[i]);
			}
		}
	}
	LUN_TICK(1);
	// ---- workgroup 0 stores A01 (lanes along the rows)
	if (blockIdx.x == 0) {
		for (int idx = tid; idx < LUN_W * nr; idx += 256) {
			const int d = idx & (LUN_W - 1), c = idx >> 6;
			B[d + (idx_t) c * cs] = Us[d * LUN_UP + c];
		}
	}
	LUN_TICK(2);
	}
	// ---- the tiles, two register sets in turn
#pragma unroll 1
	for (int idx = 0; r_a >= 0; idx += 2) {
		r_b = tile_r0(idx + 1);
		if (r_b >= 0)
			tile_load(r_b, b_b, acc_b);
		tile_finish(r_a, b_a, acc_a);
		if (r_b < 0)
			break;
		r_a = tile_r0(idx + 2);
		if (r_a >= 0)
			tile_load(r_a, b_a, acc_a);
		tile_finish(r_b, b_b, acc_b);
	}
	LUN_TICK(3);
#ifdef FH_LU_TIMING
	if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
		atomicAdd(&g_node_phase[7], 1ull);
#endif
}

// The same with the net permutation prepared ONCE (laswp_compose_list_kernel): the look-ahead driver applies a panel's
// 512 interchanges to up to four column ranges, each with thousands of workgroups -- rebuilding the permutation in every
// workgroup (512 dependent steps per entry) was most of those launches' time (profiles/r02_lu_timeline.txt).
__global__ __launch_bounds__(1024) void laswp_compose_list_kernel(const int *__restrict__ piv, int nt, int row_base, int *dst, int *src)
{
	__shared__ int s_piv[LASWP_SMALL_NT];
	const int tid = threadIdx.x;
	for (int j = tid; j < nt; j += 1024)
		s_piv[j] = piv[j] - row_base;
	__syncthreads();
	for (int e = tid; e < 2 * nt; e += 1024) {
		const int d = e < nt ? e : s_piv[e - nt];
		int pos = -1;
		if (e < nt || d >= nt) {
			pos = d;
			for (int j = nt - 1; j >= 0; --j) {
				const int pj = s_piv[j];
				pos = pos == j ? pj : (pos == pj ? j : pos);
			}
			if (pos == d)
				pos = -1; // row stays where it is
		}
		dst[e] = d;
		src[e] = pos;
	}
}

template <typename T>
__global__ __launch_bounds__(256) void laswp_list_kernel(T *B, idx_t rs, idx_t cs, int ncols, const int *__restrict__ dst,
							 const int *__restrict__ src, int ne)
{
	__shared__ int s_dst[2 * LASWP_SMALL_NT], s_src[2 * LASWP_SMALL_NT];
	__shared__ T tmp[LASWP_CC * 2 * LASWP_SMALL_NT];
	const int tid = threadIdx.x;
	for (int e = tid; e < ne; e += 256) {
		s_dst[e] = dst[e];
		s_src[e] = src[e];
	}
	__syncthreads();
	const int c0 = blockIdx.x * LASWP_CC;
	const int nc = min(LASWP_CC, ncols - c0);
	for (int idx0 = tid; idx0 < nc * ne; idx0 += 256 * 8) {
		T v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int idx = idx0 + u * 256;
			const bool in = idx < nc * ne;
			const int c = in ? idx / ne : 0, e = in ? idx - c * ne : 0;
			const int sr = in ? s_src[e] : -1;
			v[u] = B[sr >= 0 ? (idx_t) sr * rs + (idx_t) (c0 + c) * cs : (idx_t) 0]; // unconditional load, clamped address
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int idx = idx0 + u * 256;
			if (idx < nc * ne)
				tmp[idx] = v[u];
		}
	}
	__syncthreads();
	for (int idx = tid; idx < nc * ne; idx += 256) {
		const int c = idx / ne, e = idx - c * ne;
		if (s_src[e] >= 0)
			B[(idx_t) s_dst[e] * rs + (idx_t) (c0 + c) * cs] = tmp[idx];
	}
}

struct LaswpList {
	int *dst = nullptr, *src = nullptr; // 2 * nt entries each (device)
	int nt = 0;
};
static void laswp_compose_list(const int *piv, int nt, int row_base, LaswpList &l)
{
	FH_CHECK(nt <= LASWP_SMALL_NT, "laswp list: too many interchanges");
	l.nt = nt;
	if (nt == 0)
		return;
	hipLaunchKernelGGL(laswp_compose_list_kernel, dim3(1), dim3(1024), 0, ctx().stream, piv, nt, row_base, l.dst, l.src);
	FH_HIP(hipGetLastError());
}
template <typename T> static void laswp_list_dev(MatV<T> B, const LaswpList &l)
{
	if (B.nrows == 0 || B.ncols == 0 || l.nt == 0)
		return;
	hipLaunchKernelGGL(laswp_list_kernel<T>, dim3((unsigned) ((B.ncols + LASWP_CC - 1) / LASWP_CC)), dim3(256), 0, ctx().stream, B.p, B.rs, B.cs,
			   (int) B.ncols, l.dst, l.src, 2 * l.nt);
	FH_HIP(hipGetLastError());
}

// Applies the transpositions (j <-> piv[j] - row_base), j < nt, to all columns of B (B's row 0 is the
// row the first transposition refers to).
template <typename T> static void laswp_dev(MatV<T> B, const int *piv, int nt, int row_base, T *top_out = nullptr, const T *L00 = nullptr, idx_t lcs = 0)
{
	if (B.nrows == 0 || B.ncols == 0 || nt == 0)
		return;
	const int nrows = (int) B.nrows;
	hipStream_t s = ctx().stream;
	if (nt <= LASWP_SMALL_NT) {
		hipLaunchKernelGGL(laswp_small_kernel<T>, dim3((unsigned) ((B.ncols + LASWP_CC - 1) / LASWP_CC)), dim3(256), 0, s, B.p,
				   B.rs, B.cs, nrows, (int) B.ncols, piv, nt, row_base, top_out, (nt == LUN_W_HOST && top_out) ? L00 : (const T *) nullptr, lcs);
		FH_HIP(hipGetLastError());
		return;
	}
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
	xwg_u64 *gran;	    // [LU2_NSLOT][LU2_GMAX][LU2_GSLOT] tagged granules (zeroed once per factorization)
	xwg_u64 *gran_diag; // [LU2_NSLOT][2 * LU_W]
	xwg_u64 epoch_base; // epochs consumed by earlier leaf launches of this factorization
	unsigned char *wws = nullptr; // LW_WS_BYTES of getrf_wpanel_kernel's exchange records (zeroed once per factorization)
	T *ttop = nullptr;	      // LUN_W x LU_FLAT_MAXW scalars: the interchanged top blocks (64 x 64 each, one per 64-column group right of a leaf)
				      // that lu_node64_kernel reads -- up to LU_FLAT_MAXW / 64 - 1 groups from a flat panel or a staged last leaf
	int *status;
	bool general = false;		 // every leaf on the non-cooperative path (rerun after an exchange timeout, debug switch)
};

// rows per workgroup of the cooperative kernel for a leaf of w columns (registers: RPT x W scalars per thread)
// Leaf shapes: W columns x RPT rows per thread with W * RPT = 64 (fp64) or 128 (fp32) register scalars per thread.
// Every workgroup of the cooperative kernel must be resident (one 512-thread workgroup per CU at this register
// footprint), so tall panels trade leaf width for rows per workgroup: 64 columns up to 512 (1024) rows per
// workgroup, ... 8 columns up to 4096 (8192).
template <typename T> static int leaf_rpt(int w) { return (sizeof(T) == 8 ? 64 : 128) / w; }
template <typename T> static int leaf_rows_per_wg(int w) { return LU2_NT * leaf_rpt<T>(w); }

static int resident_workgroups()
{
	Ctx &c = ctx();
	if (c.la_state > 0 && c.stream == c.la_panel)
		return c.la_panel_cus;
	static const int ncu = [&]() { // initialised once, thread safe (function-local static)
		hipDeviceProp_t prop;
		FH_HIP(hipGetDeviceProperties(&prop, c.device));
		return prop.multiProcessorCount;
	}();
	return ncu;
}

// widest leaf whose workgroups are all resident for a panel of m rows when `cap` workgroups fit the device at once;
// 0 if not even the narrowest shape fits (pure host logic: faer_hip_debug_lu_leaf_width)
int lu_leaf_width(idx_t m, int elem_bytes, int cap)
{
	for (int w = LU_W; w >= 8; w /= 2) {
		const idx_t rows = (idx_t) LU2_NT * ((elem_bytes == 8 ? 64 : 128) / w);
		if ((m + rows - 1) / rows <= (idx_t) cap)
			return w;
	}
	return 0;
}
// The distributed driver factors the look-ahead panel on the CU-masked panel stream only when its cooperative leaves keep,
// on those `panel_cus` CUs, the width they would have on the whole chip (pure host logic: faer_hip_debug_dist_two_streams_ok)
bool dist_two_streams_ok(idx_t panel_rows, int elem_bytes, int panel_cus, int all_cus)
{
	if (panel_rows <= 0)
		return true;
	const int w = lu_leaf_width(panel_rows, elem_bytes, panel_cus);
	return w > 0 && w == lu_leaf_width(panel_rows, elem_bytes, all_cus);
}
// 0: the panel is taller than the cooperative kernel can keep resident -> the non-cooperative leaf (getrf_leaf_general)
template <typename T> static int leaf_width_for(idx_t m) { return lu_leaf_width(m, (int) sizeof(T), resident_workgroups()); }

// faer_hip_partial_piv_lu_lend_copy: a copy of the NEXT LU's input, with the shape and element size it was made for.  The
// thread's next getrf_dev takes it at ENTRY, before anything can throw (LentGuard), whatever its size or type, and uses it
// only if shape and type are its own: a stale pointer never survives a call and is never read with another extent.
struct LuLent {
	const void *p = nullptr;
	idx_t nrows = 0, ncols = 0;
	int elem_bytes = 0;
};
static thread_local LuLent t_lu_lent;
void lu_lend_copy(const void *p, idx_t nrows, idx_t ncols, int elem_bytes) { t_lu_lent = LuLent{p, nrows, ncols, elem_bytes}; }
// Above this size the library does not copy A on its own (one read + one write of A per call: VERDICT r04 item 5); below
// it the copy costs < 1 % of the factorization (512 MiB: 0.25 ms against the 30 ms of an N = 8192 fp64 LU) and a caller of
// the reference's FFI surface, which has no way to lend anything, still gets a completed factorization after a timeout.
static constexpr size_t LU_AUTO_BACKUP_BYTES = (size_t) 512 << 20;
static std::atomic<int> g_lu_force_general{0}; // faer_hip_debug_lu_force_general: tests run the whole suite of shapes on the fallback
void lu_force_general(int on) { g_lu_force_general.store(on); }
// faer_hip_debug_lu_plan: the switch-over points of the look-ahead driver (0 = the tuned default), so that tests can drive its three
// phases -- pipelined bulk-bound steps, plain bulk-bound steps, staged panel-bound steps -- and the transitions between them at
// N ~ 2-6 k instead of only at the full benchmark size
static std::atomic<long> g_lu_plan_nb2_from{0}, g_lu_plan_pipe_from{0}, g_lu_plan_la_min{0};
void lu_debug_plan(long nb2_from, long pipe_from, long la_min)
{
	g_lu_plan_nb2_from.store(nb2_from);
	g_lu_plan_pipe_from.store(pipe_from);
	g_lu_plan_la_min.store(la_min);
}

// ------------------------------------------------------------------------------------------------
// Non-cooperative leaf: the same unblocked elimination (factor.rs:19-67: first largest |a| of the column, interchange,
// reciprocal scaling, rank-1 update with fma(l, -u, dst)) as three plain launches per column -- no workgroup has to be
// resident together with any other, so it takes ANY number of rows and cannot time out.  It is the fallback of the
// cooperative panel kernel: panels taller than its residency limit (the reference has no such limit,
// lu/partial_pivoting/factor.rs:234-295) and the rerun after an exchange timeout (GPU shared with other work).
// ~3 launches x 64 columns per leaf: slow, by design never on the fast path.
// ------------------------------------------------------------------------------------------------
template <typename T> struct LugArgs {
	T *P;
	idx_t rs, cs;
	int m, w, j;
	int *piv;
	int row_base;
	double *pv; // per workgroup: best |a| ...
	int *pr;    // ... and its row
	int nwg;
	T *urow; // row j after the interchange (w entries)
};

template <typename T> __global__ __launch_bounds__(256) void lug_argmax_kernel(const LugArgs<T> a)
{
	__shared__ double sv[256];
	__shared__ int sr[256];
	const int tid = threadIdx.x;
	double bv = 0.0;
	int br = INT_MAX;
	// ascending rows per thread, strictly greater: the first largest of the thread's rows
	for (long i = (long) a.j + (long) blockIdx.x * 256 + tid; i < a.m; i += (long) gridDim.x * 256) {
		const double av = fabs((double) a.P[i * a.rs + (idx_t) a.j * a.cs]);
		if (av > bv) {
			bv = av;
			br = (int) i;
		}
	}
	sv[tid] = bv;
	sr[tid] = br;
	__syncthreads();
	for (int o = 128; o > 0; o >>= 1) {
		if (tid < o && better(sv[tid + o], sr[tid + o], sv[tid], sr[tid])) {
			sv[tid] = sv[tid + o];
			sr[tid] = sr[tid + o];
		}
		__syncthreads();
	}
	if (tid == 0) {
		a.pv[blockIdx.x] = sv[0];
		a.pr[blockIdx.x] = sr[0];
	}
}

template <typename T> __global__ __launch_bounds__(256) void lug_pivot_kernel(const LugArgs<T> a)
{
	__shared__ double sv[256];
	__shared__ int sr[256];
	__shared__ int s_p;
	const int tid = threadIdx.x;
	double bv = 0.0;
	int br = INT_MAX;
	for (int g = tid; g < a.nwg; g += 256)
		if (better(a.pv[g], a.pr[g], bv, br)) {
			bv = a.pv[g];
			br = a.pr[g];
		}
	sv[tid] = bv;
	sr[tid] = br;
	__syncthreads();
	for (int o = 128; o > 0; o >>= 1) {
		if (tid < o && better(sv[tid + o], sr[tid + o], sv[tid], sr[tid])) {
			sv[tid] = sv[tid + o];
			sr[tid] = sr[tid + o];
		}
		__syncthreads();
	}
	if (tid == 0) {
		const int p = (sv[0] > 0.0 && sr[0] != INT_MAX) ? sr[0] : a.j; // zero / NaN-only column: no interchange
		s_p = p;
		a.piv[a.j] = a.row_base + p;
	}
	__syncthreads();
	const int p = s_p;
	for (int c = tid; c < a.w; c += 256) {
		T *xj = a.P + (idx_t) a.j * a.rs + (idx_t) c * a.cs, *xp = a.P + (idx_t) p * a.rs + (idx_t) c * a.cs;
		const T vj = *xj, vp = *xp;
		if (p != a.j) {
			*xj = vp;
			*xp = vj;
		}
		a.urow[c] = vp; // row j after the interchange (p == j: vp == vj)
	}
}

template <typename T> __global__ __launch_bounds__(256) void lug_update_kernel(const LugArgs<T> a)
{
	__shared__ T su[LU_WMAX];
	const int tid = threadIdx.x;
	if (tid < a.w)
		su[tid] = a.urow[tid];
	__syncthreads();
	const T inv = (T) 1 / su[a.j];
	for (long i = (long) a.j + 1 + (long) blockIdx.x * 256 + tid; i < a.m; i += (long) gridDim.x * 256) {
		T *row = a.P + i * a.rs;
		const T l = row[(idx_t) a.j * a.cs] * inv;
		row[(idx_t) a.j * a.cs] = l;
		for (int c = a.j + 1; c < a.w; ++c)
			row[(idx_t) c * a.cs] = fh_fma(l, -su[c], row[(idx_t) c * a.cs]); // rank_update_imp: fma(l_i, -u_c, dst)
	}
}

template <typename T> static void getrf_leaf_general(MatV<T> P, int col0, int row_base, LuWork<T> &wk)
{
	const idx_t m = P.nrows;
	const int w = (int) P.ncols;
	FH_CHECK(w <= LU_WMAX, "getrf leaf: panel too wide");
	hipStream_t s = ctx().stream;
	int nwg = (int) ((m + 255) / 256);
	if (nwg > 1024)
		nwg = 1024;
	if (nwg < 1)
		nwg = 1;
	// (released on return while the launches may still be queued: the pool hands a buffer back to the SAME stream only)
	Scratch pvb((size_t) nwg * sizeof(double)), prb((size_t) nwg * sizeof(int)), ub((size_t) LU_WMAX * sizeof(T));
	LugArgs<T> a;
	a.P = P.p;
	a.rs = P.rs;
	a.cs = P.cs;
	a.m = (int) m;
	a.w = w;
	a.piv = wk.piv + col0;
	a.row_base = row_base;
	a.pv = pvb.as<double>();
	a.pr = prb.as<int>();
	a.nwg = nwg;
	a.urow = ub.as<T>();
	const int steps = w < (int) m ? w : (int) m;
	for (int j = 0; j < steps; ++j) {
		a.j = j;
		hipLaunchKernelGGL(lug_argmax_kernel<T>, dim3(nwg), dim3(256), 0, s, a);
		hipLaunchKernelGGL(lug_pivot_kernel<T>, dim3(1), dim3(256), 0, s, a);
		if (j + 1 < (int) m)
			hipLaunchKernelGGL(lug_update_kernel<T>, dim3(nwg), dim3(256), 0, s, a);
	}
	FH_HIP(hipGetLastError());
}

// (A second-generation panel kernel -- one hop per column on the dependent chain, logical row indices instead of physical
// interchanges -- lived here through round 2: N = 16384 took 145.1 ms with it against 141.0 ms, profiles/r02_exp_lu_panel.txt;
// removed in round 3 as VERDICT r02 asked, it is in the history of this file.)
template <typename T, int W> static void launch_leaf(int G, hipStream_t s, const Panel2Args<T> &a)
{
	hipLaunchKernelGGL((getrf_panel2_kernel<T, W, (sizeof(T) == 8 ? 64 : 128) / W>), dim3(G), dim3(LU2_NT), 0, s, a);
}

// timing build: 16 device words that receive the per-phase tick sums of the panel kernel (lu_dump_timing)
static unsigned long long *g_lu_phase = nullptr;
static unsigned long long *lu_phase_words()
{
#ifdef FH_LU_TIMING
	if (!g_lu_phase) {
		FH_HIP(hipMalloc(&g_lu_phase, 16 * sizeof(unsigned long long)));
		FH_HIP(hipMemset(g_lu_phase, 0, 16 * sizeof(unsigned long long)));
	}
#endif
	return g_lu_phase;
}
void lu_dump_timing()
{
#ifdef FH_LU_TIMING
	{
		unsigned long long d[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		FH_HIP(hipDeviceSynchronize());
		FH_HIP(hipMemcpyFromSymbol(d, HIP_SYMBOL(g_node_phase), sizeof(d)));
		FH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_node_phase), z, sizeof(z)));
		if (d[7])
			fprintf(stderr, "node kernel phases (s_memtime ticks per launch, workgroup (0,0) wave 0, %llu launches): prefetch + L00 + top in %.0f | solve %.0f | A01 store %.0f | tiles %.0f\n",
				d[7], (double) d[0] / d[7], (double) d[1] / d[7], (double) d[2] / d[7], (double) d[3] / d[7]);
	}
	if (!g_lu_phase)
		return;
	unsigned long long h[16];
	FH_HIP(hipDeviceSynchronize());
	FH_HIP(hipMemcpy(h, g_lu_phase, sizeof(h), hipMemcpyDeviceToHost));
	FH_HIP(hipMemset(g_lu_phase, 0, sizeof(h)));
	const double n = h[8] ? (double) h[8] : 1.0;
	fprintf(stderr,
		"wpanel phases (s_memtime ticks per column, wg 0 / wave 0, %llu columns): sweep %.0f | barrier+result %.0f | relabel+scale+col J+1 %.0f | candidate+combine+publish %.0f | "
		"row record wait+correct %.0f | rank-1 update %.0f || per 8 columns: rotate %.0f || header sweeps checked per column %.2f\n",
		h[8], h[0] / n, h[7] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, 8.0 * h[5] / n, h[6] / n);
	fprintf(stderr, "wpanel: ticks per group of 8 columns and leaf, groups 0..6: %.0f %.0f %.0f %.0f %.0f %.0f %.0f\n", 64.0 * h[9] / n, 64.0 * h[10] / n, 64.0 * h[11] / n,
		64.0 * h[12] / n, 64.0 * h[13] / n, 64.0 * h[14] / n, 64.0 * h[15] / n);
#endif
}

template <typename T> static void getrf_leaf(MatV<T> P, int col0, int row_base, LuWork<T> &wk)
{
	const idx_t m = P.nrows;
	const int w = (int) P.ncols;
	int lw = leaf_width_for<T>(m);
	if (wk.general || lw == 0 || g_lu_force_general.load()) {
		getrf_leaf_general<T>(P, col0, row_base, wk);
		return;
	}
	FH_CHECK(w <= lw, "getrf leaf: panel too wide");
	// a panel of at most 512 (fp64) / 1024 (fp32) rows: one workgroup holds it in registers, nothing to exchange (lu_small_leaf.h)
	constexpr idx_t SMALL_ROWS = sizeof(T) == 8 ? 512 : 1024;
	if (w <= LW_W && w <= m && m <= SMALL_ROWS) {
		hipStream_t s = ctx().stream;
		ProfScope prof(1, (double) w);
		if (m <= 256)
			hipLaunchKernelGGL((getrf_small_leaf_kernel<T, 4>), dim3(1), dim3(256), 0, s, P.p, P.rs, P.cs, (int) m, w, wk.piv + col0, row_base);
		else if (m <= 512)
			hipLaunchKernelGGL((getrf_small_leaf_kernel<T, 8>), dim3(1), dim3(512), 0, s, P.p, P.rs, P.cs, (int) m, w, wk.piv + col0, row_base);
		else
			hipLaunchKernelGGL((getrf_small_leaf_kernel<T, (sizeof(T) == 8 ? 8 : 16)>), dim3(1), dim3(sizeof(T) == 8 ? 512 : 1024), 0, s, P.p, P.rs, P.cs,
					   (int) m, w, wk.piv + col0, row_base);
		FH_HIP(hipGetLastError());
		return;
	}
	if (lw == LU_W && w <= m && wk.wws) {
		// round-4 kernel (lu_wpanel.h): one exchange per column.  64 x RPT rows per wavefront; four wavefronts per workgroup
		// (one per SIMD) while that many workgroups are resident, else eight
		constexpr int RPT = sizeof(T) == 8 ? 1 : 2;
		const int cap = resident_workgroups();
		const int g4 = (int) ((m + 256 * RPT - 1) / (256 * RPT)), g8 = (int) ((m + 512 * RPT - 1) / (512 * RPT));
		const bool four = g4 <= cap && g4 <= LW_GMAX;
		if (four || g8 <= LW_GMAX) { // (g8 <= cap: that is what lw == LU_W says; more than LW_GMAX workgroups: the old kernel)
			WPanelArgs<T> a;
			a.P = P.p;
			a.rs = P.rs;
			a.cs = P.cs;
			a.m = (int) m;
			a.w = w;
			a.piv = wk.piv + col0;
			a.row_base = row_base;
			a.ws = wk.wws;
			a.epoch_base = (unsigned) wk.epoch_base;
			a.status = wk.status;
			a.phase = lu_phase_words();
			hipStream_t s = ctx().stream;
			{
				// the exchange needs all workgroups resident: at least one per compute unit must fit (registers, LDS)
				static const int occ = []() {
					int o4 = 0, o8 = 0;
					if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o4, getrf_wpanel_kernel<T, RPT, 4, 2>, 256, 0) != hipSuccess)
						o4 = 0;
					if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o8, getrf_wpanel_kernel<T, RPT, 8, 2>, 512, 0) != hipSuccess)
						o8 = 0;
					return o4 < o8 ? o4 : o8;
				}();
				FH_CHECK(occ >= 1, "partial_piv_lu: the cooperative panel kernel does not fit a compute unit");
			}
			ProfScope prof(1, (double) w);
			if (four)
				hipLaunchKernelGGL((getrf_wpanel_kernel<T, RPT, 4, 2>), dim3(g4), dim3(256), 0, s, a);
			else
				hipLaunchKernelGGL((getrf_wpanel_kernel<T, RPT, 8, 2>), dim3(g8), dim3(512), 0, s, a);
			FH_HIP(hipGetLastError());
			wk.epoch_base += (xwg_u64) (((w < (int) m ? w : (int) m) + 7) & ~7); // the kernel runs whole groups of 8 column steps
			return;
		}
	}
	while (lw / 2 >= w && lw > 8)
		lw /= 2; // a narrower panel fits the narrower (taller) shape just as well: fewer workgroups to synchronise
	const int R = leaf_rows_per_wg<T>(lw);
	int G = (int) ((m + R - 1) / R);
	if (G < 1)
		G = 1;
	FH_CHECK(G <= LU2_GMAX, "partial_piv_lu: too many workgroups in the cooperative panel kernel");
	Panel2Args<T> a;
	a.P = P.p;
	a.rs = P.rs;
	a.cs = P.cs;
	a.m = (int) m;
	a.w = w;
	a.piv = wk.piv + col0;
	a.row_base = row_base;
	a.gran = wk.gran;
	a.gran_diag = wk.gran_diag;
	a.epoch_base = wk.epoch_base;
	a.status = wk.status;
	hipStream_t s = ctx().stream;
	switch (lw) {
	case 64:
		launch_leaf<T, 64>(G, s, a);
		break;
	case 32:
		launch_leaf<T, 32>(G, s, a);
		break;
	case 16:
		launch_leaf<T, 16>(G, s, a);
		break;
	default:
		launch_leaf<T, 8>(G, s, a);
		break;
	}
	FH_HIP(hipGetLastError());
	const int steps = w < (int) m ? w : (int) m;
	if (G > 1)
		wk.epoch_base += (xwg_u64) steps;
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
	{
		const int lw = leaf_width_for<T>(m);
		const bool gen = wk.general || lw == 0 || g_lu_force_general.load(); // non-cooperative leaves: always LU_W columns
		if (n <= (gen ? LU_W : lw)) {
			getrf_leaf<T>(P, col0, row_base, wk);
			return;
		}
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
	MatV<T> A00 = P.sub(0, 0, bs, bs), A01 = P.sub(0, bs, bs, n - bs), A10 = P.sub(bs, 0, m - bs, bs),
		A11 = P.sub(bs, bs, m - bs, n - bs);
	if (bs == LUN_W && n - bs <= LUN_W && P.rs == 1 && wk.ttop && lu_node_offsets_ok<T>(m, P.cs)) {
		// the 128-column nodes: interchanges (+ a copy of the interchanged top block), then solve and product in one launch
		laswp_dev<T>(right, wk.piv + col0, (int) bs, row_base, wk.ttop, (const T *) P.p, P.cs);
		const idx_t below = m - bs;
		const unsigned nwg = below > 0 ? (unsigned) ((below + LUN_ROWS - 1) / LUN_ROWS) : 1u;
		hipLaunchKernelGGL(lu_node64_kernel<T>, dim3(nwg), dim3(256), 0, ctx().stream, P.p, P.cs, (int) m, (int) (n - bs), (const T *) wk.ttop, 1);
		FH_HIP(hipGetLastError());
	} else {
		laswp_dev<T>(right, wk.piv + col0, (int) bs, row_base);
		trsm_lower_dev<T>(A00.c(), true, A01);
		gemm_dev<T>(A11, DST_FULL, true, A10.c(), A01.c(), (T) -1);
	}
	getrf_rec<T>(A11, col0 + (int) bs, row_base + (int) bs, wk);
	// the right half's transpositions act on the rows below bs of the left half (factor.rs:127-185)
	laswp_dev<T>(A10, wk.piv + col0 + bs, (int) (n - bs < m - bs ? n - bs : m - bs), row_base + (int) bs);
}

// ------------------------------------------------------------------------------------------------
// Flat right-looking panel (round 5): the panels of the look-ahead driver below.  getrf_rec's binary recursion puts, per 512
// columns, 4 fused 64-column nodes, 2 nodes of 128 (interchanges, substitution leaf, K = 128 product) and one of 256
// (interchanges, a three-launch solve, K = 256 product) on the panel stream's dependent chain -- every one of them bound by its
// own latency on the 32 reserved CUs.  Here every 64-column leaf is applied at once to ALL panel columns right of it: one
// interchange launch + ONE launch of the fused node kernel (its column groups along blockIdx.y), K = 64.  Same operations per
// entry as the recursion (factor.rs:98-117 with bs = 64 at every level), regrouped; same pivots.
//   after_leaf(j): called behind leaf j and the interchanges it makes on the panel's earlier columns `cols_from .. 64 j`
//   (the staged driver starts the bulk stream's work on a finished part there); `left_from(j)`: first panel column leaf j's
//   interchanges are applied to at once (the staged driver defers the ones that reach into an earlier part).
// ------------------------------------------------------------------------------------------------
template <typename T> static bool flat_panel_ok(MatV<T> P, const LuWork<T> &wk)
{
	return P.rs == 1 && wk.ttop && P.ncols % LUN_W == 0 && P.ncols <= LU_FLAT_MAXW && P.nrows >= P.ncols && lu_node_offsets_ok<T>(P.nrows, P.cs) && !wk.general &&
	       !g_lu_force_general.load() && leaf_width_for<T>(P.nrows) == LU_W;
}
template <typename T, typename LeftFrom, typename AfterLeaf>
static void getrf_panel_flat(MatV<T> P, int col0, int row_base, LuWork<T> &wk, LeftFrom left_from, AfterLeaf after_leaf)
{
	const idx_t m = P.nrows, w = P.ncols;
	const idx_t nl = w / LUN_W;
	for (idx_t j = 0; j < nl; ++j) {
		const idx_t c = j * LUN_W;
		getrf_leaf<T>(P.sub(c, c, m - c, LUN_W), col0 + (int) c, row_base + (int) c, wk);
		// the leaf's transpositions on the panel columns to its left (factor.rs:127-185)
		const idx_t lf = left_from(j);
		if (c > lf)
			laswp_dev<T>(P.sub(c, lf, m - c, c - lf), wk.piv + col0 + c, (int) LUN_W, row_base + (int) c);
		after_leaf(j);
		const idx_t nr = w - c - LUN_W;
		if (nr > 0) {
			// ... and on the columns to its right, which then take the leaf's update: A01 <- L00^-1 A01, A11 -= A10 A01
			FH_CHECK(nr <= LU_FLAT_MAXW - LUN_W, "lu: more column groups than LuWork::ttop holds");
			laswp_dev<T>(P.sub(c, c + LUN_W, m - c, nr), wk.piv + col0 + c, (int) LUN_W, row_base + (int) c, wk.ttop, (const T *) (P.p + c + c * P.cs), P.cs);
			const idx_t below = m - c - LUN_W;
			unsigned nwg = below > 0 ? (unsigned) ((below + LUN_ROWS - 1) / LUN_ROWS) : 1u;
			const unsigned ncg = (unsigned) ((nr + LUN_W - 1) / LUN_W);
			// (every workgroup solves the 64 x 64 system of its column group before it updates: with many row chunks and several
			// column groups a workgroup takes several chunks -- two workgroups per compute unit of the stream in all)
			// (measured: 2 workgroups per CU 88.1-88.2 ms, 1: 88.7, 3-4: 88.5, one chunk per workgroup 88.9-89.2 -- profiles/r05_exp_lu_driver.txt)
			const unsigned cap = (unsigned) (2 * ctx().stream_cus()) / ncg;
			if (nwg > cap)
				nwg = cap < 1u ? 1u : cap;
			hipLaunchKernelGGL(lu_node64_kernel<T>, dim3(nwg, ncg), dim3(256), 0, ctx().stream, P.p + c + c * P.cs, P.cs, (int) (m - c), (int) nr,
					   (const T *) wk.ttop, 1);
			FH_HIP(hipGetLastError());
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Right-looking driver with look-ahead for large matrices (same idea as potrf.hip): steps of LU_LA_NB columns,
//     [panel stream]  P_k : recursive panel factorization above on the m_k x nb panel (cooperative leaves; <= 16
//                     workgroups for 16384 rows, they fit the CUs reserved for this stream)
//     [bulk stream]   columns of panel k+1 first: interchanges, U = L_kk^-1 A_k,k+1, A_{>k,k+1} -= L_{>k,k} U
//                     -> releases P_{k+1} on the panel stream
//     [bulk stream]   the same three operations on all remaining columns, then the interchanges of the
//                     columns left of the panel
// so the chain of per-column pivot exchanges (the serial part of LU) overlaps with the trailing GEMMs.
// Same operations per entry as the reference's recursion (factor.rs:68-187) regrouped by block columns.
// ------------------------------------------------------------------------------------------------
constexpr idx_t LU_LA_NB = 512;

template <typename T> static void getrf_lookahead(MatV<T> A, LuWork<T> &wk, hipStream_t caller)
{
	Ctx &c = ctx();
	const idx_t m = A.nrows, n = A.ncols; // n <= m
	// Step plan: LU_LA_NB-column panels while the bulk stream is the critical one (wide steps keep its trailing products at
	// K = 512); once the panel chain is critical (fewer than `nb2_from` rows below the panel) the steps narrow to `nb2` columns:
	// the top node of a panel's recursion -- interchanges, a solve against half the panel, a product on the reserved CUs -- is
	// then done by the (idle) bulk stream as part of its update, and what sits between two panels is half as long.
	// (measured: profiles/r05_exp_lu_driver.txt -- N = 16384 92.4 -> 90.5 ms, N = 8192 34.0 -> 30.8 ms; a third, 128-column level
	// and other switch-over points change nothing)
	// (second half of round 5, with the leaf-wise hand-over: switch-over at 12288 / 11264 / 10240 / 9216 / 8192 rows: 89.3 / 88.0 / 86.8-86.9 /
	// 86.5-86.6 / 86.8 ms -- 9216, for the bulk_bound threshold below as well)
	constexpr idx_t LU_LA_NB2 = 256;
	const idx_t LU_LA_NB2_FROM = g_lu_plan_nb2_from.load() > 0 ? (idx_t) g_lu_plan_nb2_from.load() : 9216;
	std::vector<idx_t> J;
	J.push_back(0);
	while (J.back() < n) {
		const idx_t j0 = J.back();
		idx_t w = m - j0 - LU_LA_NB >= LU_LA_NB2_FROM ? LU_LA_NB : LU_LA_NB2;
		if (w > n - j0)
			w = n - j0;
		J.push_back(j0 + w);
	}
	const idx_t nsteps = (idx_t) J.size() - 1;
	auto Jat = [&](idx_t k) { return k < (idx_t) J.size() ? J[(size_t) k] : n; };
	c.reset_events();
	hipEvent_t e0 = c.next_event();
	FH_HIP(hipEventRecord(e0, caller));
	stream_wait(c.la_bulk, e0);
	stream_wait(c.la_panel, e0);
	hipEvent_t ev_panel;
	// net row permutation of a panel's interchanges, composed once (laswp_compose_list_kernel) and shared by all the interchange
	// launches for it.  Round 4: composed on the BULK stream at the start of the step that applies it (22 us that used to sit
	// on the panel stream's chain).
	static_assert(LU_LA_NB <= LASWP_SMALL_NT, "panel interchange list");
	// (a ring of lists: in the pipelined steps below the side stream composes step k + 1's list while the bulk stream still
	// applies step k's to the columns left of the panel)
	constexpr idx_t NLIST = 4;
	Scratch listb((size_t) NLIST * 2 * 2 * LU_LA_NB * sizeof(int));
	LaswpList full;
	auto use_list = [&](idx_t k) {
		full.dst = listb.as<int>() + (size_t) (k % NLIST) * 4 * LU_LA_NB;
		full.src = full.dst + 2 * LU_LA_NB;
	};
	use_list(0);
	// Lending the panel stream's idle CUs to the big product of a step.  While the bulk stream is the critical
	// one the panel stream finishes panel k + 1 before the bulk stream has finished product k and then idles.  The product's tiles
	// are handed out through per-XCD counters (GemmExtra::ticket); a HELPER launch of the same product, queued on the panel stream
	// right behind panel k + 1, takes tiles on the reserved CUs while more than a margin remain, and the consumers of the product
	// wait for both launches.  Same arithmetic per tile whoever computes it: results do not depend on the split.
	// MEASURED, OFF BY DEFAULT (faer_hip_debug_lend_cus(1) turns it on; profiles/r06_exp_lend.txt): the helpers take 8-25 % of a
	// product's tiles in the pipelined steps, the product ends earlier -- and N = 16384 takes 87.2-87.4 ms instead of 86.9-87.0: the
	// reserved CUs are where the side stream's interchange / solve chains run undisturbed while the panel stream idles, and with
	// helpers on them those chains (which the next product waits for) queue behind 130 us tiles on every CU of the chip.
	constexpr int LU_TICKETS = 64;
	Scratch tickb((size_t) LU_TICKETS * 8 * sizeof(int));
	FH_HIP(hipMemsetAsync(tickb.p, 0, (size_t) LU_TICKETS * 8 * sizeof(int), caller)); // (before e0: older than everything below)
	int tick_used = 0;
	struct HelpJob {
		bool on = false;
		MatV<T> C;
		MatV<const T> A, B;
		int *ticket = nullptr;
		int wgs = 0;
		hipEvent_t ev_in = nullptr;
		idx_t c0 = 0; // first column of the helped product
	} help;
	hipEvent_t ev_help_prev = nullptr; // the previous step's helper launch (panel stream)
	idx_t help_prev_c0 = 0;
	const bool lend = (g_lend_cus.load() & 1) != 0;
	// the big product of a step on the bulk stream; `helped`: a helper launch will follow on the panel stream
	auto big_product = [&](MatV<T> Cd, MatV<const T> Ad, MatV<const T> Bd, bool helped, idx_t c0) {
		const idx_t tiles = ((Cd.nrows + 127) / 128) * ((Cd.ncols + 127) / 128);
		if (!helped || !lend || tiles < 2048 || tick_used >= LU_TICKETS) {
			gemm_dev<T>(Cd, DST_FULL, true, Ad, Bd, (T) -1);
			return;
		}
		GemmExtra<T> ex;
		ex.ticket = tickb.as<int>() + 8 * (tick_used++);
		help.on = true;
		help.C = Cd;
		help.A = Ad;
		help.B = Bd;
		help.ticket = ex.ticket;
		help.wgs = (int) (tiles / 8 + 64);
		help.c0 = c0;
		help.ev_in = c.next_event();
		FH_HIP(hipEventRecord(help.ev_in, c.la_bulk)); // (the product's inputs are final: everything older on the bulk stream)
		gemm_dev<T>(Cd, DST_FULL, true, Ad, Bd, (T) -1, &ex);
	};
	// Pipelined steps (while the bulk stream is the critical one).  Between two trailing products the bulk stream used to run
	// the chain "interchanges, solve, product" for the next panel's columns (~0.3 ms busy) and then wait for the side stream's
	// chain on the far columns (~0.25 ms idle): ~0.55 ms per step in which most of the chip does nothing.  The columns right of
	// the panel are now cut at a fixed column jA into a near group A (with the next panel's columns) and a far group B, and ALL
	// chains run on the side stream one product ahead: the chain of step k + 1 on A runs beside the product of step k on B, the
	// chain on B beside the product of step k + 1 on A.  The bulk stream issues nothing but the two products per step.
	// It pays while a product is much longer than a chain: down to 13312 rows below the panel (N = 16384: the first six steps;
	// 89.7-90.1 -> 88.2-88.5 ms; down to 10240 rows 89.0 -- later the chains, which run 2-4 x slower among a product's workgroups,
	// no longer fit beside half a product: profiles/r05_exp_lu_driver.txt).
	const idx_t LU_PIPE_FROM = g_lu_plan_pipe_from.load() > 0 ? (idx_t) g_lu_plan_pipe_from.load() : 13312;
	idx_t jA = 0;
	{
		// the boundary: the middle of the columns right of the next panel, taken at the middle step of this phase
		idx_t kend = 0;
		while (kend + 1 < nsteps && m - J[(size_t) kend + 1] >= LU_PIPE_FROM - 3072)
			++kend;
		const idx_t jm = Jat(kend / 2 + 2);
		jA = (jm + n) / 2 / LU_LA_NB * LU_LA_NB;
	}
	hipEvent_t ev_GA_prev = nullptr, ev_GB_prev = nullptr; // the previous step's products on the two groups (bulk stream)
	c.qr_side_streams();
	hipStream_t side = c.qr_side[0];
	// remaining rows from which the bulk stream is the critical one (mode 2 below); fewer: the panel chain is
	auto bulk_bound = [&](idx_t rows_below) { return rows_below >= LU_LA_NB2_FROM; };
	// one panel on the current stream: flat right-looking (getrf_panel_flat) where its shape allows, else the recursion
	auto panel = [&](MatV<T> P, idx_t j) {
		if (flat_panel_ok<T>(P, wk))
			getrf_panel_flat<T>(P, (int) j, (int) j, wk, [](idx_t) { return (idx_t) 0; }, [](idx_t) {});
		else
			getrf_rec<T>(P, (int) j, (int) j, wk);
	};
	bool staged = false; // the panel about to be applied has already been applied to the next panel's columns except for its last part
	idx_t staged_last = 0; // ... whose width this is (half the panel, or one leaf: see the panel part below)
	{
		StreamScope sc(c.la_panel);
		panel(A.sub(0, 0, m, J[1]), 0);
		ev_panel = c.next_event();
		FH_HIP(hipEventRecord(ev_panel, c.la_panel));
	}
	// the interchanges of panel k (pivots [j0, j0 + w)) on the columns [c0, c0 + nc), rows j0 ..
	auto swaps = [&](idx_t k, idx_t j0, idx_t w, idx_t c0, idx_t nc) {
		(void) k;
		(void) w;
		laswp_list_dev<T>(A.sub(j0, c0, m - j0, nc), full);
	};
	// brings the columns [c0, c0 + nc) up to date with panel k = [j0, j0 + w): swaps, solve, update
	auto update = [&](idx_t k, idx_t j0, idx_t w, idx_t c0, idx_t nc) {
		const idx_t j1 = j0 + w;
		swaps(k, j0, w, c0, nc);
		MatV<T> U = A.sub(j0, c0, w, nc);
		trsm_lower_dev<T>(A.sub(j0, j0, w, w).c(), true, U);
		if (m > j1)
			gemm_dev<T>(A.sub(j1, c0, m - j1, nc), DST_FULL, true, A.sub(j1, j0, m - j1, w).c(), U.c(), (T) -1);
	};
	// ---- staged update of the NEXT panel's columns (while the panel chain is the critical one, see the panel part below).
	// Stage q of panel [jp, jp + 512) on the columns [cx, cx + wx): the interchanges of its pivots [q QW, (q + 1) QW), the solve
	// against that diagonal block, the product with the QW columns below it -- on the bulk stream.
	// QW = 256 (two stages): with four stages of 128 the part left between two panels is shorter, but the factorization is 2 - 3 ms
	// SLOWER at N = 8192 .. 16384 (profiles/r04_exp_lu_stages.txt): twice the small launches beside the latency-bound panel kernel
	idx_t QW = 256; // (half of the panel being staged; set per step)
	hipEvent_t ev_stage[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	// (general form: the pivots / columns [r0, r0 + qw) of the panel)
	auto stage_update_at = [&](idx_t r0, idx_t qw, idx_t cx, idx_t wx) {
		const idx_t r1 = r0 + qw;
		laswp_dev<T>(A.sub(r0, cx, m - r0, wx), wk.piv + r0, (int) qw, (int) r0);
		MatV<T> U = A.sub(r0, cx, qw, wx);
		trsm_lower_dev<T>(A.sub(r0, r0, qw, qw).c(), true, U);
		if (m > r1)
			gemm_dev<T>(A.sub(r1, cx, m - r1, wx), DST_FULL, true, A.sub(r1, r0, m - r1, qw).c(), U.c(), (T) -1);
	};
	auto stage_update = [&](idx_t jp, idx_t q, idx_t cx, idx_t wx) { stage_update_at(jp + q * QW, QW, cx, wx); };
	// The panel [jp, jp + 512) on the panel stream with the top levels of getrf_rec's recursion written out (factor.rs:84-117,
	// :127-185), columns [lo, hi) of it: as soon as a QW-column part is final, its stage starts on the bulk stream; the
	// interchanges a later part makes on that part's columns wait for the stage, which reads them.
	std::function<void(idx_t, idx_t, idx_t, idx_t, idx_t)> staged_panel = [&](idx_t jp, idx_t cx, idx_t wx, idx_t lo, idx_t hi) {
		if (hi - lo == QW) {
			getrf_rec<T>(A.sub(jp + lo, jp + lo, m - jp - lo, QW), (int) (jp + lo), (int) (jp + lo), wk);
			const idx_t q = lo / QW;
			if (hi < 2 * QW) { // (the last part's stage opens the next step on the bulk stream)
				hipEvent_t ev_part = c.next_event();
				FH_HIP(hipEventRecord(ev_part, c.la_panel));
				StreamScope sb(c.la_bulk);
				stream_wait(c.la_bulk, ev_part);
				stage_update(jp, q, cx, wx);
				ev_stage[q] = c.next_event();
				FH_HIP(hipEventRecord(ev_stage[q], c.la_bulk));
			}
			return;
		}
		const idx_t mid = (lo + hi) / 2;
		staged_panel(jp, cx, wx, lo, mid);
		const idx_t a = jp + lo, b = jp + mid;
		laswp_dev<T>(A.sub(a, b, m - a, hi - mid), wk.piv + a, (int) (mid - lo), (int) a);
		MatV<T> U = A.sub(a, b, mid - lo, hi - mid);
		trsm_lower_dev<T>(A.sub(a, a, mid - lo, mid - lo).c(), true, U);
		gemm_dev<T>(A.sub(b, b, m - b, hi - mid), DST_FULL, true, A.sub(b, a, m - b, mid - lo).c(), U.c(), (T) -1);
		staged_panel(jp, cx, wx, mid, hi);
		stream_wait(c.la_panel, ev_stage[mid / QW - 1]); // (bulk stream order: the earlier stages are done as well)
		laswp_dev<T>(A.sub(b, a, m - b, mid - lo), wk.piv + b, (int) (hi - mid), (int) b);
	};
	for (idx_t k = 0; k < nsteps; ++k) {
		const idx_t j0 = J[(size_t) k], j1 = J[(size_t) k + 1], w = j1 - j0;
		const idx_t j2 = Jat(k + 2), w2 = j2 - j1;
		hipEvent_t ev_next = nullptr;
		use_list(k);
		{
			StreamScope sc(c.la_bulk);
			stream_wait(c.la_bulk, ev_panel);
			if (ev_help_prev)
				stream_wait(c.la_bulk, ev_help_prev); // (the helper's tiles of the previous product: done well before that product's last tile)
			// Every solve U = L_kk^-1 A is a dependent chain of ~8 launches whose length does not depend on the number of columns
			// (profiles/r04_lu_v4_timeline.txt: ~240 us per chain, three chains per step -- the first 64 columns of the next panel,
			// its other 448, the rest -- were 20 of the bulk stream's 87 ms).  How the columns right of the panel are grouped:
			//   mode 2 (while the bulk stream is the critical one: its trailing product takes longer than a panel): ONE chain of
			//          interchanges + solve for all of them, then the product on the next panel's columns (-> released), then on
			//          the rest;
			//   mode 1 (afterwards, the panel chain is critical): the next panel's columns first (all 512: with the round-4 leaf
			//          the split "first 64, then 448" of rounds 1-3 made the panel wait for the second chain), then the rest.
			// Measured in one visit (profiles/r04_exp_lu_chain_grouping.txt, N = 16384): three chains 107.9 ms, mode 1 everywhere
			// 102.4, mode 2 everywhere 101.6, mode 2 down to 10240 remaining rows 100.7.
			const int mode = bulk_bound(m - j1) ? 2 : 1;
			auto compose = [&]() { laswp_compose_list(wk.piv + j0, (int) w, (int) j0, full); };
			const bool pipe = w2 > 0 && mode == 2 && side && !staged && m > j1 && j2 + 2 * LU_LA_NB <= jA && jA < n && m - j1 >= LU_PIPE_FROM;
			if (!pipe && !(w2 > 0 && staged)) // (the staged path composes it behind the release of the next panel)
				compose();
			if (pipe) {
				hipEvent_t ev_cA = c.next_event(), ev_cB = c.next_event();
				{
					StreamScope ss(side);
					stream_wait(side, ev_panel);
					compose();
					if (ev_GA_prev)
						stream_wait(side, ev_GA_prev);
					if (ev_help_prev && help_prev_c0 < jA)
						stream_wait(side, ev_help_prev);
					// near group + the next panel's columns: one chain, then the product on the next panel's columns
					swaps(k, j0, w, j1, jA - j1);
					MatV<T> U = A.sub(j0, j1, w, jA - j1);
					trsm_lower_dev<T>(A.sub(j0, j0, w, w).c(), true, U);
					FH_HIP(hipEventRecord(ev_cA, side)); // (the bulk stream's product on A waits for the solve only)
					gemm_dev<T>(A.sub(j1, j1, m - j1, w2), DST_FULL, true, A.sub(j1, j0, m - j1, w).c(), U.sub(0, 0, w, w2).c(), (T) -1);
					ev_next = c.next_event();
					FH_HIP(hipEventRecord(ev_next, side));
					if (ev_GB_prev)
						stream_wait(side, ev_GB_prev);
					if (ev_help_prev)
						stream_wait(side, ev_help_prev);
					swaps(k, j0, w, jA, n - jA);
					trsm_lower_dev<T>(A.sub(j0, j0, w, w).c(), true, A.sub(j0, jA, w, n - jA));
					FH_HIP(hipEventRecord(ev_cB, side));
				}
				stream_wait(c.la_bulk, ev_cA);
				if (jA > j2)
					gemm_dev<T>(A.sub(j1, j2, m - j1, jA - j2), DST_FULL, true, A.sub(j1, j0, m - j1, w).c(), A.sub(j0, j2, w, jA - j2).c(), (T) -1);
				ev_GA_prev = c.next_event();
				FH_HIP(hipEventRecord(ev_GA_prev, c.la_bulk));
				stream_wait(c.la_bulk, ev_cB);
				big_product(A.sub(j1, jA, m - j1, n - jA), A.sub(j1, j0, m - j1, w).c(), A.sub(j0, jA, w, n - jA).c(), true, jA);
				ev_GB_prev = c.next_event();
				FH_HIP(hipEventRecord(ev_GB_prev, c.la_bulk));
			} else
			if (w2 > 0 && staged) {
				// last stage of the staged update of the next panel's columns (the earlier ones ran beside the rest of panel k,
				// see the panel part below): interchanges of the last QW pivots, U = L_qq^-1 (.), product with K = QW
				if (staged_last == LUN_W) {
					// the last LEAF of panel k is all that is left: its interchanges (which also gather the top block) and ONE
					// launch of the fused node kernel on the next panel's columns, which sit right behind the leaf's -- two launches
					// on the whole chip between two panels instead of "interchanges, substitution leaf, product" for 128 pivots
					const idx_t r0 = j1 - LUN_W;
					FH_CHECK(w2 <= LU_FLAT_MAXW, "lu: more column groups than LuWork::ttop holds");
					laswp_dev<T>(A.sub(r0, j1, m - r0, w2), wk.piv + r0, (int) LUN_W, (int) r0, wk.ttop, (const T *) (A.p + r0 * A.rs + r0 * A.cs), A.cs);
					const idx_t below = m - r0 - LUN_W;
					unsigned nwg = below > 0 ? (unsigned) ((below + LUN_ROWS - 1) / LUN_ROWS) : 1u;
					const unsigned ncg = (unsigned) ((w2 + LUN_W - 1) / LUN_W);
					const unsigned cap = (unsigned) (2 * ctx().stream_cus()) / ncg;
					if (nwg > cap)
						nwg = cap < 1u ? 1u : cap;
					hipLaunchKernelGGL(lu_node64_kernel<T>, dim3(nwg, ncg), dim3(256), 0, ctx().stream, A.p + r0 * A.rs + r0 * A.cs, A.cs, (int) (m - r0), (int) w2,
							   (const T *) wk.ttop, 1);
					FH_HIP(hipGetLastError());
				} else {
					QW = w / 2;
					stage_update(j0, 1, j1, w2);
				}
				ev_next = c.next_event();
				FH_HIP(hipEventRecord(ev_next, c.la_bulk));
				compose();
				if (j2 < n)
					update(k, j0, w, j2, n - j2);
			} else if (w2 > 0 && mode == 2 && side && j2 < n && m > j1) {
				// the chain of the columns behind the next panel -- interchanges, solve -- on the side stream, BESIDE the bulk
				// stream's chain and product for the next panel's columns: both chains are latency bound (a dozen small launches
				// each), together they keep the chip no busier than one, and the big product starts a chain earlier
				hipEvent_t ev_list = c.next_event(), ev_side = c.next_event();
				FH_HIP(hipEventRecord(ev_list, c.la_bulk)); // (the list is composed, everything earlier on these columns is done)
				{
					StreamScope ss(side);
					stream_wait(side, ev_list);
					swaps(k, j0, w, j2, n - j2);
					trsm_lower_dev<T>(A.sub(j0, j0, w, w).c(), true, A.sub(j0, j2, w, n - j2));
					FH_HIP(hipEventRecord(ev_side, side));
				}
				update(k, j0, w, j1, w2);
				ev_next = c.next_event();
				FH_HIP(hipEventRecord(ev_next, c.la_bulk));
				stream_wait(c.la_bulk, ev_side);
				big_product(A.sub(j1, j2, m - j1, n - j2), A.sub(j1, j0, m - j1, w).c(), A.sub(j0, j2, w, n - j2).c(), true, j2);
			} else if (w2 > 0 && mode == 2) {
				swaps(k, j0, w, j1, n - j1);
				MatV<T> U = A.sub(j0, j1, w, n - j1);
				trsm_lower_dev<T>(A.sub(j0, j0, w, w).c(), true, U);
				if (m > j1)
					gemm_dev<T>(A.sub(j1, j1, m - j1, w2), DST_FULL, true, A.sub(j1, j0, m - j1, w).c(), U.sub(0, 0, w, w2).c(), (T) -1);
				ev_next = c.next_event();
				FH_HIP(hipEventRecord(ev_next, c.la_bulk));
				if (j2 < n && m > j1)
					big_product(A.sub(j1, j2, m - j1, n - j2), A.sub(j1, j0, m - j1, w).c(), U.sub(0, w2, w, n - j2).c(), true, j2);
			} else if (w2 > 0) {
				update(k, j0, w, j1, w2);
				ev_next = c.next_event();
				FH_HIP(hipEventRecord(ev_next, c.la_bulk));
				if (j2 < n)
					update(k, j0, w, j2, n - j2);
			}
			if (!pipe) { // (a pipelined step that followed would order its chains behind this step's bulk work)
				ev_GA_prev = ev_GB_prev = c.next_event();
				FH_HIP(hipEventRecord(ev_GA_prev, c.la_bulk));
			}
			// factor.rs:127-185: the panel's transpositions act on the columns to its left as well (nothing reads those columns
			// again during the factorization; on the plain side stream instead these passes made the factorization 1.2 ms slower,
			// composed over 8 panels they cost a serial tail: profiles/r05_exp_lu_driver.txt, r02_exp_lu_panel.txt)
			if (j0 > 0)
				swaps(k, j0, w, 0, j0);
		}
		if (w2 > 0) {
			StreamScope sc(c.la_panel);
			stream_wait(c.la_panel, ev_next);
			// While the panel chain is the critical one, what sits between two panels is the bulk stream's chain "interchanges,
			// solve against the 512 x 512 triangle, product" on the next panel's columns (~330 us, profiles/r04_lu_timeline.txt).
			// Most of it can run early: as soon as a QW-column part of panel k + 1 is final, the bulk stream (idle in this
			// phase) applies it to the columns of panel k + 2 -- interchanges of its QW pivots, U = L_qq^-1 (.), product with
			// K = QW on all rows below -- beside the rest of the panel; only the last part's stage (above) is left between the
			// panels.  Row interchanges of later parts commute with a stage (they move whole rows of L and of the updated columns
			// alike); the panel stream only has to wait for a stage before it interchanges rows of that part's columns
			// (factor.rs:127-185), which the stage reads.
			const idx_t w3 = Jat(k + 3) - j2;
			const bool stage = w2 >= 256 && w2 % 256 == 0 && w3 > 0 && !bulk_bound(m - j2) && m - j1 > w2;
			MatV<T> Pn = A.sub(j1, j1, m - j1, w2);
			if (!stage) {
				panel(Pn, j1);
			} else if (flat_panel_ok<T>(Pn, wk)) {
				// the flat panel in two parts: as soon as the left half is final -- its leaves' interchanges applied inside the
				// half -- the bulk stream starts stage 0; the right half's interchanges on the left half's columns wait for it
				// (round 5, second half: the first part is everything but the LAST LEAF -- its stage runs beside that leaf, and what is
				// left between two panels is one leaf's worth of interchanges + one fused node launch; halves before)
				const idx_t nl = w2 / LUN_W;
				const idx_t s0 = lu_node_offsets_ok<T>(m - j1, A.cs) ? w2 - LUN_W : w2 / 2; // width of the first part
				const idx_t lp0 = s0 / LUN_W;						    // its leaves
				(void) nl;
				getrf_panel_flat<T>(
					Pn, (int) j1, (int) j1, wk, [&](idx_t jl) { return jl < lp0 ? (idx_t) 0 : s0; },
					[&](idx_t jl) {
						if (jl + 1 == lp0) {
							hipEvent_t ev_part = c.next_event();
							FH_HIP(hipEventRecord(ev_part, c.la_panel));
							StreamScope sb(c.la_bulk);
							stream_wait(c.la_bulk, ev_part);
							stage_update_at(j1, s0, j2, w3);
							ev_stage[0] = c.next_event();
							FH_HIP(hipEventRecord(ev_stage[0], c.la_bulk));
						}
					});
				stream_wait(c.la_panel, ev_stage[0]);
				laswp_dev<T>(A.sub(j1 + s0, j1, m - j1 - s0, s0), wk.piv + j1 + s0, (int) (w2 - s0), (int) (j1 + s0));
				staged_last = w2 - s0;
			} else {
				QW = w2 / 2;
				staged_panel(j1, j2, w3, 0, w2);
				staged_last = QW;
			}
			staged = stage;
			ev_panel = c.next_event();
			FH_HIP(hipEventRecord(ev_panel, c.la_panel));
			ev_help_prev = nullptr;
			if (help.on) { // the panel stream's CUs join the step's big product until its tiles run low
				stream_wait(c.la_panel, help.ev_in);
				GemmExtra<T> ex;
				ex.ticket = help.ticket;
				ex.helper_wgs = help.wgs;
				ex.helper_margin = 2 * (ctx().ncu > 0 ? ctx().ncu - ctx().la_panel_cus : 224) / 8 * 2;
				gemm_dev<T>(help.C, DST_FULL, true, help.A, help.B, (T) -1, &ex);
				ev_help_prev = c.next_event();
				FH_HIP(hipEventRecord(ev_help_prev, c.la_panel));
				help_prev_c0 = help.c0;
				help.on = false;
			}
		} else {
			staged = false;
			ev_help_prev = nullptr;
		}
		FH_CHECK(!help.on, "lu: a helped product without a helper launch");
	}
	hipEvent_t eb = c.next_event();
	FH_HIP(hipEventRecord(eb, c.la_bulk));
	stream_wait(caller, eb);
	hipEvent_t ep = c.next_event(); // (everything on the panel stream, a last helper launch included)
	FH_HIP(hipEventRecord(ep, c.la_panel));
	stream_wait(caller, ep);
}

template <typename T> long getrf_dev(MatV<T> A, idx_t *perm, idx_t *perm_inv)
{
	const idx_t m = A.nrows, n = A.ncols;
	const LuLent lent_rec = t_lu_lent; // consumed by this call whatever happens next
	t_lu_lent = LuLent{};
	FH_CHECK(m < (1L << 30) && n < (1L << 30), "partial_piv_lu: matrix too large");
	for (idx_t i = 0; i < m; ++i)
		perm[i] = i;
	const idx_t size = m < n ? m : n;
	long n_trans = 0;
	if (size > 0) {
		Scratch pivb((size_t) size * sizeof(int));
		const size_t gran_bytes = (size_t) LU2_NSLOT * LU2_GMAX * LU2_GSLOT * sizeof(xwg_u64), diag_bytes = (size_t) LU2_NSLOT * 2 * LU_WMAX * sizeof(xwg_u64);
		Scratch granb(gran_bytes + diag_bytes);
		Scratch misc(256);
		Scratch wwsb(LW_WS_BYTES);
		Scratch ttopb((size_t) LUN_W * LU_FLAT_MAXW * sizeof(T));
		LuWork<T> wk;
		wk.wws = wwsb.as<unsigned char>();
		wk.ttop = ttopb.as<T>();
		FH_HIP(hipMemsetAsync(wwsb.p, 0, LW_WS_BYTES, ctx().stream));
		wk.piv = pivb.as<int>();
		wk.gran = granb.as<xwg_u64>();
		wk.gran_diag = wk.gran + (size_t) LU2_NSLOT * LU2_GMAX * LU2_GSLOT;
		wk.epoch_base = 0;
		wk.status = misc.as<int>() + 8;
		FH_HIP(hipMemsetAsync(misc.p, 0, 256, ctx().stream));
		FH_HIP(hipMemsetAsync(granb.p, 0, gran_bytes + diag_bytes, ctx().stream));

		// The cooperative leaves exchange pivots between RESIDENT workgroups.  Residency is arranged, not hoped for: a leaf is
		// only launched with at most one workgroup per compute unit its stream can use (leaf_width_for / resident_workgroups,
		// checked against the kernel's occupancy in getrf_leaf), the look-ahead driver runs it on CUs no other stream of this
		// library touches, and taller panels take the non-cooperative leaf.  What remains is a GPU shared with OTHER work: a
		// workgroup may then get its CU late.  Every spin is bounded (~0.2 s); a timeout raises status word 2, the panel
		// kernels return before they store anything, and the call reports PartialPivLuStatus::Unknown through the boundary's
		// own status channel (faer-ffi/src/lib.rs:591-595) -- unless the caller has lent a copy of A
		// (faer_hip_partial_piv_lu_lend_copy): then A is restored from it and factored again on the non-cooperative leaves
		// (getrf_leaf_general), which wait for nobody.  Rounds 3-4 made that copy themselves on every call (+N^2 scalars of
		// memory, one read + one write of A per factorization: VERDICT r04 item 5); it is the caller's choice now.
		const bool force_general = g_lu_force_general.load() != 0;
		const T *lent = nullptr;
		if (lent_rec.p) {
			FH_CHECK(lent_rec.nrows == m && lent_rec.ncols == n && lent_rec.elem_bytes == (int) sizeof(T),
				 "partial_piv_lu: the lent copy was made for another shape or scalar type");
			lent = static_cast<const T *>(lent_rec.p);
		}
		// nobody lent one: the library keeps its own while that is cheap (and only where a cooperative leaf can run at all)
		std::optional<Scratch> backup;
		if (!lent && !force_general && (size_t) m * (size_t) n * sizeof(T) <= LU_AUTO_BACKUP_BYTES) {
			backup.emplace((size_t) m * (size_t) n * sizeof(T));
			copy_dev<T>(MatV<T>{backup->as<T>(), m, n, 1, m}, A.c());
			lent = backup->as<T>();
		}
		// look-ahead needs every workgroup of a cooperative leaf resident on the CUs reserved for the panel stream
		const idx_t leaf_r = leaf_rows_per_wg<T>(LU_W);
		const idx_t la_min = g_lu_plan_la_min.load() > 0 ? (idx_t) g_lu_plan_la_min.load() : 8 * LU_LA_NB;
		const bool la = !force_general && size >= la_min && ctx().lookahead_streams() && (m + leaf_r - 1) / leaf_r <= (idx_t) ctx().la_panel_cus;
		if (la)
			getrf_lookahead<T>(A.sub(0, 0, m, size), wk, ctx().stream);
		else if (flat_panel_ok<T>(A.sub(0, 0, m, size), wk)) // (one panel of the look-ahead driver: same launches, same pivots)
			getrf_panel_flat<T>(A.sub(0, 0, m, size), 0, 0, wk, [](idx_t) { return (idx_t) 0; }, [](idx_t) {});
		else
			getrf_rec<T>(A.sub(0, 0, m, size), 0, 0, wk);
		if (lent && !force_general) {
			int st2[4] = {0, 0, 0, 0};
			FH_HIP(hipMemcpyAsync(st2, wk.status, sizeof(st2), hipMemcpyDeviceToHost, ctx().stream));
			ctx().sync();
			ctx().quiesce();
			if (st2[2] != 0) {
				fprintf(stderr, "faer_hip: partial_piv_lu: the cross-workgroup exchange of the panel kernel timed out (GPU shared with other "
						"work?); redoing the factorization from the lent copy on the non-cooperative path\n");
				copy_dev<T>(A, MatV<const T>{lent, m, n, 1, m});
				FH_HIP(hipMemsetAsync(misc.p, 0, 256, ctx().stream));
				wk.general = true;
				getrf_rec<T>(A.sub(0, 0, m, size), 0, 0, wk);
			}
		}
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
		ctx().quiesce();
		if (st[2] != 0) {
			// The cooperative panel kernel did not get all its workgroups resident within its bounded spin (CUs held by
			// another process or stream: residency is not something a plain launch can be promised).  The
			// matrix is partially factored; this is reported through the boundary's own status channel
			// (PartialPivLuStatus::Unknown, faer-ffi/src/lib.rs:591-595) instead of aborting the caller's process.
			fprintf(stderr, "faer_hip: partial_piv_lu: the cross-workgroup exchange of the panel kernel timed out (GPU shared with other work?)\n");
			for (idx_t i = 0; i < m; ++i)
				perm_inv[i] = i;
			return -1;
		}
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

// `status_dev` (optional, 16 zeroed device ints owned by the caller): the kernels' status words go there and NOTHING is
// waited for -- the caller looks at word 2 (exchange timed out) once, after its own final synchronisation.  The
// distributed LU calls it this way for every block column: no host synchronisation inside its loop (dist_lu.h).
template <typename T> void getrf_panel_dev(MatV<T> P, int *piv_dev, int *status_dev)
{
	const idx_t m = P.nrows, w = P.ncols;
	FH_CHECK(w <= m, "getrf_panel: the panel must be tall");
	if (w == 0)
		return;
	const size_t gran_bytes = (size_t) LU2_NSLOT * LU2_GMAX * LU2_GSLOT * sizeof(xwg_u64), diag_bytes = (size_t) LU2_NSLOT * 2 * LU_WMAX * sizeof(xwg_u64);
	// released on return while the kernels may still be queued: the pool hands a buffer back to the SAME stream only
	Scratch granb(gran_bytes + diag_bytes);
	Scratch misc(256);
	Scratch wwsb(LW_WS_BYTES);
	Scratch ttopb((size_t) LUN_W * LU_FLAT_MAXW * sizeof(T));
	LuWork<T> wk;
	wk.wws = wwsb.as<unsigned char>();
	wk.ttop = ttopb.as<T>();
	FH_HIP(hipMemsetAsync(wwsb.p, 0, LW_WS_BYTES, ctx().stream));
	wk.piv = piv_dev;
	wk.gran = granb.as<xwg_u64>();
	wk.gran_diag = wk.gran + (size_t) LU2_NSLOT * LU2_GMAX * LU2_GSLOT;
	wk.epoch_base = 0;
	wk.status = status_dev ? status_dev : misc.as<int>() + 8;
	FH_HIP(hipMemsetAsync(misc.p, 0, 256, ctx().stream));
	FH_HIP(hipMemsetAsync(granb.p, 0, gran_bytes + diag_bytes, ctx().stream));
	if (flat_panel_ok<T>(P, wk)) // (the distributed driver's block columns: the flat right-looking panel of the look-ahead driver)
		getrf_panel_flat<T>(P, 0, 0, wk, [](idx_t) { return (idx_t) 0; }, [](idx_t) {});
	else
		getrf_rec<T>(P, 0, 0, wk);
	if (status_dev)
		return;
	int st[4] = {0, 0, 0, 0};
	FH_HIP(hipMemcpyAsync(st, wk.status, sizeof(st), hipMemcpyDeviceToHost, ctx().stream));
	ctx().sync();
	FH_CHECK(st[2] == 0, "partial_piv_lu: device exchange timed out in the panel kernel");
}
template <typename T> void laswp_rows_dev(MatV<T> B, const int *piv_dev, int nt) { laswp_dev<T>(B, piv_dev, nt, 0); }
// The same interchanges with the net permutation composed once and shared by several launches (the distributed driver applies a
// panel's pivots to three column ranges per step): `list` = 4 * nt device ints.  nt <= laswp_list_max().
int laswp_list_max() { return LASWP_SMALL_NT; }
void laswp_compose_rows_dev(const int *piv_dev, int nt, int *list)
{
	LaswpList l;
	l.dst = list;
	l.src = list + 2 * nt;
	laswp_compose_list(piv_dev, nt, 0, l);
}
template <typename T> void laswp_list_rows_dev(MatV<T> B, const int *list, int nt)
{
	LaswpList l;
	l.dst = const_cast<int *>(list);
	l.src = const_cast<int *>(list) + 2 * nt;
	l.nt = nt;
	laswp_list_dev<T>(B, l);
}
template void laswp_list_rows_dev<double>(MatV<double>, const int *, int);
template void laswp_list_rows_dev<float>(MatV<float>, const int *, int);

template void getrf_panel_dev<double>(MatV<double>, int *, int *);
template void getrf_panel_dev<float>(MatV<float>, int *, int *);
template void laswp_rows_dev<double>(MatV<double>, const int *, int);
template void laswp_rows_dev<float>(MatV<float>, const int *, int);
template long getrf_dev<double>(MatV<double>, idx_t *, idx_t *);
template long getrf_dev<float>(MatV<float>, idx_t *, idx_t *);

} // namespace fh
