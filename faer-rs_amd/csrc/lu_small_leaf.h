// Single-workgroup LU leaf: 64 columns x at most NW * 64 rows, the whole panel in registers, no cross-workgroup exchange.
//
// Same column loop as getrf_wpanel_kernel (lu_wpanel.h) -- faer/src/linalg/lu/partial_pivoting/factor.rs:19-67: first row of
// strictly largest |a| (ties: smallest row index), interchange, scaling by the reciprocal pivot, rank-1 update as
// fma(l, -u, dst) -- and bitwise the same factors and pivots; what it leaves out is that kernel's protocol.  The cooperative
// kernel publishes a header and a row record per column through memory and polls for them even when its grid is ONE workgroup:
// 2.3-2.6 us per column whatever the number of rows (kernel trace of the last steps of N = 16384: 147-165 us per 64-column
// leaf on 256 .. 1536 rows, profiles/r05_exp_lu_driver.txt item 11).  Here a column is: wavefront arg-max on the DPP network, the
// wavefront's candidate row and (|a|, label) to LDS, ONE barrier, every wavefront picks the workgroup's winner and reads its
// row back as LDS broadcasts (one address per wavefront, 16 bytes per read).
//   * rows never move: every register row carries its LABEL (the row index it would have after the interchanges so far), rows
//     are written to their label positions at the end -- as in the cooperative kernel;
//   * eight step bodies, the panel's registers rotated by 8 positions per group of steps (the column being eliminated sits at
//     a compile-time position; the code of a launch is 8 step bodies, not 64);
//   * steps past min(w, m) run on the zero padding and store nothing.
#pragma once
#include <climits>

#include "common.h"
#include "lds_blocks.h"
#include "lu_wpanel.h"

namespace fh {

template <int NW> struct alignas(16) LsShared {
	double cv[2][NW];
	int lab[2][NW];
	double trans[2][NW][LW_W]; // the wavefronts' candidate rows of the current column (slots alternate with the column parity)
};

template <typename T, int NW, int JJ>
static __device__ __forceinline__ void ls_step(T (&x)[LW_W], int &lab, LsShared<NW> &sh, int grp, int steps, int *piv, int row_base)
{
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int J = grp * 8 + JJ, par = J & 1;
	const int lim = LW_W - grp * 8; // positions < lim hold unfinished columns
	// ---- candidate of this wavefront (a real one, |a| > 0, or the diagonal row of a zero / NaN-only column: factor.rs:35-43)
	const double av = fabs((double) x[JJ]);
	const double cv = lab >= J ? (av > 0.0 ? av : (lab == J ? 0.0 : -1.0)) : -1.0;
	double wv;
	int wl;
	lw_argmax<6>(cv, lab, wv, wl);
	const bool has = wv >= 0.0;
	if (has) { // wave uniform
		const int ol = __builtin_amdgcn_readfirstlane((int) __ffsll((unsigned long long) __ballot(lab == wl && cv == wv)) - 1);
		if (lane == ol) {
#pragma unroll
			for (int c = 0; c < LW_W; ++c)
				sh.trans[par][wave][c] = (double) x[c];
		}
	}
	if (lane == 0) {
		sh.cv[par][wave] = has ? wv : -1.0;
		sh.lab[par][wave] = has ? wl : INT_MAX;
	}
	__syncthreads(); // (the only barrier of a column: slot `par` is written again two columns later, behind the next barrier)
	const double ecv = lane < NW ? sh.cv[par][lane] : -1.0;
	const int elab = lane < NW ? sh.lab[par][lane] : INT_MAX;
	double gv;
	int p;
	lw_argmax<6>(ecv, elab, gv, p);
	if (!(gv >= 0.0))
		return; // nobody has a candidate (J >= m on the padding): workgroup uniform
	const int ow = __builtin_amdgcn_readfirstlane((int) __ffsll((unsigned long long) __ballot(lane < NW && elab == p && ecv == gv)) - 1);
	const double *prow = sh.trans[par][ow]; // the pivot row by register position: broadcast reads (one address for the whole wavefront)
	// "interchange" J <-> p on the labels
	lab = lab == J ? p : (lab == p ? J : lab);
	if (tid == 0 && J < steps)
		piv[J] = row_base + p;
	// scaling by the reciprocal pivot (factor.rs:45-57), rank-1 update (factor.rs:59-64); rows labelled <= J stay bitwise untouched
	const T inv = (T) 1 / (T) prow[JJ];
	const bool upd = lab > J;
	T l = (T) 0;
	if (upd) {
		l = x[JJ] * inv;
		x[JJ] = l;
	}
	typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
	for (int cb = 0; cb < LW_W / 8; ++cb) {
		if (cb * 8 + 7 > JJ && cb * 8 < lim) {
			T u[8];
#pragma unroll
			for (int k = 0; k < 8; k += 2) {
				const d2 uu = *reinterpret_cast<const d2 *>(prow + cb * 8 + k);
				u[k] = (T) uu.x;
				u[k + 1] = (T) uu.y;
			}
			if (upd) {
#pragma unroll
				for (int k = 0; k < 8; ++k)
					if (cb * 8 + k > JJ)
						x[cb * 8 + k] = fh_fma(l, -u[k], x[cb * 8 + k]);
			}
		}
	}
}

template <typename T, int NW> __global__ __launch_bounds__(NW * 64) void getrf_small_leaf_kernel(T *P, idx_t rs, idx_t cs, int m, int w, int *piv, int row_base)
{
	__shared__ LsShared<NW> sh;
	const int tid = threadIdx.x;
	const int gr = tid; // one row per thread
	int lab = gr < m ? gr : -1; // rows past the end never take part
	T x[LW_W];
#pragma unroll
	for (int c = 0; c < LW_W; ++c) {
		const bool in = gr < m && c < w;
		const T v = P[in ? (idx_t) gr * rs + (idx_t) c * cs : (idx_t) 0];
		x[c] = in ? v : (T) 0;
	}
	const int steps = min(w, m);
	int rot = 0;
	for (int grp = 0; grp * 8 < steps; ++grp) {
		ls_step<T, NW, 0>(x, lab, sh, grp, steps, piv, row_base);
		ls_step<T, NW, 1>(x, lab, sh, grp, steps, piv, row_base);
		ls_step<T, NW, 2>(x, lab, sh, grp, steps, piv, row_base);
		ls_step<T, NW, 3>(x, lab, sh, grp, steps, piv, row_base);
		ls_step<T, NW, 4>(x, lab, sh, grp, steps, piv, row_base);
		ls_step<T, NW, 5>(x, lab, sh, grp, steps, piv, row_base);
		ls_step<T, NW, 6>(x, lab, sh, grp, steps, piv, row_base);
		ls_step<T, NW, 7>(x, lab, sh, grp, steps, piv, row_base);
		// rotate the row left by 8: the finished columns go to the tail
		T t8[8];
#pragma unroll
		for (int k = 0; k < 8; ++k)
			t8[k] = x[k];
#pragma unroll
		for (int c = 0; c + 8 < LW_W; ++c)
			x[c] = x[c + 8];
#pragma unroll
		for (int k = 0; k < 8; ++k)
			x[LW_W - 8 + k] = t8[k];
		rot += 8;
	}
	if (lab >= 0) {
#pragma unroll
		for (int c = 0; c < LW_W; ++c) {
			const int gc = (c + rot) & (LW_W - 1); // panel column of register position c
			if (gc < w)
				P[(idx_t) lab * rs + (idx_t) gc * cs] = x[c];
		}
	}
}

} // namespace fh
