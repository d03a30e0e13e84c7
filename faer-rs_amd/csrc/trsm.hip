// In-place triangular solve  X <- op(T)^-1 X  (left side) for gfx950.
//
// Replaces faer/src/linalg/triangular_solve.rs:16-604.  Same shape of
// algorithm as the reference -- recurse on the triangle, off-diagonal block by GEMM
// (triangular_solve.rs:420-604) -- but the leaf is GPU sized: instead of the reference's n <= 4 closed
// forms, one wavefront solves a 64 x 64 triangle against 64 right-hand sides:
//   * the triangle and its reciprocal diagonal sit in LDS and are read as wave-uniform broadcasts;
//   * every lane owns ONE right-hand side, held in 64 registers, and runs the column-oriented (axpy)
//     substitution: 2016 independent FMAs, no cross-lane traffic at all;
//   * X is staged through a padded LDS tile so that global accesses run along whichever of its two
//     strides is the small one (rows of a transposed Cholesky panel or columns of an LU block row);
//   * like the reference's base case (triangular_solve.rs:98-198) the diagonal enters as a reciprocal
//     computed once: x_j <- x_j * (1 / t_jj).
// Upper triangles are lower triangles on the row/column reversed views (triangular_solve.rs:578-604);
// every kernel here takes signed strides so the reversal is free.
#include <atomic>

#include "common.h"
#include "lds_blocks.h"
#include "mfma.h"
#include "trsm_pack.h"

namespace fh {

template <typename T>
__global__ __launch_bounds__(64) void trsm_leaf_kernel(const T *__restrict__ Lp, idx_t lrs, idx_t lcs, int n, int unit,
							T *Xp, idx_t xrs, idx_t xcs, int nrhs, int lanes_along_rhs)
{
	constexpr int NB = 64, XP = NB + 1;
	__shared__ T Ls[NB * NB]; // column major: Ls[j * NB + i] = L[i][j]
	__shared__ T dinv[NB];
	__shared__ T Xs[NB * XP]; // Xs[i * XP + c]
	const int lane = threadIdx.x;
	const int c0 = blockIdx.x * NB;
	const int nc = min(NB, nrhs - c0);

	// ---- stage the triangle (identity padded); 8 independent loads in flight per lane ----
	for (int e0 = lane; e0 < NB * NB; e0 += 64 * 8) {
		T v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int e = e0 + u * 64;
			const int i = e % NB, j = e / NB;
			const bool in = i < n && j < i;
			const T x = Lp[in ? (idx_t) i * lrs + (idx_t) j * lcs : (idx_t) 0];
			v[u] = in ? x : (T) 0;
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int e = e0 + u * 64;
			Ls[(e / NB) * NB + e % NB] = v[u];
		}
	}
	{
		T d = (T) 1;
		if (lane < n && !unit)
			d = (T) 1 / Lp[(idx_t) lane * lrs + (idx_t) lane * lcs];
		dinv[lane] = d;
	}
	// ---- stage X: lanes run along the dimension with the smaller stride; batches of 8 loads ----
	if (lanes_along_rhs) {
		for (int i0 = 0; i0 < n; i0 += 8) {
			T v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const bool in = i0 + u < n && lane < nc;
				const T x = Xp[in ? (idx_t) (i0 + u) * xrs + (idx_t) (c0 + lane) * xcs : (idx_t) 0];
				v[u] = in ? x : (T) 0;
			}
#pragma unroll
			for (int u = 0; u < 8; ++u)
				if (i0 + u < n)
					Xs[(i0 + u) * XP + lane] = v[u];
		}
	} else {
		for (int cc = 0; cc < nc; cc += 8) {
			T v[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const bool in = cc + u < nc && lane < n;
				const T x = Xp[in ? (idx_t) lane * xrs + (idx_t) (c0 + cc + u) * xcs : (idx_t) 0];
				v[u] = in ? x : (T) 0;
			}
#pragma unroll
			for (int u = 0; u < 8; ++u)
				if (cc + u < nc && lane < n)
					Xs[lane * XP + cc + u] = v[u];
		}
	}
	__syncthreads();

	T x[NB];
#pragma unroll
	for (int i = 0; i < NB; ++i)
		x[i] = (i < n && lane < nc) ? Xs[i * XP + lane] : (T) 0;

#pragma unroll
	for (int j = 0; j < NB; ++j) {
		if (j < n) { // wave-uniform
			const T xj = x[j] * dinv[j];
			x[j] = xj;
#pragma unroll
			for (int i = j + 1; i < NB; ++i)
				x[i] = fh_fma(-Ls[j * NB + i], xj, x[i]);
		}
	}

#pragma unroll
	for (int i = 0; i < NB; ++i)
		if (i < n)
			Xs[i * XP + lane] = x[i];
	__syncthreads();
	if (lanes_along_rhs) {
		for (int i = 0; i < n; ++i)
			if (lane < nc)
				Xp[(idx_t) i * xrs + (idx_t) (c0 + lane) * xcs] = Xs[i * XP + lane];
	} else {
		for (int c = 0; c < nc; ++c)
			if (lane < n)
				Xp[(idx_t) lane * xrs + (idx_t) (c0 + c) * xcs] = Xs[lane * XP + c];
	}
}

// ------------------------------------------------------------------------------------------------
// Many right-hand sides / large triangles.  The reference's recursion (triangular_solve.rs:452-484: solve with
// the top-left block, eliminate it from the rows below with ONE GEMM, recurse on the bottom-right block) runs
// on 128-row blocks; every off-diagonal step is an MFMA GEMM, every 128 x 128 diagonal block is solved by
// SUBSTITUTION in trsm_leaf128_kernel -- no explicit inverses anywhere (round 1 multiplied by inverted diagonal
// blocks, whose error grows with cond(T_kk); substitution is backward stable like the reference).
//
// trsm_leaf128_kernel: one workgroup = 4 wavefronts (one per SIMD), each wavefront RW = 16 or 32 right-hand sides.
//   * the packed image of the diagonal block (trsm_pack.h: 28 negated 16 x 16 tiles below the diagonal in MFMA
//     operand order, 8 diagonal tiles column by column with the reciprocal diagonal) is copied to LDS with 16-byte
//     loads, or built there from the strided triangle itself (DIRECT: single-block solves, no packing launch);
//   * the 64 x RW slice of X a wavefront works on lives in its LDS exchange tile; per 16-row tile: the rows are
//     brought up to date with the solved tiles above on the matrix cores (accumulator layout), then the 16 x 16
//     diagonal tile is solved by substitution on the vector ALU, lane = right-hand side, multipliers as LDS
//     broadcasts, the diagonal as a reciprocal like triangular_solve.rs:113;
//   * rows 64 .. 127 are loaded straight into accumulators, eliminated against the solved top half (256 MFMAs per
//     16 right-hand sides) and then solved like the top half;
//   * the MFMA phases of a wavefront scale with its share of the right-hand sides, the substitution phases wait for
//     LDS broadcasts whatever the lane count: 16 right-hand sides per wavefront as long as that needs at most one
//     workgroup per compute unit of the stream, 32 beyond (trsm_leaf128_launch);
//   * global accesses run along whichever stride of X is the small one: directly when that is the right-hand
//     side index (rows of a transposed Cholesky panel), through the tile otherwise (columns of X).
// ------------------------------------------------------------------------------------------------
constexpr int TRSM_IB = TP_NB;

// -DFH_PANEL_TIMING: s_memtime phase accounting of the substitution leaf (workgroup 0 / thread 0), printed and reset
// by faer_hip_debug_dump_timing() (timing build only: make -C csrc timing)
#ifdef FH_PANEL_TIMING
__device__ unsigned long long g_leaf128_timing[16];
#define FH_TT(i)                                                                                                         \
	do {                                                                                                             \
		if (blockIdx.x == 0 && threadIdx.x == 0) {                                                               \
			const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                    \
			atomicAdd(&g_leaf128_timing[i], now_ - tt_last);                                                 \
			tt_last = now_;                                                                                  \
		}                                                                                                        \
	} while (0)
#define FH_TT_DECL unsigned long long tt_last = __builtin_amdgcn_s_memtime()
void trsm_dump_timing()
{
	unsigned long long d[16];
	FH_HIP(hipDeviceSynchronize());
	FH_HIP(hipMemcpyFromSymbol(d, HIP_SYMBOL(g_leaf128_timing), sizeof(d)));
	const double c = d[15] ? (double) d[15] : 1.0;
	fprintf(stderr,
		"trsm leaf128 phases (s_memtime ticks per launch, wg 0 / thread 0, %llu launches): image+rows0 %.0f | rest of image+bar %.0f | subst0 %.0f | "
		"store0 %.0f | reload+rows1 %.0f | eliminate %.0f | subst1 %.0f | store1 %.0f\n",
		d[15], d[0] / c, d[1] / c, d[2] / c, d[3] / c, d[4] / c, d[5] / c, d[6] / c, d[7] / c);
	unsigned long long z[16] = {0};
	FH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_leaf128_timing), z, sizeof(z)));
}
#else
#define FH_TT(i)                                                                                                         \
	do {                                                                                                             \
	} while (0)
#define FH_TT_DECL
void trsm_dump_timing() {}
#endif

constexpr int TL_NW = 4;	   // wavefronts per workgroup of the substitution leaf
constexpr int TL_NT = TL_NW * 64;

// W block b <- packed image of the b-th 128 x 128 diagonal block of the lower triangular L
template <typename T>
__global__ __launch_bounds__(256) void trsm_pack_kernel(const T *__restrict__ Lp, idx_t lrs, idx_t lcs, int n, int unit, T *__restrict__ W)
{
	typedef TriPack<T> P;
	const int b = blockIdx.x;
	const int r0 = b * TP_NB;
	const int nb = min(TP_NB, n - r0);
	T *img = W + (size_t) b * P::SIZE;
	// alignment holes of the packed triangles must read as zeros (they are multiplied into padding lanes only, but
	// NaN garbage would still propagate): clear the two triangles first
	for (int e = threadIdx.x; e < TP_NT * P::DG_SZ; e += blockDim.x) // alignment holes of the diagonal tiles
		img[P::OFF_DG + e] = (T) 0;
	__syncthreads();
	const T *L0 = Lp + (idx_t) r0 * lrs + (idx_t) r0 * lcs;
	constexpr int U = 8;
	for (int e0 = threadIdx.x; e0 < TP_NB * TP_NB; e0 += 256 * U) {
		T v[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int e = e0 + u * 256;
			const int i = e % TP_NB, j = e / TP_NB;
			const bool in = i < nb && j <= i;
			const T x = L0[in ? (idx_t) i * lrs + (idx_t) j * lcs : (idx_t) 0];
			v[u] = in ? x : (T) 0;
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int e = e0 + u * 256;
			const int i = e % TP_NB, j = e / TP_NB;
			if (j > i)
				continue;
			bool neg;
			const int ps = P::pos(i, j, neg);
			const T val = i == j ? ((unit || i >= nb) ? (T) 1 : (T) 1 / v[u]) : v[u]; // the diagonal enters as its reciprocal
			img[ps] = neg ? -val : val;
		}
	}
}

template <typename T> void trsm_pack_dev(MatV<const T> L, bool unit, T *W)
{
	const idx_t n = L.nrows;
	if (n == 0)
		return;
	const idx_t nblk = (n + TP_NB - 1) / TP_NB;
	hipLaunchKernelGGL(trsm_pack_kernel<T>, dim3((unsigned) nblk), dim3(256), 0, ctx().stream, L.p, L.rs, L.cs, (int) n, unit ? 1 : 0, W);
	FH_HIP(hipGetLastError());
}
template void trsm_pack_dev<double>(MatV<const double>, bool, double *);
template void trsm_pack_dev<float>(MatV<const float>, bool, float *);

// DIRECT: there is no packed image in memory -- the workgroup packs the triangle (img = T(0, 0), strides trs / tcs,
// `unit`) into LDS itself.  Used for single-block solves (n <= 128), where a separate packing launch (~20 us on the
// dependent chain of an LU panel) would serve one leaf launch only.
template <typename T, bool DIRECT, int RW>
__global__ __launch_bounds__(TL_NT) void trsm_leaf128_kernel(const T *__restrict__ img, int n, T *Xp, idx_t xss, idx_t xcs, int nrhs,
							     int lanes_along_rhs, idx_t trs, idx_t tcs, int unit)
{
	typedef TriPack<T> P;
	static_assert(RW == 16 || RW == 32, "right-hand sides per wavefront");
	constexpr int XP = RW + 1;   // pitch of the per-wave 64 x RW exchange tile
	constexpr int JT = RW / 16;  // 16-column groups of right-hand sides per wavefront (MFMA N tiles)
	constexpr int NT = TL_NT;
	extern __shared__ __attribute__((aligned(16))) unsigned char tl_smem[];
	T *Ls = reinterpret_cast<T *>(tl_smem); // packed image
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	T *Xs = Ls + P::SIZE + wave * (TP_H * XP); // this wave's exchange tile: Xs[s * XP + c]
	const int c0 = (blockIdx.x * TL_NW + wave) * RW;
	const int nc = min(RW, nrhs - c0); // may be <= 0 for the trailing waves of the last workgroup
	const bool act = lane < nc;

	// Memory phases are organised by ROUND TRIPS, not by data: every batch below is a set of independent loads in
	// flight together (a leaf spends ~1.5 us per dependent round trip when the chip is busy), and the code is kept
	// SMALL (run-time loops around 16- / 32-deep batches): the whole kernel has to stay inside the 64 KB instruction
	// cache, straight-line code runs at the speed of instruction fetch (profiles/r02_trsm_leaf.txt).
	typedef int v4i __attribute__((ext_vector_type(4)));
	const v4i *isrc = reinterpret_cast<const v4i *>(img);
	v4i *idst = reinterpret_cast<v4i *>(Ls);
	constexpr int NV = (int) (P::BYTES / 16), UI = 6; // image: batches of 6 vectors per thread
	const bool two = n > TP_H;
	const int ns0 = min(TP_H, n), ns1 = two ? n - TP_H : 0;

	// rows s0 .. s0+ns-1 of this wave's RW right-hand sides -> tile (zero padded), lanes along the small stride
	auto load_half = [&](int s0, int ns) {
		if (lanes_along_rhs) {
#pragma unroll 1
			for (int i0 = 0; i0 < TP_H; i0 += 32) {
				T v[32];
#pragma unroll
				for (int u = 0; u < 32; ++u) {
					const bool in = i0 + u < ns && act;
					const T t = Xp[in ? (idx_t) (s0 + i0 + u) * xss + (idx_t) (c0 + lane) * xcs : (idx_t) 0];
					v[u] = in ? t : (T) 0;
				}
				if (lane < RW) {
#pragma unroll
					for (int u = 0; u < 32; ++u)
						Xs[(i0 + u) * XP + lane] = v[u];
				}
			}
		} else {
			T v[RW];
#pragma unroll
			for (int u = 0; u < RW; ++u) {
				const bool in = u < nc && lane < ns;
				const T t = Xp[in ? (idx_t) (s0 + lane) * xss + (idx_t) (c0 + u) * xcs : (idx_t) 0];
				v[u] = in ? t : (T) 0;
			}
#pragma unroll
			for (int u = 0; u < RW; ++u)
				Xs[lane * XP + u] = v[u];
		}
		__builtin_amdgcn_wave_barrier();
	};
	// rows s0 .. s0+ns-1 of the solution, tile -> X
	auto store_half = [&](int s0, int ns) {
		__builtin_amdgcn_wave_barrier();
		if (lanes_along_rhs) {
#pragma unroll 16
			for (int i = 0; i < TP_H; ++i)
				if (i < ns && act)
					Xp[(idx_t) (s0 + i) * xss + (idx_t) (c0 + lane) * xcs] = Xs[i * XP + lane];
		} else {
#pragma unroll 1
			for (int c8 = 0; c8 < RW; c8 += 16) {
				T w[16];
#pragma unroll
				for (int u = 0; u < 16; ++u)
					w[u] = Xs[lane * XP + c8 + u];
#pragma unroll
				for (int u = 0; u < 16; ++u)
					if (c8 + u < nc && lane < ns)
						Xp[(idx_t) (s0 + lane) * xss + (idx_t) (c0 + c8 + u) * xcs] = w[u];
			}
		}
		__builtin_amdgcn_wave_barrier();
	};

	FH_TT_DECL;
	if constexpr (DIRECT) {
		// the lower triangle of T straight from the matrix (strided) into the packed image: thread = row, 32 columns per
		// batch of independent loads, the two halves of the workgroup take alternate batches (two round trips each);
		// cheap index arithmetic (the column is wave uniform)
		__shared__ int dgo[TP_TS];
		if (threadIdx.x < TP_TS)
			dgo[threadIdx.x] = P::dg_off(threadIdx.x);
		for (int e = threadIdx.x; e < TP_NT * P::DG_SZ; e += NT)
			Ls[P::OFF_DG + e] = (T) 0; // alignment holes of the diagonal tiles
		__syncthreads();
		if (nc > 0)
			load_half(0, ns0);
		{
			const int i = threadIdx.x & (TP_NB - 1), bi = i >> 4, ii = i & 15;
			const bool in_i = i < n;
#pragma unroll 1
			for (int j0 = (threadIdx.x >> 7) * 32; j0 < TP_NB; j0 += 32 * (NT / TP_NB)) {
				T v[32];
				if (j0 < n) {
#pragma unroll
					for (int u = 0; u < 32; ++u) {
						const int j = j0 + u;
						const bool ld = in_i && j <= i;
						const T t = img[ld ? (idx_t) i * trs + (idx_t) j * tcs : (idx_t) 0];
						v[u] = ld ? t : (T) 0;
					}
				} else {
#pragma unroll
					for (int u = 0; u < 32; ++u)
						v[u] = (T) 0; // identity padding only: no round trip
				}
#pragma unroll
				for (int u = 0; u < 32; ++u) {
					const int j = j0 + u, bj = j >> 4, jj = j & 15;
					if (j > i)
						continue;
					if (bi > bj)
						Ls[P::od_tile(bi, bj) + (jj >> 2) * 64 + (jj & 3) * 16 + ii] = -v[u];
					else
						Ls[P::OFF_DG + bi * P::DG_SZ + dgo[jj] + (ii - jj)] = i == j ? ((unit || !in_i) ? (T) 1 : (T) 1 / v[u]) : v[u];
				}
			}
		}
		FH_TT(0);
	} else {
		// trip 1: first image batch + the top rows; then the rest of the image
		{
			v4i va[UI];
#pragma unroll
			for (int u = 0; u < UI; ++u)
				va[u] = isrc[min((int) threadIdx.x + u * NT, NV - 1)];
			if (nc > 0)
				load_half(0, ns0);
#pragma unroll
			for (int u = 0; u < UI; ++u)
				if ((int) threadIdx.x + u * NT < NV)
					idst[threadIdx.x + u * NT] = va[u];
		}
		FH_TT(0);
#pragma unroll 1
		for (int e0 = UI * NT + (int) threadIdx.x; e0 < NV; e0 += UI * NT) {
			v4i va[UI];
#pragma unroll
			for (int u = 0; u < UI; ++u)
				va[u] = isrc[min(e0 + u * NT, NV - 1)];
#pragma unroll
			for (int u = 0; u < UI; ++u)
				if (e0 + u * NT < NV)
					idst[e0 + u * NT] = va[u];
		}
	}
	__syncthreads(); // the image is in LDS
	FH_TT(1);
	if (nc <= 0)
		return; // the trailing wavefronts of the last workgroup helped with the image only

	// One 64-row half, X resident in the tile (Xs[row * XP + rhs]); `bt0`: its first tile row in the 128-block.
	//   per 16-row tile bi:  C (accumulator layout) <- tile;  C -= T[bi][bj] X_bj for the solved tiles bj < bi on the
	//   matrix cores (A = the negated tile from the image, B = rows of X_bj read from the tile);  C -> tile;  then the
	//   16 x 16 diagonal tile by substitution, lane = right-hand side (diagonal as a reciprocal, triangular_solve.rs:113)
	typedef typename Mfma<T>::acc_t acc_t;
	const int l15 = lane & 15, lhi = lane >> 4;
	const bool rl = lane < RW; // lanes that own a right-hand side in the substitution
	auto solve_half = [&](int bt0) {
#pragma unroll 1
		for (int bi = 0; bi < 4; ++bi) {
			if (bi > 0) {
				acc_t C[JT];
#pragma unroll
				for (int jt = 0; jt < JT; ++jt)
#pragma unroll
					for (int r = 0; r < 4; ++r)
						C[jt][r] = Xs[(16 * bi + Mfma<T>::row(r, lhi)) * XP + 16 * jt + l15];
#pragma unroll 1
				for (int bj = 0; bj < bi; ++bj) {
					const T *At = Ls + P::od_tile(bt0 + bi, bt0 + bj);
#pragma unroll
					for (int kk = 0; kk < 4; ++kk) {
						const T av = At[kk * 64 + lane];
#pragma unroll
						for (int jt = 0; jt < JT; ++jt)
							C[jt] = Mfma<T>::run(av, Xs[(16 * bj + 4 * kk + lhi) * XP + 16 * jt + l15], C[jt]);
					}
				}
#pragma unroll
				for (int jt = 0; jt < JT; ++jt)
#pragma unroll
					for (int r = 0; r < 4; ++r)
						Xs[(16 * bi + Mfma<T>::row(r, lhi)) * XP + 16 * jt + l15] = C[jt][r];
				__builtin_amdgcn_wave_barrier();
			}
			T x16[TP_TS];
#pragma unroll
			for (int i = 0; i < TP_TS; ++i)
				x16[i] = Xs[(16 * bi + i) * XP + (rl ? lane : 0)];
			const T *dg = Ls + P::OFF_DG + (bt0 + bi) * P::DG_SZ;
#pragma unroll
			for (int j = 0; j < TP_TS; ++j) {
				const T xj = x16[j] * dg[P::dg_off(j)];
				x16[j] = xj;
#pragma unroll
				for (int i = j + 1; i < TP_TS; ++i)
					x16[i] = fh_fma(-dg[P::dg_off(j) + (i - j)], xj, x16[i]);
			}
			if (rl) {
#pragma unroll
				for (int i = 0; i < TP_TS; ++i)
					Xs[(16 * bi + i) * XP + lane] = x16[i];
			}
			__builtin_amdgcn_wave_barrier();
		}
	};

	// ---- half 0
	solve_half(0);
	FH_TT(2);
	store_half(0, ns0);
	FH_TT(3);
	if (!two)
		return;
	// ---- half 1: rows 64 .. 127 come straight from X in accumulator layout, C -= T[4 + bi'][bj] X_bj against the solved
	// top half (still in the tile) on the matrix cores, all four tiles of C in registers; only then do they replace the
	// top half in the tile and get solved like it
	{
		acc_t C[4][JT];
#pragma unroll
		for (int bi = 0; bi < 4; ++bi)
#pragma unroll
			for (int jt = 0; jt < JT; ++jt)
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const int row = TP_H + 16 * bi + Mfma<T>::row(r, lhi), rhs = c0 + 16 * jt + l15;
					const bool in = row < n && rhs < nrhs;
					const T t = Xp[in ? (idx_t) row * xss + (idx_t) rhs * xcs : (idx_t) 0];
					C[bi][jt][r] = in ? t : (T) 0;
				}
		FH_TT(4);
#pragma unroll 1
		for (int bj = 0; bj < 4; ++bj) {
#pragma unroll
			for (int kk = 0; kk < 4; ++kk) {
				T bv[JT];
#pragma unroll
				for (int jt = 0; jt < JT; ++jt)
					bv[jt] = Xs[(16 * bj + 4 * kk + lhi) * XP + 16 * jt + l15];
#pragma unroll
				for (int bi = 0; bi < 4; ++bi) {
					const T av = Ls[P::od_tile(4 + bi, bj) + kk * 64 + lane];
#pragma unroll
					for (int jt = 0; jt < JT; ++jt)
						C[bi][jt] = Mfma<T>::run(av, bv[jt], C[bi][jt]);
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int bi = 0; bi < 4; ++bi)
#pragma unroll
			for (int jt = 0; jt < JT; ++jt)
#pragma unroll
				for (int r = 0; r < 4; ++r)
					Xs[(16 * bi + Mfma<T>::row(r, lhi)) * XP + 16 * jt + l15] = C[bi][jt][r];
		__builtin_amdgcn_wave_barrier();
	}
	FH_TT(5);
	solve_half(4);
	FH_TT(6);
	store_half(TP_H, ns1);
	FH_TT(7);
#ifdef FH_PANEL_TIMING
	if (blockIdx.x == 0 && threadIdx.x == 0)
		atomicAdd(&g_leaf128_timing[15], 1ull);
#endif
}

template <typename T, int RW> static size_t trsm_leaf128_lds()
{
	return TriPack<T>::BYTES + (size_t) TL_NW * TP_H * (RW + 1) * sizeof(T);
}

template <typename T, bool DIRECT, int RW>
static void trsm_leaf128_go(const T *src, idx_t n, MatV<T> X, int along_rhs, idx_t trs, idx_t tcs, int unit)
{
	// raise the dynamic LDS limit once per DEVICE and instance: the attribute belongs to the device that is current at the
	// time of the call, and a thread bound to a second GPU must not launch without it (atomic flags: several threads)
	static std::atomic<unsigned long long> attr_done{0}; // bit d: device d
	const size_t lds = trsm_leaf128_lds<T, RW>();
	const int dev = ctx().device;
	const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
	if (bit == 0 || !(attr_done.load(std::memory_order_acquire) & bit)) {
		FH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&trsm_leaf128_kernel<T, DIRECT, RW>), hipFuncAttributeMaxDynamicSharedMemorySize,
					   (int) lds));
		attr_done.fetch_or(bit, std::memory_order_release);
	}
	const idx_t k = X.ncols;
	const dim3 grid((unsigned) ((k + TL_NW * RW - 1) / (TL_NW * RW)));
	hipLaunchKernelGGL((trsm_leaf128_kernel<T, DIRECT, RW>), grid, dim3(TL_NT), lds, ctx().stream, src, (int) n, X.p, X.rs, X.cs, (int) k, along_rhs, trs,
			   tcs, unit);
	FH_HIP(hipGetLastError());
}

// X (n <= 128 rows) <- T^-1 X with the packed image of T, or (img == nullptr) with the triangle Ld itself.
// Four wavefronts per workgroup (one per SIMD), 16 right-hand sides per wavefront while that still gives at most one
// workgroup per compute unit of the stream, else 32: the matrix-core phases of a wavefront shrink with its share of the
// right-hand sides, the substitution phases do not care (they wait for LDS broadcasts).
template <typename T> static void trsm_leaf128_launch(const T *img, MatV<T> X, MatV<const T> Ld = MatV<const T>{nullptr, 0, 0, 0, 0}, bool unit = false)
{
	const idx_t n = X.nrows, k = X.ncols;
	if (n == 0 || k == 0)
		return;
	FH_CHECK(n <= TP_NB && k < (1L << 31), "trsm leaf: shape");
	auto ab = [](idx_t v) { return v < 0 ? -v : v; };
	const int along_rhs = ab(X.cs) <= ab(X.rs) ? 1 : 0;
	const bool narrow = (k + TL_NW * 16 - 1) / (TL_NW * 16) <= (idx_t) ctx().stream_cus();
	if (img) {
		if (narrow)
			trsm_leaf128_go<T, false, 16>(img, n, X, along_rhs, 0, 0, 0);
		else
			trsm_leaf128_go<T, false, 32>(img, n, X, along_rhs, 0, 0, 0);
	} else {
		if (narrow)
			trsm_leaf128_go<T, true, 16>(Ld.p, n, X, along_rhs, Ld.rs, Ld.cs, unit ? 1 : 0);
		else
			trsm_leaf128_go<T, true, 32>(Ld.p, n, X, along_rhs, Ld.rs, Ld.cs, unit ? 1 : 0);
	}
}

// W == nullptr: the leaves pack their diagonal block themselves (DIRECT), `unit` says how to read its diagonal
template <typename T> static void trsm_rec(MatV<const T> L, MatV<T> X, const T *W, idx_t b0, bool unit = false)
{
	const idx_t n = L.nrows, k = X.ncols;
	if (n <= TRSM_IB) {
		if (W)
			trsm_leaf128_launch<T>(W + (size_t) b0 * TriPack<T>::SIZE, X);
		else
			trsm_leaf128_launch<T>(nullptr, X, L, unit);
		return;
	}
	const idx_t nblk = (n + TRSM_IB - 1) / TRSM_IB;
	const idx_t top = (nblk / 2) * TRSM_IB;
	MatV<T> Xt = X.sub(0, 0, top, k), Xb = X.sub(top, 0, n - top, k);
	trsm_rec<T>(L.sub(0, 0, top, top), Xt, W, b0, unit);
	gemm_dev<T>(Xb, DST_FULL, true, L.sub(top, 0, n - top, top), Xt.c(), (T) -1);
	trsm_rec<T>(L.sub(top, top, n - top, n - top), Xb, W, b0 + top / TRSM_IB, unit);
}

// X <- L^-1 X with the packed images of L's 128 x 128 diagonal blocks already in W (block i at
// W + i * TriPack<T>::SIZE): the Cholesky leaves produce them as a by-product (potrf.hip).
template <typename T> void trsm_lower_pre_dev(MatV<const T> L, MatV<T> X, const T *W)
{
	FH_CHECK(L.nrows == L.ncols && X.nrows == L.nrows, "trsm: shape mismatch");
	if (L.nrows == 0 || X.ncols == 0)
		return;
	trsm_rec<T>(L, X, W, 0);
}

// triangular_solve.rs:200-215
static idx_t trsm_block_size(idx_t n)
{
	const idx_t base_rem = n / 2;
	idx_t r;
	if (n >= 32)
		r = (base_rem + 15) / 16 * 16;
	else if (n >= 16)
		r = (base_rem + 7) / 8 * 8;
	else if (n >= 8)
		r = (base_rem + 3) / 4 * 4;
	else
		r = base_rem;
	return n - r;
}

template <typename T> void trsm_lower_dev(MatV<const T> L, bool unit, MatV<T> X)
{
	FH_CHECK(L.nrows == L.ncols && X.nrows == L.nrows, "trsm: shape mismatch");
	const idx_t n = L.nrows, k = X.ncols;
	if (n == 0 || k == 0)
		return;
	if (n <= TRSM_IB && (n > 64 || k >= 64)) { // one block: the leaf packs the triangle itself (no packing launch)
		trsm_leaf128_launch<T>(nullptr, X, L, unit);
		return;
	}
	if (n > 64 || k >= 64) { // substitution leaves on the diagonal blocks + MFMA products off the diagonal
		// every leaf packs its own diagonal block from the triangle (as fast as copying a prepared image, and one launch
		// less on a chain that is all launches) (a separate packing launch is kept as the other branch)
		const bool prepack = false;
		if (prepack) {
			const idx_t nblk = (n + TRSM_IB - 1) / TRSM_IB;
			Scratch wb((size_t) nblk * TriPack<T>::BYTES);
			trsm_pack_dev<T>(L, unit, wb.as<T>());
			trsm_rec<T>(L, X, wb.as<T>(), 0);
		} else {
			trsm_rec<T>(L, X, nullptr, 0, unit);
		}
		return;
	}
	// tiny solve: one wavefront stages the triangle itself (a single launch)
	auto ab = [](idx_t v) { return v < 0 ? -v : v; };
	const int along_rhs = ab(X.cs) <= ab(X.rs) ? 1 : 0;
	FH_CHECK(k < (1L << 31), "trsm: too many right-hand sides");
	hipLaunchKernelGGL(trsm_leaf_kernel<T>, dim3((unsigned) ((k + 63) / 64)), dim3(64), 0, ctx().stream, L.p, L.rs, L.cs, (int) n,
			   unit ? 1 : 0, X.p, X.rs, X.cs, (int) k, along_rhs);
	FH_HIP(hipGetLastError());
}

template <typename T> void trsm_upper_dev(MatV<const T> U, bool unit, MatV<T> X)
{
	trsm_lower_dev<T>(U.rev_rows().rev_cols(), unit, X.rev_rows());
}

template void trsm_lower_pre_dev<double>(MatV<const double>, MatV<double>, const double *);
template void trsm_lower_pre_dev<float>(MatV<const float>, MatV<float>, const float *);
template void trsm_lower_dev<double>(MatV<const double>, bool, MatV<double>);
template void trsm_lower_dev<float>(MatV<const float>, bool, MatV<float>);
template void trsm_upper_dev<double>(MatV<const double>, bool, MatV<double>);
template void trsm_upper_dev<float>(MatV<const float>, bool, MatV<float>);

} // namespace fh
