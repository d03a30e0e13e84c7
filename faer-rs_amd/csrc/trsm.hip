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
#include "common.h"
#include "lds_blocks.h"
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
// trsm_leaf128_kernel: one workgroup = 2 wavefronts, each wavefront 64 right-hand sides, one per lane.
//   * the packed image of the diagonal block (trsm_pack.h: two packed 64 x 64 triangles, the 64 x 64 block
//     between them, reciprocal diagonal) is copied to LDS with 16-byte loads; every multiplier is then a
//     wave-uniform LDS broadcast;
//   * half 0: the lane's 64 values sit in registers, column-oriented substitution (2016 FMAs, the diagonal
//     enters as a reciprocal like triangular_solve.rs:113);  half 1: b_i - sum_j t_ij x_j for the 64 rows below
//     with the solved x_j still in registers (64 x 64 FMAs per lane, four partial sums), then the same
//     substitution code on the second triangle.  fp64 vector FMA runs at the MFMA rate on gfx950, so nothing
//     is lost against a matrix-core formulation and no layout change is needed between the phases;
//   * global accesses run along whichever stride of X is the small one: directly when that is the right-hand
//     side index (rows of a transposed Cholesky panel), through a padded LDS tile otherwise (columns of X).
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

constexpr int TL_XP = TP_H + 1; // pitch of the per-wave 64 x 64 exchange tile

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
	for (int e = threadIdx.x; e < P::TRI; e += blockDim.x) { // the strictly upper parts of the diagonal 8 x 8 blocks
		img[P::OFF_T00 + e] = (T) 0;
		img[P::OFF_T11 + e] = (T) 0;
	}
	for (int e = threadIdx.x; e < P::PAD; e += blockDim.x)
		img[P::SIZE - P::PAD + e] = (T) 0;
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
			const T val = i == j ? ((unit || i >= nb) ? (T) 1 : (T) 1 / v[u]) : v[u]; // the diagonal enters as its reciprocal
			if (i < TP_H)
				img[P::OFF_T00 + P::tri_pos(i, j)] = val;
			else if (j >= TP_H)
				img[P::OFF_T11 + P::tri_pos(i - TP_H, j - TP_H)] = val;
			else
				img[P::OFF_T10 + (i - TP_H) * TP_H + j] = val;
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

// ---- software-pipelined LDS streams -------------------------------------------------------------------------------
// Every multiplier of the substitution is a wave-uniform LDS broadcast.  Left to the compiler, each pair of FMAs waits
// for its own ds_read_b128 (~100 cycles with one wavefront per SIMD: 24 cycles per FMA measured, profiles/
// r02_trsm_leaf.txt).  The packed triangle is therefore consumed as ONE linear stream of 16-byte reads with TL_D of
// them in flight: read R is waited for with lgkmcnt(TL_D - 1), its FMAs issue, read R + TL_D goes out into the same
// ring register.  The reads and waits are inline asm (the compiler neither reorders nor counts them); LDS operations
// complete in order, so any LDS traffic the compiler adds around them only makes the waits stricter.
constexpr int TL_D = 16;

template <typename T> struct V16;
template <> struct V16<double> {
	typedef double type __attribute__((ext_vector_type(2)));
};
template <> struct V16<float> {
	typedef float type __attribute__((ext_vector_type(4)));
};
template <int OFF, typename V> static __device__ __forceinline__ void lds_read128(V &dst, unsigned base)
{
	asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(OFF));
}
template <int CNT, typename V> static __device__ __forceinline__ void lds_wait(V &v)
{
	asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(CNT));
}

// x <- tri^-1 x.  One group of 8 columns per iteration of a run-time loop (trsm_pack.h): the 8 x 8 diagonal block
// first (column oriented, the diagonal enters as a reciprocal), then the blocks of 8 rows below it -- the code of all
// seven is there, the triangle leaves early -- and the registers rotate by 8 so that the next group is at 0 .. 7 again.
// Linear read index inside a group: L = ((pb * 8 + jj) * 8) / EPR + r.
template <typename T, int PB, int JJ, int RD> struct SubstRead {
	typedef TriPack<T> P;
	typedef typename V16<T>::type V;
	static constexpr int EPR = P::ALIGN;
	static constexpr int RPC = 8 / EPR; // reads per column of a block
	static constexpr int L = (PB * 8 + JJ) * RPC + RD;
	static __device__ __forceinline__ void run(T (&x)[TP_H], T (&xj)[8], V (&ring)[TL_D], unsigned base)
	{
		constexpr int S = L % TL_D;
		lds_wait<TL_D - 1>(ring[S]);
#pragma unroll
		for (int u = 0; u < EPR; ++u) {
			const int e = RD * EPR + u; // row inside the block
			if (PB == 0) {
				if (e == JJ) {
					xj[JJ] = x[JJ] * ring[S][u];
					x[JJ] = xj[JJ];
				} else if (e > JJ) {
					x[e] = fh_fma(-ring[S][u], xj[JJ], x[e]);
				}
			} else {
				x[PB * 8 + e] = fh_fma(-ring[S][u], xj[JJ], x[PB * 8 + e]);
			}
		}
		lds_read128<(L + TL_D) * 16>(ring[S], base); // may run past the group: harmless read-ahead (image pad)
		if constexpr (RD + 1 < RPC)
			SubstRead<T, PB, JJ, RD + 1>::run(x, xj, ring, base);
		else if constexpr (JJ + 1 < 8)
			SubstRead<T, PB, JJ + 1, 0>::run(x, xj, ring, base);
	}
};
template <typename T, int PB, int NBLK> struct SubstBlocks {
	typedef typename V16<T>::type V;
	static __device__ __forceinline__ void run(T (&x)[TP_H], T (&xj)[8], V (&ring)[TL_D], unsigned base)
	{
		if constexpr (PB < NBLK) {
			SubstRead<T, PB, 0, 0>::run(x, xj, ring, base);
			SubstBlocks<T, PB + 1, NBLK>::run(x, xj, ring, base);
		}
	}
};
template <typename T, int R> struct PipeFill { // the first TL_D reads of a stream starting at `base`
	typedef typename V16<T>::type V;
	static __device__ __forceinline__ void run(V (&ring)[TL_D], unsigned base)
	{
		if constexpr (R < TL_D) {
			lds_read128<R * 16>(ring[R], base);
			PipeFill<T, R + 1>::run(ring, base);
		}
	}
};
template <typename T, int R> struct PipeDrain {
	typedef typename V16<T>::type V;
	static __device__ __forceinline__ void run(V (&ring)[TL_D])
	{
		if constexpr (R < TL_D) {
			lds_wait<0>(ring[R]);
			PipeDrain<T, R + 1>::run(ring);
		}
	}
};
// One group: the diagonal 8 x 8 block, then NBLK - 1 blocks of 8 rows below it with NO test whether the triangle
// has that many left: register positions past the end of the triangle are dead (their rows went to `out` when
// they were finished), the FMAs on them and the multipliers read for them (whatever follows the group in the
// image) are wasted work that keeps the code regular.  Finished rows 8 g .. 8 g + 7 go to out[(8 g + k) * TL_XP].
template <typename T, int NBLK>
static __device__ __forceinline__ void tl_subst_group(T (&x)[TP_H], typename V16<T>::type (&ring)[TL_D], const T *tri, int g, T *out)
{
	typedef TriPack<T> P;
	const unsigned base = (unsigned) (size_t) (tri + P::goff(g)); // LDS byte address of the group
	T xj[8];
	PipeFill<T, 0>::run(ring, base);
	SubstBlocks<T, 0, NBLK>::run(x, xj, ring, base);
	// drain the read-ahead while the ring registers are still live (a read landing in a register the compiler has
	// reused for something else would corrupt it: the reads are asm, their unused outputs look dead to it)
	PipeDrain<T, 0>::run(ring);
#pragma unroll
	for (int k = 0; k < 8; ++k)
		out[(g * 8 + k) * TL_XP] = xj[k];
#pragma unroll
	for (int c = 0; c + 8 < TP_H; ++c)
		x[c] = x[c + 8];
}
// x <- tri^-1 x; the solution goes to out[row * TL_XP] (the caller's column of the exchange tile), x is destroyed.
// Groups 0 .. 3 run the 8-block code (6.5 needed on average), groups 4 .. 7 a 4-block copy (2.5 needed).
template <typename T> static __device__ __forceinline__ void tl_subst(T (&x)[TP_H], const T *tri, T *out)
{
	typename V16<T>::type ring[TL_D];
#pragma unroll 1
	for (int g = 0; g < 4; ++g)
		tl_subst_group<T, 8>(x, ring, tri, g, out);
#pragma unroll 1
	for (int g = 4; g < 8; ++g)
		tl_subst_group<T, 4>(x, ring, tri, g, out);
}

// acc[u] = sum_j T10(i0 + u, j) x_j for TL_EB rows at a time: the reads walk the TL_EB rows column pair by column pair
// (so that consecutive FMAs feed different accumulators), same pipelining
constexpr int TL_EB = 8;
template <typename T, int L> struct ElimRead {
	typedef TriPack<T> P;
	typedef typename V16<T>::type V;
	static constexpr int EPR = P::ALIGN;
	static constexpr int RPR = TP_H / EPR;	 // reads per row
	static constexpr int NL = TL_EB * RPR;	 // reads per block of rows
	static constexpr int off(int l) { return ((l % TL_EB) * TP_H + (l / TL_EB) * EPR) * (int) sizeof(T); } // row l % EB, column group l / EB
	static __device__ __forceinline__ void run(const T (&x)[TP_H], T (&acc)[TL_EB][2], V (&ring)[TL_D], unsigned base)
	{
		if constexpr (L < NL) {
			constexpr int S = L % TL_D;
			constexpr int u = L % TL_EB, kk = L / TL_EB;
			lds_wait<(L + TL_D <= NL ? TL_D - 1 : NL - 1 - L)>(ring[S]);
#pragma unroll
			for (int e = 0; e < EPR; ++e)
				acc[u][e & 1] = fh_fma(ring[S][e], x[kk * EPR + e], acc[u][e & 1]);
			if constexpr (L + TL_D < NL)
				lds_read128<off(L + TL_D)>(ring[S], base);
			ElimRead<T, L + 1>::run(x, acc, ring, base);
		}
	}
};
template <typename T, int L> struct ElimFill {
	typedef typename V16<T>::type V;
	static __device__ __forceinline__ void run(V (&ring)[TL_D], unsigned base)
	{
		if constexpr (L < TL_D) {
			lds_read128<ElimRead<T, 0>::off(L)>(ring[L], base);
			ElimFill<T, L + 1>::run(ring, base);
		}
	}
};

template <typename T>
__global__ __launch_bounds__(128) void trsm_leaf128_kernel(const T *__restrict__ img, int n, T *Xp, idx_t xss, idx_t xcs, int nrhs,
							   int lanes_along_rhs)
{
	typedef TriPack<T> P;
	extern __shared__ __attribute__((aligned(16))) unsigned char tl_smem[];
	T *Ls = reinterpret_cast<T *>(tl_smem); // packed image
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	T *Xs = Ls + P::SIZE + wave * (TP_H * TL_XP); // this wave's 64 x 65 exchange tile: Xs[s * TL_XP + c]
	const int c0 = (blockIdx.x * 2 + wave) * 64;
	const int nc = min(64, nrhs - c0); // may be <= 0 for the second wave of the last workgroup
	const bool act = lane < nc;

	// Memory phases are organised by ROUND TRIPS, not by data: every batch below is a set of independent loads in
	// flight together (a leaf spends ~1.5 us per dependent round trip when the chip is busy), and the code is kept
	// SMALL (run-time loops around 16- / 32-deep batches): the whole kernel has to stay inside the 64 KB instruction
	// cache, straight-line code runs at the speed of instruction fetch (profiles/r02_trsm_leaf.txt).
	typedef int v4i __attribute__((ext_vector_type(4)));
	const v4i *isrc = reinterpret_cast<const v4i *>(img);
	v4i *idst = reinterpret_cast<v4i *>(Ls);
	constexpr int NV = (int) (P::BYTES / 16), UI = 12; // image: batches of 12 vectors per thread
	const bool two = n > TP_H;
	const int ns0 = min(TP_H, n), ns1 = two ? n - TP_H : 0;

	// rows s0 .. s0+ns-1 of this wave's 64 right-hand sides -> tile (zero padded), lanes along the small stride
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
#pragma unroll
				for (int u = 0; u < 32; ++u)
					Xs[(i0 + u) * TL_XP + lane] = v[u];
			}
		} else {
#pragma unroll 1
			for (int c8 = 0; c8 < TP_H; c8 += 32) {
				T v[32];
#pragma unroll
				for (int u = 0; u < 32; ++u) {
					const bool in = c8 + u < nc && lane < ns;
					const T t = Xp[in ? (idx_t) (s0 + lane) * xss + (idx_t) (c0 + c8 + u) * xcs : (idx_t) 0];
					v[u] = in ? t : (T) 0;
				}
#pragma unroll
				for (int u = 0; u < 32; ++u)
					Xs[lane * TL_XP + c8 + u] = v[u];
			}
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
					Xp[(idx_t) (s0 + i) * xss + (idx_t) (c0 + lane) * xcs] = Xs[i * TL_XP + lane];
		} else {
#pragma unroll 1
			for (int c8 = 0; c8 < TP_H; c8 += 16) {
				T w[16];
#pragma unroll
				for (int u = 0; u < 16; ++u)
					w[u] = Xs[lane * TL_XP + c8 + u];
#pragma unroll
				for (int u = 0; u < 16; ++u)
					if (c8 + u < nc && lane < ns)
						Xp[(idx_t) (s0 + lane) * xss + (idx_t) (c0 + c8 + u) * xcs] = w[u];
			}
		}
		__builtin_amdgcn_wave_barrier();
	};

	FH_TT_DECL;
	// trip 1: first image batch + the top rows; then the rest of the image
	{
		v4i va[UI];
#pragma unroll
		for (int u = 0; u < UI; ++u)
			va[u] = isrc[min((int) threadIdx.x + u * 128, NV - 1)];
		load_half(0, ns0);
#pragma unroll
		for (int u = 0; u < UI; ++u)
			if ((int) threadIdx.x + u * 128 < NV)
				idst[threadIdx.x + u * 128] = va[u];
	}
	FH_TT(0);
#pragma unroll 1
	for (int e0 = UI * 128 + (int) threadIdx.x; e0 < NV; e0 += UI * 128) {
		v4i va[UI];
#pragma unroll
		for (int u = 0; u < UI; ++u)
			va[u] = isrc[min(e0 + u * 128, NV - 1)];
#pragma unroll
		for (int u = 0; u < UI; ++u)
			if (e0 + u * 128 < NV)
				idst[e0 + u * 128] = va[u];
	}
	__syncthreads(); // the image is in LDS
	FH_TT(1);
	if (nc <= 0)
		return; // the second wavefront of the last workgroup helped with the image only: the LDS pipe is the other one's

	// ---- half 0: substitution on T00 (lane = right-hand side, register = row); the solution goes back to the tile
	T y[TP_H];
#pragma unroll
	for (int i = 0; i < TP_H; ++i)
		y[i] = Xs[i * TL_XP + lane];
	tl_subst<T>(y, Ls + P::OFF_T00, Xs + lane);
	FH_TT(2);
	store_half(0, ns0);
	FH_TT(3);
	if (!two)
		return;
	// ---- half 1: b_i - sum_j T10(i, j) y_j with the solved y_j in registers and the rows of T10 as LDS broadcasts,
	// then the same substitution on T11
#pragma unroll
	for (int i = 0; i < TP_H; ++i)
		y[i] = Xs[i * TL_XP + lane];
	__builtin_amdgcn_wave_barrier();
	load_half(TP_H, ns1);
	FH_TT(4);
#pragma unroll 1
	for (int i0 = 0; i0 < TP_H; i0 += TL_EB) {
		typename V16<T>::type ring[TL_D];
		T acc[TL_EB][2];
#pragma unroll
		for (int u = 0; u < TL_EB; ++u)
			acc[u][0] = acc[u][1] = (T) 0;
		const unsigned base = (unsigned) (size_t) (Ls + P::OFF_T10 + i0 * TP_H);
		ElimFill<T, 0>::run(ring, base);
		ElimRead<T, 0>::run(y, acc, ring, base);
#pragma unroll
		for (int u = 0; u < TL_EB; ++u)
			Xs[(i0 + u) * TL_XP + lane] -= acc[u][0] + acc[u][1];
	}
	__builtin_amdgcn_wave_barrier();
	FH_TT(5);
#pragma unroll
	for (int i = 0; i < TP_H; ++i)
		y[i] = Xs[i * TL_XP + lane];
	tl_subst<T>(y, Ls + P::OFF_T11, Xs + lane);
	FH_TT(6);
	store_half(TP_H, ns1);
	FH_TT(7);
#ifdef FH_PANEL_TIMING
	if (blockIdx.x == 0 && threadIdx.x == 0)
		atomicAdd(&g_leaf128_timing[15], 1ull);
#endif
}

template <typename T> static size_t trsm_leaf128_lds()
{
	return TriPack<T>::BYTES + (size_t) 2 * TP_H * TL_XP * sizeof(T);
}

// X (n <= 128 rows) <- T^-1 X with the packed image of T
template <typename T> static void trsm_leaf128_launch(const T *img, MatV<T> X)
{
	const idx_t n = X.nrows, k = X.ncols;
	if (n == 0 || k == 0)
		return;
	FH_CHECK(n <= TP_NB && k < (1L << 31), "trsm leaf: shape");
	static bool attr_done = false; // raise the dynamic LDS limit once per process and type
	if (!attr_done) {
		FH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&trsm_leaf128_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
					   (int) trsm_leaf128_lds<T>()));
		attr_done = true;
	}
	auto ab = [](idx_t v) { return v < 0 ? -v : v; };
	const int along_rhs = ab(X.cs) <= ab(X.rs) ? 1 : 0;
	hipLaunchKernelGGL(trsm_leaf128_kernel<T>, dim3((unsigned) ((k + 127) / 128)), dim3(128), trsm_leaf128_lds<T>(), ctx().stream, img, (int) n,
			   X.p, X.rs, X.cs, (int) k, along_rhs);
	FH_HIP(hipGetLastError());
}

template <typename T> static void trsm_rec(MatV<const T> L, MatV<T> X, const T *W, idx_t b0)
{
	const idx_t n = L.nrows, k = X.ncols;
	if (n <= TRSM_IB) {
		trsm_leaf128_launch<T>(W + (size_t) b0 * TriPack<T>::SIZE, X);
		return;
	}
	const idx_t nblk = (n + TRSM_IB - 1) / TRSM_IB;
	const idx_t top = (nblk / 2) * TRSM_IB;
	MatV<T> Xt = X.sub(0, 0, top, k), Xb = X.sub(top, 0, n - top, k);
	trsm_rec<T>(L.sub(0, 0, top, top), Xt, W, b0);
	gemm_dev<T>(Xb, DST_FULL, true, L.sub(top, 0, n - top, top), Xt.c(), (T) -1);
	trsm_rec<T>(L.sub(top, top, n - top, n - top), Xb, W, b0 + top / TRSM_IB);
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
	if (n > 64 || k >= 64) { // packed diagonal blocks + substitution leaves + MFMA products off the diagonal
		const idx_t nblk = (n + TRSM_IB - 1) / TRSM_IB;
		Scratch wb((size_t) nblk * TriPack<T>::BYTES);
		trsm_pack_dev<T>(L, unit, wb.as<T>());
		trsm_rec<T>(L, X, wb.as<T>(), 0);
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
