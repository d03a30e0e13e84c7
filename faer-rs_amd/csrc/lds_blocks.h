// LDS-resident 128 x 128 block routines shared by the Cholesky leaf (potrf.hip) and the triangular
// inversion kernel behind TRSM (trsm.hip).  One 512-thread workgroup (8 wavefronts, 2 per SIMD, up to 256
// VGPRs each) owns one block.
//
// Block image: column major, S[c * LDS_LDP + r], pitch 130 elements.  With that pitch both MFMA operand
// shapes -- "16 consecutive rows x 4 columns" (A-type) and "4 consecutive rows x 16 columns" (B-type) -- stay
// at most 2-way bank conflicted (MI355X_MICROARCH.md, LDS table: ds_read_b64 bank = (addr / 4) mod 64), and a
// full fp64 block is 133,120 B of the CU's 160 KiB.
#pragma once
#include "common.h"
#include "mfma.h"

namespace fh {

constexpr int LDS_NB = 128;  // block dimension
constexpr int LDS_LDP = 130; // pitch (elements)
constexpr int LDS_NT = 512;  // threads per workgroup
constexpr int LDS_NW = LDS_NT / 64;

// wave-uniform broadcast of lane `src` (compile-time or uniform) through the scalar unit
static __device__ __forceinline__ double lane_bcast(double v, int src)
{
	int lo = __double2loint(v), hi = __double2hiint(v);
	lo = __builtin_amdgcn_readlane(lo, src);
	hi = __builtin_amdgcn_readlane(hi, src);
	return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ float lane_bcast(float v, int src)
{
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// acc = A(16 x K) * B(K x 16) for one wavefront; A starts at (ar, ac), B at (br, bc) of the block image.
template <typename T>
static __device__ __forceinline__ typename Mfma<T>::acc_t lds_tile_prod(const T *S, int ar, int ac, int br, int bc, int K, int lane)
{
	typedef typename Mfma<T>::acc_t acc_t;
	const int l15 = lane & 15, lhi = lane >> 4;
	acc_t acc = (acc_t) (T) 0;
	const T *pa = S + (ac + lhi) * LDS_LDP + ar + l15; // A[i = l15][k = lhi]
	const T *pb = S + (bc + l15) * LDS_LDP + br + lhi; // B[k = lhi][j = l15]
	for (int kk = 0; kk < K; kk += 4)
		acc = Mfma<T>::run(pa[kk * LDS_LDP], pb[kk], acc);
	return acc;
}

// writes sign * acc to the 16 x 16 tile at (cr, cc): register r of lane l is element (row(r, l >> 4), l & 15)
template <typename T>
static __device__ __forceinline__ void lds_tile_store(T *S, int cr, int cc, typename Mfma<T>::acc_t acc, T sign, int lane)
{
	const int l15 = lane & 15, lhi = lane >> 4;
#pragma unroll
	for (int r = 0; r < 4; ++r)
		S[(cc + l15) * LDS_LDP + cr + Mfma<T>::row(r, lhi)] = sign * acc[r];
}

// In-place inverse of the lower triangular 128 x 128 block in S (strict upper part must be zero on entry and
// is zero on exit; `unit` ignores the stored diagonal and produces a unit diagonal).  All 1024 threads call it.
//   level 16: the eight 16 x 16 diagonal blocks by forward substitution (one wavefront each, lane = column);
//   levels 32 / 64 / 128: inv([L11 0; L21 L22]) = [W11 0; -W22 L21 W11, W22] -- two chained MFMA products per
//   level, each output tile owned by one wavefront, the products overwrite L21 in place.
template <typename T> static __device__ void lds_tri_inv_inplace(T *S, int unit)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	{
		T w[16];
		const bool act = lane < 16;
		const int r0 = wave * 16; // LDS_NW == 8 diagonal blocks
		const int c = lane & 15;
		T dl = (T) 1;
		if (!unit)
			dl = (T) 1 / S[(r0 + c) * LDS_LDP + r0 + c];
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			T li[16];
#pragma unroll
			for (int k = 0; k < i; ++k)
				li[k] = S[(r0 + k) * LDS_LDP + r0 + i]; // L[i][k]: one address per wave (broadcast)
			T s = (i == c) ? (T) 1 : (T) 0;
#pragma unroll
			for (int k = 0; k < i; ++k)
				s = fh_fma(-li[k], w[k], s);
			w[i] = s * lane_bcast(dl, i);
			asm volatile("" ::: "memory"); // keep the next rows' loads from piling up in registers
		}
		__syncthreads();
		if (act) {
#pragma unroll
			for (int i = 0; i < 16; ++i)
				S[(r0 + c) * LDS_LDP + r0 + i] = w[i]; // w[i] == 0 for i < c
		}
		__syncthreads();
	}
#pragma unroll
	for (int h = 16; h <= 64; h *= 2) {
		const int ht = h / 16;			   // tiles per block side
		const int tpp = ht * ht;		   // tiles per pair
		const int total = (LDS_NB / (2 * h)) * tpp; // 4, 8, 16 output tiles: at most 2 per wavefront
		typename Mfma<T>::acc_t acc[2];
		int orow[2], ocol[2], base[2];
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			const int t = wave + q * LDS_NW;
			const int pair = t / tpp, tt = t % tpp;
			base[q] = pair * 2 * h;
			orow[q] = base[q] + h + (tt % ht) * 16; // output tile inside the L21 block
			ocol[q] = base[q] + (tt / ht) * 16;
		}
		// T = L21 * W11
#pragma unroll
		for (int q = 0; q < 2; ++q)
			if (wave + q * LDS_NW < total)
				acc[q] = lds_tile_prod<T>(S, orow[q], base[q], base[q], ocol[q], h, lane);
		__syncthreads();
#pragma unroll
		for (int q = 0; q < 2; ++q)
			if (wave + q * LDS_NW < total)
				lds_tile_store<T>(S, orow[q], ocol[q], acc[q], (T) 1, lane);
		__syncthreads();
		// W21 = -W22 * T
#pragma unroll
		for (int q = 0; q < 2; ++q)
			if (wave + q * LDS_NW < total)
				acc[q] = lds_tile_prod<T>(S, orow[q], base[q] + h, base[q] + h, ocol[q], h, lane);
		__syncthreads();
#pragma unroll
		for (int q = 0; q < 2; ++q)
			if (wave + q * LDS_NW < total)
				lds_tile_store<T>(S, orow[q], ocol[q], acc[q], (T) -1, lane);
		__syncthreads();
	}
}

// global -> LDS: lower triangle of the n x n matrix at A (any strides), identity padded to LDS_NB, zero above.
// Eight independent loads per thread are in flight at a time (one HBM round trip per batch, not per element).
template <typename T> static __device__ void lds_load_lower(T *S, const T *A, idx_t rs, idx_t cs, int n)
{
	constexpr int U = 8;
	static_assert(LDS_NB * LDS_NB % (LDS_NT * U) == 0, "load batches");
	for (int e0 = threadIdx.x; e0 < LDS_NB * LDS_NB; e0 += LDS_NT * U) {
		T v[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int e = e0 + u * LDS_NT;
			const int i = e % LDS_NB, k = e / LDS_NB;
			const bool in = i < n && k <= i;
			const T x = A[in ? (idx_t) i * rs + (idx_t) k * cs : (idx_t) 0]; // unconditional load, clamped address
			v[u] = in ? x : ((i == k) ? (T) 1 : (T) 0);
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int e = e0 + u * LDS_NT;
			S[(e / LDS_NB) * LDS_LDP + e % LDS_NB] = v[u];
		}
	}
}

// LDS -> global: the n x n leading part of the block image to a column major (ld) or strided destination;
// lower_only skips the strict upper part
template <typename T> static __device__ void lds_store_block(const T *S, T *A, idx_t rs, idx_t cs, int n, bool lower_only)
{
	constexpr int U = 8;
	for (int e0 = threadIdx.x; e0 < LDS_NB * LDS_NB; e0 += LDS_NT * U) {
		T v[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int e = e0 + u * LDS_NT;
			v[u] = S[(e / LDS_NB) * LDS_LDP + e % LDS_NB];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int e = e0 + u * LDS_NT;
			const int i = e % LDS_NB, k = e / LDS_NB;
			if (i < n && k < n && (!lower_only || k <= i))
				A[(idx_t) i * rs + (idx_t) k * cs] = v[u];
		}
	}
}

} // namespace fh
