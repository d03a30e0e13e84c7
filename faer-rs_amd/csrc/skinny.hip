// Tall-skinny shapes of the product for gfx950: one dimension in the hundreds of thousands, the other two at most
// a few dozen.  They are what the block-reflector steps of a tall QR are made of (faer/src/linalg/householder.rs:
// 438-604 apply_block_householder..., qr/no_pivoting/factor.rs:160-301: V^H A, V^H V and A -= V (T V^H A) with
// 8 .. 32 columns on each side) and they are pure HBM streams: a 128 x 128 MFMA tile would be 94 % padding and its
// K loop one memory round trip per 16 rows.  Two streaming kernels instead:
//   update:  C (M x N) <- [C +] alpha A (M x K) B (K x N),  K <= 16, N <= 32, M >= 256 (measured: ahead of the MFMA tile from there), unit stride along m in A and C.
//            One thread per row (R rows in flight): its K values of A stay in registers, B sits in LDS and is read
//            as wave-uniform broadcasts, every C element is loaded and stored exactly once, lanes along m.
//   reduce:  C (M x N) <- [C +] alpha A (M x K) B (K x N),  M, N <= 16, K large, unit stride along k in A and B.
//            8 x 8 register blocks of C per thread, lanes along k, the K range split over workgroups; the per-lane
//            partial blocks meet in LDS, the workgroup partials in a workspace that splitk_reduce adds in a fixed
//            order (deterministic, like every split reduction of this library).
// Algorithmic bytes: (M K + 2 M N) sizeof(T) for the update, K (M + N) sizeof(T) for the reduction; roofline = HBM.
#include "common.h"

namespace fh {

static inline idx_t iabs3(idx_t x) { return x < 0 ? -x : x; }

// ------------------------------------------------------------------------------------------------
// update
// ------------------------------------------------------------------------------------------------
template <typename T, int KMAX, int R>
__global__ __launch_bounds__(256) void skinny_update_kernel(int M, int N, int K, T *c, idx_t crs, idx_t ccs, const T *__restrict__ a, idx_t ars,
							    idx_t acs, const T *__restrict__ b, idx_t brs, idx_t bcs, T alpha, int add)
{
	__shared__ T Bs[KMAX * 33];
	const int tid = threadIdx.x;
	for (int e = tid; e < KMAX * 32; e += 256) {
		const int k = e / 32, n = e % 32;
		Bs[k * 33 + n] = (k < K && n < N) ? b[(idx_t) k * brs + (idx_t) n * bcs] : (T) 0;
	}
	const int m0 = blockIdx.x * 256 * R + tid;
	T av[R][KMAX];
#pragma unroll
	for (int r = 0; r < R; ++r) {
		const int m = min(m0 + r * 256, M - 1); // clamped: unconditional loads
#pragma unroll
		for (int k = 0; k < KMAX; ++k)
			av[r][k] = a[(idx_t) m * ars + (idx_t) min(k, K - 1) * acs];
	}
	__syncthreads();
	constexpr int U = 4; // columns of C in flight per row
	for (int n0 = 0; n0 < N; n0 += U) {
		T cv[R][U];
		if (add) {
#pragma unroll
			for (int r = 0; r < R; ++r) {
				const int m = min(m0 + r * 256, M - 1);
#pragma unroll
				for (int u = 0; u < U; ++u)
					cv[r][u] = c[(idx_t) m * crs + (idx_t) min(n0 + u, N - 1) * ccs];
			}
		}
		T acc[R][U];
#pragma unroll
		for (int r = 0; r < R; ++r)
#pragma unroll
			for (int u = 0; u < U; ++u)
				acc[r][u] = (T) 0;
#pragma unroll
		for (int k = 0; k < KMAX; ++k) {
			if (k < K) { // wave uniform
				T bv[U];
#pragma unroll
				for (int u = 0; u < U; ++u)
					bv[u] = Bs[k * 33 + n0 + u]; // n0 + u <= 31; columns >= N hold zeros
#pragma unroll
				for (int r = 0; r < R; ++r)
#pragma unroll
					for (int u = 0; u < U; ++u)
						acc[r][u] = fh_fma(av[r][k], bv[u], acc[r][u]);
			}
		}
#pragma unroll
		for (int r = 0; r < R; ++r) {
			const int m = m0 + r * 256;
#pragma unroll
			for (int u = 0; u < U; ++u)
				if (m < M && n0 + u < N)
					c[(idx_t) m * crs + (idx_t) (n0 + u) * ccs] = add ? fh_fma(alpha, acc[r][u], cv[r][u]) : alpha * acc[r][u];
		}
	}
}

template <typename T, int KMAX, int R>
static void launch_update(MatV<T> C, bool add, MatV<const T> A, MatV<const T> B, T alpha)
{
	const idx_t M = C.nrows;
	const unsigned blocks = (unsigned) ((M + 256 * R - 1) / (256 * R));
	hipLaunchKernelGGL((skinny_update_kernel<T, KMAX, R>), dim3(blocks), dim3(256), 0, ctx().stream, (int) M, (int) C.ncols, (int) A.ncols, C.p,
			   C.rs, C.cs, A.p, A.rs, A.cs, B.p, B.rs, B.cs, alpha, add ? 1 : 0);
	FH_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// reduce
// ------------------------------------------------------------------------------------------------
// grid.x = K slices, grid.y = 8 x 8 blocks of C (row block fastest); ws: slice z at ws + z * M * N, column major M x N
template <typename T>
__global__ __launch_bounds__(256) void skinny_reduce_kernel(int M, int N, int K, const T *__restrict__ a, idx_t ars, idx_t acs,
							    const T *__restrict__ b, idx_t brs, idx_t bcs, int k_per_slice, T *ws)
{
	__shared__ T red[64 * 65];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int mblocks = (M + 7) / 8;
	const int mi = (blockIdx.y % mblocks) * 8, nj = (blockIdx.y / mblocks) * 8;
	const int k0 = blockIdx.x * k_per_slice, k1 = min(K, k0 + k_per_slice);
	T acc[8][8];
#pragma unroll
	for (int i = 0; i < 8; ++i)
#pragma unroll
		for (int j = 0; j < 8; ++j)
			acc[i][j] = (T) 0;
	// rows (of the tall operands) p = k0 + tid, + 256, ...; two per iteration => 32 independent loads in flight
	for (int p0 = k0 + tid; p0 < k1; p0 += 512) {
		T av[2][8], bv[2][8];
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int p = min(p0 + h * 256, K - 1);
#pragma unroll
			for (int i = 0; i < 8; ++i)
				av[h][i] = a[(idx_t) min(mi + i, M - 1) * ars + (idx_t) p * acs];
#pragma unroll
			for (int j = 0; j < 8; ++j)
				bv[h][j] = b[(idx_t) p * brs + (idx_t) min(nj + j, N - 1) * bcs];
		}
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			if (p0 + h * 256 < k1) {
#pragma unroll
				for (int i = 0; i < 8; ++i)
#pragma unroll
					for (int j = 0; j < 8; ++j)
						acc[i][j] = fh_fma(av[h][i], bv[h][j], acc[i][j]);
			}
		}
	}
	// 256 partial blocks -> 1: wave by wave through LDS (pitch 65: conflict free both ways), lanes of wave 0 own one
	// element each; fixed order
	T tot = (T) 0;
	for (int w = 0; w < 4; ++w) {
		if (wave == w) {
#pragma unroll
			for (int i = 0; i < 8; ++i)
#pragma unroll
				for (int j = 0; j < 8; ++j)
					red[lane * 65 + i * 8 + j] = acc[i][j];
		}
		__syncthreads();
		if (tid < 64) {
			T s[4] = {0, 0, 0, 0};
#pragma unroll
			for (int l = 0; l < 64; l += 4)
#pragma unroll
				for (int u = 0; u < 4; ++u)
					s[u] += red[(l + u) * 65 + tid];
			tot += (s[0] + s[1]) + (s[2] + s[3]);
		}
		__syncthreads();
	}
	if (tid < 64) {
		const int m = mi + tid / 8, n = nj + tid % 8;
		if (m < M && n < N)
			ws[((size_t) blockIdx.x * N + n) * M + m] = tot;
	}
}

// ------------------------------------------------------------------------------------------------
// dispatch: true if the product was done here
// ------------------------------------------------------------------------------------------------
template <typename T> bool skinny_dev(MatV<T> C, bool add, MatV<const T> A, MatV<const T> B, T alpha)
{
	idx_t m = C.nrows, n = C.ncols, k = A.ncols;
	const idx_t LONG = 256;
	// ---- update, possibly on the transposed problem (C^T = B^T A^T)
	if (n >= LONG && m <= 32 && k <= 16 && iabs3(C.cs) == 1 && iabs3(B.cs) == 1) {
		MatV<T> Ct = C.t();
		MatV<const T> At = B.t(), Bt = A.t();
		C = Ct;
		A = At;
		B = Bt;
		std::swap(m, n);
	}
	// (K = 32 is left to the MFMA kernel: measured 133 us here against 94 us there at 5e5 x 32 x 32 fp32)
	if (m >= LONG && m < (1L << 31) && n <= 32 && k <= 16 && iabs3(C.rs) == 1 && iabs3(A.rs) == 1) {
		// rows per thread: few -- the launch needs thousands of workgroups in flight to cover the HBM latency
		// (measured at 5e5 rows: 8 rows per thread = 245 workgroups ran at 1.5 TB/s)
		// (2 rows per thread: 1 and 4 measured slower in round 1)
		if (k <= 8)
			launch_update<T, 8, 2>(C, add, A, B, alpha);
		else
			launch_update<T, 16, 2>(C, add, A, B, alpha);
		return true;
	}
	// ---- reduce
	if (k >= LONG && k < (1L << 31) && m <= 16 && n <= 16 && iabs3(A.cs) == 1 && iabs3(B.rs) == 1) {
		const idx_t blocks = ((m + 7) / 8) * ((n + 7) / 8);
		const idx_t wgs = 512;
		idx_t slices = wgs / blocks;
		idx_t kps = (k + slices - 1) / slices;
		kps = (kps + 511) / 512 * 512;
		slices = (k + kps - 1) / kps;
		Scratch wsb((size_t) slices * (size_t) m * (size_t) n * sizeof(T));
		hipLaunchKernelGGL(skinny_reduce_kernel<T>, dim3((unsigned) slices, (unsigned) blocks), dim3(256), 0, ctx().stream, (int) m, (int) n, (int) k,
				   A.p, A.rs, A.cs, B.p, B.rs, B.cs, (int) kps, wsb.as<T>());
		FH_HIP(hipGetLastError());
		splitk_reduce_dev<T>(C, wsb.as<T>(), (int) slices, alpha, add);
		return true;
	}
	return false;
}

template bool skinny_dev<double>(MatV<double>, bool, MatV<const double>, MatV<const double>, double);
template bool skinny_dev<float>(MatV<float>, bool, MatV<const float>, MatV<const float>, float);

} // namespace fh
