// fp64 / fp32 GEMM for gfx950 on the MFMA pipe.
//
// Replaces the kernel behind faer's `gemm_call!` (faer/src/linalg/matmul/mod.rs:1312-1453, i.e.
// private_gemm_x86::gemm with DstKind::{Full,Lower,Upper}) -- SURVEY.md section 8a rows a6 / a12 / a14.
//
// Design (CDNA4-first, not a port of a CPU packing GEMM):
//   * one workgroup = WM x WN waves, block tile BM x BN, K step BK = 16;
//   * each wave owns a (BM/WM) x (BN/WN) sub-tile as TM x TN accumulators of v_mfma_{f64,f32}_16x16x4
//     (one VGPR pair / one VGPR per operand per lane, 64 / 32 cycles per instruction): the pipe is fed
//     with 8 LDS reads per 16 MFMAs, so LDS bandwidth is irrelevant and the kernel is MFMA-issue bound;
//   * A and B tiles are staged through LDS with a register prefetch of the next K tile (global loads
//     for tile t+1 are in flight while tile t is multiplied) and two LDS buffers => ONE barrier per K tile;
//   * operands may have ANY signed element strides.  Two loader shapes per operand pick the coalesced
//     direction: "MN-major" (unit stride along the m / n index, LDS image [k][mn], row pitch == 16 mod 32
//     elements) and "K-major" (unit stride along k, LDS image [mn][k], pitch BK+2): both images are
//     bank-conflict free for the ds_read_b64 / ds_read_b32 fragment reads (MI355X_MICROARCH.md, LDS table);
//   * the MFMA is issued with the operands swapped (D' = tile^T) so that a lane's 16 neighbours hold 16
//     consecutive rows of C: 128-byte contiguous stores into a column-major dst;
//   * blockIdx is remapped XCD-aware (each XCD's private L2 gets a contiguous run of tiles) and then
//     rastered in groups of 8 tile rows; DstKind::Lower enumerates only the tiles that touch the lower
//     triangle;
//   * deep-K products with few output tiles use split-K: the slices write raw partial sums to a workspace which a
//     second small kernel adds in a fixed order (deterministic; hardware atomics would make every run differ in
//     the last bits);
//   * the accumulate epilogue loads the old values of a 16-column group together before it stores them (one
//     memory round trip per group instead of one per element, see gemm_kernel_p);
//   * level-2 and tall-skinny shapes never reach this file's kernels: gemv.hip / skinny.hip stream them.
#include <type_traits>

#include "common.h"
#include "mfma.h"

namespace fh {

template <typename T> struct GemmArgs {
	int M, N, K;
	T *dst;
	idx_t drs, dcs;
	const T *a;
	idx_t ars, acs; // element (m,k) at a[m*ars + k*acs]
	const T *b;
	idx_t brs, bcs; // element (k,n) at b[k*brs + n*bcs]
	T alpha;
	int add;    // 1: dst += ; 0: dst =
	int lower;  // 1: only i >= j written
	int atomic; // split-K: 1 = atomicAdd(alpha * partial) (unused now), 2 = raw partial sums to `ws` (deterministic)
	T *ws;	    // split-K workspace: slice z at ws + z * M * N, column major M x N

	int k_per_split;
	int ntm, ntn;
	int tri_enum; // lower && square tiles: grid enumerates the lower triangle of tiles only
	const void *row_idx;
	const void *col_idx;
	int idx64;
	const T *diag;
	idx_t diag_stride;
	int a_struct, b_struct; // FaerBlock codes of lhs (m x k) and rhs (k x n); EXTRA kernels only
	int dst_strict;		// lower && strict: only i > j written
	int epi_serial;		// 1: accumulate epilogue as one read-modify-write per element (A/B switch, see gemm_kernel_p)
	int k_trim;		// GemmExtra::k_trim (pipelined kernel only)
	int tri_off;		// tri_enum: first tile of the enumeration (tiles of the skipped leading rows)
	int raster_g;		// tile rows per raster group (pipelined kernel)
	int stair_nb, stair_gap, stair_row0; // GemmExtra::stair_*: "lower" is tested against the column n + (n / stair_nb) * stair_gap - stair_row0
	// pipelined kernel, interior tiles of a plain column-major dst (see "fast tile I/O" there): 0 = off, 1 = replace (dst = alpha acc),
	// 2 / 3 = accumulate with alpha == +1 / -1: the accumulators START from the old dst values (-dst for 3, stored negated)
	int fast_io;
	// Tiles handed out through per-XCD counters instead of blockIdx (GemmExtra::ticket, pipelined kernel only): 8 zeroed device
	// ints.  XCD x owns the contiguous share of the `ticket_total` logical tile ids that xcd_remap would give it and hands them
	// out in order to whichever workgroup asks; a workgroup of the main launch (role 0, grid = all tiles) whose own XCD has
	// run dry takes from the next XCD that still has tiles, and returns at once when nobody has.  A HELPER launch (role 1, any
	// grid) is the same kernel with the same arguments on another stream -- the idle CUs of the look-ahead drivers' panel
	// stream -- whose workgroups take a tile only while more than `ticket_margin` remain in their XCD's share (so that the
	// helpers are done well before the main launch is) and never steal.
	int *ticket;
	int ticket_total, ticket_role, ticket_margin, helper_wgs;
};

// column index the lower-part test of dst uses (GemmExtra::stair_nb: a staircase instead of a diagonal)
template <typename G> static __device__ __forceinline__ int lower_col(const G &g, int n)
{
	return g.stair_nb ? n + (n / g.stair_nb) * g.stair_gap - g.stair_row0 : n;
}

// FaerBlock membership test (faer/src/linalg/matmul/triangular.rs:906-977)
static __device__ __forceinline__ bool in_block(int s, int i, int j)
{
	return s == 0 || ((s == 1) & (i >= j)) || ((s == 2) & (i <= j)) || (((s == 3) | (s == 5)) & (i > j)) ||
	       (((s == 4) | (s == 6)) & (i < j));
}

static __device__ __forceinline__ idx_t load_idx(const void *p, int idx64, int i)
{
	return idx64 ? (idx_t) reinterpret_cast<const unsigned long long *>(p)[i]
		     : (idx_t) reinterpret_cast<const unsigned int *>(p)[i];
}

// blockIdx.x -> tile coordinates.  (1) XCD remap: hardware deals block b to XCD b % 8, so give XCD x
// the contiguous range of logical ids [start_x, start_x + count_x) (bijective for any grid size);
// (2) logical id -> (tm, tn) in groups of 8 tile rows, column fastest inside the group's rows.
static __device__ __forceinline__ int xcd_remap(int b, int nblocks)
{
	const int q = nblocks >> 3, r = nblocks & 7;
	const int xcd = b & 7, idx = b >> 3;
	return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename T, int BM, int BN, int BK, int WM, int WN, bool AKM, bool BKM, bool EXTRA>
__global__ __launch_bounds__(WM *WN * 64, (WM * WN == 4 ? 2 : 1)) void gemm_kernel(const GemmArgs<T> g)
{
	constexpr int NT = WM * WN * 64;
	constexpr int WTM = BM / WM, WTN = BN / WN; // wave tile
	constexpr int TM = WTM / 16, TN = WTN / 16;
	constexpr int SA = AKM ? BK + 2 : BM + 16; // LDS pitch (elements)
	constexpr int SB = BKM ? BK + 2 : BN + 16;
	constexpr int A_SZ = AKM ? BM * SA : BK * SA;
	constexpr int B_SZ = BKM ? BN * SB : BK * SB;
	constexpr int A_CNT = BM * BK / NT, B_CNT = BN * BK / NT; // elements per thread per tile
	static_assert(BM * BK % NT == 0 && BN * BK % NT == 0, "tile/threads");
	static_assert(NT % BM == 0 || AKM, "MN loader needs NT % BM == 0");
	static_assert(NT % BN == 0 || BKM, "MN loader needs NT % BN == 0");
	static_assert(NT % BK == 0, "K loader needs NT % BK == 0");
	typedef typename Mfma<T>::acc_t acc_t;

	__shared__ T smem[2 * (A_SZ + B_SZ)];

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int wm = wave % WM, wn = wave / WM;
	const int l15 = lane & 15, lhi = lane >> 4;

	// ---- tile coordinates
	int tm, tn;
	{
		const int nblocks = gridDim.x;
		const int pid = xcd_remap(blockIdx.x, nblocks);
		if (g.tri_enum) {
			// pid enumerates (i, j), j <= i, row by row
			int i = (int) ((sqrtf(8.0f * (float) pid + 1.0f) - 1.0f) * 0.5f);
			while ((i + 1) * (i + 2) / 2 <= pid)
				++i;
			while (i * (i + 1) / 2 > pid)
				--i;
			tm = i;
			tn = pid - i * (i + 1) / 2;
		} else {
			constexpr int G = 8;
			const int per_group = G * g.ntn;
			const int grp = pid / per_group;
			const int first_m = grp * G;
			const int gsz = min(g.ntm - first_m, G);
			const int rem = pid - grp * per_group;
			tm = first_m + rem % gsz;
			tn = rem / gsz;
		}
	}
	const int m_off = tm * BM, n_off = tn * BN;
	if (g.lower && m_off + BM - 1 < lower_col(g, n_off))
		return; // tile entirely above the diagonal

	int k_begin = blockIdx.z * g.k_per_split;
	int k_end = min(g.K, k_begin + g.k_per_split);
	if (EXTRA) {
		// skip the K range where a triangular operand is identically zero for this tile
		const int as = g.a_struct, bs = g.b_struct;
		if (as == 1 || as == 3 || as == 5) // lhs lower: k <= m
			k_end = min(k_end, m_off + BM);
		if (as == 2 || as == 4 || as == 6) // lhs upper: k >= m
			k_begin = max(k_begin, m_off / BK * BK);
		if (bs == 1 || bs == 3 || bs == 5) // rhs lower: k >= n
			k_begin = max(k_begin, n_off / BK * BK);
		if (bs == 2 || bs == 4 || bs == 6) // rhs upper: k <= n
			k_end = min(k_end, n_off + BN);
	}
	const bool k_empty = k_begin >= k_end;
	if (k_empty && g.atomic != 2 && (g.add || g.atomic))
		return; // nothing to accumulate (Replace, and a split-K slice, still have to write zeros)

	// ---- loader state.  MN-major: thread owns one mn index and A_CNT k's (stride KSTEP);
	//      K-major: thread owns one k and A_CNT mn's (stride MSTEP).
	constexpr int A_KSTEP = AKM ? 0 : NT / BM, A_MSTEP = AKM ? NT / BK : 0;
	constexpr int B_KSTEP = BKM ? 0 : NT / BN, B_NSTEP = BKM ? NT / BK : 0;
	const int a_mn = AKM ? tid / BK : tid % BM; // first mn (tile local)
	const int a_k = AKM ? tid % BK : tid / BM;  // first k (tile local)
	const int b_mn = BKM ? tid / BK : tid % BN;
	const int b_k = BKM ? tid % BK : tid / BN;

	T ra[A_CNT], rb[B_CNT];
	unsigned amask = 0, bmask = 0; // validity of ra[] / rb[]; applied at LDS-store time so that the
				       // global loads stay in flight across the MFMA section

	// per-thread base pointers (64-bit math once); per-load offsets are wave-uniform
	const int a_mn0 = m_off + a_mn, b_mn0 = n_off + b_mn;
	const T *a_base = g.a + (AKM ? (idx_t) 0 : (idx_t) min(a_mn0, g.M - 1) * g.ars) + (idx_t) a_k * g.acs;
	const T *b_base = g.b + (BKM ? (idx_t) 0 : (idx_t) min(b_mn0, g.N - 1) * g.bcs) + (idx_t) b_k * g.brs;
	unsigned a_mnmask = 0, b_mnmask = 0;
#pragma unroll
	for (int i = 0; i < A_CNT; ++i)
		a_mnmask |= (unsigned) (a_mn0 + (AKM ? i * A_MSTEP : 0) < g.M) << i;
#pragma unroll
	for (int i = 0; i < B_CNT; ++i)
		b_mnmask |= (unsigned) (b_mn0 + (BKM ? i * B_NSTEP : 0) < g.N) << i;

	auto load_a = [&](int k0, bool kcheck) {
		amask = a_mnmask;
#pragma unroll
		for (int i = 0; i < A_CNT; ++i) {
			const int mn = a_mn0 + (AKM ? i * A_MSTEP : 0);
			const int k = k0 + a_k + (AKM ? 0 : i * A_KSTEP);
			int kc = k;
			if (kcheck) {
				kc = min(k, k_end - 1);
				if (k >= k_end)
					amask &= ~(1u << i);
			}
			const T *p = a_base + (idx_t) (kc - a_k) * g.acs;
			if (AKM)
				p += (idx_t) min(mn, g.M - 1) * g.ars;
			T v = *p;
			if (EXTRA) {
				const int as = g.a_struct;
				if (!in_block(as, mn, k))
					v = ((((as == 5) | (as == 6)) & (mn == k)) != 0) ? (T) 1 : (T) 0;
			}
			ra[i] = v;
		}
	};
	auto load_b = [&](int k0, bool kcheck) {
		bmask = b_mnmask;
#pragma unroll
		for (int i = 0; i < B_CNT; ++i) {
			const int mn = b_mn0 + (BKM ? i * B_NSTEP : 0);
			const int k = k0 + b_k + (BKM ? 0 : i * B_KSTEP);
			int kc = k;
			if (kcheck) {
				kc = min(k, k_end - 1);
				if (k >= k_end)
					bmask &= ~(1u << i);
			}
			const T *p = b_base + (idx_t) (kc - b_k) * g.brs;
			if (BKM)
				p += (idx_t) min(mn, g.N - 1) * g.bcs;
			T v = *p;
			if (EXTRA) {
				const int bs = g.b_struct;
				if (!in_block(bs, k, mn))
					v = ((((bs == 5) | (bs == 6)) & (mn == k)) != 0) ? (T) 1 : (T) 0;
				if (g.diag)
					v *= g.diag[(idx_t) kc * g.diag_stride];
			}
			rb[i] = v;
		}
	};
	auto store_a = [&](T *sa) {
#pragma unroll
		for (int i = 0; i < A_CNT; ++i) {
			const T v = (amask >> i) & 1u ? ra[i] : (T) 0;
			if (AKM)
				sa[(a_mn + i * A_MSTEP) * SA + a_k] = v;
			else
				sa[(a_k + i * A_KSTEP) * SA + a_mn] = v;
		}
	};
	auto store_b = [&](T *sb) {
#pragma unroll
		for (int i = 0; i < B_CNT; ++i) {
			const T v = (bmask >> i) & 1u ? rb[i] : (T) 0;
			if (BKM)
				sb[(b_mn + i * B_NSTEP) * SB + b_k] = v;
			else
				sb[(b_k + i * B_KSTEP) * SB + b_mn] = v;
		}
	};

	acc_t acc[TM][TN];
#pragma unroll
	for (int i = 0; i < TM; ++i)
#pragma unroll
		for (int j = 0; j < TN; ++j)
			acc[i][j] = (acc_t) (T) 0;

	auto compute = [&](const T *sa) {
		const T *sb = sa + A_SZ;
#pragma unroll
		for (int kk = 0; kk < BK / 4; ++kk) {
			T af[TM], bf[TN];
#pragma unroll
			for (int i = 0; i < TM; ++i) {
				const int m = wm * WTM + i * 16 + l15;
				af[i] = AKM ? sa[m * SA + kk * 4 + lhi] : sa[(kk * 4 + lhi) * SA + m];
			}
#pragma unroll
			for (int j = 0; j < TN; ++j) {
				const int n = wn * WTN + j * 16 + l15;
				bf[j] = BKM ? sb[n * SB + kk * 4 + lhi] : sb[(kk * 4 + lhi) * SB + n];
			}
#pragma unroll
			for (int i = 0; i < TM; ++i)
#pragma unroll
				for (int j = 0; j < TN; ++j)
					acc[i][j] = Mfma<T>::run(bf[j], af[i], acc[i][j]);
		}
	};

	// K loop: nk tiles; tiles 0 .. nk-2 are full, the last one may be partial (k-checked loads).
	// Iteration t multiplies tile t out of LDS buffer t&1 while the global loads of tile t+1 are in
	// flight; they land in the other buffer after the MFMA section.  One barrier per tile.
	const int nk = k_empty ? 0 : (k_end - k_begin + BK - 1) / BK;
	constexpr int STAGE = A_SZ + B_SZ;
	if (nk > 0) {
		const bool tail = nk == 1;
		load_a(k_begin, tail);
		load_b(k_begin, tail);
		store_a(smem);
		store_b(smem + A_SZ);
	}
	__syncthreads();
	int kt = 0;
	for (; kt + 2 < nk; ++kt) { // tile kt+1 is full
		const int k0 = k_begin + (kt + 1) * BK;
		load_a(k0, false);
		load_b(k0, false);
		compute(smem + (kt & 1) * STAGE);
		T *nxt = smem + ((kt + 1) & 1) * STAGE;
		store_a(nxt);
		store_b(nxt + A_SZ);
		__syncthreads();
	}
	if (kt + 1 < nk) { // tile kt+1 is the last one
		const int k0 = k_begin + (kt + 1) * BK;
		load_a(k0, true);
		load_b(k0, true);
		compute(smem + (kt & 1) * STAGE);
		T *nxt = smem + ((kt + 1) & 1) * STAGE;
		store_a(nxt);
		store_b(nxt + A_SZ);
		__syncthreads();
		++kt;
	}
	if (nk > 0)
		compute(smem + (kt & 1) * STAGE);

	// ---- epilogue: lane (l15, lhi), reg r of acc[i][j] holds
	//      C[m_off + wm*WTM + i*16 + l15][n_off + wn*WTN + j*16 + row(r, lhi)]
	// (accumulate mode: the old values of a 16-column group are loaded together before the group is stored -- see
	// gemm_kernel_p)
#pragma unroll
	for (int j = 0; j < TN; ++j) {
		if (n_off + wn * WTN + j * 16 >= g.N)
			continue; // the whole 16-column group lies outside dst (wave uniform)
		T *ptr[4][TM];
		T old[4][TM];
		unsigned okmask = 0;
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int n = n_off + wn * WTN + j * 16 + Mfma<T>::row(r, lhi);
			const bool n_ok = n < g.N;
			const int nlow = lower_col(g, n);
			const idx_t ncol = (n_ok && g.col_idx) ? load_idx(g.col_idx, g.idx64, n) : (idx_t) n;
#pragma unroll
			for (int i = 0; i < TM; ++i) {
				const int m = m_off + wm * WTM + i * 16 + l15;
				const bool ok = n_ok && m < g.M && !(g.lower && (m < nlow || (g.dst_strict && m == nlow)));
				const idx_t mrow = (ok && g.row_idx) ? load_idx(g.row_idx, g.idx64, m) : (idx_t) m;
				ptr[r][i] = !ok ? g.dst : g.atomic == 2 ? g.ws + ((size_t) blockIdx.z * g.N + n) * g.M + m : g.dst + mrow * g.drs + ncol * g.dcs;
				okmask |= (unsigned) ok << (r * TM + i);
			}
		}
		if (g.add && !g.atomic) {
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				if (n_off + wn * WTN + j * 16 + Mfma<T>::row(r, 0) >= g.N)
					continue; // wave uniform: no lane has a valid column in this register
#pragma unroll
				for (int i = 0; i < TM; ++i) {
					if (m_off + wm * WTM + i * 16 >= g.M)
						continue; // wave uniform
					old[r][i] = *ptr[r][i]; // unconditional: masked lanes point at dst(0, 0) (a per-lane branch per element
								 // would bring back one memory round trip per element)
				}
			}
		}
#pragma unroll
		for (int r = 0; r < 4; ++r)
#pragma unroll
			for (int i = 0; i < TM; ++i) {
				if (!((okmask >> (r * TM + i)) & 1u))
					continue;
				const T v = acc[i][j][r];
				if (g.atomic == 2)
					*ptr[r][i] = v; // raw slice sums, reduced in a fixed order afterwards
				else if (g.atomic)
					atomicAdd(ptr[r][i], g.alpha * v);
				else if (g.add)
					*ptr[r][i] = fh_fma(g.alpha, v, old[r][i]);
				else
					*ptr[r][i] = g.alpha * v;
			}
	}
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined variant of the dense kernel (no triangular operands / diag): same tiles, same LDS
// images, same epilogue as gemm_kernel above, but the K loop is written as ONE dense MFMA stream per wave:
//   * MFMA fragments are double buffered in registers: the ds_reads of k-step kk+1 are issued before the
//     16 (4) MFMAs of k-step kk, so no MFMA waits on an LDS round trip;
//   * the global loads of tile t+1 are spread over the MFMAs of k-step 0, the LDS stores of tile t+1 over
//     the MFMAs of the last k-step (a v_mfma_f64_16x16x4 holds the pipe for 64 cycles = 16 issue slots, so
//     one memory instruction per MFMA is free);
//   * per-thread running pointers replace the per-load 64-bit stride multiplications;
//   * the only bubble left per K tile is  barrier -> first fragment read of the next tile, which the
//     second workgroup on the CU (2 waves per SIMD) covers.
// sched_group_barrier pins the interleave (cdna_hip_programming.md T19); it is a scheduling hint only.
// ------------------------------------------------------------------------------------------------
#define FH_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

// timing build (make timing): where a workgroup's time goes, s_memtime stamps of thread 0 -- entry -> first tile in LDS ->
// end of the K loop -> epilogue's stores issued -> stores acknowledged; sums over all workgroups (gemm_dump_timing)
#ifdef FH_GEMM_TIMING
__device__ unsigned long long g_gemm_phase[16];
#define FH_GT(i)                                                                                                         \
	do {                                                                                                             \
		const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                             \
		gt_acc[i] += now_ - gt_last;                                                                             \
		gt_last = now_;                                                                                          \
	} while (0)
#else
#define FH_GT(i)                                                                                                         \
	do {                                                                                                             \
	} while (0)
#endif

// Buffer-addressed access to a dst tile: address = descriptor base (SGPRs, per wavefront) + per-lane byte offset (one VGPR,
// computed once) + wave-uniform byte offset (an SGPR per element, scalar unit) -- no vector ALU instruction per element.
typedef unsigned int fh_u32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct TileIO;
template <> struct TileIO<double> {
	static __device__ __forceinline__ double load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
	{
		return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int) voff, (int) soff, 0));
	}
	static __device__ __forceinline__ void store(double v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
	{
		__builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fh_u32x2, v), r, (int) voff, (int) soff, 0);
	}
};
template <> struct TileIO<float> {
	static __device__ __forceinline__ float load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
	{
		return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int) voff, (int) soff, 0));
	}
	static __device__ __forceinline__ void store(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
	{
		__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, (int) voff, (int) soff, 0);
	}
};

// compile-time unrolled interleave pattern for one k-step of NMMA MFMAs (the builtin needs constants):
// KIND 0: MFMA + fragment ds_read + global loads; KIND 1: MFMA + fragment ds_read; KIND 2: MFMA + LDS stores
template <int I, int NMMA, int NFR, int NLD, int KIND> struct SgbStep {
	static __device__ __forceinline__ void run()
	{
		if constexpr (I < NMMA) {
			FH_SGB(0x008, 1);
			if constexpr (KIND != 2 && I < NFR)
				FH_SGB(0x100, 1);
			constexpr int CNT = (I + 1) * NLD / NMMA - I * NLD / NMMA;
			if constexpr (KIND == 0 && CNT > 0)
				FH_SGB(0x020, CNT);
			if constexpr (KIND == 2 && CNT > 0)
				FH_SGB(0x200, CNT);
			SgbStep<I + 1, NMMA, NFR, NLD, KIND>::run();
		}
	}
};

template <typename T, int BM, int BN, int BK, int WM, int WN, bool AKM, bool BKM, int PF>
__global__ __launch_bounds__(WM *WN * 64, (WM * WN == 4 ? 2 : 1)) void gemm_kernel_p(const GemmArgs<T> g)
{
	constexpr int NT = WM * WN * 64;
	constexpr int WTM = BM / WM, WTN = BN / WN; // wave tile
	constexpr int TM = WTM / 16, TN = WTN / 16;
	constexpr int SA = AKM ? BK + 2 : BM + 16; // LDS pitch (elements)
	constexpr int SB = BKM ? BK + 2 : BN + 16;
	constexpr int A_SZ = AKM ? BM * SA : BK * SA;
	constexpr int B_SZ = BKM ? BN * SB : BK * SB;
	constexpr int A_CNT = BM * BK / NT, B_CNT = BN * BK / NT; // elements per thread per tile
	static_assert(BM * BK % NT == 0 && BN * BK % NT == 0, "tile/threads");
	static_assert(NT % BM == 0 || AKM, "MN loader needs NT % BM == 0");
	static_assert(NT % BN == 0 || BKM, "MN loader needs NT % BN == 0");
	static_assert(NT % BK == 0 && BK == 16, "K loader needs NT % BK == 0");
	typedef typename Mfma<T>::acc_t acc_t;

	__shared__ T smem[2 * (A_SZ + B_SZ)];

	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int wm = wave % WM, wn = wave / WM;
	const int l15 = lane & 15, lhi = lane >> 4;
#ifdef FH_GEMM_TIMING
	unsigned long long gt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long gt_last = __builtin_amdgcn_s_memtime();
#endif

	// ---- tile coordinates (same mapping as gemm_kernel)
	int tm, tn;
	int pid_t = 0;
	if (g.ticket) {
		__shared__ int s_pid;
		if (tid == 0) {
			unsigned xcc;
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
			const int T_ = g.ticket_total, q = T_ >> 3, r = T_ & 7;
			const int x0 = (int) (xcc & 7u);
			int got = -1;
			if (g.ticket_role) {
				const int cnt = q + (x0 < r ? 1 : 0), start = x0 < r ? x0 * (q + 1) : r * (q + 1) + (x0 - r) * q;
				const int cur = __hip_atomic_load(&g.ticket[x0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (cnt - cur > g.ticket_margin) {
					const int t = atomicAdd(&g.ticket[x0], 1);
					if (t < cnt)
						got = start + t;
				}
			} else {
				for (int i = 0; i < 8 && got < 0; ++i) {
					const int x = (x0 + i) & 7;
					const int cnt = q + (x < r ? 1 : 0), start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
					if (cnt <= 0)
						continue;
					const int t = atomicAdd(&g.ticket[x], 1);
					if (t < cnt)
						got = start + t;
				}
			}
			s_pid = got;
		}
		__syncthreads();
		pid_t = s_pid;
		if (pid_t < 0)
			return;
	}
	{
		const int nblocks = gridDim.x;
		const int pid = (g.ticket ? pid_t : xcd_remap(blockIdx.x, nblocks)) + g.tri_off;
		if (g.tri_enum && BN == 2 * BM) {
			// lower trapezoid of BM x 2 BM tiles: tile row tm holds tn = 0 .. tm / 2, i.e. rows 2 p and 2 p + 1 hold p + 1
			// tiles each and p (p + 1) tiles precede row 2 p
			int q = (int) ((sqrtf(4.0f * (float) pid + 1.0f) - 1.0f) * 0.5f);
			while ((q + 1) * (q + 2) <= pid)
				++q;
			while (q * (q + 1) > pid)
				--q;
			const int rem = pid - q * (q + 1);
			tm = rem < q + 1 ? 2 * q : 2 * q + 1;
			tn = rem < q + 1 ? rem : rem - (q + 1);
		} else if (g.tri_enum) {
			int i = (int) ((sqrtf(8.0f * (float) pid + 1.0f) - 1.0f) * 0.5f);
			while ((i + 1) * (i + 2) / 2 <= pid)
				++i;
			while (i * (i + 1) / 2 > pid)
				--i;
			tm = i;
			tn = pid - i * (i + 1) / 2;
		} else {
			const int G = g.raster_g;
			const int per_group = G * g.ntn;
			const int grp = pid / per_group;
			const int first_m = grp * G;
			const int gsz = min(g.ntm - first_m, G);
			const int rem = pid - grp * per_group;
			tm = first_m + rem % gsz;
			tn = rem / gsz;
			// trimmed K loops: the tiles with the long loops go first (longest-processing-time order)
			if (g.k_trim == 1)
				tn = g.ntn - 1 - tn;
		}
	}
	const int m_off = tm * BM, n_off = tn * BN;
	if (g.lower && m_off + BM - 1 < lower_col(g, n_off))
		return; // tile entirely above the diagonal

	const int k_begin = blockIdx.z * g.k_per_split;
	int k_end = min(g.K, k_begin + g.k_per_split);
	if (g.k_trim == 1)
		k_end = min(k_end, n_off + BN);
	else if (g.k_trim == 2)
		k_end = min(k_end, m_off + BM);
	const bool k_empty = k_begin >= k_end;
	if (k_empty && g.atomic != 2 && (g.add || g.atomic))
		return;

	// ---- loader state (see gemm_kernel).  MN-major: one mn, A_CNT k's; K-major: one k, A_CNT mn's.
	constexpr int A_KSTEP = AKM ? 0 : NT / BM, A_MSTEP = AKM ? NT / BK : 0;
	constexpr int B_KSTEP = BKM ? 0 : NT / BN, B_NSTEP = BKM ? NT / BK : 0;
	const int a_mn = AKM ? tid / BK : tid % BM;
	const int a_k = AKM ? tid % BK : tid / BM;
	const int b_mn = BKM ? tid / BK : tid % BN;
	const int b_k = BKM ? tid % BK : tid / BN;
	const int a_mn0 = m_off + a_mn, b_mn0 = n_off + b_mn;

	// PF register sets: tile t + PF is in flight from HBM while tile t is multiplied (PF = 1 for the
	// 128 x 128 tile whose accumulators need the registers; deeper for the small tiles, whose launches are
	// short and latency bound: one HBM round trip per K tile would otherwise dominate them)
	T ra[PF][A_CNT], rb[PF][B_CNT];
	unsigned amask[PF], bmask[PF];
	unsigned a_mnmask = 0, b_mnmask = 0;
#pragma unroll
	for (int i = 0; i < A_CNT; ++i)
		a_mnmask |= (unsigned) (a_mn0 + (AKM ? i * A_MSTEP : 0) < g.M) << i;
#pragma unroll
	for (int i = 0; i < B_CNT; ++i)
		b_mnmask |= (unsigned) (b_mn0 + (BKM ? i * B_NSTEP : 0) < g.N) << i;

	// running pointers: element (first mn, k_tile + first k) of the current tile
	const T *pa = g.a + (AKM ? (idx_t) 0 : (idx_t) min(a_mn0, g.M - 1) * g.ars) + (idx_t) (k_begin + a_k) * g.acs;
	const T *pb = g.b + (BKM ? (idx_t) 0 : (idx_t) min(b_mn0, g.N - 1) * g.bcs) + (idx_t) (k_begin + b_k) * g.brs;
	const idx_t a_tile_step = (idx_t) BK * g.acs, b_tile_step = (idx_t) BK * g.brs;
	const idx_t a_kstep = (idx_t) A_KSTEP * g.acs, b_kstep = (idx_t) B_KSTEP * g.brs; // wave uniform

	// Full tiles (no k checks) are loaded through buffer descriptors: base = a wave-uniform pointer that advances with the K
	// tile (scalar unit), one per-lane byte offset computed here, a scalar offset per element -- no vector ALU instruction per
	// load inside the K loop (the 64-bit pointer arithmetic of the pointer form was ~40 of the loop's ~70 per 16 columns of K).
	// MN-major operands keep the row clamp in the lane offset; the K-major B operand needs no clamp at all: the descriptor ends
	// where column N begins, columns beyond it read as zero.  (K-major A keeps the pointer form.)  The host routes operands
	// with negative strides or offsets beyond 32 bits to the non-pipelined kernel (gemm_dev).
	constexpr unsigned TS = (unsigned) sizeof(T);
	const T *a_ub = g.a + (idx_t) k_begin * g.acs; // (uniform)
	const T *b_ub = g.b + (idx_t) k_begin * g.brs + (BKM ? (idx_t) n_off * g.bcs : (idx_t) 0);
	const unsigned a_vo = AKM ? 0u : (unsigned) ((idx_t) min(a_mn0, g.M - 1) * g.ars + (idx_t) a_k * g.acs) * TS;
	const unsigned b_vo = BKM ? (unsigned) ((idx_t) b_k * g.brs + (idx_t) b_mn * g.bcs) * TS
				  : (unsigned) ((idx_t) min(b_mn0, g.N - 1) * g.bcs + (idx_t) b_k * g.brs) * TS;
	const idx_t b_ext = (idx_t) (g.N - n_off) * g.bcs * (idx_t) TS;
	const unsigned b_lim = BKM ? (b_ext < (idx_t) 0xfffffff0u ? (unsigned) b_ext : 0xfffffff0u) : 0x7fffffffu;
	// The hardware's range check covers the per-lane offset only, NOT the scalar offset: with the column step of
	// element i in the scalar offset, an edge tile would read up to BN - 1 columns past the end of B (masked afterwards, but
	// outside the operand).  The column steps of a K-major B therefore travel in per-lane offsets prepared here, once per tile
	// (B_CNT registers, no vector ALU inside the K loop either): every element of every lane is checked against the descriptor,
	// which ends where column N begins.
	unsigned b_vo_i[BKM ? B_CNT : 1];
#pragma unroll
	for (int i = 0; i < (BKM ? B_CNT : 1); ++i)
		b_vo_i[i] = b_vo + (unsigned) (i * B_NSTEP * (int) g.bcs) * TS;
	// (the eight-wavefront 128 x 256 tile loads only B this way: descriptors for both 71.4 -> 70.0 TFLOP/s at N = 8192, for A alone
	// 69.8, for B alone 72.0; the four-wavefront tiles gain 4 % with both: profiles/r04_gemm_tile_phases.txt)
	constexpr bool BUF_A = !AKM && WM * WN == 4, BUF_B = true;
	auto load_a = [&](T (&ra_)[A_CNT], unsigned &amask_) {
		amask_ = a_mnmask;
		if constexpr (!BUF_A) {
#pragma unroll
			for (int i = 0; i < A_CNT; ++i)
				ra_[i] = *(AKM ? pa + (idx_t) min(a_mn0 + i * A_MSTEP, g.M - 1) * g.ars : pa + (idx_t) i * a_kstep);
		} else {
			const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) a_ub, 0, (int) 0x7fffffff, 0x00020000);
#pragma unroll
			for (int i = 0; i < A_CNT; ++i)
				ra_[i] = TileIO<T>::load(rs, a_vo, (unsigned) (i * A_KSTEP * (int) g.acs) * TS);
		}
	};
	auto load_b = [&](T (&rb_)[B_CNT], unsigned &bmask_) {
		bmask_ = b_mnmask;
		if constexpr (!BUF_B) {
#pragma unroll
			for (int i = 0; i < B_CNT; ++i)
				rb_[i] = *(BKM ? pb + (idx_t) min(b_mn0 + i * B_NSTEP, g.N - 1) * g.bcs : pb + (idx_t) i * b_kstep);
		} else {
			const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) b_ub, 0, (int) b_lim, 0x00020000);
#pragma unroll
			for (int i = 0; i < B_CNT; ++i)
				rb_[i] = BKM ? TileIO<T>::load(rs, b_vo_i[BKM ? i : 0], 0u) : TileIO<T>::load(rs, b_vo, (unsigned) (i * B_KSTEP * (int) g.brs) * TS);
		}
	};
	// last (possibly partial) tile starting at k0: k >= k_end is clamped to k_end - 1 and masked
	auto load_a_tail = [&](T (&ra_)[A_CNT], unsigned &amask_, int k0) {
		amask_ = a_mnmask;
#pragma unroll
		for (int i = 0; i < A_CNT; ++i) {
			const int k = k0 + a_k + (AKM ? 0 : i * A_KSTEP);
			const int back = max(k - (k_end - 1), 0);
			if (back > 0)
				amask_ &= ~(1u << i);
			const T *p = AKM ? pa + (idx_t) min(a_mn0 + i * A_MSTEP, g.M - 1) * g.ars : pa + (idx_t) i * a_kstep;
			ra_[i] = *(p - (idx_t) back * g.acs);
		}
	};
	auto load_b_tail = [&](T (&rb_)[B_CNT], unsigned &bmask_, int k0) {
		bmask_ = b_mnmask;
#pragma unroll
		for (int i = 0; i < B_CNT; ++i) {
			const int k = k0 + b_k + (BKM ? 0 : i * B_KSTEP);
			const int back = max(k - (k_end - 1), 0);
			if (back > 0)
				bmask_ &= ~(1u << i);
			const T *p = BKM ? pb + (idx_t) min(b_mn0 + i * B_NSTEP, g.N - 1) * g.bcs : pb + (idx_t) i * b_kstep;
			rb_[i] = *(p - (idx_t) back * g.brs);
		}
	};
	auto store_a = [&](T *sa, const T (&ra_)[A_CNT], unsigned amask_) {
#pragma unroll
		for (int i = 0; i < A_CNT; ++i) {
			const T v = (amask_ >> i) & 1u ? ra_[i] : (T) 0;
			if (AKM)
				sa[(a_mn + i * A_MSTEP) * SA + a_k] = v;
			else
				sa[(a_k + i * A_KSTEP) * SA + a_mn] = v;
		}
	};
	auto store_b = [&](T *sb, const T (&rb_)[B_CNT], unsigned bmask_) {
#pragma unroll
		for (int i = 0; i < B_CNT; ++i) {
			const T v = (bmask_ >> i) & 1u ? rb_[i] : (T) 0;
			if (BKM)
				sb[(b_mn + i * B_NSTEP) * SB + b_k] = v;
			else
				sb[(b_k + i * B_KSTEP) * SB + b_mn] = v;
		}
	};

	acc_t acc[TM][TN];
#pragma unroll
	for (int i = 0; i < TM; ++i)
#pragma unroll
		for (int j = 0; j < TN; ++j)
			acc[i][j] = (acc_t) (T) 0;

	// ---- fast tile I/O.  The two workgroups of a CU share its SIMDs, and while one of them streams MFMAs every vector ALU
	// instruction of the OTHER one waits for a gap in that stream (~64 cycles each, profiles/r04_gemm_tile_phases.txt): the
	// general epilogue below -- 64-bit address arithmetic, masks and an fma per element, ~1500 vector ALU instructions per
	// wavefront -- took 95 000 (accumulate) to 143 000 (replace) cycles of a K = 1024 tile's 640 000.  Interior tiles of a plain
	// column-major dst (everything but the edge / diagonal tiles of the products the factorizations issue) address dst through
	// a buffer descriptor instead (TileIO: no vector ALU per element), and an accumulating product with alpha = +-1 loads the
	// old tile INTO the accumulators here, so that its epilogue is 64 stores (alpha = -1: from -dst, stored negated -- exact).
	const bool fast = g.fast_io != 0 && m_off + BM <= g.M && n_off + BN <= g.N && (!g.lower || m_off >= n_off + BN);
	__amdgpu_buffer_rsrc_t crs;
	unsigned cvoff = 0;
	if (fast) {
		const int wv = __builtin_amdgcn_readfirstlane(wave);
		T *cbase = g.dst + (idx_t) (m_off + (wv % WM) * WTM) + (idx_t) (n_off + (wv / WM) * WTN) * g.dcs;
		crs = __builtin_amdgcn_make_buffer_rsrc((void *) cbase, 0, (int) 0x7fffffff, 0x00020000);
		cvoff = (unsigned) (l15 + Mfma<T>::row(0, lhi) * (int) g.dcs) * (unsigned) sizeof(T);
	}
	// (issued BEHIND the first A / B tile's loads, negated behind that tile's LDS stores: one memory round trip for both)
	const bool cstart = fast && g.fast_io >= 2;
	auto load_c_tile = [&]() {
#pragma unroll
		for (int j = 0; j < TN; ++j)
#pragma unroll
			for (int r = 0; r < 4; ++r)
#pragma unroll
				for (int i = 0; i < TM; ++i) {
					const unsigned soff = (unsigned) ((j * 16 + Mfma<T>::row(r, 0)) * (int) g.dcs + i * 16) * (unsigned) sizeof(T);
					acc[i][j][r] = TileIO<T>::load(crs, cvoff, soff);
				}
	};
	auto negate_c_tile = [&]() {
#pragma unroll
		for (int j = 0; j < TN; ++j)
#pragma unroll
			for (int r = 0; r < 4; ++r)
#pragma unroll
				for (int i = 0; i < TM; ++i)
					acc[i][j][r] = -acc[i][j][r];
	};

	// fragment double buffer
	T fa[2][TM], fb[2][TN];
	const int a_frag_off = AKM ? (wm * WTM + l15) * SA + lhi : lhi * SA + wm * WTM + l15;
	const int b_frag_off = BKM ? (wn * WTN + l15) * SB + lhi : lhi * SB + wn * WTN + l15;
	auto read_frag = [&](const T *st, int kk, int buf) {
		const T *sa = st + a_frag_off + (AKM ? kk * 4 : kk * 4 * SA);
		const T *sb = st + A_SZ + b_frag_off + (BKM ? kk * 4 : kk * 4 * SB);
#pragma unroll
		for (int i = 0; i < TM; ++i)
			fa[buf][i] = sa[AKM ? i * 16 * SA : i * 16];
#pragma unroll
		for (int j = 0; j < TN; ++j)
			fb[buf][j] = sb[BKM ? j * 16 * SB : j * 16];
	};
	auto mma = [&](int buf) {
#pragma unroll
		for (int i = 0; i < TM; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
				acc[i][j] = Mfma<T>::run(fb[buf][j], fa[buf][i], acc[i][j]);
	};
	constexpr int NMMA = TM * TN;	  // MFMAs per k-step
	constexpr int NFR = TM + TN;	  // fragment ds_reads per k-step
	constexpr int NLD = A_CNT + B_CNT; // global loads / LDS stores per tile

	const int nk = k_empty ? 0 : (k_end - k_begin + BK - 1) / BK;
	constexpr int STAGE = A_SZ + B_SZ;
	if constexpr (PF == 1) {
		if (nk > 0) {
			if (nk == 1) {
				load_a_tail(ra[0], amask[0], k_begin);
				load_b_tail(rb[0], bmask[0], k_begin);
			} else {
				load_a(ra[0], amask[0]);
				load_b(rb[0], bmask[0]);
			}
			if (cstart)
				load_c_tile();
			store_a(smem, ra[0], amask[0]);
			store_b(smem + A_SZ, rb[0], bmask[0]);
			if (cstart && g.fast_io == 3)
				negate_c_tile();
		}
		__syncthreads();
		FH_GT(0);
		if (nk > 0)
			read_frag(smem, 0, 0);
		int kt = 0;
		for (; kt + 2 < nk; ++kt) { // tile kt+1 is full
			const T *cur = smem + (kt & 1) * STAGE;
			T *nxt = smem + ((kt + 1) & 1) * STAGE;
			pa += a_tile_step;
			pb += b_tile_step;
			a_ub += a_tile_step;
			b_ub += b_tile_step;
			load_a(ra[0], amask[0]);
			load_b(rb[0], bmask[0]);
			read_frag(cur, 1, 1);
			mma(0);
			// k-step 0: one global load per MFMA, the fragment reads of k-step 1 up front
			SgbStep<0, NMMA, NFR, NLD, 0>::run();
			read_frag(cur, 2, 0);
			mma(1);
			SgbStep<0, NMMA, NFR, NLD, 1>::run();
			read_frag(cur, 3, 1);
			mma(0);
			SgbStep<0, NMMA, NFR, NLD, 1>::run();
			store_a(nxt, ra[0], amask[0]);
			store_b(nxt + A_SZ, rb[0], bmask[0]);
			mma(1);
			// last k-step: the LDS stores of the next tile ride behind the MFMAs
			SgbStep<0, NMMA, NFR, NLD, 2>::run();
			__syncthreads();
			read_frag(nxt, 0, 0);
		}
		if (kt + 1 < nk) { // tile kt+1 is the last one (k-checked loads)
			const T *cur = smem + (kt & 1) * STAGE;
			T *nxt = smem + ((kt + 1) & 1) * STAGE;
			pa += a_tile_step;
			pb += b_tile_step;
			a_ub += a_tile_step;
			b_ub += b_tile_step;
			load_a_tail(ra[0], amask[0], k_begin + (kt + 1) * BK);
			load_b_tail(rb[0], bmask[0], k_begin + (kt + 1) * BK);
			read_frag(cur, 1, 1);
			mma(0);
			read_frag(cur, 2, 0);
			mma(1);
			read_frag(cur, 3, 1);
			mma(0);
			store_a(nxt, ra[0], amask[0]);
			store_b(nxt + A_SZ, rb[0], bmask[0]);
			mma(1);
			__syncthreads();
			read_frag(nxt, 0, 0);
			++kt;
		}
		if (nk > 0) {
			const T *cur = smem + (kt & 1) * STAGE;
			read_frag(cur, 1, 1);
			mma(0);
			read_frag(cur, 2, 0);
			mma(1);
			read_frag(cur, 3, 1);
			mma(0);
			mma(1);
		}
	} else {
		// ring of PF register sets; pa / pb always point at the next tile to load (tile index `tl`)
		if (cstart) {
			load_c_tile();
			if (g.fast_io == 3)
				negate_c_tile();
		}
		int tl = 0;
		auto load_next = [&](T (&ra_)[A_CNT], unsigned &amask_, T (&rb_)[B_CNT], unsigned &bmask_) {
			if (tl == nk - 1) { // last tile: k-checked
				load_a_tail(ra_, amask_, k_begin + tl * BK);
				load_b_tail(rb_, bmask_, k_begin + tl * BK);
			} else {
				load_a(ra_, amask_);
				load_b(rb_, bmask_);
			}
			pa += a_tile_step;
			pb += b_tile_step;
			a_ub += a_tile_step;
			b_ub += b_tile_step;
			++tl;
		};
#pragma unroll
		for (int d = 0; d < PF; ++d)
			if (d < nk)
				load_next(ra[d], amask[d], rb[d], bmask[d]);
		if (nk > 0) {
			store_a(smem, ra[0], amask[0]);
			store_b(smem + A_SZ, rb[0], bmask[0]);
		}
		__syncthreads();
		if (nk > 0)
			read_frag(smem, 0, 0);
		for (int kt = 0; kt < nk; kt += PF) {
#pragma unroll
			for (int d = 0; d < PF; ++d) {
				const int t = kt + d;
				if (t < nk) { // block uniform
					const T *cur = smem + (t & 1) * STAGE;
					T *nxt = smem + ((t + 1) & 1) * STAGE;
					// slot d (tile t) already sits in LDS: refill it with tile t + PF
					if (t + PF < nk)
						load_next(ra[d], amask[d], rb[d], bmask[d]);
					read_frag(cur, 1, 1);
					mma(0);
					read_frag(cur, 2, 0);
					mma(1);
					read_frag(cur, 3, 1);
					mma(0);
					if (t + 1 < nk) {
						const int dn = (d + 1) % PF; // constant after unrolling
						store_a(nxt, ra[dn], amask[dn]);
						store_b(nxt + A_SZ, rb[dn], bmask[dn]);
					}
					mma(1);
					if (t + 1 < nk) {
						__syncthreads();
						read_frag(nxt, 0, 0);
					}
				}
			}
		}
	}

	FH_GT(1);
	if (fast) {
		// (scaled IN PLACE first, then stored straight from the accumulators: a temporary per element makes the compiler wait
		// for older stores before it reuses the temporary's registers)
		const T scale = g.fast_io == 1 ? g.alpha : (g.fast_io == 3 ? (T) -1 : (T) 1);
		if (scale != (T) 1) { // (uniform)
#pragma unroll
			for (int j = 0; j < TN; ++j)
#pragma unroll
				for (int r = 0; r < 4; ++r)
#pragma unroll
					for (int i = 0; i < TM; ++i)
						acc[i][j][r] *= scale;
		}
#pragma unroll
		for (int j = 0; j < TN; ++j)
#pragma unroll
			for (int r = 0; r < 4; ++r)
#pragma unroll
				for (int i = 0; i < TM; ++i) {
					const unsigned soff = (unsigned) ((j * 16 + Mfma<T>::row(r, 0)) * (int) g.dcs + i * 16) * (unsigned) sizeof(T);
					TileIO<T>::store(acc[i][j][r], crs, cvoff, soff);
				}
#ifdef FH_GEMM_TIMING
		FH_GT(2);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		FH_GT(3);
		if (tid == 0 && PF == 1) {
			for (int i = 0; i < 8; ++i)
				atomicAdd(&g_gemm_phase[i], gt_acc[i]);
			atomicAdd(&g_gemm_phase[8], 1ull);
		}
#endif
		return;
	}
	// ---- epilogue: lane (l15, lhi), reg r of acc[i][j] holds C(m_off + wm*WTM + i*16 + l15, n_off + wn*WTN + j*16 +
	// row(r, lhi)).  Accumulate mode first LOADS the 4 * TM old values of one 16-column group together and only then
	// stores them (measured: prefetching the next group ahead of the stores is slower again): written as one
	// read-modify-write per element the compiler has to keep every load behind the previous
	// store (same base pointer, run-time strides), i.e. 16 * TN dependent memory round trips per wave -- ~30 us per
	// 128 x 128 tile, 10 % of a K = 1024 update (profiles/r01_exp_syrk_rates.txt).
#pragma unroll
	for (int j = 0; j < TN; ++j) {
		if (n_off + wn * WTN + j * 16 >= g.N)
			continue; // the whole 16-column group lies outside dst (wave uniform)
		T *ptr[4][TM];
		T old[4][TM];
		unsigned okmask = 0;
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int n = n_off + wn * WTN + j * 16 + Mfma<T>::row(r, lhi);
			const bool n_ok = n < g.N;
			const int nlow = lower_col(g, n);
			const idx_t ncol = (n_ok && g.col_idx) ? load_idx(g.col_idx, g.idx64, n) : (idx_t) n;
#pragma unroll
			for (int i = 0; i < TM; ++i) {
				const int m = m_off + wm * WTM + i * 16 + l15;
				const bool ok = n_ok && m < g.M && !(g.lower && (m < nlow || (g.dst_strict && m == nlow)));
				const idx_t mrow = (ok && g.row_idx) ? load_idx(g.row_idx, g.idx64, m) : (idx_t) m;
				ptr[r][i] = !ok ? g.dst : g.atomic == 2 ? g.ws + ((size_t) blockIdx.z * g.N + n) * g.M + m : g.dst + mrow * g.drs + ncol * g.dcs;
				okmask |= (unsigned) ok << (r * TM + i);
			}
		}
		if (g.add && !g.atomic && g.epi_serial) {
#pragma unroll
			for (int r = 0; r < 4; ++r)
#pragma unroll
				for (int i = 0; i < TM; ++i)
					if ((okmask >> (r * TM + i)) & 1u)
						*ptr[r][i] = fh_fma(g.alpha, acc[i][j][r], *ptr[r][i]);
			continue;
		}
		if (g.add && !g.atomic) {
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				if (n_off + wn * WTN + j * 16 + Mfma<T>::row(r, 0) >= g.N)
					continue; // wave uniform: no lane has a valid column in this register
#pragma unroll
				for (int i = 0; i < TM; ++i) {
					if (m_off + wm * WTM + i * 16 >= g.M)
						continue; // wave uniform
					old[r][i] = *ptr[r][i]; // unconditional: masked lanes point at dst(0, 0) (a per-lane branch per element
								 // would bring back one memory round trip per element)
				}
			}
		}
#pragma unroll
		for (int r = 0; r < 4; ++r)
#pragma unroll
			for (int i = 0; i < TM; ++i) {
				if (!((okmask >> (r * TM + i)) & 1u))
					continue;
				const T v = acc[i][j][r];
				if (g.atomic == 2)
					*ptr[r][i] = v; // raw slice sums, reduced in a fixed order afterwards
				else if (g.atomic)
					atomicAdd(ptr[r][i], g.alpha * v);
				else if (g.add)
					*ptr[r][i] = fh_fma(g.alpha, v, old[r][i]);
				else
					*ptr[r][i] = g.alpha * v;
			}
#ifdef FH_GEMM_TIMING
		if (j < 4)
			FH_GT(4 + j);
#endif
	}
#ifdef FH_GEMM_TIMING
	FH_GT(2);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	FH_GT(3);
	if (tid == 0 && PF == 1) {
		for (int i = 0; i < 8; ++i)
			atomicAdd(&g_gemm_phase[i], gt_acc[i]);
		atomicAdd(&g_gemm_phase[8], 1ull);
	}
#endif
}

// ------------------------------------------------------------------------------------------------
// zero / constant fill restricted to a DstKind (K == 0 with Replace, split-K pre-zero)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void fill_kernel(T *p, idx_t rs, idx_t cs, idx_t M, idx_t N, int kind, T value, const void *row_idx,
			    const void *col_idx, int idx64)
{
	const idx_t total = M * N;
	for (idx_t e = (idx_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (idx_t) gridDim.x * blockDim.x) {
		const idx_t i = e % M, j = e / M;
		const int kd = kind & 3;
		if ((kd == DST_LOWER && i < j) || (kd == DST_UPPER && i > j) || ((kind & 4) && i == j))
			continue;
		const idx_t r = row_idx ? load_idx(row_idx, idx64, (int) i) : i;
		const idx_t c = col_idx ? load_idx(col_idx, idx64, (int) j) : j;
		p[r * rs + c * cs] = value;
	}
}

static inline idx_t iabs(idx_t x) { return x < 0 ? -x : x; }

// second pass of the deterministic split-K: dst(kind) <- [dst +] alpha * sum_z ws[z].  The slice range is cut into
// 16 fixed segments summed by 16 threads per element (independent loads in flight), the segment sums are
// combined in segment order: the result depends on the split count only, never on scheduling.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(T *dst, idx_t drs, idx_t dcs, int M, int N, const T *__restrict__ ws,
							    int splits, T alpha, int add, int lower, int strict)
{
	__shared__ T part[16][17];
	const idx_t total = (idx_t) M * N;
	const int le = threadIdx.x & 15, seg = threadIdx.x >> 4;
	const idx_t e = (idx_t) blockIdx.x * 16 + le;
	const int zs = (splits + 15) / 16;
	const int z0 = seg * zs, z1 = min(splits, z0 + zs);
	T acc[4] = {0, 0, 0, 0};
	if (e < total) {
		int z = z0;
		for (; z + 4 <= z1; z += 4) {
#pragma unroll
			for (int u = 0; u < 4; ++u)
				acc[u] += ws[(size_t) (z + u) * total + e];
		}
		for (; z < z1; ++z)
			acc[0] += ws[(size_t) z * total + e];
	}
	part[seg][le] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
	__syncthreads();
	if (seg == 0 && e < total) {
		const int m = (int) (e % M), n = (int) (e / M);
		if (lower && (m < n || (strict && m == n)))
			return;
		T sum = (T) 0;
#pragma unroll
		for (int k = 0; k < 16; ++k)
			sum += part[k][le];
		T *p = dst + (idx_t) m * drs + (idx_t) n * dcs;
		*p = add ? fh_fma(alpha, sum, *p) : alpha * sum;
	}
}

template <typename T> void splitk_reduce_dev(MatV<T> C, const T *ws, int splits, T alpha, bool add)
{
	const idx_t total = C.nrows * C.ncols;
	if (total == 0)
		return;
	hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned) ((total + 15) / 16)), dim3(256), 0, ctx().stream, C.p, C.rs, C.cs, (int) C.nrows,
			   (int) C.ncols, ws, splits, alpha, add ? 1 : 0, 0, 0);
	FH_HIP(hipGetLastError());
}
template void splitk_reduce_dev<double>(MatV<double>, const double *, int, double, bool);
template void splitk_reduce_dev<float>(MatV<float>, const float *, int, float, bool);

void gemm_dump_timing()
{
#ifdef FH_GEMM_TIMING
	unsigned long long h[16];
	FH_HIP(hipDeviceSynchronize());
	FH_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_phase), sizeof(h)));
	unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	FH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_phase), z, sizeof(z)));
	const double n = h[8] ? (double) h[8] : 1.0;
	fprintf(stderr, "gemm tile phases (shader cycles per workgroup, wave 0; %llu tiles): first tile in LDS %.0f | K loop %.0f | epilogue groups %.0f %.0f %.0f %.0f (+%.0f) | stores acknowledged %.0f\n",
		h[8], h[0] / n, h[1] / n, h[4] / n, h[5] / n, h[6] / n, h[7] / n, h[2] / n, h[3] / n);
#endif
}

template <typename T> static void fill_ext(MatV<T> A, DstKind kind, T value, const GemmExtra<T> *ex)
{
	if (A.nrows == 0 || A.ncols == 0)
		return;
	// put the smaller stride on the fast index (only when no index arrays are involved)
	const bool swap = !ex && iabs(A.cs) < iabs(A.rs);
	MatV<T> V = swap ? A.t() : A;
	DstKind k = kind;
	if (swap && kind != DST_FULL)
		k = kind == DST_LOWER ? DST_UPPER : DST_LOWER;
	const idx_t total = V.nrows * V.ncols;
	idx_t blocks = (total + 255) / 256;
	if (blocks > 65536)
		blocks = 65536;
	hipLaunchKernelGGL(fill_kernel<T>, dim3((unsigned) blocks), dim3(256), 0, ctx().stream, V.p, V.rs, V.cs, V.nrows,
			   V.ncols, (int) k | (ex && ex->dst_strict && kind != DST_FULL ? 4 : 0), value, ex ? (swap ? ex->col_idx : ex->row_idx) : nullptr,
			   ex ? (swap ? ex->row_idx : ex->col_idx) : nullptr, ex ? ex->idx64 : 1);
	FH_HIP(hipGetLastError());
}

template <typename T> void fill_dev(MatV<T> A, DstKind kind, T value) { fill_ext<T>(A, kind, value, nullptr); }

// ------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, bool EXTRA>
static void launch_cfg(const GemmArgs<T> &g, bool akm, bool bkm, int splits)
{
	constexpr int BK = 16;
	constexpr int NT = WM * WN * 64;
	int nblocks = g.tri_enum ? g.ntm * (g.ntm + 1) / 2 : g.ntm * g.ntn;
	dim3 grid((unsigned) nblocks, 1, (unsigned) splits), block(NT);
	hipStream_t s = ctx().stream;
	if (akm && bkm)
		hipLaunchKernelGGL((gemm_kernel<T, BM, BN, BK, WM, WN, true, true, EXTRA>), grid, block, 0, s, g);
	else if (akm && !bkm)
		hipLaunchKernelGGL((gemm_kernel<T, BM, BN, BK, WM, WN, true, false, EXTRA>), grid, block, 0, s, g);
	else if (!akm && bkm)
		hipLaunchKernelGGL((gemm_kernel<T, BM, BN, BK, WM, WN, false, true, EXTRA>), grid, block, 0, s, g);
	else
		hipLaunchKernelGGL((gemm_kernel<T, BM, BN, BK, WM, WN, false, false, EXTRA>), grid, block, 0, s, g);
	FH_HIP(hipGetLastError());
}

// tiles of the lower triangle in the first `rows` tile rows: square tiles, or BM x 2 BM tiles (tile row tm holds tm / 2 + 1)
static inline int tri_tiles(int rows, bool wide)
{
	if (!wide)
		return rows * (rows + 1) / 2;
	const int p = rows / 2;
	return (rows & 1) ? (p + 1) * (p + 1) : p * (p + 1);
}

// software-pipelined dense kernel (gemm_kernel_p)
template <typename T, int BM, int BN, int WM, int WN> static void launch_cfg_p(const GemmArgs<T> &g, bool akm, bool bkm, int splits)
{
	constexpr int BK = 16;
	constexpr int PF = (BM * BN >= 128 * 128) ? 1 : 4; // register prefetch depth (tiles)
	constexpr int NT = WM * WN * 64;
	int nblocks = g.tri_enum ? tri_tiles(g.ntm, BN == 2 * BM) - g.tri_off : g.ntm * g.ntn;
	if (g.ticket && g.ticket_role) // helper launch: as many workgroups as asked for (at most one per tile)
		nblocks = g.helper_wgs < nblocks ? g.helper_wgs : nblocks;
	dim3 grid((unsigned) nblocks, 1, (unsigned) splits), block(NT);
	hipStream_t s = ctx().stream;
	const unsigned dyn = 0u;
	if (akm && bkm)
		hipLaunchKernelGGL((gemm_kernel_p<T, BM, BN, BK, WM, WN, true, true, PF>), grid, block, dyn, s, g);
	else if (akm && !bkm)
		hipLaunchKernelGGL((gemm_kernel_p<T, BM, BN, BK, WM, WN, true, false, PF>), grid, block, dyn, s, g);
	else if (!akm && bkm)
		hipLaunchKernelGGL((gemm_kernel_p<T, BM, BN, BK, WM, WN, false, true, PF>), grid, block, dyn, s, g);
	else
		hipLaunchKernelGGL((gemm_kernel_p<T, BM, BN, BK, WM, WN, false, false, PF>), grid, block, dyn, s, g);
	FH_HIP(hipGetLastError());
}

template <typename T>
void gemm_dev(MatV<T> C, DstKind kind, bool add, MatV<const T> A, MatV<const T> B, T alpha, const GemmExtra<T> *extra)
{
	FH_CHECK(A.nrows == C.nrows && B.ncols == C.ncols && A.ncols == B.nrows, "gemm: shape mismatch");
	idx_t m = C.nrows, n = C.ncols, k = A.ncols;
	if (m == 0 || n == 0)
		return;
	if (k == 0) { // faer/src/linalg/matmul/mod.rs:1190-1198
		if (!add)
			fill_ext<T>(C, kind, (T) 0, extra);
		return;
	}
	FH_CHECK(m < (1L << 31) && n < (1L << 31) && k < (1L << 31), "gemm: dimension too large");
	GemmExtra<T> ex;
	if (extra)
		ex = *extra;
	// level-2 shapes leave for the streaming kernels (matmul/mod.rs:1215-1310: matvec / rank_update dispatch)
	if (kind == DST_FULL && !ex.row_idx && !ex.col_idx && !ex.diag && !ex.a_struct && !ex.b_struct && !ex.inplace) {
		if (k == 1 && m * n >= 1) {
			rank1_dev<T>(C, add, A.p, A.rs, B.p, B.cs, alpha);
			return;
		}
		if (n == 1 && gemv_dev<T>(m, k, A, B.p, B.rs, C.p, C.rs, alpha, add))
			return;
		if (m == 1 && gemv_dev<T>(n, k, B.t(), A.p, A.cs, C.p, C.cs, alpha, add))
			return;
		// one dimension huge, the other two tiny: HBM streams as well (block-reflector steps of a tall QR)
		if (skinny_dev<T>(C, add, A, B, alpha))
			return;
	}
	const bool indexed = ex.row_idx || ex.col_idx;
	// Upper(dst) == Lower(dst^T); dst^T = B^T diag A^T.  Also prefer the unit dst stride along m.
	bool transpose = (kind == DST_UPPER) || (kind == DST_FULL && iabs(C.cs) == 1 && iabs(C.rs) != 1 && !indexed);
	if (transpose) {
		MatV<T> Ct = C.t();
		MatV<const T> At = B.t(), Bt = A.t();
		C = Ct;
		A = At;
		B = Bt;
		std::swap(ex.row_idx, ex.col_idx);
		if (ex.inplace)
			ex.inplace = 3 - ex.inplace; // the aliased operand changes sides with the transposition
		if (ex.k_trim)
			ex.k_trim = 3 - ex.k_trim;
		std::swap(m, n);
		static const int tr[7] = {0, 2, 1, 4, 3, 6, 5}; // FaerBlock of the transposed operand
		const int as = tr[ex.b_struct], bs = tr[ex.a_struct];
		ex.a_struct = as;
		ex.b_struct = bs;
		if (kind == DST_UPPER)
			kind = DST_LOWER;
		// diag sits between lhs and rhs in both orientations; the loader applies it to the rhs rows (k)
	}

	GemmArgs<T> g;
	g.M = (int) m;
	g.N = (int) n;
	g.K = (int) k;
	g.dst = C.p;
	g.drs = C.rs;
	g.dcs = C.cs;
	g.a = A.p;
	g.ars = A.rs;
	g.acs = A.cs;
	g.b = B.p;
	g.brs = B.rs;
	g.bcs = B.cs;
	g.alpha = alpha;
	g.add = add ? 1 : 0;
	g.lower = kind == DST_LOWER ? 1 : 0;
	g.atomic = 0;
	g.ws = nullptr;
	g.row_idx = ex.row_idx;
	g.col_idx = ex.col_idx;
	g.idx64 = ex.idx64;
	g.diag = ex.diag;
	g.diag_stride = ex.diag_stride;
	g.a_struct = ex.a_struct;
	g.b_struct = ex.b_struct;
	g.dst_strict = ex.dst_strict ? 1 : 0;
	g.k_trim = ex.k_trim;
	g.tri_off = 0;
	g.stair_nb = (int) ex.stair_nb;
	g.stair_gap = (int) ex.stair_gap;
	g.stair_row0 = (int) ex.stair_row0;
	{
		g.raster_g = 8; // (2 ... 32 measured within 0.3 % of each other at N = 8192: profiles/notes/DESIGN_history_r01_r05.md 3.1)
	}
	g.epi_serial = 0; // (1: one read-modify-write per element, the pre-round-1-fix epilogue; kept for the record, profiles/r01_exp_syrk_rates.txt)
	const bool extra_path = ex.diag || ex.a_struct || ex.b_struct;
	if (ex.k_trim || ex.tri_skip || ex.stair_nb)
		FH_CHECK(!extra_path && !indexed && !ex.inplace && ctx().gemm_variant < 10, "gemm: k_trim / tri_skip / stair_nb need the plain dense kernel");
	if (ex.stair_nb)
		FH_CHECK(g.lower && !transpose && ex.stair_nb > 0 && ex.stair_gap >= 0 && ex.stair_row0 >= 0, "gemm: stair_nb needs a lower, untransposed dst");

	// loader shapes: K-major when the k stride is the unit one (and the mn stride is not)
	const bool akm = iabs(A.cs) == 1 && iabs(A.rs) != 1;
	const bool bkm = iabs(B.rs) == 1 && iabs(B.cs) != 1;

	// tile config.  ctx().gemm_variant: 0 = auto (pipelined kernels); 1 / 2 = force 128 / 64 tiles (pipelined);
	// 11 / 12 = force 128 / 64 tiles on the non-pipelined kernel (A/B measurements).
	// Measured on MI355X (profiles/r01_call2_gemm_variants.txt): the pipelined 128 x 128 tile is ahead of the
	// 64 x 64 tile from N = 2048 (256 big tiles) upwards and 12 % ahead at N = 8192; below that the small tile
	// gives the 256 CUs more workgroups.
	const int variant = ctx().gemm_variant;
	const idx_t tiles128 = ((m + 127) / 128) * ((n + 127) / 128) / (kind == DST_LOWER ? 2 : 1);
	bool big = tiles128 >= 256;
	if (variant == 1 || variant == 11 || variant == 3)
		big = true;
	if (variant == 2 || variant == 12)
		big = false;
	// variant 6 (A/B, tools/gpu_qr_square_gemm_ab.py): short-wide / tall-narrow FULL outputs with a deep K -- the V^H A products of the QR
	// block applications, 128 x n with K = rows -- on 128 x 128 tiles with more K slices instead of 64 x 64 tiles
	if ((variant == 6 || (variant == 0 && ex.prefer_big_tiles)) && !big && kind == DST_FULL && k >= 2048 && ((m >= 128 && n >= 1024) || (n >= 128 && m >= 1024)))
		big = true;
	// Eight wavefronts of 64 x 64 per workgroup (128 x 256 block tile, one workgroup per CU) for large plain FULL products
	// with a deep K: the A tile is shared by four wavefront columns, the B tile by two rows -- 25 % fewer global loads and
	// LDS stores per flop than four wavefronts on 128 x 128, same LDS reads per MFMA.  DGEMM N = 8192: 68.6 -> 71.3 TFLOP/s
	// in one visit (256 x 128: 70.9; raster group sizes 2 .. 32 make no difference; four wavefronts of 128 x 64 on the same
	// block tile spill in the main loop: 12 TFLOP/s).  One workgroup per CU means nobody covers a workgroup's accumulate
	// epilogue, so the short-K updates of the factorizations stay on the 128 x 128 tile (tools/gpu_gemm_wide_ab.py:
	// r = 8192: K = 512 -2 %, K = 2048 +1.4 %, K = 8192 +2.9 %; r = 15360: +-0.3 %), and so does every lower dst (its
	// 128 x 256 tiles on the diagonal waste more: -5 .. -11 %; the trapezoid enumeration is kept reachable for tests).
	// variant 3 forces the wide tile (full and square lower), 5 forbids it.
	bool legacy = variant >= 10;
	{
		// the pipelined kernel addresses full operand tiles through 32-bit buffer offsets (see its loaders): operands with
		// negative strides, or whose per-lane / per-element offsets inside a tile row do not fit, go to the non-pipelined kernel
		const idx_t ts = (idx_t) sizeof(T);
		const bool a_ok = A.rs >= 0 && A.cs >= 0 && (akm || ((m - 1) * A.rs + 32 * A.cs) * ts < (1L << 31));
		// (K-major B: lane offset <= 15 brs + 31 bcs, element offsets <= (BN - 16) bcs with BN <= 256; its descriptor ends where
		// column N begins, which is only the end of the last column's 16-row tile if the columns are >= 16 elements apart:
		// broadcast views (cs == 0) and overlapping columns (0 < cs < 16) keep the pointer loaders -- ADVICE r04)
		const bool b_ok = B.rs >= 0 && B.cs >= 0 &&
				  (bkm ? (B.cs >= 16 && (16 * B.rs + 256 * B.cs) * ts < (1L << 31)) : ((n - 1) * B.cs + 32 * B.rs) * ts < (1L << 31));
		if (!(a_ok && b_ok) && !legacy) {
			if (ex.tri_skip && !ex.k_trim && !ex.stair_nb && !ex.inplace) {
				// (the Cholesky look-ahead's merged update on a view the buffer-addressed loaders cannot take -- negative
				// strides, a huge leading dimension: ADVICE r04.)  The pointer-addressed kernel has no tile skip: the
				// rectangle below the skipped rows, then the lower triangle right of it, as two plain products.
				const idx_t sk = ex.tri_skip;
				FH_CHECK(kind == DST_LOWER && m == n && sk < m, "gemm: tri_skip needs a square lower dst");
				gemm_dev<T>(C.sub(sk, 0, m - sk, sk), DST_FULL, add, A.sub(sk, 0, m - sk, k), B.sub(0, 0, k, sk), alpha, nullptr);
				gemm_dev<T>(C.sub(sk, sk, m - sk, m - sk), DST_LOWER, add, A.sub(sk, 0, m - sk, k), B.sub(0, sk, k, m - sk), alpha, nullptr);
				return;
			}
			FH_CHECK(!ex.k_trim && !ex.tri_skip && !ex.stair_nb && !ex.inplace, "gemm: operand strides out of range for this product");
			legacy = true;
		}
	}
	const bool plain = !ex.inplace && !ex.diag && !ex.a_struct && !ex.b_struct && !ex.row_idx && !ex.col_idx;
	const bool wide_ok = plain && (kind == DST_FULL ? !ex.k_trim : (m == n));
	const idx_t tiles_wide = ((m + 127) / 128) * ((n + 255) / 256);
	const bool wide_auto = kind == DST_FULL && k >= 2048 && tiles_wide >= 512;
	const int wide = wide_ok && big && variant != 5 && !legacy && (variant == 3 || (variant == 0 && wide_auto)) ? 1 : 0;
	// in-place product (ex.inplace): the aliased operand and dst share rows (transposed orientation: A and C
	// share their rows, one tile must cover all of N) or columns (B and C share their columns, one tile must
	// cover all of M).  A 32-wide tile along the free dimension keeps the launch wide for skinny panels.
	int bm = big ? 128 : 64, bn = bm;
	int shape = big ? 0 : 1; // 0: 128x128  1: 64x64  2: 32x128  3: 128x32
	if (ex.inplace) {
		FH_CHECK(!extra_path && kind == DST_FULL && !indexed, "gemm: in-place product must be a plain one");
		// long free dimension: the 128 x 128 tile (fewer, fatter workgroups; one per CU is enough to fill the
		// chip); short one: 32-wide tiles keep the launch wide.  Either way ONE tile spans the aliased dimension.
		if (ex.inplace == 2) { // A aliases C: all of N inside one tile
			FH_CHECK(n <= 128 && k == n, "gemm: in-place (A) product needs N == K <= 128");
			if (m >= 4096) {
				bm = bn = 128;
				shape = 0;
			} else {
				bm = 32;
				bn = 128;
				shape = 2;
			}
		} else { // B aliases C: all of M inside one tile
			FH_CHECK(m <= 128 && k == m, "gemm: in-place (B) product needs M == K <= 128");
			if (n >= 4096) {
				bm = bn = 128;
				shape = 0;
			} else {
				bm = 128;
				bn = 32;
				shape = 3;
			}
		}
	}
	if (extra_path) {
		bm = bn = 64;
		shape = 1;
	}
	if (wide) {
		bm = 128;
		bn = 256;
		shape = 5;
	}
	g.ntm = (int) ((m + bm - 1) / bm);
	g.ntn = (int) ((n + bn - 1) / bn);
	g.tri_enum = (g.lower && m == n && !ex.stair_nb) ? 1 : 0;
	if (ex.tri_skip) {
		FH_CHECK(g.tri_enum && ex.tri_skip % bm == 0 && ex.tri_skip < m, "gemm: tri_skip needs a square lower dst and a tile-aligned skip");
		const int st = (int) (ex.tri_skip / bm);
		g.tri_off = tri_tiles(st, bn == 2 * bm);
	}

	// split-K for few-tile / deep-K products
	idx_t tiles = g.tri_enum ? (idx_t) tri_tiles(g.ntm, bn == 2 * bm) - g.tri_off : (idx_t) g.ntm * g.ntn;
	int splits = 1;
	// Deep-K products with few output tiles are split along K.  Rectangular outputs (the V^H A / V^H V products of QR:
	// K = rows of the panel, a few tiles of output) split from K = 1024 in slices of >= 256 (measured: square QR
	// N = 4096 99.6 -> 80.2 ms); triangular outputs (the diagonal-block updates of the blocked Cholesky, which sit
	// on its critical chain) keep the coarser rule, the finer one cost the N = 16384 factorization 0.6 ms.
	const idx_t splitk_mink = 1024, splitk_chunk = 256;
	const idx_t mink = kind == DST_FULL ? splitk_mink : 4096, chunk = kind == DST_FULL ? splitk_chunk : 1024;
	if (tiles < 256 && k >= mink && !indexed && !ex.k_trim && !ex.stair_nb) {
		splits = (int) ((512 + tiles - 1) / tiles);
		idx_t max_splits = k / chunk;
		if (splits > max_splits)
			splits = (int) max_splits;
		if (splits > 1024)
			splits = 1024;
		if (splits < 1)
			splits = 1;
	}
	idx_t kps = (k + splits - 1) / splits;
	kps = (kps + 15) / 16 * 16;
	splits = (int) ((k + kps - 1) / kps);
	g.k_per_split = (int) kps;
	// deep-K products are split over workgroups; the slices meet in a workspace and are added in a fixed order by
	// a second small kernel, so the result does not depend on scheduling (hardware atomics would be faster by one
	// launch but make every QR run differ in the last bits)
	Scratch wsb(splits > 1 ? (size_t) splits * (size_t) m * (size_t) n * sizeof(T) : 256);
	g.ws = wsb.as<T>();
	if (splits > 1)
		g.atomic = 2;

	if (ex.inplace)
		FH_CHECK(splits == 1, "gemm: in-place product cannot be split along K");
	g.fast_io = 0;
	{
		// (the wave-uniform byte offsets of a tile, < 64 dcs sizeof(T), and the per-lane ones, < 16 dcs sizeof(T), are 32-bit)
		if (plain && !legacy && splits == 1 && C.rs == 1 && C.cs > 0 && C.cs < (1L << 21) && !ex.stair_nb) {
			if (!add)
				g.fast_io = 1;
			else if (alpha == (T) 1)
				g.fast_io = 2;
			else if (alpha == (T) -1)
				g.fast_io = 3;
		}
	}
	g.ticket = nullptr;
	g.ticket_total = g.ticket_role = g.ticket_margin = g.helper_wgs = 0;
	if (ex.ticket) {
		FH_CHECK(!extra_path && !legacy && shape == 0 && splits == 1 && !ex.inplace, "gemm: ticketed launches are for the 128 x 128 pipelined tile");
		g.ticket = ex.ticket;
		g.ticket_role = ex.helper_wgs > 0 ? 1 : 0;
		g.ticket_margin = ex.helper_margin;
		g.ticket_total = (int) tiles;
		g.helper_wgs = ex.helper_wgs; // (a helper launch's grid: launch_cfg_p)
	}
	// (profile class 0: the big pipelined tiles -- the trailing updates of the factorizations, the headline product)
	const double out_elems = g.tri_enum ? 0.5 * (double) m * (double) (m + 1) - 0.5 * (double) ex.tri_skip * (double) (ex.tri_skip + 1) : (double) m * (double) n;
	ProfScope prof(!extra_path && !legacy && (shape == 0 || shape == 5) ? 0 : -1, 2.0 * (double) k * out_elems);
	prof.sp.d[0] = (long) m;
	prof.sp.d[1] = (long) n;
	prof.sp.d[2] = (long) k;
	prof.sp.d[3] = g.tri_enum ? 1 + (long) ex.tri_skip : 0;
	if (extra_path)
		launch_cfg<T, 64, 64, 2, 2, true>(g, akm, bkm, splits); // triangular operands / diag scaling
	else if (shape == 5)
		launch_cfg_p<T, 128, 256, 2, 4>(g, akm, bkm, splits);
	else if (shape == 2)
		launch_cfg_p<T, 32, 128, 1, 4>(g, akm, bkm, splits);
	else if (shape == 3)
		launch_cfg_p<T, 128, 32, 4, 1>(g, akm, bkm, splits);
	else if (legacy && shape == 0)
		launch_cfg<T, 128, 128, 2, 2, false>(g, akm, bkm, splits);
	else if (legacy)
		launch_cfg<T, 64, 64, 2, 2, false>(g, akm, bkm, splits);
	else if (shape == 0)
		launch_cfg_p<T, 128, 128, 2, 2>(g, akm, bkm, splits);
	else
		launch_cfg_p<T, 64, 64, 2, 2>(g, akm, bkm, splits);
	if (splits > 1) {
		const idx_t total = m * n;
		const idx_t blocks = (total + 15) / 16;
		hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned) blocks), dim3(256), 0, ctx().stream, C.p, C.rs, C.cs, (int) m, (int) n,
				   wsb.as<T>(), splits, alpha, add ? 1 : 0, g.lower, g.dst_strict);
		FH_HIP(hipGetLastError());
	}
}

// faer/src/linalg/matmul/triangular.rs:1246-1495: only the structured part of each operand is accessed
// (strict => diagonal is 0, unit => diagonal is 1) and only the structured part of dst is written.
template <typename T>
void matmul_triangular_dev(MatV<T> C, int c_s, bool add, MatV<const T> A, int a_s, MatV<const T> B, int b_s, T alpha)
{
	GemmExtra<T> ex;
	ex.a_struct = a_s;
	ex.b_struct = b_s;
	DstKind kind = DST_FULL;
	if (c_s == 1 || c_s == 3 || c_s == 5)
		kind = DST_LOWER;
	else if (c_s == 2 || c_s == 4 || c_s == 6)
		kind = DST_UPPER;
	ex.dst_strict = c_s >= 3;
	gemm_dev<T>(C, kind, add, A, B, alpha, &ex);
}

template <typename T> double gemm_time_ms(MatV<T> C, MatV<const T> A, MatV<const T> B, int iters)
{
	hipEvent_t e0, e1;
	FH_HIP(hipEventCreate(&e0));
	FH_HIP(hipEventCreate(&e1));
	gemm_dev<T>(C, DST_FULL, false, A, B, (T) 1); // warm
	FH_HIP(hipEventRecord(e0, ctx().stream));
	for (int i = 0; i < iters; ++i)
		gemm_dev<T>(C, DST_FULL, false, A, B, (T) 1);
	FH_HIP(hipEventRecord(e1, ctx().stream));
	FH_HIP(hipEventSynchronize(e1));
	float ms = 0;
	FH_HIP(hipEventElapsedTime(&ms, e0, e1));
	FH_HIP(hipEventDestroy(e0));
	FH_HIP(hipEventDestroy(e1));
	return (double) ms / iters;
}

// ------------------------------------------------------------------------------------------------
// MFMA issue-rate probe (register-only): the measured ceiling next to the datasheet peak.
// ------------------------------------------------------------------------------------------------
template <typename T> __global__ __launch_bounds__(256) void mfma_peak_kernel(T *out, int iters)
{
	typedef typename Mfma<T>::acc_t acc_t;
	acc_t c0 = (acc_t) (T) 0, c1 = c0, c2 = c0, c3 = c0;
	T a = (T) (threadIdx.x & 7) * (T) 0.125, b = (T) 1.0 - (T) (threadIdx.x & 3) * (T) 0.25;
	for (int i = 0; i < iters; ++i) {
		c0 = Mfma<T>::run(a, b, c0);
		c1 = Mfma<T>::run(a, b, c1);
		c2 = Mfma<T>::run(a, b, c2);
		c3 = Mfma<T>::run(a, b, c3);
	}
	acc_t s = c0 + c1 + c2 + c3;
	if (s[0] + s[1] + s[2] + s[3] == (T) 123456789)
		out[0] = s[0];
}

double mfma_peak_tflops(bool f64, int iters)
{
	ctx().ensure_device();
	hipEvent_t e0, e1;
	FH_HIP(hipEventCreate(&e0));
	FH_HIP(hipEventCreate(&e1));
	Scratch out(64);
	const int blocks = 256 * 8; // 8 workgroups of 4 waves per CU
	hipStream_t s = ctx().stream;
	for (int rep = 0; rep < 2; ++rep) {
		if (rep == 1)
			FH_HIP(hipEventRecord(e0, s));
		if (f64)
			hipLaunchKernelGGL(mfma_peak_kernel<double>, dim3(blocks), dim3(256), 0, s, out.as<double>(), iters);
		else
			hipLaunchKernelGGL(mfma_peak_kernel<float>, dim3(blocks), dim3(256), 0, s, out.as<float>(), iters);
	}
	FH_HIP(hipEventRecord(e1, s));
	FH_HIP(hipEventSynchronize(e1));
	float ms = 0;
	FH_HIP(hipEventElapsedTime(&ms, e0, e1));
	FH_HIP(hipEventDestroy(e0));
	FH_HIP(hipEventDestroy(e1));
	const double flops = (double) blocks * 4 /*waves*/ * (double) iters * 4 /*mfma*/ * 2.0 * 16 * 16 * 4;
	return flops / (ms * 1e-3) / 1e12;
}

template void gemm_dev<double>(MatV<double>, DstKind, bool, MatV<const double>, MatV<const double>, double,
			       const GemmExtra<double> *);
template void gemm_dev<float>(MatV<float>, DstKind, bool, MatV<const float>, MatV<const float>, float,
			      const GemmExtra<float> *);
template void matmul_triangular_dev<double>(MatV<double>, int, bool, MatV<const double>, int, MatV<const double>, int,
					    double);
template void matmul_triangular_dev<float>(MatV<float>, int, bool, MatV<const float>, int, MatV<const float>, int, float);
template double gemm_time_ms<double>(MatV<double>, MatV<const double>, MatV<const double>, int);
template double gemm_time_ms<float>(MatV<float>, MatV<const float>, MatV<const float>, int);
template void fill_dev<double>(MatV<double>, DstKind, double);
template void fill_dev<float>(MatV<float>, DstKind, float);

} // namespace fh
