// Cooperative LU panel kernel of round 4: 64 columns x all rows in one launch, ONE cross-workgroup exchange per column.
//
// Replaces the column loop of faer/src/linalg/lu/partial_pivoting/factor.rs:19-67 (lu_in_place_unblocked: first row of
// strictly largest |a|, interchange, scaling by the reciprocal pivot, rank-1 update as fma(l, -u, dst)) for a leaf of
// the recursion (factor.rs:68-187).  Same arithmetic per entry and the same pivots as getrf_panel2_kernel (getrf.hip),
// which stays the fallback for panels taller than this kernel keeps resident; what changes is the dependent chain per
// column (profiles/r03_lu_kernel_stats.csv: 4.25 us per column, two fabric round trips + three workgroup barriers + the
// whole rank-1 update between two pivot searches):
//
//   * rows never move inside the leaf.  Every register row carries a LABEL = the row index it would have after the
//     interchanges so far; "interchange J <-> p" swaps two labels, ties are decided on labels, rows are written to their
//     label positions at the end.  Nobody needs the displaced diagonal row, so there is nothing to patch;
//   * a wavefront keeps 64 (fp64) or 128 (fp32) rows, one or two per lane, all 64 panel columns in registers in the
//     ROTATED order of getrf_panel2_kernel (the column being eliminated sits at a compile-time position);
//   * per column ONE exchange of 64-byte headers {label, a = x_c[J], s = x_c[J + 1], l = l_{J-1}[c]} of each workgroup's
//     candidate row c.  A header is enough to bring column J + 1 up to date and search it: the next header goes out
//     ~0.25 us after the previous sweep ended;
//   * everything else LAGS one column behind, in the shadow of the next exchange: the winner's whole row is published
//     together with the header (updated through step J - 2 only), fetched after the next header is on its way and
//     CORRECTED by the consumer, u_J[c] = fma(l_{J-1}[p], -u_{J-1}[c], record[c]) -- the operation its owner applies to
//     it, bit for bit -- then the rank-1 update of the columns >= J + 2 runs;
//   * inside a workgroup the 4 or 8 wavefronts combine their candidates through LDS (the only barrier of a column);
//     EVERY wavefront then sweeps the <= 256 headers itself, so nothing is broadcast back.
// The schedule is modelled in numpy (tests/diag/proto_lu_wpanel.py, tests/test_lu_wpanel_proto.py: pivots and factors
// bitwise equal to the unblocked elimination, ties and zero columns included).
//
// Exchange records are data-tagged granules (xwg.h, recipe R2 of cdna_hip_programming.md Guideline 16): a double travels
// as {tag, high word, tag, low word} in ONE 16-byte write-through store and is read back with sc1 loads; a reader that
// finds both tags has the value.  Tags = column epoch, never 0; the workspace is zeroed once per factorization.  Header
// slots alternate with the column parity, row-record slots with the column modulo 4: a workgroup can be at most one
// sweep ahead of the slowest one, and a row record is consumed before its consumer publishes the next header but one.
// Every spin is bounded; a timeout raises status word 2 (getrf_dev reruns on the non-cooperative leaves).
#pragma once
#include <climits>

#include "common.h"
#include "lds_blocks.h"
#include "xwg.h"

namespace fh {

static __device__ __forceinline__ bool better(double av, int ar, double bv, int br)
{
	return av > bv || (av == bv && ar < br);
}

// Wave-wide arg-max of (|a|, row) with the smaller row winning ties, on the DPP network instead of LDS-crossbar
// shuffles: quad_perm + row_half_mirror + row_mirror reduce each row of 16 lanes, row_bcast:15 / row_bcast:31
// carry the partial results across the four rows, lane 63 ends up with the wave result and broadcasts it.
// (six data-parallel steps of ~7 VALU instructions each instead of eighteen ds_bpermute round trips.)
static __device__ __forceinline__ void wave_argmax2(double &v, int &r)
{
#define FH_DPP_STEP(ctrl, rmask)                                                                                         \
	do {                                                                                                             \
		const int lo_ = __double2loint(v), hi_ = __double2hiint(v);                                              \
		const int olo_ = __builtin_amdgcn_update_dpp(lo_, lo_, ctrl, rmask, 0xf, false);                         \
		const int ohi_ = __builtin_amdgcn_update_dpp(hi_, hi_, ctrl, rmask, 0xf, false);                         \
		const int or_ = __builtin_amdgcn_update_dpp(r, r, ctrl, rmask, 0xf, false);                              \
		const double ov_ = __hiloint2double(ohi_, olo_);                                                         \
		if (better(ov_, or_, v, r)) {                                                                            \
			v = ov_;                                                                                         \
			r = or_;                                                                                         \
		}                                                                                                        \
	} while (0)
	FH_DPP_STEP(0xB1, 0xf);	 // quad_perm [1,0,3,2]
	FH_DPP_STEP(0x4E, 0xf);	 // quad_perm [2,3,0,1]
	FH_DPP_STEP(0x141, 0xf); // row_half_mirror
	FH_DPP_STEP(0x140, 0xf); // row_mirror: every lane of a row now holds the row's best
	FH_DPP_STEP(0x142, 0xa); // row_bcast:15 into rows 1 and 3
	FH_DPP_STEP(0x143, 0xc); // row_bcast:31 into rows 2 and 3
#undef FH_DPP_STEP
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
	v = __hiloint2double(hi, lo);
	r = __builtin_amdgcn_readlane(r, 63);
}

// The same reduction for the new kernel, without a branch: the values first (v_max_f64 on the DPP network: two moves and
// one max per step), then the smallest label among the lanes that hold the maximum (v_min_i32 with a DPP operand).  Values
// are never NaN here (candidates are |a| > 0, 0 for the diagonal row of a zero column, -1 for "none").  STEPS = 6: all 64
// lanes; STEPS = 3: lanes 0-7 only (the wavefronts of a workgroup).  ~40 instructions instead of ~100.
template <int STEPS> static __device__ __forceinline__ void lw_argmax(double v, int r, double &M, int &L)
{
	double m = v;
#define FH_DPP_MAX(ctrl, rmask)                                                                                          \
	do {                                                                                                             \
		const int lo_ = __double2loint(m), hi_ = __double2hiint(m);                                              \
		const int olo_ = __builtin_amdgcn_update_dpp(lo_, lo_, ctrl, rmask, 0xf, false);                         \
		const int ohi_ = __builtin_amdgcn_update_dpp(hi_, hi_, ctrl, rmask, 0xf, false);                         \
		m = __builtin_fmax(m, __hiloint2double(ohi_, olo_));                                                     \
	} while (0)
	FH_DPP_MAX(0xB1, 0xf);
	FH_DPP_MAX(0x4E, 0xf);
	FH_DPP_MAX(0x141, 0xf);
	if (STEPS == 6) {
		FH_DPP_MAX(0x140, 0xf);
		FH_DPP_MAX(0x142, 0xa);
		FH_DPP_MAX(0x143, 0xc);
	}
#undef FH_DPP_MAX
	constexpr int SRC = STEPS == 6 ? 63 : 0;
	M = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m), SRC), __builtin_amdgcn_readlane(__double2loint(m), SRC));
	int rr = v == M ? r : INT_MAX;
#define FH_DPP_MIN(ctrl, rmask) rr = min(rr, __builtin_amdgcn_update_dpp(rr, rr, ctrl, rmask, 0xf, false))
	FH_DPP_MIN(0xB1, 0xf);
	FH_DPP_MIN(0x4E, 0xf);
	FH_DPP_MIN(0x141, 0xf);
	if (STEPS == 6) {
		FH_DPP_MIN(0x140, 0xf);
		FH_DPP_MIN(0x142, 0xa);
		FH_DPP_MIN(0x143, 0xc);
	}
#undef FH_DPP_MIN
	L = __builtin_amdgcn_readlane(rr, SRC);
}

constexpr int LW_W = 64;	    // leaf width = wavefront size: lane c <-> register position c of a published row
constexpr int LW_GMAX = 256;	    // workgroups per panel
constexpr int LW_HDR_BYTES = 64;    // {label}, {a}, {s}, {l}: four 16-byte granule pairs
constexpr int LW_ROW_BYTES = LW_W * 16;
constexpr int LW_NSH = 2; // header slots (column parity)
constexpr int LW_NSR = 4; // row-record slots (column modulo 4)
constexpr size_t LW_HDR_WS = (size_t) LW_NSH * LW_GMAX * LW_HDR_BYTES;
constexpr size_t LW_ROW_WS = (size_t) LW_NSR * LW_GMAX * LW_ROW_BYTES;
constexpr size_t LW_WS_BYTES = LW_HDR_WS + LW_ROW_WS;
constexpr int LW_SPIN_MAX = 1 << 20;

typedef unsigned int lw_u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct WPanelArgs {
	T *P;
	idx_t rs, cs;
	int m, w;
	int *piv; // piv[j] = row_base + pivot row
	int row_base;
	unsigned char *ws; // LW_WS_BYTES: headers, then row records
	unsigned epoch_base;
	int *status;
	unsigned long long *phase; // timing build: per-phase tick sums of workgroup 0 / wave 0
};

template <int NW> struct WPanelShared {
	double cv[2][NW]; // the wavefronts' candidates of the current column, slots alternate with the column parity
	int lab[2][NW];
	double trans[NW][LW_W]; // per wavefront: the candidate row on its way from one lane's registers to 64 lanes
};

static __device__ __forceinline__ lw_u32x4 lw_pack(unsigned tag, double v)
{
	const unsigned long long b = (unsigned long long) __double_as_longlong(v);
	lw_u32x4 q;
	q.x = tag;
	q.y = (unsigned) (b >> 32);
	q.z = tag;
	q.w = (unsigned) b;
	return q;
}
static __device__ __forceinline__ double lw_unpack(lw_u32x4 q)
{
	return __longlong_as_double((long long) (((unsigned long long) q.y << 32) | (unsigned long long) q.w));
}
static __device__ __forceinline__ bool lw_ok(lw_u32x4 q, unsigned tag) { return q.x == tag && q.z == tag; }
// sc1 (aux = 16): write-through stores, loads served past the L1 (MI355X_MICROARCH.md, inter-workgroup visibility)
// (voff: the lane's part of the byte offset, soff: the wave-uniform part -- kept in an SGPR so that the eight unrolled step
// bodies do not each pin their own address registers)
static __device__ __forceinline__ lw_u32x4 lw_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
	return __builtin_amdgcn_raw_buffer_load_b128(r, (int) voff, (int) soff, 16);
}
static __device__ __forceinline__ void lw_store(lw_u32x4 q, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
	__builtin_amdgcn_raw_buffer_store_b128(q, r, (int) voff, (int) soff, 16);
}

#ifdef FH_LU_TIMING
#define LW_TICK(slot)                                                                                                   \
	do {                                                                                                            \
		const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                           \
		if (blockIdx.x == 0 && threadIdx.x == 0)                                                               \
			tk[slot] += now_ - t_last;                                                                      \
		t_last = now_;                                                                                          \
	} while (0)
#else
#define LW_TICK(slot)                                                                                                   \
	do {                                                                                                            \
	} while (0)
#endif

// Sweep of the G headers of column J by one wavefront (lane t reads workgroups t, t + 64, ...): winner's label p,
// workgroup gw and header values; false if nobody has a candidate (then there is no row record either).  A timeout sets `dead` (wave uniform): the kernel then runs to its end without waiting
// for anything and without storing anything -- no early exits, the eight step bodies stay one straight line of code.
static __device__ __forceinline__ bool lw_sweep(__amdgpu_buffer_rsrc_t hr, int G, unsigned tag, int J, int lane, bool &dead, int &p, int &gw, double &a,
						double &s, double &l)
{
	double bcv, ba, bs, bl;
	int blab, bg;
	for (int spin = 0;; ++spin) {
		bool ok = true;
		bcv = -1.0;
		blab = INT_MAX;
		ba = bs = bl = 0.0;
		bg = 0;
		for (int t = lane; t < G; t += 64) {
			const unsigned voff = (unsigned) (t * LW_HDR_BYTES), soff = (unsigned) ((J & 1) * G * LW_HDR_BYTES);
			const lw_u32x4 h0 = lw_load(hr, voff, soff), h1 = lw_load(hr, voff + 16, soff), h2 = lw_load(hr, voff + 32, soff),
				       h3 = lw_load(hr, voff + 48, soff);
			ok = ok && lw_ok(h0, tag) && lw_ok(h1, tag) && lw_ok(h2, tag) && lw_ok(h3, tag);
			const int lb = (int) h0.y;
			const double av = lw_unpack(h1), fa = fabs(av);
			// a published candidate is a real one (|a| > 0) or the diagonal row of a zero / NaN-only column (factor.rs:35-43)
			const double cv = lb == INT_MAX ? -1.0 : (fa > 0.0 ? fa : 0.0);
			if (better(cv, lb, bcv, blab)) {
				bcv = cv;
				blab = lb;
				ba = av;
				bs = lw_unpack(h2);
				bl = lw_unpack(h3);
				bg = t;
			}
		}
		if (dead || __all(ok))
			break;
		if (spin >= LW_SPIN_MAX) {
			dead = true;
			break;
		}
		__builtin_amdgcn_s_sleep(1);
	}
	double wv;
	int wl;
	lw_argmax<6>(bcv, blab, wv, wl);
	// (no candidate at all: cannot happen for J < m, the row labelled J is always one; the padded steps J >= m keep the diagonal)
	const bool any = wv >= 0.0;
	const unsigned long long bal = (unsigned long long) __ballot(blab == wl && bcv == wv);
	const int wlane = any && bal != 0ull ? __builtin_amdgcn_readfirstlane((int) __ffsll(bal) - 1) : 0;
	p = any ? wl : J;
	gw = any ? __builtin_amdgcn_readlane(bg, wlane) : 0;
	a = lane_bcast(ba, wlane);
	s = lane_bcast(bs, wlane);
	l = lane_bcast(bl, wlane);
	return any;
}

// Candidate of column Jn (at register position POS) among the rows labelled >= Jn: wavefront arg-max, combination of the
// workgroup's NW wavefronts through LDS, then the wavefront that owns the workgroup's candidate publishes the header
// {label, a = x[POS], s = x[POS + 1], l = lprev} and the row record.  Contains the one barrier of a column.
template <typename T, int RPT, int NW, int POS>
static __device__ __forceinline__ void lw_publish(const WPanelArgs<T> &a, __amdgpu_buffer_rsrc_t hr, __amdgpu_buffer_rsrc_t rr, T (&x)[RPT][LW_W],
						  const int (&lab)[RPT], const T (&lprev)[RPT], WPanelShared<NW> &sh, int Jn, int rot, int G)
{
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int g = blockIdx.x;
	const unsigned tag = a.epoch_base + (unsigned) Jn + 1u;
	const int par = Jn & 1;
	double bcv = -1.0;
	int blab = INT_MAX, bi = 0;
#pragma unroll
	for (int i = 0; i < RPT; ++i) {
		const int lb = lab[i];
		const double av = fabs((double) x[i][POS]);
		const double cv = lb >= Jn ? (av > 0.0 ? av : (lb == Jn ? 0.0 : -1.0)) : -1.0;
		if (better(cv, lb, bcv, blab)) {
			bcv = cv;
			blab = lb;
			bi = i;
		}
	}
	double wv;
	int wl;
	lw_argmax<6>(bcv, blab, wv, wl);
	const bool has = wv >= 0.0;
	int ol = 0, oi = 0;
	double ca = 0.0, cs = 0.0, cl = 0.0;
	if (has) { // wave uniform
		ol = __builtin_amdgcn_readfirstlane((int) __ffsll((unsigned long long) __ballot(blab == wl && bcv == wv)) - 1);
		oi = __builtin_amdgcn_readlane(bi, ol);
		T va = x[0][POS], vs = x[0][POS + 1], vl = lprev[0];
#pragma unroll
		for (int i = 1; i < RPT; ++i)
			if (oi == i) {
				va = x[i][POS];
				vs = x[i][POS + 1];
				vl = lprev[i];
			}
		ca = (double) lane_bcast(va, ol);
		cs = (double) lane_bcast(vs, ol);
		cl = (double) lane_bcast(vl, ol);
	}
	if (lane == 0) {
		sh.cv[par][wave] = has ? wv : -1.0;
		sh.lab[par][wave] = has ? wl : INT_MAX;
	}
	__syncthreads();
	const double ecv = lane < NW ? sh.cv[par][lane] : -1.0;
	const int elab = lane < NW ? sh.lab[par][lane] : INT_MAX;
	double gv;
	int gl;
	lw_argmax<3>(ecv, elab, gv, gl); // (NW <= 8: lanes 0-7 hold the entries)
	const bool ghas = gv >= 0.0;
	int ow = 0; // a workgroup without candidate: wavefront 0 publishes "none"
	if (ghas)
		ow = __builtin_amdgcn_readfirstlane((int) __ffsll((unsigned long long) __ballot(lane < NW && elab == gl && ecv == gv)) - 1);
	if (wave != ow)
		return;
	{
		const double hv = lane == 1 ? ca : (lane == 2 ? cs : cl);
		lw_u32x4 q = lw_pack(tag, hv);
		if (lane == 0) {
			q.y = (unsigned) (ghas ? gl : INT_MAX);
			q.w = 0u;
		}
		if (lane < 4)
			lw_store(q, hr, (unsigned) (lane * 16), (unsigned) ((par * G + g) * LW_HDR_BYTES));
	}
	if (!ghas)
		return;
	// the candidate row: one lane's registers -> LDS -> lane c stores register position c = panel column (c + rot) mod 64
#pragma unroll
	for (int i = 0; i < RPT; ++i)
		if (oi == i) { // wave uniform
			if (lane == ol) {
#pragma unroll
				for (int c = 0; c < LW_W; ++c)
					sh.trans[wave][c] = (double) x[i][c];
			}
		}
	__builtin_amdgcn_wave_barrier();
	const double rv = sh.trans[wave][lane];
	__builtin_amdgcn_wave_barrier();
	lw_store(lw_pack(tag, rv), rr, (unsigned) (((lane + rot) & 63) * 16), (unsigned) (((Jn & 3) * G + g) * LW_ROW_BYTES));
}

// One column step (column J = 8 grp + JJ at register position JJ).  Steps J >= steps (a leaf narrower than a multiple of 8
// columns) run like any other on the zero padding and store nothing.
template <typename T, int RPT, int NW, int JJ>
static __device__ __forceinline__ void lw_step(const WPanelArgs<T> &a, __amdgpu_buffer_rsrc_t hr, __amdgpu_buffer_rsrc_t rr, T (&x)[RPT][LW_W],
					       int (&lab)[RPT], T &uprev, WPanelShared<NW> &sh, int grp, int G, int steps, int nsteps, bool &dead
#ifdef FH_LU_TIMING
					       ,
					       unsigned long long (&tk)[8], unsigned long long &t_last
#endif
)
{
	const int tid = threadIdx.x, lane = tid & 63;
	const int J = grp * 8 + JJ;
	const int rot = grp * 8;
	const int lim = LW_W - rot; // positions < lim hold unfinished columns
	const unsigned tag = a.epoch_base + (unsigned) J + 1u;
	// ---- 1. the headers of column J
	int p, gw;
	double da, ds, dl;
	const bool any = lw_sweep(hr, G, tag, J, lane, dead, p, gw, da, ds, dl);
	LW_TICK(0);
	// ---- 2. the winner's row record is fetched while the next header is prepared (used in 5.)
	const unsigned rvoff = (unsigned) (((lane + rot) & 63) * 16), rsoff = (unsigned) (((J & 3) * G + gw) * LW_ROW_BYTES);
	lw_u32x4 rv = lw_load(rr, rvoff, rsoff);
	// ---- 3. u_J[J + 1] from the header: the published s lags one step (factor.rs:59-64: fma(l, -u, dst))
	const T lp = (T) dl;
	const T uJ1 = fh_fma(lp, -lane_bcast(uprev, JJ + 1), (T) ds);
	// "interchange" J <-> p on the labels
#pragma unroll
	for (int i = 0; i < RPT; ++i)
		lab[i] = lab[i] == J ? p : (lab[i] == p ? J : lab[i]);
	if (blockIdx.x == 0 && tid == 0 && J < steps)
		a.piv[J] = a.row_base + p;
	// scaling by the reciprocal pivot (factor.rs:45-57) and the update of column J + 1 alone
	const T inv = (T) 1 / (T) da;
	T l[RPT];
#pragma unroll
	for (int i = 0; i < RPT; ++i) {
		const bool act = lab[i] > J;
		const T li = x[i][JJ] * inv;
		l[i] = act ? li : (T) 0;
		x[i][JJ] = act ? li : x[i][JJ];
		const T nx = fh_fma(li, -uJ1, x[i][JJ + 1]);
		x[i][JJ + 1] = act && JJ + 1 < lim ? nx : x[i][JJ + 1];
	}
	LW_TICK(1);
	// ---- 4. candidate of column J + 1, header and row record on their way
	if (J + 1 < nsteps)
		lw_publish<T, RPT, NW, JJ + 1>(a, hr, rr, x, lab, l, sh, J + 1, rot, G);
	LW_TICK(2);
	// ---- 5. the winner's row of column J, corrected by the step it lags behind
	for (int spin = 0;; ++spin) {
		if (dead || !any || __all(lw_ok(rv, tag)))
			break;
		if (spin >= LW_SPIN_MAX) {
			dead = true;
			break;
		}
		__builtin_amdgcn_s_sleep(1);
		rv = lw_load(rr, rvoff, rsoff);
	}
	const T ucur = fh_fma(lp, -uprev, (T) lw_unpack(rv));
	LW_TICK(3);
	// ---- 6. rank-1 update of the columns >= J + 2 (positions in blocks of 8, a block takes part while it holds unfinished
	//         columns; the multipliers of the pivot row reach the FMAs through the scalar unit, as in getrf_panel2_kernel)
#pragma unroll
	for (int cb = 0; cb < LW_W / 8; ++cb) {
		if (cb * 8 + 7 > JJ + 1 && cb * 8 < lim) {
			T u[8];
#pragma unroll
			for (int k = 0; k < 8; ++k)
				u[k] = lane_bcast(ucur, cb * 8 + k);
#pragma unroll
			for (int i = 0; i < RPT; ++i) {
				const bool act = lab[i] > J;
#pragma unroll
				for (int k = 0; k < 8; ++k)
					if (cb * 8 + k > JJ + 1) {
						const T nx = fh_fma(l[i], -u[k], x[i][cb * 8 + k]);
						x[i][cb * 8 + k] = act ? nx : x[i][cb * 8 + k];
					}
			}
		}
	}
	uprev = ucur;
	LW_TICK(4);
}

// grid = G workgroups of NW wavefronts, all resident; wavefront v of workgroup g owns the rows [(g NW + v) 64 RPT, +64 RPT)
template <typename T, int RPT, int NW> __global__ __launch_bounds__(NW * 64) void getrf_wpanel_kernel(const WPanelArgs<T> a)
{
	__shared__ WPanelShared<NW> sh;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int G = gridDim.x;
	const int r0 = (blockIdx.x * NW + wave) * 64 * RPT;
	const int w = a.w;
	const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc((void *) a.ws, 0, (int) LW_HDR_WS, 0x00020000);
	const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *) (a.ws + LW_HDR_WS), 0, (int) LW_ROW_WS, 0x00020000);
	T x[RPT][LW_W];
	int lab[RPT];
#pragma unroll
	for (int i = 0; i < RPT; ++i) {
		const int gr = r0 + i * 64 + lane;
		lab[i] = gr < a.m ? gr : -1; // rows past the end never take part
#pragma unroll
		for (int c = 0; c < LW_W; ++c) {
			const bool in = gr < a.m && c < w;
			const T v = a.P[in ? (idx_t) gr * a.rs + (idx_t) c * a.cs : (idx_t) 0];
			x[i][c] = in ? v : (T) 0;
		}
	}
	const int steps = min(w, a.m);
	const int nsteps = (steps + 7) & ~7; // whole groups of 8 steps
	T uprev = (T) 0;
	bool dead = false;
#ifdef FH_LU_TIMING
	unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long t_last = __builtin_amdgcn_s_memtime();
#define LW_TARGS , tk, t_last
#else
#define LW_TARGS
#endif
	{
		T l0[RPT];
#pragma unroll
		for (int i = 0; i < RPT; ++i)
			l0[i] = (T) 0;
		lw_publish<T, RPT, NW, 0>(a, hr, rr, x, lab, l0, sh, 0, 0, G);
	}
	int rot = 0;
	for (int grp = 0; grp * 8 < steps; ++grp) {
		lw_step<T, RPT, NW, 0>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead LW_TARGS);
		lw_step<T, RPT, NW, 1>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead LW_TARGS);
		lw_step<T, RPT, NW, 2>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead LW_TARGS);
		lw_step<T, RPT, NW, 3>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead LW_TARGS);
		lw_step<T, RPT, NW, 4>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead LW_TARGS);
		lw_step<T, RPT, NW, 5>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead LW_TARGS);
		lw_step<T, RPT, NW, 6>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead LW_TARGS);
		lw_step<T, RPT, NW, 7>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead LW_TARGS);
		// rotate every row left by 8: the finished columns go to the tail
#pragma unroll
		for (int i = 0; i < RPT; ++i) {
			T t8[8];
#pragma unroll
			for (int k = 0; k < 8; ++k)
				t8[k] = x[i][k];
#pragma unroll
			for (int c = 0; c + 8 < LW_W; ++c)
				x[i][c] = x[i][c + 8];
#pragma unroll
			for (int k = 0; k < 8; ++k)
				x[i][LW_W - 8 + k] = t8[k];
		}
		uprev = __shfl(uprev, (lane + 8) & 63);
		rot += 8;
		LW_TICK(5);
	}
#undef LW_TARGS
	if (dead) { // wave uniform: an exchange timed out, nothing is stored (getrf_dev restores A and reruns without this kernel)
		if (lane == 0)
			atomicExch(a.status + 2, 1);
		return;
	}
#pragma unroll
	for (int i = 0; i < RPT; ++i) {
		const int dr = lab[i]; // the row's position after the leaf's interchanges
#pragma unroll
		for (int c = 0; c < LW_W; ++c) {
			const int gc = (c + rot) & (LW_W - 1); // panel column of register position c
			if (dr >= 0 && gc < w)
				a.P[(idx_t) dr * a.rs + (idx_t) gc * a.cs] = x[i][c];
		}
	}
#ifdef FH_LU_TIMING
	if (blockIdx.x == 0 && tid == 0 && a.phase) {
		for (int k = 0; k < 8; ++k)
			atomicAdd(a.phase + k, tk[k]);
		atomicAdd(a.phase + 8, (unsigned long long) steps);
	}
#endif
}

} // namespace fh
