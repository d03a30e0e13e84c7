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
constexpr int LW_GMAX = 32;	    // workgroups per panel = two rounds of a header sweep (taller panels: getrf_panel2_kernel)
constexpr int LW_HDR_BYTES = 64;    // {label}, {a}, {s}, {l}: four 16-byte granule pairs
#ifndef LW_HDR_STRIDE
#define LW_HDR_STRIDE 320 // bytes between the header records of consecutive workgroups (64: LU N = 16384 ~0.8 ms slower, 1088 / 4160: no better -- profiles/r04_exp_lu_header_stride.txt)
#endif
constexpr int LW_ROW_BYTES = LW_W * 16;
constexpr int LW_NSH = 2; // header slots (column parity)
constexpr int LW_NSR = 4; // row-record slots (column modulo 4)
constexpr size_t LW_HDR_WS = (size_t) LW_NSH * LW_GMAX * LW_HDR_STRIDE;
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
	double res_a[2], res_s[2], res_l[2]; // what wavefront 0 found in the headers of the current column
	int res_p[2], res_gw[2], res_flags[2];
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

// One sweep over the G <= 16 RND headers of a column as ONE wavefront reads it: lane 4 t + k holds piece k ({label}, {a},
// {s}, {l}) of the records t, t + 16, ... -- 16 bytes per lane and round, four adjacent lanes per 64-byte record, so a round is
// 16 full cache lines (a lane per record and four loads per lane asked for every line four times: with all 256 wavefronts of
// a panel sweeping, three sweeps deep, the 2 KB of headers became a hot spot that slowed every memory access of the kernel --
// profiles/r04_lu_panel_phases_v2_all_waves_poll_3_deep.txt).  Only wavefront 0 of a workgroup sweeps; the others execute the
// SAME load instructions on one line of their own workgroup's record (`voff`) and never look at the result: with the number
// of memory instructions per step fixed, the compiler's counted waits (s_waitcnt vmcnt(N)) let a sweep stay in flight across
// the other waits of a step -- behind a wave-uniform branch every wait in its shadow became vmcnt(0), i.e. a full round trip
// (profiles/r04_lu_panel_phases_v3_one_sweeper_branchy.txt).
template <int RND> struct LwSweep { // (plain members, not an array: an indexed array inside the struct is kept in scratch memory)
	lw_u32x4 r0, r1;
};
template <int RND> static __device__ __forceinline__ void lw_sweep_issue(LwSweep<RND> &w, __amdgpu_buffer_rsrc_t hr, int G, int J, unsigned voff)
{
	const unsigned soff = (unsigned) ((J & 1) * LW_GMAX * LW_HDR_STRIDE);
	w.r0 = lw_load(hr, voff, soff);
	w.r1 = lw_load(hr, voff + 16u * LW_HDR_STRIDE, soff);
	static_assert(RND == 2, "two rounds of 16 records");
}
template <int RND> static __device__ __forceinline__ bool lw_sweep_ok(const LwSweep<RND> &w, unsigned tag, int G, int lane)
{
	const int t = lane >> 2;
	bool ok = (t >= G || lw_ok(w.r0, tag)) && (t + 16 >= G || lw_ok(w.r1, tag));
	return ok;
}
// "The old contents of this sweep are used here": placed right before a sweep's registers are re-issued, it keeps the register
// allocator from handing them to something else while a stale re-issued load may still be in flight (the compiler then waits
// for that load before the FIRST write to the register -- a full round trip right behind the successful check,
// profiles/r04_lu_panel_phases_v3_one_sweeper_branchy.txt); here the old load has long returned and the wait is free.
template <int RND> static __device__ __forceinline__ void lw_sweep_retire(const LwSweep<RND> &w)
{
	asm volatile("" ::"v"(w.r0), "v"(w.r1));
}
// value of quad lane Q (0..3) in every lane of the quad
template <int Q> static __device__ __forceinline__ unsigned lw_quad(unsigned v)
{
	return (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, Q * 0x55, 0xf, 0xf, true);
}

// The sweep of column J, by wavefront 0 of every workgroup: winner's label p, workgroup gw and header values; false if nobody
// has a candidate (then there is no row record either).  THREE sweeps are in flight when it starts -- A, B, C were issued by
// the previous step behind its row-record wait, in the middle of and behind its rank-1 update -- so a header is seen a
// fraction of a memory round trip after it lands instead of a whole one later.  A timeout sets `dead` (wave
// uniform): the kernel then runs to its end without waiting for anything and without storing anything -- no early exits, the
// eight step bodies stay one straight line of code.
template <int RND>
static __device__ __forceinline__ bool lw_sweep(__amdgpu_buffer_rsrc_t hr, int G, unsigned tag, int J, int lane, bool &dead, LwSweep<RND> &A,
						LwSweep<RND> &B, LwSweep<RND> &C, int &p, int &gw, double &a, double &s, double &l
#ifdef FH_LU_TIMING
						,
						unsigned long long &nsweeps
#endif
)
{
	const unsigned voff = (unsigned) ((lane >> 2) * LW_HDR_STRIDE + (lane & 3) * 16);
	// A, B, C are only ever written by the UNCONDITIONAL issues of lw_step: re-issued inside a branch here they would become
	// phi values, and the register copies at the join READ -- i.e. wait for -- the sweeps that are still in flight
	// (profiles/r04_lu_panel_phases_v3_one_sweeper_branchy.txt).  A sweep that comes back stale is simply dropped; if all three
	// are stale (a slow workgroup), a plain load-check loop on its own register pair takes over.
	// The sweep that succeeded is copied member by member inside its own branch: a select over A, B, C would read the two
	// that are still in flight as well, and a struct assignment sends all of them to scratch memory.
	lw_u32x4 q0, q1;
#define FH_LW_TAKE(S)                                                                                                    \
	do {                                                                                                             \
		q0 = S.r0;                                                                                               \
		q1 = S.r1;                                                                                               \
		/* (opaque to the optimizer: it would merge the three copies into ONE load through a pointer phi, which */   \
		/* keeps the sweeps in scratch memory) */                                                                  \
		asm volatile("" : "+v"(q0), "+v"(q1));                                                                   \
	} while (0)
#ifdef FH_LU_TIMING
	nsweeps += 1;
#endif
	if (dead || __all(lw_sweep_ok<RND>(A, tag, G, lane))) {
		FH_LW_TAKE(A);
	} else {
#ifdef FH_LU_TIMING
		nsweeps += 1;
#endif
		if (__all(lw_sweep_ok<RND>(B, tag, G, lane))) {
			FH_LW_TAKE(B);
		} else {
#ifdef FH_LU_TIMING
			nsweeps += 1;
#endif
			if (__all(lw_sweep_ok<RND>(C, tag, G, lane))) {
				FH_LW_TAKE(C);
			} else {
				LwSweep<RND> S;
				for (int spin = 0;; ++spin) {
#ifdef FH_LU_TIMING
					nsweeps += 1;
#endif
					lw_sweep_issue<RND>(S, hr, G, J, voff);
					if (__all(lw_sweep_ok<RND>(S, tag, G, lane)))
						break;
					if (spin >= LW_SPIN_MAX) {
						dead = true;
						break;
					}
					__builtin_amdgcn_s_sleep(1);
				}
				FH_LW_TAKE(S);
			}
		}
	}
#undef FH_LW_TAKE
	// per quad: the best of its (up to RND) records; every lane keeps ITS piece of that record
	double bcv = -1.0;
	int blab = INT_MAX, bg = 0;
	unsigned phi = 0u, plo = 0u;
	const int t = lane >> 2;
#define FH_LW_ROUND(q, off)                                                                                              \
	if (t + off < G) {                                                                                               \
		const int lb = (int) lw_quad<0>(q.y);                                                                    \
		const double av = __hiloint2double((int) lw_quad<1>(q.y), (int) lw_quad<1>(q.w)), fa = fabs(av);         \
		/* a published candidate is a real one (|a| > 0) or the diagonal row of a zero / NaN-only column (factor.rs:35-43) */ \
		const double cv = lb == INT_MAX ? -1.0 : (fa > 0.0 ? fa : 0.0);                                          \
		if (better(cv, lb, bcv, blab)) {                                                                         \
			bcv = cv;                                                                                        \
			blab = lb;                                                                                       \
			bg = t + off;                                                                                    \
			phi = q.y;                                                                                       \
			plo = q.w;                                                                                       \
		}                                                                                                        \
	}
	FH_LW_ROUND(q0, 0)
	FH_LW_ROUND(q1, 16)
#undef FH_LW_ROUND
	double wv;
	int wl;
	lw_argmax<6>(bcv, blab, wv, wl);
	// (no candidate at all: cannot happen for J < m, the row labelled J is always one; the padded steps J >= m keep the diagonal)
	const bool any = wv >= 0.0;
	const unsigned long long bal = (unsigned long long) __ballot(blab == wl && bcv == wv);
	const int qb = any && bal != 0ull ? (__builtin_amdgcn_readfirstlane((int) __ffsll(bal) - 1) & ~3) : 0;
	p = any ? wl : J;
	gw = any ? __builtin_amdgcn_readlane(bg, qb) : 0;
	a = __hiloint2double(__builtin_amdgcn_readlane((int) phi, qb + 1), __builtin_amdgcn_readlane((int) plo, qb + 1));
	s = __hiloint2double(__builtin_amdgcn_readlane((int) phi, qb + 2), __builtin_amdgcn_readlane((int) plo, qb + 2));
	l = __hiloint2double(__builtin_amdgcn_readlane((int) phi, qb + 3), __builtin_amdgcn_readlane((int) plo, qb + 3));
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
			lw_store(q, hr, (unsigned) (lane * 16), (unsigned) ((par * LW_GMAX + g) * LW_HDR_STRIDE));
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
template <typename T, int RPT, int NW, int RND, int JJ>
static __device__ __forceinline__ void lw_step(const WPanelArgs<T> &a, __amdgpu_buffer_rsrc_t hr, __amdgpu_buffer_rsrc_t rr, T (&x)[RPT][LW_W],
					       int (&lab)[RPT], T &uprev, WPanelShared<NW> &sh, int grp, int G, int steps, int nsteps, bool &dead, unsigned svoff,
					       LwSweep<RND> &A, LwSweep<RND> &B, LwSweep<RND> &C
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
	// ---- 1. the headers of column J: wavefront 0 sweeps, the others wait for its result at the barrier
	const int wave0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
	const int rpar = J & 1;
	if (wave0) {
		int p0, gw0;
		double a0, s0, l0;
#ifdef FH_LU_TIMING
		const bool any0 = lw_sweep<RND>(hr, G, tag, J, lane, dead, A, B, C, p0, gw0, a0, s0, l0, tk[6]);
#else
		const bool any0 = lw_sweep<RND>(hr, G, tag, J, lane, dead, A, B, C, p0, gw0, a0, s0, l0);
#endif
		if (lane == 0) {
			sh.res_p[rpar] = p0;
			sh.res_gw[rpar] = gw0;
			sh.res_flags[rpar] = (any0 ? 1 : 0) | (dead ? 2 : 0);
			sh.res_a[rpar] = a0;
			sh.res_s[rpar] = s0;
			sh.res_l[rpar] = l0;
		}
	}
	LW_TICK(0);
	__syncthreads();
	const int p = __builtin_amdgcn_readfirstlane(sh.res_p[rpar]), gw = __builtin_amdgcn_readfirstlane(sh.res_gw[rpar]);
	const int rflags = __builtin_amdgcn_readfirstlane(sh.res_flags[rpar]);
	const bool any = (rflags & 1) != 0;
	dead = dead || (rflags & 2) != 0;
	const double da = sh.res_a[rpar], ds = sh.res_s[rpar], dl = sh.res_l[rpar];
	LW_TICK(7);
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
		l[i] = (T) 0;
		if (lab[i] > J) {
			l[i] = x[i][JJ] * inv;
			x[i][JJ] = l[i];
			if (JJ + 1 < lim)
				x[i][JJ + 1] = fh_fma(l[i], -uJ1, x[i][JJ + 1]);
		}
	}
	LW_TICK(1);
	// ---- 4. candidate of column J + 1, header and row record on their way; the first sweep of the next column follows them
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
	// the first sweep of column J + 1 (every wavefront issues it, see LwSweep); the other two follow inside / behind the update
	lw_sweep_retire<RND>(A);
	lw_sweep_issue<RND>(A, hr, G, J + 1, svoff);
	LW_TICK(3);
	// ---- 6. rank-1 update of the columns >= J + 2 (positions in blocks of 8, a block takes part while it holds unfinished
	//         columns; the multipliers of the pivot row reach the FMAs through the scalar unit, as in getrf_panel2_kernel;
	//         rows labelled <= J stay bitwise untouched: l * u could be NaN for an infinite u)
#pragma unroll
	for (int cb = 0; cb < LW_W / 8; ++cb) {
		if (cb == 4) {
			lw_sweep_retire<RND>(B);
			lw_sweep_issue<RND>(B, hr, G, J + 1, svoff);
		}
		if (cb * 8 + 7 > JJ + 1 && cb * 8 < lim) {
			T u[8];
#pragma unroll
			for (int k = 0; k < 8; ++k)
				u[k] = lane_bcast(ucur, cb * 8 + k);
#pragma unroll
			for (int i = 0; i < RPT; ++i) {
				if (lab[i] > J) {
#pragma unroll
					for (int k = 0; k < 8; ++k)
						if (cb * 8 + k > JJ + 1)
							x[i][cb * 8 + k] = fh_fma(l[i], -u[k], x[i][cb * 8 + k]);
				}
			}
		}
	}
	lw_sweep_retire<RND>(C);
	lw_sweep_issue<RND>(C, hr, G, J + 1, svoff);
	uprev = ucur;
	LW_TICK(4);
}

// grid = G <= 16 RND workgroups of NW wavefronts, all resident; wavefront v of workgroup g owns the rows [(g NW + v) 64 RPT, +64 RPT)
template <typename T, int RPT, int NW, int RND> __global__ __launch_bounds__(NW * 64) void getrf_wpanel_kernel(const WPanelArgs<T> a)
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
	// wavefront 0 sweeps all headers; the others issue the same loads on their own workgroup's record and ignore them (LwSweep)
	const unsigned svoff = wave == 0 ? (unsigned) ((lane >> 2) * LW_HDR_STRIDE + (lane & 3) * 16) : (unsigned) (blockIdx.x * LW_HDR_STRIDE + (lane & 3) * 16);
	LwSweep<RND> A, B, C;
	lw_sweep_issue<RND>(A, hr, G, 0, svoff);
	lw_sweep_issue<RND>(B, hr, G, 0, svoff);
	lw_sweep_issue<RND>(C, hr, G, 0, svoff);
	int rot = 0;
	for (int grp = 0; grp * 8 < steps; ++grp) {
#ifdef FH_LU_TIMING
		const unsigned long long t_grp0 = __builtin_amdgcn_s_memtime();
#endif
		lw_step<T, RPT, NW, RND, 0>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead, svoff, A, B, C LW_TARGS);
		lw_step<T, RPT, NW, RND, 1>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead, svoff, A, B, C LW_TARGS);
		lw_step<T, RPT, NW, RND, 2>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead, svoff, A, B, C LW_TARGS);
		lw_step<T, RPT, NW, RND, 3>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead, svoff, A, B, C LW_TARGS);
		lw_step<T, RPT, NW, RND, 4>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead, svoff, A, B, C LW_TARGS);
		lw_step<T, RPT, NW, RND, 5>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead, svoff, A, B, C LW_TARGS);
		lw_step<T, RPT, NW, RND, 6>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead, svoff, A, B, C LW_TARGS);
		lw_step<T, RPT, NW, RND, 7>(a, hr, rr, x, lab, uprev, sh, grp, G, steps, nsteps, dead, svoff, A, B, C LW_TARGS);
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
#ifdef FH_LU_TIMING
		if (blockIdx.x == 0 && tid == 0 && a.phase && grp < 7)
			atomicAdd(a.phase + 9 + grp, __builtin_amdgcn_s_memtime() - t_grp0); // per group of 8 columns: is the first pass over the code slower?
#endif
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
