// Shared host-side infrastructure of libfaer_hip.so (gfx950 only, no compatibility layers).
//
//  * MatV<T>: the strided view every driver works on.  It is faer's MatView
//    {ptr, nrows, ncols, row_stride, col_stride} (faer/src/mat/mod.rs:7-13) with strides in elements.
//  * Ctx: per-thread device / stream / scratch state.  faer's entry points may be called from any
//    rayon worker => everything mutable is thread_local.
//  * Staged<T>: host<->device staging so that a host-resident faer::Mat works unchanged.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <atomic>
#include <vector>

#include "../../include/faer_hip.h"

namespace fh {

[[noreturn]] inline void die(const char *msg, const char *file, int line)
{
	fprintf(stderr, "faer_hip: fatal: %s (%s:%d)\n", msg, file, line);
	fflush(stderr);
	abort();
}
#define FH_CHECK(cond, msg)                                                                                             \
	do {                                                                                                            \
		if (!(cond))                                                                                            \
			::fh::die(msg, __FILE__, __LINE__);                                                             \
	} while (0)
#define FH_HIP(expr)                                                                                                    \
	do {                                                                                                            \
		hipError_t e_ = (expr);                                                                                 \
		if (e_ != hipSuccess)                                                                                   \
			::fh::die(hipGetErrorString(e_), __FILE__, __LINE__);                                           \
	} while (0)

typedef long idx_t;

// typed fused multiply-add: __builtin_fma is the DOUBLE builtin -- called with floats it converts to fp64 and back
// (three extra conversions per operation and the quarter-rate fp64 pipe); templates must use this overload pair
static __device__ __forceinline__ double fh_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
static __device__ __forceinline__ float fh_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T> struct MatV {
	T *p;
	idx_t nrows, ncols, rs, cs;

	MatV sub(idx_t r0, idx_t c0, idx_t nr, idx_t nc) const { return MatV{p + r0 * rs + c0 * cs, nr, nc, rs, cs}; }
	MatV t() const { return MatV{p, ncols, nrows, cs, rs}; }
	MatV rev_rows() const { return MatV{nrows > 0 ? p + (nrows - 1) * rs : p, nrows, ncols, -rs, cs}; }
	MatV rev_cols() const { return MatV{ncols > 0 ? p + (ncols - 1) * cs : p, nrows, ncols, rs, -cs}; }
	MatV<const T> c() const { return MatV<const T>{p, nrows, ncols, rs, cs}; }
};

// ------------------------------------------------------------------------------------------------
// per-thread context
// ------------------------------------------------------------------------------------------------
struct Ctx {
	int device = -1;
	hipStream_t stream = nullptr;
	int gemm_variant = 0;
	// scratch pool: simple stack of device buffers reused across calls of this thread
	struct Buf {
		void *p;
		size_t bytes;
		bool used;
		hipStream_t owner; // stream whose work last used the buffer: reuse is stream ordered only on that stream
	};
	std::vector<Buf> pool;
	int *status = nullptr; // 16 ints of device status words
	int *host_ints = nullptr; // 16 pinned host ints: read-back target of the drivers that return a status per call
	int *pinned_ints();
	// Look-ahead execution (factorizations): two internal streams on DISJOINT sets of CUs
	// (hipExtStreamCreateWithCUMask): `la_bulk` runs the trailing-matrix GEMMs on most of the chip, `la_panel`
	// the latency-bound diagonal-block / panel work on a few reserved CUs, so neither queue can starve the other.
	hipStream_t la_bulk = nullptr, la_panel = nullptr;
	int la_state = 0; // 0: not tried, 1: available, -1: unavailable
	int la_panel_cus = 32;
	std::vector<hipEvent_t> la_events;
	size_t la_next_event = 0;
	bool lookahead_streams(); // creates the streams on first use; false if the runtime refuses
	// Plain side stream + events of the one-pass QR (tsqr.hip): ONE stream for both of its roles, created before the two
	// CU-masked streams above.  The runtime maps streams onto 4 hardware queues by default; as the 4th and 5th stream of the
	// process the QR's side streams made every kernel of the QR -- on the caller's stream too -- 1.3-6x slower (bench.py:
	// QR after an LLT in the same process 2.78 ms against 2.04; profiles/r03_qr_stream_order.txt).
	hipStream_t qr_side[2] = {nullptr, nullptr};
	hipEvent_t qr_ev[4] = {nullptr, nullptr, nullptr, nullptr};
	void qr_side_streams();
	hipEvent_t next_event();  // timing-disabled events, recycled per factorization (reset_events)
	void reset_events() { la_next_event = 0; }
	int ncu = 0;
	int stream_cus(); // compute units the current stream can occupy (the look-ahead streams are CU masked)
	// Kernel-class profile (faer_hip_prof_begin / faer_hip_prof_end, bench.py's per-workload `roofline`): while on, every launch
	// of a dominant kernel class is bracketed by two timing events on ITS stream; prof_end adds the spans up per class.
	// Classes: 0 big-tile MFMA products (units: flop), 1 LU panel kernel (columns), 2 one-pass QR update (algorithmic bytes),
	// 3 one-pass QR Gram (bytes), 4 one-pass QR panel kernel (launches), 5 Cholesky leaf (columns).
	struct ProfSpan {
		hipEvent_t a, b;
		int cls;
		double units;
		long d[4] = {0, 0, 0, 0}; // class 0: m, n, k of the product, 1 + tri_skip for a Lower destination (0: Full)
		int sid = 0;		  // 0 caller's stream, 1 bulk, 2 panel, 3 side
	};
	static constexpr int PROF_CLASSES = 6;
	bool prof_on = false;
	std::vector<ProfSpan> prof_spans;
	std::vector<hipEvent_t> prof_pool;
	hipEvent_t prof_event();

	void ensure_device();
	void *alloc(size_t bytes); // returns a device buffer valid until release()
	void release(void *p);
	void sync() { FH_HIP(hipStreamSynchronize(stream)); }
	void set_stream(hipStream_t s); // the caller's stream (faer_hip_set_stream); internal code switches `stream` directly
	// after a point where every internal stream has been joined AND the caller's stream synchronised: free
	// buffers may be handed to any stream again
	void quiesce();
};
Ctx &ctx();
void debug_stream_xcc(int which, int nblocks, unsigned *out_host); // which: 0 caller's stream, 1 bulk, 2 panel
void prof_collect(double *out, double *spans = nullptr, size_t cap = 0, size_t *nspans = nullptr); // Ctx::PROF_CLASSES x {ms, launches, units} of the spans recorded since prof_on; optionally one record of 8 doubles per span (ctx.hip)
double xwg_hop_us(int iters); // idle-chip hand-off latency between two workgroups, microseconds (ctx.hip)
void ctx_shutdown(); // releases the calling thread's look-ahead streams / events; safe without a device

// stream `s` waits for event `e`
inline void stream_wait(hipStream_t s, hipEvent_t e) { FH_HIP(hipStreamWaitEvent(s, e, 0)); }

// RAII: run the enclosed launches on another stream
struct StreamScope {
	hipStream_t saved;
	explicit StreamScope(hipStream_t s) : saved(ctx().stream) { ctx().stream = s; }
	~StreamScope() { ctx().stream = saved; }
	StreamScope(const StreamScope &) = delete;
	StreamScope &operator=(const StreamScope &) = delete;
};

// RAII: one profiled launch (no-op unless Ctx::prof_on)
struct ProfScope {
	Ctx::ProfSpan sp;
	bool on;
	ProfScope(int cls, double units) : on(ctx().prof_on && cls >= 0)
	{
		if (!on)
			return;
		sp.cls = cls;
		sp.units = units;
		{
			Ctx &c = ctx();
			sp.sid = c.stream == c.la_bulk ? 1 : (c.stream == c.la_panel ? 2 : (c.qr_side[0] && c.stream == c.qr_side[0] ? 3 : 0));
		}
		sp.a = ctx().prof_event();
		sp.b = ctx().prof_event();
		FH_HIP(hipEventRecord(sp.a, ctx().stream));
	}
	~ProfScope() // may run while an exception unwinds: never throws, a span whose end cannot be recorded is dropped
	{
		if (!on)
			return;
		if (hipEventRecord(sp.b, ctx().stream) != hipSuccess) {
			(void) hipGetLastError();
			sp.cls = -1;
		}
		try {
			ctx().prof_spans.push_back(sp);
		} catch (...) {
		}
	}
	ProfScope(const ProfScope &) = delete;
	ProfScope &operator=(const ProfScope &) = delete;
};

// RAII scratch
struct Scratch {
	void *p;
	explicit Scratch(size_t bytes) : p(ctx().alloc(bytes)) {}
	~Scratch() { ctx().release(p); }
	Scratch(const Scratch &) = delete;
	Scratch &operator=(const Scratch &) = delete;
	template <typename T> T *as() const { return static_cast<T *>(p); }
};

bool is_device_ptr(const void *p);

// Span of memory (in elements, relative to .p) touched by a strided view.
template <typename T> inline void view_span(const MatV<T> &v, idx_t &lo, idx_t &hi)
{
	lo = 0;
	hi = 0;
	if (v.nrows == 0 || v.ncols == 0)
		return;
	idx_t r = (v.nrows - 1) * v.rs, c = (v.ncols - 1) * v.cs;
	lo = (r < 0 ? r : 0) + (c < 0 ? c : 0);
	hi = (r > 0 ? r : 0) + (c > 0 ? c : 0);
}

// Host->device staging of one operand.  If the view already lives in device memory it is used in place;
// otherwise EXACTLY the elements of the view are copied into a dense device buffer (and copied back on
// destruction when `writeback` is set).  The gaps of a strided host view -- the parent matrix's entries between
// the columns of a `submatrix_mut`, which other rayon workers may be writing at the same time -- are never read
// and never written (round 1 staged the whole contiguous span and wrote it back, gaps included).
//   * unit row stride, positive column stride (a faer::Mat or a block of one): one hipMemcpy2DAsync, column major;
//   * unit column stride, positive row stride (a transposed view): the same, row major;
//   * anything else (negative / non-unit strides): packed through a temporary host buffer.
template <typename T> struct Staged {
	typedef typename std::remove_const<T>::type U;
	MatV<T> dev;
	MatV<T> host;
	void *buf = nullptr;
	U *pack = nullptr; // host-side packing buffer (mode 3)
	int mode = 0;	   // 0: device operand, 1: column-major 2-D copy, 2: row-major 2-D copy, 3: host packed (column major)
	bool writeback = false;

	// host pitch (elements) of the 2-D copy; a single column / row may carry any outer stride (even 0)
	idx_t hpitch() const
	{
		if (mode == 1)
			return host.ncols == 1 ? host.nrows : host.cs;
		return host.nrows == 1 ? host.ncols : host.rs;
	}

	Staged(MatV<T> v, bool copy_in, bool writeback_)
	{
		dev = v;
		host = v;
		if (v.nrows == 0 || v.ncols == 0 || is_device_ptr(v.p))
			return;
		const size_t bytes = (size_t) v.nrows * (size_t) v.ncols * sizeof(T);
		buf = ctx().alloc(bytes);
		writeback = writeback_;
		const idx_t pitch_limit = (idx_t) 1 << 30; // bytes; beyond it (never for a real Mat) fall back to packing
		if (v.rs == 1 && (v.cs >= v.nrows || v.ncols == 1) && v.cs * (idx_t) sizeof(T) < pitch_limit) {
			mode = 1;
			dev = MatV<T>{static_cast<T *>(buf), v.nrows, v.ncols, 1, v.nrows};
		} else if (v.cs == 1 && (v.rs >= v.ncols || v.nrows == 1) && v.rs * (idx_t) sizeof(T) < pitch_limit) {
			mode = 2;
			dev = MatV<T>{static_cast<T *>(buf), v.nrows, v.ncols, v.ncols, 1};
		} else {
			mode = 3;
			dev = MatV<T>{static_cast<T *>(buf), v.nrows, v.ncols, 1, v.nrows};
			pack = static_cast<U *>(malloc(bytes));
			FH_CHECK(pack != nullptr, "staging: out of host memory");
		}
		if (!copy_in)
			return;
		if (mode == 1) {
			FH_HIP(hipMemcpy2DAsync(buf, (size_t) v.nrows * sizeof(T), (const void *) v.p, (size_t) hpitch() * sizeof(T),
						(size_t) v.nrows * sizeof(T), (size_t) v.ncols, hipMemcpyHostToDevice, ctx().stream));
		} else if (mode == 2) {
			FH_HIP(hipMemcpy2DAsync(buf, (size_t) v.ncols * sizeof(T), (const void *) v.p, (size_t) hpitch() * sizeof(T),
						(size_t) v.ncols * sizeof(T), (size_t) v.nrows, hipMemcpyHostToDevice, ctx().stream));
		} else {
			for (idx_t j = 0; j < v.ncols; ++j)
				for (idx_t i = 0; i < v.nrows; ++i)
					pack[j * v.nrows + i] = v.p[i * v.rs + j * v.cs];
			FH_HIP(hipMemcpyAsync(buf, pack, bytes, hipMemcpyHostToDevice, ctx().stream));
			FH_HIP(hipStreamSynchronize(ctx().stream)); // `pack` is reused by the write-back
		}
	}
	~Staged()
	{
		if (!buf)
			return;
		if (writeback) {
			U *hp = const_cast<U *>(host.p);
			if (mode == 1) {
				FH_HIP(hipMemcpy2DAsync((void *) hp, (size_t) hpitch() * sizeof(T), buf, (size_t) host.nrows * sizeof(T),
							(size_t) host.nrows * sizeof(T), (size_t) host.ncols, hipMemcpyDeviceToHost, ctx().stream));
				FH_HIP(hipStreamSynchronize(ctx().stream));
			} else if (mode == 2) {
				FH_HIP(hipMemcpy2DAsync((void *) hp, (size_t) hpitch() * sizeof(T), buf, (size_t) host.ncols * sizeof(T),
							(size_t) host.ncols * sizeof(T), (size_t) host.nrows, hipMemcpyDeviceToHost, ctx().stream));
				FH_HIP(hipStreamSynchronize(ctx().stream));
			} else {
				FH_HIP(hipMemcpyAsync(pack, buf, (size_t) host.nrows * (size_t) host.ncols * sizeof(T), hipMemcpyDeviceToHost,
						      ctx().stream));
				FH_HIP(hipStreamSynchronize(ctx().stream));
				for (idx_t j = 0; j < host.ncols; ++j)
					for (idx_t i = 0; i < host.nrows; ++i)
						hp[i * host.rs + j * host.cs] = pack[j * host.nrows + i];
			}
		}
		free(pack);
		ctx().release(buf);
	}
	Staged(const Staged &) = delete;
	Staged &operator=(const Staged &) = delete;
};

// ------------------------------------------------------------------------------------------------
// device-level drivers (operands are device memory); defined in the .hip files
// ------------------------------------------------------------------------------------------------
enum DstKind { DST_FULL = 0, DST_LOWER = 1, DST_UPPER = 2 };

template <typename T> struct GemmExtra {
	const void *row_idx = nullptr; // device memory, itype-wide indices
	const void *col_idx = nullptr;
	int idx64 = 1;
	const T *diag = nullptr; // device memory
	idx_t diag_stride = 0;
	int a_struct = 0, b_struct = 0; // FaerBlock codes of the operands (triangular products)
	bool dst_strict = false;	// with DST_LOWER / DST_UPPER: leave the diagonal untouched
	// in-place product dst = alpha * A * B with dst aliasing an operand whose other factor is a square
	// K x K (K <= 128) matrix: 1 = B aliases dst (dst = S * dst), 2 = A aliases dst (dst = dst * S).
	// (The value refers to the operands as passed; gemm_dev swaps it when it transposes the problem.)
	int inplace = 0;
	// dense-kernel shortcuts for the factorization drivers (plain products only):
	//   k_trim 1: rhs(k, n) is zero for k > n (upper triangular rhs with an explicitly zeroed lower part), 2: lhs(m, k)
	//   is zero for k > m -- each tile stops its K loop at the end of its diagonal block instead of multiplying zeros;
	//   tri_skip (DST_LOWER, square): the first `tri_skip` rows of the lower triangle (a multiple of 128) are left
	//   untouched, i.e. one launch covers "block column below the leading block + remaining lower square".
	//   stair_nb / stair_gap (DST_LOWER, any shape): the "diagonal" is a staircase -- column n of dst stands for column
	//   n + (n / stair_nb) * stair_gap of the matrix it is a part of (the owned block columns of a 1-D block-cyclic
	//   partition next to each other: dist_llt.h), entries above it are left untouched.  stair_row0: row 0 of dst is row
	//   stair_row0 of that picture (a row chunk of the staircase: dst(i, n) is written iff i + stair_row0 >= that column).
	int k_trim = 0;
	idx_t tri_skip = 0;
	idx_t stair_nb = 0, stair_gap = 0, stair_row0 = 0;
	// Lending idle CUs to a big product (128 x 128 pipelined tile): `ticket` = 8 zeroed device ints; the launch's tiles are then
	// handed out through per-XCD counters (gemm.hip, GemmArgs::ticket).  helper_wgs == 0: the main launch; > 0: a HELPER launch
	// of that many workgroups -- the same call with the same operands on another stream -- which takes tiles only while more
	// than helper_margin remain in an XCD's share.  The caller orders consumers of dst behind BOTH launches.
	int *ticket = nullptr;
	int helper_wgs = 0, helper_margin = 0;
	// short-wide / tall-narrow FULL output with a deep K (the V^H A product of a QR block application, 128 x n with K = rows): 128 x 128
	// tiles with more K slices instead of the 64 x 64 tiles the tile-count rule picks (square QR N = 8192: -3 %, tools/gpu_qr_square_gemm_ab.py)
	bool prefer_big_tiles = false;
};

// dst(kind) <- [dst +] alpha * A * diag * B        (gemm.hip)
template <typename T>
void gemm_dev(MatV<T> C, DstKind kind, bool add, MatV<const T> A, MatV<const T> B, T alpha,
	      const GemmExtra<T> *extra = nullptr);
// level-2 shapes (gemv.hip): matrix-vector product and rank-1 update, HBM streams outside the MFMA kernel
template <typename T> bool gemv_dev(idx_t m, idx_t k, MatV<const T> A, const T *x, idx_t xs, T *y, idx_t ys, T alpha, bool add);
template <typename T> void rank1_dev(MatV<T> C, bool add, const T *a, idx_t as, const T *b, idx_t bs, T alpha);
// extras.hip / trsm.hip: triangular inverse (triangular_inverse.rs) and helpers of the reconstruct / inverse entry points
template <typename T> void tri_invert_lower_dev(MatV<T> dst, MatV<const T> src, bool unit);
template <typename T> void ldlt_scale_lower_dev(MatV<T> out, MatV<const T> L, const T *d, idx_t ds);
template <typename T> void ldlt_inverse_prepare_dev(MatV<T> W, const T *d, idx_t ds);
template <typename T> void zero_then_upper_dev(MatV<T> out, const MatV<const T> *R);
// fplu.hip: LU with full pivoting (perm arrays are host memory); returns the transposition count
void fplu_debug_inplace(int on); // fplu.hip: 1 = the in-place two-launch path (A/B tests)
template <typename T> long full_piv_lu_dev(MatV<T> A, idx_t *row_perm, idx_t *row_perm_inv, idx_t *col_perm, idx_t *col_perm_inv);
// qr.hip: QR with column pivoting (perm arrays are host memory); returns the transposition count
template <typename T> long colpiv_qr_dev(MatV<T> A, MatV<T> H, idx_t *col_perm, idx_t *col_perm_inv);
// pure host planning logic, exported for the CPU tests (faer_hip_debug_*)
std::vector<idx_t> llt_plan(idx_t n, idx_t tail_rows, idx_t nb2);
int lu_leaf_width(idx_t m, int elem_bytes, int resident_workgroups);
bool dist_two_streams_ok(idx_t panel_rows, int elem_bytes, int panel_cus, int all_cus);
void lu_force_general(int on); // debug: every LU leaf on the non-cooperative path
extern std::atomic<int> g_lend_cus; // faer_hip_debug_lend_cus: the look-ahead drivers lend the panel stream's idle CUs to their big products (ctx.hip; A/B, tests)
void lu_debug_plan(long nb2_from, long pipe_from, long la_min); // debug: switch-over points of the look-ahead LU driver (0 = default)
void lu_lend_copy(const void *device_copy, idx_t nrows, idx_t ncols, int elem_bytes); // the calling thread's next LU may restore A from it after an exchange timeout (getrf.hip)
bool rccl_is_builtin_wait(FaerHipWaitFn fn); // rccl_transport.hip: is this the built-in transport's wait (takes any stream)
bool loop_is_builtin_wait(FaerHipWaitFn fn); // loop_transport.hip: the loop-back transport of the tests (ranks = threads on one GPU)
void level2_debug_force_memory_bodies(int on); // debug: tridiag / bidiag / Hessenberg vector kernels never keep their columns in registers
void tsqr_debug_fused(int on); // debug: 0 = the one-pass QR runs update and Gram as separate launches (rounds 3-5), 1 = fused with look-ahead (default)
void tsqr_debug_f64(int on); // debug: 0 = fp64 matrices never take the one-pass QR path
void tsqr_debug_shape_rule(long min_rows, long min_aspect); // debug: shape rule of the whole-matrix one-pass QR path (0 = default)
void tsqr_debug_panels(int on); // debug: 0 = the classic QR path never factors a panel on the one-pass path
long qr_last_one_pass_columns(); // debug: columns the one-pass QR path completed in this thread's last factorization (-1: not taken)
// tall-skinny shapes (skinny.hip): streaming kernels; false if the shape / strides do not qualify
template <typename T> bool skinny_dev(MatV<T> C, bool add, MatV<const T> A, MatV<const T> B, T alpha);
// C <- [C +] alpha * sum_z ws[z] (slices of nrows x ncols, column major), fixed summation order (gemm.hip)
template <typename T> void splitk_reduce_dev(MatV<T> C, const T *ws, int splits, T alpha, bool add);
// average ms per launch of the dense kernel (hipEvents on the ctx stream)
template <typename T> double gemm_time_ms(MatV<T> C, MatV<const T> A, MatV<const T> B, int iters);
double mfma_peak_tflops(bool f64, int iters);

// dst(structure) <- [dst +] alpha * A(structure) * B(structure)   (trmm.hip)
template <typename T>
void matmul_triangular_dev(MatV<T> C, int c_s, bool add, MatV<const T> A, int a_s, MatV<const T> B, int b_s, T alpha);

// X <- op(T)^-1 X, T lower triangular n x n, X n x k (trsm.hip); upper handled by reversal
template <typename T> void trsm_lower_dev(MatV<const T> L, bool unit, MatV<T> X);
template <typename T> void trsm_upper_dev(MatV<const T> U, bool unit, MatV<T> X);
// same as trsm_lower_dev with the packed images (trsm_pack.h) of L's 128 x 128 diagonal blocks already in W
template <typename T> void trsm_lower_pre_dev(MatV<const T> L, MatV<T> X, const T *W);
template <typename T> void trsm_pack_dev(MatV<const T> L, bool unit, T *W);
void trsm_dump_timing(); // timing build only (no-op otherwise)
void lu_dump_timing();   // timing build only: per-phase ticks of the LU panel kernel (getrf.hip)
void gemm_dump_timing(); // timing build only: per-phase ticks of the pipelined GEMM tile (gemm.hip)

// in-place lower Cholesky; returns >=0 regularization count or -(index+1)   (potrf.hip)
template <typename T> long potrf_lower_dev(MatV<T> A, T reg_delta, T reg_eps);
template <typename T> void potrf_panel_dev(MatV<T> P, T reg_delta, T reg_eps, int *status_dev, idx_t offset);

// in-place unit-lower L D L^T without pivoting (potrf.hip); L strictly below the diagonal, D on it; `signs_host`: n int8
// expected pivot signs or NULL; returns >= 0 regularization count or -(index + 1) of the zero pivot
template <typename T> long sytrf_lower_dev(MatV<T> A, T reg_delta, T reg_eps, const signed char *signs_host);

// partial pivot LU; perm/perm_inv are HOST arrays of idx_t (m entries)   (getrf.hip)
template <typename T> long getrf_dev(MatV<T> A, idx_t *perm, idx_t *perm_inv);

// pieces of the LU driver used by the distributed driver (dist.hip): in-place factorization of an m x w panel
// (w <= m) with the pivots left on the device (piv_dev[j] = row swapped with row j, relative to the panel), and
// the application of such a transposition list to the rows of another block
template <typename T> void getrf_panel_dev(MatV<T> P, int *piv_dev, int *status_dev = nullptr);
template <typename T> void laswp_rows_dev(MatV<T> B, const int *piv_dev, int nt);
int laswp_list_max(); // most interchanges one composed list holds
void laswp_compose_rows_dev(const int *piv_dev, int nt, int *list_4nt); // net permutation of (j <-> piv[j]), j < nt, composed once ...
template <typename T> void laswp_list_rows_dev(MatV<T> B, const int *list_4nt, int nt); // ... and applied to any number of column ranges

// Householder QR without pivoting (qr.hip); H is block_size x min(m,n) (device). returns rank
template <typename T> long geqrf_dev(MatV<T> A, MatV<T> H, idx_t blocking_threshold);
// evd/tridiag.rs:274: A (self-adjoint, lower triangle used) -> T + reflectors, H: block Householder factors (qr.hip)
template <typename T> void tridiag_dev(MatV<T> A, MatV<T> H);
// svd/bidiag.rs:47: A -> upper bidiagonal + reflectors, Hl / Hr: block Householder factors (qr.hip)
template <typename T> void bidiag_dev(MatV<T> A, MatV<T> Hl, MatV<T> Hr);
// evd/hessenberg.rs:549: A -> upper Hessenberg + reflectors, H: block Householder factors (qr.hip)
template <typename T> void hessenberg_dev(MatV<T> A, MatV<T> H);
template <typename T>
void apply_householder_sequence_left_dev(MatV<const T> V, MatV<const T> H, MatV<T> M, bool transpose);

// small utility kernels (util.hip)
template <typename T> void fill_dev(MatV<T> A, DstKind kind, T value);
template <typename T> void copy_dev(MatV<T> dst, MatV<const T> src);
template <typename T> void gather_rows_dev(MatV<T> dst, MatV<const T> src, const idx_t *perm_dev); // dst[i,:] = src[perm[i],:]
template <typename T> void scale_rows_recip_dev(MatV<T> X, const T *d, idx_t ds);		       // X[i,:] *= 1 / d[i]

} // namespace fh
