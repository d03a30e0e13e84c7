// Multi-GPU entry points: 1-D block-cyclic columns, one process per GPU, RCCL (or any transport) through the
// caller's broadcast callback -- SURVEY.md section 8e.  The orchestration lives in dist_lu.h (backend template);
// this file is its device backend: every operation is one of the library's own HIP drivers on the calling
// thread's stream, the broadcast is handed a DEVICE buffer.
#include "common.h"
#include <memory>
#include "dist_llt.h"
#include "dist_lu.h"
using namespace fh;

namespace {

// per-thread record of the last distributed factorization (faer_hip_dist_last_stats): device time between its first and last
// launch on the caller's stream, and the time of the panel factorizations THIS rank owned (timing events on the stream they
// ran on) -- what bench.py prints per rank so that a scaling curve can be read against the model of DESIGN.md section 4
struct DistStats {
	double total_ms = 0, panel_ms = 0;
	int panels = 0;
};
thread_local DistStats g_dist_stats;

struct PhaseTimer {
	std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
	void begin()
	{
		hipEvent_t a, b;
		FH_HIP(hipEventCreate(&a));
		FH_HIP(hipEventCreate(&b));
		FH_HIP(hipEventRecord(a, ctx().stream));
		ev.emplace_back(a, b);
	}
	void end() { FH_HIP(hipEventRecord(ev.back().second, ctx().stream)); }
	double harvest() // after the run has been synchronised
	{
		double ms = 0;
		for (auto &p : ev) {
			float t = 0;
			if (hipEventElapsedTime(&t, p.first, p.second) == hipSuccess)
				ms += t;
			(void) hipEventDestroy(p.first);
			(void) hipEventDestroy(p.second);
		}
		ev.clear();
		return ms;
	}
};

// dst(c, k) = P(c + (c / nb) * gap, k): the rows of the received panel that belong to the owned block columns, contiguous
// (the rhs of the staircase product of dist_llt.h); dst column major with leading dimension ld
template <typename T> __global__ void gather_stair_kernel(const T *P, long prs, long pcs, long ncols, long w, long nb, long gap, T *dst, long ld)
{
	const long c = (long) blockIdx.x * blockDim.x + threadIdx.x;
	const long k = blockIdx.y;
	if (c < ncols)
		dst[k * ld + c] = P[(c + (c / nb) * gap) * prs + k * pcs];
}

template <typename S> struct DeviceBackend {
	typedef S T;
	PhaseTimer t_panel, t_total;
	struct View {
		T *p;
		long nrows, ncols, rs, cs;
	};
	FaerHipComm comm;

	static MatV<T> mv(View v) { return MatV<T>{v.p, v.nrows, v.ncols, v.rs, v.cs}; }
	int *lu_status = nullptr; // 16 zeroed device ints: the panel kernels' status words, read once at the end
	void factor_panel(View P, int *piv_out)
	{
		t_panel.begin();
		getrf_panel_dev<T>(mv(P), piv_out, lu_status);
		t_panel.end();
	}
	// The orchestration applies the pivots of panel k to up to three column ranges per step (left of the panel, the look-ahead
	// block column, everything right of it).  The net permutation is composed ONCE per (step, pivot buffer) and the launches
	// gather / scatter through it, as in the single-GPU driver: laswp_small_kernel rebuilt it in every workgroup, 199 us per
	// launch, 18 of a rank's 110 ms at N = 16384 (profiles/r06_exp_dist.txt).  All interchanges of a step are issued on one
	// stream, in order; two list buffers alternate so that a step never rewrites the list of the step before.
	long step_id = 0;
	const int *list_piv = nullptr;
	long list_step = -1;
	int list_nt = 0;
	std::unique_ptr<Scratch> lists;
	void laswp(View B, const int *piv, int nt)
	{
		if (nt > laswp_list_max() || nt < 64) {
			laswp_rows_dev<T>(mv(B), piv, nt);
			return;
		}
		const size_t cap = (size_t) 4 * laswp_list_max();
		if (!lists)
			lists.reset(new Scratch(2 * cap * sizeof(int)));
		int *list = lists->as<int>() + (size_t) (step_id & 1) * cap;
		if (list_piv != piv || list_step != step_id || list_nt != nt) {
			laswp_compose_rows_dev(piv, nt, list);
			list_piv = piv;
			list_step = step_id;
			list_nt = nt;
		}
		laswp_list_rows_dev<T>(mv(B), list, nt);
	}
	void trsm_unit_lower(View L, View X) { trsm_lower_dev<T>(mv(L).c(), true, mv(X)); }
	void gemm_sub(View C, View A, View B) { gemm_dev<T>(mv(C), DST_FULL, true, mv(A).c(), mv(B).c(), (T) -1); }
	void pack(View src, T *dst) { copy_dev<T>(MatV<T>{dst, src.nrows, src.ncols, 1, src.nrows}, mv(src).c()); }
	void bcast(void *buf, size_t bytes, int root)
	{
		FH_CHECK(comm.bcast != nullptr, "dist: FaerHipComm.bcast is NULL");
		comm.bcast(comm.user, buf, bytes, root);
	}
	// The caller's stream is usually the NULL stream (torch's default stream), and the look-ahead streams are blocking streams
	// (hipExtStreamCreateWithCUMask takes no flags): every command on the null stream -- an event record, even a wait -- is
	// ordered against ALL of them, so one transport call or hand-shake on the caller's stream in the middle of a step stalls the
	// bulk stream for a cross-queue round trip (one-rank Cholesky over the RCCL transport: 170 us between two pack launches,
	// 83.7 ms against 66.3 with a callback that does nothing; profiles/r06_exp_dist.txt).  With the built-in RCCL transport
	// (`quiet`: its calls only need ctx().stream, whichever stream that is) a run therefore leaves the caller's stream alone
	// between its first two-stream step and run_end: broadcasts start from the stream that packed (owner) or last read (receiver)
	// the buffer, waits are taken by the bulk stream, the panel stream joins the bulk stream directly.
	// (the loop-back transport of the tests has the same contract: loop_transport.hip)
	bool builtin_wait() const { return rccl_is_builtin_wait(comm.wait) || loop_is_builtin_wait(comm.wait); }
	bool quiet() const { return comm.ibcast && comm.wait && builtin_wait(); }
	bool bcast_from_panel = false; // dist_lu.h: the next broadcast ships what the panel stream has just packed (set by ahead_join)
	hipEvent_t ev_join = nullptr;  // quiet: panel-stream work the bulk stream has to join before its next section
	// asynchronous pair when the transport offers one, else the blocking broadcast at `begin`
	void bcast_begin(void *buf, size_t bytes, int root, int slot)
	{
		if (comm.ibcast && comm.wait) {
			hipStream_t cur = ctx().stream;
			if (quiet() && prev_two && cur == caller)
				ctx().stream = bcast_from_panel ? ctx().la_panel : ctx().la_bulk;
			comm.ibcast(comm.user, buf, bytes, root, slot);
			ctx().stream = cur;
		} else {
			bcast(buf, bytes, root);
		}
		bcast_from_panel = false;
	}
	// The transport's wait orders a stream behind the transfer.  The built-in RCCL transport's wait is a stream-wait on an event
	// and takes ctx().stream as it is -- the panel / bulk stream that first reads the chunk (dist_llt.h), and only that stream
	// waits.  A callback transport (torch.distributed handles) orders the stream of ITS world, the caller's: the wait is taken
	// there and handed to the internal stream through an event -- that event also carries whatever else the caller's stream is
	// waiting for at that point (the owner's look-ahead pauses), so with a callback transport the rest of update k can start
	// later than the transfer alone would demand (a dedicated helper stream would be a fifth stream of this process,
	// see Ctx::qr_side_streams for what that costs).
	void bcast_wait(int slot)
	{
		if (!(comm.ibcast && comm.wait))
			return;
		hipStream_t cur = ctx().stream;
		if (caller && cur != caller && !builtin_wait()) {
			ctx().stream = caller;
			comm.wait(comm.user, slot);
			hipEvent_t e = ctx().next_event();
			FH_HIP(hipEventRecord(e, caller));
			ctx().stream = cur;
			stream_wait(cur, e);
		} else if (quiet() && prev_two && cur == caller) {
			// (every consumer of a received panel, and every later writer of its buffer, runs on or behind the bulk stream)
			ctx().stream = ctx().la_bulk;
			comm.wait(comm.user, slot);
			ctx().stream = cur;
		} else {
			comm.wait(comm.user, slot);
			if (cur == caller)
				caller_dirty = true;
		}
	}
	// ---- two-stream schedule inside the rank (dist_lu.h): the rest of update k on the bulk stream, the look-ahead part
	// (update of block column k+1 + its panel factorization, cooperative leaves on the reserved CUs) on the panel stream
	bool two = false;     // the two internal streams are available
	bool two_now = false; // ... and used in the current step
	long two_min_work = 0; // trailing entries (rows x local columns right of the panel) from which a step uses both streams (round 6: always)
	hipStream_t caller = nullptr;
	hipEvent_t ev0 = nullptr, ev_bulk = nullptr, ev_ahead = nullptr;
	void streams_init()
	{
		caller = ctx().stream;
		two = !getenv("FAER_HIP_DIST_ONE_STREAM") && ctx().lookahead_streams();
		if (const char *e = getenv("FAER_HIP_DIST_TWO_MIN"))
			two_min_work = atol(e);
		if (two)
			ctx().reset_events();
	}
	// Rounds 2-5 used both streams only in steps with >= 1e8 trailing entries on the rank: the look-ahead part then
	// INCLUDED the update of block column k + 1, on the panel stream's 32 CUs (~1 ms per step), and that only paid beside a
	// long rest.  Round 6 runs that update on the bulk stream in front of the rest (dist_lu.h), so the panel stream carries the
	// panel alone, as in the single-GPU driver, which keeps its two streams down to the last step: one rank, N = 16384,
	// threshold 1e8 / 3e7 / 0: 191.7 / 156.7 / 123.6 ms (profiles/r06_exp_dist.txt).  FAER_HIP_DIST_TWO_MIN overrides.
	// `caller_dirty`: the caller's stream has taken a dependency since the last hand-shake that the internal streams do not have
	// (a transfer awaited there, the panel stream joined into it).  While it has not, a step that uses both streams like the step
	// before it is ordered by the streams themselves and keeps the old (long complete) ev0: the hand-shake "caller waits for the
	// bulk stream, records ev0, the bulk stream waits for ev0" cost two cross-stream hops (~70 us) in front of every step of the
	// one-rank Cholesky (profiles/r06_exp_dist.txt).
	bool caller_dirty = true, prev_two = false;
	void step_begin(long local_trailing_entries, long next_panel_rows)
	{
		++step_id;
		if (!two)
			return;
		// the look-ahead panel is factored on the CU-masked panel stream: its cooperative leaves must fit those CUs at the
		// width they would have on the whole chip (dist_two_streams_ok, getrf.hip) -- beyond 131072 fp64 rows they do not
		// fit at all and the leaf would abort
		two_now = local_trailing_entries >= two_min_work && dist_two_streams_ok(next_panel_rows, (int) sizeof(T), ctx().la_panel_cus, ctx().ncu > 0 ? ctx().ncu : 256);
		const bool keep = two_now && prev_two && !caller_dirty && ev0; // the streams order this step against the last one themselves
		// the bulk stream's reads of the panel buffer that the next receive overwrites, and its writes to the columns the
		// look-ahead part touches, are older than everything issued from here on
		if (!(keep && quiet())) {
			if (ev_bulk) {
				stream_wait(caller, ev_bulk);
				ev_bulk = nullptr;
			}
			if (ev_join) {
				stream_wait(caller, ev_join);
				ev_join = nullptr;
			}
		}
		if (!two_now) {
			prev_two = false;
			return;
		}
		if (!keep) {
			ev0 = ctx().next_event();
			FH_HIP(hipEventRecord(ev0, caller));
			caller_dirty = false;
		}
		prev_two = true;
	}
	// quiet: what the panel stream did in the step before (ahead_join) is joined by the bulk stream's next section
	void bulk_joins_panel()
	{
		if (ev_join) {
			stream_wait(ctx().la_bulk, ev_join);
			ev_join = nullptr;
		}
	}
	void rest_begin()
	{
		if (!two_now)
			return;
		stream_wait(ctx().la_bulk, ev0);
		bulk_joins_panel();
		ctx().stream = ctx().la_bulk;
	}
	void rest_end()
	{
		if (!two_now)
			return;
		ev_bulk = ctx().next_event();
		FH_HIP(hipEventRecord(ev_bulk, ctx().la_bulk));
		ctx().stream = caller;
	}
	// the update of the look-ahead block column: on the bulk stream, in front of the rest of the step (dist_lu.h)
	hipEvent_t ev_cols = nullptr;
	void ahead_cols_begin()
	{
		if (!two_now)
			return;
		stream_wait(ctx().la_bulk, ev0);
		bulk_joins_panel();
		ctx().stream = ctx().la_bulk;
	}
	void ahead_cols_end()
	{
		if (!two_now)
			return;
		ev_cols = ctx().next_event();
		FH_HIP(hipEventRecord(ev_cols, ctx().la_bulk));
		ctx().stream = caller;
	}
	void ahead_begin()
	{
		if (!two_now)
			return;
		stream_wait(ctx().la_panel, ev0);
		if (ev_cols) {
			stream_wait(ctx().la_panel, ev_cols);
			ev_cols = nullptr;
		}
		ctx().stream = ctx().la_panel;
	}
	void ahead_end()
	{
		if (!two_now)
			return;
		ev_ahead = ctx().next_event();
		FH_HIP(hipEventRecord(ev_ahead, ctx().la_panel));
		ctx().stream = caller;
	}
	// inside the look-ahead part (dist_llt.h: the broadcast of a chunk): the caller's stream joins the panel stream's work so
	// far and takes the next launches / transport calls; the panel stream goes on afterwards
	hipStream_t paused = nullptr;
	void ahead_pause()
	{
		paused = nullptr;
		if (!two_now || ctx().stream == caller || quiet())
			return; // (quiet: the broadcast starts from the stream that packed the chunk)
		paused = ctx().stream;
		hipEvent_t e = ctx().next_event();
		FH_HIP(hipEventRecord(e, paused));
		stream_wait(caller, e);
		ctx().stream = caller;
	}
	void ahead_resume()
	{
		if (paused)
			ctx().stream = paused;
		paused = nullptr;
	}
	// dist_llt.h: the solve of the new panel's rows -- on the bulk stream, behind the diagonal block (panel stream)
	void ahead_solve_begin()
	{
		if (!two_now)
			return;
		stream_wait(ctx().la_bulk, ev0);
		if (ev_ahead)
			stream_wait(ctx().la_bulk, ev_ahead);
		ctx().stream = ctx().la_bulk;
	}
	void ahead_solve_end()
	{
		if (!two_now)
			return;
		ev_bulk = ctx().next_event();
		FH_HIP(hipEventRecord(ev_bulk, ctx().la_bulk));
		ctx().stream = caller;
	}
	void ahead_join()
	{
		if (two_now && ev_ahead) {
			if (quiet()) { // the broadcast starts from the panel stream; the bulk stream joins it before its next section
				ev_join = ev_ahead;
				bcast_from_panel = true;
			} else {
				stream_wait(caller, ev_ahead); // the broadcast of the new panel is ordered behind it
				caller_dirty = true;
			}
			ev_ahead = nullptr;
		}
	}
	void run_end()
	{
		if (two && (ev_bulk || prev_two)) {
			// everything queued on the bulk stream so far -- with the built-in transport that includes the last waits for this
			// rank's own broadcasts, which still read the workspace the caller may release after the call
			hipEvent_t e = ctx().next_event();
			FH_HIP(hipEventRecord(e, ctx().la_bulk));
			stream_wait(caller, e);
		}
		if (two && ev_join)
			stream_wait(caller, ev_join);
		ev_bulk = ev_join = nullptr;
	}
	void copy_ints(int *dst, const int *src, size_t n)
	{
		FH_HIP(hipMemcpyAsync(dst, src, n * sizeof(int), hipMemcpyDeviceToDevice, ctx().stream));
	}
	void zero_ints(int *p, size_t n) { FH_HIP(hipMemsetAsync(p, 0, n * sizeof(int), ctx().stream)); }
	void from_host(int *dst, const int *src, size_t n)
	{
		FH_HIP(hipMemcpyAsync(dst, src, n * sizeof(int), hipMemcpyHostToDevice, ctx().stream));
		ctx().sync(); // `src` is a stack buffer of the caller
	}
	// ---- Cholesky (dist_llt.h)
	T reg_delta = (T) 0, reg_eps = (T) 0;
	void potrf_panel(View P, long offset, int *status)
	{
		t_panel.begin();
		potrf_panel_dev<T>(mv(P), reg_delta, reg_eps, status, (idx_t) offset);
		t_panel.end();
	}
	// X <- X L^-T (cholesky/ldlt/factor.rs:422-426, expressed like the reference as L \ X^T); counted as panel time
	void solve_rows(View L, View X)
	{
		if (X.nrows <= 0)
			return;
		t_panel.begin();
		trsm_lower_dev<T>(mv(L).c(), false, mv(X).t());
		t_panel.end();
	}
	void syrk_sub(View C, View A, View Bt) { gemm_dev<T>(mv(C), DST_LOWER, true, mv(A).c(), mv(Bt).t().c(), (T) -1); }
	void gemm_sub_nt(View C, View A, View Bt)
	{
		if (C.nrows > 0 && C.ncols > 0)
			gemm_dev<T>(mv(C), DST_FULL, true, mv(A).c(), mv(Bt).t().c(), (T) -1);
	}
	void gather_stair(View P, long ncols, long nb, long gap, T *dst, long ld)
	{
		if (ncols <= 0 || P.ncols <= 0)
			return;
		hipLaunchKernelGGL(gather_stair_kernel<T>, dim3((unsigned) ((ncols + 255) / 256), (unsigned) P.ncols), dim3(256), 0, ctx().stream, P.p, P.rs,
				   P.cs, ncols, P.ncols, nb, gap, dst, ld);
		FH_HIP(hipGetLastError());
	}
	void syrk_stair_sub(View C, View A, View Bt, long nb, long gap, long row0)
	{
		if (C.nrows <= 0 || C.ncols <= 0)
			return;
		GemmExtra<T> ex;
		ex.stair_nb = (idx_t) nb;
		ex.stair_gap = (idx_t) gap;
		ex.stair_row0 = (idx_t) row0;
		gemm_dev<T>(mv(C), DST_LOWER, true, mv(A).c(), mv(Bt).t().c(), (T) -1, &ex);
	}
	void to_host(int *dst, const int *src, size_t n)
	{
		FH_HIP(hipMemcpyAsync(dst, src, n * sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
		ctx().sync();
	}
};

template <typename T>
FaerPartialPivLuStatus dist_lu_api(FaerMatMut A_local, size_t n_global, size_t nb, FaerSliceMut pf, FaerSliceMut pb, FaerHipComm comm,
				   void *panel_ws)
{
	typedef DeviceBackend<T> B;
	const long m = (long) A_local.nrows, n = (long) n_global;
	FH_CHECK(comm.world_size >= 1 && comm.rank >= 0 && comm.rank < comm.world_size, "dist lu: bad communicator");
	FH_CHECK(nb >= 1 && (long) nb <= m, "dist lu: block width must be in [1, nrows]");
	FH_CHECK((size_t) A_local.ncols == DistLu<B>::local_ncols(n_global, nb, comm.rank, comm.world_size),
		 "dist lu: A_local has the wrong number of columns for this rank");
	FH_CHECK((long) pf.len == m && (long) pb.len == m, "dist lu: perm slices must have nrows entries");
	FH_CHECK(is_device_ptr(A_local.ptr) && is_device_ptr(panel_ws), "dist lu: A_local and panel_ws must be device memory");
	FH_CHECK(A_local.row_stride == 1, "dist lu: A_local must be column major");
	B be;
	be.comm = comm;
	be.streams_init();
	Scratch stb(64);
	be.lu_status = stb.as<int>();
	FH_HIP(hipMemsetAsync(stb.p, 0, 64, ctx().stream)); // older than everything the run queues on any stream (step_begin)
	typename B::View Av{static_cast<T *>(A_local.ptr), m, (long) A_local.ncols, 1, (long) A_local.col_stride};
	const long size = m < n ? m : n;
	std::vector<int> piv((size_t) size);
	be.t_total.begin();
	DistLu<B>::run(be, Av, m, n, (long) nb, comm.rank, comm.world_size, static_cast<T *>(panel_ws), piv.data());
	be.t_total.end();
	int lst[4] = {0, 0, 0, 0};
	FH_HIP(hipMemcpyAsync(lst, be.lu_status, sizeof(lst), hipMemcpyDeviceToHost, ctx().stream));
	ctx().sync(); // the run joined both internal streams into the caller's
	g_dist_stats.panels = (int) be.t_panel.ev.size();
	g_dist_stats.panel_ms = be.t_panel.harvest();
	g_dist_stats.total_ms = be.t_total.harvest();
	if (be.two)
		ctx().quiesce();
	if (lst[2] != 0) { // a cooperative panel kernel of THIS rank gave up waiting for its peers (getrf.hip): factors are garbage
		FaerPartialPivLuStatus bad;
		memset(&bad, 0, sizeof(bad));
		bad.tag = FaerPartialPivLuStatus_Unknown;
		return bad;
	}
	// lu/partial_pivoting/factor.rs:274-277: perm = identity with the transpositions applied in order
	unsigned long long *f = static_cast<unsigned long long *>(pf.ptr), *b = static_cast<unsigned long long *>(pb.ptr);
	for (long i = 0; i < m; ++i)
		f[i] = (unsigned long long) i;
	size_t nt = 0;
	for (long j = 0; j < size; ++j)
		if (piv[(size_t) j] != j) {
			std::swap(f[j], f[piv[(size_t) j]]);
			++nt;
		}
	for (long i = 0; i < m; ++i)
		b[f[i]] = (unsigned long long) i;
	FaerPartialPivLuStatus st;
	memset(&st, 0, sizeof(st));
	st.tag = FaerPartialPivLuStatus_Ok;
	st.ok.transposition_count = nt;
	return st;
}

template <typename T>
FaerLltStatus dist_llt_api(FaerMatMut A_local, size_t n_global, size_t nb, FaerLltRegularization reg, FaerHipComm comm, void *panel_ws)
{
	typedef DeviceBackend<T> B;
	const long n = (long) n_global;
	FH_CHECK(comm.world_size >= 1 && comm.rank >= 0 && comm.rank < comm.world_size, "dist llt: bad communicator");
	FH_CHECK(nb >= 1 && (long) nb <= (n > 0 ? n : 1), "dist llt: block width must be in [1, n]");
	FH_CHECK((long) A_local.nrows == n, "dist llt: A_local must have n rows");
	FH_CHECK((size_t) A_local.ncols == DistLu<B>::local_ncols(n_global, nb, comm.rank, comm.world_size),
		 "dist llt: A_local has the wrong number of columns for this rank");
	FH_CHECK(A_local.row_stride == 1, "dist llt: A_local must be column major");
	FaerLltStatus st;
	memset(&st, 0, sizeof(st));
	st.tag = FaerLltStatus_Ok;
	if (n == 0)
		return st;
	FH_CHECK(is_device_ptr(panel_ws) && (A_local.ncols == 0 || is_device_ptr(A_local.ptr)), "dist llt: A_local and panel_ws must be device memory");
	B be;
	be.comm = comm;
	be.reg_delta = reg.dynamic_regularization_delta ? *static_cast<const T *>(reg.dynamic_regularization_delta) : (T) 0;
	be.reg_eps = reg.dynamic_regularization_epsilon ? *static_cast<const T *>(reg.dynamic_regularization_epsilon) : (T) 0;
	be.streams_init(); // the two-stream schedule of the LU (step_begin / ahead_* / rest_*: no-ops without it -- ADVICE r03)
	// (rounds 2-5: both streams only beside >= 1e8 trailing entries, because the look-ahead part put its level-3 work on the panel
	// stream's 32 CUs; now only the diagonal block's chain runs there -- dist_llt.h -- and every step uses both streams)
	typename B::View Av{static_cast<T *>(A_local.ptr), n, (long) A_local.ncols, 1, (long) A_local.col_stride};
	be.t_total.begin();
	const long r = DistLlt<B>::run(be, Av, n, (long) nb, comm.rank, comm.world_size, static_cast<T *>(panel_ws));
	be.t_total.end();
	ctx().sync(); // the run joined both internal streams into the caller's
	if (be.two)
		ctx().quiesce();
	g_dist_stats.panels = (int) be.t_panel.ev.size();
	g_dist_stats.panel_ms = be.t_panel.harvest();
	g_dist_stats.total_ms = be.t_total.harvest();
	if (r >= 0) {
		st.ok.dynamic_regularization_count = (size_t) r;
	} else {
		st.tag = FaerLltStatus_NonPositivePivot;
		st.non_positive_pivot.index = (size_t) (-r - 1);
	}
	return st;
}

} // namespace

extern "C" {
void faer_hip_dist_last_stats(double *out3)
{
	out3[0] = g_dist_stats.total_ms;
	out3[1] = g_dist_stats.panel_ms;
	out3[2] = (double) g_dist_stats.panels;
}
size_t faer_hip_dist_llt_ws_scalars(size_t n, size_t nb, FaerHipDType dtype)
{
	return dtype == FaerHipDType_F64 ? DistLlt<DeviceBackend<double>>::ws_scalars((long) n, (long) nb)
					 : DistLlt<DeviceBackend<float>>::ws_scalars((long) n, (long) nb);
}
FaerLltStatus faer_hip_dist_llt_f64(FaerMatMut A, size_t n, size_t nb, FaerLltRegularization reg, FaerHipComm comm, void *ws)
{
	return dist_llt_api<double>(A, n, nb, reg, comm, ws);
}
FaerLltStatus faer_hip_dist_llt_f32(FaerMatMut A, size_t n, size_t nb, FaerLltRegularization reg, FaerHipComm comm, void *ws)
{
	return dist_llt_api<float>(A, n, nb, reg, comm, ws);
}
size_t faer_hip_dist_local_ncols(size_t n, size_t nb, int rank, int world_size)
{
	return DistLu<DeviceBackend<double>>::local_ncols(n, nb, rank, world_size);
}
size_t faer_hip_dist_panel_ws_scalars(size_t nrows, size_t nb, FaerHipDType dtype)
{
	return dtype == FaerHipDType_F64 ? DistLu<DeviceBackend<double>>::ws_scalars((long) nrows, (long) nb)
					 : DistLu<DeviceBackend<float>>::ws_scalars((long) nrows, (long) nb);
}
FaerPartialPivLuStatus faer_hip_dist_partial_piv_lu_f64(FaerMatMut A, size_t n, size_t nb, FaerSliceMut pf, FaerSliceMut pb, FaerHipComm comm,
							  void *ws)
{
	return dist_lu_api<double>(A, n, nb, pf, pb, comm, ws);
}
FaerPartialPivLuStatus faer_hip_dist_partial_piv_lu_f32(FaerMatMut A, size_t n, size_t nb, FaerSliceMut pf, FaerSliceMut pb, FaerHipComm comm,
							  void *ws)
{
	return dist_lu_api<float>(A, n, nb, pf, pb, comm, ws);
}
}
