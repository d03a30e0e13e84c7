// TEST INFRASTRUCTURE ONLY (never linked into libfaer_hip.so): a host-memory backend for the distributed-LU
// orchestration template of faer-rs_amd/csrc/dist_lu.h, so that the world_size > 1 control flow (ownership,
// ONE broadcast per block column, interchanges / solve / update on the right columns) can run under gloo on a
// machine without a GPU.  The arithmetic here is a plain restatement of lu_in_place_unblocked
// (faer/src/linalg/lu/partial_pivoting/factor.rs:19-67) and of the obvious triple loops.
#include <cmath>
#include <cstring>
#include <vector>

#include "../faer-rs_amd/csrc/dist_llt.h"
#include "../faer-rs_amd/csrc/dist_lu.h"

typedef void (*BcastFn)(void *user, void *buf, size_t bytes, int root);

struct HostBackend {
	typedef double T;
	struct View {
		T *p;
		long nrows, ncols, rs, cs;
	};
	BcastFn cb;
	void *user;
	long n_bcast = 0;
	size_t bytes_bcast = 0;

	static T &at(View v, long i, long j) { return v.p[i * v.rs + j * v.cs]; }
	void factor_panel(View P, int *piv_out)
	{
		const long m = P.nrows, w = P.ncols;
		for (long j = 0; j < w; ++j) {
			// factor.rs:35-43: first row of strictly largest |a|, all-zero column keeps the diagonal
			long p = j;
			double best = 0.0;
			for (long i = j; i < m; ++i) {
				const double av = std::fabs(at(P, i, j));
				if (av > best) {
					best = av;
					p = i;
				}
			}
			piv_out[j] = (int) p;
			if (p != j)
				for (long c = 0; c < w; ++c)
					std::swap(at(P, j, c), at(P, p, c));
			const double inv = 1.0 / at(P, j, j);
			for (long i = j + 1; i < m; ++i) {
				at(P, i, j) *= inv;
				const double l = at(P, i, j);
				for (long c = j + 1; c < w; ++c)
					at(P, i, c) = std::fma(l, -at(P, j, c), at(P, i, c));
			}
		}
	}
	void laswp(View B, const int *piv, int nt)
	{
		for (int j = 0; j < nt; ++j)
			if (piv[j] != j)
				for (long c = 0; c < B.ncols; ++c)
					std::swap(at(B, j, c), at(B, piv[j], c));
	}
	void trsm_unit_lower(View L, View X)
	{
		for (long c = 0; c < X.ncols; ++c)
			for (long i = 0; i < L.nrows; ++i) {
				double s = at(X, i, c);
				for (long k = 0; k < i; ++k)
					s -= at(L, i, k) * at(X, k, c);
				at(X, i, c) = s;
			}
	}
	void gemm_sub(View C, View A, View B)
	{
		for (long j = 0; j < C.ncols; ++j)
			for (long i = 0; i < C.nrows; ++i) {
				double s = 0.0;
				for (long k = 0; k < A.ncols; ++k)
					s += at(A, i, k) * at(B, k, j);
				at(C, i, j) -= s;
			}
	}
	void pack(View src, T *dst)
	{
		for (long j = 0; j < src.ncols; ++j)
			for (long i = 0; i < src.nrows; ++i)
				dst[j * src.nrows + i] = at(src, i, j);
	}
	void bcast(void *buf, size_t bytes, int root)
	{
		++n_bcast;
		bytes_bcast += bytes;
		cb(user, buf, bytes, root);
	}
	void to_host(int *dst, const int *src, size_t n) { std::memcpy(dst, src, n * sizeof(int)); }
	// the look-ahead protocol of the drivers with a blocking transport: begin = the broadcast itself, wait = no-op;
	// `begun` / `waited` let the tests check that every broadcast is started exactly once and awaited exactly once
	// (slots 0 / 1: dist_lu.h; 0 .. 7: the chunks of dist_llt.h, which may be waited for once per context)
	long begun[8] = {0, 0, 0, 0, 0, 0, 0, 0}, waited[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	long pending[8] = {0, 0, 0, 0, 0, 0, 0, 0}, waits_without_transfer = 0;
	void bcast_begin(void *buf, size_t bytes, int root, int slot)
	{
		++begun[slot];
		++pending[slot];
		bcast(buf, bytes, root);
	}
	void bcast_wait(int slot)
	{
		++waited[slot];
		if (begun[slot] == 0)
			++waits_without_transfer; // a wait for a slot nothing was ever started on
		pending[slot] = 0;
	}
	// scheduling hooks of the asynchronous device backend: nothing to do on the host
	void step_begin(long, long) {}
	void rest_begin() {}
	void rest_end() {}
	void ahead_cols_begin() {}
	void ahead_cols_end() {}
	void ahead_begin() {}
	void ahead_end() {}
	void ahead_join() {}
	void ahead_pause() {}
	void ahead_solve_begin() {}
	void ahead_solve_end() {}
	void ahead_resume() {}
	void run_end() {}
	void copy_ints(int *dst, const int *src, size_t n) { std::memcpy(dst, src, n * sizeof(int)); }
	void zero_ints(int *p, size_t n) { std::memset(p, 0, n * sizeof(int)); }
	void from_host(int *dst, const int *src, size_t n) { std::memcpy(dst, src, n * sizeof(int)); }
	// ---- Cholesky: plain restatement of cholesky/ldlt/factor.rs:122-174 (llt) on the tall panel, column by column
	void potrf_panel(View P, long offset, int *status)
	{
		const long rows = P.nrows, w = P.ncols;
		for (long j = 0; j < w; ++j) {
			double d = at(P, j, j);
			for (long k = 0; k < j; ++k)
				d -= at(P, j, k) * at(P, j, k);
			if (!(d > 0.0)) {
				if (status[0] == 0)
					status[0] = (int) (offset + j + 1);
				return;
			}
			const double l = std::sqrt(d), inv = 1.0 / l;
			at(P, j, j) = l;
			for (long i = j + 1; i < rows; ++i) {
				double s = at(P, i, j);
				for (long k = 0; k < j; ++k)
					s -= at(P, i, k) * at(P, j, k);
				at(P, i, j) = s * inv;
			}
		}
	}
	// X <- X L^-T, row by row (cholesky/ldlt/factor.rs:422-426)
	void solve_rows(View L, View X)
	{
		for (long i = 0; i < X.nrows; ++i)
			for (long j = 0; j < X.ncols; ++j) {
				double s = at(X, i, j);
				for (long k = 0; k < j; ++k)
					s -= at(X, i, k) * at(L, j, k);
				at(X, i, j) = s * (1.0 / at(L, j, j));
			}
	}
	void gemm_sub_nt(View C, View A, View Bt)
	{
		for (long j = 0; j < C.ncols; ++j)
			for (long i = 0; i < C.nrows; ++i) {
				double s = 0.0;
				for (long k = 0; k < A.ncols; ++k)
					s += at(A, i, k) * at(Bt, j, k);
				at(C, i, j) -= s;
			}
	}
	void gather_stair(View P, long ncols, long nb, long gap, double *dst, long ld)
	{
		for (long k = 0; k < P.ncols; ++k)
			for (long c = 0; c < ncols; ++c)
				dst[k * ld + c] = at(P, c + (c / nb) * gap, k);
	}
	void syrk_stair_sub(View C, View A, View Bt, long nb, long gap, long row0)
	{
		for (long c = 0; c < C.ncols; ++c) {
			const long first = c + (c / nb) * gap - row0;
			for (long i = first > 0 ? first : 0; i < C.nrows; ++i) {
				double s = 0.0;
				for (long k = 0; k < A.ncols; ++k)
					s += at(A, i, k) * at(Bt, c, k);
				at(C, i, c) -= s;
			}
		}
	}
	void syrk_sub(View C, View A, View Bt)
	{
		for (long j = 0; j < C.ncols; ++j)
			for (long i = j; i < C.nrows; ++i) { // rows 0..ncols-1: lower part only
				double s = 0.0;
				for (long k = 0; k < A.ncols; ++k)
					s += at(A, i, k) * at(Bt, j, k);
				at(C, i, j) -= s;
			}
	}
};

extern "C" {
// returns the number of broadcasts issued; stats[0] = bytes broadcast
long test_dist_lu_f64(double *a_local, long m, long local_ncols, long ld, long n, long nb, int rank, int world, BcastFn cb, void *user,
		      int *piv_out, unsigned long long *stats)
{
	HostBackend be;
	be.cb = cb;
	be.user = user;
	std::vector<double> ws(fh::DistLu<HostBackend>::ws_scalars(m, nb));
	HostBackend::View A{a_local, m, local_ncols, 1, ld};
	fh::DistLu<HostBackend>::run(be, A, m, n, nb, rank, world, ws.data(), piv_out);
	stats[0] = be.bytes_bcast;
	stats[1] = (unsigned long long) (be.begun[0] + be.begun[1]);
	stats[2] = (unsigned long long) (be.waited[0] + be.waited[1]);
	stats[3] = (unsigned long long) (be.begun[0] > be.begun[1] ? be.begun[0] - be.begun[1] : be.begun[1] - be.begun[0]);
	return be.n_bcast;
}
// returns what DistLlt::run returns; stats[0] = bytes broadcast, stats[1] = number of broadcasts
long test_dist_llt_f64(double *a_local, long n, long local_ncols, long ld, long nb, int rank, int world, BcastFn cb, void *user,
		       unsigned long long *stats)
{
	HostBackend be;
	be.cb = cb;
	be.user = user;
	std::vector<double> ws(fh::DistLlt<HostBackend>::ws_scalars(n, nb));
	HostBackend::View A{a_local, n, local_ncols, 1, ld};
	const long r = fh::DistLlt<HostBackend>::run(be, A, n, nb, rank, world, ws.data());
	long begun = 0, waited = 0, unawaited = 0;
	for (int sl = 0; sl < 8; ++sl) {
		begun += be.begun[sl];
		waited += be.waited[sl];
		unawaited += be.pending[sl];
	}
	stats[0] = be.bytes_bcast;
	stats[1] = (unsigned long long) be.n_bcast;
	stats[2] = (unsigned long long) begun;
	stats[3] = (unsigned long long) waited;
	stats[4] = (unsigned long long) (unawaited + be.waits_without_transfer); // must be 0: every chunk awaited, no stray waits
	fh::DistLlt<HostBackend>::wire(n, nb, sizeof(double), reinterpret_cast<long *>(&stats[5]), reinterpret_cast<size_t *>(&stats[6]));
	return r;
}
long test_dist_local_ncols(long n, long nb, int rank, int world) { return (long) fh::DistLu<HostBackend>::local_ncols(n, nb, rank, world); }
}
