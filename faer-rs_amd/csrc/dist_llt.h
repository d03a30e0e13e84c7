// Distributed Cholesky (lower) over a 1-D block-cyclic column partition: one process per
// GPU, ONE broadcast per block column -- the factored column panel L[k:, k] from its owner -- and purely local
// rank-nb updates of the owned block columns.  Same mathematics as the single-GPU driver
// (cholesky/ldlt/factor.rs:367-498 with is_llt): panel = Cholesky of the diagonal block + solve of the rows below
// (:407-433), trailing update lower(A11) -= L10 L10^T (:436-446) restricted to the columns a rank owns.
//
// Look-ahead: the owner of block column k+1 updates and factors that column FIRST and starts its broadcast; every
// rank posts the receive before it runs the rest of update k, so the transfer of panel k+1 (and the latency-bound
// panel factorization on its owner) overlaps with the trailing updates of step k.  Two panel buffers alternate.
// On the owner the look-ahead part is issued first (an asynchronous backend queues it on its panel stream) and the
// rest of the update second (bulk stream): the two run concurrently inside the rank, as in dist_lu.h.
//
// The rest of update k is ONE product per rank, not one per owned block column: the owned block columns right of the
// panel lie next to each other in A_local, the rows of the panel that belong to them are gathered into a contiguous
// operand, and the product writes the lower part under a STAIRCASE (block column i of the range starts `world` blocks
// further down than block column i - 1).  Per entry the arithmetic is that of the block-by-block update.
//
// Template over a backend like dist_lu.h (device backend in dist.hip, host backend under tests/).  Backend B:
//   typedef scalar T;  struct View { T *p; long nrows, ncols, rs, cs; };
//   void potrf_panel(View P, long offset, int *status)   -- P: rows x w, top w x w block = diagonal block (lower part
//                                                           referenced); in place: L_kk and the solved rows below.
//                                                           status (backend memory): [0] = first failing global index
//                                                           + 1 (kept if already set), [1] += regularisation count
//   void syrk_sub(View C, View A, View Bt)               -- C -= A * Bt^T; rows 0..C.ncols-1 of C: lower part only
//   void gather_stair(View P, long ncols, long nb, long gap, T *dst)
//                                                        -- dst (ncols x P.ncols, column major, ld = ncols) <- the rows
//                                                           c + (c / nb) * gap of P, c < ncols
//   void syrk_stair_sub(View C, View A, View Bt, long nb, long gap)
//                                                        -- C(i, c) -= (A Bt^T)(i, c) for i >= c + (c / nb) * gap
//   void step_begin(long local_trailing_entries, long next_panel_rows) / rest_begin() / rest_end() / ahead_begin() /
//        ahead_end() / ahead_join() / run_end()          -- scheduling hooks as in dist_lu.h
//   void pack(View src, T *dst)                          -- contiguous column-major copy into the panel buffer
//   void bcast_begin(void *buf, size_t bytes, int root, int slot) / void bcast_wait(int slot)
//   void bcast(void *buf, size_t bytes, int root)        -- blocking (status exchange at the end)
//   void to_host(int *dst, const int *src, size_t n)     -- synchronising
//   void zero_ints(int *p, size_t n)
#pragma once
#include <cstddef>
#include <cstdint>

namespace fh {

template <class B> struct DistLlt {
	typedef typename B::T T;
	typedef typename B::View View;

	static size_t hdr_scalars() { return (4 * sizeof(int) + sizeof(T) - 1) / sizeof(T); }
	static size_t buf_scalars(long n, long nb) { return (size_t) n * (size_t) nb; }
	// [status: 4 ints][panel buffer 0][panel buffer 1][gathered panel rows of the owned block columns]
	static size_t ws_scalars(long n, long nb) { return hdr_scalars() + 3 * buf_scalars(n, nb); }

	// A_local: n x local_ncols (this rank's block columns in increasing global order, full height; only the lower
	// triangle of the global matrix is referenced or written).  Returns -(index + 1) for the first non-positive
	// pivot (global index, identical on every rank), else the dynamic regularisation count.
	static long run(B &be, View A_local, long n, long nb, int rank, int world, T *ws)
	{
		const long nblk = (n + nb - 1) / nb;
		int *status = reinterpret_cast<int *>(ws);
		T *bufs = ws + hdr_scalars();
		const size_t bsz = buf_scalars(n, nb);
		auto buf = [&](long k) { return bufs + (size_t) (k & 1) * bsz; };
		auto width = [&](long b) { return (b + 1) * nb <= n ? nb : n - b * nb; };
		auto local_col0 = [&](long b) {
			long c = 0;
			for (long bb = rank; bb < b; bb += world)
				c += width(bb);
			return c;
		};
		auto view = [&](long r0, long c0, long nr, long nc) {
			return View{A_local.p + r0 * A_local.rs + c0 * A_local.cs, nr, nc, A_local.rs, A_local.cs};
		};
		auto factor_and_pack = [&](long k) { // owner of block column k
			const long j0 = k * nb, w = width(k), rows = n - j0, lc = local_col0(k);
			be.potrf_panel(view(j0, lc, rows, w), j0, status);
			be.pack(view(j0, lc, rows, w), buf(k));
		};
		auto update = [&](long k, long b) { // block column b (owned by this rank, b > k) -= panel k
			const long j0 = k * nb, w = width(k), rows = n - j0;
			const long bc0 = b * nb, bw = width(b), lc = local_col0(b), off = bc0 - j0;
			View P{buf(k), rows, w, 1, rows};
			be.syrk_sub(view(bc0, lc, n - bc0, bw), View{P.p + off, rows - off, w, 1, rows}, View{P.p + off, bw, w, 1, rows});
		};
		T *gbuf = bufs + 2 * bsz;
		// the rest of update k: all owned block columns right of the panel (without block k + 1 if this rank brings it up
		// to date in the look-ahead part) in one staircase product
		auto rest = [&](long k, bool skip_next) {
			be.rest_begin();
			long b0 = -1, c_first = 0, c_all = 0; // first block of the range, its first local column, local columns in all
			for (long b = rank; b < nblk; b += world) {
				if (b0 < 0 && b > k && !(skip_next && b == k + 1)) {
					b0 = b;
					c_first = c_all;
				}
				c_all += width(b);
			}
			if (b0 >= 0) {
				const long j0 = k * nb, w = width(k), rows = n - j0, off = b0 * nb - j0, nc = c_all - c_first;
				View Pk{buf(k) + off, rows - off, w, 1, rows};
				be.gather_stair(Pk, nc, nb, (world - 1) * nb, gbuf);
				be.syrk_stair_sub(view(b0 * nb, c_first, n - b0 * nb, nc), Pk, View{gbuf, nc, w, 1, nc}, nb, (world - 1) * nb);
			}
			be.rest_end();
		};
		be.zero_ints(status, 4);
		if (rank == 0 % world)
			factor_and_pack(0);
		be.bcast_begin(buf(0), (size_t) n * (size_t) width(0) * sizeof(T), 0, 0);
		for (long k = 0; k < nblk; ++k) {
			be.bcast_wait((int) (k & 1));
			const bool ahead = k + 1 < nblk;
			const int next_owner = ahead ? (int) ((k + 1) % world) : -1;
			{ // trailing entries this rank updates in this step, rows of the next panel: does the step use both streams?
				long right = 0;
				for (long b = rank; b < nblk; b += world)
					if (b > k)
						right += width(b);
				be.step_begin((n - k * nb) * right / 2, n - (k + 1) * nb);
			}
			if (ahead && rank == next_owner) {
				be.ahead_begin();
				update(k, k + 1);
				factor_and_pack(k + 1);
				be.ahead_end();
				rest(k, true); // runs beside the panel on an asynchronous backend
				be.ahead_join();
				be.bcast_begin(buf(k + 1), (size_t) (n - (k + 1) * nb) * (size_t) width(k + 1) * sizeof(T), next_owner, (int) ((k + 1) & 1));
			} else {
				if (ahead) // post the receive before the updates: the transfer overlaps them
					be.bcast_begin(buf(k + 1), (size_t) (n - (k + 1) * nb) * (size_t) width(k + 1) * sizeof(T), next_owner, (int) ((k + 1) & 1));
				rest(k, false);
			}
		}
		be.run_end();
		// ---- outcome: every rank knows only about the panels it factored; combine (first failure, summed count)
		int mine[4] = {0, 0, 0, 0};
		be.to_host(mine, status, 4);
		long first_fail = 0, count = 0;
		for (int r = 0; r < world; ++r) {
			int st[4] = {0, 0, 0, 0};
			if (world > 1) {
				// the status words travel through the (now idle) panel buffer 0
				int *slot = reinterpret_cast<int *>(buf(0));
				if (r == rank)
					be.from_host(slot, mine, 4);
				be.bcast(slot, 4 * sizeof(int), r);
				be.to_host(st, slot, 4);
			} else {
				st[0] = mine[0];
				st[1] = mine[1];
			}
			if (st[0] != 0 && (first_fail == 0 || st[0] < first_fail))
				first_fail = st[0];
			count += st[1];
		}
		return first_fail != 0 ? -first_fail : count;
	}
};

} // namespace fh
