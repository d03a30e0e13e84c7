// Distributed Cholesky (lower) over a 1-D block-cyclic column partition: one process per
// GPU, the solved rows L[k+1:, k] of every block column broadcast from its owner, purely local rank-nb updates of the
// owned block columns.  Same mathematics as the single-GPU driver (cholesky/ldlt/factor.rs:367-498 with is_llt):
// panel = Cholesky of the diagonal block + solve of the rows below (:407-433), trailing update lower(A11) -= L10 L10^T
// (:436-446) restricted to the columns a rank owns.
//
// Round 4: the panel travels in ROW CHUNKS.  Once L_kk is known the rows of A21 L_kk^-T are independent of each other
// (factor.rs:422-426), so the owner packs and broadcasts the rows below the diagonal block in up to LLT_NCH
// block-aligned chunks (round 6: solved in one go before -- the solve chain is latency bound, not row bound) and every receiver
// updates the rows of a chunk as soon as that chunk has arrived: the rows of the panel that multiply an owned column lie in the same or an
// earlier chunk (lower triangle).  The owner of block column k + 1 does the same in its look-ahead part: diagonal block
// after chunk 0 of panel k, the rows' update chunk by chunk, one solve, then chunk by chunk "pack, start the broadcast".  With the whole panel as
// ONE message (rounds 2-3) every step carried "panel + transfer of up to 134 MB" on its critical chain; now it carries one
// diagonal block and one chunk.  The diagonal block itself is needed by nobody else and stays home.
//
// Look-ahead: the owner of block column k+1 updates and factors that column before (most of) the rest of update k and
// starts its broadcasts; every rank posts the receives before it runs the rest of update k.  Two panel buffers alternate.
// On the owner (round 6) only the diagonal block's factorization is queued on the panel stream of an asynchronous backend;
// the update of block column k+1, the solve of its rows and the rest of update k share the bulk stream in the order
// "diagonal block's update | rows' update, first half of the rest | solve, pack, broadcast | second half of the rest".
//
// The rest of update k is one staircase product per rank AND CHUNK, not one per owned block column: the owned block
// columns right of the panel lie next to each other in A_local, the rows of the panel that belong to them are gathered
// into a contiguous operand as their chunks arrive, and the product writes the lower part under a STAIRCASE (block
// column i of the range starts `world` blocks further down than block column i - 1).  Per entry the arithmetic is that
// of the block-by-block update.
//
// Template over a backend like dist_lu.h (device backend in dist.hip, host backend under tests/).  Backend B:
//   typedef scalar T;  struct View { T *p; long nrows, ncols, rs, cs; };
//   void potrf_panel(View P, long offset, int *status)   -- here: P = the w x w diagonal block (lower part referenced), in
//                                                           place.  status (backend memory): [0] = first failing global
//                                                           index + 1 (kept if already set), [1] += regularisation count
//   void solve_rows(View L, View X)                      -- X <- X L^-T (rows below the diagonal block, any subset)
//   void syrk_sub(View C, View A, View Bt)               -- C -= A * Bt^T; rows 0..C.ncols-1 of C: lower part only
//   void gemm_sub_nt(View C, View A, View Bt)            -- C -= A * Bt^T (all of C)
//   void gather_stair(View P, long ncols, long nb, long gap, T *dst, long ld)
//                                                        -- dst(c, :) <- row c + (c / nb) * gap of P, c < ncols (dst column
//                                                           major with leading dimension ld)
//   void syrk_stair_sub(View C, View A, View Bt, long nb, long gap, long row0)
//                                                        -- C(i, c) -= (A Bt^T)(i, c) for i + row0 >= c + (c / nb) * gap
//   void step_begin(long local_trailing_entries, long next_panel_rows) / rest_begin() / rest_end() / ahead_begin() /
//        ahead_end() / ahead_join() / ahead_cols_begin() / ahead_cols_end() / run_end()
//                                                        -- scheduling hooks as in dist_lu.h (rest_begin / rest_end may bracket
//                                                           several parts of one step's rest)
//   void ahead_solve_begin() / ahead_solve_end()         -- level-3 work of the look-ahead part that needs the diagonal block
//                                                           factored between ahead_begin / ahead_end
//   void ahead_pause() / ahead_resume()                  -- inside ahead_solve_*: what follows (a broadcast) is issued on the
//                                                           caller's context, ordered behind the look-ahead work so far
//   void pack(View src, T *dst)                          -- contiguous column-major copy into the panel buffer
//   void bcast_begin(void *buf, size_t bytes, int root, int slot) / void bcast_wait(int slot)   -- slots 0 .. 2 LLT_NCH - 1;
//                                                           a slot may be waited for more than once (once per context)
//   void bcast(void *buf, size_t bytes, int root)        -- blocking (status exchange at the end)
//   void to_host(int *dst, const int *src, size_t n)     -- synchronising
//   void zero_ints(int *p, size_t n)
#pragma once
#include <cstddef>
#include <cstdint>

namespace fh {

constexpr int LLT_NCH = 4; // row chunks of a panel (fewer when fewer block rows remain)

template <class B> struct DistLlt {
	typedef typename B::T T;
	typedef typename B::View View;

	static size_t hdr_scalars() { return (4 * sizeof(int) + sizeof(T) - 1) / sizeof(T); }
	static size_t buf_scalars(long n, long nb) { return (size_t) n * (size_t) nb; }
	// [status: 4 ints][panel buffer 0][panel buffer 1][gathered panel rows of the owned block columns]
	static size_t ws_scalars(long n, long nb) { return hdr_scalars() + 3 * buf_scalars(n, nb); }

	// chunks of panel k: the block rows k + 1 .. nblk - 1 in nch nearly equal block-aligned pieces; chunk c = rows [g[c], g[c + 1])
	struct Plan {
		int nch;
		long g[LLT_NCH + 1];
	};
	static Plan plan(long k, long nblk, long n, long nb)
	{
		Plan p;
		const long tb = nblk - k - 1;
		p.nch = (int) (tb < LLT_NCH ? tb : LLT_NCH);
		for (int c = 0; c <= LLT_NCH; ++c)
			p.g[c] = n;
		for (int c = 0; c < p.nch; ++c)
			p.g[c] = (k + 1 + (c * tb) / p.nch) * nb;
		return p;
	}
	// number of chunk broadcasts and their bytes for an n x n matrix (what the tests expect on the wire)
	static void wire(long n, long nb, size_t scalar_bytes, long *messages, size_t *bytes)
	{
		const long nblk = (n + nb - 1) / nb;
		*messages = 0;
		*bytes = 0;
		for (long k = 0; k < nblk; ++k) {
			const Plan p = plan(k, nblk, n, nb);
			const long w = (k + 1) * nb <= n ? nb : n - k * nb;
			*messages += p.nch;
			if (p.nch > 0)
				*bytes += (size_t) (n - p.g[0]) * (size_t) w * scalar_bytes;
		}
	}

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
		auto slot = [&](long k, int c) { return (int) ((k & 1) * LLT_NCH + c); };
		// rows [r0, r1) of panel k inside its chunk c, as packed in the panel buffer (chunk c: column major, ld = its rows)
		auto chunk_rows = [&](long k, const Plan &p, int c, long r0, long r1) {
			const long w = width(k), ld = p.g[c + 1] - p.g[c];
			T *base = buf(k) + (size_t) (p.g[c] - p.g[0]) * (size_t) w;
			return View{base + (r0 - p.g[c]), r1 - r0, w, 1, ld};
		};
		// each (context, chunk) waits for the chunk's transfer at most once: 0 = look-ahead part, 1 = rest of the update
		bool waited[2][LLT_NCH];
		auto wait_chunk = [&](long k, int c, int ctx) {
			if (!waited[ctx][c]) {
				be.bcast_wait(slot(k, c));
				waited[ctx][c] = true;
			}
		};
		// owner of block column k, its diagonal block factored: ALL rows below are solved in one go (a substitution leaf is latency
		// bound -- ~35 us for 4096 rows or for 16384 -- so chunk-wise solves cost nch times the chain: 1.0 against 0.3 ms per step
		// at N = 16384, profiles/r06_exp_dist.txt), then chunk by chunk: pack, start the broadcast
		auto solve_panel = [&](long k, const Plan &p) {
			const long j0 = k * nb, w = width(k), lc = local_col0(k);
			if (p.nch > 0)
				be.solve_rows(view(j0, lc, w, w), view(p.g[0], lc, n - p.g[0], w));
		};
		auto produce_chunk = [&](long k, const Plan &p, int c) {
			const long w = width(k), lc = local_col0(k), rows = p.g[c + 1] - p.g[c];
			View dst = chunk_rows(k, p, c, p.g[c], p.g[c + 1]);
			be.pack(view(p.g[c], lc, rows, w), dst.p);
			be.ahead_pause();
			be.bcast_begin(dst.p, (size_t) rows * (size_t) w * sizeof(T), (int) (k % world), slot(k, c));
			be.ahead_resume();
		};
		auto post_receives = [&](long k) {
			const Plan p = plan(k, nblk, n, nb);
			for (int c = 0; c < p.nch; ++c) {
				View dst = chunk_rows(k, p, c, p.g[c], p.g[c + 1]);
				be.bcast_begin(dst.p, (size_t) dst.nrows * (size_t) width(k) * sizeof(T), (int) (k % world), slot(k, c));
			}
		};
		T *gbuf = bufs + 2 * bsz;
		// the rest of update k: all owned block columns right of the panel (without block k + 1 if this rank brings it up
		// to date in the look-ahead part), one staircase product per chunk of the panel as the chunks arrive.  It can be
		// issued in two parts (chunks [0, c_end) first, the others later): the owner of block column k + 1 puts the solve of
		// its new panel between them (below).
		struct Rest {
			long b0 = -1, c_first = 0, nc = 0, nc_done = 0, b_next = 0;
			int c_next = 0;
		};
		auto rest_init = [&](long k, bool skip_next) {
			Rest s;
			long c_all = 0; // first block of the range, its first local column, local columns in all
			for (long b = rank; b < nblk; b += world) {
				if (s.b0 < 0 && b > k && !(skip_next && b == k + 1)) {
					s.b0 = b;
					s.c_first = c_all;
				}
				c_all += width(b);
			}
			s.nc = c_all - s.c_first;
			s.b_next = s.b0;
			return s;
		};
		auto rest_run = [&](long k, const Plan &p, Rest &s, int c_end) {
			be.rest_begin();
			if (s.b0 >= 0) {
				const long gap = (world - 1) * nb, b0 = s.b0;
				for (int c = s.c_next; c < c_end; ++c) {
					if (p.g[c + 1] <= b0 * nb)
						continue; // rows above the first owned column: nothing to update
					wait_chunk(k, c, 1);
					// rows of the owned blocks that start inside this chunk
					long ncols_c = 0;
					const long b_first = s.b_next;
					while (s.b_next < nblk && s.b_next * nb < p.g[c + 1]) {
						ncols_c += width(s.b_next);
						s.b_next += world;
					}
					if (ncols_c > 0)
						be.gather_stair(chunk_rows(k, p, c, b_first * nb, p.g[c + 1]), ncols_c, nb, gap, gbuf + s.nc_done, s.nc);
					s.nc_done += ncols_c;
					const long r0 = p.g[c] > b0 * nb ? p.g[c] : b0 * nb;
					be.syrk_stair_sub(view(r0, s.c_first, p.g[c + 1] - r0, s.nc_done), chunk_rows(k, p, c, r0, p.g[c + 1]),
							  View{gbuf, s.nc_done, width(k), 1, s.nc}, nb, gap, r0 - b0 * nb);
				}
			}
			if (c_end > s.c_next)
				s.c_next = c_end;
			be.rest_end();
		};
		be.zero_ints(status, 4);
		{
			const Plan p0 = plan(0, nblk, n, nb);
			if (rank == 0 % world) {
				be.potrf_panel(view(0, local_col0(0), width(0), width(0)), 0, status);
				solve_panel(0, p0);
				for (int c = 0; c < p0.nch; ++c)
					produce_chunk(0, p0, c);
			} else {
				post_receives(0);
			}
		}
		for (long k = 0; k < nblk; ++k) {
			const Plan p = plan(k, nblk, n, nb);
			for (int c = 0; c < LLT_NCH; ++c)
				waited[0][c] = waited[1][c] = false;
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
				// Look-ahead part: block column k + 1 is brought up to date, factored and sent chunk by chunk.  Round 6: only the
				// DIAGONAL block's factorization (a chain of latency-bound leaves) runs on the panel stream; the level-3 work --
				// the update of block column k + 1 and the solve of the rows below its diagonal block -- runs on the bulk stream
				// in front of / between the parts of the rest of update k, as the single-GPU driver does (potrf.hip): rounds 2-5
				// had all of it on the panel stream's few CUs, or everything on one stream (one rank, N = 16384: 100-105 ms
				// against 34 ms for the single-GPU driver, profiles/r06_exp_dist.txt).
				const Plan q = plan(k + 1, nblk, n, nb);
				const long d0 = (k + 1) * nb, w1 = width(k + 1), lc1 = local_col0(k + 1);
				View Bt1 = chunk_rows(k, p, 0, d0, d0 + w1); // the rows of panel k that belong to block column k + 1
				be.ahead_cols_begin();
				wait_chunk(k, 0, 0);
				be.syrk_sub(view(d0, lc1, w1, w1), Bt1, Bt1);
				be.ahead_cols_end();
				be.ahead_begin();
				be.potrf_panel(view(d0, lc1, w1, w1), d0, status);
				be.ahead_end();
				// the rows below the diagonal block: updated beside its factorization
				be.ahead_cols_begin();
				for (int c = 0; c < p.nch; ++c) {
					const long lo = q.g[0] > p.g[c] ? q.g[0] : p.g[c], hi = p.g[c + 1];
					if (lo >= hi)
						continue;
					wait_chunk(k, c, 0);
					be.gemm_sub_nt(view(lo, lc1, hi - lo, w1), chunk_rows(k, p, c, lo, hi), Bt1);
				}
				be.ahead_cols_end();
				// the first chunks of the rest of update k fill the time the diagonal block takes; everybody else needs panel
				// k + 1 only after THEIR rest of update k
				Rest rs = rest_init(k, true);
				rest_run(k, p, rs, p.nch / 2);
				be.ahead_solve_begin(); // (behind the diagonal block)
				solve_panel(k + 1, q);
				for (int cq = 0; cq < q.nch; ++cq)
					produce_chunk(k + 1, q, cq);
				be.ahead_solve_end();
				rest_run(k, p, rs, p.nch); // (the bulk stream has joined the panel stream in ahead_solve_begin: no ahead_join)
			} else {
				if (ahead) // post the receives before the updates: the transfers overlap them
					post_receives(k + 1);
				Rest rs = rest_init(k, false);
				rest_run(k, p, rs, p.nch);
			}
			// every chunk of panel k has been waited for by somebody on this rank before its buffer is reused two steps later
			for (int c = 0; c < p.nch; ++c)
				if (!waited[0][c] && !waited[1][c])
					wait_chunk(k, c, 1);
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
