// Distributed partial-pivot LU over a 1-D block-cyclic column partition: one process per
// GPU, ONE exchange per block column -- a broadcast of {factored panel, its pivots} from the owner -- and purely
// local row interchanges / triangular solves / trailing updates.  The caller owns the transport
// (FaerHipComm::bcast: RCCL through torch.distributed in bench.py, gloo in the CPU tests).
//
// The orchestration is a template over a backend so that the world_size > 1 control flow can be exercised
// without a GPU: dist.hip instantiates it with the device backend (the HIP kernels of getrf/trsm/gemm), the CPU
// test suite with a small host backend that lives under tests/ (never linked into libfaer_hip.so).
//
// Same mathematics as the single-GPU driver (lu/partial_pivoting/factor.rs:68-187): the panel is factored by the
// same recursive code on the owner, so pivots follow the reference's rule; the row interchanges reach every
// column of every rank (factor.rs:127-185), A01 <- L00^-1 A01 and A11 -= A10 A01 run on the owners of the columns.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace fh {

// Backend concept (B):
//   typedef scalar T;
//   struct View { T *p; long nrows, ncols, rs, cs; }   -- element (i, j) at p[i * rs + j * cs]
//   void factor_panel(View P, int *piv_out)              -- in-place LU of the m x w panel; piv_out: w ints in the
//                                                           same memory space as the matrix ("device" ints),
//                                                           piv_out[j] = row (relative to the panel's row 0) swapped with j
//   void laswp(View Bm, const int *piv, int nt)          -- applies (j <-> piv[j]), j < nt, to the rows of Bm
//   void trsm_unit_lower(View L, View X)                 -- X <- L^-1 X
//   void gemm_sub(View C, View A, View Bm)               -- C -= A * Bm
//   void pack(View src, T *dst)                          -- contiguous column-major copy into the panel buffer
//   void bcast_begin(void *buf, size_t bytes, int root, int slot) / void bcast_wait(int slot)
//                                                        -- collective on the backend's memory space; may be
//                                                           asynchronous (slot = 0 / 1, at most one in flight each)
//   void step_begin(long local_trailing_entries, long next_panel_rows) / void rest_begin() / rest_end() / ahead_cols_begin() /
//   ahead_cols_end() / ahead_begin() / ahead_end() / ahead_join()
//                                                        -- scheduling hooks (no-ops for a synchronous backend): the
//                                                           device backend runs the update of block column k+1 and then
//                                                           the "rest" updates of a step on the bulk stream, and the panel
//                                                           of block column k+1 on the CU-masked panel stream as soon as
//                                                           its columns are up to date
//   void copy_ints(int *dst, const int *src, size_t n)   -- inside the backend's memory space, stream ordered
//   void to_host(int *dst, const int *src, size_t n)     -- pivots back to the host (synchronising)
//
// Look-ahead: the owner of block column k+1 brings that column up to date with panel k and factors it FIRST, then
// starts its broadcast; every rank posts the receive before it runs the rest of update k, so the transfer of
// panel k+1 and the latency-bound panel factorization on its owner overlap with the trailing updates of step k.
// On the owner itself the update of block column k+1 is issued first, on the stream that also takes the REST of update k
// (an asynchronous backend: its bulk stream, most of the chip), the panel factorization behind it on the panel stream, the
// rest last: panel k+1 and the rest of update k run concurrently inside the rank, exactly like the two streams of the
// single-GPU driver (getrf.hip, getrf_lookahead, "mode 1").  Rounds 2-5 ran the update of block column k+1 on the panel
// stream's 32 CUs in front of the panel: ~1 ms per step on the critical chain (profiles/r05_bench_dryrun_dist_one_rank.json).
// Two panel buffers alternate; the pivots stay in the backend's memory until the end (no host synchronisation
// inside the loop).
template <class B> struct DistLu {
	typedef typename B::T T;
	typedef typename B::View View;

	static size_t hdr_scalars(long nints) { return ((size_t) nints * sizeof(int) + sizeof(T) - 1) / sizeof(T); }
	static size_t buf_scalars(long m, long nb) { return hdr_scalars(nb) + (size_t) m * (size_t) nb; }
	// [all pivots: m ints][panel buffer 0: nb pivots + packed panel][panel buffer 1]
	static size_t ws_scalars(long m, long nb) { return hdr_scalars(m) + 2 * buf_scalars(m, nb); }

	static size_t local_ncols(size_t n, size_t nb, int rank, int world)
	{
		const size_t nblk = (n + nb - 1) / nb;
		size_t cols = 0;
		for (size_t b = (size_t) rank; b < nblk; b += (size_t) world)
			cols += (b + 1) * nb <= n ? nb : n - b * nb;
		return cols;
	}

	// A_local: m x local_ncols (this rank's block columns, in increasing global order, contiguous).
	// panel_ws: >= ws_scalars(m, nb) scalars in the backend's memory space.
	// piv_host: min(m, n) ints, filled on every rank with ABSOLUTE pivot rows.
	static void run(B &be, View A_local, long m, long n, long nb, int rank, int world, T *panel_ws, int *piv_host)
	{
		const long size = m < n ? m : n;
		const long nblk = (size + nb - 1) / nb; // block columns that get factored
		const long nblk_all = (n + nb - 1) / nb;
		auto local_col0 = [&](long b) { // first local column of global block b (owned by this rank)
			long c = 0;
			for (long bb = rank; bb < b; bb += world)
				c += (bb + 1) * nb <= n ? nb : n - bb * nb;
			return c;
		};
		auto view = [&](long r0, long c0, long nr, long nc) {
			return View{A_local.p + r0 * A_local.rs + c0 * A_local.cs, nr, nc, A_local.rs, A_local.cs};
		};
		int *piv_all = reinterpret_cast<int *>(panel_ws); // pivots of every block, relative to their panel's row 0
		T *bufs = panel_ws + hdr_scalars(m);
		const size_t bsz = buf_scalars(m, nb);
		auto piv_of = [&](long k) { return reinterpret_cast<int *>(bufs + (size_t) (k & 1) * bsz); };
		auto panel_of = [&](long k) { return bufs + (size_t) (k & 1) * bsz + hdr_scalars(nb); };
		auto fw = [&](long k) { return (k * nb + nb <= size) ? nb : size - k * nb; }; // columns factored in block k
		auto bytes_of = [&](long k) { return (hdr_scalars(nb) + (size_t) (m - k * nb) * (size_t) fw(k)) * sizeof(T); };
		auto factor_and_pack = [&](long k) { // owner of block column k
			const long j0 = k * nb, w = fw(k), rows = m - j0, lc = local_col0(k);
			be.factor_panel(view(j0, lc, rows, w), piv_of(k));
			be.pack(view(j0, lc, rows, w), panel_of(k));
		};
		// applies panel k to block column b of this rank: interchanges everywhere, solve + update right of the panel
		auto update = [&](long k, long b) {
			const long j0 = k * nb, w = fw(k), rows = m - j0;
			const long wcols = (j0 + nb <= n) ? nb : n - j0; // columns block k really has (m < n tail)
			const int *piv = piv_of(k);
			View Lp{panel_of(k), rows, w, 1, rows}; // packed panel: unit lower trapezoid (U11 in its top triangle)
			const long bc0 = b * nb;
			const long bw = (bc0 + nb <= n) ? nb : n - bc0;
			const long lc = local_col0(b);
			if (b == k) {
				// the owner's panel columns were swapped by the panel factorization itself; a wider block
				// (m < n tail) still has columns to the right of the factored ones
				if (wcols > w) {
					View R = view(j0, lc + w, rows, wcols - w);
					be.laswp(R, piv, (int) w);
					be.trsm_unit_lower(View{Lp.p, w, w, Lp.rs, Lp.cs}, View{R.p, w, R.ncols, R.rs, R.cs});
					if (rows > w)
						be.gemm_sub(View{R.p + w * R.rs, rows - w, R.ncols, R.rs, R.cs},
							    View{Lp.p + w * Lp.rs, rows - w, w, Lp.rs, Lp.cs}, View{R.p, w, R.ncols, R.rs, R.cs});
				}
				return;
			}
			View Bk = view(j0, lc, rows, bw);
			be.laswp(Bk, piv, (int) w);
			if (b > k) {
				View top{Bk.p, w, bw, Bk.rs, Bk.cs};
				be.trsm_unit_lower(View{Lp.p, w, w, Lp.rs, Lp.cs}, top); // A01 <- L00^-1 A01 (factor.rs:98-107)
				if (rows > w)
					be.gemm_sub(View{Bk.p + w * Bk.rs, rows - w, bw, Bk.rs, Bk.cs},
						    View{Lp.p + w * Lp.rs, rows - w, w, Lp.rs, Lp.cs}, top); // A11 -= A10 A01 (:108-117)
			}
		};
		// The ROOT of a broadcast reads its own buffer: it does not wait for the transfer before it applies the panel, only
		// before the slot is written again (two steps later).  With one rank, and wherever a rank owns consecutive panels, the
		// transport is then off the chain "panel k+1 -> update of block column k+2" (round 6: every wait is still taken exactly
		// once, tests/test_dist_lu.py).
		bool sent[2] = {false, false}; // this rank's own broadcast from the slot has not been awaited yet
		auto wait_own = [&](int slot) {
			if (sent[slot]) {
				be.bcast_wait(slot);
				sent[slot] = false;
			}
		};
		if (nblk > 0) {
			if (rank == 0)
				factor_and_pack(0);
			be.bcast_begin(piv_of(0), bytes_of(0), 0, 0);
			sent[0] = rank == 0;
		}
		for (long k = 0; k < nblk; ++k) {
			if ((int) (k % world) != rank)
				be.bcast_wait((int) (k & 1));
			const bool ahead = k + 1 < nblk;
			const int next_owner = ahead ? (int) ((k + 1) % world) : -1;
			{ // local columns right of block k x remaining rows: how much trailing work this rank has beside the next panel
				long right = 0;
				for (long b = rank; b < nblk_all; b += world)
					if (b > k)
						right += (b * nb + nb <= n) ? nb : n - b * nb;
				be.step_begin((m - k * nb) * right, m - (k + 1) * nb); // trailing entries this rank updates in this step; rows of the next panel
			}
			// The rest of update k on ALL the local block columns at once: the columns of the blocks left of the panel are one
			// contiguous range of A_local (interchanges only), those of the blocks right of it another one (interchanges, ONE
			// triangular solve, ONE product) -- three operations per step instead of three per owned block column: fewer,
			// larger launches (the host was the bottleneck of a rank with many block columns, profiles/r02_dist_overlap.txt),
			// and the product runs on the large tiles.  Per column the arithmetic is the same as block by block.
			auto rest = [&]() {
				be.rest_begin();
				be.copy_ints(piv_all + k * nb, piv_of(k), (size_t) fw(k)); // (needed at the very end only: off the caller's chain)
				const bool skip_next = ahead && rank == next_owner; // block k + 1 is brought up to date by the look-ahead part
				long cl = 0, cr = 0, ctot = 0;			    // local columns: left of block k / up to the right range / all
				for (long b = rank; b < nblk_all; b += world) {
					const long bw = (b * nb + nb <= n) ? nb : n - b * nb;
					if (b < k)
						cl += bw;
					if (b <= k || (skip_next && b == k + 1))
						cr += bw;
					ctot += bw;
				}
				const long j0 = k * nb, w = fw(k), rows = m - j0;
				const int *piv = piv_of(k);
				View Lp{panel_of(k), rows, w, 1, rows};
				if (cl > 0)
					be.laswp(view(j0, 0, rows, cl), piv, (int) w); // factor.rs:127-185
				if (k % world == rank)
					update(k, k); // (only a block wider than its factored part has anything left to do)
				if (ctot > cr) {
					View Bk = view(j0, cr, rows, ctot - cr);
					be.laswp(Bk, piv, (int) w);
					View top{Bk.p, w, Bk.ncols, Bk.rs, Bk.cs};
					be.trsm_unit_lower(View{Lp.p, w, w, Lp.rs, Lp.cs}, top); // A01 <- L00^-1 A01 (factor.rs:98-107)
					if (rows > w)
						be.gemm_sub(View{Bk.p + w * Bk.rs, rows - w, Bk.ncols, Bk.rs, Bk.cs},
							    View{Lp.p + w * Lp.rs, rows - w, w, Lp.rs, Lp.cs}, top); // A11 -= A10 A01 (:108-117)
				}
				be.rest_end();
			};
			if (ahead && rank == next_owner) {
				// The look-ahead part is ISSUED first: it is a chain of latency-bound launches that has to start at once,
				// while issuing the rest of the update (dozens of launches per owned block column) keeps the host busy for
				// as long as the panel runs (profiles/r02_dist_overlap.txt: issued the other way round, the panel stream only
				// started when the host had finished queueing the rest -- the two streams ran one after the other).  The
				// caller's stream joins the panel stream only AFTER the rest has been queued, so a blocking transport that
				// synchronises in bcast_begin does not hold the rest back either.
				be.ahead_cols_begin();
				update(k, k + 1);
				be.ahead_cols_end();
				be.ahead_begin();
				wait_own((int) ((k + 1) & 1)); // (block k - 1 has left the buffer that panel k + 1 is packed into)
				factor_and_pack(k + 1);
				be.ahead_end();
				rest(); // runs beside the panel on an asynchronous backend
				be.ahead_join();
				be.bcast_begin(piv_of(k + 1), bytes_of(k + 1), next_owner, (int) ((k + 1) & 1));
				sent[(k + 1) & 1] = true;
			} else {
				if (ahead) { // post the receive before the updates: the transfer overlaps them
					wait_own((int) ((k + 1) & 1));
					be.bcast_begin(piv_of(k + 1), bytes_of(k + 1), next_owner, (int) ((k + 1) & 1));
				}
				rest();
			}
		}
		wait_own(0);
		wait_own(1);
		be.run_end();
		std::vector<int> rel((size_t) size);
		if (size > 0)
			be.to_host(rel.data(), piv_all, (size_t) size);
		for (long j = 0; j < size; ++j)
			piv_host[j] = (int) ((j / nb) * nb + rel[(size_t) j]);
	}
};

} // namespace fh
