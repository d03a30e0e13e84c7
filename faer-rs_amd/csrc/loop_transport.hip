// Loop-back transport: the ranks of a distributed factorization as THREADS of one process sharing one GPU.
//
// The boxes this library is developed on have one GPU; RCCL refuses two ranks on one device, so the stream schedule a rank runs over
// the BUILT-IN transport (dist.hip `quiet`: broadcasts start from the stream that packed or last read the buffer, waits are taken by
// the internal streams, nothing on the caller's stream between the first step and the end) could only run with ONE rank.  This
// transport has the same contract as rccl_transport.hip -- ibcast / wait only need ctx().stream of the calling thread, whichever
// stream that is -- and moves the data with device-to-device copies between the ranks' buffers, so tests/test_gpu_dist_threads.py runs
// that schedule with 2 and 3 ranks against the oracle.  It is test infrastructure inside the product library (the transport has to see
// ctx().stream), never a multi-GPU transport: every rank must be a thread of the calling process bound to the same device.
//
// Protocol per (slot, generation) -- every rank calls ibcast for every broadcast, in the same order (the drivers' wire plan):
//   root      records `ready` on ctx().stream (behind the pack), publishes {buffer, ready event};
//   receiver  waits ON THE HOST until the root has published this generation, then on its transport stream: wait for its own
//             ctx().stream (the buffer's last readers), wait for the root's `ready`, copy, record `copied`; publishes the event;
//   wait      receiver: ctx().stream waits for its `copied`; root (before it reuses the buffer): waits on the host until every receiver
//             has published, then ctx().stream waits for their `copied` events.
// Host waits never deadlock: a rank only waits for a call the others make earlier in the common order than anything they wait for.
#include "common.h"

#include <condition_variable>
#include <mutex>

using namespace fh;

namespace {

constexpr int LT_SLOTS = 9; // FAER_HIP_COMM_SLOTS + the blocking form
constexpr int LT_MAXR = 8;
constexpr int LT_RING = 16; // generations of one slot that may be in flight between the fastest and the slowest rank

struct LoopGroup {
	int world = 1;
	std::mutex mu;
	std::condition_variable cv;
	struct Entry {
		long gen = 0; // generation this entry describes (0: none yet)
		const void *root_buf = nullptr;
		hipEvent_t root_ready = nullptr;
		int copied_cnt = 0;
		hipEvent_t copied[LT_MAXR] = {};
	} ring[LT_SLOTS][LT_RING];
};

struct LoopRank {
	LoopGroup *g = nullptr;
	int rank = 0;
	long gen[LT_SLOTS] = {};      // broadcasts this rank has taken part in, per slot
	bool was_root[LT_SLOTS] = {}; // ... and its role in the last one
	hipStream_t ts = nullptr;
	hipEvent_t my_ready = nullptr;
	hipEvent_t ev[LT_SLOTS][LT_RING] = {}; // root: `ready`, receiver: `copied` of generation g in slot s
	double n_bcast = 0, bytes = 0;
	hipEvent_t event(int slot, long g)
	{
		hipEvent_t &e = ev[slot][g % LT_RING];
		if (!e)
			FH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		return e;
	}
};

void loop_start(LoopRank *r, void *buf, size_t bytes, int root, int slot)
{
	LoopGroup *G = r->g;
	FH_CHECK(slot >= 0 && slot < LT_SLOTS && root >= 0 && root < G->world, "loop-back transport: bad slot / root");
	const long g = ++r->gen[slot];
	LoopGroup::Entry &en = G->ring[slot][g % LT_RING];
	hipStream_t cur = ctx().stream;
	hipEvent_t e = r->event(slot, g);
	r->was_root[slot] = r->rank == root;
	r->n_bcast += 1;
	r->bytes += (double) bytes;
	if (r->rank == root) {
		FH_HIP(hipEventRecord(e, cur)); // the panel was packed on this stream
		std::lock_guard<std::mutex> lk(G->mu);
		FH_CHECK(en.gen < g, "loop-back transport: a slot's ring was overrun");
		en.gen = g;
		en.root_buf = buf;
		en.root_ready = e;
		en.copied_cnt = 0;
		G->cv.notify_all();
		return;
	}
	const void *src;
	hipEvent_t rdy;
	{
		std::unique_lock<std::mutex> lk(G->mu);
		G->cv.wait(lk, [&] { return en.gen >= g; });
		FH_CHECK(en.gen == g, "loop-back transport: a slot's ring was overrun");
		src = en.root_buf;
		rdy = en.root_ready;
	}
	FH_HIP(hipEventRecord(r->my_ready, cur)); // the buffer's last readers on this rank
	FH_HIP(hipStreamWaitEvent(r->ts, r->my_ready, 0));
	FH_HIP(hipStreamWaitEvent(r->ts, rdy, 0));
	FH_HIP(hipMemcpyAsync(buf, src, bytes, hipMemcpyDeviceToDevice, r->ts));
	FH_HIP(hipEventRecord(e, r->ts));
	{
		std::lock_guard<std::mutex> lk(G->mu);
		en.copied[r->rank] = e;
		en.copied_cnt += 1;
		G->cv.notify_all();
	}
}

void loop_ibcast(void *user, void *buf, size_t bytes, int root, int slot) { loop_start(static_cast<LoopRank *>(user), buf, bytes, root, slot); }

void loop_wait(void *user, int slot)
{
	LoopRank *r = static_cast<LoopRank *>(user);
	LoopGroup *G = r->g;
	const long g = r->gen[slot];
	if (g == 0)
		return;
	hipStream_t cur = ctx().stream;
	if (!r->was_root[slot]) {
		FH_HIP(hipStreamWaitEvent(cur, r->event(slot, g), 0));
		return;
	}
	LoopGroup::Entry &en = G->ring[slot][g % LT_RING];
	hipEvent_t evs[LT_MAXR];
	{
		std::unique_lock<std::mutex> lk(G->mu);
		FH_CHECK(en.gen == g, "loop-back transport: a slot's ring was overrun");
		G->cv.wait(lk, [&] { return en.copied_cnt >= G->world - 1; });
		for (int q = 0; q < G->world; ++q)
			evs[q] = q == r->rank ? nullptr : en.copied[q];
	}
	for (int q = 0; q < G->world; ++q)
		if (evs[q])
			FH_HIP(hipStreamWaitEvent(cur, evs[q], 0));
}

void loop_bcast(void *user, void *buf, size_t bytes, int root)
{
	loop_start(static_cast<LoopRank *>(user), buf, bytes, root, LT_SLOTS - 1);
	loop_wait(user, LT_SLOTS - 1);
}

} // namespace

namespace fh {
bool loop_is_builtin_wait(FaerHipWaitFn fn) { return fn == loop_wait; }
} // namespace fh

extern "C" {

void *faer_hip_loopback_group_create(int world_size)
{
	FH_CHECK(world_size >= 1 && world_size <= LT_MAXR, "loop-back transport: 1 .. 8 ranks");
	LoopGroup *G = new LoopGroup;
	G->world = world_size;
	return G;
}

// per rank, on the thread that will run the rank (binds the thread to its device and creates the rank's transport stream)
void *faer_hip_loopback_rank_create(void *group, int rank)
{
	LoopGroup *G = static_cast<LoopGroup *>(group);
	FH_CHECK(G != nullptr && rank >= 0 && rank < G->world, "loop-back transport: bad arguments");
	ctx();
	LoopRank *r = new LoopRank;
	r->g = G;
	r->rank = rank;
	FH_HIP(hipStreamCreateWithFlags(&r->ts, hipStreamNonBlocking));
	FH_HIP(hipEventCreateWithFlags(&r->my_ready, hipEventDisableTiming));
	return r;
}

FaerHipComm faer_hip_loopback_comm(void *rank_handle)
{
	FH_CHECK(rank_handle != nullptr, "loop-back transport: NULL handle");
	LoopRank *r = static_cast<LoopRank *>(rank_handle);
	FaerHipComm c;
	memset(&c, 0, sizeof(c));
	c.rank = r->rank;
	c.world_size = r->g->world;
	c.bcast = loop_bcast;
	c.user = r;
	c.ibcast = loop_ibcast;
	c.wait = loop_wait;
	return c;
}

// out2: {broadcasts this rank took part in, their bytes}
void faer_hip_loopback_stats(void *rank_handle, double *out2)
{
	LoopRank *r = static_cast<LoopRank *>(rank_handle);
	out2[0] = r->n_bcast;
	out2[1] = r->bytes;
}

void faer_hip_loopback_rank_destroy(void *rank_handle)
{
	if (!rank_handle)
		return;
	LoopRank *r = static_cast<LoopRank *>(rank_handle);
	(void) hipStreamSynchronize(r->ts);
	(void) hipEventDestroy(r->my_ready);
	for (auto &row : r->ev)
		for (hipEvent_t e : row)
			if (e)
				(void) hipEventDestroy(e);
	(void) hipStreamDestroy(r->ts);
	delete r;
}

void faer_hip_loopback_group_destroy(void *group) { delete static_cast<LoopGroup *>(group); }
}
