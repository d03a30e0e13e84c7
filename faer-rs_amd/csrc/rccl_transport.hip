// Direct RCCL transport for the distributed drivers (dist_lu.h / dist_llt.h): ncclBroadcast on a DEDICATED stream,
// ordered against the calling thread's stream with events only -- no host synchronisation, no Python in the loop.
// (The other transport is the caller's callback, e.g. torch.distributed through faer-rs_amd/__init__.py.)
//
// librccl is opened with dlopen on first use: libfaer_hip.so itself keeps depending on libamdhip64 alone, and a
// single-GPU user never loads RCCL.  The communicator is created from a 128-byte ncclUniqueId that rank 0 obtains
// (faer_hip_rccl_unique_id) and the application ships to the other ranks by whatever means it already has.
#include <dlfcn.h>

#include <utility>
#include <vector>

#include "common.h"

using namespace fh;

namespace {

struct NcclUniqueId {
	char internal[128];
};
typedef void *NcclComm;
constexpr int NCCL_CHAR = 0; // ncclInt8 / ncclChar

struct RcclApi {
	void *handle = nullptr;
	int (*GetUniqueId)(NcclUniqueId *) = nullptr;
	int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
	int (*Broadcast)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
	int (*CommDestroy)(NcclComm) = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	int (*CommCount)(NcclComm, int *) = nullptr;
};

RcclApi *rccl_api()
{
	static RcclApi api;
	static bool tried = false;
	if (!tried) {
		tried = true;
		const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
		for (const char *n : names) {
			api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
			if (api.handle)
				break;
		}
		if (api.handle) {
			api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
			api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
			api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(dlsym(api.handle, "ncclBroadcast"));
			api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
			api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
			api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.handle, "ncclCommCount"));
			if (!api.GetUniqueId || !api.CommInitRank || !api.Broadcast || !api.CommDestroy)
				api.handle = nullptr;
		}
	}
	return api.handle ? &api : nullptr;
}

struct Rccl {
	NcclComm comm = nullptr;
	hipStream_t stream = nullptr;
	static constexpr int SLOTS = 8; // dist_llt.h: two panels x LLT_NCH chunks in flight (dist_lu.h uses slots 0 / 1)
	hipEvent_t ready = nullptr, done[SLOTS + 1] = {}; // slots of the drivers + the blocking form (index SLOTS)
	int rank = 0, world = 1;
	// statistics since the last faer_hip_rccl_stats: broadcasts issued, bytes, their device time.  The timing events are a
	// fixed ring, created on first use and destroyed with the transport: a caller that never asks for statistics holds at
	// most 2 * TIMED events however many panels it broadcasts (one pair per broadcast was an unbounded leak: ADVICE r03).
	static constexpr int TIMED = 64;
	hipEvent_t t0[TIMED] = {}, t1[TIMED] = {};
	bool pending[TIMED] = {};
	int next_timed = 0;
	double timed_ms = 0;
	double n_bcast = 0, bytes = 0;
	// folds the pair in slot i into timed_ms if its broadcast has finished (never waits)
	void harvest(int i)
	{
		if (!pending[i])
			return;
		float t = 0;
		const hipError_t e = hipEventElapsedTime(&t, t0[i], t1[i]);
		if (e == hipSuccess)
			timed_ms += t;
		else
			(void) hipGetLastError(); // not finished yet: this broadcast stays untimed
		pending[i] = false;
	}
};

void rccl_check(int rc, const char *what)
{
	if (rc != 0) {
		RcclApi *a = rccl_api();
		fprintf(stderr, "faer_hip: fatal: %s: %s\n", what, (a && a->GetErrorString) ? a->GetErrorString(rc) : "RCCL error");
		fflush(stderr);
		abort();
	}
}

void rccl_start(Rccl *r, void *buf, size_t bytes, int root, int slot)
{
	RcclApi *a = rccl_api();
	hipStream_t cur = ctx().stream;
	FH_HIP(hipEventRecord(r->ready, cur)); // the panel was packed on the caller's stream
	FH_HIP(hipStreamWaitEvent(r->stream, r->ready, 0));
	const int ti = r->next_timed;
	r->next_timed = (ti + 1) % Rccl::TIMED;
	r->harvest(ti); // the slot's previous broadcast (TIMED broadcasts ago)
	if (!r->t0[ti]) {
		FH_HIP(hipEventCreate(&r->t0[ti]));
		FH_HIP(hipEventCreate(&r->t1[ti]));
	}
	FH_HIP(hipEventRecord(r->t0[ti], r->stream));
	rccl_check(a->Broadcast(buf, buf, bytes, NCCL_CHAR, root, r->comm, r->stream), "ncclBroadcast");
	FH_HIP(hipEventRecord(r->t1[ti], r->stream));
	r->pending[ti] = true;
	r->n_bcast += 1;
	r->bytes += (double) bytes;
	FH_HIP(hipEventRecord(r->done[slot], r->stream));
}

void rccl_ibcast(void *user, void *buf, size_t bytes, int root, int slot)
{
	FH_CHECK(slot >= 0 && slot < Rccl::SLOTS, "rccl transport: slot out of range");
	rccl_start(static_cast<Rccl *>(user), buf, bytes, root, slot);
}
void rccl_wait(void *user, int slot)
{
	Rccl *r = static_cast<Rccl *>(user);
	FH_HIP(hipStreamWaitEvent(ctx().stream, r->done[slot], 0)); // stream ordered: the host does not block
}
void rccl_bcast(void *user, void *buf, size_t bytes, int root)
{
	Rccl *r = static_cast<Rccl *>(user);
	rccl_start(r, buf, bytes, root, Rccl::SLOTS);
	FH_HIP(hipStreamWaitEvent(ctx().stream, r->done[Rccl::SLOTS], 0));
}

} // namespace

namespace fh {
// dist.hip: the built-in transport's wait only needs ctx().stream, whichever stream that is
bool rccl_is_builtin_wait(FaerHipWaitFn fn) { return fn == rccl_wait; }
} // namespace fh

extern "C" {

int faer_hip_rccl_unique_id(void *out128)
{
	RcclApi *a = rccl_api();
	if (!a || !out128)
		return 1;
	NcclUniqueId id;
	memset(&id, 0, sizeof(id));
	if (a->GetUniqueId(&id) != 0)
		return 2;
	memcpy(out128, &id, sizeof(id));
	return 0;
}

void *faer_hip_rccl_create(const void *id128, int rank, int world_size)
{
	RcclApi *a = rccl_api();
	FH_CHECK(a != nullptr, "rccl transport: librccl.so could not be loaded");
	FH_CHECK(id128 != nullptr && world_size >= 1 && rank >= 0 && rank < world_size, "rccl transport: bad arguments");
	ctx(); // binds the thread to its device
	Rccl *r = new Rccl;
	r->rank = rank;
	r->world = world_size;
	NcclUniqueId id;
	memcpy(&id, id128, sizeof(id));
	rccl_check(a->CommInitRank(&r->comm, world_size, id, rank), "ncclCommInitRank");
	FH_HIP(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
	FH_HIP(hipEventCreateWithFlags(&r->ready, hipEventDisableTiming));
	for (hipEvent_t &e : r->done)
		FH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	return r;
}

FaerHipComm faer_hip_rccl_comm(void *handle)
{
	FH_CHECK(handle != nullptr, "rccl transport: NULL handle");
	Rccl *r = static_cast<Rccl *>(handle);
	FaerHipComm c;
	memset(&c, 0, sizeof(c));
	c.rank = r->rank;
	c.world_size = r->world;
	c.bcast = rccl_bcast;
	c.user = r;
	c.ibcast = rccl_ibcast;
	c.wait = rccl_wait;
	return c;
}

// out4: {ranks the communicator reports (ncclCommCount), broadcasts, bytes, device milliseconds inside ncclBroadcast} since
// the previous call; synchronises the transport's stream
void faer_hip_rccl_stats(void *handle, double *out4)
{
	FH_CHECK(handle != nullptr && out4 != nullptr, "rccl transport: NULL argument");
	Rccl *r = static_cast<Rccl *>(handle);
	RcclApi *a = rccl_api();
	int cnt = -1;
	if (a && a->CommCount && r->comm)
		(void) a->CommCount(r->comm, &cnt);
	FH_HIP(hipStreamSynchronize(r->stream));
	for (int i = 0; i < Rccl::TIMED; ++i)
		r->harvest(i);
	const double ms = r->timed_ms;
	r->timed_ms = 0;
	out4[0] = (double) cnt;
	out4[1] = r->n_bcast;
	out4[2] = r->bytes;
	out4[3] = ms;
	r->n_bcast = 0;
	r->bytes = 0;
}

void faer_hip_rccl_destroy(void *handle)
{
	if (!handle)
		return;
	Rccl *r = static_cast<Rccl *>(handle);
	RcclApi *a = rccl_api();
	(void) hipStreamSynchronize(r->stream);
	if (a && r->comm)
		(void) a->CommDestroy(r->comm);
	(void) hipEventDestroy(r->ready);
	for (hipEvent_t e : r->done)
		(void) hipEventDestroy(e);
	for (int i = 0; i < Rccl::TIMED; ++i)
		if (r->t0[i]) {
			(void) hipEventDestroy(r->t0[i]);
			(void) hipEventDestroy(r->t1[i]);
		}
	(void) hipStreamDestroy(r->stream);
	delete r;
}
}
