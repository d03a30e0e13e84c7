// Per-thread runtime context: device binding, stream, scratch pool, pointer classification.
#include <type_traits>

#include "common.h"

namespace fh {

static thread_local Ctx g_ctx;
static const hipStream_t ANY_STREAM = reinterpret_cast<hipStream_t>(~(uintptr_t) 0);

Ctx &ctx()
{
	g_ctx.ensure_device();
	return g_ctx;
}

void ctx_shutdown()
{
	Ctx &c = g_ctx; // no ensure_device(): must be callable (as a no-op) on a machine without a GPU
	if (c.device < 0)
		return;
	(void) hipStreamSynchronize(c.stream);
	for (hipEvent_t e : c.la_events)
		(void) hipEventDestroy(e);
	c.la_events.clear();
	c.la_next_event = 0;
	if (c.la_state > 0) {
		(void) hipStreamSynchronize(c.la_bulk);
		(void) hipStreamSynchronize(c.la_panel);
		(void) hipStreamDestroy(c.la_bulk);
		(void) hipStreamDestroy(c.la_panel);
		c.la_bulk = c.la_panel = nullptr;
	}
	c.la_state = 0;
	if (c.qr_side[0]) {
		(void) hipStreamSynchronize(c.qr_side[0]);
		(void) hipStreamDestroy(c.qr_side[0]);
		c.qr_side[0] = c.qr_side[1] = nullptr;
		for (hipEvent_t &e : c.qr_ev) {
			(void) hipEventDestroy(e);
			e = nullptr;
		}
	}
	if (c.host_ints) {
		(void) hipHostFree(c.host_ints);
		c.host_ints = nullptr;
	}
}

int Ctx::stream_cus()
{
	if (ncu == 0) {
		hipDeviceProp_t prop;
		FH_HIP(hipGetDeviceProperties(&prop, device));
		ncu = prop.multiProcessorCount;
	}
	if (la_state > 0 && stream == la_panel)
		return la_panel_cus;
	if (la_state > 0 && stream == la_bulk)
		return ncu - la_panel_cus;
	return ncu;
}

void Ctx::ensure_device()
{
	if (device >= 0)
		return;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	// No CPU fallback by design: the product path needs a gfx950 device.
	FH_CHECK(e == hipSuccess && n > 0, "no HIP device available (libfaer_hip has no CPU fallback)");
	int d = 0;
	FH_HIP(hipGetDevice(&d));
	hipDeviceProp_t prop;
	FH_HIP(hipGetDeviceProperties(&prop, d));
	FH_CHECK(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "libfaer_hip is built for gfx950 (MI355X) only");
	device = d;
}

int *Ctx::pinned_ints()
{
	if (!host_ints)
		FH_HIP(hipHostMalloc(reinterpret_cast<void **>(&host_ints), 16 * sizeof(int), hipHostMallocDefault));
	return host_ints;
}

void Ctx::qr_side_streams()
{
	if (qr_side[0])
		return;
	// ONE stream for both roles: with the caller's stream and the two CU-masked ones that makes four -- a fifth stream
	// shares a hardware queue with another one (the runtime maps streams onto 4 of them by default) and whatever was
	// created last ran 1.3 - 6 x slower, on every stream (profiles/r03_qr_stream_order.txt)
	FH_HIP(hipStreamCreateWithFlags(&qr_side[0], hipStreamNonBlocking));
	qr_side[1] = qr_side[0];
	for (hipEvent_t &e : qr_ev)
		FH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
}

bool Ctx::lookahead_streams()
{
	if (la_state != 0)
		return la_state > 0;
	qr_side_streams(); // (see common.h: the order of creation matters)
	la_state = -1;
	if (getenv("FAER_HIP_NO_LOOKAHEAD"))
		return false;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess)
		return false;
	const int ncu = prop.multiProcessorCount;
	if (la_panel_cus < 8 || la_panel_cus > ncu / 2 || ncu > 1024)
		return false;
	// CU i of the mask is enabled by bit i; the panel stream gets the LAST `la_panel_cus` CUs.  Measured
	// (profiles/r01_exp_masks.txt): workgroups of a masked stream still land on all 8 XCDs for every layout -- the
	// dispatcher deals workgroups to the XCDs round robin and the mask only selects CUs inside each, so a stream cannot be
	// confined to one XCD (and its L2); every 8th CU instead leaves XCDs without panel / bulk CUs and is 20 % slower, the
	// first CUs instead of the last ones make no difference.
	const int layout = 0;
	uint32_t mb[32], mp[32];
	memset(mb, 0, sizeof(mb));
	memset(mp, 0, sizeof(mp));
	for (int i = 0; i < ncu; ++i) {
		const bool panel = layout == 0 ? i >= ncu - la_panel_cus : layout == 1 ? i < la_panel_cus : i % 8 == 7;
		uint32_t *m = panel ? mp : mb;
		m[i / 32] |= 1u << (i % 32);
	}
	const uint32_t words = (uint32_t) ((ncu + 31) / 32);
	hipStream_t b = nullptr, p = nullptr;
	if (hipExtStreamCreateWithCUMask(&b, words, mb) != hipSuccess) {
		(void) hipGetLastError();
		return false;
	}
	if (hipExtStreamCreateWithCUMask(&p, words, mp) != hipSuccess) {
		(void) hipGetLastError();
		(void) hipStreamDestroy(b);
		return false;
	}
	la_bulk = b;
	la_panel = p;
	la_state = 1;
	return true;
}

// debugging aid: XCC id (low 4 bits) and HW_ID of the CU each block of a small grid lands on
__global__ void xcc_probe_kernel(unsigned *out)
{
	unsigned xcc, hw;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	// keep the block resident for a while so that a small grid spreads over the enabled CUs
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	while (__builtin_amdgcn_s_memtime() - t0 < 20000)
		__builtin_amdgcn_s_sleep(8);
	if (threadIdx.x == 0) {
		out[2 * blockIdx.x] = xcc;
		out[2 * blockIdx.x + 1] = hw;
	}
}

void debug_stream_xcc(int which, int nblocks, unsigned *out_host)
{
	Ctx &c = ctx();
	hipStream_t s = c.stream;
	if (which != 0) {
		FH_CHECK(c.lookahead_streams(), "debug_stream_xcc: look-ahead streams unavailable");
		s = which == 1 ? c.la_bulk : c.la_panel;
	}
	unsigned *d = nullptr;
	FH_HIP(hipMalloc(&d, (size_t) nblocks * 2 * sizeof(unsigned)));
	hipLaunchKernelGGL(xcc_probe_kernel, dim3(nblocks), dim3(512), 0, s, d);
	FH_HIP(hipGetLastError());
	FH_HIP(hipStreamSynchronize(s));
	FH_HIP(hipMemcpy(out_host, d, (size_t) nblocks * 2 * sizeof(unsigned), hipMemcpyDeviceToHost));
	FH_HIP(hipFree(d));
}

// Box-population probe: two resident workgroups (the dispatcher deals consecutive workgroups to different
// XCDs) bounce a data-tagged granule (xwg.h, recipe R2) `iters` times; returns microseconds per one-way hand-off on an otherwise
// idle chip.  The latency-bound kernels (LU panel, substitution leaves) track this number: ~0.8 us on the fast boxes of the pool.
__global__ void xwg_hop_kernel(unsigned long long *gran, int iters, int *status)
{
	const int me = blockIdx.x, other = 1 - me;
	if (threadIdx.x != 0)
		return;
	for (int r = 1; r <= iters; ++r) {
		if (me == 0)
			__hip_atomic_store(gran + 0, ((unsigned long long) r << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		bool ok = false;
		for (int spin = 0; spin < (1 << 22); ++spin) {
			const unsigned long long v = __hip_atomic_load(gran + (me == 0 ? 16 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if ((unsigned) (v >> 32) == (unsigned) r) {
				ok = true;
				break;
			}
		}
		if (!ok) {
			atomicExch(status, 1);
			return;
		}
		if (me == 1)
			__hip_atomic_store(gran + 16, ((unsigned long long) r << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		(void) other;
	}
}

double xwg_hop_us(int iters)
{
	Ctx &c = ctx();
	if (iters < 1)
		iters = 1;
	Scratch buf(512);
	FH_HIP(hipMemsetAsync(buf.p, 0, 512, c.stream));
	unsigned long long *g = buf.as<unsigned long long>();
	int *st = reinterpret_cast<int *>(g + 32);
	hipEvent_t e0, e1;
	FH_HIP(hipEventCreate(&e0));
	FH_HIP(hipEventCreate(&e1));
	hipLaunchKernelGGL(xwg_hop_kernel, dim3(2), dim3(64), 0, c.stream, g, 8, st); // warm (code fetch)
	FH_HIP(hipMemsetAsync(buf.p, 0, 512, c.stream));
	FH_HIP(hipEventRecord(e0, c.stream));
	hipLaunchKernelGGL(xwg_hop_kernel, dim3(2), dim3(64), 0, c.stream, g, iters, st);
	FH_HIP(hipEventRecord(e1, c.stream));
	FH_HIP(hipEventSynchronize(e1));
	float ms = 0;
	FH_HIP(hipEventElapsedTime(&ms, e0, e1));
	int h = 0;
	FH_HIP(hipMemcpy(&h, st, sizeof(int), hipMemcpyDeviceToHost));
	FH_HIP(hipEventDestroy(e0));
	FH_HIP(hipEventDestroy(e1));
	if (h != 0)
		return -1.0; // (the two workgroups were not resident together: a busy GPU)
	return (double) ms * 1e3 / (2.0 * iters);
}

std::atomic<int> g_lend_cus{0}; // (measured: no gain, see potrf.hip / getrf.hip -- off by default)

hipEvent_t Ctx::prof_event()
{
	if (!prof_pool.empty()) {
		hipEvent_t e = prof_pool.back();
		prof_pool.pop_back();
		return e;
	}
	hipEvent_t e;
	FH_HIP(hipEventCreate(&e));
	return e;
}

// out: PROF_CLASSES x {milliseconds inside the class's launches, launches, units}; the caller has synchronised.
// spans (optional): per recorded launch {class, ms, units, d0, d1, d2, d3 + 16 * stream id, start in ms after the first recorded launch}
void prof_collect(double *out, double *spans, size_t cap, size_t *nspans)
{
	Ctx &c = ctx();
	for (int i = 0; i < Ctx::PROF_CLASSES * 3; ++i)
		out[i] = 0.0;
	size_t ns = 0;
	hipEvent_t first = nullptr;
	for (Ctx::ProfSpan &sp : c.prof_spans) {
		float ms = 0;
		if (sp.cls >= 0 && sp.cls < Ctx::PROF_CLASSES) {
			if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
				out[3 * sp.cls + 0] += ms;
				out[3 * sp.cls + 1] += 1.0;
				out[3 * sp.cls + 2] += sp.units;
				if (!first)
					first = sp.a;
				if (spans && ns < cap) {
					float t0 = 0;
					if (hipEventElapsedTime(&t0, first, sp.a) != hipSuccess) {
						(void) hipGetLastError();
						t0 = 0;
					}
					double *r = spans + 8 * ns++;
					r[0] = sp.cls;
					r[1] = ms;
					r[2] = sp.units;
					r[3] = (double) sp.d[0];
					r[4] = (double) sp.d[1];
					r[5] = (double) sp.d[2];
					r[6] = (double) sp.d[3] + 65536.0 * sp.sid;
					r[7] = t0;
				}
			} else {
				// spans of an unfinished profile (prof_begin drops them without synchronising): hipErrorNotReady would
				// stay behind as the thread's last error and fail the next launch check
				(void) hipGetLastError();
			}
		}
		c.prof_pool.push_back(sp.a);
		c.prof_pool.push_back(sp.b);
	}
	c.prof_spans.clear();
	if (nspans)
		*nspans = ns;
}

hipEvent_t Ctx::next_event()
{
	if (la_next_event == la_events.size()) {
		hipEvent_t e;
		FH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		la_events.push_back(e);
	}
	return la_events[la_next_event++];
}

void *Ctx::alloc(size_t bytes)
{
	if (bytes == 0)
		bytes = 256;
	bytes = (bytes + 255) & ~(size_t) 255;
	// best fit among the free buffers last used on THIS stream (a buffer released by host code may still be in
	// use by queued kernels: handing it to work on the same stream is ordered, to another stream it is a race)
	int best = -1;
	for (size_t i = 0; i < pool.size(); ++i)
		if (!pool[i].used && (pool[i].owner == stream || pool[i].owner == ANY_STREAM) && pool[i].bytes >= bytes &&
		    (best < 0 || pool[i].bytes < pool[best].bytes))
			best = (int) i;
	if (best >= 0 && pool[best].bytes <= 2 * bytes + (1 << 20)) {
		pool[best].used = true;
		pool[best].owner = stream;
		return pool[best].p;
	}
	// No buffer free on THIS stream.  Free buffers last used on OTHER streams become reusable once that work has
	// finished: poll (never wait) before growing the pool, so that a caller that changes streams
	// (faer_hip_set_stream) or paths that never reach quiesce() do not accumulate one set of buffers per stream.
	{
		std::vector<hipStream_t> idle, busy;
		for (auto &b : pool) {
			if (b.used || b.owner == stream || b.owner == ANY_STREAM)
				continue;
			// only streams this library knows to be alive are polled here: its own two and the null stream (a stream of
			// the caller may have been destroyed since; those are polled in set_stream(), while the caller still holds them)
			if (!(b.owner == nullptr || (la_state > 0 && (b.owner == la_bulk || b.owner == la_panel))))
				continue;
			bool known = false, is_idle = false;
			for (hipStream_t q : idle)
				if (q == b.owner)
					known = is_idle = true;
			for (hipStream_t q : busy)
				if (q == b.owner)
					known = true;
			if (!known) {
				const hipError_t qe = hipStreamQuery(b.owner);
				if (qe != hipSuccess && qe != hipErrorNotReady)
					(void) hipGetLastError(); // e.g. a stream the caller destroyed: leave its buffers alone
				is_idle = qe == hipSuccess;
				(is_idle ? idle : busy).push_back(b.owner);
			}
			if (is_idle)
				b.owner = ANY_STREAM;
		}
		if (!idle.empty()) {
			best = -1;
			for (size_t i = 0; i < pool.size(); ++i)
				if (!pool[i].used && pool[i].owner == ANY_STREAM && pool[i].bytes >= bytes && (best < 0 || pool[i].bytes < pool[best].bytes))
					best = (int) i;
			if (best >= 0 && pool[best].bytes <= 2 * bytes + (1 << 20)) {
				pool[best].used = true;
				pool[best].owner = stream;
				return pool[best].p;
			}
		}
	}
	void *p = nullptr;
	hipError_t e = hipMalloc(&p, bytes);
	if (e != hipSuccess) {
		// free cached buffers and retry once
		FH_HIP(hipStreamSynchronize(stream));
		for (size_t i = 0; i < pool.size();) {
			if (!pool[i].used) {
				(void) hipFree(pool[i].p);
				pool.erase(pool.begin() + i);
			} else
				++i;
		}
		FH_HIP(hipMalloc(&p, bytes));
	}
	pool.push_back(Buf{p, bytes, true, stream});
	return p;
}

// the caller moves to another stream: its free buffers can be handed to any stream as soon as the old stream is idle
// (polled, never waited for; the caller still owns `stream` at this point, so the query is safe)
void Ctx::set_stream(hipStream_t s)
{
	if (s == stream)
		return;
	if (device >= 0) {
		bool any = false;
		for (auto &b : pool)
			any = any || (!b.used && b.owner == stream);
		if (any && hipStreamQuery(stream) == hipSuccess) {
			for (auto &b : pool)
				if (!b.used && b.owner == stream)
					b.owner = ANY_STREAM;
		} else {
			(void) hipGetLastError();
		}
	}
	stream = s;
}

void Ctx::quiesce()
{
	for (auto &b : pool)
		if (!b.used)
			b.owner = ANY_STREAM;
}

void Ctx::release(void *p)
{
	for (auto &b : pool)
		if (b.p == p) {
			b.used = false;
			return;
		}
	die("release of unknown scratch buffer", __FILE__, __LINE__);
}

bool is_device_ptr(const void *p)
{
	if (!p)
		return false;
	hipPointerAttribute_t attr;
	hipError_t e = hipPointerGetAttributes(&attr, p);
	if (e != hipSuccess) {
		(void) hipGetLastError(); // plain host memory: clear the sticky error
		return false;
	}
	return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

} // namespace fh
