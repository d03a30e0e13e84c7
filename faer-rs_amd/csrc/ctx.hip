// Per-thread runtime context: device binding, stream, scratch pool, pointer classification.
#include <type_traits>

#include "common.h"

namespace fh {

static thread_local Ctx g_ctx;

Ctx &ctx()
{
	g_ctx.ensure_device();
	return g_ctx;
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

void *Ctx::alloc(size_t bytes)
{
	if (bytes == 0)
		bytes = 256;
	bytes = (bytes + 255) & ~(size_t) 255;
	// best fit among free buffers
	int best = -1;
	for (size_t i = 0; i < pool.size(); ++i)
		if (!pool[i].used && pool[i].bytes >= bytes && (best < 0 || pool[i].bytes < pool[best].bytes))
			best = (int) i;
	if (best >= 0 && pool[best].bytes <= 2 * bytes + (1 << 20)) {
		pool[best].used = true;
		return pool[best].p;
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
	pool.push_back(Buf{p, bytes, true});
	return p;
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
