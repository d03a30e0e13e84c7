// All-to-all exchange among G resident workgroups, as the LU panel kernel does it once per column -- without the kernel around it.
//   hipcc --offload-arch=gfx950 -O2 -o tools/xwg_a2a_probe tools/xwg_a2a_probe.hip && tools/xwg_a2a_probe
// Every round: each workgroup publishes a 64-byte header (tagged granules, write-through stores), then waits until it has seen the
// headers of ALL G workgroups, then "computes" for D shader cycles.  Reported: microseconds per round for several protocols and D.
//   mode 0  one record per workgroup, everybody sweeps all records (2 x 16-byte loads per lane), re-issued until complete
//   mode 1  the same with TWO sweeps in flight (the second issued before the first is checked)
//   mode 2  push: every workgroup stores its header into the inbox of every consumer; a consumer polls its own 2 KB inbox
//   mode 3  one 8-byte granule per workgroup instead of 64 bytes (lane t reads record t: one load)
//   mode 4  mode 3 + reduction by atomics: atomicMax of {value, g} into one word + atomicAdd of a counter; poll the counter
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                                              \
	do {                                                                                                               \
		hipError_t e_ = (x);                                                                                       \
		if (e_ != hipSuccess) {                                                                                    \
			fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                        \
			exit(1);                                                                                           \
		}                                                                                                          \
	} while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int STRIDE = 320; // bytes between records (as in lu_wpanel.h)
constexpr int GMAX = 32;
constexpr int SPIN = 1 << 20;

static __device__ __forceinline__ u32x4 ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
	return __builtin_amdgcn_raw_buffer_load_b128(r, (int) voff, (int) soff, 16);
}
static __device__ __forceinline__ void st(u32x4 q, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
	__builtin_amdgcn_raw_buffer_store_b128(q, r, (int) voff, (int) soff, 16);
}
static __device__ __forceinline__ void busy(unsigned long long cycles)
{
	if (cycles == 0)
		return;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	while (__builtin_amdgcn_s_memtime() - t0 < cycles) {
	}
}

__global__ __launch_bounds__(256) void a2a_kernel(unsigned char *ws, int iters, int mode, unsigned long long D, int *status, unsigned long long *sink)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int g = blockIdx.x, G = gridDim.x;
	if (wave != 0)
		return;
	const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc((void *) ws, 0, 1 << 22, 0x00020000);
	unsigned long long acc = 0;
	bool dead = false;
	for (int r = 1; r <= iters && !dead; ++r) {
		const unsigned tag = (unsigned) r;
		const unsigned par = (unsigned) (r & 1);
		if (mode == 0 || mode == 1) {
			if (lane < 4) {
				u32x4 q = {tag, (unsigned) (g * 7 + r), tag, (unsigned) lane};
				st(q, hr, (unsigned) (lane * 16), (par * GMAX + g) * STRIDE);
			}
			const unsigned voff = (unsigned) ((lane >> 2) * STRIDE + (lane & 3) * 16), soff = par * GMAX * STRIDE;
			const int t = lane >> 2;
			u32x4 a0 = ld(hr, voff, soff), a1 = ld(hr, voff + 16u * STRIDE, soff);
			u32x4 b0 = a0, b1 = a1;
			if (mode == 1) {
				b0 = ld(hr, voff, soff);
				b1 = ld(hr, voff + 16u * STRIDE, soff);
			}
			for (int spin = 0;; ++spin) {
				bool ok = (t >= G || (a0.x == tag && a0.z == tag)) && (t + 16 >= G || (a1.x == tag && a1.z == tag));
				if (__all(ok)) {
					acc += a0.y + a1.y;
					break;
				}
				if (spin > SPIN) {
					dead = true;
					break;
				}
				asm volatile("" ::: "memory"); // (the buffer loads below must be re-issued, not hoisted)
				__builtin_amdgcn_s_sleep(1);
				if (mode == 1) {
					a0 = b0;
					a1 = b1;
					b0 = ld(hr, voff, soff);
					b1 = ld(hr, voff + 16u * STRIDE, soff);
				} else {
					a0 = ld(hr, voff, soff);
					a1 = ld(hr, voff + 16u * STRIDE, soff);
				}
			}
		} else if (mode == 5 || mode == 6) {
			// mode 5: 32-byte records (two 16-byte granules, lanes 2 t and 2 t + 1), ONE load instruction sweeps all 32 records
			// mode 6: 16-byte records, lane t reads record t
			const unsigned rb = mode == 5 ? 32u : 16u, rstride = 128u;
			if (lane < (mode == 5 ? 2 : 1)) {
				u32x4 q = {tag, (unsigned) (g * 7 + r), tag, (unsigned) lane};
				st(q, hr, (unsigned) (lane * 16), (par * GMAX + g) * rstride);
			}
			const int t = mode == 5 ? lane >> 1 : lane;
			const unsigned voff = (unsigned) (t * rstride + (mode == 5 ? (lane & 1) * 16 : 0)), soff = par * GMAX * rstride;
			(void) rb;
			u32x4 a0 = ld(hr, voff, soff);
			for (int spin = 0;; ++spin) {
				bool ok = t >= G || (a0.x == tag && a0.z == tag);
				if (__all(ok)) {
					acc += a0.y;
					break;
				}
				if (spin > SPIN) {
					dead = true;
					break;
				}
				asm volatile("" ::: "memory");
				a0 = ld(hr, voff, soff);
			}
		} else if (mode == 8 || mode == 9) {
			// 16-byte records through GLOBAL instructions (scalar base + 32-bit lane offset, sc1) instead of buffer instructions;
			// mode 9: 64-byte records, two loads per sweep (the LU kernel's layout)
			const unsigned rstride = mode == 8 ? 128u : (unsigned) STRIDE;
			const unsigned long long base = (unsigned long long) ws + (unsigned long long) par * GMAX * rstride;
			if (lane < (mode == 8 ? 1 : 4)) {
				u32x4 q = {tag, (unsigned) (g * 7 + r), tag, (unsigned) lane};
				const unsigned so = (unsigned) (g * rstride + lane * 16);
				asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(so), "v"(q), "s"(base) : "memory");
			}
			const int t = mode == 8 ? lane : lane >> 2;
			const unsigned vo = mode == 8 ? (unsigned) (lane * rstride) : (unsigned) ((lane >> 2) * rstride + (lane & 3) * 16);
			for (int spin = 0;; ++spin) {
				u32x4 a0, a1 = {tag, 0u, tag, 0u};
				asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(a0) : "v"(vo), "s"(base) : "memory");
				if (mode == 9) {
					const unsigned vo1 = vo + 16u * rstride;
					asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(a1) : "v"(vo1), "s"(base) : "memory");
				}
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				bool ok = (t >= G || (a0.x == tag && a0.z == tag)) && (mode == 8 || t + 16 >= G || (a1.x == tag && a1.z == tag));
				if (__all(ok)) {
					acc += a0.y + a1.y;
					break;
				}
				if (spin > SPIN) {
					dead = true;
					break;
				}
			}
		} else if (mode == 10) {
			// 8-byte granules through BUFFER instructions (b64, sc1): lane t reads record t (128 bytes apart)
			typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
			if (lane == 0) {
				u32x2 q = {(unsigned) (g * 7 + r), tag};
				__builtin_amdgcn_raw_buffer_store_b64(q, hr, 0, (int) ((par * GMAX + g) * 128u), 16);
			}
			for (int spin = 0;; ++spin) {
				asm volatile("" ::: "memory");
				u32x2 v = {0u, tag};
				if (lane < G)
					v = __builtin_amdgcn_raw_buffer_load_b64(hr, (int) (lane * 128), (int) (par * GMAX * 128u), 16);
				if (__all(v.y == tag)) {
					acc += v.x;
					break;
				}
				if (spin > SPIN) {
					dead = true;
					break;
				}
			}
		} else if (mode == 7) {
			// mode 0 without the s_sleep between sweeps
			if (lane < 4) {
				u32x4 q = {tag, (unsigned) (g * 7 + r), tag, (unsigned) lane};
				st(q, hr, (unsigned) (lane * 16), (par * GMAX + g) * STRIDE);
			}
			const unsigned voff = (unsigned) ((lane >> 2) * STRIDE + (lane & 3) * 16), soff = par * GMAX * STRIDE;
			const int t = lane >> 2;
			for (int spin = 0;; ++spin) {
				asm volatile("" ::: "memory");
				u32x4 a0 = ld(hr, voff, soff), a1 = ld(hr, voff + 16u * STRIDE, soff);
				bool ok = (t >= G || (a0.x == tag && a0.z == tag)) && (t + 16 >= G || (a1.x == tag && a1.z == tag));
				if (__all(ok)) {
					acc += a0.y + a1.y;
					break;
				}
				if (spin > SPIN) {
					dead = true;
					break;
				}
			}
		} else if (mode == 2) {
			// inbox of consumer c: [par][c][GMAX records of 64 bytes]; lane l < 4 G stores piece l & 3 into inbox (l >> 2) ... two rounds
			const unsigned base = par * (GMAX * GMAX * 64);
			for (int c0 = 0; c0 < G; c0 += 16) {
				const int c = c0 + (lane >> 2);
				if (c < G) {
					u32x4 q = {tag, (unsigned) (g * 7 + r), tag, (unsigned) (lane & 3)};
					st(q, hr, (unsigned) ((c * GMAX + g) * 64 + (lane & 3) * 16), base);
				}
			}
			const unsigned voff = (unsigned) ((g * GMAX + (lane >> 2)) * 64 + (lane & 3) * 16);
			const int t = lane >> 2;
			for (int spin = 0;; ++spin) {
				asm volatile("" ::: "memory");
				u32x4 a0 = ld(hr, voff, base), a1 = ld(hr, voff + 16u * 64u, base);
				bool ok = (t >= G || (a0.x == tag && a0.z == tag)) && (t + 16 >= G || (a1.x == tag && a1.z == tag));
				if (__all(ok)) {
					acc += a0.y + a1.y;
					break;
				}
				if (spin > SPIN) {
					dead = true;
					break;
				}
			}
		} else if (mode == 3) {
			unsigned long long *gr = reinterpret_cast<unsigned long long *>(ws) + par * 4096;
			if (lane == 0)
				__hip_atomic_store(gr + g * 16, ((unsigned long long) tag << 32) | (unsigned) (g * 7 + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			for (int spin = 0;; ++spin) {
				unsigned long long v = lane < G ? __hip_atomic_load(gr + lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long) tag << 32);
				if (__all((unsigned) (v >> 32) == tag)) {
					acc += (unsigned) v;
					break;
				}
				if (spin > SPIN) {
					dead = true;
					break;
				}
			}
		} else {
			unsigned long long *word = reinterpret_cast<unsigned long long *>(ws) + (r & 3) * 64;
			unsigned *cnt = reinterpret_cast<unsigned *>(ws + 65536) + (r & 3) * 64;
			if (lane == 0) {
				__hip_atomic_fetch_max(word, ((unsigned long long) (unsigned) (g * 7 + r) << 8) | (unsigned) g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			const unsigned want = (unsigned) (((r + 3) / 4) * G); // the counter of slot r & 3 after its ((r + 3) / 4)-th use
			for (int spin = 0;; ++spin) {
				unsigned v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (v >= want) {
					acc += __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					break;
				}
				if (spin > SPIN) {
					dead = true;
					break;
				}
			}
		}
		busy(D);
	}
	if (dead && lane == 0)
		atomicExch(status, 1);
	if (lane == 0)
		sink[g] = acc;
}

int main()
{
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const int ncu = prop.multiProcessorCount;
	unsigned char *ws;
	int *status;
	unsigned long long *sink;
	const size_t WS = 1 << 22;
	CK(hipMalloc(&ws, WS));
	CK(hipMalloc(&status, 4));
	CK(hipMalloc(&sink, 8 * 64));
	// the panel stream's CU set: the last 32 CUs
	uint32_t mask[32];
	memset(mask, 0, sizeof(mask));
	for (int i = ncu - 32; i < ncu; ++i)
		mask[i / 32] |= 1u << (i % 32);
	hipStream_t s;
	CK(hipExtStreamCreateWithCUMask(&s, (uint32_t) ((ncu + 31) / 32), mask));
	const int iters = 3000;
	for (int G : {32, 16}) {
		for (int mode : {0, 10, 3}) {
			for (unsigned long long D : {0ull, 2300ull}) {
				CK(hipMemsetAsync(ws, 0, WS, s));
				CK(hipMemsetAsync(status, 0, 4, s));
				hipLaunchKernelGGL(a2a_kernel, dim3(G), dim3(256), 0, s, ws, 16, mode, D, status, sink);
				CK(hipMemsetAsync(ws, 0, WS, s));
				hipEvent_t e0, e1;
				CK(hipEventCreate(&e0));
				CK(hipEventCreate(&e1));
				CK(hipEventRecord(e0, s));
				hipLaunchKernelGGL(a2a_kernel, dim3(G), dim3(256), 0, s, ws, iters, mode, D, status, sink);
				CK(hipEventRecord(e1, s));
				CK(hipEventSynchronize(e1));
				float ms = 0;
				CK(hipEventElapsedTime(&ms, e0, e1));
				int st_ = 0;
				CK(hipMemcpy(&st_, status, 4, hipMemcpyDeviceToHost));
				printf("G %2d mode %d D %4llu cycles: %.3f us per round%s\n", G, mode, D, ms * 1e3 / iters, st_ ? "  TIMEOUT" : "");
				fflush(stdout);
				CK(hipEventDestroy(e0));
				CK(hipEventDestroy(e1));
			}
		}
	}
	return 0;
}
