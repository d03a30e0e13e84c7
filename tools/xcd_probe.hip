// Where do the workgroups of a CU-masked stream land, and what does a hand-off between two of them cost?
//   hipcc --offload-arch=gfx950 -O2 -o tools/xcd_probe tools/xcd_probe.hip && tools/xcd_probe
// For each mask layout: XCC_ID / SE / CU of 64 resident workgroups, and the ping-pong latency (tagged granule, write-through
// store -> polling load, as the LU panel kernel exchanges its headers) between workgroup 0 and each of a few partners.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define CK(x)                                                                                                              \
	do {                                                                                                               \
		hipError_t e_ = (x);                                                                                       \
		if (e_ != hipSuccess) {                                                                                    \
			fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                        \
			exit(1);                                                                                           \
		}                                                                                                          \
	} while (0)

__global__ void where_kernel(unsigned *out, unsigned long long hold)
{
	unsigned xcc, hw;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	while (__builtin_amdgcn_s_memtime() - t0 < hold)
		__builtin_amdgcn_s_sleep(8);
	if (threadIdx.x == 0) {
		out[2 * blockIdx.x] = xcc;
		out[2 * blockIdx.x + 1] = hw;
	}
}

// workgroup 0 and workgroup `partner` bounce a granule `iters` times; every other workgroup just records where it is and leaves
__global__ void pingpong_kernel(unsigned long long *gran, int partner, int iters, unsigned *where, int *status, int sc_mode)
{
	unsigned xcc, hw;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	if (threadIdx.x == 0) {
		where[2 * blockIdx.x] = xcc;
		where[2 * blockIdx.x + 1] = hw;
	}
	const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == (unsigned) partner ? 1 : -1);
	if (me < 0) {
		// stay resident for a while so that the partner really is the partner-th workgroup of a full set
		const unsigned long long t0 = __builtin_amdgcn_s_memtime();
		while (__builtin_amdgcn_s_memtime() - t0 < 200000)
			__builtin_amdgcn_s_sleep(8);
		return;
	}
	if (threadIdx.x != 0)
		return;
	unsigned long long *mine = gran + (me == 0 ? 0 : 16), *theirs = gran + (me == 0 ? 16 : 0);
	for (int r = 1; r <= iters; ++r) {
		if (me == 0) {
			if (sc_mode == 0)
				__hip_atomic_store(mine, ((unsigned long long) r << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			else
				__hip_atomic_store(mine, ((unsigned long long) r << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		bool ok = false;
		for (int spin = 0; spin < (1 << 22); ++spin) {
			unsigned long long v;
			if (sc_mode == 0)
				v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			else
				v = __builtin_nontemporal_load(theirs);
			if ((unsigned) (v >> 32) == (unsigned) r) {
				ok = true;
				break;
			}
		}
		if (!ok) {
			atomicExch(status, 1);
			return;
		}
		if (me == 1) {
			if (sc_mode == 0)
				__hip_atomic_store(mine, ((unsigned long long) r << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			else
				__hip_atomic_store(mine, ((unsigned long long) r << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	}
}

int main(int argc, char **argv)
{
	int ncu = 0;
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	ncu = prop.multiProcessorCount;
	printf("device %s, %d CUs\n", prop.gcnArchName, ncu);
	const int nwg = 32;
	unsigned *d_where;
	unsigned long long *d_gran;
	int *d_status;
	CK(hipMalloc(&d_where, 2 * 256 * sizeof(unsigned)));
	CK(hipMalloc(&d_gran, 512));
	CK(hipMalloc(&d_status, 4));
	struct Layout {
		const char *name;
		std::vector<int> cus;
	};
	std::vector<Layout> layouts;
	{
		Layout l;
		l.name = "none(plain stream)";
		layouts.push_back(l);
		Layout a{"first32", {}}, b{"last32", {}}, c{"every8th(i%8==0)", {}}, d{"every8th(i%8==7)", {}}, e{"first4_of_each_32", {}}, f{"first64", {}},
			g{"i%8==0,first16", {}}, h{"first32_even", {}};
		for (int i = 0; i < ncu; ++i) {
			if (i < 32)
				a.cus.push_back(i);
			if (i >= ncu - 32)
				b.cus.push_back(i);
			if (i % 8 == 0)
				c.cus.push_back(i);
			if (i % 8 == 7)
				d.cus.push_back(i);
			if (i % 32 < 4)
				e.cus.push_back(i);
			if (i < 64)
				f.cus.push_back(i);
			if (i % 8 == 0 && i < 128)
				g.cus.push_back(i);
			if (i < 64 && i % 2 == 0)
				h.cus.push_back(i);
		}
		layouts.push_back(a);
		layouts.push_back(b);
		layouts.push_back(c);
		layouts.push_back(d);
		layouts.push_back(e);
		layouts.push_back(f);
		layouts.push_back(g);
		layouts.push_back(h);
	}
	for (auto &L : layouts) {
		hipStream_t s;
		if (L.cus.empty()) {
			CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
		} else {
			uint32_t mask[32];
			memset(mask, 0, sizeof(mask));
			for (int cu : L.cus)
				mask[cu / 32] |= 1u << (cu % 32);
			CK(hipExtStreamCreateWithCUMask(&s, (uint32_t) ((ncu + 31) / 32), mask));
		}
		// placement of `nwg` resident 256-thread workgroups
		hipLaunchKernelGGL(where_kernel, dim3(nwg), dim3(256), 0, s, d_where, 400000ull);
		CK(hipStreamSynchronize(s));
		std::vector<unsigned> w(2 * nwg);
		CK(hipMemcpy(w.data(), d_where, w.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
		std::map<unsigned, int> per_xcc;
		std::string seq;
		for (int i = 0; i < nwg; ++i) {
			per_xcc[w[2 * i] & 0xF]++;
			char buf[32];
			// HW_ID (gfx9): [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se
			snprintf(buf, sizeof(buf), "%u:%u.%u ", w[2 * i] & 0xF, (w[2 * i + 1] >> 13) & 7, (w[2 * i + 1] >> 8) & 15);
			seq += buf;
		}
		printf("\n[%s] %zu CUs enabled; workgroups per XCC:", L.name, L.cus.size());
		for (auto &kv : per_xcc)
			printf(" %u:%d", kv.first, kv.second);
		printf("\n   xcc:se.cu of workgroups 0..%d: %s\n", nwg - 1, seq.c_str());
		// hop latency workgroup 0 <-> partner
		for (int partner : {1, 2, 3, 4, 7, 8, 9, 16, 31}) {
			for (int mode = 0; mode < 1; ++mode) {
				CK(hipMemsetAsync(d_gran, 0, 512, s));
				CK(hipMemsetAsync(d_status, 0, 4, s));
				hipEvent_t e0, e1;
				CK(hipEventCreate(&e0));
				CK(hipEventCreate(&e1));
				const int iters = 2000;
				hipLaunchKernelGGL(pingpong_kernel, dim3(nwg), dim3(256), 0, s, d_gran, partner, 8, d_where, d_status, mode);
				CK(hipMemsetAsync(d_gran, 0, 512, s));
				CK(hipEventRecord(e0, s));
				hipLaunchKernelGGL(pingpong_kernel, dim3(nwg), dim3(256), 0, s, d_gran, partner, iters, d_where, d_status, mode);
				CK(hipEventRecord(e1, s));
				CK(hipEventSynchronize(e1));
				float ms = 0;
				CK(hipEventElapsedTime(&ms, e0, e1));
				int st = 0;
				CK(hipMemcpy(&st, d_status, 4, hipMemcpyDeviceToHost));
				CK(hipMemcpy(w.data(), d_where, w.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
				printf("   hop wg0(xcc %u) <-> wg%d(xcc %u): %.3f us one way%s\n", w[0] & 0xF, partner, w[2 * partner] & 0xF,
				       (ms * 1e3) / (2.0 * iters), st ? "  TIMEOUT" : "");
				CK(hipEventDestroy(e0));
				CK(hipEventDestroy(e1));
			}
		}
		CK(hipStreamDestroy(s));
	}
	return 0;
}
