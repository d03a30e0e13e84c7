// Cross-workgroup exchange inside one launch (cooperative panel kernels of LU and QR).
//
// gfx950 has 8 XCDs with private, mutually non-coherent L2s and per-CU L1s that are never refreshed by other
// CUs' stores, so data handed between resident workgroups must bypass them (MI355X_MICROARCH.md, "Workgroup
// dispatch, XCD placement & inter-workgroup visibility"; cdna_hip_programming.md Guideline 16, recipe R1):
//   producer : payload with relaxed AGENT-scope 8-byte stores (sc1, write-through)  ->  the storing wave drains
//              them (s_waitcnt vmcnt(0))  ->  ONE lane stores the workgroup's flag = epoch (relaxed, agent);
//   consumer : ONE wave polls the G flags relaxed (s_sleep between sweeps, bounded)  ->  __syncthreads()  ->
//              payload read back with relaxed AGENT-scope loads (sc1: served by the memory side, never by a
//              stale L1 / foreign L2 line).
// No release/acquire fence is needed on either side (each costs ~1.7 us on this part); a full all-to-all round
// costs one store drain + one flag hop + the payload reads.  Epochs are monotonic over the whole factorization
// (flags are zeroed once per call by the host), payload slots are double buffered by the caller (epoch parity):
// a workgroup can only be one round ahead of the slowest one, because publishing round e+1 requires having
// consumed round e, which requires everybody's flag >= e.
#pragma once
#include <hip/hip_runtime.h>

namespace fh {

typedef unsigned long long xwg_u64;

static __device__ __forceinline__ void xwg_store(double *p, double v)
{
	__hip_atomic_store(reinterpret_cast<xwg_u64 *>(p), (xwg_u64) __double_as_longlong(v), __ATOMIC_RELAXED,
			   __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ double xwg_load(const double *p)
{
	return __longlong_as_double((long long) __hip_atomic_load(reinterpret_cast<const xwg_u64 *>(p), __ATOMIC_RELAXED,
								 __HIP_MEMORY_SCOPE_AGENT));
}
static __device__ __forceinline__ void xwg_store_i(int *p, int v)
{
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ int xwg_load_i(const int *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// R2 granule: {tag (high 32 bits), 32 payload bits} in ONE naturally aligned 8-byte write-through store; a
// reader that finds the expected tag has the payload (no flag, no fence, no store drain).  Granule memory must
// be zeroed once per factorization and tags must never be 0.
static __device__ __forceinline__ void xwg_store_gran(xwg_u64 *p, unsigned tag, unsigned value)
{
	__hip_atomic_store(p, ((xwg_u64) tag << 32) | (xwg_u64) value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ xwg_u64 xwg_load_gran(const xwg_u64 *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by the wave that issued the payload stores (all lanes): drain them, then lane `leader` raises the flag.
static __device__ __forceinline__ void xwg_publish(xwg_u64 *flags, int g, xwg_u64 epoch, bool leader)
{
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if (leader)
		__hip_atomic_store(flags + g, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every thread of the workgroup calls it; wave 0 polls.  Returns false on timeout (bounded spins).
static __device__ __forceinline__ bool xwg_wait_all(const xwg_u64 *flags, int G, xwg_u64 epoch, int *s_flag)
{
	const int tid = threadIdx.x;
	if (tid < 64) {
		int ok = 0;
		for (int spin = 0; spin < (1 << 21); ++spin) {
			bool all = true;
			for (int t = tid; t < G; t += 64)
				all = all && (__hip_atomic_load(flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch);
			if (__all(all)) {
				ok = 1;
				break;
			}
			__builtin_amdgcn_s_sleep(1);
		}
		if (tid == 0)
			*s_flag = ok;
	}
	__syncthreads();
	return *s_flag != 0;
}

} // namespace fh
