// v_mfma_{f64,f32}_16x16x4 wrappers shared by the GEMM kernel and the LDS-resident block kernels.
//
// D = A * B + C on one wavefront: lane l supplies a = A[i = l & 15][k = l >> 4] and b = B[k = l >> 4][j = l & 15];
// result register r of lane l is D[i = row(r, l >> 4)][j = l & 15] (the f64 and f32 forms differ in row()).
#pragma once
#include <hip/hip_runtime.h>

namespace fh {

template <typename T> struct Mfma;
template <> struct Mfma<double> {
	typedef double acc_t __attribute__((ext_vector_type(4)));
	static __device__ __forceinline__ acc_t run(double a, double b, acc_t c)
	{
		return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
	}
	// f64 16x16x4 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
	static __device__ __forceinline__ int row(int r, int lhi) { return lhi + 4 * r; }
};
template <> struct Mfma<float> {
	typedef float acc_t __attribute__((ext_vector_type(4)));
	static __device__ __forceinline__ acc_t run(float a, float b, acc_t c)
	{
		return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
	}
	// f32 16x16x4 C/D map: col = lane & 15, row = (lane >> 4) * 4 + reg
	static __device__ __forceinline__ int row(int r, int lhi) { return lhi * 4 + r; }
};

} // namespace fh
