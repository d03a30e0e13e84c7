// What HBM rate can a column-major tall-skinny matrix (5e5 x 256 fp32, the one-pass QR's shape) be streamed at, by access pattern?
// The update / Gram kernels of csrc/tsqr.hip run at 3.3-3.7 / 2.5 TB/s; a linear float4 copy reaches 6.3 (MI355X_MICROARCH.md).
// This probe measures in-place "x *= s" (read + write, the update's traffic shape) and a read-only sum (the Gram's) with
//   LPC  lanes per column inside one wave-wide 16-byte load: 16 / 32 / 64 -> 256 / 512 / 1024-byte runs per column,
//   U    loads in flight per lane and batch (two batches are double buffered),
//   WPC  workgroups per CU (persistent grid),
// and the order in which a wave walks its (row block, column batch) pairs.
// Build: hipcc -O3 --offload-arch=gfx950 tools/tall_stream_probe.hip -o tools/tall_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                                                                  \
	do {                                                                                                                   \
		hipError_t e_ = (x);                                                                                           \
		if (e_ != hipSuccess) {                                                                                        \
			fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                                   \
			exit(1);                                                                                               \
		}                                                                                                              \
	} while (0)

__global__ __launch_bounds__(256) void linear_rw(f32x4 *p, long n4, float s)
{
	for (long i = (long) blockIdx.x * 256 + threadIdx.x; i < n4; i += (long) gridDim.x * 256) {
		f32x4 v = p[i];
		p[i] = v * s;
	}
}
__global__ __launch_bounds__(256) void linear_ro(const f32x4 *p, long n4, float *out)
{
	f32x4 acc = {0, 0, 0, 0};
	for (long i = (long) blockIdx.x * 256 + threadIdx.x; i < n4; i += (long) gridDim.x * 256)
		acc += p[i];
	if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f)
		out[0] = 1.f;
}

// MODE 0: in place scale, 1: read only.  ORDER 0: a wave takes one row block and sweeps the columns; 1: the workgroup's four waves take
// the SAME row block and interleave column batches (wave w: batches w, w + 4, ...): fewer rows, more columns per CU at a time
template <int LPC, int U, int MODE, int ORDER> __global__ __launch_bounds__(256) void tall_kernel(float *A, long ld, int m, int n, float s, float *out)
{
	constexpr int CPI = 64 / LPC;	 // columns per wave-wide load
	constexpr int RPW = 4 * LPC;	 // rows per wave block
	constexpr int CPB = U * CPI;	 // columns per batch
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int lr = 4 * (lane % LPC), lc = lane / LPC;
	const int nrb = (m + RPW - 1) / RPW, ncb = (n + CPB - 1) / CPB;
	f32x4 acc = {0, 0, 0, 0};
	const long nwaves = (long) gridDim.x * 4, wid = (long) blockIdx.x * 4 + wv;
	const long total = ORDER == 0 ? nrb : (long) nrb;
	for (long rb = ORDER == 0 ? wid : blockIdx.x; rb < total; rb += ORDER == 0 ? nwaves : gridDim.x) {
		const int r = (int) rb * RPW + lr;
		const bool rok = r + 3 < m;
		f32x4 cur[U], nxt[U];
		auto load = [&](f32x4 (&v)[U], int cb) {
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const int c = cb * CPB + u * CPI + lc;
				v[u] = f32x4{0, 0, 0, 0};
				if (rok && c < n)
					v[u] = *reinterpret_cast<const f32x4 *>(A + (long) c * ld + r);
			}
		};
		const int cb0 = ORDER == 0 ? 0 : wv, cbs = ORDER == 0 ? 1 : 4;
		if (cb0 < ncb)
			load(cur, cb0);
		for (int cb = cb0; cb < ncb; cb += cbs) {
			if (cb + cbs < ncb)
				load(nxt, cb + cbs);
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const int c = cb * CPB + u * CPI + lc;
				if (MODE == 0) {
					if (rok && c < n)
						*reinterpret_cast<f32x4 *>(A + (long) c * ld + r) = cur[u] * s;
				} else {
					acc += cur[u];
				}
			}
#pragma unroll
			for (int u = 0; u < U; ++u)
				cur[u] = nxt[u];
		}
	}
	if (MODE == 1 && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f)
		out[0] = 1.f;
}

template <typename F> static double time_ms(F f, int reps)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	f();
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	for (int i = 0; i < reps; ++i)
		f();
	CK(hipEventRecord(b));
	CK(hipEventSynchronize(b));
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	return ms / reps;
}

int main(int argc, char **argv)
{
	const int m = argc > 1 ? atoi(argv[1]) : 500000, n = argc > 2 ? atoi(argv[2]) : 256;
	const long ld = (m + 15) / 16 * 16;
	float *A, *out;
	CK(hipMalloc(&A, (size_t) ld * n * 4));
	CK(hipMalloc(&out, 64));
	CK(hipMemset(A, 0, (size_t) ld * n * 4));
	const double bytes = (double) m * n * 4;
	const int reps = 10;
	printf("# m=%d n=%d ld=%ld  bytes per pass %.1f MB\n", m, n, ld, bytes / 1e6);
	for (int wpc : {2, 4, 8}) {
		double t = time_ms([&] { hipLaunchKernelGGL(linear_rw, dim3(256 * wpc), dim3(256), 0, 0, (f32x4 *) A, ld * n / 4, 1.0f); }, reps);
		printf("linear rw  wpc=%d  %.1f us  %.2f TB/s\n", wpc, t * 1e3, 2 * (double) ld * n * 4 / t / 1e9);
		t = time_ms([&] { hipLaunchKernelGGL(linear_ro, dim3(256 * wpc), dim3(256), 0, 0, (const f32x4 *) A, ld * n / 4, out); }, reps);
		printf("linear ro  wpc=%d  %.1f us  %.2f TB/s\n", wpc, t * 1e3, (double) ld * n * 4 / t / 1e9);
	}
#define RUN(LPC, U, MODE, ORDER, WPC)                                                                                          \
	{                                                                                                                      \
		double t = time_ms([&] { hipLaunchKernelGGL((tall_kernel<LPC, U, MODE, ORDER>), dim3(256 * WPC), dim3(256), 0, 0, A, ld, m, n, 1.0f, out); }, reps); \
		printf("tall %s lpc=%2d (%4d B runs) U=%2d order=%d wpc=%d  %.1f us  %.2f TB/s\n", MODE == 0 ? "rw" : "ro", LPC, LPC * 16, U, ORDER, WPC,  \
		       t * 1e3, (MODE == 0 ? 2 : 1) * bytes / t / 1e9);                                                         \
	}
	RUN(16, 16, 0, 0, 1) RUN(32, 16, 0, 0, 1) RUN(64, 16, 0, 0, 1)
	RUN(16, 16, 0, 0, 2) RUN(32, 16, 0, 0, 2) RUN(64, 16, 0, 0, 2)
	RUN(32, 8, 0, 0, 2) RUN(64, 8, 0, 0, 2) RUN(32, 8, 0, 0, 4) RUN(64, 8, 0, 0, 4) RUN(64, 4, 0, 0, 8)
	RUN(32, 16, 0, 1, 1) RUN(64, 16, 0, 1, 1) RUN(32, 16, 0, 1, 2) RUN(64, 16, 0, 1, 2) RUN(64, 8, 0, 1, 4)
	RUN(16, 16, 1, 0, 2) RUN(32, 16, 1, 0, 2) RUN(64, 16, 1, 0, 2) RUN(16, 8, 1, 0, 4) RUN(32, 8, 1, 0, 4) RUN(64, 8, 1, 0, 4)
	RUN(64, 16, 1, 1, 2) RUN(64, 8, 1, 1, 4) RUN(64, 4, 1, 0, 8)
	return 0;
}
