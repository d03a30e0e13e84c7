// A C++ client of libfaer_hip.so written against the REFERENCE's own C++ wrapper, faer-ffi/faer.hpp (not in this
// repository; tests/test_cabi_client.py passes -I<generated dir>, see below).  faer.hpp spells the symbols
// `libfaer_v0_24_*` while faer.h / the Rust side export `libfaer_v0_23_*` (SURVEY.md section 8b caveat i), and its
// dispatch tables (faer.hpp:17-64) take the address of all six scalar variants of every entry point (caveat ii):
// the test generates a copy of faer.h with the v0_24 spelling next to verbatim copies of faer.hpp / quad.hpp in its
// build directory, and libfaer_hip.so provides the v0_24 names plus aborting stubs for fx128 / c32 / c64 / cx128
// (csrc/abi_compat.c).  Linking this file proves both.  `compute` runs LLT and QR through the wrapper's templates.
// (LU is left to ref_client.c: faer.hpp's Slice::ffi() passes BYTES as SliceMut.len while the Rust side and this
// library read an element count -- caveat iii, a defect of the reference wrapper.)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "faer.hpp"

extern "C" int faer_hip_device_count(void);

static int fails = 0;
#define CHECK(cond)                                                                                                    \
	do {                                                                                                           \
		if (!(cond)) {                                                                                         \
			std::fprintf(stderr, "hpp_client: FAILED %s (line %d)\n", #cond, __LINE__);                     \
			++fails;                                                                                       \
		}                                                                                                      \
	} while (0)

template <typename T> static void host_only()
{
	namespace llt = faer::linalg::cholesky::llt;
	namespace qr = faer::linalg::qr::no_pivoting;
	faer::Par seq {faer::Par::Seq, 0};
	auto p = llt::factor::params<T>();
	CHECK(p.recursion_threshold == 64 && p.block_size == 128);
	auto l = llt::factor::in_place_scratch<T>(777, seq);
	CHECK(l.size >= 777 * sizeof(T));
	auto lq = qr::factor::in_place_scratch<T>(1000, 100, 32, seq);
	CHECK(lq.size >= 32 * 100 * sizeof(T));
}

template <typename T> static void compute(double tol)
{
	namespace llt = faer::linalg::cholesky::llt;
	namespace qr = faer::linalg::qr::no_pivoting;
	const size_t n = 200, k = 3;
	faer::Par seq {faer::Par::Seq, 0};
	std::vector<T> a(n * n), l(n * n), b(n * k), x(n * k);
	unsigned long long s = 88172645463325252ull;
	auto rnd = [&]() {
		s ^= s << 13, s ^= s >> 7, s ^= s << 17;
		return (T) ((double) (s >> 11) / 9007199254740992.0 - 0.5);
	};
	std::vector<T> g(n * n);
	for (auto &v : g)
		v = rnd();
	for (auto &v : b)
		v = rnd();
	for (size_t j = 0; j < n; ++j)
		for (size_t i = 0; i < n; ++i) {
			double acc = i == j ? (double) n : 0.0;
			for (size_t q = 0; q < n; ++q)
				acc += (double) g[i + q * n] * (double) g[j + q * n];
			a[i + j * n] = (T) acc;
		}
	l = a;
	llt::factor::in_place<T>(faer::Mat<T> {l.data(), n, n, 1, (ptrdiff_t) n});
	x = b;
	llt::solve::in_place<T>(faer::Mat<T const> {l.data(), n, n, 1, (ptrdiff_t) n}, faer::Conj {faer::Conj::No},
				faer::Mat<T> {x.data(), n, k, 1, (ptrdiff_t) n}, seq);
	double worst = 0;
	for (size_t c = 0; c < k; ++c)
		for (size_t i = 0; i < n; ++i) {
			double r = -(double) b[i + c * n];
			for (size_t j = 0; j < n; ++j)
				r += (double) a[i + j * n] * (double) x[j + c * n];
			worst = std::fmax(worst, std::fabs(r));
		}
	CHECK(worst < tol);
	// QR of G (square): solve G y = b
	const size_t bs = 16;
	std::vector<T> qrm = g, h(bs * n, (T) 0);
	qr::factor::in_place<T>(faer::Mat<T> {qrm.data(), n, n, 1, (ptrdiff_t) n}, faer::Mat<T> {h.data(), bs, n, 1, (ptrdiff_t) bs}, seq);
	x = b;
	faer::Mat<T const> QB {qrm.data(), n, n, 1, (ptrdiff_t) n}, QC {h.data(), bs, n, 1, (ptrdiff_t) bs};
	qr::solve::in_place<T>(QB, QC, QB, faer::Conj {faer::Conj::No}, faer::Mat<T> {x.data(), n, k, 1, (ptrdiff_t) n}, seq);
	worst = 0;
	for (size_t c = 0; c < k; ++c)
		for (size_t i = 0; i < n; ++i) {
			double r = -(double) b[i + c * n];
			for (size_t j = 0; j < n; ++j)
				r += (double) g[i + j * n] * (double) x[j + c * n];
			worst = std::fmax(worst, std::fabs(r));
		}
	CHECK(worst < tol * 100);
}

int main(int argc, char **argv)
{
	host_only<double>();
	host_only<float>();
	if (argc > 1 && std::strcmp(argv[1], "compute") == 0) {
		if (faer_hip_device_count() < 1) {
			std::fprintf(stderr, "hpp_client: no gfx950 device\n");
			return 2;
		}
		compute<double>(1e-10);
		compute<float>(2e-2);
	}
	if (fails == 0)
		std::printf("hpp_client ok (%s)\n", argc > 1 ? argv[1] : "host-only");
	return fails == 0 ? 0 : 1;
}
