// Multi-GPU entry points: 1-D block-cyclic columns, one process per GPU, RCCL (or any transport) through the
// caller's broadcast callback -- SURVEY.md section 8e.  The orchestration lives in dist_lu.h (backend template);
// this file is its device backend: every operation is one of the library's own HIP drivers on the calling
// thread's stream, the broadcast is handed a DEVICE buffer.
#include "common.h"
#include "dist_lu.h"
using namespace fh;

namespace {

template <typename S> struct DeviceBackend {
	typedef S T;
	struct View {
		T *p;
		long nrows, ncols, rs, cs;
	};
	FaerHipComm comm;

	static MatV<T> mv(View v) { return MatV<T>{v.p, v.nrows, v.ncols, v.rs, v.cs}; }
	void factor_panel(View P, int *piv_out) { getrf_panel_dev<T>(mv(P), piv_out); }
	void laswp(View B, const int *piv, int nt) { laswp_rows_dev<T>(mv(B), piv, nt); }
	void trsm_unit_lower(View L, View X) { trsm_lower_dev<T>(mv(L).c(), true, mv(X)); }
	void gemm_sub(View C, View A, View B) { gemm_dev<T>(mv(C), DST_FULL, true, mv(A).c(), mv(B).c(), (T) -1); }
	void pack(View src, T *dst) { copy_dev<T>(MatV<T>{dst, src.nrows, src.ncols, 1, src.nrows}, mv(src).c()); }
	void bcast(void *buf, size_t bytes, int root)
	{
		FH_CHECK(comm.bcast != nullptr, "dist lu: FaerHipComm.bcast is NULL");
		comm.bcast(comm.user, buf, bytes, root);
	}
	void to_host(int *dst, const int *src, size_t n)
	{
		FH_HIP(hipMemcpyAsync(dst, src, n * sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
		ctx().sync();
	}
};

template <typename T>
FaerPartialPivLuStatus dist_lu_api(FaerMatMut A_local, size_t n_global, size_t nb, FaerSliceMut pf, FaerSliceMut pb, FaerHipComm comm,
				   void *panel_ws)
{
	typedef DeviceBackend<T> B;
	const long m = (long) A_local.nrows, n = (long) n_global;
	FH_CHECK(comm.world_size >= 1 && comm.rank >= 0 && comm.rank < comm.world_size, "dist lu: bad communicator");
	FH_CHECK(nb >= 1 && (long) nb <= m, "dist lu: block width must be in [1, nrows]");
	FH_CHECK((size_t) A_local.ncols == DistLu<B>::local_ncols(n_global, nb, comm.rank, comm.world_size),
		 "dist lu: A_local has the wrong number of columns for this rank");
	FH_CHECK((long) pf.len == m && (long) pb.len == m, "dist lu: perm slices must have nrows entries");
	FH_CHECK(is_device_ptr(A_local.ptr) && is_device_ptr(panel_ws), "dist lu: A_local and panel_ws must be device memory");
	FH_CHECK(A_local.row_stride == 1, "dist lu: A_local must be column major");
	B be;
	be.comm = comm;
	typename B::View Av{static_cast<T *>(A_local.ptr), m, (long) A_local.ncols, 1, (long) A_local.col_stride};
	const long size = m < n ? m : n;
	std::vector<int> piv((size_t) size);
	DistLu<B>::run(be, Av, m, n, (long) nb, comm.rank, comm.world_size, static_cast<T *>(panel_ws), piv.data());
	// lu/partial_pivoting/factor.rs:274-277: perm = identity with the transpositions applied in order
	unsigned long long *f = static_cast<unsigned long long *>(pf.ptr), *b = static_cast<unsigned long long *>(pb.ptr);
	for (long i = 0; i < m; ++i)
		f[i] = (unsigned long long) i;
	size_t nt = 0;
	for (long j = 0; j < size; ++j)
		if (piv[(size_t) j] != j) {
			std::swap(f[j], f[piv[(size_t) j]]);
			++nt;
		}
	for (long i = 0; i < m; ++i)
		b[f[i]] = (unsigned long long) i;
	FaerPartialPivLuStatus st;
	memset(&st, 0, sizeof(st));
	st.tag = FaerPartialPivLuStatus_Ok;
	st.ok.transposition_count = nt;
	return st;
}

} // namespace

extern "C" {
size_t faer_hip_dist_local_ncols(size_t n, size_t nb, int rank, int world_size)
{
	return DistLu<DeviceBackend<double>>::local_ncols(n, nb, rank, world_size);
}
size_t faer_hip_dist_panel_ws_scalars(size_t nrows, size_t nb, FaerHipDType dtype)
{
	return dtype == FaerHipDType_F64 ? DistLu<DeviceBackend<double>>::ws_scalars((long) nrows, (long) nb)
					 : DistLu<DeviceBackend<float>>::ws_scalars((long) nrows, (long) nb);
}
FaerPartialPivLuStatus faer_hip_dist_partial_piv_lu_f64(FaerMatMut A, size_t n, size_t nb, FaerSliceMut pf, FaerSliceMut pb, FaerHipComm comm,
							  void *ws)
{
	return dist_lu_api<double>(A, n, nb, pf, pb, comm, ws);
}
FaerPartialPivLuStatus faer_hip_dist_partial_piv_lu_f32(FaerMatMut A, size_t n, size_t nb, FaerSliceMut pf, FaerSliceMut pb, FaerHipComm comm,
							  void *ws)
{
	return dist_lu_api<float>(A, n, nb, pf, pb, comm, ws);
}
}
