// Multi-GPU entry points (placeholder until the single-GPU path is parity green).
#include "common.h"
using namespace fh;
extern "C" {
size_t faer_hip_dist_local_ncols(size_t n, size_t nb, int rank, int world_size)
{
	const size_t nblk = (n + nb - 1) / nb;
	size_t cols = 0;
	for (size_t b = (size_t) rank; b < nblk; b += (size_t) world_size)
		cols += (b + 1) * nb <= n ? nb : n - b * nb;
	return cols;
}
FaerPartialPivLuStatus faer_hip_dist_partial_piv_lu_f64(FaerMatMut, size_t, size_t, FaerSliceMut, FaerSliceMut, FaerHipComm, void *)
{
	die("dist lu: not implemented yet", __FILE__, __LINE__);
}
FaerPartialPivLuStatus faer_hip_dist_partial_piv_lu_f32(FaerMatMut, size_t, size_t, FaerSliceMut, FaerSliceMut, FaerHipComm, void *)
{
	die("dist lu: not implemented yet", __FILE__, __LINE__);
}
}
