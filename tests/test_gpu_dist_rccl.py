"""The built-in RCCL transport (csrc/rccl_transport.hip: ncclBroadcast on a dedicated stream, event ordered) and the
two-stream schedule inside a rank (csrc/dist.hip: rest of update k on the bulk stream, look-ahead panel on the
CU-masked panel stream), exercised with ONE rank on one GPU -- RCCL refuses two ranks on the same device, so the
multi-rank control flow is covered by test_gpu_dist_two_ranks.py (gloo transport) and the CPU gloo tests.
The results must equal the single-GPU factorization by the same library: identical pivots, factors within tolerance."""
import numpy as np
import pytest

from gpu_util import init_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def transport():
    F = init_gpu()
    try:
        uid = F.RcclTransport.unique_id()
    except RuntimeError as e:
        pytest.skip(str(e))
    t = F.RcclTransport(uid, 0, 1)
    yield t
    t.close()


@pytest.mark.parametrize("n,nb", [(1536, 256), (4096, 512), (3000, 384)])
def test_rccl_transport_lu_single_rank(transport, n, nb):
    import torch

    F = init_gpu()
    g = torch.Generator(device="cuda").manual_seed(n)
    a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
    loc = a.clone()
    fwd, bwd, cnt = F.dist_partial_piv_lu(loc, n, nb, 0, 1, transport=transport)
    F.synchronize()
    ref = a.clone()
    p, _, c = F.partial_piv_lu_factor_in_place(ref)
    F.synchronize()
    assert np.array_equal(fwd, p) and cnt == c
    assert (loc - ref).abs().max().item() <= 256 * n * 2.3e-16 * max(1.0, ref.abs().max().item())


def test_rccl_transport_llt_single_rank(transport):
    import torch

    F = init_gpu()
    n, nb = 2048, 256
    g = torch.Generator(device="cuda").manual_seed(3)
    b = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g)
    a = (b @ b.t() + n * torch.eye(n, dtype=torch.float64, device="cuda")).t().contiguous().t()
    loc = a.clone()
    assert F.dist_llt(loc, n, nb, 0, 1, transport=transport) == 0
    F.synchronize()
    ref = a.clone()
    assert F.llt_factor_in_place(ref) == 0
    F.synchronize()
    assert (torch.tril(loc) - torch.tril(ref)).abs().max().item() <= 256 * n * 2.3e-16 * ref.abs().max().item()


def test_one_stream_and_two_stream_schedules_agree(transport, monkeypatch):
    """FAER_HIP_DIST_ONE_STREAM=1 runs the whole rank on the caller's stream (round 1's schedule): same pivots, same
    factors bit for bit (the kernels and their order per entry are the same; only the stream assignment differs)"""
    import torch

    F = init_gpu()
    n, nb = 2560, 512
    g = torch.Generator(device="cuda").manual_seed(9)
    a = torch.randn((n, n), dtype=torch.float64, device="cuda", generator=g).t()
    l2 = a.clone()
    f2, _, _ = F.dist_partial_piv_lu(l2, n, nb, 0, 1, transport=transport)
    F.synchronize()
    monkeypatch.setenv("FAER_HIP_DIST_ONE_STREAM", "1")
    l1 = a.clone()
    f1, _, _ = F.dist_partial_piv_lu(l1, n, nb, 0, 1, transport=transport)
    F.synchronize()
    assert np.array_equal(f1, f2)
    assert (l1 - l2).abs().max().item() <= 64 * n * 2.3e-16 * max(1.0, l1.abs().max().item())
