"""A plan kept on a DeviceCSR must not outlive the arrays it regroups: device.spmm(algo=0) keeps AUTO's plan — a
copy of A's entries — on the matrix; an in-place change of A's tensors (torch's version counters) or another tensor in
their place makes the next product rebuild it.  Checked against the oracle (src/matmul.cpp:150-185)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_kept_plan_follows_in_place_changes_of_the_matrix(gpu):
    import torch
    from matrixextra_amd import device as D, synth
    from oracle import oracle as O
    m, K, n = 100_000, 100_000, 128
    p, j, x = synth.csr_fixed(m, K, 32, seed=3)
    B_host = synth.dense_normal(K, n, seed=4)
    A = D.DeviceCSR.from_host(p, j, x, K)
    B = torch.from_numpy(B_host).cuda()

    def check(scale, jj=j):
        C = D.spmm(A, B, colmajor=False)                    # AUTO, plan kept
        rows = 512
        ref = np.zeros(rows * n)
        O.gemm_csr_drm_as_drm(rows, n, p[:rows + 1], jj[:p[rows]].copy(), (x[:p[rows]] * scale).copy(), B_host.reshape(-1), n, ref, n, 1, True)
        np.testing.assert_allclose(C[:rows].cpu().numpy(), ref.reshape(rows, n), rtol=1e-11, atol=1e-11)

    check(1.0)
    assert A._plan is not None and A._plan_limited          # the planned kernel ran and its plan is kept
    check(1.0)                                              # same plan again
    A.values.mul_(2.0)                                      # in place: the kept plan holds the OLD values
    check(2.0)
    A.values = A.values * 0.5                               # another tensor in its place
    check(1.0)
    j2 = j.copy()
    j2[: p[1]] = np.sort((j2[: p[1]] + 7) % K)              # the first row points elsewhere
    A.indices.copy_(torch.from_numpy(j2).cuda())            # in place again
    check(1.0, j2)
    # a plan built without AUTO's padding limit (spmm_planned on a fresh matrix) is not silently taken for AUTO's
    A2 = D.DeviceCSR.from_host(p, j, x, K)
    D.spmm_planned(A2, B, colmajor=False)
    assert A2._plan is not None and not A2._plan_limited
    D.spmm(A2, B, colmajor=False)
    assert A2._plan_limited
