"""GPU tests of the planes GEMM (renet_pack_planes + renet_gemm_planes): pre-split bf16x6 operands, LDS-DMA staging,
K-contiguous (ds_read_b128) and K-strided (ds_read_b64_tr_b16) fragment paths -- against fp64 at the fp32-class
tolerance of the other GEMM kernels, for every role combination, ragged sizes and split-K."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    import renet_hip
    renet_hip.lib()
    return torch.device('cuda:0')


def test_pack_planes_reconstructs_the_input(dev):
    import renet_hip as K
    rng = np.random.RandomState(0)
    for r, c in ((1, 1), (5, 130), (128, 128), (300, 37), (1024, 600)):
        x = (rng.randn(r, c) * 3).astype(np.float32)
        x[0, 0] = 0.0
        pl = K.pack_planes(torch.from_numpy(x).to(dev))
        p = pl.p.float().cpu().numpy()
        assert p.shape == (3, (r + 127) // 128 * 128, (c + 127) // 128 * 128)
        rec = p[0].astype(np.float64) + p[1] + p[2]
        np.testing.assert_allclose(rec[:r, :c], x, rtol=2.0 ** -24, atol=1e-38)
        assert np.all(p[:, r:, :] == 0) and np.all(p[:, :, c:] == 0)               # zero padding
        # a strided view (row slice of a wider matrix)
        wide = torch.from_numpy(np.concatenate((x, x), axis=1)).to(dev)
        pl2 = K.pack_planes(wide[:, :c])
        assert torch.equal(pl2.p, pl.p)


@pytest.mark.parametrize('m,n,k', [(1, 1, 1), (7, 5, 3), (128, 128, 32), (257, 130, 71), (1024, 777, 600),
                                   (300, 200, 4097), (96, 100, 5000), (1024, 600, 2300)])
@pytest.mark.parametrize('a_tr,b_tr', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_planes_gemm_matches_fp64(dev, m, n, k, a_tr, b_tr):
    import renet_hip as K
    rng = np.random.RandomState(m * 131 + n * 17 + k + a_tr * 2 + b_tr)
    a = rng.uniform(-1, 1, (k, m) if a_tr else (m, k)).astype(np.float32)         # asymmetric operands
    b = rng.uniform(-1, 1, (k, n) if b_tr else (n, k)).astype(np.float32)
    bias = rng.uniform(-1, 1, n).astype(np.float32)
    pa, pb = K.pack_planes(torch.from_numpy(a).to(dev)), K.pack_planes(torch.from_numpy(b).to(dev))
    ref = (a.T if a_tr else a).astype(np.float64) @ (b if b_tr else b.T).astype(np.float64)
    for sk in (None, 1, 3):
        out = K.gemm_planes(pa, bool(a_tr), pb, bool(b_tr), bias=torch.from_numpy(bias).to(dev), split_k=sk)
        np.testing.assert_allclose(out.cpu().numpy(), ref + bias, rtol=1e-5, atol=1e-5 * max(1, k) ** 0.5)
    # alpha / beta accumulation into an existing tensor
    c0 = rng.uniform(-1, 1, (m, n)).astype(np.float32)
    out = torch.from_numpy(c0.copy()).to(dev)
    K.gemm_planes(pa, bool(a_tr), pb, bool(b_tr), out=out, alpha=0.5, beta=1.0)
    np.testing.assert_allclose(out.cpu().numpy(), 0.5 * ref + c0, rtol=1e-5, atol=1e-5 * max(1, k) ** 0.5)
    # same numbers as the in-loop-split bf16x6 kernel (identical term products; only the summation order may differ)
    t = K.gemm(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), ta=bool(a_tr), tb=not b_tr, mode='bf16x6', split_k=1)
    out1 = K.gemm_planes(pa, bool(a_tr), pb, bool(b_tr), split_k=1)
    np.testing.assert_allclose(out1.cpu().numpy(), t.cpu().numpy(), rtol=2e-6, atol=2e-6 * max(1, k) ** 0.5)
