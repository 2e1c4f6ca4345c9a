"""GPU: the brute-force kNN producer returns scikit-learn's exact neighbours (the reference's
CPU implementation, neighbors.py:106-107,242) and squared-l2 distances."""

import numpy as np
import pytest

from tests.util import to_dev

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,k", [(1, 5), (8, 10), (40, 30)])
def test_matches_sklearn(d, k):
    from sklearn.neighbors import NearestNeighbors

    from muygpys_amd.neighbors import NN_Wrapper

    rng = np.random.default_rng(d)
    X = rng.normal(size=(3000, d))
    Q = rng.normal(size=(257, d))
    nn = NN_Wrapper(to_dev(X, torch.float64), k, chunk=100)
    idx, dist = nn.get_nns(to_dev(Q, torch.float64))
    ref = NearestNeighbors(n_neighbors=k, algorithm="brute").fit(X)
    rd, ri = ref.kneighbors(Q)
    assert np.array_equal(idx.cpu().numpy(), ri)
    np.testing.assert_allclose(dist.cpu().numpy(), rd**2, rtol=1e-9, atol=1e-12)
    bi = rng.choice(3000, size=100, replace=False)
    bidx, bdist = nn.get_batch_nns(to_dev(bi))
    rd, ri = NearestNeighbors(n_neighbors=k + 1, algorithm="brute").fit(X).kneighbors(X[bi])
    assert np.array_equal(bidx.cpu().numpy(), ri[:, 1:])
    np.testing.assert_allclose(bdist.cpu().numpy(), rd[:, 1:] ** 2, rtol=1e-9, atol=1e-12)
    assert not (bidx.cpu().numpy() == bi[:, None]).any()


def test_end_to_end_with_true_neighbours():
    """kNN producer -> fused posterior vs the oracle on the same neighbourhoods."""
    from muygpys_amd.fused import KernelSpec, posterior_mean_var
    from muygpys_amd.neighbors import NN_Wrapper
    from oracle import muygps_oracle as orc
    from tests.util import RTOL, assert_close

    rng = np.random.default_rng(5)
    X = rng.normal(size=(5000, 40))
    y = np.sin(X[:, 0]) + 0.1 * rng.normal(size=5000)
    bi = np.arange(0, 5000, 10)
    Xd, yd, bid = to_dev(X, torch.float32), to_dev(y, torch.float32), to_dev(bi)
    ni, _ = NN_Wrapper(Xd, 30).get_batch_nns(bid)
    mean, var = posterior_mean_var(KernelSpec("matern15", "l2", 5.0, 1e-3), Xd, Xd, bid, ni, yd)
    m_ref, v_ref = orc.posterior_mean_var(orc.Spec("matern15", "l2", 5.0, 1e-3), X, X, bi, ni.cpu().numpy(), y)
    assert_close(mean.cpu().numpy(), m_ref, RTOL["float32"], "mean")
    assert_close(var.cpu().numpy(), v_ref, RTOL["float32"], "var")
