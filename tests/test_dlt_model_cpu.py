"""CPU: the dealt-lower-triangle elimination of the fp64 64-slot kernels (csrc/mgp_fused_wave_kernel.h, phase 4D;
docs/HISTORY.md sec. 4.1e), restated in numpy with the kernel's own index maps -- `dlt_col_start`, pair -> (slot, lane),
which slots a step touches, where the pivot and the Schur block sit -- and compared with a direct solve.  The GPU
parity tests (tests/test_gpu_jit.py, tests/test_gpu_fused.py) check the kernel; this pins the scheme itself: posting the
raw column in lane order, junk in finished columns and in the odd upper-triangle elements never reaching a live entry,
one multiply and two FMAs per live slot."""

import numpy as np
import pytest


def dlt_col_start(c, nr2):
    """First pair of column c (mgp_fused_wave_kernel.h: dlt_col_start)."""
    m, b = divmod(c, 2)
    return 2 * m * nr2 - m * (m - 1) + b * (nr2 - m)


def dealt_elimination(K, c, Y, kout=1.0):
    """-> (var, mean (R,), ykinvy (R,), slot-steps) of the augmented system [[K, .], [c^T, kout, .], [Y^T, 0, 0]]."""
    k, R = K.shape[0], Y.shape[1]
    N = k + 1 + R
    nr2 = (N + 1) // 2
    cs = [dlt_col_start(cc, nr2) for cc in range(N + 1)]
    assert all(cs[cc + 1] - cs[cc] == nr2 - cc // 2 for cc in range(N))
    npair = cs[N]
    nsl = (npair + 63) // 64
    S = np.zeros((2 * nr2, N))
    S[:k, :k] = np.tril(K)
    S[k, :k], S[k, k] = c, kout
    S[k + 1:k + 1 + R, :k] = Y.T
    # exchange image in dealt order; what the kernel never writes (upper-triangle halves of odd columns' first pairs, the
    # phantom row of an odd N, the tail of the last slot) is junk
    D = np.full((nsl * 64, 2), 1e30)
    for cc in range(N):
        for i in range(cc, N):
            D[cs[cc] + i // 2 - cc // 2, i & 1] = S[i, cc]
    e = np.arange(nsl * 64)
    col = np.minimum(np.searchsorted(np.asarray(cs), e, side="right") - 1, N - 1)
    row2 = np.where(e < npair, 2 * (e - np.asarray(cs)[col] + col // 2), 0)  # g_dlt_meta: element offsets 2 r and c
    steps = 0
    for j in range(k):
        ep = cs[j]
        p = D[ep, j & 1]                       # pivot: pair cs2(j), element j & 1, lane ep & 63 of slot ep >> 6
        assert p > 0
        s0, s1 = cs[j] >> 6, (cs[j + 1] - 1) >> 6
        stage = {}                               # staging area: ALL pairs of the posting slot(s), in lane order
        for s in range(s0, s1 + 1):
            for lane in range(64):
                stage[64 * (s - s0) + lane] = D[64 * s + lane].copy()
        cj = cs[j] - 64 * s0 - j // 2            # pair (r, j) lies cj + r pairs into the staging area
        first = cs[j + 1] >> 6                   # first slot with a pair of a column right of j
        for s in range(first, nsl):
            for lane in range(64):
                q = 64 * s + lane
                wp = stage.get(cj + row2[q] // 2, np.array([1e30, 1e30]))
                cc = col[q]
                wc = stage.get(cj + cc // 2, np.array([1e30, 1e30]))[cc & 1]
                D[q] += wp * (-(wc / p))
            steps += 1

    def entry(i, cc):
        return D[cs[cc] + i // 2 - cc // 2, i & 1]

    q = k
    return entry(q, q), np.array([-entry(q + 1 + r, q) for r in range(R)]), np.array([-entry(q + 1 + r, q + 1 + r) for r in range(R)]), steps


@pytest.mark.parametrize("k,R", [(50, 1), (33, 1), (40, 2), (61, 1), (62, 1), (58, 4), (10, 1), (31, 3)])
def test_dealt_elimination_matches_a_direct_solve(k, R):
    rng = np.random.default_rng(k * 7 + R)
    X = rng.normal(size=(k + 1, 4))
    Kf = np.exp(-0.5 * ((X[:, None] - X[None]) ** 2).sum(-1))
    K, c = Kf[:k, :k] + 1e-3 * np.eye(k), Kf[k, :k]
    Y = rng.normal(size=(k, R))
    var, mean, yk, steps = dealt_elimination(K, c, Y)
    w = np.linalg.solve(K, c)
    np.testing.assert_allclose(var, 1.0 - c @ w, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(mean, w @ Y, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(yk, np.einsum("kr,kr->r", Y, np.linalg.solve(K, Y)), rtol=1e-8)
    if (k, R) == (50, 1):
        assert steps == 217  # docs/HISTORY.md sec. 4.1e: 217 slot updates = 434 FMAs against 688 group updates of the row form
