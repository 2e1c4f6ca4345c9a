"""GPU parity of the backward kernel (mgp_posterior_backward_*) and of the layers built on it:
against the reference's own autograd results (tests/golden/grad_*.npz), against the oracle's
reverse sweep on random shapes, and through a short deep-kernel training run."""

import numpy as np
import pytest

from oracle import muygps_oracle as orc
from tests.conftest import grad_spec
from tests.util import RTOL, assert_close, to_dev

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _run(spec_o, xq_np, x_np, bi_np, ni_np, y_np, gm_np, gv_np, dtype, shared, hetero=False):
    """Product path: autograd.posterior forward + .backward(); returns numpy grads."""
    from muygpys_amd.autograd import posterior
    from muygpys_amd.fused import KernelSpec

    td = getattr(torch, dtype)
    x = to_dev(x_np, td).requires_grad_(True)
    xq = x if shared else to_dev(xq_np, td).requires_grad_(True)
    y = to_dev(y_np, td).requires_grad_(True)
    ls = to_dev(np.atleast_1d(spec_o.length_scale), td).requires_grad_(True)
    if np.ndim(spec_o.noise) == 0:
        nz = torch.tensor(float(spec_o.noise), device="cuda", dtype=td, requires_grad=True)
    else:
        nz = to_dev(spec_o.noise, td).requires_grad_(True)
    spec = KernelSpec(kernel=spec_o.kernel, metric=spec_o.metric, length_scale=ls, noise=nz)
    mean, var = posterior(spec, xq, x, to_dev(bi_np), to_dev(ni_np), y)
    gm = to_dev(gm_np.reshape(mean.shape), td)
    ((mean * gm).sum() + (var * to_dev(gv_np, td)).sum()).backward()
    torch.cuda.synchronize()
    out = dict(mean=mean.detach().cpu().numpy(), var=var.detach().cpu().numpy(), x=x.grad.cpu().numpy(),
               y=y.grad.cpu().numpy(), ls=ls.grad.cpu().numpy(), noise=nz.grad.cpu().numpy())
    if not shared:
        out["xq"] = xq.grad.cpu().numpy()
    return out


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_backward_matches_reference_autograd(grad_golden, dtype):
    g, meta = grad_golden, grad_golden["meta"]
    if dtype == "float32" and meta["d"] < 10 and not meta["hetero"] and float(meta["eps"]) < 1e-3:
        pytest.skip("fp32 at tiny nugget / low d is ill-conditioned (reference skips it too, tests/gp.py:514-519)")
    spec, xq = grad_spec(meta, g)
    shared = not meta["separate_test"]
    out = _run(spec, xq, g["features"], g["batch_indices"], g["nn_indices"], g["targets"], g["grad_mean"],
               g["grad_var"], dtype, shared, meta["hetero"])
    rtol = RTOL[dtype] * (3 if dtype == "float32" else 1)  # gradients amplify the solve's conditioning
    assert_close(out["mean"].reshape(g["mean"].shape), g["mean"], RTOL[dtype], "mean")
    assert_close(out["var"], g["var"], RTOL[dtype], "var")
    assert_close(out["x"], g["g_features"], rtol, "g_features")
    if not shared:
        assert_close(out["xq"], g["g_test_features"], rtol, "g_test_features")
    assert_close(out["y"].reshape(g["g_targets"].shape), g["g_targets"], rtol, "g_targets")
    assert_close(out["ls"], g["g_length_scale"], rtol, "g_length_scale")
    if meta["hetero"]:
        assert_close(out["noise"], g["g_noise_table"], rtol, "g_noise_table")
    else:
        assert_close(out["noise"].reshape(()), g["g_noise"], rtol, "g_noise")


CASES = [
    # kernel, metric, aniso, d, k, R, b
    ("matern15", "l2", False, 40, 30, 1, 700),
    ("matern25", "l2", True, 8, 50, 1, 300),
    ("rbf", "F2", False, 40, 64, 4, 200),
    ("matern05", "l2", False, 5, 17, 2, 400),
    ("maternInf", "l2", True, 70, 12, 1, 150),   # d > one LDS feature chunk
    ("matern15", "l2", False, 3, 90, 1, 64),     # k above one wave
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}-{c[1]}-{'aniso' if c[2] else 'iso'}-d{c[3]}-k{c[4]}-R{c[5]}" for c in CASES])
def test_backward_random_vs_oracle(case):
    kernel, metric, aniso, d, k, R, b = case
    rng = np.random.default_rng(100 + CASES.index(case))
    n = 1500
    X = rng.normal(size=(n, d))
    Y = np.sin(X @ rng.normal(size=(d, R)) / np.sqrt(d)) + 0.1 * rng.normal(size=(n, R))
    bi = rng.choice(n, size=b, replace=False)
    ni = np.stack([rng.choice(np.setdiff1d(np.arange(n), [i]), size=k, replace=False) for i in bi])
    # typical squared distance is 2d: a length scale near sqrt(d) keeps the kernel argument O(1)
    ls = np.sqrt(d) * rng.uniform(0.7, 1.5, size=d) if aniso else float(np.sqrt(d))
    spec = orc.Spec(kernel, metric, ls, 1e-2)
    gm, gv = rng.normal(size=(b, R)), rng.normal(size=b)
    ref = orc.posterior_vjp(spec, X, X, bi, ni, Y, gm, gv)
    out = _run(spec, X, X, bi, ni, Y, gm, gv, "float64", True)
    assert_close(out["x"], ref["train_features"] + ref["test_features"], 1e-5, "g_features")
    assert_close(out["y"], ref["targets"], 1e-5, "g_targets")
    assert_close(out["ls"], ref["length_scale"], 1e-5, "g_length_scale")
    assert_close(out["noise"].reshape(()), ref["noise"], 1e-5, "g_noise")


def test_backward_wave_kernel_anisotropy_fp32():
    """Per-feature length scales on the wave-per-neighbourhood backward kernel (csrc/mgp_backward_wave.hip, round 3:
    rows scaled in place, length-scale partials out of the feature sweep), at the headline shape and with two
    neighbourhoods per wave: every gradient against the fp64 oracle's vector-Jacobian product, and against the
    LDS workgroup kernel the call used to fall back to (same maths, fp64)."""
    rng = np.random.default_rng(321)
    n, d, k, R, b = 1500, 40, 30, 1, 513
    X = rng.normal(size=(n, d))
    Y = np.sin(X @ rng.normal(size=(d, R)) / np.sqrt(d)) + 0.1 * rng.normal(size=(n, R))
    bi = rng.choice(n, size=b, replace=False)
    ni = np.stack([rng.choice(np.setdiff1d(np.arange(n), [i]), size=k, replace=False) for i in bi])
    ls = np.sqrt(d) * rng.uniform(0.7, 1.5, size=d)
    spec = orc.Spec("matern15", "l2", ls, 1e-2)
    gm, gv = rng.normal(size=(b, R)), rng.normal(size=b)
    ref = orc.posterior_vjp(spec, X, X, bi, ni, Y, gm, gv)
    out = _run(spec, X, X, bi, ni, Y, gm, gv, "float32", True)
    rtol = RTOL["float32"] * 3
    assert_close(out["x"], ref["train_features"] + ref["test_features"], rtol, "g_features")
    assert_close(out["y"], ref["targets"], rtol, "g_targets")
    assert_close(out["ls"], ref["length_scale"], rtol, "g_length_scale")
    assert_close(out["noise"].reshape(()), ref["noise"], rtol, "g_noise")
    # fp64 at d = 8 (two neighbourhoods of 12 + 2 slots per wave; whole 16-byte groups of two doubles)
    d2, k2 = 8, 12
    X2 = rng.normal(size=(n, d2))
    ni2 = np.stack([rng.choice(np.setdiff1d(np.arange(n), [i]), size=k2, replace=False) for i in bi])
    ls2 = np.sqrt(d2) * rng.uniform(0.7, 1.5, size=d2)
    spec2 = orc.Spec("rbf", "F2", ls2, 1e-2)
    ref2 = orc.posterior_vjp(spec2, X2, X2, bi, ni2, Y, gm, gv)
    out2 = _run(spec2, X2, X2, bi, ni2, Y, gm, gv, "float64", True)
    assert_close(out2["x"], ref2["train_features"] + ref2["test_features"], 1e-5, "g_features (fp64)")
    assert_close(out2["ls"], ref2["length_scale"], 1e-5, "g_length_scale (fp64)")


@pytest.mark.parametrize("aniso", [False, True])
@pytest.mark.parametrize("d,k", [(48, 30), (64, 30), (64, 45), (44, 20)])
def test_backward_wave_kernel_long_rows_fp32(d, k, aniso):
    """Rows of more than ten 16-byte groups: the feature-cotangent sweep of the wave kernel runs in two passes over half
    the groups each (round 4; the single pass spilled hundreds of registers at d = 48 / 64).  32- and 64-slot
    instantiations, iso- and anisotropic, every gradient against the fp64 oracle's vector-Jacobian product."""
    rng = np.random.default_rng(1000 + d + k + int(aniso))
    n, R, b = 1200, 2, 257
    X = rng.normal(size=(n, d))
    Y = np.sin(X @ rng.normal(size=(d, R)) / np.sqrt(d)) + 0.1 * rng.normal(size=(n, R))
    bi = rng.choice(n, size=b, replace=False)
    ni = np.stack([rng.choice(np.setdiff1d(np.arange(n), [i]), size=k, replace=False) for i in bi])
    ls = np.sqrt(d) * rng.uniform(0.7, 1.5, size=d) if aniso else float(np.sqrt(d))
    spec = orc.Spec("matern25", "l2", ls, 1e-2)
    gm, gv = rng.normal(size=(b, R)), rng.normal(size=b)
    ref = orc.posterior_vjp(spec, X, X, bi, ni, Y, gm, gv)
    out = _run(spec, X, X, bi, ni, Y, gm, gv, "float32", True)
    rtol = RTOL["float32"] * 3
    assert_close(out["x"], ref["train_features"] + ref["test_features"], rtol, "g_features")
    assert_close(out["y"], ref["targets"], rtol, "g_targets")
    assert_close(out["ls"], ref["length_scale"], rtol, "g_length_scale")
    assert_close(out["noise"].reshape(()), ref["noise"], rtol, "g_noise")


def test_backward_only_mean_or_only_var():
    """A loss that reads only one output hands the kernel a NULL cotangent for the other."""
    from muygpys_amd.autograd import posterior
    from muygpys_amd.fused import KernelSpec

    rng = np.random.default_rng(5)
    n, d, k, b = 500, 6, 12, 90
    X = rng.normal(size=(n, d))
    Y = rng.normal(size=n)
    bi = rng.choice(n, size=b, replace=False)
    ni = np.stack([rng.choice(np.setdiff1d(np.arange(n), [i]), size=k, replace=False) for i in bi])
    spec_o = orc.Spec("matern15", "l2", 2.0, 1e-2)
    gm, gv = rng.normal(size=b), rng.normal(size=b)
    for which in ("mean", "var"):
        x = to_dev(X, torch.float64).requires_grad_(True)
        mean, var = posterior(KernelSpec("matern15", "l2", 2.0, 1e-2), x, x, to_dev(bi), to_dev(ni),
                              to_dev(Y, torch.float64))
        if which == "mean":
            (mean * to_dev(gm, torch.float64)).sum().backward()
            ref = orc.posterior_vjp(spec_o, X, X, bi, ni, Y, gm, np.zeros(b))
        else:
            (var * to_dev(gv, torch.float64)).sum().backward()
            ref = orc.posterior_vjp(spec_o, X, X, bi, ni, Y, np.zeros(b), gv)
        assert_close(x.grad.cpu().numpy(), ref["train_features"] + ref["test_features"], 1e-5, which)


def test_backward_rejects_oversized_neighbourhoods():
    from muygpys_amd import _lib
    from muygpys_amd.autograd import posterior
    from muygpys_amd.fused import KernelSpec

    kmax = _lib.load().mgp_max_nn_count_backward(8)
    assert 64 <= kmax < 200
    x = torch.randn(kmax + 50, 2, device="cuda", dtype=torch.float64, requires_grad=True)
    ni = torch.arange(kmax + 1, device="cuda")[None, :] + 1
    with pytest.raises(ValueError):
        posterior(KernelSpec(), x, x, torch.zeros(1, dtype=torch.int64, device="cuda"), ni, x[:, 0].detach())


def _toy_problem(dtype, n=600, d=12, k=16, b=200, R=2, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    X = torch.randn(n, d, generator=g, dtype=dtype)
    W = torch.randn(d, R, generator=g, dtype=dtype) / d**0.5
    Y = torch.sin(X @ W) + 0.05 * torch.randn(n, R, generator=g, dtype=dtype)
    return X.cuda(), Y.cuda()


def test_muygps_layer_trains_embedding():
    """Deep-kernel flow of tests/torch/muygps_torch.py:42-170 in miniature: an embedding followed by
    MuyGPs_layer, lool loss, Adam; the loss must fall and gradients must reach the embedding."""
    from muygpys_amd.gp import MuyGPS
    from muygpys_amd.gp.deformation import Isotropy, l2
    from muygpys_amd.gp.hyperparameter import ScalarParam
    from muygpys_amd.gp.kernels import Matern
    from muygpys_amd.gp.noise import HomoscedasticNoise
    from muygpys_amd.neighbors import NN_Wrapper
    from muygpys_amd.torch import MuyGPs_layer

    torch.manual_seed(0)
    X, Y = _toy_problem(torch.float32)
    n, k, b = X.shape[0], 16, 200
    nbrs = NN_Wrapper(X, k, nn_method="exact")
    bi = torch.randperm(n, device="cuda")[:b].sort().values
    ni = nbrs.get_batch_nns(bi)[0]
    model = MuyGPS(
        kernel=Matern(smoothness=ScalarParam(1.5), deformation=Isotropy(l2, length_scale=ScalarParam(2.0))),
        noise=HomoscedasticNoise(1e-2),
    )

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.embedding = torch.nn.Sequential(torch.nn.Linear(12, 10), torch.nn.PReLU(1), torch.nn.Linear(10, 6))
            self.GP_layer = MuyGPs_layer(model, bi, ni, Y[bi], Y[ni])

        def forward(self, x):
            return self.GP_layer(self.embedding(x))

    net = Net().cuda()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    losses = []
    for _ in range(25):
        opt.zero_grad()
        pred, var = net(X)
        assert pred.shape == (b, 2) and var.shape == (b,)
        loss = (((pred - Y[bi]) ** 2) / var[:, None] + torch.log(var)[:, None]).sum()
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.embedding.parameters())
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0] - 1e-3 * abs(losses[0]), losses


def test_muygps_layer_gradient_equals_torch_autograd_of_dense_ops():
    """x.grad from the layer == autograd through a dense torch restatement of the same maths
    (cdist-free: explicit differences, torch.linalg.solve), fp64."""
    from muygpys_amd.gp import MuyGPS
    from muygpys_amd.gp.deformation import Isotropy, l2
    from muygpys_amd.gp.hyperparameter import ScalarParam
    from muygpys_amd.gp.kernels import Matern
    from muygpys_amd.gp.noise import HomoscedasticNoise
    from muygpys_amd.torch import MuyGPs_layer

    X, Y = _toy_problem(torch.float64, n=300, d=5, k=9, b=70, R=2, seed=3)
    n, k, b = 300, 9, 70
    g = torch.Generator().manual_seed(1)
    bi = torch.randperm(n, generator=g)[:b].cuda()
    ni = torch.stack([torch.randperm(n, generator=g)[:k] for _ in range(b)]).cuda()
    ni = torch.where(ni == bi[:, None], (ni + 1) % n, ni)
    model = MuyGPS(
        kernel=Matern(smoothness=ScalarParam(2.5), deformation=Isotropy(l2, length_scale=ScalarParam(1.7))),
        noise=HomoscedasticNoise(1e-2),
    )
    layer = MuyGPs_layer(model, bi, ni, Y[bi], Y[ni])
    x1 = X.clone().requires_grad_(True)
    pred, var = layer(x1)
    gm = torch.randn(pred.shape, dtype=torch.float64, device="cuda")
    gv = torch.randn(var.shape, dtype=torch.float64, device="cuda")
    ((pred * gm).sum() + (var * gv).sum()).backward()

    x2 = X.clone().requires_grad_(True)
    cd = torch.linalg.norm(x2[bi][:, None, :] - x2[ni], dim=-1) / 1.7
    pd = torch.linalg.norm(x2[ni][:, :, None, :] - x2[ni][:, None, :, :] + 0.0, dim=-1) / 1.7

    def m25(r):
        t = r * 5**0.5
        return (1 + t + t**2 / 3) * torch.exp(-t)

    Kin = m25(pd) + 1e-2 * torch.eye(k, dtype=torch.float64, device="cuda")
    Kc = m25(cd)
    F = torch.linalg.solve(Kin, Kc[..., None])
    pred2 = (F.transpose(-1, -2) @ Y[ni])[:, 0, :]
    var2 = 1 - (F[..., 0] * Kc).sum(-1)
    ((pred2 * gm).sum() + (var2 * gv).sum()).backward()
    assert_close(pred.detach().cpu().numpy(), pred2.detach().cpu().numpy(), 1e-5, "pred")
    assert_close(x1.grad.cpu().numpy(), x2.grad.cpu().numpy(), 1e-5, "x.grad")


def test_multivariate_layer_shapes_and_grad():
    from muygpys_amd.gp import MuyGPS
    from muygpys_amd.gp.deformation import Isotropy, l2
    from muygpys_amd.gp.hyperparameter import ScalarParam
    from muygpys_amd.gp.kernels import Matern
    from muygpys_amd.gp.noise import HomoscedasticNoise
    from muygpys_amd.torch import MultivariateMuyGPs_layer

    X, Y = _toy_problem(torch.float32, n=400, d=6, k=10, b=80, R=2, seed=2)
    g = torch.Generator().manual_seed(4)
    bi = torch.randperm(400, generator=g)[:80].cuda()
    ni = torch.stack([torch.randperm(400, generator=g)[:10] for _ in range(80)]).cuda()
    models = [
        MuyGPS(kernel=Matern(smoothness=ScalarParam(0.5), deformation=Isotropy(l2, length_scale=ScalarParam(ell))),
               noise=HomoscedasticNoise(1e-2))
        for ell in (1.0, 2.5)
    ]
    layer = MultivariateMuyGPs_layer(models, bi, ni, Y[bi], Y[ni])
    x = X.clone().requires_grad_(True)
    pred, var = layer(x)
    assert pred.shape == (80, 2) and var.shape == (80, 2)
    (pred.sum() + var.sum()).backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0
    # column r equals a single-model layer with model r
    from muygpys_amd.torch import MuyGPs_layer

    single = MuyGPs_layer(models[1], bi, ni, Y[bi][:, 1:], Y[ni][:, :, 1:])
    p1, v1 = single(X)
    assert_close(p1.reshape(-1).cpu().numpy(), pred[:, 1].detach().cpu().numpy(), 1e-3, "col 1 mean")
    assert_close(v1.reshape(-1).cpu().numpy(), var[:, 1].detach().cpu().numpy(), 1e-3, "col 1 var")


def _svdk(layer_cls, gp_model, bi, ni, Y, in_dim):
    class SVDK(torch.nn.Module):
        """embedding -> GP layer, the shape of MuyGPyS/_test/torch_utils.py:10-93."""

        def __init__(self):
            super().__init__()
            self.embedding = torch.nn.Sequential(
                torch.nn.Linear(in_dim, 30), torch.nn.ELU(1), torch.nn.Linear(30, 10), torch.nn.ELU(1)
            )
            self.batch_indices, self.batch_nn_indices = bi, ni
            self.batch_targets, self.batch_nn_targets = Y[bi], Y[ni]
            self.GP_layer = layer_cls(gp_model, bi, ni, Y[bi], Y[ni])
            self.deformation = self.GP_layer.deformation

        def forward(self, x):
            return self.GP_layer(self.embedding(x))

    return SVDK().cuda()


@pytest.mark.parametrize("multivariate", [False, True])
def test_train_deep_kernel_flow(multivariate):
    """tests/torch/muygps_torch.py:42-170 on the hip backend: sample a batch, train the deep kernel for
    a few iterations with the lool loss, refresh neighbours in the embedded space, predict."""
    from muygpys_amd.examples.muygps_torch import predict_model, train_deep_kernel_muygps, update_nearest_neighbors
    from muygpys_amd.gp import MuyGPS
    from muygpys_amd.gp.deformation import Isotropy, l2
    from muygpys_amd.gp.hyperparameter import ScalarParam
    from muygpys_amd.gp.kernels import Matern
    from muygpys_amd.gp.noise import HomoscedasticNoise
    from muygpys_amd.neighbors import NN_Wrapper
    from muygpys_amd.optimize.batch import sample_batch
    from muygpys_amd.torch import MultivariateMuyGPs_layer, MuyGPs_layer

    torch.manual_seed(1)
    train_count, test_count, feature_count, nn_count, batch_count = 1000, 100, 40, 30, 500
    R = 2 if multivariate else 1
    g = torch.Generator().manual_seed(0)
    X = torch.randn(train_count + test_count, feature_count, generator=g)
    W = torch.randn(feature_count, R, generator=g) / feature_count**0.5
    Yall = torch.sin(X @ W) + 0.05 * torch.randn(train_count + test_count, R, generator=g)
    Xtr, Xte, Ytr, Yte = X[:train_count].cuda(), X[train_count:].cuda(), Yall[:train_count].cuda(), Yall[train_count:].cuda()

    nbrs = NN_Wrapper(Xtr, nn_count, nn_method="exact")
    bi, ni = sample_batch(nbrs, batch_count, train_count)
    assert bi.shape == (batch_count,) and ni.shape == (batch_count, nn_count) and ni.dtype == torch.int64
    assert not (ni == bi[:, None]).any()

    def one():
        return MuyGPS(
            kernel=Matern(smoothness=ScalarParam(0.5), deformation=Isotropy(l2, length_scale=ScalarParam(1.0))),
            noise=HomoscedasticNoise(1e-3),
        )

    if multivariate:
        model = _svdk(MultivariateMuyGPs_layer, [one(), one()], bi, ni, Ytr, feature_count)
    else:
        model = _svdk(MuyGPs_layer, one(), bi, ni, Ytr, feature_count)
    before = [p.detach().clone() for p in model.embedding.parameters()]
    nbrs2, trained = train_deep_kernel_muygps(
        model=model, train_features=Xtr, train_responses=Ytr, batch_indices=bi, nbrs_lookup=nbrs,
        training_iterations=10, optimizer_method=torch.optim.Adam, learning_rate=1e-3, scheduler_decay=0.95,
        loss_function="lool", update_frequency=1,
    )
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, trained.embedding.parameters()))
    assert nbrs2.feature_count == 10 and trained.batch_nn_indices.shape == (batch_count, nn_count)
    trained.eval()
    pred, var = predict_model(model=trained, test_features=Xte, train_features=Xtr, train_responses=Ytr,
                              nbrs_lookup=nbrs2, nn_count=nn_count)
    assert pred.shape == (test_count, R)
    assert var.shape == ((test_count, R) if multivariate else (test_count,))
    assert torch.isfinite(pred).all() and (var > 0).all()
    mse = float(((pred - Yte) ** 2).sum() / test_count)
    assert mse <= 3.0  # the reference's own acceptance bar (tests/torch/muygps_torch.py:56,166-168)
    assert mse < float((Yte**2).sum() / test_count)  # and better than predicting zero
    nbrs3, _ = update_nearest_neighbors(trained, Xtr, Ytr, bi, nn_count)
    assert nbrs3.train_count == train_count


def test_differentiable_losses_match_torch_expression():
    from muygpys_amd.optimize.loss import lool_fn, lool_fn_unscaled, mse_fn

    torch.manual_seed(0)
    p = torch.randn(300, device="cuda", dtype=torch.float64, requires_grad=True)
    t = torch.randn(300, device="cuda", dtype=torch.float64)
    v = (torch.rand(300, device="cuda", dtype=torch.float64) + 0.1).requires_grad_(True)
    for fn, ref in [
        (lambda: lool_fn_unscaled(p, t, v), lambda: ((p - t) ** 2 / v + torch.log(v)).sum()),
        (lambda: lool_fn(p, t, v, 1.7), lambda: ((p - t) ** 2 / (1.7 * v) + torch.log(1.7 * v)).sum()),
        (lambda: mse_fn(p, t), lambda: ((p - t) ** 2).mean()),
    ]:
        p.grad = v.grad = None
        a = fn()
        a.backward()
        gp, gv = p.grad.clone(), None if v.grad is None else v.grad.clone()
        p.grad = v.grad = None
        b = ref()
        b.backward()
        assert_close(a.detach().cpu().numpy(), b.detach().cpu().numpy(), 1e-10, "value")
        assert_close(gp.cpu().numpy(), p.grad.cpu().numpy(), 1e-10, "d/dpred")
        if gv is not None:
            assert_close(gv.cpu().numpy(), v.grad.cpu().numpy(), 1e-10, "d/dvar")


DLT_CASES = [
    # kernel, metric, aniso, b   (k = 50, d = 8, one response: BASELINE config 4's shape)
    ("matern15", "l2", True, 700), ("matern25", "l2", True, 300), ("maternInf", "l2", True, 130), ("matern05", "l2", True, 257),
    ("matern15", "l2", False, 300), ("rbf", "F2", False, 200), ("rbf", "F2", True, 64), ("matern05", "l2", False, 65),
]


@pytest.mark.parametrize("case", DLT_CASES, ids=[f"{c[0]}-{c[1]}-{'aniso' if c[2] else 'iso'}-b{c[3]}" for c in DLT_CASES])
@pytest.mark.parametrize("which", ["mean+var", "mean", "var"])
def test_hyper_parameter_backward_on_the_dealt_triangle(case, which):
    """Round 6: with no feature cotangent asked for, the fp64 k = 50, d = 8 backward runs on the dealt-triangle FORWARD
    kernel (csrc/mgp_backward_dlt.hip: saved factor, back-substitution, pair cotangents in the pair scheme's layout)
    instead of the row-per-lane kernel at one wave per SIMD.  Every hyper-parameter gradient -- length scale(s), noise,
    responses -- against the oracle's vector-Jacobian product (reference: torch autograd over
    torch/muygps_layer.py:129-164).  (That the launch IS the new kernel's is asserted in the next test, through the C
    entry point: autograd runs the backward in a thread of its own, and `mgp_last_kernel_name` is per thread.)"""
    from muygpys_amd.autograd import posterior
    from muygpys_amd.fused import KernelSpec

    kernel, metric, aniso, b = case
    d, k = 8, 50
    rng = np.random.default_rng(900 + DLT_CASES.index(case))
    n = 2500
    X = rng.normal(size=(n, d))
    Y = np.sin(X @ rng.normal(size=(d, 1)) / np.sqrt(d)) + 0.1 * rng.normal(size=(n, 1))
    bi = rng.choice(n, size=b, replace=False)
    ni = np.stack([rng.choice(np.setdiff1d(np.arange(n), [i]), size=k, replace=False) for i in bi])
    ls = np.sqrt(d) * rng.uniform(0.6, 1.6, size=d) if aniso else float(np.sqrt(d)) * 1.1
    spec_o = orc.Spec(kernel, metric, ls, 3e-2)
    gm = rng.normal(size=(b, 1)) if "mean" in which else np.zeros((b, 1))
    gv = rng.normal(size=b) if "var" in which else np.zeros(b)
    ref = orc.posterior_vjp(spec_o, X, X, bi, ni, Y, gm, gv)
    td = torch.float64
    x = to_dev(X, td)  # (no gradient: the hyper-parameter form)
    y = to_dev(Y, td).requires_grad_(True)
    lst = to_dev(np.atleast_1d(ls), td).requires_grad_(True)
    nz = torch.tensor(3e-2, device="cuda", dtype=td, requires_grad=True)
    mean, var = posterior(KernelSpec(kernel=kernel, metric=metric, length_scale=lst, noise=nz), x, x, to_dev(bi), to_dev(ni), y)
    loss = 0.0
    if "mean" in which:
        loss = loss + (mean * to_dev(gm.reshape(mean.shape), td)).sum()
    if "var" in which:
        loss = loss + (var * to_dev(gv, td)).sum()
    loss.backward()
    torch.cuda.synchronize()
    assert_close(y.grad.cpu().numpy(), ref["targets"], 1e-5, "g_targets")
    assert_close(lst.grad.cpu().numpy(), np.atleast_1d(ref["length_scale"]), 1e-5, "g_length_scale")
    assert_close(nz.grad.cpu().numpy().reshape(()), ref["noise"], 1e-5, "g_noise")


def test_dealt_triangle_backward_loocv_form_and_table_noise():
    """The LOOCV form (cotangent of y^T K^-1 y chained in, mgp_loocv_backward_*) and a per-point noise table through
    the same kernel: against the row-per-lane / workgroup kernels' results on a shape they share (the same C entry with
    feature cotangents requested, which the new launcher declines)."""
    from muygpys_amd import _lib

    rng = np.random.default_rng(31)
    n, d, k, b = 3000, 8, 50, 900
    td = torch.float64
    X = to_dev(rng.normal(size=(n, d)), td)
    y = to_dev(rng.normal(size=(n, 1)), td)
    bi = to_dev(rng.choice(n, size=b, replace=False))
    ni = to_dev(rng.integers(0, n, size=(b, k)))
    ni = torch.where(ni == bi[:, None], (ni + 1) % n, ni)
    ls = to_dev(np.sqrt(d) * rng.uniform(0.7, 1.4, size=d), td)
    gm, gv, gy = (to_dev(rng.normal(size=b), td) for _ in range(3))
    noise_tab = to_dev(rng.uniform(1e-2, 5e-2, size=n), td)

    def run(mode, nz_t, with_features, loocv):
        g_l = torch.zeros((b, d), device="cuda", dtype=td)
        g_n = torch.zeros((b, k), device="cuda", dtype=td)
        g_t = torch.zeros_like(y)
        g_x = torch.zeros_like(X) if with_features else None
        info = torch.zeros(1, device="cuda", dtype=torch.int32)
        if loocv:
            rc = _lib.fn("loocv_backward", td)(_lib.ptr(X), d, _lib.ptr(bi), _lib.ptr(ni), b, k, _lib.ptr(y), mode, 2e-2, _lib.ptr(nz_t),
                                               2, 0, _lib.ptr(ls), d, _lib.ptr(gm), _lib.ptr(gv), _lib.ptr(gy), _lib.ptr(g_l), _lib.ptr(g_n),
                                               _lib.ptr(info), _lib.stream_ptr())
        else:
            rc = _lib.fn("posterior_backward", td)(_lib.ptr(X), _lib.ptr(X), d, _lib.ptr(bi), _lib.ptr(ni), b, k, _lib.ptr(y), 1, mode, 2e-2,
                                                   _lib.ptr(nz_t), 2, 0, _lib.ptr(ls), d, _lib.ptr(gm), _lib.ptr(gv), _lib.ptr(g_x), _lib.ptr(g_x),
                                                   _lib.ptr(g_t), _lib.ptr(g_l), _lib.ptr(g_n), _lib.ptr(info), _lib.stream_ptr())
        assert rc == 0 and int(info.item()) == 0
        torch.cuda.synchronize()
        return g_l.cpu().numpy(), g_n.cpu().numpy(), g_t.cpu().numpy(), _lib.last_kernel()

    for mode, nz_t in ((_lib.NOISE_SCALAR, None), (_lib.NOISE_TABLE, noise_tab)):
        new = run(mode, nz_t, False, False)
        old = run(mode, nz_t, True, False)
        assert "backward" in new[3] and "double,64,50,1,8" in new[3], new[3]
        for a_, b_, what in zip(new[:3], old[:3], ("g_ls", "g_noise", "g_targets")):
            assert_close(a_, b_, 1e-9, f"{what} (noise mode {mode})")
    # the LOOCV form: the new kernel against the same launch with the dealt-triangle path switched off is a process-wide
    # switch, so compare with the chain rule instead: gy = 0 must reproduce the plain form's length-scale partials
    gy0 = gy
    gy = torch.zeros_like(gy0)
    lo0 = run(_lib.NOISE_SCALAR, None, False, True)
    pl0 = run(_lib.NOISE_SCALAR, None, False, False)
    assert_close(lo0[0], pl0[0], 1e-12, "LOOCV form at zero y^T K^-1 y cotangent")
    gy = gy0
    lo1 = run(_lib.NOISE_SCALAR, None, False, True)
    assert np.abs(lo1[0] - lo0[0]).max() > 1e-6  # (the third cotangent does flow)


@pytest.mark.parametrize("kernel,aniso,k,d,b", [("matern25", True, 40, 6, 300), ("matern15", False, 62, 16, 129), ("rbf", True, 31, 2, 500)])
def test_dealt_triangle_backward_of_run_time_compiled_shapes(kernel, aniso, k, d, b):
    """Any fp64 shape of the dealt-triangle kernels (33 <= nn_count + 2 <= 64, even feature count): the backward
    instantiation compiled at run time (mgp_jit_prepare_backward, cached on disk), the same oracle comparison, and the
    launch reports it."""
    from muygpys_amd import _lib

    metric = "F2" if kernel == "rbf" else "l2"
    kid = {"rbf": 0, "matern05": 1, "matern15": 2, "matern25": 3, "maternInf": 4}[kernel]
    rc = _lib.load().mgp_jit_prepare_backward(8, k, d, kid)
    if rc == -2:
        pytest.skip("no hiprtc on this machine: the row-per-lane kernels serve the shape")
    assert rc == 0
    rng = np.random.default_rng(1200 + k)
    n = 2000
    X = rng.normal(size=(n, d))
    Y = np.sin(X @ rng.normal(size=(d, 1)) / np.sqrt(d)) + 0.1 * rng.normal(size=(n, 1))
    bi = rng.choice(n, size=b, replace=False)
    ni = np.stack([rng.choice(np.setdiff1d(np.arange(n), [i]), size=k, replace=False) for i in bi])
    ls = np.sqrt(d) * rng.uniform(0.7, 1.5, size=d) if aniso else float(np.sqrt(d))
    spec_o = orc.Spec(kernel, metric, ls, 2e-2)
    gm, gv = rng.normal(size=(b, 1)), rng.normal(size=b)
    ref = orc.posterior_vjp(spec_o, X, X, bi, ni, Y, gm, gv)
    td = torch.float64
    Xd, yd, lsd = to_dev(X, td), to_dev(Y, td), to_dev(np.atleast_1d(ls), td)
    g_l = torch.zeros((b, lsd.numel()), device="cuda", dtype=td)
    g_n = torch.zeros((b, k), device="cuda", dtype=td)
    g_t = torch.zeros_like(yd)
    info = torch.zeros(1, device="cuda", dtype=torch.int32)
    bid, nid, gmd, gvd = to_dev(bi), to_dev(ni), to_dev(gm, td), to_dev(gv, td)
    kid = {"rbf": 0, "matern05": 1, "matern15": 2, "matern25": 3, "maternInf": 4}[kernel]
    rc = _lib.fn("posterior_backward", td)(_lib.ptr(Xd), _lib.ptr(Xd), d, _lib.ptr(bid), _lib.ptr(nid), b, k, _lib.ptr(yd), 1, 0, 2e-2, None,
                                           kid, 0 if metric == "l2" else 1, _lib.ptr(lsd), lsd.numel(), _lib.ptr(gmd), _lib.ptr(gvd), None, None,
                                           _lib.ptr(g_t), _lib.ptr(g_l), _lib.ptr(g_n), _lib.ptr(info), _lib.stream_ptr())
    assert rc == 0 and int(info.item()) == 0
    name = _lib.last_kernel()
    assert "backward" in name and "run-time compiled" in name and f"double,64,{k},1,{d}" in name, name
    assert_close(g_l.sum(0).cpu().numpy(), np.atleast_1d(ref["length_scale"]), 1e-5, "g_length_scale")
    assert_close(g_n.sum().cpu().numpy().reshape(()), ref["noise"], 1e-5, "g_noise")
    assert_close(g_t.cpu().numpy(), ref["targets"], 1e-5, "g_targets")


ROW_CASES = [
    # dtype, kernel, metric, aniso, k, d, b      (17 <= k + 2 <= 32: the row-per-lane form; BASELINE config 3's shape first)
    ("float32", "matern15", "l2", False, 30, 40, 1001), ("float32", "matern15", "l2", True, 30, 40, 600),
    ("float32", "matern25", "l2", True, 30, 40, 257), ("float32", "rbf", "F2", False, 30, 40, 300),
    ("float32", "matern05", "l2", True, 30, 40, 200),    # (difference form: no Gram for the Matern-1/2 kernel)
    ("float32", "maternInf", "l2", True, 20, 16, 333), ("float32", "matern15", "l2", False, 25, 8, 129),
    ("float64", "matern15", "l2", True, 30, 40, 400), ("float64", "matern25", "l2", False, 16, 6, 90),
    # small neighbourhoods ride in the 32-slot kernel; fp32 with 33 .. 64 slots: the 64-slot one, whole rows of multipliers
    ("float32", "matern15", "l2", True, 5, 4, 500), ("float64", "rbf", "F2", False, 3, 2, 77), ("float64", "matern15", "l2", True, 10, 8, 301),
    ("float32", "matern15", "l2", True, 50, 8, 300), ("float32", "matern25", "l2", False, 62, 40, 70), ("float32", "matern05", "l2", True, 40, 16, 129),
    # fp32 rows of up to 128 features in one stage (the reference's torch tutorial: k = 30 on a 100-dimensional embedding)
    ("float32", "matern15", "l2", False, 30, 100, 300), ("float32", "rbf", "F2", True, 30, 100, 129), ("float32", "matern05", "l2", True, 20, 72, 200),
    ("float32", "matern25", "l2", True, 50, 128, 65),
]


@pytest.mark.parametrize("case", ROW_CASES, ids=[f"{c[0]}-{c[1]}-{c[2]}-{'aniso' if c[3] else 'iso'}-k{c[4]}-d{c[5]}" for c in ROW_CASES])
def test_hyper_parameter_backward_row_per_lane_form(case):
    """Round 6, second half: the 32-slot static shapes (BASELINE config 3: k = 30, d = 40, fp32) take their
    hyper-parameter gradients from the forward kernel too -- row per lane, the multipliers kept in the exchange image,
    Gram form, pipelined gather (csrc/mgp_backward_dlt.hip, BWD_ROW).  Every gradient against the oracle's
    vector-Jacobian product (reference: torch autograd over torch/muygps_layer.py:129-164)."""
    from muygpys_amd import _lib

    dtype, kernel, metric, aniso, k, d, b = case
    kid = {"rbf": 0, "matern05": 1, "matern15": 2, "matern25": 3, "maternInf": 4}[kernel]
    es = 4 if dtype == "float32" else 8
    rc = _lib.load().mgp_jit_prepare_backward(es, k, d, kid)
    if rc == -2 and not (es == 4 and k == 30 and d == 40):
        pytest.skip("no hiprtc on this machine: the older kernels serve the shape")
    rng = np.random.default_rng(1500 + ROW_CASES.index(case))
    n = 3000
    X = rng.normal(size=(n, d))
    Y = np.sin(X @ rng.normal(size=(d, 1)) / np.sqrt(d)) + 0.1 * rng.normal(size=(n, 1))
    bi = rng.choice(n, size=b, replace=False)
    ni = np.stack([rng.choice(np.setdiff1d(np.arange(n), [i]), size=k, replace=False) for i in bi])
    ls = np.sqrt(d) * rng.uniform(0.7, 1.5, size=d) if aniso else float(np.sqrt(d))
    spec_o = orc.Spec(kernel, metric, ls, 2e-2)
    gm, gv = rng.normal(size=(b, 1)), rng.normal(size=b)
    ref = orc.posterior_vjp(spec_o, X, X, bi, ni, Y, gm, gv)
    td = getattr(torch, dtype)
    Xd, yd, lsd = to_dev(X, td), to_dev(Y, td), to_dev(np.atleast_1d(ls), td)
    g_l = torch.zeros((b, lsd.numel()), device="cuda", dtype=td)
    g_n = torch.zeros((b, k), device="cuda", dtype=td)
    g_t = torch.zeros_like(yd)
    info = torch.zeros(1, device="cuda", dtype=torch.int32)
    bid, nid, gmd, gvd = to_dev(bi), to_dev(ni), to_dev(gm, td), to_dev(gv, td)
    rc = _lib.fn("posterior_backward", td)(_lib.ptr(Xd), _lib.ptr(Xd), d, _lib.ptr(bid), _lib.ptr(nid), b, k, _lib.ptr(yd), 1, 0, 2e-2, None,
                                           kid, 0 if metric == "l2" else 1, _lib.ptr(lsd), lsd.numel(), _lib.ptr(gmd), _lib.ptr(gvd), None, None,
                                           _lib.ptr(g_t), _lib.ptr(g_l), _lib.ptr(g_n), _lib.ptr(info), _lib.stream_ptr())
    assert rc == 0 and int(info.item()) == 0
    name = _lib.last_kernel()
    assert "backward" in name and f",{32 if k + 2 <= 32 else 64},{k},1,{d}," in name, name
    rtol = 1e-5 if dtype == "float64" else 3e-3
    assert_close(g_l.double().sum(0).cpu().numpy(), np.atleast_1d(ref["length_scale"]), rtol, "g_length_scale")
    assert_close(g_n.double().sum().cpu().numpy().reshape(()), ref["noise"], rtol, "g_noise")
    assert_close(g_t.double().cpu().numpy(), ref["targets"], rtol, "g_targets")


@pytest.mark.parametrize("case", ROW_CASES, ids=[f"{c[0]}-{c[1]}-{c[2]}-{'aniso' if c[3] else 'iso'}-k{c[4]}-d{c[5]}" for c in ROW_CASES])
def test_feature_cotangents_row_per_lane_form(case):
    """Round 6, last part: the same instantiations also emit the FEATURE cotangents (the gradient an embedding network in
    front of the model asks for: reference torch/muygps_layer.py:129-164 under autograd) -- the pair cotangents laid out
    as a symmetric image, one sweep over the tile's rows, atomic adds along the features of a row.  Separate query and
    neighbour tables (two gradient buffers) and one shared table (one buffer, both kinds of row accumulate in it), with
    every other gradient in the same launch, against the oracle's vector-Jacobian product."""
    from muygpys_amd import _lib

    dtype, kernel, metric, aniso, k, d, b = case
    kid = {"rbf": 0, "matern05": 1, "matern15": 2, "matern25": 3, "maternInf": 4}[kernel]
    es = 4 if dtype == "float32" else 8
    rc = _lib.load().mgp_jit_prepare_backward(es, k, d, kid)
    if rc == -2 and not (es == 4 and k == 30 and d == 40):
        pytest.skip("no hiprtc on this machine: the older kernels serve the shape")
    rng = np.random.default_rng(2500 + ROW_CASES.index(case))
    n, nq = 3000, b + 17
    X = rng.normal(size=(n, d))
    Xq = rng.normal(size=(nq, d))
    Y = np.sin(X @ rng.normal(size=(d, 1)) / np.sqrt(d)) + 0.1 * rng.normal(size=(n, 1))
    ls = np.sqrt(d) * rng.uniform(0.7, 1.5, size=d) if aniso else float(np.sqrt(d))
    spec_o = orc.Spec(kernel, metric, ls, 2e-2)
    gm, gv = rng.normal(size=(b, 1)), rng.normal(size=b)
    td = getattr(torch, dtype)
    rtol = 1e-5 if dtype == "float64" else 3e-3
    for shared in (False, True):
        if shared:
            bi = rng.choice(n, size=b, replace=False)
            ni = np.stack([rng.choice(np.setdiff1d(np.arange(n), [i]), size=k, replace=False) for i in bi])
            xq_np = X
        else:
            bi = rng.choice(nq, size=b, replace=False)
            ni = np.stack([rng.choice(n, size=k, replace=False) for _ in bi])
            xq_np = Xq
        ref = orc.posterior_vjp(spec_o, xq_np, X, bi, ni, Y, gm, gv)
        Xd, Xqd, yd, lsd = to_dev(X, td), to_dev(xq_np, td), to_dev(Y, td), to_dev(np.atleast_1d(ls), td)
        g_l = torch.zeros((b, lsd.numel()), device="cuda", dtype=td)
        g_n = torch.zeros((b, k), device="cuda", dtype=td)
        g_t = torch.zeros_like(yd)
        g_x = torch.zeros_like(Xd)
        g_q = g_x if shared else torch.zeros_like(Xqd)
        info = torch.zeros(1, device="cuda", dtype=torch.int32)
        bid, nid, gmd, gvd = to_dev(bi), to_dev(ni), to_dev(gm, td), to_dev(gv, td)
        rc = _lib.fn("posterior_backward", td)(_lib.ptr(Xd if shared else Xqd), _lib.ptr(Xd), d, _lib.ptr(bid), _lib.ptr(nid), b, k,
                                               _lib.ptr(yd), 1, 0, 2e-2, None, kid, 0 if metric == "l2" else 1, _lib.ptr(lsd), lsd.numel(),
                                               _lib.ptr(gmd), _lib.ptr(gvd), _lib.ptr(g_q), _lib.ptr(g_x), _lib.ptr(g_t), _lib.ptr(g_l),
                                               _lib.ptr(g_n), _lib.ptr(info), _lib.stream_ptr())
        assert rc == 0 and int(info.item()) == 0
        name = _lib.last_kernel()
        assert "backward" in name and f",{32 if k + 2 <= 32 else 64},{k},1,{d}," in name, name
        if shared:
            assert_close(g_x.double().cpu().numpy(), ref["train_features"] + ref["test_features"], rtol, "g_features (one table)")
        else:
            assert_close(g_x.double().cpu().numpy(), ref["train_features"], rtol, "g_train_features")
            assert_close(g_q.double().cpu().numpy(), ref["test_features"], rtol, "g_test_features")
        assert_close(g_l.double().sum(0).cpu().numpy(), np.atleast_1d(ref["length_scale"]), rtol, "g_length_scale")
        assert_close(g_n.double().sum().cpu().numpy().reshape(()), ref["noise"], rtol, "g_noise")
        assert_close(g_t.double().cpu().numpy(), ref["targets"], rtol, "g_targets")
    # only the neighbour rows' cotangents asked for: the query rows' stay out (and nothing else is written)
    g_x2 = torch.zeros_like(Xd)
    rc = _lib.fn("posterior_backward", td)(_lib.ptr(Xd), _lib.ptr(Xd), d, _lib.ptr(bid), _lib.ptr(nid), b, k, _lib.ptr(yd), 1, 0, 2e-2, None,
                                           kid, 0 if metric == "l2" else 1, _lib.ptr(lsd), lsd.numel(), _lib.ptr(gmd), _lib.ptr(gvd), None,
                                           _lib.ptr(g_x2), None, None, None, _lib.ptr(info), _lib.stream_ptr())
    assert rc == 0 and "backward" in _lib.last_kernel()
    assert_close(g_x2.double().cpu().numpy(), ref["train_features"], rtol, "g_train_features alone")


@pytest.mark.parametrize("dtype,kernel,aniso,k,d,R,b", [("float32", "matern15", False, 30, 40, 10, 401), ("float32", "rbf", True, 30, 40, 3, 300),
                                                       ("float64", "matern25", True, 10, 8, 2, 257), ("float32", "matern15", True, 50, 8, 4, 130)])
def test_backward_on_the_forward_kernel_with_several_responses(dtype, kernel, aniso, k, d, R, b):
    """Several responses (the reference's torch tutorial trains on ten one-hot columns, docs/examples/torch_tutorial.ipynb)
    on the row-per-lane backward instantiations: the system still has ONE right-hand side, the combined column
    Y g_mean formed as the rows' responses are fetched; the responses' cotangents are g_mean,r a_j.  Everything against
    the oracle's vector-Jacobian product."""
    from muygpys_amd import _lib

    kid = {"rbf": 0, "matern05": 1, "matern15": 2, "matern25": 3, "maternInf": 4}[kernel]
    es = 4 if dtype == "float32" else 8
    rc = _lib.load().mgp_jit_prepare_backward(es, k, d, kid)
    if rc == -2 and not (es == 4 and k == 30 and d == 40):
        pytest.skip("no hiprtc on this machine: the older kernels serve the shape")
    rng = np.random.default_rng(77 + R)
    n = 2500
    X = rng.normal(size=(n, d))
    Y = np.sin(X @ rng.normal(size=(d, R)) / np.sqrt(d)) + 0.1 * rng.normal(size=(n, R))
    bi = rng.choice(n, size=b, replace=False)
    ni = np.stack([rng.choice(np.setdiff1d(np.arange(n), [i]), size=k, replace=False) for i in bi])
    ls = np.sqrt(d) * rng.uniform(0.7, 1.5, size=d) if aniso else float(np.sqrt(d))
    metric = "F2" if kernel == "rbf" else "l2"
    spec_o = orc.Spec(kernel, metric, ls, 2e-2)
    gm, gv = rng.normal(size=(b, R)), rng.normal(size=b)
    ref = orc.posterior_vjp(spec_o, X, X, bi, ni, Y, gm, gv)
    td = getattr(torch, dtype)
    Xd, yd, lsd = to_dev(X, td), to_dev(Y, td), to_dev(np.atleast_1d(ls), td)
    g_l = torch.zeros((b, lsd.numel()), device="cuda", dtype=td)
    g_n = torch.zeros((b, k), device="cuda", dtype=td)
    g_t, g_x = torch.zeros_like(yd), torch.zeros_like(Xd)
    info = torch.zeros(1, device="cuda", dtype=torch.int32)
    bid, nid, gmd, gvd = to_dev(bi), to_dev(ni), to_dev(gm, td), to_dev(gv, td)
    rc = _lib.fn("posterior_backward", td)(_lib.ptr(Xd), _lib.ptr(Xd), d, _lib.ptr(bid), _lib.ptr(nid), b, k, _lib.ptr(yd), R, 0, 2e-2, None,
                                           kid, 0 if metric == "l2" else 1, _lib.ptr(lsd), lsd.numel(), _lib.ptr(gmd), _lib.ptr(gvd),
                                           _lib.ptr(g_x), _lib.ptr(g_x), _lib.ptr(g_t), _lib.ptr(g_l), _lib.ptr(g_n), _lib.ptr(info),
                                           _lib.stream_ptr())
    assert rc == 0 and int(info.item()) == 0
    name = _lib.last_kernel()
    assert "backward" in name and f",{32 if k + 2 <= 32 else 64},{k},1,{d}," in name, name
    rtol = 1e-5 if dtype == "float64" else 3e-3
    assert_close(g_x.double().cpu().numpy(), ref["train_features"] + ref["test_features"], rtol, "g_features")
    assert_close(g_t.double().cpu().numpy(), ref["targets"], rtol, "g_targets")
    assert_close(g_l.double().sum(0).cpu().numpy(), np.atleast_1d(ref["length_scale"]), rtol, "g_length_scale")
    assert_close(g_n.double().sum().cpu().numpy().reshape(()), ref["noise"], rtol, "g_noise")
    # only the variance's cotangent (no g_mean): the combined column is zero
    g_x.zero_(); g_t.zero_()
    rc = _lib.fn("posterior_backward", td)(_lib.ptr(Xd), _lib.ptr(Xd), d, _lib.ptr(bid), _lib.ptr(nid), b, k, _lib.ptr(yd), R, 0, 2e-2, None,
                                           kid, 0 if metric == "l2" else 1, _lib.ptr(lsd), lsd.numel(), None, _lib.ptr(gvd),
                                           _lib.ptr(g_x), _lib.ptr(g_x), _lib.ptr(g_t), _lib.ptr(g_l), _lib.ptr(g_n), _lib.ptr(info),
                                           _lib.stream_ptr())
    assert rc == 0
    ref0 = orc.posterior_vjp(spec_o, X, X, bi, ni, Y, np.zeros_like(gm), gv)
    assert_close(g_x.double().cpu().numpy(), ref0["train_features"] + ref0["test_features"], rtol, "g_features (variance only)")
    assert float(g_t.abs().max()) == 0.0
