"""CPU oracle for the MuyGPyS batched local-GP hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the reference's *numpy backend* for the
path named in BASELINE.json (distance tensors -> Matern/RBF kernel -> nugget ->
per-neighbourhood solve -> posterior mean / variance / sigma_sq / LOOCV loss).
It executes the same operation sequence as the reference (materialise the
difference tensors, reduce, elementwise kernel, ``linalg.solve`` twice ...), so it
doubles as the "what the reference does on a CPU" baseline in ``bench.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``muygpys_amd/`` imports it: the
product path is the HIP library and fails loudly when that is missing.

Parity pin: every function here is checked against outputs of the real reference
(imported from /root/reference in the build container) through the committed
fixtures ``tests/golden/*.npz`` (generator: ``tests/golden/make_golden.py``), see
``tests/test_oracle_golden.py``.

All citations are ``/root/reference/src/MuyGPyS/...`` file:line.
"""

from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# T1-T4  tensors                                  _src/gp/tensors/numpy.py
# --------------------------------------------------------------------------


def crosswise_tensor(data, nn_data, data_indices, nn_indices):
    """_src/gp/tensors/numpy.py:47-58 -- (b,k,d) query-minus-neighbour differences."""
    locations = data[data_indices]
    points = nn_data[nn_indices]
    if data.ndim == 1:
        return locations[..., :, None, None] - points[..., None]
    return locations[..., :, None, :] - points


def pairwise_tensor(data, nn_indices):
    """_src/gp/tensors/numpy.py:61-69 -- (b,k,k,d) all-pairs differences."""
    points = data[nn_indices]
    if data.ndim == 1:
        return points[..., :, None, None] - points[..., None, :, None]
    return points[..., None, :] - points[..., None, :, :]


def F2(diffs):
    """_src/gp/tensors/numpy.py:89-90."""
    return np.sum(diffs**2, axis=-1)


def l2(diffs):
    """_src/gp/tensors/numpy.py:93-94."""
    return np.sqrt(F2(diffs))


def batch_features_tensor(features, batch_indices):
    """_src/gp/tensors/numpy.py:40-44."""
    return features[batch_indices]


def make_heteroscedastic_tensor(measurement_noise, batch_nn_indices):
    """_src/gp/tensors/numpy.py:11-15."""
    return measurement_noise[batch_nn_indices]


def fast_nn_update(train_nn_indices):
    """_src/gp/tensors/numpy.py:97-108 -- prepend self, drop the last neighbour."""
    n = train_nn_indices.shape[0]
    return np.concatenate(
        (np.arange(n, dtype=train_nn_indices.dtype)[:, None], train_nn_indices[:, :-1]),
        axis=1,
    )


# --------------------------------------------------------------------------
# D1/D2  deformation                 gp/deformation/{isotropy,anisotropy,metric}.py
# --------------------------------------------------------------------------

METRICS = ("l2", "F2")


def metric_reduce(diffs, metric):
    return l2(diffs) if metric == "l2" else F2(diffs)


def apply_length_scale(dists, length_scale, metric):
    """gp/deformation/metric.py:241 (l2: x/l) and :264 (F2: x/l**2)."""
    if metric == "l2":
        return dists / length_scale
    return dists / length_scale**2


def isotropy(dists, length_scale, metric):
    """gp/deformation/isotropy.py:60-89 -- dists are already metric-reduced."""
    return apply_length_scale(dists, length_scale, metric)


def anisotropy(diffs, length_scales, metric):
    """gp/deformation/anisotropy.py:43-70 -- metric(diffs / l_vec)."""
    length_scales = np.asarray(length_scales)
    if diffs.shape[-1] != length_scales.shape[0]:
        raise ValueError(
            f"Difference tensor of shape {diffs.shape} must have final dimension "
            f"size of {len(length_scales)}"
        )
    return metric_reduce(diffs / length_scales, metric)


# --------------------------------------------------------------------------
# K1/K2  kernels                                   _src/gp/kernels/numpy.py
# --------------------------------------------------------------------------


def rbf_fn(squared_dists):
    """_src/gp/kernels/numpy.py:12-13."""
    return np.exp(-squared_dists / 2.0)


def matern_05_fn(dists):
    """:16-17."""
    return np.exp(-dists)


def matern_15_fn(dists):
    """:20-22."""
    K = dists * np.sqrt(3)
    return (1.0 + K) * np.exp(-K)


def matern_25_fn(dists):
    """:25-27."""
    K = dists * np.sqrt(5)
    return (1.0 + K + K**2 / 3.0) * np.exp(-K)


def matern_inf_fn(dists):
    """:30-31."""
    return np.exp(-(dists**2) / 2.0)


def matern_gen_fn(dists, smoothness):
    """:34-43 -- general smoothness through the modified Bessel function (scipy.special.kv, the
    reference's own dependency); zeros become eps.  Returns a new array (the reference overwrites
    its argument)."""
    from scipy.special import gamma, kv

    K = np.array(dists, dtype=np.float64, copy=True)
    K[K == 0.0] += np.finfo(float).eps
    tmp = np.sqrt(2 * smoothness) * K
    return (2 ** (1.0 - smoothness)) / gamma(smoothness) * tmp**smoothness * kv(smoothness, tmp)


KERNELS = {
    "rbf": rbf_fn,
    "matern05": matern_05_fn,
    "matern15": matern_15_fn,
    "matern25": matern_25_fn,
    "maternInf": matern_inf_fn,
}


def kernel_for_smoothness(nu):
    """gp/kernels/matern.py:61-81 -- special-case selection for fixed nu."""
    table = {0.5: "matern05", 1.5: "matern15", 2.5: "matern25", np.inf: "maternInf"}
    if nu not in table:
        return lambda dists: matern_gen_fn(dists, nu)
    return table[nu]


# --------------------------------------------------------------------------
# N1/N2  noise                                       _src/gp/noise/numpy.py
# --------------------------------------------------------------------------


def homoscedastic_perturb(Kin, noise_variance):
    """_src/gp/noise/numpy.py:9-14 (3-D case)."""
    if Kin.ndim != 3:
        raise ValueError(
            "homoscedastic perturbation is not implemented for tensors of "
            f"shape {Kin.shape}"
        )
    return Kin + noise_variance * np.eye(Kin.shape[1])


def heteroscedastic_perturb(Kin, noise_variances):
    """_src/gp/noise/numpy.py:56-67 -- Kin[b,i,i] += eps[b,i]."""
    ret = Kin.copy()
    b, k, _ = Kin.shape
    idx = np.arange(k)
    ret[:, idx, idx] += noise_variances.reshape(b, k)
    return ret


# --------------------------------------------------------------------------
# S1-S3  solves                 _src/gp/muygps/numpy.py, _src/optimize/scale/numpy.py
# --------------------------------------------------------------------------


def posterior_mean(Kin, Kcross, nn_targets):
    """_src/gp/muygps/numpy.py:17-41 -- F = solve(Kin, Kcross); mean = F^T Y.

    Kin (b,k,k), Kcross (b,k), nn_targets (b,k) or (b,k,R) -> (b,) or (b,R).
    """
    b, k, _ = Kin.shape
    Y = nn_targets.reshape(b, k, -1)
    F = np.linalg.solve(Kin, Kcross.reshape(b, k, 1))
    ret = np.swapaxes(F, -2, -1) @ Y  # (b,1,R)
    return ret.reshape((b,) + nn_targets.shape[2:])


def diagonal_variance(Kin, Kcross, Kout=1.0):
    """_src/gp/muygps/numpy.py:44-67 -- Kout - Kcross^T solve(Kin, Kcross)."""
    b, k, _ = Kin.shape
    Kc = Kcross.reshape(b, k, 1)
    F = np.linalg.solve(Kin, Kc)
    Kpost = np.swapaxes(F, -2, -1) @ Kc
    return Kout - Kpost.reshape(b)


def analytic_scale_optim_unnormalized(Kin, nn_targets):
    """_src/optimize/scale/numpy.py:9-15 -- sum_b y^T K^-1 y."""
    Y = np.atleast_3d(nn_targets)
    return np.sum(np.einsum("ijk,ijk->ik", Y, np.linalg.solve(Kin, Y)))


def analytic_scale_optim(Kin, nn_targets):
    """_src/optimize/scale/numpy.py:18-34 -- / (b*k); R>1 is rejected like the reference."""
    b, k, _ = Kin.shape
    Y = nn_targets.reshape(b, k, 1)
    return analytic_scale_optim_unnormalized(Kin, Y) / (b * k)


def analytic_scale_per_response(Kin, nn_targets):
    """Build extension: sigma_sq_r = sum_b y_r^T K^-1 y_r / (b k).

    The reference only reaches this through the deprecated per-model loop
    gp/multivariate_muygps.py:375-382; the maths is Appendix A per column.
    """
    b, k, _ = Kin.shape
    Y = nn_targets.reshape(b, k, -1)
    S = np.linalg.solve(Kin, Y)
    return np.einsum("bkr,bkr->r", Y, S) / (b * k)


def fast_posterior_mean_precompute(Kin, train_nn_targets_fast):
    """_src/gp/muygps/numpy.py:88-95 -- coefficients C_i = K_i^-1 y_i."""
    if train_nn_targets_fast.ndim == 2:
        train_nn_targets_fast = train_nn_targets_fast[:, :, None]
    return np.squeeze(np.linalg.solve(Kin, train_nn_targets_fast))


def fast_posterior_mean(Kcross, coeffs_tensor):
    """_src/gp/muygps/numpy.py:70-77."""
    return np.squeeze(np.einsum("ij,ijk->ik", Kcross, np.atleast_3d(coeffs_tensor)))


# --------------------------------------------------------------------------
# L1/L2  losses                                    _src/optimize/loss/numpy.py
# --------------------------------------------------------------------------


def mse_fn(predictions, targets):
    """:22-31."""
    return np.sum((predictions - targets) ** 2) / np.prod(predictions.shape)


def lool_fn_unscaled(predictions, targets, variances):
    """:34-44 (1-D variance branch only; full-covariance branch is out of scope)."""
    if variances.ndim != 1:
        raise NotImplementedError("full-covariance lool is outside the hot path")
    return np.sum(np.divide((predictions - targets) ** 2, variances) + np.log(variances))


def lool_fn(predictions, targets, variances, scale):
    """:54-61."""
    return lool_fn_unscaled(predictions, targets, scale * variances)


def pseudo_huber_fn(predictions, targets, boundary_scale=1.5):
    """:64-72."""
    return boundary_scale**2 * np.sum(
        np.sqrt(1 + np.divide(targets - predictions, boundary_scale) ** 2) - 1
    )


def looph_fn(predictions, targets, variances, scale, boundary_scale=3.0):
    """:75-117."""
    v = scale * variances
    if v.ndim != 1:
        raise ValueError("looph does not yet support multivariate inference")
    bs2 = boundary_scale**2
    return np.sum(
        2 * bs2 * (np.sqrt(1 + np.divide((targets - predictions) ** 2, bs2 * v)) - 1)
        + np.log(v)
    )


# --------------------------------------------------------------------------
# sharding rule                                          _src/mpi_utils.py:36-41
# --------------------------------------------------------------------------


def chunk_sizes(count, size):
    """floor(count/size) rows each, remainder to the LAST ranks."""
    floor = int(count / size)
    remainder = count - floor * size
    return [floor + 1 if i >= (size - remainder) else floor for i in range(size)]


# --------------------------------------------------------------------------
# H1-H4  composed pipelines (what the reference's callers do, in its order)
# --------------------------------------------------------------------------


class Spec:
    """Plain description of a model: kernel, metric, deformation, noise.

    kernel: one of KERNELS, or a callable on (scaled) distances such as
    ``lambda r: matern_gen_fn(r, nu)``; metric: "l2"|"F2"; length_scale: scalar (Isotropy) or
    (d,) array (Anisotropy); noise: scalar (homoscedastic) or (N,) per-training-
    point variances (heteroscedastic, gathered with nn_indices like
    _src/gp/tensors/numpy.py:11-15).
    """

    def __init__(self, kernel="matern15", metric="l2", length_scale=1.0, noise=0.0):
        self.kernel = kernel
        self.metric = metric
        self.length_scale = length_scale
        self.noise = noise

    @property
    def anisotropic(self):
        return np.ndim(self.length_scale) == 1

    @property
    def heteroscedastic(self):
        return np.ndim(self.noise) >= 1


def kernel_tensors(spec, crosswise_diffs, pairwise_diffs):
    """gp/kernels/matern.py:148-168 / rbf.py: K = fn(deformation(diffs)).

    For Isotropy the reference hands *distances* to the kernel (isotropy.py:92-161
    reduce with the metric at tensor-construction time); for Anisotropy the raw
    differences (anisotropy.py:73-143).  Both orders are numerically what is done
    here: reduce, then divide (Isotropy) / divide, then reduce (Anisotropy).
    """
    fn = spec.kernel if callable(spec.kernel) else KERNELS[spec.kernel]
    if spec.anisotropic:
        Kc = fn(anisotropy(crosswise_diffs, spec.length_scale, spec.metric))
        Kin = fn(anisotropy(pairwise_diffs, spec.length_scale, spec.metric))
    else:
        Kc = fn(isotropy(metric_reduce(crosswise_diffs, spec.metric), spec.length_scale, spec.metric))
        Kin = fn(isotropy(metric_reduce(pairwise_diffs, spec.metric), spec.length_scale, spec.metric))
    return Kc, Kin


def perturb(spec, Kin, nn_indices):
    """gp/noise/homoscedastic.py:90-115 / heteroscedastic.py."""
    if spec.heteroscedastic:
        return heteroscedastic_perturb(
            Kin, make_heteroscedastic_tensor(np.asarray(spec.noise), nn_indices)
        )
    return homoscedastic_perturb(Kin, spec.noise)


def posterior_mean_var(spec, test_features, train_features, batch_indices, nn_indices, train_targets):
    """MuyGPS.make_predict_tensors (gp/muygps.py:406-475) -> kernel -> posterior_mean
    (:164-211) -> *unscaled* posterior variance (gp/variance.py:40).

    Returns (mean (b,) or (b,R), var (b,)).  Two solves of the same matrix, like
    the reference.
    """
    cd = crosswise_tensor(test_features, train_features, batch_indices, nn_indices)
    pd = pairwise_tensor(train_features, nn_indices)
    Kc, Kin = kernel_tensors(spec, cd, pd)
    Kin = perturb(spec, Kin, nn_indices)
    Y = train_targets[nn_indices]
    return posterior_mean(Kin, Kc, Y), diagonal_variance(Kin, Kc, 1.0)


def posterior_mean_var_chunked(spec, test_features, train_features, batch_indices, nn_indices,
                               train_targets, chunk=4096):
    """Same, over row chunks to bound the (chunk,k,k,d) temporary."""
    means, variances = [], []
    for s in range(0, nn_indices.shape[0], chunk):
        m, v = posterior_mean_var(
            spec, test_features, train_features, batch_indices[s:s + chunk],
            nn_indices[s:s + chunk], train_targets,
        )
        means.append(m)
        variances.append(v)
    return np.concatenate(means), np.concatenate(variances)


def sigma_sq(spec, train_features, nn_indices, train_targets):
    """MuyGPS.optimize_scale, gp/muygps.py:373-403 + scale.py:205-217 (iteration_count 1)."""
    pd = pairwise_tensor(train_features, nn_indices)
    fn = spec.kernel if callable(spec.kernel) else KERNELS[spec.kernel]
    if spec.anisotropic:
        Kin = fn(anisotropy(pd, spec.length_scale, spec.metric))
    else:
        Kin = fn(isotropy(metric_reduce(pd, spec.metric), spec.length_scale, spec.metric))
    Kin = perturb(spec, Kin, nn_indices)
    Y = train_targets[nn_indices]
    if Y.ndim == 3 and Y.shape[2] > 1:
        return analytic_scale_per_response(Kin, Y)
    return analytic_scale_optim(Kin, Y)


def loocv_terms(spec, train_features, batch_indices, nn_indices, train_targets):
    """The body of the objective (optimize/loss.py:158-176): mean, sigma_sq, unscaled var."""
    mean, var = posterior_mean_var(
        spec, train_features, train_features, batch_indices, nn_indices, train_targets
    )
    scale = sigma_sq(spec, train_features, nn_indices, train_targets)
    return mean, var, scale, train_targets[batch_indices]


def loocv_objective(spec, train_features, batch_indices, nn_indices, train_targets, loss="lool"):
    """optimize/objective.py:101-103: returns -loss (optimize/loss.py:94,174)."""
    mean, var, scale, y = loocv_terms(spec, train_features, batch_indices, nn_indices, train_targets)
    if loss == "lool":
        return -lool_fn(mean, y, var, scale)
    if loss == "mse":
        return -mse_fn(mean, y)
    if loss == "looph":
        return -looph_fn(mean, y, var, scale)
    if loss == "huber":
        return -pseudo_huber_fn(mean, y)
    raise ValueError(loss)


# --------------------------------------------------------------------------
# Backward pass: what torch autograd computes over the reference's torch backend
# (torch/muygps_layer.py:129-164 -> loss.sum().backward(), examples/muygps_torch.py:425-437).
# Restated as an explicit reverse sweep over the SAME operation sequence as
# posterior_mean_var above (each forward op's adjoint, last to first), with the dense
# (b,k,k,d) intermediates materialised like autograd keeps them.
# Parity pin: tests/golden/grad_*.npz (generator tests/golden/make_golden_grad.py).
# --------------------------------------------------------------------------

KERNEL_DERIVS = {
    # d kernel / d (its argument)
    "rbf": lambda s: -0.5 * np.exp(-s / 2.0),
    "matern05": lambda r: -np.exp(-r),
    "matern15": lambda r: -3.0 * r * np.exp(-np.sqrt(3) * r),
    "matern25": lambda r: -(5.0 / 3.0) * r * (1.0 + np.sqrt(5) * r) * np.exp(-np.sqrt(5) * r),
    "maternInf": lambda r: -r * np.exp(-(r**2) / 2.0),
}


def _metric_adjoint(diffs, reduced, g_reduced, metric):
    """Adjoint of _l2 / _F2 (_src/gp/tensors/torch.py:81-86); torch.norm's subgradient at 0 is 0."""
    if metric == "F2":
        return 2.0 * g_reduced[..., None] * diffs
    safe = np.where(reduced > 0, reduced, 1.0)
    return np.where(reduced[..., None] > 0, g_reduced[..., None] * diffs / safe[..., None], 0.0)


def _kernel_adjoint(spec, diffs, g_K):
    """Adjoint of kernel_tensors for one difference tensor: returns (g_diffs, g_length_scale)."""
    dfn = KERNEL_DERIVS[spec.kernel]
    if spec.anisotropic:
        ls = np.asarray(spec.length_scale, dtype=diffs.dtype)
        scaled = diffs / ls
        arg = metric_reduce(scaled, spec.metric)
        g_scaled = _metric_adjoint(scaled, arg, g_K * dfn(arg), spec.metric)
        g_ls = -(g_scaled * diffs / ls**2).reshape(-1, ls.shape[0]).sum(axis=0)
        return g_scaled / ls, g_ls
    ell = float(spec.length_scale)
    p = 1 if spec.metric == "l2" else 2
    dist = metric_reduce(diffs, spec.metric)
    g_arg = g_K * dfn(dist / ell**p)
    g_ls = np.array([-(p * g_arg * dist / ell ** (p + 1)).sum()])
    return _metric_adjoint(diffs, dist, g_arg / ell**p, spec.metric), g_ls


def posterior_vjp(spec, test_features, train_features, batch_indices, nn_indices, train_targets,
                  grad_mean, grad_var):
    """Cotangents of (mean, var) = posterior_mean_var(...) contracted with (grad_mean, grad_var).

    Returns a dict: ``test_features`` (n_q,d), ``train_features`` (n,d), ``targets`` (n,R),
    ``length_scale`` (1,) or (d,), ``noise`` (scalar, or (n,) for a heteroscedastic table).
    When the query table IS the training table (LOOCV) add the first two.
    """
    tg = train_targets if train_targets.ndim == 2 else train_targets[:, None]
    gm = grad_mean if grad_mean.ndim == 2 else grad_mean[:, None]
    b, k = nn_indices.shape
    cd = crosswise_tensor(test_features, train_features, batch_indices, nn_indices)
    pd = pairwise_tensor(train_features, nn_indices)
    Kc, Kin0 = kernel_tensors(spec, cd, pd)
    Kin = perturb(spec, Kin0, nn_indices)
    Y = tg[nn_indices]                                   # (b,k,R)
    A = np.linalg.solve(Kin, Kc[..., None])[..., 0]      # F = Kin^-1 Kcross, (b,k)

    # mean = F^T Y ; var = 1 - F^T Kcross
    g_A = np.einsum("br,bkr->bk", gm, Y) - grad_var[:, None] * Kc
    g_Y = A[:, :, None] * gm[:, None, :]
    g_Kc = -grad_var[:, None] * A
    # F = solve(Kin, Kcross):  g_Kcross += Kin^-T g_F ;  g_Kin = -(Kin^-T g_F) F^T
    t = np.linalg.solve(np.swapaxes(Kin, -1, -2), g_A[..., None])[..., 0]
    g_Kc = g_Kc + t
    g_Kin = -t[:, :, None] * A[:, None, :]
    # perturb: Kin = Kin0 + diag(noise)
    g_diag = np.einsum("bii->bi", g_Kin)
    if spec.heteroscedastic:
        g_noise = np.zeros_like(np.asarray(spec.noise, dtype=g_diag.dtype))
        np.add.at(g_noise, nn_indices, g_diag)
    else:
        g_noise = g_diag.sum()
    g_cd, g_ls_c = _kernel_adjoint(spec, cd, g_Kc)
    g_pd, g_ls_p = _kernel_adjoint(spec, pd, g_Kin)
    # crosswise[b,i] = q[batch[b]] - x[nn[b,i]] ; pairwise[b,i,j] = x[nn[b,i]] - x[nn[b,j]]
    g_q = np.zeros_like(test_features, dtype=g_cd.dtype)
    g_x = np.zeros_like(train_features, dtype=g_cd.dtype)
    np.add.at(g_q, batch_indices, g_cd.sum(axis=1))
    np.add.at(g_x, nn_indices, -g_cd)
    np.add.at(g_x, nn_indices, g_pd.sum(axis=2))
    np.add.at(g_x, nn_indices, -g_pd.sum(axis=1))
    g_t = np.zeros_like(tg, dtype=g_Y.dtype)
    np.add.at(g_t, nn_indices, g_Y)
    return {
        "test_features": g_q,
        "train_features": g_x,
        "targets": g_t if train_targets.ndim == 2 else g_t[:, 0],
        "length_scale": g_ls_c + g_ls_p,
        "noise": g_noise,
    }
