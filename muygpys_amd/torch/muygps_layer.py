"""``MuyGPs_layer``: the local-GP head of a deep-kernel model, on the HIP backend.

Reference: torch/muygps_layer.py:24-164 (``MuyGPs_layer``) and
torch/multivariate_muygps_layer.py:24-154 (``MultivariateMuyGPs_layer``).  There the layer runs
the torch backend's crosswise/pairwise tensors -> kernel -> two ``linalg.solve`` and lets autograd
walk back through the ``(b,k,k,d)`` intermediates.  Here ``forward`` issues the same functor calls
with ``lazy=True``, i.e. as handles that ONE fused launch resolves; because the embedded
features require grad the launch is the differentiable one (``muygpys_amd.autograd``) and
``backward`` is one HIP vector-Jacobian kernel that scatters straight into ``x.grad``.
"""

from __future__ import annotations

import torch
from torch import nn

from muygpys_amd import lazy
from muygpys_amd.gp.deformation import Isotropy
from muygpys_amd.gp.hyperparameter import ScalarParam
from muygpys_amd.gp.muygps import MuyGPS


def _targets_table(batch_nn_indices: torch.Tensor, batch_nn_targets: torch.Tensor, train_count: int):
    """The layer is handed gathered responses ``(b, k, R)`` (muygps_layer.py:88-92); the fused kernels
    gather by index from a ``(train_count, R)`` table.  Rows never referenced stay zero."""
    b, k = batch_nn_indices.shape
    nt = batch_nn_targets.reshape(b * k, -1)
    table = torch.zeros((train_count, nt.shape[1]), device=nt.device, dtype=nt.dtype)
    table[batch_nn_indices.reshape(-1)] = nt
    return table


class MuyGPs_layer(nn.Module):
    """A MuyGPs model as a ``torch.nn.Module`` (same constructor and return values as the reference).

    ``forward(x)`` takes the embedded training features ``(train_count, embed_dim)`` and returns
    ``(predictions (b, R), variances (b,))`` for the batch given at construction; only Isotropy with a
    scalar length scale is accepted, like muygps_layer.py:107-121.
    """

    def __init__(self, muygps_model: MuyGPS, batch_indices, batch_nn_indices, batch_targets, batch_nn_targets):
        super().__init__()
        if not isinstance(muygps_model.kernel.deformation, Isotropy):
            raise NotImplementedError(
                f"MuyGPyS/torch optimization does not support {type(muygps_model.kernel.deformation)} deformations"
            )
        if not isinstance(muygps_model.kernel.deformation.length_scale, ScalarParam):
            raise NotImplementedError(
                "MuyGPyS/torch optimization does not support "
                f"{type(muygps_model.kernel.deformation.length_scale)} length scales"
            )
        self.muygps_model = muygps_model
        self.deformation = muygps_model.kernel.deformation
        self.length_scale = muygps_model.kernel.deformation.length_scale._val
        self.batch_indices = batch_indices
        self.batch_nn_indices = batch_nn_indices
        self.batch_targets = batch_targets
        self.batch_nn_targets = batch_nn_targets

    def forward(self, x):
        self.muygps_model._make()
        crosswise = self.deformation.crosswise_tensor(x, x, self.batch_indices, self.batch_nn_indices, lazy=True)
        pairwise = self.deformation.pairwise_tensor(x, self.batch_nn_indices, lazy=True)
        Kcross = self.muygps_model.kernel(crosswise)
        Kin = self.muygps_model.kernel(pairwise)
        nn_targets = lazy.LazyTargets(
            _targets_table(self.batch_nn_indices, self.batch_nn_targets.to(x.dtype), x.shape[0]),
            self.batch_nn_indices,
        )
        predictions = self.muygps_model.posterior_mean(Kin, Kcross, nn_targets)
        variances = self.muygps_model.posterior_variance(Kin, Kcross)
        return predictions, variances


class MultivariateMuyGPs_layer(nn.Module):
    """One independent MuyGPs model per response column (multivariate_muygps_layer.py:99-154).

    ``multivariate_muygps_model`` is anything with a ``models`` sequence of :class:`MuyGPS` (the
    reference's deprecated ``MultivariateMuyGPS`` container is not part of this package).  Returns
    ``(predictions (b, R), variances (b, R))``."""

    def __init__(self, multivariate_muygps_model, batch_indices, batch_nn_indices, batch_targets, batch_nn_targets):
        super().__init__()
        models = getattr(multivariate_muygps_model, "models", multivariate_muygps_model)
        self.multivariate_muygps_model = multivariate_muygps_model
        self.models = list(models)
        self.deformation = self.models[0].kernel.deformation
        self.batch_indices = batch_indices
        self.batch_nn_indices = batch_nn_indices
        self.batch_targets = batch_targets
        self.batch_nn_targets = batch_nn_targets

    def forward(self, x):
        table = _targets_table(self.batch_nn_indices, self.batch_nn_targets.to(x.dtype), x.shape[0])
        if table.shape[1] != len(self.models):
            raise ValueError(f"{len(self.models)} models for {table.shape[1]} response columns")
        predictions, variances = [], []
        for r, model in enumerate(self.models):
            model._make()
            deformation = model.kernel.deformation
            crosswise = deformation.crosswise_tensor(x, x, self.batch_indices, self.batch_nn_indices, lazy=True)
            pairwise = deformation.pairwise_tensor(x, self.batch_nn_indices, lazy=True)
            Kcross, Kin = model.kernel(crosswise), model.kernel(pairwise)
            nn_targets = lazy.LazyTargets(table[:, r].contiguous(), self.batch_nn_indices)
            predictions.append(model.posterior_mean(Kin, Kcross, nn_targets).reshape(-1))
            variances.append(model.posterior_variance(Kin, Kcross).reshape(-1))
        return torch.stack(predictions, dim=1), torch.stack(variances, dim=1)
