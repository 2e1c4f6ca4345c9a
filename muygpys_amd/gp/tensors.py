"""Public tensor helpers (reference contract: src/MuyGPyS/gp/tensors.py:23-153)."""

from muygpys_amd._src.gp.tensors import (
    _batch_features_tensor,
    _fast_nn_update,
    _make_fast_predict_tensors,
    _make_heteroscedastic_tensor,
)


def make_heteroscedastic_tensor(measurement_noise, batch_nn_indices):
    """(batch, nn) nugget tensor for ``HeteroscedasticNoise`` (tensors.py:23-49)."""
    return _make_heteroscedastic_tensor(measurement_noise, batch_nn_indices)


def fast_nn_update(train_nn_indices):
    """tensors.py:52-90: prepend each point to its own neighbourhood, drop the farthest."""
    return _fast_nn_update(train_nn_indices)


def make_fast_predict_tensors(batch_nn_indices, train_features, train_targets):
    """tensors.py:93-131."""
    return _make_fast_predict_tensors(batch_nn_indices, train_features, train_targets)


def batch_features_tensor(features, batch_indices):
    """tensors.py:134-153."""
    return _batch_features_tensor(features, batch_indices)
