"""Tensor family: gathers, difference tensors, metric reductions (reference name list: _src/gp/tensors/__init__.py:8-31)."""

from muygpys_amd._src.util import export_backend

__all__ = export_backend(
    __name__,
    globals(),
    """
    _make_fast_predict_tensors
    _batch_features_tensor
    _crosswise_differences
    _crosswise_tensor
    _pairwise_differences
    _pairwise_tensor
    _fast_nn_update
    _make_heteroscedastic_tensor
    _F2
    _l2
    """,
)
