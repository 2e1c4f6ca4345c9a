from muygpys_amd._src.util import _collect_implementation

(
    _make_fast_predict_tensors,
    _batch_features_tensor,
    _crosswise_differences,
    _crosswise_tensor,
    _pairwise_differences,
    _pairwise_tensor,
    _fast_nn_update,
    _make_heteroscedastic_tensor,
    _F2,
    _l2,
) = _collect_implementation(
    "muygpys_amd._src.gp.tensors",
    "_make_fast_predict_tensors",
    "_batch_features_tensor",
    "_crosswise_differences",
    "_crosswise_tensor",
    "_pairwise_differences",
    "_pairwise_tensor",
    "_fast_nn_update",
    "_make_heteroscedastic_tensor",
    "_F2",
    "_l2",
)
