"""Batch sampling for training (reference: optimize/batch.py:183-228, ``sample_batch``)."""

from __future__ import annotations

from typing import Tuple

import torch

from muygpys_amd.neighbors import NN_Wrapper


def sample_batch(nbrs_lookup: NN_Wrapper, batch_count: int, train_count: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """``batch_count`` training rows drawn uniformly without replacement (all rows when
    ``train_count <= batch_count``) and their ``nn_count`` nearest neighbours, self excluded.
    Returns ``(batch_indices (b,), batch_nn_indices (b, nn_count))`` as int64 device tensors."""
    device = nbrs_lookup.train.device
    if train_count > batch_count:
        batch_indices = torch.randperm(train_count, device=device)[:batch_count]
    else:
        batch_indices = torch.arange(train_count, device=device, dtype=torch.int64)
    batch_nn_indices, _ = nbrs_lookup.get_batch_nns(batch_indices)
    return batch_indices, batch_nn_indices
