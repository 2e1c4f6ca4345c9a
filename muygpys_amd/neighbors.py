"""Exact brute-force k-nearest-neighbour index producer on the GPU.

The step immediately upstream of the hot path: the reference wraps scikit-learn's exact KNN
(or hnswlib) on the CPU (src/MuyGPyS/neighbors.py:32-262).  ``NN_Wrapper`` here keeps that
interface -- ``get_nns(test)`` / ``get_batch_nns(batch_indices)`` returning ``(indices int64,
squared-l2 distances)`` with the self-match dropped for batch queries (neighbors.py:207-211,
246-250) -- and computes it exactly, in row chunks, as a dense ``|q|^2 + |x|^2 - 2 q.x``
contraction (the one place on the path where a GEMM is the natural shape: rocBLAS/MFMA through
``torch.matmul``) followed by ``topk``.  Candidate distances are then recomputed in difference
form for the k winners, so the returned distances carry no cancellation error.
"""

from __future__ import annotations

from typing import Tuple

import torch


class NN_Wrapper:
    def __init__(self, train: torch.Tensor, nn_count: int, nn_method: str = "exact", chunk: int = 4096, **kwargs):
        if nn_method.lower() != "exact":
            raise NotImplementedError(f"Nearest Neighbor algorithm {nn_method} is not implemented.")
        if not (isinstance(train, torch.Tensor) and train.is_cuda):
            raise TypeError("NN_Wrapper takes a torch tensor on the ROCm device")
        self.train = (train[:, None] if train.ndim == 1 else train).contiguous()
        self.train_count, self.feature_count = self.train.shape
        self.nn_count = int(nn_count)
        self.nn_method = "exact"
        self.chunk = int(chunk)
        self._sq = (self.train.double() ** 2).sum(1).to(self.train.dtype)

    def get_nns(self, test: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """neighbors.py:129-167: nn_count nearest training rows of every test row."""
        test = test[:, None] if test.ndim == 1 else test
        return self._get_nns(test, self.nn_count)

    def get_batch_nns(self, batch_indices: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """neighbors.py:169-211: neighbours of training rows, the row itself excluded.  (The
        reference drops column 0 of a k+1 query and relies on it being the point itself; here the
        self-match is masked explicitly, so duplicate points cannot displace it.)"""
        q = self.train[batch_indices]
        return self._get_nns(q, self.nn_count, exclude=batch_indices)

    def _get_nns(self, samples, nn_count, exclude=None):
        n = samples.shape[0]
        idx = torch.empty((n, nn_count), dtype=torch.int64, device=samples.device)
        dist = torch.empty((n, nn_count), dtype=samples.dtype, device=samples.device)
        for s in range(0, n, self.chunk):
            q = samples[s:s + self.chunk].to(self.train.dtype)
            d2 = self._sq[None, :] - 2.0 * (q @ self.train.T) + (q * q).sum(1)[:, None]
            if exclude is not None:
                rows = torch.arange(q.shape[0], device=q.device)
                d2[rows, exclude[s:s + self.chunk]] = float("inf")
            _, cand = d2.topk(nn_count, dim=1, largest=False)
            # exact squared distances of the winners, difference form, then final order
            diff = q[:, None, :] - self.train[cand]
            dd = (diff * diff).sum(-1)
            order = dd.argsort(dim=1, stable=True)
            idx[s:s + self.chunk] = cand.gather(1, order)
            dist[s:s + self.chunk] = dd.gather(1, order)
        return idx, dist
