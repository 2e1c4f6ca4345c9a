"""Exact brute-force k-nearest-neighbour index producer on the GPU.

The step immediately upstream of the hot path: the reference wraps scikit-learn's exact KNN
(or hnswlib) on the CPU (src/MuyGPyS/neighbors.py:32-262).  ``NN_Wrapper`` here keeps that
interface -- ``get_nns(test)`` / ``get_batch_nns(batch_indices)`` returning ``(indices int64,
squared-l2 distances)`` with the self-match dropped for batch queries (neighbors.py:207-211,
246-250) -- and computes it exactly.

* fp32, ``d % 4 == 0``, ``d <= 64``, ``k <= 64`` (the shapes of the hot path): the fused MFMA
  scan ``mgp_knn_scan_f32`` (``csrc/mgp_knn.hip``) -- distances on the matrix cores, one compare
  per pair against the query's running k-th best, rare survivors merged into per-query k-best
  lists -- after the lists have been initialised from the first rows of the table on the dense
  path.  Nothing of size (queries x train) is ever materialised.
* anything else: the dense path, in row chunks, ``|q|^2 + |x|^2 - 2 q.x`` through
  ``torch.matmul`` (rocBLAS) followed by ``topk``.

Either way the k winners are then re-measured in difference form and sorted, so the returned
distances carry no cancellation error and ties resolve by distance then index order of the sort.
"""

from __future__ import annotations

import os

from typing import Tuple

import torch

from . import _lib

SCAN_INIT_ROWS = 4096  # rows of the table the k-best lists are initialised from (dense path)


class NN_Wrapper:
    def __init__(self, train: torch.Tensor, nn_count: int, nn_method: str = "exact", chunk: int = 4096,
                 use_scan: bool = True, scan_kind: str = "auto", shuffle: bool = True, **kwargs):
        if nn_method.lower() != "exact":
            raise NotImplementedError(f"Nearest Neighbor algorithm {nn_method} is not implemented.")
        if not (isinstance(train, torch.Tensor) and train.is_cuda):
            raise TypeError("NN_Wrapper takes a torch tensor on the ROCm device")
        # the table is kept centred on its mean (queries are shifted alike): distances are unchanged,
        # and every Gram-form quantity below (|q|^2 + |x|^2 - 2 q.x, the split-bf16 margin) is computed
        # at the scale of the data's spread instead of its offset from the origin
        table = train[:, None] if train.ndim == 1 else train
        self._mean = table.double().mean(0).to(table.dtype)
        self.train_count, self.feature_count = table.shape
        # Tables the scan kernels serve are also stored in a fixed pseudo-random row order: a table
        # sorted in space (a grid, a time series) would hand a query all of its neighbours within one or
        # two consecutive tiles and overflow the kernels' candidate queues; shuffled, candidates arrive
        # evenly and the first rows are a fair sample for the initial k-best lists.  ``_perm`` maps
        # stored position -> caller's row, ``_inv`` the other way; both are None for the identity.
        self._perm = self._inv = None
        if (shuffle and use_scan and table.dtype == torch.float32 and self.feature_count % 4 == 0 and 4 <= self.feature_count <= 64
                and self.train_count > 2 * SCAN_INIT_ROWS):
            gen = torch.Generator(device=table.device).manual_seed(0x5CA9)
            self._perm = torch.randperm(self.train_count, device=table.device, generator=gen)
            self._inv = torch.empty_like(self._perm)
            self._inv[self._perm] = torch.arange(self.train_count, device=table.device)
            self.train = (table[self._perm] - self._mean).contiguous()
        else:
            self.train = (table - self._mean).contiguous()
        self.nn_count = int(nn_count)
        self.nn_method = "exact"
        self.chunk = int(chunk)
        self.use_scan = bool(use_scan)
        if scan_kind not in ("auto", "bf16x3", "f32"):
            raise ValueError("scan_kind must be 'auto', 'bf16x3' or 'f32'")
        # measured (1 M x 1 M, end to end): split-bf16 pre-filter 0.47 s (d = 40) / 0.36 s (d = 8),
        # plain fp32 scan 0.90 s / 0.43 s
        self.scan_kind = "bf16x3" if scan_kind == "auto" else scan_kind
        self._packed_train, self._packed_qmax, self._packed_layout = None, None, None
        self.last_overflow = None  # per-query overflow flags of the most recent scan (device int32)
        self._sq = (self.train.double() ** 2).sum(1).to(self.train.dtype)
        # the scan kernel reads |x|^2 in whole 64-row tiles: +inf past the end (never a neighbour)
        pad = (-self.train_count) % 64
        self._sq_scan = torch.cat([self._sq, torch.full((pad,), float("inf"), device=self._sq.device, dtype=self._sq.dtype)])

    def get_nns(self, test: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """neighbors.py:129-167: nn_count nearest training rows of every test row."""
        test = test[:, None] if test.ndim == 1 else test
        return self._get_nns(test, self.nn_count)

    def get_batch_nns(self, batch_indices: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """neighbors.py:169-211: neighbours of training rows, the row itself excluded.  (The
        reference drops column 0 of a k+1 query and relies on it being the point itself; here the
        self-match is masked explicitly, so duplicate points cannot displace it.)"""
        pos = batch_indices if self._inv is None else self._inv[batch_indices]
        return self._get_nns(self.train[pos], self.nn_count, exclude=pos, centred=True)

    def _get_nns(self, samples, nn_count, exclude=None, centred=False):
        if not centred:
            samples = samples.to(self.train.dtype) - self._mean
        out = self._scan_nns(samples, nn_count, exclude)
        idx, dist = out if out is not None else self._dense_nns(samples, nn_count, exclude)
        return (idx if self._perm is None else self._perm[idx]), dist

    def _scan_supported(self, samples, nn_count) -> bool:
        d = self.feature_count
        return (
            self.use_scan and self.train.dtype == torch.float32 and samples.dtype == torch.float32
            and d % 4 == 0 and 4 <= d <= 64 and 1 <= nn_count <= 64
            and self.train_count > 2 * SCAN_INIT_ROWS and self.train_count < 2**31
        )

    @staticmethod
    def _split_bf16(v: torch.Tensor):
        hi = v.to(torch.bfloat16)
        return hi, (v - hi.float()).to(torch.bfloat16)

    @classmethod
    def _pack_bf16(cls, x: torch.Tensor, slot_a, slot_b) -> torch.Tensor:
        """Rows [hi(KP) | lo(KP)] of the split x = hi + lo (bf16 each), KP = 16 ceil((d + 2) / 16); the
        last two slots of each part hold the split of ``slot_a`` / ``slot_b`` (scalars or (n,) tensors):
        the threshold terms the scan kernel evaluates on the matrix cores (csrc/mgp_knn.hip)."""
        n, d = x.shape
        kp = (d + 2 + 15) // 16 * 16
        packed = torch.zeros((n, 2 * kp), device=x.device, dtype=torch.bfloat16)
        packed[:, :d], packed[:, kp:kp + d] = cls._split_bf16(x)
        for pos, val in ((kp - 2, slot_a), (kp - 1, slot_b)):
            val = torch.as_tensor(val, device=x.device, dtype=torch.float32).expand(n)
            packed[:, pos], packed[:, kp + pos] = cls._split_bf16(val)
        return packed

    @classmethod
    def _pack_bf16_d8(cls, x: torch.Tensor, slot_a, slot_b) -> torch.Tensor:
        """d <= 8 (mgp_knn_scan_bf16x2_d8): rows [hi(8) | lo(8) | T(8)].  The threshold terms are one product of the T
        parts: a table row (``slot_a`` = c, ``slot_b`` = 1) carries T = [hi(c), lo(c), 1, 1, 0 ...], a query row
        (``slot_a`` = 1, ``slot_b`` = 0) T = [1, 1, 0, 0, 0 ...] -- the kernel writes the split of -thr into slots 2, 3."""
        n, d = x.shape
        packed = torch.zeros((n, 24), device=x.device, dtype=torch.bfloat16)
        packed[:, :d], packed[:, 8:8 + d] = cls._split_bf16(x)
        if torch.is_tensor(slot_a):  # table rows
            packed[:, 16], packed[:, 17] = cls._split_bf16(slot_a.to(torch.float32))
            packed[:, 18:20] = 1.0
        else:  # query rows
            packed[:, 16:18] = 1.0
        return packed

    def _scan_nns(self, samples, nn_count, exclude=None):
        """Fused MFMA scan (see the module docstring); None when the shape is not covered."""
        if not self._scan_supported(samples, nn_count):
            return None
        q = samples.contiguous()
        m, k = q.shape[0], int(nn_count)
        if m == 0:
            return None
        # exact k-best over the first rows of the table (Gram-form distances, like the scan's).  How many rows: a query
        # collects ~ k / rows-seen candidates per row and its queue (16 entries) is emptied once per tile at the
        # earliest, so the first tile -- 128 rows for the short packed rows, 64 otherwise -- should bring two at most
        # (P(Poisson(2) >= 17) ~ 5e-11): tile * k / 2 rows, in steps of 1 024, SCAN_INIT_ROWS at most.  torch's topk
        # over these rows was a fifth of the search's time at 1 M x 1 M, k = 30 (round 5)
        tile_rows = 128 if (self.scan_kind == "bf16x3" and self.feature_count + 2 <= 16) else 64
        init_rows = min(SCAN_INIT_ROWS, max(1024, -(-(tile_rows * k // 2) // 1024) * 1024))
        head = self.train[:init_rows]
        best_d = torch.empty((m, k), device=q.device, dtype=torch.float32)
        best_i = torch.empty((m, k), device=q.device, dtype=torch.int32)
        qn = (q * q).sum(1)
        init_chunk = max(self.chunk, 16384)  # (a 16 384 x 4 096 distance block is 268 MB; fewer, larger launches)
        for s in range(0, m, init_chunk):
            qq = q[s:s + init_chunk]
            d2 = self._sq[None, :init_rows] - 2.0 * (qq @ head.T) + qn[s:s + init_chunk, None]
            if exclude is not None:
                ex = exclude[s:s + init_chunk]
                inside = ex < init_rows
                rows = torch.arange(qq.shape[0], device=q.device)[inside]
                d2[rows, ex[inside]] = float("inf")
            # the k smallest per row, unordered (csrc/mgp_knn_select.hip: one wave per row, bisection on the key bits);
            # torch.topk's multi-block radix select on a million short rows was a sixth of the search (round 5)
            rc_sel = _lib.load().mgp_topk_rows_f32(_lib.ptr(d2), d2.shape[0], d2.shape[1], d2.stride(0), k,
                                                   _lib.ptr(best_d[s:s + init_chunk]), _lib.ptr(best_i[s:s + init_chunk]),
                                                   _lib.stream_ptr())
            if rc_sel == -2:  # (more than 4 096 columns: not this path's sizes)
                bd, bi = d2.topk(k, dim=1, largest=False)
                best_d[s:s + init_chunk] = bd
                best_i[s:s + init_chunk] = bi.to(torch.int32)
            else:
                _lib.check(rc_sel, "mgp_topk_rows_f32")
        overflow = torch.zeros((m,), device=q.device, dtype=torch.int32)
        ex64 = None if exclude is None else exclude.to(torch.int64).contiguous()
        if self.scan_kind == "bf16x3":
            # the split-bf16 kernel keeps exact (difference-form) distances in the lists
            for s in range(0, m, 65536):
                diff = q[s:s + 65536, None, :] - self.train[best_i[s:s + 65536].to(torch.int64)]
                best_d[s:s + 65536] = (diff * diff).sum(-1)
            # table rows carry c = -|x|^2/2 + 2^-14 QMAX |x| (raised by its own split error) and a 1;
            # query rows a 1 and a slot the kernel fills with -(|q|^2 - tau)/2
            # QMAX only has to bound |q| from above (a larger margin lets a few more near misses
            # through to the exact re-measurement): the packed table is rebuilt only when a batch
            # exceeds the bound it was packed for, with head-room so that it rarely does
            qmax = float(qn.max().sqrt())
            # d <= 8: the two-chain kernel on 24-slot rows (round 5), else three chains on [hi(KP) | lo(KP)]
            d8 = self.feature_count <= 8 and os.environ.get("MUYGPYS_HIP_KNN_D8", "1") != "0"  # (0: A/B against the three-chain kernel)
            pack = self._pack_bf16_d8 if d8 else self._pack_bf16
            scan = _lib.load().mgp_knn_scan_bf16x2_d8 if d8 else _lib.load().mgp_knn_scan_bf16x3
            # (the cache remembers WHICH layout it holds: the A/B switch may be flipped on a live object)
            if self._packed_train is None or self._packed_layout != d8 or qmax > self._packed_qmax:
                self._packed_layout = d8
                self._packed_qmax = 1.25 * qmax
                c = -0.5 * self._sq + (2.0**-14 * self._packed_qmax) * self._sq.sqrt()
                c = c + 2.0**-15 * c.abs()
                self._packed_train = pack(self.train, c, 1.0)
            packed_q = pack(q, 1.0, 0.0)
            rc = scan(
                _lib.ptr(self.train), _lib.ptr(self._packed_train), _lib.ptr(self._sq_scan), self.train_count,
                self.feature_count, _lib.ptr(q), _lib.ptr(packed_q), _lib.ptr(qn), _lib.ptr(ex64), m, k,
                init_rows, _lib.ptr(best_d), _lib.ptr(best_i), _lib.ptr(overflow), _lib.stream_ptr(),
            )
        else:
            rc = _lib.load().mgp_knn_scan_f32(
                _lib.ptr(self.train), _lib.ptr(self._sq_scan), self.train_count, self.feature_count,
                _lib.ptr(q), _lib.ptr(qn), _lib.ptr(ex64), m, k, init_rows,
                _lib.ptr(best_d), _lib.ptr(best_i), _lib.ptr(overflow), _lib.stream_ptr(),
            )
        if rc == -2:  # MGP_EUNSUPPORTED (alignment)
            return None
        _lib.check(rc, "mgp_knn_scan_f32")
        # the winners re-measured in difference form, put in order and mapped to the caller's row numbers in one launch
        # (csrc/mgp_knn_select.hip) -- was gather (m, k, d) / subtract / square / sum / argsort / gather / gather
        idx = torch.empty((m, k), dtype=torch.int64, device=q.device)
        dist = torch.empty((m, k), dtype=q.dtype, device=q.device)
        rc_fin = _lib.load().mgp_knn_finish_f32(_lib.ptr(q), _lib.ptr(self.train), self.feature_count, _lib.ptr(best_i), m, k,
                                                None, _lib.ptr(idx), _lib.ptr(dist), _lib.stream_ptr())
        if rc_fin == -2:
            cand = best_i.to(torch.int64)
            for s in range(0, m, 65536):
                c = cand[s:s + 65536]
                diff = q[s:s + 65536, None, :] - self.train[c]
                dd = (diff * diff).sum(-1)
                order = dd.argsort(dim=1, stable=True)
                idx[s:s + 65536] = c.gather(1, order)
                dist[s:s + 65536] = dd.gather(1, order)
        else:
            _lib.check(rc_fin, "mgp_knn_finish_f32")
        self.last_overflow = overflow
        redo = overflow.nonzero().reshape(-1)
        if redo.numel():  # queues overflowed (adversarial row order): those queries go dense
            ri, rd = self._dense_nns(q[redo], k, None if exclude is None else exclude[redo])
            idx[redo], dist[redo] = ri, rd
        return idx, dist

    DENSE_BUDGET_BYTES = 4 << 30  # (chunk x train_count) distance matrix + topk temporaries

    def _dense_nns(self, samples, nn_count, exclude=None):
        n = samples.shape[0]
        idx = torch.empty((n, nn_count), dtype=torch.int64, device=samples.device)
        dist = torch.empty((n, nn_count), dtype=samples.dtype, device=samples.device)
        # query rows per pass from a memory budget: the distance matrix of a pass is chunk x train_count
        per_row = 3 * self.train_count * self.train.element_size()
        chunk = max(1, min(self.chunk, self.DENSE_BUDGET_BYTES // max(per_row, 1)))
        for s in range(0, n, chunk):
            q = samples[s:s + chunk].to(self.train.dtype)
            d2 = self._sq[None, :] - 2.0 * (q @ self.train.T) + (q * q).sum(1)[:, None]
            if exclude is not None:
                rows = torch.arange(q.shape[0], device=q.device)
                d2[rows, exclude[s:s + chunk]] = float("inf")
            _, cand = d2.topk(nn_count, dim=1, largest=False)
            # exact squared distances of the winners, difference form, then final order
            diff = q[:, None, :] - self.train[cand]
            dd = (diff * diff).sum(-1)
            order = dd.argsort(dim=1, stable=True)
            idx[s:s + chunk] = cand.gather(1, order)
            dist[s:s + chunk] = dd.gather(1, order)
        return idx, dist
