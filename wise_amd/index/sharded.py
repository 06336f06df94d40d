"""Row-sharded flat IP index: one process per GPU, each rank owns a contiguous slice of the rows.

Per query batch: local HIP scan+top-k on every rank -> ONE all-gather of the per-shard
(score, id)[nq,k] lists packed in one int64 buffer (RCCL over xGMI; 16*nq*k bytes per rank, latency-bound) -> the
same k-way merge kernel on every rank, so every rank returns the global result.
The reference has no distributed path; this is the only collective the search needs.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from .. import _lib
from .flat_ip import FlatIPIndex


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous rows [lo, hi) of rank `rank`; sizes differ by at most one row."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


def merge_device(Ds: torch.Tensor, Is: torch.Tensor, k: int):
    """[parts,nq,k] device tensors -> [nq,k] via wise_topk_merge."""
    lib = _lib.lib()
    parts, nq, kk = Ds.shape
    D = torch.empty(nq, k, dtype=torch.float32, device=Ds.device)
    I = torch.empty(nq, k, dtype=torch.int64, device=Ds.device)
    _lib.check(lib.wise_topk_merge(Ds.contiguous().data_ptr(), Is.contiguous().data_ptr(), parts, nq, kk,
                                   D.data_ptr(), I.data_ptr(), _lib.stream_ptr()), "wise_topk_merge")
    return D, I


class ShardedFlatIPIndex:
    """Every rank constructs it around its own local FlatIPIndex (rows [lo,hi) of the global index,
    ids already global).  `search_device` is collective: all ranks call it with the same queries."""

    def __init__(self, local: FlatIPIndex, group: Optional[dist.ProcessGroup] = None,
                 local_search: Optional[Callable] = None, merge: Optional[Callable] = None,
                 always_exchange: bool = False):
        """always_exchange: run the all-gather and the merge even when the group has ONE rank (the collective and the
        merge kernel can then be exercised on a one-GPU box: tests/test_gpu_sharded.py)."""
        self.local = local
        self.group = group
        self.d = local.d
        self.always_exchange = bool(always_exchange)
        # injection points exist for the CPU (gloo) tests only; the product path is the HIP one
        self._local_search = local_search or local.search_device
        self._merge = merge or merge_device
        self._xchg = None
        self.last_exchange_bytes = 0    # payload this rank contributed to the last all-gather

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def ntotal(self) -> int:
        n = torch.tensor([self.local.ntotal], dtype=torch.int64,
                         device=self.local.device if dist.is_initialized() and dist.get_backend(self.group) == "nccl"
                         else "cpu")
        if dist.is_initialized() and (self.world > 1 or self.always_exchange):
            dist.all_reduce(n, group=self.group)
        return int(n.item())

    def search_device(self, q: torch.Tensor, k: int):
        D, I = self._local_search(q, k)
        if not dist.is_initialized() or (self.world == 1 and not self.always_exchange):
            return D, I
        W = self.world
        nq = D.shape[0]
        # ONE collective per query batch: scores and ids travel as one int64 buffer per rank — plane 0 the fp32 score
        # bits (as int32 values), plane 1 the ids — 16 * nq * k bytes (160 B at nq = 1, k = 10).  The exchange is
        # latency-bound (at 8 ranks the local scan of a 10M-row index is ~0.2 ms), so the number of collectives is what
        # counts, not their payload.  Buffers are kept between calls.
        key = (nq, k, D.device)
        if self._xchg is None or self._xchg[0] != key:
            self._xchg = (key, torch.empty(2, nq, k, dtype=torch.int64, device=D.device),
                          torch.empty(W, 2, nq, k, dtype=torch.int64, device=D.device))
        _, send, recv = self._xchg
        send[0].copy_(D.contiguous().view(torch.int32))        # exact: int32 -> int64 and back keeps the float's bits
        send[1].copy_(I)
        dist.all_gather_into_tensor(recv.view(W * 2 * nq, k), send.view(2 * nq, k), group=self.group)
        Ds = recv[:, 0].to(torch.int32).view(torch.float32)
        Is = recv[:, 1].contiguous()
        self.last_exchange_bytes = send.numel() * 8
        return self._merge(Ds, Is, k)

    def search(self, x, k: int):
        """faiss signature, collective: every rank calls it with the same x and gets the global result."""
        import numpy as np

        q = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self.local.device)
        D, I = self.search_device(q, int(k))
        return D.cpu().numpy(), I.cpu().numpy()

    def reconstruct_batch(self, ids):
        """IndexIDMap::reconstruct_batch over the shards (api/routes.py:1078), collective: every rank looks the ids up in
        its own rows (a row of NaN where it does not hold the id — wise_reconstruct_batch), ONE all-gather of the [n,d]
        answers, and each row is taken from the rank that had it.  Ids no rank holds stay NaN."""
        import numpy as np

        mine = torch.from_numpy(np.ascontiguousarray(self.local.reconstruct_batch(ids), dtype=np.float32))
        if not dist.is_initialized() or (self.world == 1 and not self.always_exchange):
            return mine.numpy()
        W = self.world
        on_device = dist.get_backend(self.group) == "nccl"
        mine = mine.to(self.local.device) if on_device else mine
        allr = torch.empty((W,) + tuple(mine.shape), dtype=torch.float32, device=mine.device)
        dist.all_gather_into_tensor(allr.view(W * mine.shape[0], -1), mine, group=self.group)
        have = ~torch.isnan(allr[:, :, 0])                        # [W, n]
        owner = have.to(torch.int8).argmax(dim=0)                 # first rank that holds the id (ids are unique)
        out = allr[owner, torch.arange(mine.shape[0], device=mine.device)]
        return out.cpu().numpy()
