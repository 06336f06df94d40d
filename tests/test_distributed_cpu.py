"""CPU test (-m "not gpu"): the N>1 search path (row sharding, all-gather of per-shard top-k, merge)
with world_size 2 over gloo.  Local search and merge are played by the oracle here (no GPU in this
container); on the GPU box the same class runs the HIP kernels over RCCL (bench.py --gpus N)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, d, k, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ip_topk_ref
    from wise_amd.index.sharded import ShardedFlatIPIndex, shard_range

    X = np.random.default_rng(5).standard_normal((N, d)).astype(np.float32)  # every rank regenerates, keeps its slice
    Q = np.random.default_rng(6).standard_normal((5, d)).astype(np.float32)
    lo, hi = shard_range(N, rank, world)

    class Local:  # stands where FlatIPIndex stands on a GPU
        def __init__(self, dim, n):
            self.d, self.ntotal, self.device = dim, n, torch.device("cpu")

        def search_device(self, q, kk):
            raise RuntimeError("not used: local_search is injected")

    def local_search(q, kk):
        D, I = ip_topk_ref.ip_topk(X[lo:hi], q.numpy(), kk, id_base=lo + 1)
        return torch.from_numpy(D), torch.from_numpy(I)

    def merge(Ds, Is, kk):
        D, I = ip_topk_ref.merge_topk(Ds.numpy(), Is.numpy(), kk)
        return torch.from_numpy(D), torch.from_numpy(I)

    idx = ShardedFlatIPIndex(Local(d, hi - lo), local_search=local_search, merge=merge)
    assert idx.world == world
    assert idx.ntotal == N
    D, I = idx.search_device(torch.from_numpy(Q), k)
    np.savez(Path(out_dir) / f"rank{rank}.npz", D=D.numpy(), I=I.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_search_world2(tmp_path):
    from oracle import ip_topk_ref

    N, d, k, world = 1001, 64, 10, 2
    mp.spawn(_worker, args=(world, _free_port(), N, d, k, str(tmp_path)), nprocs=world, join=True)
    X = np.random.default_rng(5).standard_normal((N, d)).astype(np.float32)
    Q = np.random.default_rng(6).standard_normal((5, d)).astype(np.float32)
    Dr, Ir = ip_topk_ref.ip_topk(X, Q, k, id_base=1)
    for r in range(world):
        g = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(g["I"], Ir), f"rank {r}: sharded ids differ from the unsharded search"
        assert np.array_equal(g["D"], Dr)
