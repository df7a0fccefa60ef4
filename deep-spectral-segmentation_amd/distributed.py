"""Multi-GPU layer: one process per GPU, round-robin image sharding, ONE gather at the end.

The reference hot path is single-process (its only parallelism is a CPU ``multiprocessing.Pool``,
extract/extract_utils.py:142).  Images are independent in both stages, so the path
shards embarrassingly:

  * rank ``r`` owns items ``i % world == r`` of the SORTED, de-duplicated work list - the order
    extract_utils.py:23 / extract.py:279 define;
  * there is no collective on the data path;
  * results are collected on rank 0 once, at the end (RCCL over xGMI when the backend is ``nccl``): sizes first
    (an int64 ``[n_r, 3]`` table of item id, N, K per result), then ONE flat f32 payload per rank (eigenvectors then
    eigenvalues of every result, back to back) received point to point - xGMI is point-to-point, so the gather is 7
    concurrent receives, one per link, of exactly the bytes each rank produced; with <= 180 MB for 10k images (C4) or
    <= 5 GB (C5, mixed N) it is bounded by per-link bandwidth (~153 GB/s), not by a ring.  Item ids travel as
    int64, never through a float.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def rank_world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def local_device() -> torch.device:
    if torch.cuda.is_available():
        idx = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(idx)
        return torch.device("cuda", idx)
    raise RuntimeError("no GPU visible: the extract hot path has no CPU fallback")


def init_process_group(backend: Optional[str] = None) -> Tuple[int, int]:
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("DSS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = local_device()
        dist.init_process_group(backend=backend, **kw)
    return rank_world()


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin shard of ``range(n_items)`` (BASELINE.json configs[3])."""
    return list(range(rank, n_items, world))


# ------------------------------------------------------------------------------------------------------------------
# Variable-size records (BASELINE config 5: mixed image sizes -> a different N per result): sizes first, then one
# flat payload per rank.
def pack_records(ids, eigenvalues, eigenvectors) -> Tuple[torch.Tensor, torch.Tensor]:
    """One or several same-shape groups of results -> ``(meta int64 [n, 3] = (id, N, K), payload f32 flat)``.
    ``ids [n]``, ``eigenvalues [n, K]``, ``eigenvectors [n, K, N]``; pass lists of such tensors for several groups
    (each with its own N / K).  Per result the payload holds the ``K*N`` eigenvector values, then the ``K`` eigenvalues."""
    if torch.is_tensor(ids):
        ids, eigenvalues, eigenvectors = [ids], [eigenvalues], [eigenvectors]
    metas, flats = [], []
    for i, ev, vec in zip(ids, eigenvalues, eigenvectors):
        n, k, nn = vec.shape
        dev = vec.device
        metas.append(torch.stack((i.to(dev, torch.int64), torch.full((n,), nn, dtype=torch.int64, device=dev),
                                  torch.full((n,), k, dtype=torch.int64, device=dev)), dim=1))
        flats.append(torch.cat((vec.reshape(n, -1).float(), ev.float()), dim=1).reshape(-1))
    return torch.cat(metas), torch.cat(flats)


def unpack_records(meta: torch.Tensor, payload: torch.Tensor):
    """Inverse of ``pack_records``: list of ``(id, eigenvalues [K], eigenvectors [K, N])`` in ``meta`` order."""
    out, off = [], 0
    for item, n, k in meta.tolist():
        vec = payload[off:off + k * n].reshape(k, n)
        val = payload[off + k * n:off + k * n + k]
        out.append((item, val, vec))
        off += k * n + k
    assert off == payload.numel(), (off, payload.numel())
    return out


def gather_records_to_root(meta: torch.Tensor, payload: torch.Tensor):
    """The run's single collection step: every rank's ``(meta, payload)`` -> rank 0, which returns them concatenated
    and ordered by item id (``None`` on the other ranks).  Two rounds: the ``[world, 2]`` table of (records, floats)
    per rank, then point-to-point receives of exactly those sizes (all posted at once: 7 concurrent xGMI links)."""
    rank, world = rank_world()
    if dist.is_available() and dist.is_initialized():   # also with ONE rank: the sizes round then still goes through the backend
        dev = payload.device if dist.get_backend() != "gloo" else torch.device("cpu")  # gloo moves host tensors
        meta, payload = meta.to(dev).contiguous(), payload.to(dev).contiguous()
        mine = torch.tensor([meta.shape[0], payload.numel()], dtype=torch.int64, device=dev)
        sizes = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, sizes, dst=0)                                    # round 1: sizes
        if rank != 0:
            if meta.shape[0] > 0:            # a rank with nothing to send posts nothing (rank 0 posts no receive for it)
                ops = [dist.P2POp(dist.isend, meta.reshape(-1), 0), dist.P2POp(dist.isend, payload, 0)]
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            return None
        metas, flats, ops = [meta], [payload], []
        for r in range(1, world):
            n_r, f_r = (int(v) for v in sizes[r].tolist())
            if n_r == 0:                     # fewer items than ranks: no zero-byte point-to-point operations
                continue
            metas.append(torch.empty((n_r, 3), dtype=torch.int64, device=dev))
            flats.append(torch.empty((f_r,), dtype=torch.float32, device=dev))
            ops += [dist.P2POp(dist.irecv, metas[-1].view(-1), r), dist.P2POp(dist.irecv, flats[-1], r)]
        if ops:                                                            # round 2: the flat payloads
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        meta, payload = torch.cat(metas), torch.cat(flats)
    if meta.shape[0] == 0:                   # nothing anywhere (an empty shard list): nothing to order
        return meta, payload
    # order by item id: per-record offsets into the concatenated payload
    lens = meta[:, 1] * meta[:, 2] + meta[:, 2]
    offs = torch.cumsum(lens, 0) - lens
    order = torch.argsort(meta[:, 0])
    if bool((lens == lens[0]).all()):      # one N: a plain row permutation
        payload = payload.reshape(meta.shape[0], -1)[order].reshape(-1)
    else:
        payload = torch.cat([payload[int(o):int(o) + int(l)] for o, l in zip(offs[order].tolist(), lens[order].tolist())])
    return meta[order], payload
