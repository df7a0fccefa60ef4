"""Multi-GPU layer: one process per GPU, round-robin image sharding, ONE gather at the end.

The reference hot path is single-process (its only parallelism is a CPU ``multiprocessing.Pool``,
extract/extract_utils.py:142).  Images are independent in both stages, so the path
shards embarrassingly:

  * rank ``r`` owns items ``i % world == r`` of the SORTED, de-duplicated work list - the order
    extract_utils.py:23 / extract.py:279 define;
  * there is no collective on the data path;
  * results are collected on rank 0 with one gather (RCCL over xGMI when the backend is ``nccl``): each
    rank contributes one packed f32 buffer ``[n_r, K*N + K + 1]`` (eigenvectors, eigenvalues, item id).
    xGMI is point-to-point, so a gather-to-root is 7 concurrent receives, one per link; with <= 180 MB
    for 10k images it is bounded by per-link bandwidth (~153 GB/s), not by a ring.
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


def pack_results(ids: torch.Tensor, eigenvalues: torch.Tensor, eigenvectors: torch.Tensor) -> torch.Tensor:
    """``ids [n]``, ``eigenvalues [n, K]``, ``eigenvectors [n, K, N]`` -> f32 ``[n, K*N + K + 1]``."""
    n = ids.shape[0]
    return torch.cat((eigenvectors.reshape(n, -1).float(), eigenvalues.float(),
                      ids.to(eigenvalues.device).float().view(n, 1)), dim=1).contiguous()


def unpack_results(buf: torch.Tensor, K: int, N: int):
    n = buf.shape[0]
    vec = buf[:, :K * N].reshape(n, K, N)
    val = buf[:, K * N:K * N + K]
    ids = buf[:, -1].round().long()
    return ids, val, vec


def gather_to_root(packed: torch.Tensor, n_total: int):
    """Gather every rank's packed rows on rank 0 and return them ordered by item id (rank 0) or ``None``.
    Shards differ by at most one row; they are padded to the common maximum so one ``gather`` suffices."""
    rank, world = rank_world()
    if world == 1 or not dist.is_initialized():
        order = torch.argsort(packed[:, -1])
        return packed[order]
    width = packed.shape[1]
    n_max = (n_total + world - 1) // world
    dev = packed.device
    if dist.get_backend() == "gloo":
        dev = torch.device("cpu")  # gloo gathers host tensors (CPU tests, single-GPU debugging)
    padded = torch.full((n_max, width), -1.0, dtype=torch.float32, device=dev)
    padded[: packed.shape[0]] = packed.to(dev)
    out = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, out, dst=0)
    if rank != 0:
        return None
    allrows = torch.cat(out)
    allrows = allrows[allrows[:, -1] >= 0]
    return allrows[torch.argsort(allrows[:, -1])]
