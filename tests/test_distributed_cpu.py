"""CPU, world_size 2 over gloo: the N>1 path (round-robin shard + single gather to rank 0)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dss_amd  # noqa: F401
from dss_amd import distributed

K, N, TOTAL = 3, 11, 9


def _fake_result(i):
    g = torch.Generator().manual_seed(1000 + i)
    return torch.randn(K, generator=g), torch.randn(K, N, generator=g)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert distributed.rank_world() == (rank, world)
    mine = distributed.shard_indices(TOTAL, rank, world)
    vals = torch.stack([_fake_result(i)[0] for i in mine])
    vecs = torch.stack([_fake_result(i)[1] for i in mine])
    packed = distributed.pack_results(torch.tensor(mine), vals, vecs)
    out = distributed.gather_to_root(packed, TOTAL)
    if rank == 0:
        ids, v, e = distributed.unpack_results(out, K, N)
        ok = ids.tolist() == list(range(TOTAL))
        for i in range(TOTAL):
            rv, re = _fake_result(i)
            ok = ok and torch.equal(v[i], rv) and torch.equal(e[i], re)
        q.put(ok)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok
