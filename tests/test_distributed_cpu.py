"""CPU, world_size 2 over gloo: the N>1 path - round-robin shards (uneven) and the sizes-first / flat-payload gather,
for results of ONE N (BASELINE config 4) and of DIFFERENT N (config 5), item ids as int64."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dss_amd  # noqa: F401
from dss_amd import distributed

K, N, TOTAL = 3, 11, 9          # 9 items over 2 ranks: shards of 5 and 4
BIG_ID = (1 << 40) + 7          # an id no float32 can carry


def _fake_result(i, n=N, k=K):
    g = torch.Generator().manual_seed(1000 + i)
    return torch.randn(k, generator=g), torch.randn(k, n, generator=g)


def _shape_of(i):               # mixed sizes: two different N (and K) interleaved over the items
    return (N, K) if i % 3 else (N + 6, K + 1)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert distributed.rank_world() == (rank, world)
    mine = distributed.shard_indices(TOTAL, rank, world)
    assert len(mine) == (5 if rank == 0 else 4)
    # ---- one N (config 4): the gathered payload is a plain row permutation ----------------------------------
    vals = torch.stack([_fake_result(i)[0] for i in mine])
    vecs = torch.stack([_fake_result(i)[1] for i in mine])
    out = distributed.gather_records_to_root(*distributed.pack_records(torch.tensor(mine), vals, vecs))
    ok = True
    if rank == 0:
        recs = distributed.unpack_records(*out)
        ok = [r[0] for r in recs] == list(range(TOTAL))
        for i, (_, v, e) in enumerate(recs):
            rv, re = _fake_result(i)
            ok = ok and torch.equal(v, rv) and torch.equal(e, re)
    else:
        assert out is None
    # ---- mixed N: sizes first, then one flat payload per rank --------------------------------------------
    groups = {}
    for i in mine:
        groups.setdefault(_shape_of(i), []).append(i)
    ids, evs, vcs = [], [], []
    for (n, k), items in groups.items():
        ids.append(torch.tensor([BIG_ID + i for i in items], dtype=torch.int64))
        evs.append(torch.stack([_fake_result(i, n, k)[0] for i in items]))
        vcs.append(torch.stack([_fake_result(i, n, k)[1] for i in items]))
    got = distributed.gather_records_to_root(*distributed.pack_records(ids, evs, vcs))
    if rank == 0:
        meta, payload = got
        assert meta.dtype == torch.int64
        recs = distributed.unpack_records(meta, payload)
        ok = ok and [r[0] for r in recs] == [BIG_ID + i for i in range(TOTAL)]
        for i, (item, val, vec) in enumerate(recs):
            n, k = _shape_of(i)
            rv, re = _fake_result(i, n, k)
            ok = ok and tuple(vec.shape) == (k, n) and torch.equal(val, rv) and torch.equal(vec, re)
        q.put(ok)
    else:
        assert got is None
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


# ------------------------------------------------------------------------------------------------------------------
# world 8 (BASELINE configs[3] / [4]: one node, eight ranks): uneven shards, mixed N / K, ranks with NOTHING to send
def _worker8(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = distributed.shard_indices(total, rank, world)
    assert mine == list(range(rank, total, world))
    groups = {}
    for i in mine:
        groups.setdefault(_shape_of(i), []).append(i)
    ids, evs, vcs = [], [], []
    for (n, k), items in groups.items():
        ids.append(torch.tensor([BIG_ID + i for i in items], dtype=torch.int64))
        evs.append(torch.stack([_fake_result(i, n, k)[0] for i in items]))
        vcs.append(torch.stack([_fake_result(i, n, k)[1] for i in items]))
    if mine:
        meta, payload = distributed.pack_records(ids, evs, vcs)
    else:           # fewer items than ranks: this rank holds no record at all
        meta, payload = torch.empty((0, 3), dtype=torch.int64), torch.empty((0,), dtype=torch.float32)
    got = distributed.gather_records_to_root(meta, payload)
    if rank == 0:
        recs = distributed.unpack_records(*got)
        ok = [r[0] for r in recs] == [BIG_ID + i for i in range(total)]
        for i, (item, val, vec) in enumerate(recs):
            n, k = _shape_of(i)
            rv, re = _fake_result(i, n, k)
            ok = ok and tuple(vec.shape) == (k, n) and torch.equal(val, rv) and torch.equal(vec, re)
        q.put(ok)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def _run_world8(total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, total, q)) for r in range(8)]
    for p in procs:
        p.start()
    ok = q.get(timeout=240)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ok


def test_shard_and_gather_world8_uneven_mixed_sizes():
    """61 items over 8 ranks (shards of 8 and 7), two result shapes interleaved, int64 ids: every item once, in order."""
    _run_world8(61)


def test_gather_world8_with_empty_ranks():
    """5 items over 8 ranks: three ranks have nothing to send - no zero-byte point-to-point operation is posted."""
    _run_world8(5)


def test_gather_single_process_and_empty():
    """No process group: the gather is the identity (ordered by id); an empty record set stays empty."""
    ids = torch.tensor([5, 2, 9], dtype=torch.int64)
    ev = torch.stack([_fake_result(int(i))[0] for i in ids])
    vc = torch.stack([_fake_result(int(i))[1] for i in ids])
    meta, payload = distributed.gather_records_to_root(*distributed.pack_records(ids, ev, vc))
    assert meta[:, 0].tolist() == [2, 5, 9]
    for (item, val, vec) in distributed.unpack_records(meta, payload):
        assert torch.equal(val, _fake_result(item)[0]) and torch.equal(vec, _fake_result(item)[1])
    meta, payload = distributed.gather_records_to_root(torch.empty((0, 3), dtype=torch.int64), torch.empty((0,)))
    assert meta.shape == (0, 3) and payload.numel() == 0
