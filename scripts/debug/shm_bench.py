"""Host-side plumbing microbench of the CLI's worker-process path: /dev/shm block creation + page-locking, JPEG decode,
H2D out of a block (registered or not) against torch's own pinned memory."""
import sys, time, os, mmap
sys.path.insert(0, os.getcwd())
import torch, dss_amd
from dss_amd import extract, pthfast, synthetic
from PIL import Image
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
dev = torch.device("cuda")
rt = torch.cuda.cudart()


def h2d(t, n=64, piece=691200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        x = t[(i % 16) * piece:(i % 16 + 1) * piece].to(dev, non_blocking=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3


size = 24 << 20
for flags in (None, 0, 1, 2, 3):
    path = f"/dev/shm/dss_bench_{flags}"
    fd = os.open(path, os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600); os.ftruncate(fd, size); m = mmap.mmap(fd, size); os.close(fd)
    t = torch.frombuffer(m, dtype=torch.uint8); t.zero_()
    rc = None
    if flags is not None:
        t0 = time.perf_counter(); rc = rt.cudaHostRegister(t.data_ptr(), size, flags); dt = time.perf_counter() - t0
    enq, tot = h2d(t)
    print(f"shm block, hipHostRegister flags={flags}: rc={rc} is_pinned={t.is_pinned()} H2D of 0.69 MB: enqueue {enq:.3f} ms, done {tot:.3f} ms")
    if flags is not None:
        rt.cudaHostUnregister(t.data_ptr())
    del t; os.unlink(path)
p = torch.empty(size, dtype=torch.uint8).pin_memory()
enq, tot = h2d(p)
print(f"torch pin_memory(): is_pinned={p.is_pinned()} H2D of 0.69 MB: enqueue {enq:.3f} ms, done {tot:.3f} ms")
s = torch.empty(size, dtype=torch.uint8).share_memory_(); s.zero_()
rc = rt.cudaHostRegister(s.data_ptr(), size, 0)
enq, tot = h2d(s)
print(f"torch share_memory_() + register: rc={rc} is_pinned={s.is_pinned()} H2D: enqueue {enq:.3f} ms, done {tot:.3f} ms")
d = torch.empty(128 * 900 * 384, dtype=torch.float32, device=dev)
for name, host in (("registered shm", s), ("pinned", p)):
    h = host[:16 << 20].view(torch.float32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): h.copy_(d[:h.numel()], non_blocking=True)
    torch.cuda.synchronize(); print(f"D2H 16 MiB into {name}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")

Image.fromarray(synthetic.synthetic_image(1, 480, 480)).save("/tmp/a.jpg", quality=95)
t0 = time.perf_counter()
for _ in range(50): a = pthfast.decode_rgb("/tmp/a.jpg")
print(f"decode_rgb: {(time.perf_counter() - t0) / 50 * 1e3:.2f} ms per 480x480 JPEG q95")
for cnt in (10, 18, 50):
    t0 = time.perf_counter(); blk = extract._ShmBlocks(cnt, 24 << 20); [blk.add() for _ in range(cnt)]; dt = time.perf_counter() - t0
    print(f"_ShmBlocks({cnt} x 24 MiB): {dt * 1e3:.0f} ms, is_pinned={blk.tensors[0].is_pinned()}")
    if cnt != 50: blk.close()
r = pthfast.decode_chunk(blk.paths[3], blk.size, ["/tmp/a.jpg"] * 16)
torch.cuda.synchronize(); t0 = time.perf_counter()
for off, shape in r:
    x = blk.tensors[3][off:off + 691200].view(shape).to(dev, non_blocking=True)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"16 H2D out of a filled block: enqueue {(t1 - t0) * 1e3:.2f} ms, done {(time.perf_counter() - t0) * 1e3:.2f} ms")
blk.close()
