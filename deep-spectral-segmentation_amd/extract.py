"""``extract_features`` / ``extract_eigs`` - MI355X-native drop-in for the two hot commands of the
reference's ``extract/extract.py`` (:21-116 and :119-280).  Same command names, flag names, defaults,
skip-if-exists behaviour and ``.pth`` schemas, so the reference's downstream
scripts (object-localization/main.py:254-272, extract.py:283-426) consume the outputs unchanged.

    python deep-spectral-segmentation_amd/extract.py extract_features \
        --images_list ./data/VOC2012/lists/images.txt --images_root ./data/VOC2012/images \
        --output_dir ./data/VOC2012/features/dino_vits16 --model_name dino_vits16 --batch_size 1
    python deep-spectral-segmentation_amd/extract.py extract_eigs \
        --images_root ./data/VOC2012/images --features_dir ./data/VOC2012/features/dino_vits16 \
        --which_matrix laplacian --output_dir ./data/VOC2012/eigs/laplacian --K 5

Under ``python -m torch.distributed.run --nproc-per-node N`` every rank takes the items
``i % world_size == rank`` of the sorted work list (one process per GPU, no collective on the data path).

Scope: ``which_matrix`` in {'laplacian', 'matting_laplacian'} (``lapnorm`` True or False), 'affinity' and
'affinity_svd' - the ``extract_eigs`` defaults, the README recipes and the reference's other solver branches, including
feature upsampling (``image_downsample_factor``) and the colour-affinity fusion (``image_color_lambda > 0`` with
``which_color_matrix`` 'knn' or 'rw': W_color is built on the GPU, extract_utils.knn_affinity / rw_affinity).  The dead
``affinity_torch`` branch raises ``NotImplementedError``.

The consumers of the eigen files keep the reference's names and file formats too:
``extract_single_region_segmentations`` (:383-426), ``extract_multi_region_segmentations`` (:283-377),
``extract_bboxes`` (:429-495), ``extract_bbox_features`` (:498-544).  (``extract_bbox_clusters`` /
``extract_semantic_segmentations`` / ``extract_crf_segmentations`` are outside SURVEY.md §8 and not provided.)
"""
from __future__ import annotations

import ast
import inspect
import os
import threading
import sys
from concurrent.futures import ThreadPoolExecutor
from functools import partial
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import torch

if __package__ in (None, ""):  # executed as a script: make the package importable as `dss_amd`
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import dss_amd  # noqa: F401
    from dss_amd import extract_utils as utils
    from dss_amd import pthfast
    from dss_amd import spectral
    from dss_amd.distributed import init_process_group, rank_world, local_device
    from dss_amd import extract as _pkg   # worker processes resolve their entry points in the PACKAGE module
else:
    _pkg = sys.modules[__name__]
    from . import extract_utils as utils
    from . import pthfast
    from . import spectral
    from .distributed import init_process_group, rank_world, local_device

_DTYPES = {"float16": torch.float16, "fp16": torch.float16, "f16": torch.float16, "half": torch.float16,
           "bfloat16": torch.bfloat16, "bf16": torch.bfloat16}


_IO_THREADS = max(8, min(32, (os.cpu_count() or 8) // 2))


class _StageClock:
    """Wall-clock totals per pipeline stage, printed at the end of a command when ``$DSS_CLI_TIMING`` is set."""

    def __init__(self):
        import time
        self.t, self.acc, self.on = time.perf_counter, {}, bool(os.environ.get("DSS_CLI_TIMING"))

    def __call__(self, name: str):
        clock = self

        class _ctx:
            def __enter__(self):
                self.t0 = clock.t()

            def __exit__(self, *exc):
                clock.acc[name] = clock.acc.get(name, 0.0) + clock.t() - self.t0
                return False

        return _ctx()

    def report(self, what: str):
        if self.on:
            print(f"[dss] {what} stage totals (s): " + ", ".join(f"{k} {v:.2f}" for k, v in self.acc.items()))


def _cap_host_threads():
    """The commands' host-side torch work is copies and slicing: on a 256-core box the default intra-op pool (one thread
    per core, spinning after every parallel region) only takes cycles from the decoder / loader / saver processes."""
    if torch.get_num_threads() > 8:
        torch.set_num_threads(8)


def _start_method() -> str:
    """How the I/O worker processes are started.  ``forkserver`` (round 5): ONE clean helper process (no HIP context, never
    imports torch) is spawned, imports what the workers need once (``_FORKSERVER_PRELOAD``) and forks every worker from
    there in milliseconds - eighty workers are up within the first half second of a run.  Rounds 3-4 used ``spawn``: every
    worker a fresh interpreter that imports numpy / PIL itself, twelve at a time (48 at once took 3.1 s to deliver a first
    chunk), so a 20 480-image run spent its first three seconds at a quarter of its decoders.  ``$DSS_IO_START_METHOD=fork``
    was measured on the GPU box: workers are up in milliseconds too, but the parent - which holds the HIP context - then
    runs its own copies 4-5x slower (extract_eigs 352 vs 633 images/s on 2 048 files), so it is not an option."""
    return os.environ.get("DSS_IO_START_METHOD", "forkserver")


_FORKSERVER_PRELOAD = ["dss_amd.pthfast", "numpy", "zipfile", "zlib", "pickle", "mmap", "PIL.Image", "PIL.ImageOps",
                       "PIL.JpegImagePlugin", "PIL.PngImagePlugin"]


class _spawn_without_main:
    """``spawn`` children normally re-import the parent's ``__main__`` - a caller's script without an
    ``if __name__ == "__main__"`` guard would then run again inside every I/O worker.  The workers here only need this
    package (their entry points are ``dss_amd.pthfast`` functions), so ``__main__`` is hidden from multiprocessing while
    they are started.  Re-entrant and thread-safe: the pools grow from helper threads (``_StaggeredPool._grow``) while the
    main thread may be starting another pool - the first one in hides ``__main__``, the last one out puts it back."""

    _lock = None
    _depth = 0
    _saved: dict = {}

    def __enter__(self):
        import threading

        cls = _spawn_without_main
        if cls._lock is None:
            cls._lock = threading.Lock()
        with cls._lock:
            if cls._depth == 0:
                main = sys.modules.get("__main__")
                cls._saved = {a: getattr(main, a) for a in ("__file__", "__spec__") if hasattr(main, a)}
                if "__file__" in cls._saved:
                    del main.__file__
                if main is not None:
                    main.__spec__ = None
            cls._depth += 1

    def __exit__(self, *exc):
        cls = _spawn_without_main
        with cls._lock:
            cls._depth -= 1
            if cls._depth == 0:
                main = sys.modules.get("__main__")
                for a, v in cls._saved.items():
                    setattr(main, a, v)
                cls._saved = {}


def _bounded_map(pool: ThreadPoolExecutor, fn, items, window: int):
    """``pool.map`` with at most ``window`` items in flight (decoded images are large: bound the memory)."""
    from collections import deque

    pending = deque()
    for it in items:
        pending.append(pool.submit(fn, it))
        if len(pending) >= window:
            yield pending.popleft().result()
    while pending:
        yield pending.popleft().result()


def _atomic_write(writer, obj, path: str):
    """``writer(obj, path)`` through a temporary name: a killed run (or an eigen file that is being REWRITTEN because a requested
    segmentation PNG was missing) never leaves a half-written file for the next run's skip-if-exists to trust."""
    tmp = f"{path}.tmp{os.getpid()}_{threading.get_ident() & 0xffff:x}"
    writer(obj, tmp)
    os.replace(tmp, path)


class _AsyncSaver:
    """``torch.save`` off the critical path for SMALL runs: a thread pool (measured: serial saves of 1.4 MB feature files
    cap the CLI at ~120 images/s; threads ~210 - pickling and the zip writer hold the GIL).  Large runs use ``_FastSaver``
    (torch-free worker processes).  ``close()`` waits for everything and re-raises the first error."""

    procs = 0   # (no worker processes: what `_FastSaver.procs` is compared with)

    def __init__(self, threads: int = _IO_THREADS, max_pending: int = 1024):
        self.pool = ThreadPoolExecutor(max_workers=threads)
        self.futures: List = []
        self.waited = 0
        self.max_pending = max_pending

    def submit(self, obj, path: str, writer=torch.save):
        self.futures.append(self.pool.submit(_atomic_write, writer, obj, path))
        if len(self.futures) - self.waited >= self.max_pending:  # back-pressure: bound the queued work
            upto = self.waited + self.max_pending // 2
            for f in self.futures[self.waited:upto]:
                f.result()
            self.waited = upto

    def submit_batch(self, kind: str, tensors: Tuple[torch.Tensor, ...], items: List[tuple], chunk: int = 16,
                     slot: Optional[int] = None):
        """``items`` (one tuple per file, see ``_SAVE_BUILDERS[kind]``) index into the batch ``tensors`` (host tensors): the files
        are built here and saved by the threads."""
        for it in items:
            self.submit(*_SAVE_BUILDERS[kind](tensors, it), writer=_SAVE_WRITERS.get(kind, torch.save))

    def close(self):
        for f in self.futures:
            f.result()
        self.futures.clear()
        self.waited = 0
        self.pool.shutdown()

    def abort(self):
        self.pool.shutdown(wait=False, cancel_futures=True)


class _FastSaver:
    """The savers of a LARGE run (round 5): torch-free worker processes (``pthfast.save_chunk / save_eigs / save_pngs`` - the
    archive `torch.save` would write, assembled directly) instead of processes that import torch to call it.  A saver is up
    in ~0.3 s instead of ~2 s (round 4: first feature file after 3.8 s) and costs a plain interpreter, so there can be
    dozens: a 1.4 MB feature file is ~2.7 ms of CRC + page-cache copy whoever writes it, and sixteen writers were what
    capped `extract_features` at ~5 800 images/s (`drain: shared block` = waiting for a ring slot to be written out).
    Features travel through a ring of page-locked /dev/shm blocks the GPU copies into (``block`` / ``submit_features`` /
    ``wait_slot``); eigenpairs and label maps are small and go through the pool's pipe (``submit_batch``)."""

    RING = 6

    def __init__(self, processes: int):
        self.pool = _StaggeredPool(processes)
        self.procs = processes                       # (truthy, like _AsyncSaver.procs: "worker processes are in use")
        self.slots = [None] * self.RING              # (path, mmap, u8 tensor, bytes) per ring slot
        self.pending: List[List] = [[] for _ in range(self.RING + 1)]   # results per slot; [RING] = the pipe jobs
        self.tag = f"{os.getpid()}_{id(self) & 0xffff:x}"

    def block(self, slot: int, nbytes: int) -> torch.Tensor:
        """The u8 tensor over ring block ``slot`` (grown if the batch needs more; ``wait_slot`` first)."""
        import mmap

        cur = self.slots[slot]
        if cur is None or cur[3] < nbytes:
            self._free(slot)
            path = f"/dev/shm/dss_{self.tag}_s{slot}_{nbytes}"
            fd = os.open(path, os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
            try:
                os.posix_fallocate(fd, 0, nbytes)
                m = mmap.mmap(fd, nbytes)
            finally:
                os.close(fd)
            t = torch.frombuffer(m, dtype=torch.uint8)
            try:
                torch.cuda.cudart().cudaHostRegister(t.data_ptr(), nbytes, 0)
            except Exception:  # pragma: no cover - pageable copies still work
                pass
            self.slots[slot] = cur = (path, m, t, nbytes)
        return cur[2]

    def _free(self, slot: int):
        cur, self.slots[slot] = self.slots[slot], None
        if cur is None:
            return
        try:
            torch.cuda.cudart().cudaHostUnregister(cur[2].data_ptr())
        except Exception:  # pragma: no cover
            pass
        try:
            os.unlink(cur[0])
        except OSError:
            pass

    def submit_features(self, slot: int, items: List[tuple], chunk: int = 8):
        path, _, _, nbytes = self.slots[slot]
        try:
            os.utime(path)                           # a live block must not look hours-old to another run's _sweep_stale
        except OSError:
            pass
        for s in range(0, len(items), chunk):
            self.pending[slot].append(self.pool.apply_async(pthfast.save_chunk, (path, nbytes, "features", items[s:s + chunk])))

    def submit_batch(self, kind: str, tensors: Tuple[torch.Tensor, ...], items: List[tuple], chunk: int = 32, slot=None):
        """``_AsyncSaver.submit_batch`` for the small kinds: "eigs" (tensors = (eigenvalues [B, K], eigenvectors [B, K, N]), item =
        (row, out path, problem)) and "png" (tensors = (u8 maps [B, N],), item = (row, out path, rows, cols))."""
        if kind == "eigs":
            ev, vec = (t.numpy() for t in tensors)
            jobs, fn = [(ev[j], vec[j], out) for j, out, _ in items], pthfast.save_eigs
        elif kind == "png":
            maps = tensors[0].numpy()
            jobs, fn = [(maps[j].reshape(hp, wp), out) for j, out, hp, wp in items], pthfast.save_pngs
        else:
            raise ValueError(kind)
        done = self.pending[self.RING]
        while len(done) > 256:                       # bound the results kept (and surface a worker's error early)
            done.pop(0).get(timeout=600)
        for s in range(0, len(jobs), chunk):
            done.append(self.pool.apply_async(fn, (jobs[s:s + chunk],)))

    def wait_slot(self, slot: int):
        for r in self.pending[slot]:
            r.get(timeout=600)                       # re-raises what the worker raised
        self.pending[slot].clear()

    def close(self):
        try:
            for slot in range(self.RING + 1):
                self.wait_slot(slot)
            self.pool.__exit__(None, None, None)
        except BaseException:
            self.pool.__exit__(RuntimeError, None, None)
            raise
        finally:
            for slot in range(self.RING):
                self._free(slot)

    def abort(self):
        """The run failed (worker / GPU / decode error, Ctrl-C): stop the workers and give the page-locked /dev/shm blocks back
        NOW - `_sweep_stale` would only collect them hours later (up to RING blocks of ~0.8 GB each)."""
        try:
            self.pool.__exit__(RuntimeError, None, None)
        finally:
            for slot in range(self.RING):
                self._free(slot)


def _build_feature_file(tensors, item):
    (k,), (j, idx, file, model_name, patch_size, shape, out) = tensors, item
    return _feature_dict(k[j:j + 1].clone(), idx, file, model_name, patch_size, shape), out


def _build_eig_file(tensors, item):
    # schema of extract/extract.py:235,243-244: eigenvalues [K] f32, eigenvectors [K, N] f32; the 'affinity' branch
    # stores its eigenvalues as a raw numpy array (:171,243) - kept, consumers load it that way
    (ev, vec), (j, out, problem) = tensors, item
    vals = ev[j].clone()
    return {"eigenvalues": vals.numpy() if problem == "affinity" else vals, "eigenvectors": vec[j].clone()}, out


def _build_png_file(tensors, item):
    (labels,), (j, out, hp, wp) = tensors, item       # u8 [B, N] label / mask maps -> one 8-bit PNG per image
    return labels[j].reshape(hp, wp).numpy(), out


def _write_png(arr, path: str):
    from PIL import Image

    Image.fromarray(arr).save(path, format="PNG")     # (explicit: the name may be a temporary one, see _atomic_write)


_SAVE_BUILDERS = {"features": _build_feature_file, "eigs": _build_eig_file, "png": _build_png_file}
_SAVE_WRITERS = {"png": _write_png}     # everything else: torch.save


class _ShmBlocks:
    """Up to ``count`` blocks of ``size`` bytes in /dev/shm that worker processes fill (``pthfast`` maps them by path)
    and the GPU reads: every block is wrapped as a u8 tensor and page-locked (best effort; measured on the GPU box: a
    0.69 MB H2D takes 0.024 ms out of a registered block, 0.145 ms out of an unregistered one).  Blocks are created one
    at a time (``add``) so that the first workers are busy while the later blocks are still being set up."""

    def __init__(self, count: int, size: int):
        self.count, self.size, self.paths, self.maps, self.tensors = count, size, [], [], []
        self._sweep_stale()

    STALE_AFTER_S = 6 * 3600

    @staticmethod
    def _sweep_stale():
        """Blocks of runs that were killed before their ``finally`` (names carry the owner's pid): remove those that are
        OURS (same uid), whose process is gone AND that nobody has touched for hours - /dev/shm may be shared with other
        PID namespaces (containers started with --ipc=host), where a live run's pid looks dead from here."""
        import time

        try:
            names = [n for n in os.listdir("/dev/shm") if n.startswith("dss_")]
        except OSError:
            return
        now, uid = time.time(), os.getuid()
        for n in names:
            path = os.path.join("/dev/shm", n)
            try:
                st = os.stat(path)
                if st.st_uid != uid or now - max(st.st_mtime, st.st_atime) < _ShmBlocks.STALE_AFTER_S:
                    continue
                os.kill(int(n.split("_")[1]), 0)          # raises if no such process
            except (ProcessLookupError, ValueError, IndexError):
                try:
                    os.unlink(path)
                except OSError:
                    pass
            except OSError:
                pass                      # vanished meanwhile, or somebody else's live process

    def add(self) -> int:
        import mmap

        i = len(self.paths)
        path = f"/dev/shm/dss_{os.getpid()}_{id(self) & 0xffff:x}_{i}"
        fd = os.open(path, os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
        try:
            os.posix_fallocate(fd, 0, self.size)   # the pages exist (zeroed by the kernel, no fault per page) before
            m = mmap.mmap(fd, self.size)           # they are page-locked
        finally:
            os.close(fd)
        t = torch.frombuffer(m, dtype=torch.uint8)
        try:
            torch.cuda.cudart().cudaHostRegister(t.data_ptr(), self.size, 0)
        except Exception:  # pragma: no cover - pageable copies still work
            pass
        self.paths.append(path), self.maps.append(m), self.tensors.append(t)
        return i

    def close(self):
        for t in self.tensors:
            try:
                torch.cuda.cudart().cudaHostUnregister(t.data_ptr())
            except Exception:  # pragma: no cover
                pass
        self.tensors.clear()
        for path in self.paths:
            try:
                os.unlink(path)
            except OSError:
                pass
        self.paths.clear()


def _pump_chunks(fn, chunks: List[list], extra_args: tuple, processes: int, block_bytes: int):
    """Runs ``fn(block path, block size, chunk, *extra_args)`` of ``pthfast`` for every chunk on ``processes`` torch-free
    worker processes, each call filling one page-locked /dev/shm block.  Yields ``(entries, block tensor u8, release)``
    in chunk order; the consumer enqueues its copies out of the block on the current stream and then calls
    ``release()`` - the block goes back to the workers once those copies have finished."""
    from collections import deque

    import time
    t_start = time.perf_counter()
    pool = _StaggeredPool(processes)                 # the first wave boots while the blocks are pinned
    blocks = _ShmBlocks(processes + 2, block_bytes)
    first = None
    try:
        with pool:
            free, events = deque(), [None] * blocks.count
            pending, nxt = deque(), 0

            def releaser(b):
                def release():
                    events[b] = torch.cuda.Event()
                    events[b].record()
                    free.append(b)
                return release

            while nxt < len(chunks) or pending:
                while (free or len(blocks.paths) < blocks.count) and nxt < len(chunks) and len(pending) < pool.capacity() + 2:
                    b = free.popleft() if free else blocks.add()
                    if events[b] is not None:
                        events[b].synchronize()     # the copies out of this block have finished
                        try:                        # (mmap writes do not refresh a tmpfs file's times: a LIVE run's blocks must
                            os.utime(blocks.paths[b])   # not look hours-old to another run's _sweep_stale)
                        except OSError:
                            pass
                    pending.append((pool.apply_async(fn, (blocks.paths[b], blocks.size, chunks[nxt]) + extra_args), b))
                    nxt += 1
                res, b = pending.popleft()
                got = res.get(timeout=600)          # a lost worker must not hang the run
                if first is None:
                    first = time.perf_counter() - t_start
                yield got, blocks.tensors[b], releaser(b)
    finally:
        blocks.close()
        if os.environ.get("DSS_CLI_TIMING") and first is not None:
            print(f"[dss] {fn.__name__}: {processes} workers ({_start_method()}, waves of {pool.WAVE}), first chunk after {first:.2f} s, "
                  f"all {len(chunks)} chunks after {time.perf_counter() - t_start:.2f} s")


class _StaggeredPool:
    """``processes`` torch-free worker processes started in WAVES: forty-eight interpreters booting at once took 3.1 s to
    deliver their first chunk on the GPU box (twelve: 1.0 s) - so twelve start, and while they already work the next
    twelve are started from a helper thread, and so on.  ``apply_async`` goes to the least loaded pool that is up;
    ``capacity()`` = workers up so far (the producer keeps that many chunks + 2 in flight)."""

    WAVE = 12          # spawn: interpreters booting at once; the fork server's children cost milliseconds: waves of 32

    def __init__(self, processes: int):
        import threading
        import torch.multiprocessing as mp

        self.ctx = mp.get_context(_start_method())
        if _start_method() == "forkserver":
            self.ctx.set_forkserver_preload(_FORKSERVER_PRELOAD)   # (no '__main__': the helper never runs the caller's script)
        self.pools, self.load, self.sizes = [], [], []
        self.lock = threading.Lock()
        self.closing = False
        if _start_method() != "spawn":
            self.WAVE = 32
        self._start(min(self.WAVE, processes))
        self.rest = processes - self.sizes[0]
        self.thread = threading.Thread(target=self._grow, daemon=True)
        self.thread.start()

    def _start(self, n: int):
        with self.lock:
            if self.closing:                 # (a wave that would start behind __exit__'s back would never be closed)
                return
        with _spawn_without_main():
            pool = self.ctx.Pool(n)
        with self.lock:
            if self.closing and self.pools:  # __exit__ has begun meanwhile: this wave is not wanted any more
                pool.terminate()
                pool.join()
                return
            self.pools.append(pool)
            self.load.append(0)
            self.sizes.append(n)

    def _grow(self):
        import time
        while self.rest > 0 and not self.closing:
            try:   # the previous wave has finished booting when each of its workers has answered once
                self.pools[-1].map(pthfast.noop, range(self.sizes[-1]), chunksize=1)
            except Exception:
                return
            if self.closing:
                return
            n = min(self.WAVE, self.rest)
            self._start(n)
            self.rest -= n
            time.sleep(0.01)

    def capacity(self) -> int:
        with self.lock:
            return sum(self.sizes)

    def apply_async(self, fn, args):
        with self.lock:
            i = min(range(len(self.pools)), key=lambda j: self.load[j] / self.sizes[j])
            self.load[i] += 1
            pool = self.pools[i]

        def done(_res, i=i):
            with self.lock:
                self.load[i] -= 1

        return pool.apply_async(fn, args, callback=done, error_callback=done)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        with self.lock:
            self.closing = True
        self.thread.join(timeout=60)         # (its one blocking step is a wave of workers answering their first call)
        with self.lock:
            pools = list(self.pools)
        for p in pools:
            p.terminate() if exc[0] is not None else p.close()
        for p in pools:
            p.join()
        return False


def _iter_features(files, which_features: str, processes: int, window: int, device: Optional[torch.device] = None):
    """Yields ``(data_dict without the features, features [N, D] f32)`` for every file of ``files``, in order.  Small
    runs: ``torch.load`` on a thread pool, host tensors.  Large runs on a GPU: ``processes`` torch-free loader processes
    (``pthfast.load_chunk``) read chunks of 16 files straight into page-locked /dev/shm blocks; the rows are copied to
    ``device`` from there (one async H2D per run of same-shape files) and yielded as DEVICE tensors."""
    if processes <= 0 or device is None or device.type != "cuda":
        with ThreadPoolExecutor(max_workers=_IO_THREADS) as pool:
            yield from _bounded_map(pool, lambda f: _load_features(str(f), which_features), files, window)
        return
    per = 16
    chunks = [[str(f) for f in files[s:s + per]] for s in range(0, len(files), per)]
    # a file's size bounds its feature bytes AS STORED; the loaders hand them over as f32: f16 / bf16 features (half the
    # file) need twice the file's size in the block - sized for that, or half of such files would fall back to torch.load
    biggest = 2 * max(os.path.getsize(f) for f in files)
    for entries, block, release in _pump_chunks(pthfast.load_chunk, chunks, (which_features,), processes, per * biggest):
        i = 0
        while i < len(entries):
            meta, off, shape = entries[i]
            if meta is None:                # (None, file, reason): not a plain feature archive
                yield _load_features(off, which_features)
                i += 1
                continue
            j, nbytes = i + 1, 4 * shape[0] * shape[1]
            while j < len(entries) and entries[j][0] is not None and entries[j][2] == shape \
                    and entries[j][1] == off + (j - i) * nbytes:
                j += 1
            rows = block[off:off + (j - i) * nbytes].view(torch.float32).view((j - i,) + tuple(shape))
            dev = rows.to(device, non_blocking=True)
            for n in range(j - i):
                yield entries[i + n][0], dev[n]
            i = j
        release()


def _iter_images(dataset, todo, processes: int, window: int, device: torch.device):
    """Yields ``(u8 RGB [H, W, 3] image, file name)`` for every ``(index, output)`` of ``todo``, in order.  Small runs:
    PIL on a thread pool, page-locked host tensors.  Large runs: ``processes`` torch-free decoder processes
    (``pthfast.decode_chunk``) fill page-locked /dev/shm blocks and the images are yielded as DEVICE tensors."""
    if processes <= 0 or device.type != "cuda":
        def decode(t):
            img, file, _ = dataset[t[0]]
            return img.pin_memory() if device.type == "cuda" else img, file   # the copy into page-locked memory: here

        with ThreadPoolExecutor(max_workers=_IO_THREADS) as pool:  # decode pool (the reference: 8 loader processes)
            yield from _bounded_map(pool, decode, todo, window)
        return
    per = 16
    names = [dataset.filenames[i] for i, _ in todo]
    paths = [str(n if dataset.root is None else dataset.root / n) for n in names]
    chunks = [paths[s:s + per] for s in range(0, len(paths), per)]
    at = 0
    # 1.5 MB per image on average (VOC: <= 500 x 500 x 3 = 0.75 MB); what does not fit a block is decoded here
    for entries, block, release in _pump_chunks(pthfast.decode_chunk, chunks, (), processes, per * (3 << 19)):
        i = 0
        while i < len(entries):
            off, shape = entries[i]
            if off is None:                # did not fit the block: decode it here
                yield dataset[todo[at][0]][0].to(device), names[at]
                at, i = at + 1, i + 1
                continue
            # ONE H2D copy per run of same-shape images that lie back to back in the block (a whole chunk on a one-size
            # dataset) instead of one per image: 20 480 copy calls were ~1 s of the main thread's 8 s at 5 000 images/s
            n, j = shape[0] * shape[1] * shape[2], i + 1
            while j < len(entries) and entries[j][1] == shape and entries[j][0] == off + (j - i) * n:
                j += 1
            dev = block[off:off + (j - i) * n].view((j - i,) + tuple(shape)).to(device, non_blocking=True)
            for q in range(j - i):
                yield dev[q], names[at]
                at += 1
            i = j
        release()


def _shm_free_bytes() -> int:
    try:
        st = os.statvfs("/dev/shm")
        return st.f_bavail * st.f_frsize
    except OSError:
        return 0


def _io_processes(n_items: int, most: int = 16, env: str = "") -> int:
    """Worker processes for the per-image file I/O of a run of ``n_items`` files on this rank: none for small runs,
    otherwise a share of the host cores, at most ``most`` (``$DSS_IO_PROCESSES`` overrides every pool, ``$<env>`` this one)."""
    if env and os.environ.get(env):
        return int(os.environ[env])
    if os.environ.get("DSS_IO_PROCESSES"):
        return int(os.environ["DSS_IO_PROCESSES"])
    few = 512 if _start_method() == "spawn" else 256   # a spawned interpreter costs ~1 s, the fork server ~0.3 s once
    if n_items < few or _shm_free_bytes() < (8 << 30):   # batches travel through /dev/shm: needs room
        return 0
    local = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    return max(2, min(most, (os.cpu_count() or 8) // (4 * max(1, local))))


def _barrier():
    """End-of-stage rendezvous of the ranks (the reference's ``accelerator.wait_for_everyone``, extract.py:114)."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()


def _make_output_dir_all_ranks(output_dir: str):
    """Rank 0 runs the reference's (possibly interactive) non-empty check BEFORE any rank writes; the others wait for
    its DECISION, not just for a barrier: if rank 0 declines the prompt (``sys.exit``) or fails, every rank leaves."""
    rank, _ = rank_world()
    err = None
    if rank == 0:
        try:
            utils.make_output_dir(output_dir)
        except BaseException as e:  # SystemExit from the prompt included
            err = e
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        verdict = [None if err is None else f"{type(err).__name__}: {err}"]
        torch.distributed.broadcast_object_list(verdict, src=0)
        if verdict[0] is not None and rank != 0:
            raise SystemExit(f"[dss] rank 0 did not open {output_dir} ({verdict[0]}): rank {rank} stops too")
    if err is not None:
        raise err
    if rank != 0:
        utils.make_output_dir(output_dir, check_if_empty=False)


def _feature_dict(k: torch.Tensor, index: int, file: str, model_name: str, patch_size: int,
                  shape: Tuple[int, int, int, int]) -> dict:
    """The reference's feature-file schema (extract/extract.py:98,104-110)."""
    return {"k": k.detach().cpu(), "indices": torch.tensor(index), "file": file, "id": Path(file).stem,
            "model_name": model_name, "patch_size": patch_size, "shape": tuple(int(s) for s in shape)}


def extract_features(images_list: str, images_root: Optional[str], model_name: str, batch_size: int,
                     output_dir: str, which_block: int = -1, weights: Optional[str] = None,
                     dtype: str = "float16", synthetic_weights: Optional[int] = None, gelu: str = "auto"):
    """Extract features from a list of images (see module docstring).  ``gelu``: "erf_f16" (erf-GELU as a polynomial form on packed
    f16, f16 operands only), "erf" (DINO's exact GELU in fp32 arithmetic, ~4 % slower at D = 384) or "auto" (default: "erf_f16" for
    the D = 384 models, "erf" for D = 768); the run prints which one and which operand dtype produced its files.  ``batch_size`` is the maximum
    number of SAME-SHAPE images pushed through the ViT together; one ``B=1`` file is written per image
    whatever its value (every consumer asserts ``B == 1``, extract_utils.py:76).  ``batch_size <= 0``: as many images as
    fill FOUR whole rounds of the Linear kernels' workgroups at the first image's size (581 at 480 x 480 / patch 16) - a
    128-image forward is 0.88 of one round and the GPU side of this command then tops out near 5 000 images/s, a third of
    what the same kernels do on ~600 images (profiles/r05_cli_throughput.txt)."""
    _make_output_dir_all_ranks(output_dir)
    _cap_host_threads()
    model_name = model_name.lower()
    if not ("dino" in model_name or "mocov3" in model_name):
        raise ValueError(model_name)
    rank, world = rank_world()
    device = local_device()

    filenames = Path(images_list).read_text().splitlines()
    dataset = utils.ImagesDataset(filenames=filenames, images_root=images_root, transform=None)
    print(f"Dataset size: {len(dataset)=}")
    todo = []
    for i in range(rank, len(dataset), world):
        out = Path(output_dir) / f"{Path(dataset.filenames[i]).stem}.pth"
        if out.is_file():
            print(f"Skipping existing file {str(out)}")
            continue
        todo.append((i, out))

    bs = int(batch_size)
    clock = _StageClock()
    with clock("start savers"):   # first of all: the saver processes boot while the model is being built
        n_savers = _io_processes(len(todo), most=32, env="DSS_SAVER_PROCESSES")
        saver = _FastSaver(n_savers) if n_savers > 0 else _AsyncSaver()
    # ~6.5 ms of PIL per 480 x 480 JPEG (150 images/s per process): the ViT takes 12 000 images/s, the feature savers
    # ~1 000 files/s each - dozens of decoders (started in waves of twelve, _StaggeredPool) before the GPU is what waits
    decoders = _io_processes(len(todo), most=48) if saver.procs else 0
    try:
        with clock("model"):
            model, _, patch_size, _ = utils.get_model(model_name, device=device, dtype=_DTYPES[str(dtype).lower()],
                                                      weights=weights, synthetic_seed=synthetic_weights, gelu=gelu)
    except BaseException:
        saver.abort()      # (the saver processes were started first: a bad checkpoint path must not leave them behind)
        raise

    # One batch travels: page-locked decoded images -> async H2D + ViT -> a drain thread that waits for the batch, copies
    # the features into a page-locked shared-memory block the saver processes map and hands the files over.  The main
    # thread is back at the decoded-image queue while the GPU and the drain thread work on the previous batches.
    import queue as _queue
    import threading
    ring = {"next": 0}
    copy_stream = torch.cuda.Stream(device=device)
    drain_q: "_queue.Queue" = _queue.Queue(maxsize=3)
    drain_err: List[BaseException] = []

    def drain():
        torch.cuda.set_device(device)      # the current device is per THREAD: without this a rank with LOCAL_RANK != 0 would
        while True:                        # register its blocks (and create a context) on GPU 0 from here
            job = drain_q.get()
            if job is None:
                return
            try:
                k_dev, done, metas = job
                job = None                 # the queue item must not keep the batch's HBM alive past `del k_dev`
                slot = None
                with clock("drain: shared block"):
                    if saver.procs:
                        # a ring of page-locked /dev/shm blocks the saver processes map by path: a fresh segment per batch
                        # cost 90 ms of page faults plus a 1.5 GB/s pageable D2H (177 MB per 128 images)
                        slot = ring["next"] % saver.RING
                        ring["next"] += 1
                        saver.wait_slot(slot)          # every file cut from the block's previous contents has been written
                        nbytes = k_dev.numel() * k_dev.element_size()
                        k = saver.block(slot, nbytes)[:nbytes].view(k_dev.dtype).view(k_dev.shape)
                    else:
                        k = torch.empty(k_dev.shape, dtype=k_dev.dtype)
                with clock("drain: wait + D2H"), torch.cuda.stream(copy_stream):   # not behind the next batch's ViT
                    copy_stream.wait_event(done)
                    k_dev.record_stream(copy_stream)
                    k.copy_(k_dev, non_blocking=True)
                    copy_stream.synchronize()
                del k_dev
                with clock("drain: hand to savers"):
                    if saver.procs:
                        per = k[0].numel() * k.element_size()
                        saver.submit_features(slot, [(j * per, tuple(k.shape[1:]), idx, file, mname, psize, shp, out)
                                                     for j, idx, file, mname, psize, shp, out in metas])
                    else:
                        saver.submit_batch("features", (k,), metas)
            except BaseException as e:  # re-raised by the main thread
                drain_err.append(e)

    drainer = threading.Thread(target=drain, daemon=True)
    drainer.start()

    def flush(batch: List[Tuple[int, Path, torch.Tensor, str]]):
        if not batch:
            return
        if drain_err:
            raise drain_err[0]
        with clock("flush: enqueue H2D + ViT"):
            # no host-side stacking pass: torch.stack of 128 decoded images into fresh pageable memory cost 265 ms
            # the images arrive page-locked (decode threads) or already on the device (decoder processes): one call
            # gathers the batch - every Python-level call gives the GIL away and queues to get it back
            imgs = torch.empty((len(batch),) + tuple(batch[0][2].shape), dtype=torch.uint8, device=device)
            torch._foreach_copy_(list(imgs.unbind(0)), [b[2] for b in batch], non_blocking=True)
            k_dev = model.extract_k(imgs, which_block=which_block)
            done = torch.cuda.Event()
            done.record()
        shp = (1, 3, int(imgs.shape[1]), int(imgs.shape[2]))
        with clock("flush: drain queue full"):
            drain_q.put((k_dev, done, [(j, idx, file, model_name, patch_size, shp, str(out))
                                       for j, (idx, out, _, file) in enumerate(batch)]))
        batch.clear()

    # Real datasets (VOC) mix image sizes: bucket by shape so every ViT launch is a full same-shape batch.  At most
    # `max_pending` decoded images wait in the buckets; beyond that the fullest bucket is flushed early.
    # The batch size is PER SHAPE BUCKET (a dataset's first image says nothing about the others): `batch_size <= 0` gives each
    # shape four rounds of the K-resident Linear kernels' workgroups, and every bucket is clamped to the kernels' 32-bit row
    # limits (`bucket_batch`: M * 4 D and M * T below 2^32, the formula bench.py applies).  What waits in the buckets is bounded
    # in BYTES of decoded images (`max_pending_bytes`), not in images.
    buckets: Dict[Tuple[int, ...], List[Tuple[int, Path, torch.Tensor, str]]] = {}
    auto_bs = bs <= 0
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    rows_wg = spectral.hip.LINEAR_KRES_WIDTHS.get(model.embed_dim, (None, 256))[1]
    bs_of: Dict[Tuple[int, ...], int] = {}

    def bucket_bs(shape) -> int:
        if shape not in bs_of:
            bs_of[shape] = bucket_batch(shape[0], shape[1], patch_size, model.embed_dim, bs if not auto_bs else 0, cus, rows_wg)
        return bs_of[shape]

    max_pending_bytes, pending_bytes = 6 << 30, 0
    decoded = _iter_images(dataset, todo, decoders, 4 * (max(1, bs) if not auto_bs else 512), device)
    try:
        for idx, out in todo:
            with clock("wait for decoded image"):
                img, file = next(decoded)
            shape = tuple(img.shape)
            bucket = buckets.setdefault(shape, [])
            bucket.append((idx, out, img, file))
            pending_bytes += img.numel()
            if len(bucket) >= bucket_bs(shape):
                pending_bytes -= sum(b[2].numel() for b in bucket)
                flush(bucket)
            elif pending_bytes >= max_pending_bytes:
                fullest = max(buckets.values(), key=lambda b: sum(x[2].numel() for x in b))
                pending_bytes -= sum(b[2].numel() for b in fullest)
                flush(fullest)
        for bucket in buckets.values():
            flush(bucket)
        with clock("tail: drain thread"):
            drain_q.put(None)
            drainer.join()
        if drain_err:
            raise drain_err[0]
        with clock("tail: savers finish"):
            saver.close()
    except BaseException:
        # decode / GPU / worker error or Ctrl-C: the drain thread is told to stop, the workers are stopped and the page-locked
        # /dev/shm ring is unlinked before the error leaves (the blocks used to stay until the 6 h stale sweep)
        try:
            drain_q.put_nowait(None)
        except Exception:
            pass
        saver.abort()
        raise
    clock.report("extract_features")
    _barrier()
    print(f"Saved features to {output_dir}")


def bucket_batch(h: int, w: int, patch_size: int, embed_dim: int, batch_size: int, compute_units: int = 256,
                 rows_per_workgroup: int = 256) -> int:
    """Images per ViT forward for a bucket of ``h x w`` images.  ``batch_size > 0`` is the caller's value; ``<= 0`` asks for four
    rounds of the K-resident Linear kernels' workgroups at THIS shape.  Either way clamped so that the forward's token matrix stays
    inside the kernels' 32-bit limits (``M * 4 D`` bytes of the fp32 residual stream and ``M * T`` of the hand-over kernel's
    row -> image division, both below 0.9 * 2^32: the same cap as bench.py's ``row_cap``)."""
    tokens = (h // patch_size) * (w // patch_size) + 1
    row_cap = int(0.9 * 2 ** 32 / max(4 * embed_dim, tokens))
    want = int(batch_size) if batch_size > 0 else max(8, int(4 * compute_units * rows_per_workgroup / tokens))
    return max(1, min(want, row_cap // tokens))


def _check_eig_options(which_matrix, lapnorm, image_color_lambda, image_downsample_factor, patch_size) -> str:
    """Validates the option combination and returns the ``spectral`` problem name it maps to."""
    if which_matrix == "affinity_torch":
        raise NotImplementedError("which_matrix='affinity_torch' is dead code in the reference (torch.eig was removed)")
    if which_matrix in ("affinity", "affinity_svd"):
        return which_matrix  # these branches ignore the Laplacian / colour options (extract.py:159-172)
    if which_matrix not in ("laplacian", "matting_laplacian"):
        raise ValueError(f"unknown which_matrix {which_matrix!r}")
    return "laplacian" if lapnorm else "laplacian_unnormalized"


def _color_spec(which_matrix: str, which_color_matrix: str, image_color_lambda: float, images_root: str):
    """``None`` or ``(lambda, 'knn' | 'rw', images_root)`` for the colour-fusion branch (extract.py:197-218)."""
    if which_matrix not in ("laplacian", "matting_laplacian") or not image_color_lambda > 0:
        return None
    if which_color_matrix not in ("knn", "rw"):
        # the reference leaves W_lr undefined here and dies with a NameError
        raise ValueError(f"which_color_matrix must be 'knn' or 'rw' (got {which_color_matrix!r})")
    return float(image_color_lambda), which_color_matrix, images_root


def _color_affinity(color, image_id: str, lr: Tuple[int, int], device: torch.device) -> torch.Tensor:
    """extract.py:201-213: the image, resized (PIL bilinear, like there) to the grid the features live on, -> dense
    ``W_color [N, N]`` f32 on the device."""
    import numpy as np
    from PIL import Image

    _, which, images_root = color
    image_lr = Image.open(str(Path(images_root) / f"{image_id}.jpg")).convert("RGB").resize((lr[1], lr[0]), Image.BILINEAR)
    image_lr = torch.from_numpy(np.array(image_lr) / 255.0).to(device)
    return utils.knn_affinity(image_lr) if which == "knn" else utils.rw_affinity(image_lr)


def _load_features(features_file: str, which_features: str) -> Tuple[dict, torch.Tensor]:
    data_dict = torch.load(features_file, map_location="cpu", weights_only=False)
    feats = data_dict[which_features].squeeze()
    if feats.dim() != 2:
        raise ValueError(f"{features_file}: expected [1, N, D] features, got {tuple(data_dict[which_features].shape)}")
    return data_dict, feats.to(torch.float32)


def _upsample_spec(data_dict: dict, which_matrix: str, image_downsample_factor: Optional[int]):
    """extract.py:178-188: the Laplacian branches resize the features to the (H_pad//f, W_pad//f) grid when that
    differs from the patch grid; the affinity branches never do."""
    if which_matrix not in ("laplacian", "matting_laplacian") or image_downsample_factor is None:
        return None
    _, _, _, _, p, h_patch, w_patch, h_pad, w_pad = utils.get_image_sizes(data_dict)
    lr = (h_pad // image_downsample_factor, w_pad // image_downsample_factor)
    return None if lr == (h_patch, w_patch) else ((h_patch, w_patch), lr)


def _lr_grid(data_dict: dict, image_downsample_factor: Optional[int]) -> Tuple[int, int]:
    """(H_pad_lr, W_pad_lr) of extract.py:178-181: the grid the Laplacian branches put the affinity on."""
    _, _, _, _, p, _, _, h_pad, w_pad = utils.get_image_sizes(data_dict)
    f = p if image_downsample_factor is None else image_downsample_factor
    return h_pad // f, w_pad // f


def _run_eig_batch(items: List[Tuple[str, torch.Tensor]], K: int, normalize: bool, threshold_at_zero: bool,
                   device: torch.device, saver: Optional["_AsyncSaver"] = None, problem: str = "laplacian",
                   upsample=None, color=None, color_items: Optional[List[Tuple[str, Tuple[int, int]]]] = None,
                   segment: Optional[dict] = None):
    # straight from wherever the loader left each file's features (a shared-memory segment when worker processes read
    # them) into the device batch: no host-side torch.stack pass over 1.4 MB per image
    feats = torch.empty((len(items),) + tuple(items[0][1].shape), dtype=torch.float32, device=device)
    for j, (_, f) in enumerate(items):
        feats[j].copy_(f, non_blocking=True)
    if color is not None:
        # extract.py:191-218 with a colour term: W_comb = W_feat / max(W_feat) + lambda * W_color, dense on the device
        # (the matrix is no longer a Gram matrix of the features), packed into the solver's tile layout, same kernel.
        chunks = []
        per = max(1, (4 << 30) // (4 * (feats.shape[1] if upsample is None else upsample[1][0] * upsample[1][1]) ** 2))
        for s in range(0, len(items), per):
            w = spectral.feature_affinity_dense(feats[s:s + per], normalize, threshold_at_zero, upsample)
            for j, (image_id, lr) in enumerate(color_items[s:s + per]):
                w[j] += color[0] * _color_affinity(color, image_id, lr, device)
            chunks.append(spectral.eigs_from_dense_affinity(w, K, problem))
            del w
        ev, vec, info = (torch.cat([c[i] for c in chunks]) for i in range(3))
    else:
        # strict=False: the reference never aborts a run over one image (bare except -> second solve,
        # extract.py:228-229).  laplacian_eigs_from_features re-solves a starved image with a larger Krylov space and,
        # failing that, densely; anything still flagged is saved as it is and reported here.
        ev, vec, info = spectral.laplacian_eigs_from_features(feats, K, normalize=normalize,
                                                              threshold_at_zero=threshold_at_zero, problem=problem,
                                                              upsample=upsample, strict=False)
    bad = (info <= 0).nonzero().flatten().tolist()
    if bad:
        print(f"[dss] WARNING: eigensolver did not converge for {[items[j][0] for j in bad]} (saved as is)")
    # the eigen files first: whatever happens in the optional segmentation below, the batch's results are on their way
    ev_h, vec_h = ev.cpu(), vec.cpu()
    if saver is None:
        for j, (output_file, _) in enumerate(items):
            torch.save(*_build_eig_file((ev_h, vec_h), (j, output_file, problem)))
    else:
        saver.submit_batch("eigs", (ev_h, vec_h), [(j, output_file, problem) for j, (output_file, _) in enumerate(items)])
    if segment is not None:
        # the segmentations of extract.py:283-426 straight from the device-resident eigenvectors,
        # no .pth round trip (same algorithms on the device: threshold of the Fiedler vector; Lloyd K-means + border vote)
        hp, wp = segment["grid"]
        pngs = []
        if segment.get("single_region_dir"):
            pngs.append((segment["single_region_dir"], spectral.single_region_masks(vec, segment["threshold"])))
        if segment.get("multi_region_dir"):
            # one launch per image, seeded from the image's own name: the k-means++ draw - hence the label numbering
            # before the border vote - does not depend on batch size, shape bucket or rank count
            import zlib
            maps = [spectral.multi_region_segments(
                ev[j:j + 1], vec[j:j + 1], (hp, wp), adaptive=segment["adaptive"],
                non_adaptive_num_segments=segment["num_segments"], infer_bg_index=segment["infer_bg_index"],
                num_eigenvectors=segment["num_eigenvectors"],
                seed=(int(segment["seed"]) ^ zlib.crc32(Path(output_file).stem.encode())) & 0x7fffffff).reshape(1, -1)
                for j, (output_file, _) in enumerate(items)]
            pngs.append((segment["multi_region_dir"], torch.cat(maps)))
        for out_dir, maps in pngs:
            maps = maps.cpu()
            png_items = [(j, str(Path(out_dir) / f"{Path(output_file).stem}.png"), hp, wp)
                         for j, (output_file, _) in enumerate(items)]
            if saver is None:
                for it in png_items:
                    _write_png(*_build_png_file((maps,), it))
            else:
                saver.submit_batch("png", (maps,), png_items)


def _extract_eig(inp: Tuple[int, str], K: int, images_root: str, output_dir: str,
                 which_matrix: str = "laplacian", which_features: str = "k", normalize: bool = True,
                 lapnorm: bool = True, which_color_matrix: str = "knn", threshold_at_zero: bool = True,
                 image_downsample_factor: Optional[int] = None, image_color_lambda: float = 10):
    """One feature file -> one eigen file (same signature as the reference's ``_extract_eig``; note its
    ``image_color_lambda`` default of 10 is only reachable by calling this function directly)."""
    index, features_file = inp
    data_dict, feats = _load_features(str(features_file), which_features)
    image_id = data_dict["file"][:-4]
    output_file = str(Path(output_dir) / f"{image_id}.pth")
    if Path(output_file).is_file():
        print(f"Skipping existing file {str(output_file)}")
        return
    problem = _check_eig_options(which_matrix, lapnorm, image_color_lambda, image_downsample_factor,
                                 data_dict["patch_size"])
    utils.get_image_sizes(data_dict)  # keeps the reference's B == 1 assertion
    color = _color_spec(which_matrix, which_color_matrix, image_color_lambda, images_root)
    _run_eig_batch([(output_file, feats)], K, normalize, threshold_at_zero, local_device(), problem=problem,
                   upsample=_upsample_spec(data_dict, which_matrix, image_downsample_factor), color=color,
                   color_items=[(image_id, _lr_grid(data_dict, image_downsample_factor))])


def extract_eigs(images_root: str, features_dir: str, output_dir: str, which_matrix: str = "laplacian",
                 which_color_matrix: str = "knn", which_features: str = "k", normalize: bool = True,
                 threshold_at_zero: bool = True, lapnorm: bool = True, K: int = 20,
                 image_downsample_factor: Optional[int] = None, image_color_lambda: float = 0.0,
                 multiprocessing: int = 0, batch_size: int = 64,
                 single_region_dir: Optional[str] = None, segmentation_threshold: float = 0.0,
                 multi_region_dir: Optional[str] = None, adaptive: bool = False, non_adaptive_num_segments: int = 4,
                 infer_bg_index: bool = True, num_eigenvectors: int = 1_000_000, kmeans_seed: int = 0):
    """Extracts eigenvalues/eigenvectors from features (see module docstring).  ``multiprocessing`` is
    accepted for CLI compatibility and ignored; ``batch_size`` same-shape images share one kernel launch.

    ``single_region_dir`` / ``multi_region_dir`` (not in the reference's command): also write the segmentation PNGs of
    ``extract_single_region_segmentations`` / ``extract_multi_region_segmentations`` from the eigenvectors while they are
    still on the device (``spectral.single_region_masks`` / ``multi_region_segments``; the Laplacian branches only),
    with those commands' own options (``segmentation_threshold`` is their ``threshold``)."""
    _make_output_dir_all_ranks(output_dir)
    _cap_host_threads()
    for extra in (single_region_dir, multi_region_dir):
        if extra:
            if which_matrix not in ("laplacian", "matting_laplacian"):
                raise ValueError("the on-device segmentations use eigenvector 1 of the Laplacian branches")
            Path(extra).mkdir(parents=True, exist_ok=True)
    if multi_region_dir and int(K) < 2:
        raise ValueError("--multi_region_dir clusters eigenvectors 1.. : K must be at least 2")
    kwargs = dict(K=K, which_matrix=which_matrix, which_features=which_features,
                  which_color_matrix=which_color_matrix, normalize=normalize, threshold_at_zero=threshold_at_zero,
                  images_root=images_root, output_dir=output_dir, image_downsample_factor=image_downsample_factor,
                  image_color_lambda=image_color_lambda, lapnorm=lapnorm)
    print(kwargs)
    if multiprocessing:
        print("[dss] multiprocessing flag ignored: images are batched on the GPU")
    rank, world = rank_world()
    device = local_device()
    files = sorted(Path(features_dir).iterdir())
    mine = files[rank::world]

    color = _color_spec(which_matrix, which_color_matrix, image_color_lambda, images_root)
    pending: Dict[Tuple, List[Tuple[str, torch.Tensor]]] = {}
    pending_ids: Dict[Tuple, List[Tuple[str, Tuple[int, int]]]] = {}
    problems: Dict[Tuple, str] = {}

    def run(key):
        ids = pending_ids.pop(key)
        segment = None
        if single_region_dir or multi_region_dir:
            segment = dict(grid=ids[0][1], single_region_dir=single_region_dir, threshold=segmentation_threshold,
                           multi_region_dir=multi_region_dir, adaptive=adaptive, num_segments=non_adaptive_num_segments,
                           infer_bg_index=infer_bg_index, num_eigenvectors=num_eigenvectors, seed=kmeans_seed)
        _run_eig_batch(pending.pop(key), K, normalize, threshold_at_zero, device, saver, problems[key], key[1], color, ids,
                       segment)

    bs = max(1, int(batch_size))
    n_pending, max_pending = 0, 8 * bs   # mixed-size datasets (VOC): bound the features waiting in host RAM
    # torch.load of a 1.4 MB feature file costs ~0.6 ms and torch.save of an 18 KB eigen file less: few workers do
    nproc = _io_processes(len(mine), most=16)
    clock = _StageClock()
    with clock("start savers"):
        # (the 'affinity' branch stores its eigenvalues as a raw numpy array - extract.py:171,243 -: `torch.save` on threads)
        saver = _FastSaver(min(nproc, 8)) if nproc > 0 and which_matrix != "affinity" else _AsyncSaver()
    loaded = _iter_features(mine, which_features, nproc, 4 * bs, device)
    scheduled = set()
    try:
        while True:
            with clock("wait for loaded features"):
                nxt = next(loaded, None)
            if nxt is None:
                break
            data_dict, feats = nxt
            image_id = data_dict["file"][:-4]
            output_file = str(Path(output_dir) / f"{image_id}.pth")
            # the reference writes synchronously, so a second feature file naming the same image finds the first one's
            # output and is skipped (extract.py:141-146); here the first may still be in flight: remember what is scheduled
            pngs_there = all(Path(dd, f"{image_id}.png").is_file() for dd in (single_region_dir, multi_region_dir) if dd)
            if output_file in scheduled or (Path(output_file).is_file() and pngs_there):
                print(f"Skipping existing file {str(output_file)}")
                continue
            if Path(output_file).is_file():   # an earlier run wrote the eigen file but not the requested PNGs: redo both
                print(f"[dss] {image_id}: eigen file exists but a requested segmentation PNG does not - recomputing")
            scheduled.add(output_file)
            problem = _check_eig_options(which_matrix, lapnorm, image_color_lambda, image_downsample_factor,
                                         data_dict["patch_size"])
            utils.get_image_sizes(data_dict)
            up = _upsample_spec(data_dict, which_matrix, image_downsample_factor)
            # same feature shape, same resize target AND same grid (20 x 30 and 30 x 20 patches have the same N) share a launch
            key = (tuple(feats.shape), up, _lr_grid(data_dict, image_downsample_factor))
            pending.setdefault(key, []).append((output_file, feats))
            pending_ids.setdefault(key, []).append((image_id, _lr_grid(data_dict, image_downsample_factor)))
            problems[key] = problem
            n_pending += 1
            if len(pending[key]) < bs and n_pending >= max_pending:
                key = max(pending, key=lambda k: len(pending[k]))   # flush the fullest bucket early
            if len(pending[key]) >= bs or n_pending >= max_pending:
                n_pending -= len(pending[key])
                with clock("run batches"):
                    run(key)
        for key in list(pending):
            with clock("run batches"):
                run(key)
        with clock("tail: savers finish"):
            saver.close()
    except BaseException:
        saver.abort()        # stop the workers, unlink the /dev/shm ring before the error leaves
        raise
    clock.report("extract_eigs")
    _barrier()


# ------------------------------------------------------------------------------------------ consumers of the eigen files
# SURVEY.md §8f rows 1-2.  The commands keep the reference's names, flags, defaults, file pairing and outputs; the work
# itself runs on the device kernels the in-memory pipeline uses (`extract_eigs --single_region_dir / --multi_region_dir`
# writes the same PNGs without the .pth round trip).

def _segmentation_job(pair, output_dir: str):
    """One paired (feature file, eigen file) -> ``(png path, (rows, cols) of the patch grid, feature dict, eigen dict)``;
    ``None`` (with the reference's message) when the PNG is already there."""
    feature_path, eigs_path = pair
    meta = torch.load(feature_path, map_location="cpu", weights_only=False)
    target = Path(output_dir) / f"{Path(meta['id'])}.png"
    if target.is_file():
        print(f"Skipping existing file {str(target)}")
        return None
    sizes = utils.get_image_sizes(meta)
    return target, (int(sizes[5]), int(sizes[6])), meta, torch.load(eigs_path, map_location="cpu", weights_only=False)


def _consumer_device() -> torch.device:
    """Where the segmentation commands compute: the GPU when there is one (the kernels of `csrc/segment.hip`), the host
    otherwise - the reference ran them on the CPU, and `spectral.single_region_masks` / `multi_region_segments` / `kmeans_lloyd`
    are plain tensor code for host tensors (the two hot commands have no such fallback)."""
    return local_device() if torch.cuda.is_available() else torch.device("cpu")


def _save_label_png(labels: torch.Tensor, target: Path) -> None:
    from PIL import Image

    Image.fromarray(labels.to("cpu", torch.uint8).numpy(), mode="L").save(str(target))


def _extract_single_region_segmentations(inp: Tuple[int, Tuple[str, str]], threshold: float, output_dir: str):
    """reference extract/extract.py:383-407: the 0 / 255 mask of ``eigenvectors[1] > threshold`` on the ``(H//P, W//P)``
    grid, from ``dss_fiedler_mask``."""
    job = _segmentation_job(inp[1], output_dir)
    if job is None:
        return
    target, (rows, cols), _, eig = job
    vec = eig["eigenvectors"].to(_consumer_device(), torch.float32)[None]
    _save_label_png(spectral.single_region_masks(vec, threshold)[0].view(rows, cols), target)


def extract_single_region_segmentations(features_dir: str, eigs_dir: str, output_dir: str, threshold: float = 0.0,
                                        multiprocessing: int = 0):
    """python extract.py extract_single_region_segmentations --features_dir F --eigs_dir E --output_dir O"""
    utils.make_output_dir(output_dir)
    fn = partial(_extract_single_region_segmentations, threshold=threshold, output_dir=output_dir)
    utils.parallel_process(utils.get_paired_input_files(features_dir, eigs_dir), fn, multiprocessing)


def _extract_multi_region_segmentations(inp: Tuple[int, Tuple[str, str]], adaptive: bool, non_adaptive_num_segments: int,
                                        infer_bg_index: bool, kmeans_baseline: bool, output_dir: str,
                                        num_eigenvectors: int, seed: int = 0):
    """reference extract/extract.py:283-352: label map of a K-means over the non-constant eigenvectors (``adaptive``: as
    many segments as the largest eigengap says; ``infer_bg_index``: the segment owning the border becomes 0), on the
    patch grid or - eigenvectors of the 2x upsampled grid - on twice that.  Clustering, border vote and label swap are
    ``dss_kmeans_segments`` (one workgroup per image); problems beyond its limits (more than 8192 points, 64 coordinates or
    32 segments) and the ``kmeans_baseline`` (raw features as coordinates) take ``spectral.kmeans_lloyd`` on the device.
    The reference's ``KMeans()`` is unseeded - its labels are a random variable; here ``seed`` fixes them."""
    job = _segmentation_job(inp[1], output_dir)
    if job is None:
        return
    target, (rows, cols), meta, eig = job
    dev = _consumer_device()
    lam = eig["eigenvalues"].to(dev, torch.float32)[None]
    vec = eig["eigenvectors"].to(dev, torch.float32)[None]
    points = vec.shape[-1]
    scale = {rows * cols: 1, 4 * rows * cols: 2}.get(points)
    if scale is None:
        raise ValueError(f"{meta['id']}: {points} eigenvector entries for a {rows} x {cols} patch grid")
    grid = (scale * rows, scale * cols)
    if kmeans_baseline:
        feats = meta["k"].to(dev, torch.float32).reshape(-1, meta["k"].shape[-1])
        if feats.shape[0] != points:
            raise ValueError(f"{meta['id']}: kmeans_baseline needs one feature row per eigenvector entry")
        k = spectral.adaptive_num_segments(lam)[0] if adaptive else int(non_adaptive_num_segments)
        labels = spectral.kmeans_lloyd(feats, k, seed=seed, n_init=1 if feats.is_cuda else 4).view(grid)
        if infer_bg_index:
            labels = spectral.border_owner_to_zero(labels)
    else:
        labels = spectral.multi_region_segments(lam, vec, grid, adaptive=adaptive, infer_bg_index=infer_bg_index,
                                                non_adaptive_num_segments=non_adaptive_num_segments,
                                                num_eigenvectors=num_eigenvectors, seed=seed)[0]
    _save_label_png(labels, target)


def extract_multi_region_segmentations(features_dir: str, eigs_dir: str, output_dir: str, adaptive: bool = False,
                                       non_adaptive_num_segments: int = 4, infer_bg_index: bool = True,
                                       kmeans_baseline: bool = False, num_eigenvectors: int = 1_000_000,
                                       multiprocessing: int = 0):
    """python extract.py extract_multi_region_segmentations --features_dir F --eigs_dir E --output_dir O"""
    utils.make_output_dir(output_dir)
    fn = partial(_extract_multi_region_segmentations, adaptive=adaptive, infer_bg_index=infer_bg_index,
                 non_adaptive_num_segments=non_adaptive_num_segments, num_eigenvectors=num_eigenvectors,
                 kmeans_baseline=kmeans_baseline, output_dir=output_dir)
    utils.parallel_process(utils.get_paired_input_files(features_dir, eigs_dir), fn, multiprocessing)


def _extract_bbox(inp: Tuple[int, Tuple[str, str]], num_erode: int, num_dilate: int, skip_bg_index: bool,
                  downsample_factor: Optional[int] = None) -> dict:
    """reference extract/extract.py:429-470: the box of every segment of a label PNG after ``num_erode`` erosions and
    ``num_dilate`` dilations (``utils.segment_boxes``), ``(xmin, ymin, xmax, ymax)`` with exclusive maxima, in grid units
    and - times the patch size (or ``downsample_factor``) - in pixels.  Schema of the per-image dict: SURVEY.md §8f."""
    import numpy as np
    from PIL import Image

    feature_path, segmentation_path = inp[1]
    meta = torch.load(feature_path, map_location="cpu", weights_only=False)
    unit = int(utils.get_image_sizes(meta, downsample_factor)[4])
    labels, boxes = utils.segment_boxes(np.asarray(Image.open(str(segmentation_path))), num_erode, num_dilate,
                                        include_background=not skip_bg_index)
    return {"bboxes": boxes, "bboxes_original_resolution": [[unit * v for v in box] for box in boxes],
            "segment_indices": labels, "id": meta["id"], "format": "(xmin, ymin, xmax, ymax)"}


def extract_bboxes(features_dir: str, segmentations_dir: str, output_file: str, num_erode: int = 2, num_dilate: int = 3,
                   skip_bg_index: bool = True, downsample_factor: Optional[int] = None):
    """python extract.py extract_bboxes --features_dir F --segmentations_dir S --num_erode 2 --num_dilate 5
    --output_file bboxes.pth   (one list of per-image dicts in ONE file, reference extract/extract.py:473-495)"""
    utils.make_output_dir(str(Path(output_file).parent), check_if_empty=False)
    pairs = utils.get_paired_input_files(features_dir, segmentations_dir)
    torch.save([_extract_bbox(pair, num_erode, num_dilate, skip_bg_index, downsample_factor) for pair in pairs], output_file)
    print("Done")


def extract_bbox_features(images_root: str, bbox_file: str, model_name: str, output_file: str,
                          weights: Optional[str] = None, dtype: str = "float16",
                          synthetic_weights: Optional[int] = None):
    """python extract.py extract_bbox_features --model_name dino_vits16 --images_root I --bbox_file bboxes.pth
    --output_file bbox_features.pth   (reference extract/extract.py:498-544): the ViT's CLS output (all blocks + final
    LayerNorm, ``DinoViT.forward_cls``) of every box crop, stacked per image under ``'features'``.  Crops of equal size
    inside an image share one forward."""
    import numpy as np
    from PIL import Image

    bbox_list = torch.load(bbox_file, weights_only=False)
    total_num_boxes = sum(len(d["bboxes"]) for d in bbox_list)
    print(f"Loaded bounding box list. There are {total_num_boxes} total bounding boxes.")
    device = local_device()
    model, _, _, _ = utils.get_model(model_name.lower(), device=device, dtype=_DTYPES[str(dtype).lower()],
                                     weights=weights, synthetic_seed=synthetic_weights)
    for bbox_dict in bbox_list:
        image_filename = str(Path(images_root) / f"{bbox_dict['id']}.jpg")
        image = torch.from_numpy(np.asarray(Image.open(image_filename).convert("RGB")).copy()).to(device)  # [H, W, 3] u8
        boxes = [tuple(int(v) for v in box) for box in bbox_dict["bboxes_original_resolution"]]
        feats: List[Optional[torch.Tensor]] = [None] * len(boxes)
        for group in spectral.group_by_shape([(b[3] - b[1], b[2] - b[0]) for b in boxes], 64):
            crops = torch.stack([image[boxes[j][1]:boxes[j][3], boxes[j][0]:boxes[j][2]] for j in group])
            out = model.forward_cls(crops).cpu()
            for j, f in zip(group, out):
                feats[j] = f
        bbox_dict["features"] = torch.stack(feats, dim=0)   # raises on an image without boxes, as the reference does
    torch.save(bbox_list, output_file)
    print(f"Saved features to {output_file}")


# ------------------------------------------------------------------------------------------ CLI
COMMANDS = dict(extract_features=extract_features, extract_eigs=extract_eigs,
                extract_single_region_segmentations=extract_single_region_segmentations,
                extract_multi_region_segmentations=extract_multi_region_segmentations,
                extract_bboxes=extract_bboxes, extract_bbox_features=extract_bbox_features)


def _literal(s: str):
    try:
        return ast.literal_eval(s)
    except (ValueError, SyntaxError):
        return s


def parse_cli(argv: List[str]):
    """python-fire style: ``<command> --flag value`` or ``--flag=value``; values are Python literals when
    they parse as one (True / None / 5 / 0.5), strings otherwise; a bare ``--flag`` means True."""
    if not argv or argv[0] not in COMMANDS:
        raise SystemExit(f"usage: extract.py {{{'|'.join(COMMANDS)}}} --flag value ...")
    fn = COMMANDS[argv[0]]
    params = inspect.signature(fn).parameters
    kwargs, i = {}, 1
    while i < len(argv):
        a = argv[i]
        if not a.startswith("--"):
            raise SystemExit(f"unexpected positional argument {a!r}")
        if "=" in a:
            key, val = a[2:].split("=", 1)
            i += 1
        elif i + 1 < len(argv) and not argv[i + 1].startswith("--"):
            key, val = a[2:], argv[i + 1]
            i += 2
        else:
            key, val = a[2:], "True"
            i += 1
        key = key.replace("-", "_")
        if key == "yes":
            os.environ["DSS_ASSUME_YES"] = "1"
            continue
        if key not in params:
            raise SystemExit(f"{argv[0]}: unknown flag --{key} (known: {', '.join(params)})")
        kwargs[key] = _literal(val)
    missing = [n for n, p in params.items() if p.default is inspect.Parameter.empty and n not in kwargs]
    if missing:
        raise SystemExit(f"{argv[0]}: missing required flags: {', '.join('--' + m for m in missing)}")
    return fn, kwargs


def main(argv: Optional[List[str]] = None):
    torch.set_grad_enabled(False)  # extract/extract.py:838
    fn, kwargs = parse_cli(sys.argv[1:] if argv is None else argv)
    init_process_group()  # no-op for a single process; under torchrun: one rank per GPU, round-robin shards
    fn(**kwargs)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
