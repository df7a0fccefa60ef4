"""What the I/O worker processes of the two commands run - without torch: the feature-file reader of ``extract_eigs``,
the image decoder of ``extract_features`` and (round 5) the WRITER of both commands' ``.pth`` files.

A loader process exists to turn ``<id>.pth`` files (the reference's schema, extract/extract.py:98-110, written by
``torch.save``: a ZIP archive of STORED members - ``data.pkl`` plus one raw member per tensor storage) into feature rows
the GPU can fetch.  ``torch.load`` needs ``import torch`` in every worker (1.3-2 s before the first file is read - more
than the whole run of a 4 096-image set), an unpickled tensor per file and a second copy into shared memory.  This
module needs ``pickle``, ``zipfile``, ``mmap`` and numpy: a worker is up in ~0.3 s, parses the pickle with stand-ins for
the three torch globals it contains, and reads the feature bytes of each file STRAIGHT INTO a block of ``/dev/shm`` that
the parent has page-locked for the copy engine (disk cache -> block -> HBM, no other copy).

Anything unexpected (compressed members, non-float features, unknown pickled classes) is reported per
file; the parent then loads that file with ``torch.load`` itself.  Only ``tests/`` and ``extract.py`` import this."""
from __future__ import annotations

import mmap
import os
import pickle
import struct
import zipfile
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

_STORAGE_DTYPES = {"FloatStorage": np.float32, "HalfStorage": np.float16, "DoubleStorage": np.float64,
                   "LongStorage": np.int64, "IntStorage": np.int32, "ShortStorage": np.int16, "CharStorage": np.int8,
                   "ByteStorage": np.uint8, "BoolStorage": np.bool_}


class Unsupported(Exception):
    pass


def noop(_=None):
    """What a freshly started worker answers to say it is up (extract._StaggeredPool starts the next wave then)."""
    return None


class _StorageType:
    def __init__(self, name: str):
        self.dtype = np.dtype(_STORAGE_DTYPES[name])


class TensorRef:
    """A tensor of the archive that has not been read: member ``key``, element offset, size, stride, dtype."""

    def __init__(self, dtype, key: str, offset: int, size: Tuple[int, ...], stride: Tuple[int, ...]):
        self.dtype, self.key, self.offset, self.size, self.stride = dtype, key, int(offset), tuple(size), tuple(stride)

    @property
    def numel(self) -> int:
        n = 1
        for s in self.size:
            n *= s
        return n

    def is_contiguous(self) -> bool:
        expect = 1
        for size, stride in zip(reversed(self.size), reversed(self.stride)):
            if size != 1 and stride != expect:
                return False
            expect *= size
        return True


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    dtype, key = storage
    return TensorRef(dtype, key, storage_offset, size, stride)


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "torch._utils" and name == "_rebuild_tensor_v2":
            return _rebuild_tensor_v2
        if module == "torch" and name in _STORAGE_DTYPES:
            return _StorageType(name)
        if module == "collections" and name == "OrderedDict":
            return OrderedDict
        raise Unsupported(f"pickled global {module}.{name}")

    def persistent_load(self, pid):
        # ('storage', storage type, key, location, numel) - torch/serialization.py::_save
        if not (isinstance(pid, tuple) and len(pid) >= 3 and pid[0] == "storage" and isinstance(pid[1], _StorageType)):
            raise Unsupported(f"persistent id {pid!r}")
        return pid[1].dtype, str(pid[2])


class PthFile:
    """``with PthFile(path) as f``: ``f.obj`` is the pickled object with ``TensorRef`` in place of tensors."""

    def __init__(self, path: str):
        self.zip = zipfile.ZipFile(path)
        names = self.zip.namelist()
        pkl = [n for n in names if n.endswith("/data.pkl") or n == "data.pkl"]
        if len(pkl) != 1:
            self.zip.close()
            raise Unsupported("not a torch.save zip archive")
        self.prefix = pkl[0][:-len("data.pkl")]
        order = self.prefix + "byteorder"
        if order in names and self.zip.read(order).strip() != b"little":
            self.zip.close()
            raise Unsupported("big-endian archive")
        with self.zip.open(pkl[0]) as fh:
            self.obj = _Unpickler(fh).load()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.zip.close()
        return False

    def read_into(self, ref: TensorRef, out: memoryview) -> None:
        """The elements of ``ref`` in row-major order into ``out`` (exactly ``numel * itemsize`` bytes).  A contiguous
        tensor is read straight from the archive member; a strided VIEW (what the reference saves when it runs on the
        CPU: ``k`` is a slice of the whole qkv activation, extract/extract.py:96-98, and ``torch.save`` stores that
        storage) is gathered from the member's bytes."""
        info = self.zip.getinfo(f"{self.prefix}data/{ref.key}")
        if info.compress_type != zipfile.ZIP_STORED:
            raise Unsupported("compressed member")
        nbytes = ref.numel * ref.dtype.itemsize
        if not ref.is_contiguous():
            if any(st < 0 for st in ref.stride) or ref.numel == 0:
                raise Unsupported("strided tensor with negative strides")
            last = ref.offset + sum((sz - 1) * st for sz, st in zip(ref.size, ref.stride))
            if (last + 1) * ref.dtype.itemsize > info.file_size:
                raise Unsupported("strided tensor outside its storage")
            store = np.frombuffer(self.zip.read(info), dtype=ref.dtype)
            view = np.lib.stride_tricks.as_strided(store[ref.offset:], shape=ref.size,
                                                   strides=tuple(st * ref.dtype.itemsize for st in ref.stride))
            np.frombuffer(out, dtype=ref.dtype, count=ref.numel).reshape(ref.size)[...] = view
            return
        with self.zip.open(info) as fh:
            skip = ref.offset * ref.dtype.itemsize
            if skip:
                fh.seek(skip)
            got = fh.readinto(out) if hasattr(fh, "readinto") else None
            if got is None:
                data = fh.read(nbytes)
                out[:len(data)] = data
                got = len(data)
            while got < nbytes:      # readinto may return short counts
                more = fh.readinto(out[got:])
                if not more:
                    raise Unsupported("truncated member")
                got += more

    def read(self, ref: TensorRef) -> np.ndarray:
        arr = np.empty(ref.numel, dtype=ref.dtype)
        self.read_into(ref, memoryview(arr).cast("B"))
        return arr.reshape(ref.size)


_BLOCKS: Dict[str, Tuple[mmap.mmap, int]] = {}


def _block_slot(path: str) -> str:
    """`dss_<owner>_s<slot>_<nbytes>` -> `dss_<owner>_s<slot>`: the ring slot a block path belongs to (a slot that GROWS gets a new
    path; the mapping of its old, already unlinked block must go or its tmpfs pages stay allocated in every worker)."""
    head, _, tail = path.rpartition("_")
    return head if tail.isdigit() and head else path


def _block(path: str, size: int) -> mmap.mmap:
    hit = _BLOCKS.get(path)
    if hit is None or hit[1] != size:
        slot = _block_slot(path)
        for old in [k for k in _BLOCKS if k != path and _block_slot(k) == slot]:   # the slot was re-created at another size
            try:
                _BLOCKS.pop(old)[0].close()
            except (BufferError, ValueError):   # an exported view is still alive: dropped with it
                pass
        if hit is not None:
            try:
                hit[0].close()
            except (BufferError, ValueError):
                pass
        fd = os.open(path, os.O_RDWR)
        try:
            hit = (mmap.mmap(fd, size), size)
        finally:
            os.close(fd)
        _BLOCKS[path] = hit
    return hit[0]


def load_chunk(block_path: str, block_size: int, files: List[str], which_features: str):
    """Worker entry.  Reads ``data_dict[which_features]`` of every file of ``files`` as float32 rows, packed one after
    the other from byte 0 of the shared block.  Returns one entry per file, in order:
    ``(meta, byte offset, (N, D))`` or ``(None, file, reason)`` for a file the parent has to load itself.
    ``meta``: the pickled dict without tensors, tensors of <= 16 elements as Python values."""
    buf = memoryview(_block(block_path, block_size))
    out, used = [], 0
    for f in files:
        try:
            with PthFile(str(f)) as pth:
                d = pth.obj
                ref = d[which_features]
                if not isinstance(d, dict) or not isinstance(ref, TensorRef):
                    raise Unsupported("unexpected layout")
                shape = tuple(s for s in ref.size if s != 1)   # .squeeze()
                if len(shape) != 2:
                    raise ValueError(f"{f}: expected [1, N, D] features, got {ref.size}")
                nbytes = ref.numel * 4
                if used + nbytes > block_size:
                    raise Unsupported("block full")
                if ref.dtype == np.float32:
                    pth.read_into(ref, buf[used:used + nbytes])
                elif ref.dtype in (np.float16, np.float64):
                    np.frombuffer(buf, dtype=np.float32, count=ref.numel, offset=used)[:] = pth.read(ref).reshape(-1)
                else:
                    raise Unsupported(f"feature dtype {ref.dtype}")
                meta = {}
                for k, v in d.items():
                    if isinstance(v, TensorRef):
                        if v.numel <= 16 and k != which_features:
                            meta[k] = pth.read(v).tolist()
                    else:
                        meta[k] = v
            out.append((meta, used, shape))
            used += nbytes
        except ValueError:
            raise
        except Exception as e:   # the parent falls back to torch.load for this file
            out.append((None, str(f), f"{type(e).__name__}: {e}"))
    return out


def decode_rgb(path) -> np.ndarray:
    """One image file -> u8 RGB ``[H, W, 3]``.  cv2.imread (the reference's decoder, extract_utils.py:30) applies the
    EXIF orientation; PIL does not unless asked to."""
    from PIL import Image, ImageOps

    with Image.open(path) as im:
        return np.array(ImageOps.exif_transpose(im).convert("RGB"), dtype=np.uint8)


def decode_chunk(block_path: str, block_size: int, files: List[str]):
    """Worker entry of ``extract_features``: decodes ``files`` into the shared block, packed from byte 0.  One entry per
    file, in order: ``(byte offset, (H, W, 3))``, or ``(None, reason)`` when the image does not fit what is left of the
    block (the parent decodes it itself).  Decoding in PROCESSES: thirty decode threads keep the parent's GIL ~80 % busy
    with their Python-level steps, and every kernel launch of the ViT then queues for it (100 ms per 128-image batch)."""
    buf = np.frombuffer(_block(block_path, block_size), dtype=np.uint8)
    out, used = [], 0
    for f in files:
        img = decode_rgb(f)
        if used + img.size > block_size:
            out.append((None, "block full"))
            continue
        buf[used:used + img.size] = img.reshape(-1)
        out.append((used, tuple(img.shape)))
        used += img.size
    return out


# ---------------------------------------------------------------------------------------------------------------------
# The writer (round 5).  A saver process used to `import torch` (1.5-2 s before the first file of a run could be written:
# "first feature file after 3.8 s", profiles/r04_cli_throughput.txt) to call `torch.save` on tensors it had received
# through torch's shared-memory pickling.  What `torch.save` writes for the two schemas of this path (extract/extract.py:98-110
# features, :235,243-244 eigenvectors) is a ZIP of STORED members: `<stem>/data.pkl` - a protocol-2 pickle in which every
# tensor is `torch._utils._rebuild_tensor_v2(<persistent id of its storage>, offset, size, stride, False, OrderedDict())` -,
# `<stem>/byteorder`, one raw member `<stem>/data/<n>` per storage and `<stem>/version`.  Those few opcodes are written here
# directly; the tensor bytes go from the page-locked /dev/shm block the GPU copied them into straight into the archive
# (`zipfile` computes the CRC on the way).  `torch.load(..., weights_only=True)` reads the result like any other file
# (tests/test_host_logic.py::test_torch_free_writer_*).
_STORAGE_NAMES = {np.dtype(v).str: k for k, v in _STORAGE_DTYPES.items()}


class TensorOut:
    """A tensor to be written: a C-contiguous numpy array (any shape, one of the storage dtypes)."""

    def __init__(self, array: np.ndarray):
        self.array = np.require(array, requirements="C")      # (np.ascontiguousarray would turn a 0-dim array into [1])
        if self.array.dtype.str not in _STORAGE_NAMES:
            raise Unsupported(f"tensor dtype {self.array.dtype}")


def _pk_int(i: int) -> bytes:
    if 0 <= i < 256:
        return b"K" + bytes([i])
    if 0 <= i < 65536:
        return b"M" + struct.pack("<H", i)
    if -2 ** 31 <= i < 2 ** 31:
        return b"J" + struct.pack("<i", i)
    raw = i.to_bytes((i.bit_length() + 8) // 8, "little", signed=True)
    return b"\x8a" + bytes([len(raw)]) + raw


def _pk_str(s: str) -> bytes:
    raw = s.encode("utf-8")
    return b"X" + struct.pack("<I", len(raw)) + raw


def _pk_tuple(items: List[bytes]) -> bytes:
    if len(items) <= 3:
        return b"".join(items) + (b")", b"\x85", b"\x86", b"\x87")[len(items)]
    return b"(" + b"".join(items) + b"t"


def _pk_value(v, storages: List[np.ndarray]) -> bytes:
    if isinstance(v, TensorOut):
        a = v.array
        key = str(len(storages))
        storages.append(a)
        strides, acc = [], 1
        for n in reversed(a.shape):
            strides.append(acc)
            acc *= n
        pid = b"(" + _pk_str("storage") + b"ctorch\n" + _STORAGE_NAMES[a.dtype.str].encode() + b"\n" + _pk_str(key) + \
            _pk_str("cpu") + _pk_int(a.size) + b"t" + b"Q"
        return b"ctorch._utils\n_rebuild_tensor_v2\n" + b"(" + pid + _pk_int(0) + _pk_tuple([_pk_int(n) for n in a.shape]) + \
            _pk_tuple([_pk_int(n) for n in reversed(strides)]) + b"\x89" + b"ccollections\nOrderedDict\n" + b")R" + b"t" + b"R"
    if v is None:
        return b"N"
    if v is True or v is False:
        return b"\x88" if v else b"\x89"
    if isinstance(v, (int, np.integer)):
        return _pk_int(int(v))
    if isinstance(v, (float, np.floating)):
        return b"G" + struct.pack(">d", float(v))
    if isinstance(v, str):
        return _pk_str(v)
    if isinstance(v, tuple):
        return _pk_tuple([_pk_value(x, storages) for x in v])
    if isinstance(v, dict):
        body = b"".join(_pk_value(k, storages) + _pk_value(x, storages) for k, x in v.items())
        return b"}" + (b"(" + body + b"u" if v else b"")
    raise Unsupported(f"cannot write a {type(v).__name__}")


def dumps_pth(obj) -> Tuple[bytes, List[np.ndarray]]:
    """``(data.pkl bytes, storages in key order)`` for ``obj``: nested dicts / tuples of str, int, float, bool, None and
    ``TensorOut`` (every tensor gets its own storage, like tensors that share none in ``torch.save``)."""
    storages: List[np.ndarray] = []
    body = _pk_value(obj, storages)
    return b"\x80\x02" + body + b".", storages


def write_pth(path: str, obj) -> None:
    """``torch.save(obj, path)`` for the object kinds ``dumps_pth`` knows, without torch.  Written to ``path + '.tmp'`` and renamed:
    a killed run leaves no half-written file behind for the next run's skip-if-exists to trust."""
    pkl, storages = dumps_pth(obj)
    stem = os.path.splitext(os.path.basename(path))[0] or "archive"
    tmp = f"{path}.tmp{os.getpid()}"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_STORED, allowZip64=True) as z:
        z.writestr(f"{stem}/data.pkl", pkl)
        z.writestr(f"{stem}/byteorder", b"little")
        for i, a in enumerate(storages):
            with z.open(f"{stem}/data/{i}", "w", force_zip64=a.nbytes >= (1 << 31)) as fh:
                fh.write(memoryview(a).cast("B") if a.size else b"")
        z.writestr(f"{stem}/version", b"3\n")
    os.replace(tmp, path)


def save_chunk(block_path: str, block_size: int, kind: str, items: list):
    """Worker entry of both commands' savers: the tensors of ``items`` lie in the shared block (the GPU copied them there).
    ``kind == "features"``: item = (byte offset, (N, D), index, file, model_name, patch_size, shape, out path) -> the reference's
    feature dict (extract/extract.py:98-110: ``k`` [1, N, D] f32, ``indices`` a 0-dim int64 tensor, ...).
    ``kind == "eigs"``: item = (offset of the [K, N] f32 eigenvectors, offset of the [K] f32 eigenvalues, K, N, out path) ->
    ``{"eigenvalues", "eigenvectors"}`` (:235,243-244).  Returns the number of files written."""
    buf = _block(block_path, block_size)
    for it in items:
        if kind == "features":
            off, (n, d), index, file, model_name, patch_size, shape, out = it
            k = np.frombuffer(buf, dtype=np.float32, count=n * d, offset=off).reshape(1, n, d)
            write_pth(out, {"k": TensorOut(k), "indices": TensorOut(np.array(index, dtype=np.int64)), "file": file,
                            "id": os.path.splitext(os.path.basename(file))[0], "model_name": model_name,
                            "patch_size": int(patch_size), "shape": tuple(int(v) for v in shape)})
        elif kind == "eigs":
            voff, eoff, kk, n, out = it
            vec = np.frombuffer(buf, dtype=np.float32, count=kk * n, offset=voff).reshape(kk, n)
            val = np.frombuffer(buf, dtype=np.float32, count=kk, offset=eoff)
            write_pth(out, {"eigenvalues": TensorOut(val), "eigenvectors": TensorOut(vec)})
        else:
            raise ValueError(kind)
    return len(items)


def save_eigs(items: list):
    """Worker entry: ``items`` = ``(eigenvalues [K] f32 array, eigenvectors [K, N] f32 array, out path)`` per image (18 KB per
    image: the arrays travel through the pool's pipe, no shared block) -> the reference's eigen dict (extract/extract.py:235,243-244)."""
    for val, vec, out in items:
        write_pth(out, {"eigenvalues": TensorOut(np.asarray(val, np.float32)), "eigenvectors": TensorOut(np.asarray(vec, np.float32))})
    return len(items)


def save_pngs(items: list):
    """Worker entry: ``(u8 [rows, cols] label / mask map, out path)`` per image -> one 8-bit PNG (extract/extract.py:352,405)."""
    from PIL import Image

    for arr, out in items:
        Image.fromarray(np.asarray(arr, np.uint8)).save(out)
    return len(items)
