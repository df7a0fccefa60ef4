"""What the I/O worker processes of the two commands run - without torch: the feature-file reader of ``extract_eigs``
and the image decoder of ``extract_features``.

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


def _block(path: str, size: int) -> mmap.mmap:
    hit = _BLOCKS.get(path)
    if hit is None or hit[1] != size:
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
