#!/usr/bin/env python
"""bench.py - images/sec of the extract hot path (features + eigs) on MI355X, with the kernel roofline
and a CPU baseline (the oracle, i.e. the reference's numpy/scipy/torch-CPU path) timed beside it.

    python bench.py [--gpus N --steps K --warmup W]

``--gpus N`` with N > 1 launches itself as N ranks (``python -m torch.distributed.run --nproc-per-node N``, one process
per GPU over RCCL) unless it already runs under such a launcher (WORLD_SIZE set); the line reports ``ranks_seen``.

Workload (BASELINE.json configs[1]): dino_vits16, 480x480 synthetic VOC-shaped images, K=5.
One STEP = one batch of ``--batch`` images (default 7 ViT forwards of 290 = 2030 at the headline config):
H2D of the uint8 HWC images (pinned host memory -> HBM on a copy stream, double buffered: the batch of step s+1 travels
while step s computes - INSIDE the timed region) -> transform+crop+im2col -> ViT (HIP
LayerNorm/attention/K-resident Linear kernels, hipBLASLt for the other GEMMs) -> K features -> normalise -> affinity
-> Lanczos eigenpairs -> [K, N] eigenvectors -> D2H of every rank's own results to pinned host memory (where the CLI
writes the per-image .pth files from).  One ``B=1`` result per image, like the reference.
Multi-GPU: no collective on the data path; rank 0 collects all results once at the end (sizes first, then one flat
payload per rank, point to point over xGMI) - inside the timed region.
  * default: every rank runs ``--steps`` steps on its own images: ``"scaling": "weak"``;
  * ``--dataset D`` (BASELINE configs[3]: 10000): a fixed set of D images, item i on rank i % world, per-rank steps
    sized from the shard: ``"scaling": "strong"``.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
import dss_amd  # noqa: E402,F401
from dss_amd import distributed, hip, pipeline, spectral, synthetic  # noqa: E402
from dss_amd.vit import DinoViT  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA16_PEAK_TF = 2500.0     # dense bf16/fp16 MFMA
MFMA32_PEAK_TF = 157.3      # fp32 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12,
                    help="timed steps (12 x 4072 images = 3.6 s at the headline config: with 4, the first chunk's exposed "
                         "H2D copy and the end-of-run collection weigh 17 ms per step, with 12 or more 5-6 ms)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0,
                    help="images per step per GPU (0 = 4..8 ViT forwards, whichever fills whole rounds of eigensolver "
                         "workgroups best: 4 x 2036 = 8144 at the headline config)")
    ap.add_argument("--vit-batch", type=int, default=0,
                    help="images per ViT forward (0 = ~1.8 M token rows, sized so the token matrix fills whole waves of "
                         "workgroups: vit.wave_filling_batch; 2036 for dino_vits16 at 480x480)")
    ap.add_argument("--dataset", type=int, default=0,
                    help="strong scaling: a fixed set of this many images sharded round-robin over the ranks "
                         "(BASELINE.json configs[3]: 10000); --steps is then derived from the shard")
    ap.add_argument("--model", default="dino_vits16")
    ap.add_argument("--size", type=int, default=480)
    ap.add_argument("--K", type=int, default=5)
    ap.add_argument("--dtype", default="float16", choices=["float16", "bfloat16"])
    ap.add_argument("--w-dtype", default="u16", choices=["u16", "f32"],
                    help="storage of the affinity matrix the eigensolver streams: u16 = round(65535 w) fixed point "
                         "(default; the arithmetic stays fp32), f32 = floats")
    ap.add_argument("--companion-steps", type=int, default=2,
                    help="N=1 only: after the main measurement, time this many steps with the OTHER --w-dtype and report "
                         "them as value_w_<dtype> (0 = skip)")
    ap.add_argument("--cpu-images", type=int, default=6, help="images of the same workload timed on the CPU oracle")
    ap.add_argument("--parity-images", type=int, default=32,
                    help="images whose GPU eigenvectors are checked against the CPU oracle (the first --cpu-images of them "
                         "are the timed CPU baseline; 0 = only those)")
    ap.add_argument("--host-pin", choices=("register", "malloc", "shm"), default="malloc",
                    help="how the host image pool / result buffers are page-locked (see page_lock)")
    ap.add_argument("--distinct", type=int, default=1000,
                    help="distinct synthetic images generated per rank (BASELINE.json configs[1]: 1k synthetic images)")
    ap.add_argument("--dino-like-steps", type=int, default=1,
                    help="N=1 only: after the main measurement, time this many steps with synthetic.dino_like_state_dict "
                         "weights (outlier channels, sharp logits: the spectrum a trained DINO produces is harder for the "
                         "eigensolver than random-weight features) and report value_dino_like_weights + its passes per image")
    ap.add_argument("--resident", action="store_true",
                    help="diagnostic: keep the images resident in HBM (no H2D in the timed region; NOT the reported mode)")
    ap.add_argument("--min-warmup-seconds", type=float, default=4.0,
                    help="keep running untimed warm-up steps (beyond --warmup) until this much wall time has passed: "
                         "the first GPU process on a fresh box runs ~20 %% slower until clocks/power have ramped")
    ap.add_argument("--timer-sample", type=int, default=5,
                    help="HIP-event pairs around every N-th launch of a kernel name inside the timed region (the few-launch kernels "
                         "are always timed; odd, so that the alternating qkv / fc1 launches of one kernel name are both sampled); 1 = every launch (the events then cost ~10 %% of a step), 0 = no kernel timers")
    ap.add_argument("--affinity", default="fused", choices=["fused", "split", "fp32"],
                    help="affinity build (spectral.laplacian_eigs_from_features affinity_mode); 'fused' is the pipeline's")
    ap.add_argument("--gemm-tuning", default="table", choices=["table", "online", "off"],
                    help="DinoViT(gemm_tuning=...): shipped TunableOp table / also tune new shapes on line / leave TunableOp alone")
    ap.add_argument("--linear-kres", type=int, default=2, choices=[0, 1, 2],
                    help="DinoViT(linear_kres=...): 0 library GEMMs only, 1 K-resident qkv/proj, 2 (default) + fc1+GELU")
    ap.add_argument("--no-fuse-ln", action="store_true",
                    help="A/B arm: standalone residual + LayerNorm passes instead of the fused dss_lnlinear_* prologue")
    ap.add_argument("--no-fuse-pe", action="store_true",
                    help="A/B arm: patchify + library GEMM + position-embedding add instead of the one dss_patch_embed_p16 kernel")
    ap.add_argument("--no-fuse-k", action="store_true",
                    help="A/B arm: the hooked block's K projection as LayerNorm + library GEMM + dss_kfeatures_finalize "
                         "instead of the one dss_lnlinear_kfeatures_k384 kernel")
    ap.add_argument("--no-fuse-qkv768", action="store_true",
                    help="A/B arm (D = 768 models): norm1 -> qkv as LayerNorm + library GEMM instead of one dss_lnlinear_k768 launch")
    ap.add_argument("--library-gemm", default="lt", choices=["lt", "torch", "torch-streamk"],
                    help="who issues the Linear layers that are not hand-written kernels: lt (default) = dss_linear_lt (hipBLASLt, Stream-K "
                         "split switched off and verified off); torch = F.linear (rounds 1-5; the package import still sets Tensile's "
                         "data-parallel switch); torch-streamk = F.linear with that switch REMOVED from the environment before the first "
                         "GEMM - rounds 1-5 exactly, not reproducible at D = 768 (A/B arm only)")
    ap.add_argument("--fc2-into-stream", action="store_true",
                    help="A/B arm: fc2 adds its fp32 accumulators to the fp32 residual stream inside the GEMM's epilogue (dss_linear_lt_accumulate) "
                         "instead of leaving a half-precision tensor for the next LayerNorm prologue to add (the default; rounds 4-6)")
    ap.add_argument("--gelu", default="auto", choices=["auto", "erf", "erf_f16", "tanh_fused"],
                    help="auto (default, = DinoViT's) = erf_f16 for the D = 384 models, erf for D = 768; "
                         "erf_f16 = DINO's erf-GELU as a polynomial form on packed f16 in fc1's epilogue (f16 "
                         "operands; error budget: tests/test_host_logic.py::test_gelu_f16_poly_error_budget; bf16 falls back to erf); "
                         "erf = the same function evaluated in fp32 (the A/B arm: +3.9 %% step time); "
                         "tanh_fused = hipBLASLt epilogue (tanh approximation, NOT the reference function; diagnostic only)")
    ap.add_argument("--vit-streams", type=int, default=1,
                    help="run the ViT forwards of consecutive sub-batches on this many alternating streams (A/B arm.  2 measured "
                         "+1.7 %% in round 2 - one forward's kernels filled the other's tails - and -11 %% with round 6's kernels "
                         "(12 859 / 12 771 against 14 367 / 14 323 images/s, one box, alternating): the forwards are whole rounds "
                         "of workgroups now and two of them only compete for the same CUs' LDS and HBM; the per-kernel "
                         "HIP-event durations behind `roofline` also overlap and read long)")
    ap.add_argument("--shard-forwards", type=int, default=4,
                    help="a step that is not whole forwards of --vit-batch images (a rank's shard) is cut into at least this many forwards (chunk_counts)")
    ap.add_argument("--balanced-chunks", action="store_true",
                    help="A/B arm: cut a shard step into equal forwards (round 4: 1250 = 4 x 313) instead of whole rounds of "
                         "workgroups with a one-round lead forward")
    ap.add_argument("--tail-overlap", default="auto", choices=["auto", "on", "off"],
                    help="spectral stage of all forwards but the last on a side stream under the last forward (auto: only for a "
                         "one-step --dataset shard, where the step's tail is exposed; in the steady state it measured slower)")
    ap.add_argument("--overlap", action="store_true",
                    help="run the spectral stage of sub-batch i on a side stream under the ViT of sub-batch i+1 "
                         "(measured slower on MI355X: both stages are bandwidth-bound; default off)")
    return ap.parse_args()


def spawn_ranks_if_needed(a):
    """``python bench.py --gpus N`` on its own: re-execute under torch.distributed.run with N ranks on this node."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    print(f"[bench] --gpus {a.gpus} without a launcher: spawning {a.gpus} ranks ({' '.join(cmd[1:9])} ...)", file=sys.stderr)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def baseline_config_name(a, world):
    """Which entry of BASELINE.json `configs` this run's workload is (the line must not claim another one's)."""
    if a.model == "dino_vits16" and a.size == 480 and a.K == 5:
        if a.dataset == 10000:
            return "BASELINE.json configs[3]" + ("" if world == 8 else f", its {world}-GPU leg")
        if a.dataset == 0:
            return "BASELINE.json configs[1]" if world == 1 else f"BASELINE.json configs[1] per GPU, {world} GPUs weak-scaled"
    if a.model == "dino_vitb8" and a.size == 480 and a.K == 15 and a.dataset == 0:
        return "BASELINE.json configs[2]" if world == 1 else f"BASELINE.json configs[2] per GPU, {world} GPUs weak-scaled"
    if a.model == "dino_vits16" and a.size == 224 and a.K == 5:
        return "the shape of BASELINE.json configs[0] (its 16-image CPU run is the cpu_baseline leg's territory)"
    if a.model == "dino_vitb8" and a.K == 20:
        return "BASELINE.json configs[4] at ONE image size (the mixed 320-640 px set runs through the CLI tests)"
    return "not a BASELINE.json config"


_OVERLAP = {}
_STREAMS = {}


def step(model, imgs, K, vit_batch, overlap=False, nstreams=1, w_dtype="u16", mode="fused"):
    """One pass of the hot path over one batch: features + eigs for every image (``mode``: the affinity build)."""
    if overlap:  # spectral stage of sub-batch i on a side stream under the ViT of sub-batch i+1
        key = (id(model), K, vit_batch, mode)
        if key not in _OVERLAP:
            _OVERLAP[key] = pipeline.OverlappedExtractor(model, K, vit_batch, affinity_mode=mode)
        _, ev, vec, info = _OVERLAP[key](imgs)
        return ev, vec, info
    if nstreams > 1:  # ViT forwards of consecutive sub-batches on alternating streams (opt-in, see --vit-streams)
        cur = torch.cuda.current_stream()
        pool = _STREAMS.setdefault(nstreams, [torch.cuda.Stream() for _ in range(nstreams)])
        ks = []
        for j, s in enumerate(range(0, imgs.shape[0], vit_batch)):
            st = pool[j % nstreams]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                ks.append(model.extract_k(imgs[s:s + vit_batch]))
        for st in pool:
            cur.wait_stream(st)
    else:
        if mode == "fused" and w_dtype == "u16":
            # the K projection hands over fp32 features (what extract_features would save), their f16 copy and the
            # inverse norms in one pass (hip.kfeatures_finalize); the affinity build starts from the f16 rows
            parts = [model.extract_k_f16(imgs[s:s + vit_batch]) for s in range(0, imgs.shape[0], vit_batch)]
            k, k16, rn = (torch.cat([p_[i] for p_ in parts]) if len(parts) > 1 else parts[0][i] for i in range(3))
            return spectral.laplacian_eigs_from_features(k, K, strict=False, retry=False, w_dtype=w_dtype,
                                                         affinity_mode=mode, feats16=k16, rnorm=rn)
        ks = [model.extract_k(imgs[s:s + vit_batch]) for s in range(0, imgs.shape[0], vit_batch)]
    k = torch.cat(ks) if len(ks) > 1 else ks[0]
    # strict=False, retry=False: no device->host sync inside the step (convergence is checked once, after timing)
    return spectral.laplacian_eigs_from_features(k, K, strict=False, retry=False, w_dtype=w_dtype, affinity_mode=mode)


class ImageFeeder:
    """The u8 images travel host -> HBM inside the timed region, at the granularity of ONE ViT forward: a pinned host pool
    of ``n_distinct`` images, a ring of device buffers of ``chunk`` images, a copy stream.  ``prefetch(c, count)`` enqueues
    the copy of global chunk ``c`` (a contiguous run of the pool, so the PCIe volume is the full chunk, with no host-side
    assembly); ``get(c, count)`` makes the compute stream wait for it; ``release(c)`` marks the buffer free once the
    forward that reads it has been enqueued.  The copy of forward i + 1 runs under forward i, and a run (or a rank's
    shard) starts computing after its FIRST forward's images have arrived, not after a whole step's."""

    NBUF = 3

    def __init__(self, host_pool: torch.Tensor, chunk: int, dev, resident: bool = False):
        self.pool, self.chunk, self.n = host_pool, chunk, host_pool.shape[0]
        self.resident = resident
        if resident:
            self.dev_pool = host_pool.to(dev)
            return
        self.bufs = [torch.empty((chunk, *host_pool.shape[1:]), dtype=torch.uint8, device=dev) for _ in range(self.NBUF)]
        self.stream = torch.cuda.Stream(device=dev)
        self.ready = [torch.cuda.Event(enable_timing=True) for _ in range(self.NBUF)]
        self.began = [torch.cuda.Event(enable_timing=True) for _ in range(self.NBUF)]   # (timeline probe: see run_steps)
        self.log = []                                                                    # (chunk, images, began, ready) per copy
        self.free = [None] * self.NBUF

    def prefetch(self, c: int, count: int = 0):
        if self.resident:
            return
        b = c % self.NBUF
        count = count or self.chunk
        with torch.cuda.stream(self.stream):
            if self.free[b] is not None:
                self.stream.wait_event(self.free[b])
            self.began[b] = torch.cuda.Event(enable_timing=True)
            self.ready[b] = torch.cuda.Event(enable_timing=True)
            self.began[b].record(self.stream)
            j, off = 0, (c * self.chunk) % self.n
            while j < count:
                run = min(self.n - off, count - j)
                self.bufs[b][j:j + run].copy_(self.pool[off:off + run], non_blocking=True)
                j, off = j + run, (off + run) % self.n
            self.ready[b].record(self.stream)
            if len(self.log) < 64:
                self.log.append((c, count, self.began[b], self.ready[b]))

    def get(self, c: int, count: int = 0):
        count = count or self.chunk
        if self.resident:
            idx = (torch.arange(count, device=self.dev_pool.device) + c * self.chunk) % self.n
            return self.dev_pool[idx]
        torch.cuda.current_stream().wait_event(self.ready[c % self.NBUF])
        return self.bufs[c % self.NBUF][:count]

    def release(self, c: int):
        if self.resident:
            return
        ev = torch.cuda.Event()
        ev.record()
        self.free[c % self.NBUF] = ev


def page_lock(t: torch.Tensor, how: str = "malloc") -> torch.Tensor:
    """Page-locks a host tensor for the copy engine.  "malloc" (default): torch's own pinned allocation
    (`tensor.pin_memory()`, hipHostMalloc).  "register": `hipHostRegister` on the tensor's ordinary pages.  Measured on
    the driver's boxes, both ways round: SMALL copies (0.69 MB) out of registered /dev/shm pages take 0.024 ms against
    1.33 ms out of hipHostMalloc memory (scripts/debug/shm_bench.py - what the CLI's worker path uses), but the bench's
    48-200 MB chunk copies out of a registered numpy-allocated pool ran SLOWER (224 x 224: 13.7 k vs 34.3 k images/s;
    480 x 480: 10 971 vs 10 935), and so did a registered /dev/shm segment ("shm": 13.4 k vs 36.3 k; 11 118 vs ~10 950) -
    so the pool stays on hipHostMalloc."""
    if how == "malloc":
        return t.pin_memory()
    if how == "shm":      # a /dev/shm segment (what the CLI's worker path copies out of), then registered
        t = t.contiguous().clone().share_memory_()
    t = t.contiguous()
    rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel() * t.element_size(), 0)
    if int(rc) != 0 or not t.is_pinned():
        raise SystemExit(f"[bench] hipHostRegister failed ({rc}); run with --host-pin malloc")
    return t


ROUND_IMAGES = 0.0   # images whose token rows fill ONE round of the K-resident Linear kernel's workgroups (set in main)


SHARD_FORWARDS = 4    # forwards a one-step shard is cut into at least, while they stay above 256 images (--shard-forwards)


def chunk_counts(cnt: int, vit_batch: int, lead: bool = False, round_images: float = -1.0):
    """Images per ViT forward of a step of ``cnt`` images.  A step that is whole forwards of ``vit_batch`` images (the
    steady state: ``vit_batch`` is sized to whole rounds of workgroups by ``vit.wave_filling_batch``) runs them as they
    are.  Any other step - a rank's shard of the 8-GPU run: 1250 images = ONE step - is cut so that
      * every forward but the last is a whole number of ROUNDS of the Linear kernels' workgroups (``round_images`` images
        fill the 2 x CUs workgroup slots once; round 4 cut 1250 into 4 x 313 = 2.15 rounds each - three rounds of time
        for 2.15 of work in 51 % of the step's kernels: `lnlinear` 0.213 of peak against 0.239 in the steady state);
      * there are at least four forwards while they stay above 256 images (the copy of forward j + 1 runs under forward
        j; a one-forward step would wait for all of its bytes before computing anything);
      * ``lead``: the FIRST forward of a run is short - nothing hides its H2D copy - and takes the step's fractional round
        (1250 images = 8.59 rounds: 88 + 436 + 436 + 290 = 0.6 + 3 + 3 + 2 rounds, the first copy 61 MB instead of 216 MB;
        a partly filled round costs a whole round's time wherever it is, and at the front it is also the shortest copy)."""
    rnd = ROUND_IMAGES if round_images < 0 else round_images
    if cnt % vit_batch == 0 and not (lead and rnd > 0 and cnt == vit_batch):
        return [vit_batch] * (cnt // vit_batch)
    n = max(1, -(-cnt // vit_batch), min(SHARD_FORWARDS, cnt // 256))
    if rnd <= 0 or cnt < 2 * rnd or n == 1:
        per, extra = divmod(cnt, n)
        return [per + (1 if i < extra else 0) for i in range(n)]
    total = cnt / rnd
    whole, frac = math.floor(total), total - math.floor(total)
    if lead:      # the lead forward takes what the whole rounds leave: `frac` of a round, or 1 + frac if that would be a sliver
        body, m = (whole if frac >= 1.0 / 3.0 else whole - 1), max(1, n - 1)
    else:
        body, m = math.ceil(total), n
    m = min(max(m, -(-body // max(1, math.floor(vit_batch / rnd)))), body)     # <= vit_batch images and >= one round per forward
    rounds = [body // m + (1 if i < body % m else 0) for i in range(m)]
    sizes = [math.floor(r * rnd) for r in rounds]
    if lead:
        return [cnt - sum(sizes)] + sizes
    sizes[-1] = cnt - sum(sizes[:-1])
    return sizes


_TAIL = {}


def step_fed(model, feeder, c0, cnt, nxt, K, vit_batch, w_dtype="u16", mode="fused", lead=False, tail_overlap=False, emit=None):
    """One step whose images arrive through the feeder: forward j reads global chunk ``c0 + j``; before it is enqueued the
    copy of the NEXT chunk (this step's, or ``nxt`` = (chunk id, count) of the following step's first) is put on the copy
    stream.  ``tail_overlap`` (a rank's one-step shard): the spectral stage of every forward but the last runs on a side
    stream UNDER the last forward, so that only the last forward's images are left for the exposed tail of the step.
    ``emit(first image, eigenvalues, eigenvectors, info)`` is called for every batch of results as soon as its kernels are
    enqueued, on the stream they run on (the caller packs them and starts their D2H copy there: with ``tail_overlap`` the
    results of all forwards but the last travel to the host under the last forward too).
    Returns (eigenvalues, eigenvectors, info, chunks consumed)."""
    counts = chunk_counts(cnt, vit_batch, lead)
    f16 = mode == "fused" and w_dtype == "u16"
    parts, bufs, s0 = [], None, 0
    early = None
    for j, n in enumerate(counts):
        if j + 1 < len(counts):
            feeder.prefetch(c0 + j + 1, counts[j + 1])
        elif nxt is not None:
            feeder.prefetch(*nxt)
        imgs = feeder.get(c0 + j, n)
        if f16:
            # the forwards' hand-over kernels write straight into the step's three buffers (fp32 features, f16 copy,
            # inverse norms): no concatenation pass over 2.8 + 1.4 GB in front of the affinity build
            if bufs is None:
                npatch, dim = (imgs.shape[1] // model.patch_size) * (imgs.shape[2] // model.patch_size), model.embed_dim
                bufs = (torch.empty((cnt, npatch, dim), dtype=torch.float32, device=imgs.device),
                        torch.empty((cnt, npatch, dim), dtype=torch.float16, device=imgs.device),
                        torch.empty((cnt, npatch), dtype=torch.float32, device=imgs.device))
            model.extract_k_f16(imgs, out=tuple(b_[s0:s0 + n] for b_ in bufs))
            s0 += n
            if tail_overlap and len(counts) > 1 and j == len(counts) - 2:
                # everything up to here (all forwards but the last) goes to the spectral stage NOW, on the side stream
                main = torch.cuda.current_stream()
                side = _TAIL.setdefault(imgs.device, torch.cuda.Stream(device=imgs.device))
                ready = torch.cuda.Event()
                ready.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    early = (s0, spectral.laplacian_eigs_from_features(bufs[0][:s0], K, strict=False, retry=False, w_dtype=w_dtype,
                                                                       affinity_mode=mode, feats16=bufs[1][:s0], rnorm=bufs[2][:s0]))
                    if emit is not None:
                        emit(0, *early[1])
        else:
            parts.append(model.extract_k(imgs))
        feeder.release(c0 + j)
    if f16 and early is not None:
        e0, first = early
        last = spectral.laplacian_eigs_from_features(bufs[0][e0:], K, strict=False, retry=False, w_dtype=w_dtype,
                                                     affinity_mode=mode, feats16=bufs[1][e0:], rnorm=bufs[2][e0:])
        if emit is not None:
            emit(e0, *last)
        main = torch.cuda.current_stream()
        main.wait_stream(_TAIL[bufs[0].device])
        for t in first:
            t.record_stream(main)
        out = tuple(torch.cat((a_, b_)) for a_, b_ in zip(first, last))
        return (*out, len(counts))
    if f16:
        out = spectral.laplacian_eigs_from_features(bufs[0], K, strict=False, retry=False, w_dtype=w_dtype,
                                                    affinity_mode=mode, feats16=bufs[1], rnorm=bufs[2])
    else:
        k = torch.cat(parts) if len(parts) > 1 else parts[0]
        out = spectral.laplacian_eigs_from_features(k, K, strict=False, retry=False, w_dtype=w_dtype, affinity_mode=mode)
    if emit is not None:
        emit(0, *out)
    return (*out, len(counts))


def summarize_timers(timers, n_patches, dim, depth_attn, affinity_mode="fused"):
    """Average HIP-event duration per launch and the algorithmic work per launch."""
    out = {}
    launches = timers.get("_launches", {})
    for name, recs in timers.items():
        if name == "_launches":
            continue
        ms = [s.elapsed_time(e) for s, e, _ in recs]
        if not ms:
            continue
        avg = float(np.mean(ms))
        n_all = int(launches.get(name, len(ms)))     # every launch is counted, every --timer-sample-th one is timed
        entry = {"launches": n_all, "timed_launches": len(ms), "total_ms": round(avg * n_all, 3), "avg_ms": round(avg, 4)}
        metas = [m for _, _, m in recs]
        if name == "laplacian_eigs":
            n = metas[0]["n"]
            passes = [float(m["info"].abs().sum().item()) for m in metas]
            wb = metas[0].get("w_bytes", 4)                  # 4 (float W) or 2 (16-bit fixed-point W)
            byts = np.mean(passes) * wb * n * (n + 1) / 2    # symmetric W: w_bytes*N(N+1)/2 algorithmic bytes per pass
            entry.update(bound="hbm", achieved=byts / (avg * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                         passes_per_image=float(np.sum(passes) / sum(m["b"] for m in metas)))
        elif name == "attention":
            m = metas[0]
            flops = 4.0 * m["t"] ** 2 * m["heads"] * 64 * m["b"]
            entry.update(bound="mfma", achieved=flops / (avg * 1e-3) / 1e12, peak=MFMA16_PEAK_TF, unit="TFLOP/s")
        elif name == "affinity":
            m = metas[0]
            if m.get("f16_in"):  # f16 features + inverse norms in, packed 16-bit W out: 2ND + 4N + N(N+1) algorithmic bytes
                byts = (2.0 * m["n"] * m["d"] + 4.0 * m["n"] + 1.0 * m["n"] * (m["n"] + 1)) * m["b"]
                entry.update(bound="hbm", achieved=byts / (avg * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
            elif m.get("fused"):   # one kernel, raw f32 features in, packed 16-bit W out: 4ND + N(N+1) algorithmic bytes
                byts = (4.0 * m["n"] * m["d"] + 1.0 * m["n"] * (m["n"] + 1)) * m["b"]
                entry.update(bound="hbm", achieved=byts / (avg * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
            elif affinity_mode == "fp32":   # exact fp32 MFMA build: MFMA-bound
                flops = 1.0 * m["n"] * (m["n"] + 1) * m["d"] * m["b"]   # upper triangle: N(N+1)/2 dots of 2D flop
                entry.update(bound="mfma", achieved=flops / (avg * 1e-3) / 1e12, peak=MFMA32_PEAK_TF, unit="TFLOP/s")
            else:  # split-f16 build (normalise + Gram): HBM-bound; 4ND in + 4ND split write/read + w_bytes*N(N+1)/2 out
                byts = (4.0 * m["n"] * m["d"] + m.get("w_bytes", 4) / 2.0 * m["n"] * (m["n"] + 1)) * m["b"]
                entry.update(bound="hbm", achieved=byts / (avg * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        elif name in ("linear_kres", "lnlinear", "lnlinear_kfeatures", "library_gemm", "patch_embed"):
            # both pipes matter: 2*M*N*K flop on the matrix cores and M*N*2 output bytes (1.5 - 4x the input); the fused
            # residual + LayerNorm + Linear kernel additionally reads x f32 + the pending branch output and writes x back
            flops = np.mean([2.0 * m["m"] * m["n"] * m["k"] for m in metas])
            outb = np.mean([2.0 * m["m"] * m["n"] for m in metas])
            entry.update(bound="mfma", achieved=flops / (avg * 1e-3) / 1e12, peak=MFMA16_PEAK_TF, unit="TFLOP/s",
                         output_GBs=round(outb / (avg * 1e-3) / 1e9, 1))
            if name == "lnlinear":
                # the fused kernel sits BELOW the f16 ridge (2.5 PF / 8 TB/s = 312 FLOP/B; it moves x f32 in and out, the pending
                # branch output and its own output: ~158 FLOP/B): its roofline is min(MFMA peak, arithmetic intensity x HBM peak) =
                # the HBM side, so `bound` / `achieved` / `frac` are the bytes; the MFMA side stays in the entry (`frac_mfma`, the
                # number rounds 1-4 reported as `frac`) - achieved / attainable FLOP rate equals `frac` by construction
                hbm = np.mean([m["m"] * m["k"] * (10.0 if m["res"] else 4.0) + 2.0 * m["m"] * m["n"] for m in metas])
                ai = flops / hbm
                tfs = flops / (avg * 1e-3) / 1e12
                if ai * HBM_PEAK_GBS / 1e3 < MFMA16_PEAK_TF:
                    entry.update(bound="hbm", achieved=hbm / (avg * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
                entry.update(flop_per_byte=round(ai, 1), attainable_TFs=round(min(MFMA16_PEAK_TF, ai * HBM_PEAK_GBS / 1e3), 1),
                             mfma_TFs=round(tfs, 1), frac_mfma=round(tfs / MFMA16_PEAK_TF, 4),
                             # (`frac` changed meaning in round 5 - both views now carry their own, unambiguous key)
                             hbm_GBs=round(hbm / (avg * 1e-3) / 1e9, 1), frac_hbm=round(hbm / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            elif name == "lnlinear_kfeatures":
                # HBM-bound: x f32 (+ the pending branch output f16) in, fp32 + f16 features and the norms out (CLS rows dropped)
                hbm = np.mean([m["m"] * m["k"] * (6.0 if m["res"] else 4.0) + (m["m"] - m["m"] // m["t"]) * (6.0 * m["n"] + 4.0) for m in metas])
                entry.update(bound="hbm", achieved=hbm / (avg * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                             mfma_TFs=round(flops / (avg * 1e-3) / 1e12, 1))
        elif name == "layernorm":
            byts = np.mean([m["rows"] * m["d"] * (4 + m["out_bytes"] + (6 if m["res"] else 0)) for m in metas])
            entry.update(bound="hbm", achieved=byts / (avg * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        if "achieved" in entry:
            entry["achieved"] = round(entry["achieved"], 2)
            entry["frac"] = round(entry["achieved"] / entry["peak"], 4)
        out[name] = entry
    return out


def pmc_traffic(kernel, a):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary whose workload matches
    this run (profiles/rNN_pmc_traffic.json, FETCH_SIZE already doubled per the gfx950 correction); else None."""
    best = None
    for f in sorted((REPO / "profiles").glob("r*_pmc_traffic.json")):
        try:
            d = json.loads(f.read_text())
        except Exception:
            continue
        c = d.get("config", {})
        if (c.get("model"), c.get("size"), c.get("K"), c.get("batch"), c.get("vit_batch")) == \
                (a.model, a.size, a.K, a.batch, a.vit_batch) and kernel in d.get("kernels", {}):
            best = (d["kernels"][kernel]["hbm_bytes_per_launch"], f"profiles/{f.name}")
    return best


def cpu_quota_cores():
    """CPUs the container may actually use at once (cgroup v2 `cpu.max` / v1 `cpu.cfs_quota_us`), or None if unlimited: the
    GPU boxes show 256 cores to `nproc` under a quota of 16 - what `cores` (the threads the oracle ran) could really draw on."""
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        return None if quota == "max" else round(int(quota) / int(period), 2)
    except Exception:
        pass
    try:
        q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
        return None if q <= 0 else round(q / int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text()), 2)
    except Exception:
        return None


def cpu_baseline(model_name, sd, size, K, n_images, gpu_vecs, gpu_vals, lam_tol, n_parity=0):
    """The oracle (CPU restatement of the reference path) on the same synthetic images/weights, all host
    cores; also yields the eigenvector parity of the GPU results on those images: the rule of tests/util.check_eigs
    (isolated eigenvalues: 1 - |cos| per vector; eigenvalues closer than 1e-4: D-weighted principal angle of the
    cluster's span; every cluster is bounded by 1e-4 and reported)."""
    from oracle import spectral_ref, vit_ref
    from tests.util import build_w64, check_eigs

    ref = vit_ref.build_ref_vit(model_name, sd)
    # the reference's CPU path on the cores this container can really use: under a cgroup quota (16 CPUs on the GPU boxes, where
    # torch defaults to 128 threads and numpy's BLAS to 256) more threads than CPUs only add contention - torch's intra-op pool
    # and the BLAS / OpenMP pools behind numpy and scipy are capped at the quota
    quota = cpu_quota_cores()
    limiter = None
    if quota:
        torch.set_num_threads(max(1, math.ceil(quota)))
        try:
            from threadpoolctl import threadpool_limits
            limiter = threadpool_limits(limits=max(1, math.ceil(quota)))
        except Exception:
            limiter = None
    cores = torch.get_num_threads()
    times, worst_vec, worst_cluster, ok, clusters, draws = [], 0.0, 0.0, True, [], []
    n_total = max(n_images + 1, n_parity)
    for i in range(n_total):  # image 0 is the warm-up of the timed baseline
        img = synthetic.synthetic_image(i, size, size)
        t0 = time.perf_counter()
        k = vit_ref.ref_extract_k(ref, vit_ref.ref_preprocess(img))
        if i <= n_images:
            lam, vec = spectral_ref.ref_laplacian_eigs(k, K)   # the timed CPU path: exactly the reference's op sequence
            dt = time.perf_counter() - t0
            if i > 0:
                times.append(dt)
        # untimed: a validated draw of the same call + fp64 extra pairs to decide clusters at the edge of the K wanted
        lam, vec, ext, ndraw = spectral_ref.ref_laplacian_eigs_ext(k, K)
        draws.append(int(ndraw))
        report = []
        try:
            ce = check_eigs(gpu_vecs[i].cpu().numpy(), gpu_vals[i].cpu().numpy(), vec.numpy(), lam.numpy(),
                            what=f"img{i}", lam_tol=lam_tol, d=build_w64(k[0].numpy())[1], ext=ext, report=report)
            worst_vec = max(worst_vec, float(ce.max()))
        except AssertionError as e:
            ok = False
            print(f"[bench] parity failure: {e}", file=sys.stderr)
        worst_cluster = max([worst_cluster] + [c["err"] for c in report])
        lam_x = np.asarray(ext[0], np.float64)   # fp64 eigenvalues 0 .. K + E - 1 of the same problem
        clusters.append([{**c, "err": float(f"{c['err']:.3g}"), "max_vector_cos_err": float(f"{c['max_vector_cos_err']:.3g}"),
                          # how far the chained cluster reaches in eigenvalue (a tight pair vs a long chain of 1e-4 gaps)
                          "cluster_span": float(f"{float(lam_x[min(c['last'], len(lam_x) - 1)] - lam_x[c['first']]):.3g}")}
                         for c in report if c["kind"] != "isolated"])
    return ({"value": round(len(times) / sum(times), 3), "unit": "images/s", "cores": cores, "cpu_quota_cores": cpu_quota_cores(),
             "kind": "port",
             "sample": f"{len(times)} of the same {size}x{size} synthetic images, torch-CPU fp32 ViT + "
                       f"numpy/scipy eigsh (oracle/), 1 warm-up image excluded"},
            {"rule": "tests/util.check_eigs: every cluster of eigenvalues (gaps < 1e-4) bounded by 1e-4; isolated -> "
                     "1-|cos| per vector, cluster -> D-weighted principal angle of the spans",
             "max_cluster_err_vs_cpu": worst_cluster, "max_vector_cos_err_vs_cpu": worst_vec,
             "all_within_1e-4": ok, "eigenvalue_tol": lam_tol, "images": n_total,
             # reference draws per image (oracle/spectral_ref.ref_laplacian_eigs_ext): 1 = the reference's first ARPACK run
             # was the target; k > 1 = k - 1 runs were > 1e-5 from fp64 and re-drawn; negative = every draw was bad and
             # the fp64 solution itself is the target
             "reference_draws": draws, "images_on_fp64_target": int(sum(1 for d_ in draws if d_ < 0)),
             "non_isolated_clusters_per_image": clusters})


def run_steps(model, feeder, counts, a, rank, world, n_patches, w_dtype, first_chunk=0):
    """The timed region for this rank: ``len(counts)`` steps (``counts[s]`` images each), the u8 images copied H2D one ViT
    forward ahead of their use, every step's results streamed to pinned host memory, one collection on rank 0 at the end.
    Returns (elapsed seconds, host seconds inside the loop, info tensors, gathered (meta, payload) or None, chunks used)."""
    dev = model.device
    width = a.K * n_patches + a.K
    # (a ring of four step-sized blocks: the CLI's savers would have written a block out long before its fourth successor lands -
    #  one block per step would be 5 GB of page-locked memory at 20 steps of 14 838 images)
    n_ring = min(len(counts), 4)
    host_out = page_lock(torch.empty((n_ring, max(counts), width), dtype=torch.float32), a.host_pin)
    copy_stream = torch.cuda.Stream(device=dev)
    infos, metas, flats, d2h_log = [], [], [], []
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_start = torch.cuda.Event(enable_timing=True)
    ev_start.record()
    if not feeder.resident:
        feeder.log.clear()
    chunk = first_chunk
    lead0 = a.dataset > 0            # a rank's shard starts cold: its first forward is one round (see chunk_counts)
    feeder.prefetch(chunk, chunk_counts(counts[0], a.vit_batch, lead0)[0])
    base = 0
    for s, cnt in enumerate(counts):
        lead = lead0 and s == 0
        nchunks = len(chunk_counts(cnt, a.vit_batch, lead))
        nxt = (chunk + nchunks, chunk_counts(counts[s + 1], a.vit_batch)[0]) if s + 1 < len(counts) else None
        def emit(start, ev, vec, info, s=s, base=base):
            # one batch of results (a whole step, or with --tail-overlap its two parts), on the stream that produced it: pack,
            # then every rank streams ITS OWN [K, N] results to pinned host memory while the GPU goes on computing
            n = ev.shape[0]
            ids = (torch.arange(n, device=dev, dtype=torch.int64) + (base + start)) * world + rank   # global round-robin item ids
            meta, flat = distributed.pack_records(ids, ev, vec)
            infos.append(info), metas.append(meta), flats.append(flat)
            done = torch.cuda.Event()
            done.record()
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done)
                b_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                b_.record(copy_stream)
                host_out[s % n_ring, start:start + n].copy_(flat.view(n, width), non_blocking=True)
                e_.record(copy_stream)
                flat.record_stream(copy_stream)
                d2h_log.append((s, start, n, b_, e_))

        if a.overlap or a.vit_streams > 1:   # the opt-in variants take a whole step's images at once
            cc, parts = chunk_counts(cnt, a.vit_batch, lead), []
            for j, n in enumerate(cc):        # ring of NBUF buffers: the copy of chunk j + 1 goes out, chunk j is taken
                if j + 1 < nchunks:           # (cloned: its buffer is handed back before the step reads it), then released
                    feeder.prefetch(chunk + j + 1, cc[j + 1])
                elif nxt is not None:
                    feeder.prefetch(*nxt)
                parts.append(feeder.get(chunk + j, n).clone())
                feeder.release(chunk + j)
            ev, vec, info = step(model, torch.cat(parts) if len(parts) > 1 else parts[0], a.K, a.vit_batch, a.overlap,
                                 a.vit_streams, w_dtype, a.affinity)
            emit(0, ev, vec, info)
        else:
            step_fed(model, feeder, chunk, cnt, nxt, a.K, a.vit_batch, w_dtype, a.affinity, lead=lead, emit=emit,
                     tail_overlap=a.tail_overlap == "on" or (a.tail_overlap == "auto" and a.dataset > 0 and len(counts) == 1))
        chunk += nchunks
        base += cnt
    ev_compute = torch.cuda.Event(enable_timing=True)
    ev_compute.record()                        # the last kernel of the last step (its D2H and the collection follow)
    host_enqueue_s = time.perf_counter() - t0  # host finished enqueueing; the GPU may still be running
    # the ONE collection of the run: sizes, then every rank's flat payload point to point to rank 0 (RCCL over xGMI)
    gathered = distributed.gather_records_to_root(torch.cat(metas), torch.cat(flats))
    t_gather = time.perf_counter() - t0
    copy_stream.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tdev = dev if torch.distributed.get_backend() != "gloo" else torch.device("cpu")
        tmax = torch.tensor([elapsed], device=tdev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # where the time outside the kernels goes (GPU clock): each H2D copy of the feeder [chunk, images, start, end] relative to
    # the start of the region, and the end of the last kernel - what follows it is result D2H + collection + synchronisation
    run_steps.timeline = {"compute_done_ms": round(ev_start.elapsed_time(ev_compute), 3), "elapsed_ms": round(elapsed * 1e3, 3),
                          "host_enqueued_ms": round(host_enqueue_s * 1e3, 3), "collection_returned_ms": round(t_gather * 1e3, 3),
                          # the result copies [step, first image, images, start, end] (the last ones of the run)
                          "d2h_copies": [[s_, st_, n_, round(ev_start.elapsed_time(b_), 3), round(ev_start.elapsed_time(e_), 3)]
                                         for s_, st_, n_, b_, e_ in d2h_log[-3:]],
                          "h2d_copies": [] if feeder.resident else
                          [[c_, n_, round(ev_start.elapsed_time(b_), 3), round(ev_start.elapsed_time(r_), 3)] for c_, n_, b_, r_ in feeder.log[:12]]}
    return elapsed, host_enqueue_s, infos, gathered, chunk


def main():
    a = parse()
    spawn_ranks_if_needed(a)
    rank, world = distributed.init_process_group()
    if world != a.gpus:
        raise SystemExit(f"[bench] --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} "
                         f"(or run `python bench.py --gpus {a.gpus}` alone: it spawns its own ranks)")
    backend = torch.distributed.get_backend() if world > 1 else None
    if world > 1 and backend != "nccl" and not os.environ.get("DSS_DIST_BACKEND"):
        raise SystemExit(f"[bench] multi-GPU runs must use the nccl (= RCCL) backend, got {backend}")
    dev = distributed.local_device()
    ranks_seen = [(rank, dev.index)]
    if world > 1:
        seen = [None] * world
        torch.distributed.all_gather_object(seen, (rank, dev.index))
        ranks_seen = sorted(seen)
    dtype = {"float16": torch.float16, "bfloat16": torch.bfloat16}[a.dtype]
    dim, depth, heads, patch = synthetic.VIT_CONFIGS[a.model]
    sd = synthetic.synthetic_state_dict(a.model, 0)
    if a.library_gemm == "torch-streamk":
        os.environ.pop("TENSILE_STREAMK_DATA_PARALLEL", None)
    lib_gemm = "lt" if a.library_gemm == "lt" else "torch"
    model = DinoViT(a.model, sd, dev, dtype, gelu=a.gelu, library_gemm=lib_gemm, fc2_into_stream=a.fc2_into_stream, linear_kres=a.linear_kres, fuse_ln=not a.no_fuse_ln,
                    gemm_tuning=a.gemm_tuning, fuse_k=not a.no_fuse_k, fuse_pe=not a.no_fuse_pe, fuse_qkv768=not a.no_fuse_qkv768)
    n_patches = (a.size // patch) ** 2
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    if a.vit_batch <= 0:
        from dss_amd.vit import wave_filling_batch
        rows = hip.LINEAR_KRES_WIDTHS.get(model.embed_dim, (None, 0))[1]
        # ~0.9 M token rows per forward (1018 images at the headline config): measured round 4 on one box, images/s with
        # 290 / 435 / 580 / 1160 images per forward = 12 213 / 12 430 / 12 635 / 12 754 - per-launch tails and the ~7 us between
        # launches are paid per forward, HBM (288 GB) is nowhere near a limit (a forward's activations: < 15 GB)
        # Round 5: ~1.8 M rows (2036 images: 14 rounds of workgroups) - 1018 -> 2036 images per forward measured +1.5 % on one box
        # (13 946 / 13 921 -> 14 149 images/s); bounded by the 32-bit row arithmetic of the hand-over kernel (M * D * 4 < 2^32)
        row_cap = int(0.9 * 2 ** 32 / max(4 * model.embed_dim, n_patches + 1))   # (... and M * T < 2^32: its row -> image division)
        target = max(8, round(min(2048 * 901, row_cap) / (n_patches + 1)))
        a.vit_batch = wave_filling_batch(n_patches + 1, target, rows_per_workgroup=rows) if rows else target
        while a.vit_batch * (n_patches + 1) > row_cap:      # (wave_filling_batch may round up to 25 % above the target)
            a.vit_batch -= 1
    global ROUND_IMAGES, SHARD_FORWARDS
    rows_cu = hip.LINEAR_KRES_WIDTHS.get(model.embed_dim, (None, 0))[1]
    ROUND_IMAGES = ncu * rows_cu / (n_patches + 1) if rows_cu and a.linear_kres and not a.balanced_chunks else 0.0   # see chunk_counts
    SHARD_FORWARDS = max(1, a.shard_forwards)
    if a.batch <= 0:
        # the eigensolver runs one workgroup per image, two per CU: pick the number of ViT forwards per step (4..8: the
        # copy of forward j + 1 hides under forward j) whose image count best fills whole rounds of 2 x CUs workgroups
        fill = lambda m: (m * a.vit_batch / (2 * ncu)) / math.ceil(m * a.vit_batch / (2 * ncu))
        a.batch = max(range(4, 9), key=lambda m: (round(fill(m), 2), -m)) * a.vit_batch
    if a.dataset > 0:   # strong scaling: this rank's shard of a fixed set, in equal steps of at most the default batch
        shard = len(distributed.shard_indices(a.dataset, rank, world))
        n_steps = max(1, math.ceil(shard / a.batch))
        per = math.ceil(shard / n_steps)
        counts = [min(per, shard - i * per) for i in range(n_steps)]
        counts = [c for c in counts if c > 0]
        a.batch = max(counts)
    else:
        counts = [a.batch] * a.steps

    # synthetic images in PINNED HOST memory (rank r owns global indices r, r+world, ...)
    n_distinct = min(a.distinct, a.batch * (len(counts) + a.warmup))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as ex:   # numpy releases the GIL: ~40 ms/image serial
        host = torch.from_numpy(np.stack(list(ex.map(lambda i: synthetic.synthetic_image(rank + world * i, a.size, a.size),
                                                     range(n_distinct)))))
    host = page_lock(host, a.host_pin)
    feeder = ImageFeeder(host, a.vit_batch, dev, resident=a.resident)
    chunk_pos = 0                        # global chunk counter: every warm-up / timed step consumes its own chunks

    def warm_step(m, w_dtype, lead=False):
        """One untimed step through the same feeder path as the timed region (``lead``: cut like a run's first step)."""
        nonlocal chunk_pos
        cc = chunk_counts(a.batch, a.vit_batch, lead)
        feeder.prefetch(chunk_pos, cc[0])
        if a.overlap or a.vit_streams > 1:
            parts = []
            for j, n in enumerate(cc):
                parts.append(feeder.get(chunk_pos + j, n).clone())
                feeder.release(chunk_pos + j)
                if j + 1 < len(cc):
                    feeder.prefetch(chunk_pos + j + 1, cc[j + 1])
            out = step(m, torch.cat(parts), a.K, a.vit_batch, a.overlap, a.vit_streams, w_dtype, a.affinity)
            chunk_pos += len(cc)
            return out
        ev, vec, info, used = step_fed(m, feeder, chunk_pos, a.batch, None, a.K, a.vit_batch, w_dtype, a.affinity, lead=lead,
                                       tail_overlap=lead and a.tail_overlap != "off" and len(counts) == 1)
        chunk_pos += used
        return ev, vec, info

    from dss_amd.vit import setup_gemm_tuning
    tune = a.gemm_tuning != "off"             # "off": scripts/tune_gemm.sh drives TunableOp through PyTorch's own variables
    setup_gemm_tuning(tune_new_shapes=True, use_table=tune)   # warm-up may pick GEMM solutions for shapes missing from the table
    t_warm = time.perf_counter()
    n_warm, warm = 0, None
    while n_warm < a.warmup or time.perf_counter() - t_warm < a.min_warmup_seconds:
        # a shard run's first step is cut differently (chunk_counts `lead`): warm (and GEMM-tune) both sets of shapes
        warm = warm_step(model, a.w_dtype, lead=a.dataset > 0 and n_warm % 2 == 0)
        torch.cuda.synchronize()
        n_warm += 1
    setup_gemm_tuning(tune_new_shapes=False, use_table=tune)  # frozen for the timed region
    if warm is None:   # --warmup 0: the collection warm-up below still needs one step's results
        warm = warm_step(model, a.w_dtype)
    # warm the COLLECTION path too: RCCL sets up its point-to-point channels (one per peer) on first use - seconds, not part
    # of any steady state - and with ONE rank the ordering of the records is the first use of a sort / scan / gather kernel
    # and of their workspace allocations (measured: 7.4 ms behind the last kernel of a 1250-image shard, timeline probe) - so
    # one full-size collection of a warm-up step's results runs before the clock
    ids = torch.arange(warm[0].shape[0], device=dev, dtype=torch.int64) * world + rank
    distributed.gather_records_to_root(*distributed.pack_records(ids, warm[0], warm[1]))
    del ids
    del warm
    torch.cuda.synchronize()

    # what the HOST needs to enqueue one step (all launches of the ViT forwards + affinity + eigensolver) when the queue
    # is empty and nothing blocks it: the ceiling the launch path alone would impose
    warm_imgs = host[: min(a.batch, host.shape[0])].to(dev)
    if warm_imgs.shape[0] < a.batch:
        warm_imgs = warm_imgs.repeat((a.batch + warm_imgs.shape[0] - 1) // warm_imgs.shape[0], 1, 1, 1)[: a.batch]
    host_only_ms = float("inf")
    for _ in range(2):    # the first call may still grow the allocator's pools (hipMalloc is synchronous): take the second
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        step(model, warm_imgs, a.K, a.vit_batch, a.overlap, a.vit_streams, a.w_dtype, a.affinity)
        host_only_ms = min(host_only_ms, (time.perf_counter() - t_h) * 1e3)
    del warm_imgs
    torch.cuda.synchronize()

    hip.TIMERS = {} if a.timer_sample > 0 else None
    hip.TIMER_SAMPLE = max(a.timer_sample, 1)
    elapsed, host_enqueue_s, infos, gathered, chunk_pos = run_steps(model, feeder, counts, a, rank, world, n_patches,
                                                                    a.w_dtype, first_chunk=chunk_pos)
    timers, hip.TIMERS = (hip.TIMERS or {}), None
    timeline = getattr(run_steps, "timeline", None)
    n_images = sum(counts)
    if world > 1:
        tot = torch.tensor([n_images], device=dev, dtype=torch.int64)
        torch.distributed.all_reduce(tot)
        n_images = int(tot.item())
    if gathered is not None:
        assert gathered[0].shape[0] == n_images, (gathered[0].shape, n_images)
        ids = gathered[0][:, 0]
        assert ids.dtype == torch.int64 and bool((ids[1:] > ids[:-1]).all())   # every item exactly once, ordered

    info_all = torch.cat(infos)
    n_unconverged = int((info_all <= 0).sum().item())
    if rank == 0:
        kern = summarize_timers(timers, n_patches, dim, depth, a.affinity)
        cands = [k for k in kern if "achieved" in kern[k] and k != "library_gemm"]
        n_forwards = sum(len(chunk_counts(c, a.vit_batch, a.dataset > 0 and i == 0)) for i, c in enumerate(counts))
        steps_out = len(counts)
        roofline = None                      # --timer-sample 0: no kernel was timed, nothing to report
        if cands:
            dominant = max(cands, key=lambda k: kern[k]["total_ms"])
            d = kern[dominant]
            traffic = pmc_traffic(dominant, a)  # HBM bytes per launch from the committed rocprofv3 PMC passes, or None
            roofline = {"kernel": dominant, "bound": d["bound"], "achieved": d["achieved"], "peak": d["peak"],
                        "unit": d["unit"], "frac": d["frac"],
                        # (the fused norm -> Linear kernel: arithmetic intensity below the ridge - both sides of its roofline)
                        **({k_: d[k_] for k_ in ("flop_per_byte", "attainable_TFs", "mfma_TFs", "frac_mfma", "hbm_GBs", "frac_hbm") if k_ in d}),
                        "traffic": traffic[0] if traffic else None,
                        "traffic_unit": "bytes/launch (2*FETCH_SIZE+WRITE_SIZE)*1024", "traffic_source": traffic[1] if traffic else None,
                        "timed": f"HIP events around every {max(a.timer_sample, 1)}-th launch inside the timed region"}
        lib_sites = {}                       # hipBLASLt time per step by call site (events around every --timer-sample-th launch)
        for st, en, meta in timers.get("library_gemm", []):
            lib_sites.setdefault(meta.get("what", "?"), []).append(st.elapsed_time(en))
        n_lib_timed = max(sum(len(v) for v in lib_sites.values()), 1)
        n_lib_all = timers.get("_launches", {}).get("library_gemm", n_lib_timed)
        lib_sites = {k_: round(float(np.sum(v)) * n_lib_all / n_lib_timed / steps_out, 3) for k_, v in lib_sites.items()}
        out = {
            "metric": "images/sec end-to-end (features+eigs) at 480², K=5; eigvec cos-err vs CPU",
            "value": round(n_images / elapsed, 2), "unit": "images/s",
            "n_gpus": world, "steps": steps_out, "warmup": a.warmup, "warmup_steps_run": n_warm,
            "ms_per_step": round(elapsed / steps_out * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if a.dataset > 0 else "weak",
            "vs_baseline": None, "dtype": "f16" if dtype == torch.float16 else "bf16", "data": "synthetic",
            "config": {"workload": f"{a.model} {a.size}x{a.size} K={a.K}, {a.batch} images/step/GPU, one B=1 result "
                                   f"per image ({baseline_config_name(a, world)})" +
                                   (f"; fixed set of {a.dataset} images round-robin over {world} GPU(s)"
                                    if a.dataset > 0 else ""),
                       "distinct_images": n_distinct,
                       "images_per_step": a.batch, "images_total": n_images,
                       "vit_batch": a.vit_batch, "patches": n_patches, "weights": "synthetic trunc_normal(0.02) seed 0",
                       "vit_operands": "f16" if dtype == torch.float16 else "bf16", "accumulate": "fp32",
                       "affinity": {"fused": "f16 features + inverse norms from the K projection's hand-over kernel -> "
                                             "f16-operand Gram (fp32 accumulate, 256x128 tiles, LDS-DMA panels) -> u16 W"
                                             if a.w_dtype == "u16" else "split-f16 (hi+lo f16 terms, fp32 accumulate)",
                                    "split": "split-f16 (hi+lo f16 terms, fp32 accumulate)",
                                    "fp32": "exact fp32 MFMA"}[a.affinity],
                       "w_dtype": "u16-fixed (round(65535 w))" if a.w_dtype == "u16" else "f32",
                       "eig_arithmetic": "f32 Lanczos, f64 Rayleigh-Ritz",
                       "h2d_in_timed_region": not a.resident, "host_page_lock": {"register": "hipHostRegister", "shm": "hipHostRegister on a /dev/shm segment", "malloc": "hipHostMalloc (tensor.pin_memory)"}[a.host_pin],
                       "parallelism": f"dp{world} round-robin, 1 collection (sizes + flat payload, p2p)",
                       "stage_overlap": a.overlap, "gelu": model.gelu, "library_gemm": a.library_gemm, "fc2_into_stream": model.fc2_into_stream},
            "ranks_seen": len(ranks_seen), "rank_devices": ranks_seen, "backend": backend, "timeline": timeline,
            "roofline": roofline, "kernels": kern, "unconverged_images": n_unconverged,
            # what is NOT a hand-written kernel, and what the fused prologue removed: standalone LayerNorm launches per ViT
            # forward (23 before round 4; 1 = the last block's norm1 in front of the K projection), hipBLASLt time per step
            "layernorm_launches_per_forward": round(kern.get("layernorm", {}).get("launches", 0) / max(n_forwards, 1), 2),
            "library_gemm_ms_per_step": round(kern.get("library_gemm", {}).get("total_ms", 0.0) / steps_out, 3),
            # what each layer of THIS model actually ran on (the switches only apply where a kernel exists for the width /
            # patch size: round 4's line claimed fuse_k / fuse_pe for dino_vitb8, where neither path exists)
            "vit_paths": {"switches": {"linear_kres": a.linear_kres, "fuse_ln": not a.no_fuse_ln, "fuse_k": not a.no_fuse_k,
                                       "fuse_pe": not a.no_fuse_pe, "fuse_qkv768": not a.no_fuse_qkv768}, **model.paths()},
            "library_gemm_ms_per_step_by_site": lib_sites,
            # time until the host had enqueued a step's launches INSIDE the timed loop: it includes the waits of the
            # double-buffered image feeder on the GPU (back-pressure), not only CPU work ...
            "host_in_loop_ms_per_step": round(host_enqueue_s / steps_out * 1e3, 3),
            # ... the CPU work alone: one full step enqueued on an EMPTY queue (no sync inside, timers off)
            "host_enqueue_ms_per_step": round(host_only_ms, 3),
        }
    if world == 1 and a.companion_steps > 0 and a.dataset == 0:
        other = "f32" if a.w_dtype == "u16" else "u16"
        for i in range(2):   # warm the other storage's kernels / allocations
            warm_step(model, other)
        torch.cuda.synchronize()
        e2, _, _, _, chunk_pos = run_steps(model, feeder, [a.batch] * a.companion_steps, a, rank, world, n_patches, other,
                                           first_chunk=chunk_pos)
        out[f"value_w_{other}"] = round(a.companion_steps * a.batch / e2, 2)
    if world == 1 and a.companion_steps > 0 and a.dataset == 0 and model.gelu == "erf_f16":
        # the headline with DINO's exact erf-GELU in fp32 arithmetic instead of the packed-f16 polynomial form (both numbers
        # on the line; `extract_features --gelu erf` is this model)
        me = DinoViT(a.model, sd, dev, dtype, gelu="erf", library_gemm=lib_gemm, fc2_into_stream=a.fc2_into_stream, linear_kres=a.linear_kres, fuse_ln=not a.no_fuse_ln, gemm_tuning=a.gemm_tuning,
                     fuse_k=not a.no_fuse_k, fuse_pe=not a.no_fuse_pe, fuse_qkv768=not a.no_fuse_qkv768)
        for i in range(2):
            warm_step(me, a.w_dtype)
        torch.cuda.synchronize()
        e4, _, _, _, chunk_pos = run_steps(me, feeder, [a.batch] * a.companion_steps, a, rank, world, n_patches, a.w_dtype,
                                           first_chunk=chunk_pos)
        out["value_gelu_erf"] = round(a.companion_steps * a.batch / e4, 2)
        del me
    if world == 1 and a.dino_like_steps > 0 and a.dataset == 0:
        # the same workload with weights shaped like a trained DINO's (no checkpoint can be downloaded here): the ViT
        # costs the same, the eigensolver sees a harder spectrum - how much of the headline survives it
        dl = DinoViT(a.model, synthetic.dino_like_state_dict(a.model, 0), dev, dtype, gelu=a.gelu, library_gemm=lib_gemm, fc2_into_stream=a.fc2_into_stream, linear_kres=a.linear_kres,
                     fuse_ln=not a.no_fuse_ln, gemm_tuning=a.gemm_tuning, fuse_k=not a.no_fuse_k, fuse_pe=not a.no_fuse_pe,
                     fuse_qkv768=not a.no_fuse_qkv768)
        for i in range(2):
            warm_step(dl, a.w_dtype)
        torch.cuda.synchronize()
        e3, _, inf3, _, chunk_pos = run_steps(dl, feeder, [a.batch] * a.dino_like_steps, a, rank, world, n_patches,
                                              a.w_dtype, first_chunk=chunk_pos)
        inf3 = torch.cat(inf3)
        out["value_dino_like_weights"] = round(a.dino_like_steps * a.batch / e3, 2)
        out["passes_per_image_dino_like_weights"] = round(float(inf3.abs().float().mean().item()), 2)
        out["unconverged_images_dino_like_weights"] = int((inf3 <= 0).sum().item())
        del dl
    if rank == 0:
        if world == 1 and a.cpu_images > 0:
            n_par = max(a.cpu_images + 1, a.parity_images)
            first = step(model, host[:n_par].to(dev), a.K, a.vit_batch, a.overlap, a.vit_streams, a.w_dtype, a.affinity)
            out["cpu_baseline"], out["parity"] = cpu_baseline(a.model, sd, a.size, a.K, a.cpu_images, first[1], first[0],
                                                              lam_tol=1e-3 if dtype == torch.float16 else 1e-2,
                                                              n_parity=a.parity_images)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
