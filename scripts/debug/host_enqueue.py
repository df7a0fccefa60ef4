"""How long does the HOST need to enqueue one ViT forward (all launches), with the GPU queue empty?

    python scripts/debug/host_enqueue.py [vit_batch]
Prints the enqueue time (perf_counter around the forward call, no sync inside), the GPU time of the same forward, with
per-kernel event timers off and on, and eager vs hipGraph replay (torch.cuda.CUDAGraph capture of the same call)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd  # noqa
from dss_amd import hip, synthetic
from dss_amd.vit import DinoViT

vb = int(sys.argv[1]) if len(sys.argv) > 1 else 290
dev = torch.device("cuda", 0)
model = DinoViT("dino_vits16", synthetic.synthetic_state_dict("dino_vits16", 0), dev, torch.float16)
img = torch.from_numpy(np.stack([synthetic.synthetic_image(i, 480, 480) for i in range(4)])).to(dev)
img = img.repeat((vb + 3) // 4, 1, 1, 1)[:vb].contiguous()
for _ in range(3):
    k = model.extract_k(img)
torch.cuda.synchronize()

def run(label, timers):
    hip.TIMERS = {} if timers else None
    enq, tot = [], []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.extract_k(img)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    n = sum(len(v) for k_, v in hip.TIMERS.items() if k_ != "_launches") if timers else 0
    hip.TIMERS = None
    print(f"{label}: host enqueue of one {vb}-image forward {min(enq):.2f} ms (median {sorted(enq)[2]:.2f}), "
          f"forward incl. sync {min(tot):.2f} ms; timed launches {n}", flush=True)

run("eager, timers off", False)
run("eager, timers on ", True)
# back-to-back forwards (queue back-pressure visible): 7 forwards, then sync
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(7):
    model.extract_k(img)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"7 forwards back to back: host returned after {(t1-t0)*1e3:.1f} ms, GPU done after {(t2-t0)*1e3:.1f} ms", flush=True)
# hipGraph capture
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model.extract_k(img)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        kg = model.extract_k(img)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    ke = model.extract_k(img)
    torch.cuda.synchronize()
    print(f"graph replay vs eager: max |dk| = {(kg - ke).abs().max().item():.3e}", flush=True)
    enq, tot = [], []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.replay()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    print(f"hipGraph replay: host {min(enq):.2f} ms, forward incl. sync {min(tot):.2f} ms", flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(7):
        g.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"7 graph replays back to back: host returned after {(t1-t0)*1e3:.1f} ms, GPU done after {(t2-t0)*1e3:.1f} ms", flush=True)
except Exception as e:  # report, do not die: this is a probe
    print("hipGraph capture failed:", repr(e), flush=True)
