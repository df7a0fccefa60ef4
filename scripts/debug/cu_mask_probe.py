"""Probe: two ViT forwards running CONCURRENTLY on two HIP streams, each confined to half of the compute units
(hipExtStreamCreateWithCUMask), against the same forwards back to back on the whole chip.  Question: do one forward's
HBM-bound phases (LayerNorm prologues, store tails) overlap the other's MFMA phases once neither can fill the chip alone?
Also checks that the mask is honoured at all (a forward alone on a half-chip stream should take ~2x)."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd
from dss_amd import synthetic
from dss_amd.vit import DinoViT

hipr = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = ctypes.c_void_p()
    rc = hipr.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

dev = torch.device("cuda", 0)
B = int(os.environ.get("VIT_BATCH", 290))
model = DinoViT("dino_vits16", synthetic.synthetic_state_dict("dino_vits16", 0), dev, torch.float16)
imgs = [torch.randint(0, 255, (B, 480, 480, 3), dtype=torch.uint8, device=dev) for _ in range(4)]
def run_seq(stream=None, n=4):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        return [model.extract_k(imgs[i % 4]) for i in range(n)]
def wall(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3
t_full = wall(lambda: run_seq(None, 4))
print(f"4 forwards of {B} images back to back, whole chip: {t_full:.1f} ms ({t_full / 4:.2f} per forward)")
ALL = (1 << 256) - 1
masks = {"low/high halves": ((1 << 128) - 1, ALL ^ ((1 << 128) - 1)),
         "even/odd CUs": (int("01" * 128, 2), int("10" * 128, 2)),
         "XCD pairs (bits mod 8 < 4 / >= 4)": (sum(1 << i for i in range(256) if i % 8 < 4), sum(1 << i for i in range(256) if i % 8 >= 4))}
for name, (ma, mb) in masks.items():
    sa, sb = masked_stream(ma), masked_stream(mb)
    main = torch.cuda.current_stream()
    t_half = wall(lambda: (sa.wait_stream(main), run_seq(sa, 2), main.wait_stream(sa)))
    def both():
        sa.wait_stream(main); sb.wait_stream(main)
        for i in range(2):
            with torch.cuda.stream(sa): a = model.extract_k(imgs[2 * i])
            with torch.cuda.stream(sb): b = model.extract_k(imgs[2 * i + 1])
        main.wait_stream(sa); main.wait_stream(sb)
    t_both = wall(both)
    print(f"{name:36s}: 2 forwards alone on one half-chip stream {t_half:.1f} ms ({t_half / 2:.2f} per forward; mask honoured if ~2x {t_full / 4:.2f});"
          f" 4 forwards on two half-chip streams {t_both:.1f} ms = {100 * (t_both / t_full - 1):+.1f} % vs back to back")
# three streams, thirds
m3 = [sum(1 << i for i in range(256) if (i // 8) % 3 == r) for r in range(3)]
ss = [masked_stream(m) for m in m3]
main = torch.cuda.current_stream()
def three():
    for s in ss: s.wait_stream(main)
    for i in range(6):
        with torch.cuda.stream(ss[i % 3]): model.extract_k(imgs[i % 4])
    for s in ss: main.wait_stream(s)
t3 = wall(three)
t6 = wall(lambda: run_seq(None, 6))
print(f"6 forwards on three third-chip streams {t3:.1f} ms vs back to back {t6:.1f} ms = {100 * (t3 / t6 - 1):+.1f} %")
# unmasked two plain streams for reference
p1, p2 = torch.cuda.Stream(), torch.cuda.Stream()
def plain():
    p1.wait_stream(main); p2.wait_stream(main)
    for i in range(2):
        with torch.cuda.stream(p1): model.extract_k(imgs[2 * i])
        with torch.cuda.stream(p2): model.extract_k(imgs[2 * i + 1])
    main.wait_stream(p1); main.wait_stream(p2)
print(f"4 forwards on two UNMASKED streams {wall(plain):.1f} ms")
