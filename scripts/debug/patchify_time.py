"""Kernel-only timing of dss_preprocess_patchify at the bench's forward size (1018 images of 480x480, patch 16, f16 out)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dss_amd
from dss_amd import hip
B = int(os.environ.get("VIT_BATCH", 1018))
img = torch.randint(0, 256, (B, 480, 480, 3), dtype=torch.uint8, device="cuda")
for dt in (torch.float16, torch.float32):
    for _ in range(3): out = hip.preprocess_patchify(img, 16, dt)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(10): out = hip.preprocess_patchify(img, 16, dt)
    en.record(); torch.cuda.synchronize()
    ms = st.elapsed_time(en) / 10
    gb = (img.numel() + out.numel() * out.element_size()) / 1e9
    print(f"patchify {B} x 480x480 -> {dt}: {ms*1e3:.0f} us, {gb/ms:.2f} TB/s")
