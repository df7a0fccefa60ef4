"""GPU (-m gpu): parity of every HIP kernel, called through the C ABI (dss_amd.hip -> libdss_hip.so),
against the CPU oracle / fp64 references on seeded inputs, plus the reference goldens."""
import glob
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import dss_amd  # noqa: F401
from dss_amd import hip, spectral, synthetic
from oracle import spectral_ref, vit_ref
from tests.util import build_w64, check_eigs, d_orthonormality, golden_case, golden_ext, oracle_target

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent
DEV = "cuda"


def test_library_is_the_hip_build():
    lib = hip.load_library()
    assert lib.dss_target_arch() == b"gfx950"
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


# ----------------------------------------------------------------------------- image transform
@pytest.mark.parametrize("h,w", [(224, 224), (75, 64), (100, 130), (480, 480), (33, 47)])
def test_preprocess_chw_bit_exact(h, w):
    imgs = np.stack([synthetic.synthetic_image(i, h, w) for i in range(2)])
    out = hip.preprocess_chw(torch.from_numpy(imgs).to(DEV)).cpu()
    for i in range(2):
        assert torch.equal(out[i], vit_ref.ref_preprocess(imgs[i]))


@pytest.mark.parametrize("h,w,p", [(224, 224, 16), (75, 64, 16), (100, 130, 8), (480, 480, 16), (50, 70, 16)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_preprocess_patchify_bit_exact(h, w, p, dtype):
    imgs = np.stack([synthetic.synthetic_image(10 + i, h, w) for i in range(2)])
    out = hip.preprocess_patchify(torch.from_numpy(imgs).to(DEV), p, dtype).cpu()
    hp, wp = h // p, w // p
    for i in range(2):
        x = vit_ref.ref_preprocess(imgs[i])[:, : hp * p, : wp * p]  # crop (extract.py:82-88)
        ref = x.reshape(3, hp, p, wp, p).permute(1, 3, 0, 2, 4).reshape(hp * wp, 3 * p * p)
        assert torch.equal(out[i], ref.to(dtype))


# ----------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,d", [(901, 384), (3601, 768), (7, 64), (130, 1024), (5, 2048), (1, 4)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_layernorm_matches_fp32_reference(rows, d, out_dtype):
    g = torch.Generator().manual_seed(rows * 7 + d)
    x = torch.randn(rows, d, generator=g) * 3 + 0.5
    gamma, beta = torch.randn(d, generator=g), torch.randn(d, generator=g)
    ref = F.layer_norm(x.double(), (d,), gamma.double(), beta.double(), 1e-6)
    xd = x.to(DEV)
    out = hip.layernorm(xd, gamma.to(DEV), beta.to(DEV), 1e-6, out_dtype).cpu()
    assert torch.equal(xd.cpu(), x)  # untouched without a residual
    tol = {torch.float32: 2e-5, torch.float16: 4e-3, torch.bfloat16: 3e-2}[out_dtype]
    assert (out.double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    if out_dtype == torch.float32:  # the plain-PyTorch fp32 op of the same definition
        assert (out - F.layer_norm(x, (d,), gamma, beta, 1e-6)).abs().max().item() < 2e-5


@pytest.mark.parametrize("res_dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_layernorm_fused_residual(res_dtype):
    rows, d = 333, 384
    g = torch.Generator().manual_seed(5)
    x, r = torch.randn(rows, d, generator=g), torch.randn(rows, d, generator=g).to(res_dtype)
    gamma, beta = torch.randn(d, generator=g), torch.randn(d, generator=g)
    xd = x.to(DEV)
    out = hip.layernorm(xd, gamma.to(DEV), beta.to(DEV), 1e-6, torch.float16, residual=r.to(DEV)).cpu()
    xsum = x + r.float()
    assert torch.equal(xd.cpu(), xsum)  # residual stream updated in place, one fp32 rounding
    ref = F.layer_norm(xsum.double(), (d,), gamma.double(), beta.double(), 1e-6)
    assert (out.double() - ref).abs().max().item() <= 4e-3 * ref.abs().max().item()


def _to_planar(t2d):
    """row-major [rows, D] -> DSS_PLANAR64 [D/64, rows, 64]"""
    rows, d = t2d.shape
    return t2d.reshape(rows, d // 64, 64).permute(1, 0, 2).contiguous()


@pytest.mark.parametrize("res_dtype", [torch.float16, torch.bfloat16])
def test_layernorm_planar_residual_is_bit_identical_to_row_major(res_dtype):
    rows, d = 1803, 384
    g = torch.Generator().manual_seed(9)
    x, r = torch.randn(rows, d, generator=g), torch.randn(rows, d, generator=g).to(res_dtype)
    gamma, beta = torch.randn(d, generator=g).to(DEV), torch.randn(d, generator=g).to(DEV)
    xa, xb = x.to(DEV), x.to(DEV)
    a = hip.layernorm(xa, gamma, beta, 1e-6, torch.float16, residual=r.to(DEV))
    b = hip.layernorm(xb, gamma, beta, 1e-6, torch.float16, residual=_to_planar(r).to(DEV), residual_planar=True)
    assert torch.equal(a, b) and torch.equal(xa, xb)
    assert torch.equal(xa.cpu(), x + r.float())


# ----------------------------------------------------------------------------- Linear with K = 384 (qkv / proj / fc1)
@pytest.mark.parametrize("m,n,k", [(901, 1152, 384), (512, 384, 384), (1, 64, 384), (77, 1536, 384), (1025, 128, 384),
                                   (2 * 3601 + 5, 384, 384), (3601, 2304, 768), (256, 768, 768), (1, 64, 768),
                                   (300, 3072, 768), (513, 128, 768)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("gelu", [False, True, 2])      # 2: the packed-f16 GELU (f16 operands only; same bars as the fp32 form)
@pytest.mark.parametrize("planar", [False, True])
def test_linear_kres_matches_fp32_reference(m, n, k, dtype, gelu, planar):
    """The plain PyTorch fp32 op of the same definition (F.linear, then DINO's exact erf GELU) on the same rounded
    operands; tolerance = the output rounding of the half dtype (fp32 accumulation inside).  Shapes cover ragged
    M (not a multiple of the workgroup or the wave tile) and every (N, K) the ViT-S and ViT-B models use."""
    if gelu == 2 and dtype != torch.float16:
        pytest.skip("gelu = 2 is the packed-f16 form")
    g = torch.Generator().manual_seed(m + n + k)
    x = torch.randn(m, k, generator=g).to(dtype)
    w = (torch.randn(n, k, generator=g) * 0.05 * (384 / k) ** 0.5).to(dtype)
    b = (torch.randn(n, generator=g) * 0.2).to(dtype)
    ref = F.linear(x.float(), w.float(), b.float())
    if gelu:
        ref = F.gelu(ref)
    guard = 4                                                   # rows after the output must stay untouched
    out = hip.linear_kres(x.to(DEV), w.to(DEV), b.to(DEV), gelu=gelu, planar=planar)
    out = (hip.planar_to_rows(out) if planar else out).float().cpu()
    assert out.shape == (m, n) and guard
    tol = (1.0e-3 if dtype == torch.float16 else 8e-3) * max(1.0, ref.abs().max().item())
    assert (out - ref).abs().max().item() <= tol
    if gelu:  # the negative tail is where an erf approximation would show: absolute error well under f16 spacing
        neg = ref < -0.05
        if neg.any():
            assert (out - ref)[neg].abs().max().item() <= (5e-4 if dtype == torch.float16 else 4e-3)


@pytest.mark.parametrize("k", [384, 768])
def test_gelu_f16_epilogue_is_the_documented_arithmetic_bit_for_bit(k):
    """`gelu = 2`: with identity weights the accumulator of output (m, n) IS the f16 input A[m][n], so the kernel's output must
    equal tests.util.gelu_f16_poly (csrc/kres.h restated in numpy, every f16 operation rounding once) on EVERY finite f16 value -
    the error budget the CPU test states for that arithmetic (tests/test_host_logic.py::test_gelu_f16_poly_error_budget) is
    then the kernel's.  Also: gelu = 2 is refused for bf16 operands."""
    from tests.util import gelu_f16_poly

    allh = np.arange(0, 65536, dtype=np.uint16).view(np.float16)
    vals = allh[np.isfinite(allh)]
    m = -(-vals.size // k)
    a = np.zeros(m * k, dtype=np.float16)
    a[:vals.size] = vals
    a = torch.from_numpy(a.reshape(m, k))
    w = torch.eye(k, dtype=torch.float16)
    out = hip.linear_kres(a.to(DEV), w.to(DEV), torch.zeros(k, dtype=torch.float16, device=DEV), gelu=2).cpu().numpy()
    with np.errstate(over="ignore"):
        want = gelu_f16_poly(a.numpy())
    assert np.array_equal(out, want), int((out != want).sum())    # (values: -0 == +0; for every other finite f16 value == is bit equality)
    with pytest.raises(hip.HipLibraryError):
        hip.linear_kres(a.to(DEV).bfloat16(), w.to(DEV).bfloat16(), torch.zeros(k, dtype=torch.bfloat16, device=DEV), gelu=2)


# ----------------------------------------------------------------------------- library GEMMs behind the ABI (round 6)
@pytest.mark.parametrize("m,n,k,bias,out32", [(901 * 8, 384, 1536, True, False), (3601 * 4, 768, 3072, True, False), (3600 * 3, 768, 192, True, False),
                                             (3601 * 2, 768, 768, True, False), (3601 * 2, 2304, 768, True, False), (901 * 3, 384, 384, False, True),
                                             (1, 384, 1536, True, False), (77, 768, 768, False, True),
                                             (901 * 80, 384, 1536, True, False)])         # (from 65 536 rows: the measured tile preference)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_linear_lt_matches_fp64_and_torch(m, n, k, bias, out32, dtype):
    """dss_linear_lt (hipBLASLt with a data-parallel algorithm chosen by the library): x W^T + b against fp64 on the rounded operands
    (fp32 accumulation, one rounding of the output) and against what PyTorch's own route gives for the same call."""
    g = torch.Generator().manual_seed(m + n + k)
    x = torch.randn(m, k, generator=g).to(dtype)
    w = (torch.randn(n, k, generator=g) * 0.05).to(dtype)
    b = (torch.randn(n, generator=g) * 0.1).to(dtype) if bias else None
    out = hip.linear_lt(x.to(DEV), w.to(DEV), None if b is None else b.to(DEV), out_dtype=torch.float32 if out32 else None)
    assert out.dtype == (torch.float32 if out32 else dtype) and tuple(out.shape) == (m, n)
    ref = x.double() @ w.double().t() + (0 if b is None else b.double())
    scale = max(1.0, ref.abs().max().item())
    tol = (2e-5 if out32 else (1e-3 if dtype == torch.float16 else 8e-3)) * scale
    assert (out.cpu().double() - ref).abs().max().item() <= tol
    if not out32:
        assert (out - F.linear(x.to(DEV), w.to(DEV), None if b is None else b.to(DEV))).abs().max().item() <= 2 * tol


@pytest.mark.parametrize("m,n,k", [(901 * 8, 384, 1536), (3601 * 4, 768, 3072), (77, 384, 1536)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_linear_lt_accumulate_adds_into_the_fp32_stream(m, n, k, dtype):
    """dss_linear_lt_accumulate: X (f32) += A W^T + b in place - DINO Block's `x = x + mlp(...)` inside fc2's own epilogue.  Against
    fp64 on the rounded operands: the branch is added from the fp32 accumulators, so the error is fp32 accumulation error only (no
    rounding of the branch output to the operand type: the bar is 50x tighter than dss_linear_lt's half-precision output)."""
    g = torch.Generator().manual_seed(m + n)
    a = torch.randn(m, k, generator=g).to(dtype)
    w = (torch.randn(n, k, generator=g) * 0.05).to(dtype)
    b = torch.randn(n, generator=g) * 0.1           # fp32, like the stream
    x0 = torch.randn(m, n, generator=g) * 3.0
    x = x0.clone().to(DEV)
    out = hip.linear_lt_accumulate(a.to(DEV), w.to(DEV), b.to(DEV), x)
    assert out.data_ptr() == x.data_ptr()
    ref = x0.double() + a.double() @ w.double().t() + b.double()
    assert (x.cpu().double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # and it is what the two-step form gives up to the half-precision rounding of the branch
    two = x0.to(DEV) + hip.linear_lt(a.to(DEV), w.to(DEV), b.to(dtype).to(DEV)).float()
    assert (x - two).abs().max().item() <= (1e-3 if dtype == torch.float16 else 8e-3) * max(1.0, (ref - x0.double()).abs().max().item())


def test_linear_lt_only_takes_candidates_without_a_partial_tile_workspace():
    """The reason the wrapper exists: every gfx950 kernel of this stack's hipBLASLt is Stream-K-capable (`_SK3_` in every solution
    name) - it splits the last, partly filled round of output tiles across workgroups through a workspace, and that exchange is
    not reproducible (profiles/r06_forward_stress.txt).  With Tensile's data-parallel switch set for the process (package import /
    `dss_linear_lt` itself) the library reports NO workspace for any candidate, and `dss_linear_lt` only takes a candidate for
    which that is the case (without the switch it reports 30-64 MiB - profiles/r06_lt_describe.txt - and the wrapper refuses to
    run).  For every library GEMM shape of the two bench configurations: exactly one candidate taken, workspace 0, no
    single-buffer split-K.  Where a measured preference exists (mlp.fc2 at D = 384 from 65 536 rows: the 192 x 128 tile,
    profiles/r06_lt_tune.txt) the taken line is marked `*p` and carries that tile - IF the library offers it among its candidates
    (the build bundled with this image's PyTorch does); below the row threshold and for every other shape no preference is stated."""
    import os
    import re
    assert os.environ.get("TENSILE_STREAMK_DATA_PARALLEL") == "1"          # set at package import
    shapes = [(3601 * 24, 768, 3072), (3601 * 24, 768, 768), (3600 * 24, 768, 192), (3601 * 291, 768, 3072), (3601 * 291, 768, 768),
              (901 * 2473, 384, 1536), (901 * 291, 384, 1536), (901 * 128, 384, 1536), (901 * 64, 384, 1536), (3601 * 7, 2304, 768), (1601 * 40, 768, 3072)]
    for m, n, k in shapes:
        text = hip.linear_lt_describe(m, n, k, torch.float16)
        taken = [ln for ln in text.splitlines() if ln.startswith("* ") or ln.startswith("*p")]
        assert len(taken) == 1, text
        has_preference = (n, k) == (384, 1536) and m >= 65536
        assert ("measured preference" in text) == has_preference, text
        if taken[0].startswith("*p"):
            assert has_preference and "_MT192x128x64_MI16x16x1_" in taken[0], taken[0]
        elif has_preference:
            assert "not among the candidates" in text, text
        assert " ws=0 " in taken[0], taken[0]
        gsu = re.search(r"_GSU(\d+)_", taken[0])
        assert not (gsu and int(gsu.group(1)) > 1 and "GSUAMB" not in taken[0]), taken[0]
        print(f"[linear_lt] M={m} N={n} K={k}: took {taken[0][:150]}")


@pytest.mark.parametrize("k", [384, 768])
def test_linear_kres_does_not_write_past_the_output(k):
    m, n = 515, 128
    g = torch.Generator().manual_seed(4)
    x = torch.randn(m, k, generator=g).half().to(DEV)
    w = (torch.randn(n, k, generator=g) * 0.05).half().to(DEV)
    b = torch.zeros(n).half().to(DEV)
    for planar in (False, True):
        big = torch.full((m * n + 4096,), 7.0, dtype=torch.float16, device=DEV)
        lib = hip.load_library()
        rc = getattr(lib, hip.LINEAR_KRES_WIDTHS[k][0])(x.data_ptr(), w.data_ptr(), b.data_ptr(), big.data_ptr(), m, n, 0,
                                 hip.PLANAR64 if planar else hip.ROW_MAJOR, hip.dtype_code(torch.float16),
                                 torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.all(big[m * n:] == 7.0)
        assert not torch.any(big[: m * n] == 7.0)


def test_linear_kres_rejects_bad_shapes():
    x = torch.zeros(8, 384, dtype=torch.float16, device=DEV)
    w = torch.zeros(96, 384, dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError, match="dss_linear_k384"):
        hip.linear_k384(x, w, torch.zeros(96, dtype=torch.float16, device=DEV))     # N % 64 != 0
    with pytest.raises(ValueError, match="reduction dimension"):
        hip.linear_kres(torch.zeros(8, 512, dtype=torch.float16, device=DEV),
                        torch.zeros(64, 512, dtype=torch.float16, device=DEV),
                        torch.zeros(64, dtype=torch.float16, device=DEV))


# ----------------------------------------------------------------------------- residual add + LayerNorm + Linear in one kernel
def _lnlinear_case(m, n, k, dtype, seed, offset=0.5, outliers=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(m, k, generator=g) * 3 + offset
    if outliers:                     # what trained DINO residual streams look like: a few channels carry 150-300
        x[:, 7] += 250.0
        x[:, 100] -= 180.0
        x[:, k - 3] += 300.0
    r = (torch.randn(m, k, generator=g) * 2).to(dtype)
    w = torch.randn(n, k, generator=g) * 0.05 * (384 / k) ** 0.5
    b = torch.randn(n, generator=g) * 0.2
    gamma, beta = torch.randn(k, generator=g) * 0.5 + 1.0, torch.randn(k, generator=g) * 0.3
    return x, r, w, b, gamma, beta


@pytest.mark.parametrize("m,n,k", [(901, 1152, 384), (512, 1536, 384), (1, 64, 384), (77, 384, 384), (1025, 128, 384),
                                   (2 * 901 + 5, 1536, 384), (3601, 3072, 768), (256, 768, 768), (1, 64, 768), (513, 128, 768)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("gelu", [False, True, 2])
@pytest.mark.parametrize("res", [None, "rows", "planar"])
def test_lnlinear_matches_fp64_reference(m, n, k, dtype, gelu, res):
    """dss_lnlinear_k384/_k768 against the fp64 composition x += r; LayerNorm(x); Linear; (erf GELU) it replaces.  The
    residual stream must come back as the SAME fp32 sum the standalone LayerNorm pass writes (bit-exact); the output bar is
    the half dtype's output rounding plus the operand rounding of x (|x| <= ~15 sigma here), as for the plain kernel."""
    if gelu == 2 and dtype != torch.float16:
        pytest.skip("gelu = 2 is the packed-f16 form")
    x, r, w, b, gamma, beta = _lnlinear_case(m, n, k, dtype, m + n + k + 11)
    wg, aux = hip.lnlinear_prepare(w.to(DEV), b.to(DEV), gamma.to(DEV), beta.to(DEV), dtype)
    xd = x.to(DEV)
    rd = None if res is None else (_to_planar(r) if res == "planar" else r).to(DEV)
    out = hip.lnlinear(xd, rd, wg, aux, 1e-6, gelu=gelu, planar=(res == "planar"), residual_planar=(res == "planar"))
    out = (hip.planar_to_rows(out) if res == "planar" else out).float().cpu()
    xsum = x if res is None else x + r.float()
    assert torch.equal(xd.cpu(), xsum)
    ref = F.linear(F.layer_norm(xsum.double(), (k,), gamma.double(), beta.double(), 1e-6), w.double(), b.double())
    if gelu:
        ref = F.gelu(ref)
    assert out.shape == (m, n)
    tol = (1.5e-3 if dtype == torch.float16 else 1.2e-2) * max(1.0, ref.abs().max().item())
    assert (out.double() - ref).abs().max().item() <= tol


@pytest.mark.parametrize("m,n,k,gelu", [(901 * 64, 1152, 384, 0), (3601 * 16, 3072, 768, 1)])
def test_lnlinear_repeated_launches_give_the_same_bits(m, n, k, gelu):
    """The kernel has no atomics and a fixed reduction order: the same launch must return the same bits every time, whatever ran
    in front of it.  (Round 5: the correction term of the LAST 32-column chunk was consumed through a register copy made in front
    of the wait for its load - wrong values there in a few workgroups of 1-8 launches in 300 at these sizes, whenever the load
    took longer than the MFMA phase; scripts/debug/lnlinear_stress.py is the long form, scripts/check_async_asm.py the structural
    check.)"""
    x, r, w, b, gamma, beta = _lnlinear_case(m, n, k, torch.float16, 5)
    wg, aux = hip.lnlinear_prepare(w.to(DEV), b.to(DEV), gamma.to(DEV), beta.to(DEV), torch.float16)
    x0, rd = x.to(DEV), r.to(DEV)
    first = hip.lnlinear(x0.clone(), rd, wg, aux, 1e-6, gelu=gelu)
    junk = torch.randn(4096, 4096, device=DEV)
    for i in range(60):
        if i % 3 == 1:
            junk = junk @ junk * 1e-3
        elif i % 3 == 2:
            torch.cuda.synchronize()
        out = hip.lnlinear(x0.clone(), rd, wg, aux, 1e-6, gelu=gelu)
        assert torch.equal(out, first), (i, (out != first).nonzero()[:4].tolist())


@pytest.mark.parametrize("k,n", [(384, 1152), (768, 3072)])
def test_lnlinear_outlier_channels_and_large_mean(k, n):
    """Rows whose variance is carried by three outlier channels, rows whose mean is far from zero - up to beyond the f16 range -
    and outlier channels INSIDE the nine columns the row pivot is taken from.  The kernel rounds x - pivot to f16 once (round 5;
    rounds 4 rounded x itself: error 2^-11 |mean| / sigma per operand on large-mean rows, inf beyond 65504) where the LayerNorm ->
    Linear pair rounds the normalised value: in every case the fused kernel must be as accurate as the pair (same bar for both:
    the relative error of the large entries is what counts, and it is the same)."""
    m = 700
    cases = ((40.0, False, ()), (0.0, True, ()), (-25.0, True, ()), (3.0e4, False, ()), (1.0e5, True, ()),
             (5.0, True, (1, 5, 17)),         # one outlier in each pivot triple: every triple's median is still clean
             (-8.0, True, (0, 1, 16)))        # one triple lost (two outliers in it) + one in another: the outer median drops it
    for offset, outliers, pivot_outliers in cases:
        x, r, w, b, gamma, beta = _lnlinear_case(m, n, k, torch.float16, 99, offset=offset, outliers=outliers)
        for j, c in enumerate(pivot_outliers):
            x[:, c] += (220.0, -260.0, 190.0)[j]
        wg, aux = hip.lnlinear_prepare(w.to(DEV), b.to(DEV), gamma.to(DEV), beta.to(DEV), torch.float16)
        xa, xb = x.to(DEV), x.to(DEV)
        out = hip.lnlinear(xa, r.to(DEV), wg, aux, 1e-6).float().cpu()
        h = hip.layernorm(xb, gamma.to(DEV), beta.to(DEV), 1e-6, torch.float16, residual=r.to(DEV))
        two = hip.linear_kres(h, w.half().to(DEV), b.half().to(DEV)).float().cpu()
        assert torch.equal(xa, xb)
        xs = (x + r.float()).double()
        ref = F.linear(F.layer_norm(xs, (k,), gamma.double(), beta.double(), 1e-6), w.double(), b.double())
        assert bool(torch.isfinite(out).all()), offset
        e_fused, e_two = (out.double() - ref).abs().max().item(), (two.double() - ref).abs().max().item()
        # (at |x| ~ 1e5 the fp32 residual stream itself resolves 2^-7: both paths carry that, the fp64 reference does not)
        assert e_fused <= max(2.5 * e_two, 2e-3 * ref.abs().max().item()), (offset, outliers, pivot_outliers, e_fused, e_two)


@pytest.mark.parametrize("k", [384, 768])
def test_lnlinear_ragged_block_touches_nothing_outside(k):
    """M not a multiple of the 64-row wave tile or of the workgroup: rows past M of x, the residual and the output (all three
    embedded in larger guard buffers) stay untouched, rows below M are all written."""
    m, n = 515, 128
    g = torch.Generator().manual_seed(4)
    xbig = torch.full((m + 64, k), 3.0, device=DEV)
    xbig[:m] = torch.randn(m, k, generator=g).to(DEV)
    x0 = xbig.clone()
    r = torch.randn(m, k, generator=g).half().to(DEV)
    w, b = torch.randn(n, k, generator=g) * 0.05, torch.zeros(n)
    wg, aux = hip.lnlinear_prepare(w.to(DEV), b.to(DEV), torch.ones(k, device=DEV), torch.zeros(k, device=DEV), torch.float16)
    lib = hip.load_library()
    entry = hip.LINEAR_KRES_WIDTHS[k][0].replace("dss_linear", "dss_lnlinear")
    for planar in (False, True):
        xbig.copy_(x0)
        big = torch.full((m * n + 4096,), 7.0, dtype=torch.float16, device=DEV)
        rc = getattr(lib, entry)(xbig.data_ptr(), r.data_ptr(), hip.ROW_MAJOR, 1e-6, wg.data_ptr(), aux.data_ptr(), big.data_ptr(), m, n,
                                 0, hip.PLANAR64 if planar else hip.ROW_MAJOR, hip.dtype_code(torch.float16),
                                 torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.all(big[m * n:] == 7.0) and not torch.any(big[: m * n] == 7.0)
        assert torch.equal(xbig[m:], x0[m:]) and torch.equal(xbig[:m], x0[:m] + r.float())


def test_lnlinear_rejects_bad_arguments():
    x = torch.zeros(8, 384, device=DEV)
    wg, aux = torch.zeros(64, 384, dtype=torch.float16, device=DEV), torch.zeros(64, 2, device=DEV)
    lib = hip.load_library()
    s = torch.cuda.current_stream().cuda_stream
    out = torch.zeros(8, 64, dtype=torch.float16, device=DEV)
    assert lib.dss_lnlinear_k384(x.data_ptr(), 0, 0, 1e-6, wg.data_ptr(), 0, out.data_ptr(), 8, 64, 0, 0, 1, s) == -1   # no aux
    assert lib.dss_lnlinear_k384(x.data_ptr(), 0, 7, 1e-6, wg.data_ptr(), aux.data_ptr(), out.data_ptr(), 8, 64, 0, 0, 1, s) == -1
    assert lib.dss_lnlinear_k384(0, 0, 0, 1e-6, wg.data_ptr(), aux.data_ptr(), out.data_ptr(), 8, 64, 0, 0, 1, s) == -1
    assert lib.dss_lnlinear_k384(x.data_ptr(), 0, 0, 1e-6, wg.data_ptr(), aux.data_ptr(), out.data_ptr(), 8, 64, 0, 0, 1, s) == 0


@pytest.mark.parametrize("b,t", [(2, 901), (1, 197), (3, 65), (5, 130), (1, 3601), (7, 257)])
@pytest.mark.parametrize("res", [None, "rows", "planar"])
@pytest.mark.parametrize("k,dtype", [(384, torch.float16), (768, torch.float16), (384, torch.bfloat16), (768, torch.bfloat16)])
def test_lnlinear_kfeatures_matches_the_three_kernels_it_replaces(b, t, res, k, dtype):
    """dss_lnlinear_kfeatures (x += r; norm1; K projection; CLS drop; f16 copy; inverse norms; D = 384 / 768, f16 / bf16 operands)
    against the fp64 composition, and against LayerNorm + GEMM + dss_kfeatures_finalize: the fp32 features to the
    operand-rounding bar of the other fused kernels, k16 = f16(k32) exactly (f16 whatever the operand type), rnorm =
    1 / |k16 row| to fp32 rounding, the residual stream untouched, every output row written (image boundaries fall inside
    workgroups and inside waves at these shapes)."""
    n = k
    m = b * t
    x, r, w, bias, gamma, beta = _lnlinear_case(m, n, k, dtype, 5 * b + t + k)
    wg, aux = hip.lnlinear_prepare(w.to(DEV), bias.to(DEV), gamma.to(DEV), beta.to(DEV), dtype)
    xd = x.view(b, t, k).to(DEV)
    rd = None if res is None else (_to_planar(r).to(DEV) if res == "planar" else r.view(b, t, k).to(DEV))
    sentinel = 7.0
    bufs = (torch.full((b + 2, t - 1, k), sentinel, device=DEV), torch.full((b + 2, t - 1, k), sentinel, dtype=torch.float16, device=DEV),
            torch.full((b + 2, t - 1), sentinel, device=DEV))
    k32, k16, rn = hip.lnlinear_kfeatures(xd, rd, wg, aux, 1e-6, residual_planar=(res == "planar"), out=tuple(u[1:1 + b] for u in bufs))
    torch.cuda.synchronize()
    xsum = x if res is None else x + r.float()
    assert torch.equal(xd.cpu().view(m, k), x)       # the last reader of the stream does not store the sum
    for u in bufs:   # nothing outside the slices
        assert bool((u[0] == sentinel).all()) and bool((u[1 + b:] == sentinel).all())
    ref = F.linear(F.layer_norm(xsum.double(), (k,), gamma.double(), beta.double(), 1e-6), w.double(), bias.double())
    ref = ref.view(b, t, n)[:, 1:]
    tol = (1.5e-3 if dtype == torch.float16 else 1.2e-2) * max(1.0, ref.abs().max().item())
    assert (k32.cpu().double() - ref).abs().max().item() <= tol
    assert torch.equal(k16, k32.half())
    want_rn = 1.0 / k16.double().norm(dim=-1).clamp_min(1e-12)
    assert ((rn.double() - want_rn).abs() / want_rn).max().item() <= 1e-6
    # the three-kernel path on the same inputs: same quantity, its own roundings
    xb = x.view(b, t, k).to(DEV)
    h = hip.layernorm(xb, gamma.to(DEV), beta.to(DEV), 1e-6, dtype, residual=rd, residual_planar=(res == "planar"))
    kp = torch.mm(h.view(m, k), w.to(dtype).to(DEV).t(), out_dtype=torch.float32).view(b, t, n)
    o32, o16, orn = hip.kfeatures_finalize(kp, bias.to(DEV))
    assert (o32.cpu().double() - ref).abs().max().item() <= tol
    assert (k32 - o32).abs().max().item() <= 2 * tol


def test_lnlinear_kfeatures_rejects_bad_arguments():
    x = torch.zeros(2, 65, 384, device=DEV)
    wg, aux = torch.zeros(384, 384, dtype=torch.float16, device=DEV), torch.zeros(384, 2, device=DEV)
    k32, k16, rn = torch.zeros(2, 64, 384, device=DEV), torch.zeros(2, 64, 384, dtype=torch.float16, device=DEV), torch.zeros(2, 64, device=DEV)
    lib, s = hip.load_library(), torch.cuda.current_stream().cuda_stream
    f = lib.dss_lnlinear_kfeatures_k384
    ok = (x.data_ptr(), 0, 0, 1e-6, wg.data_ptr(), aux.data_ptr(), k32.data_ptr(), k16.data_ptr(), rn.data_ptr(), 130, 65, 1e-12, s)
    assert f(*ok) == 0
    assert f(*ok[:9], 130, 64, 1e-12, s) == -1          # T <= 64: the row remap assumes one image boundary per 64 rows
    assert f(*ok[:9], 131, 65, 1e-12, s) == -1          # M not a whole number of images
    assert f(*ok[:6], 0, *ok[7:]) == -1                 # no k32
    assert f(*ok[:2], 7, *ok[3:]) == 0                  # (res_layout is not read without a residual)
    g = lib.dss_lnlinear_kfeatures
    okg = (*ok[:11], 384, 1e-12, 1, s)
    assert g(*okg) == 0
    assert g(*ok[:11], 512, 1e-12, 1, s) == -1          # D must be 384 or 768
    assert g(*ok[:11], 384, 1e-12, 0, s) == -1          # dtype must be f16 or bf16
    with pytest.raises(ValueError):
        hip.lnlinear_kfeatures(torch.zeros(1, 64, 384, device=DEV), None, wg, aux, 1e-6)


@pytest.mark.parametrize("b,h,w,d", [(2, 480, 480, 384), (1, 224, 224, 384), (3, 37, 52, 384), (1, 16, 16, 384), (2, 130, 245, 768),
                                     (5, 64, 48, 384), (1, 333, 500, 384)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_patch_embed16_matches_transform_conv_and_position_embedding(b, h, w, d, dtype):
    """dss_patch_embed_p16 (u8 image -> fp32 residual-stream rows) against the fp64 composition it replaces: ToTensor + Normalize,
    crop to whole patches, Conv2d(3, D, 16, 16), + position embedding; and against the three-pass path (patchify + GEMM + add).
    Sizes that are not multiples of the patch (crop), of 8 (unaligned 8-byte gathers) and a single patch; row 0 of every image
    and everything outside x untouched."""
    g = torch.Generator().manual_seed(h * 7 + w + d)
    img = torch.randint(0, 256, (b, h, w, 3), generator=g, dtype=torch.uint8)
    wt = torch.randn(d, 3, 16, 16, generator=g) * 0.03
    bias = torch.randn(d, generator=g) * 0.1
    hp, wp_ = h // 16, w // 16
    n_p = hp * wp_
    pos = torch.randn(n_p, d, generator=g) * 0.5
    wp, bp = hip.patch_embed16_prepare(wt.to(DEV), bias.to(DEV), dtype)
    xbig = torch.full((b + 2, n_p + 1, d), 7.0, device=DEV)
    assert bp.dtype == torch.float32
    if (h + w) % 2:      # both ways of carrying the folded bias: in the kernel's own (half) operand, or in fp32 inside `pos`
        hip.patch_embed16(img.to(DEV), wp, bp, pos.to(DEV), xbig[1:1 + b])
    else:
        hip.patch_embed16(img.to(DEV), wp, None, (pos.to(DEV) + bp).contiguous(), xbig[1:1 + b])
    torch.cuda.synchronize()
    assert bool((xbig[0] == 7).all()) and bool((xbig[1 + b:] == 7).all()) and bool((xbig[1:1 + b, 0] == 7).all())
    mean = torch.tensor(hip.IMAGENET_MEAN, dtype=torch.float64).view(1, 3, 1, 1)
    std = torch.tensor(hip.IMAGENET_STD, dtype=torch.float64).view(1, 3, 1, 1)
    t = (img[:, :hp * 16, :wp_ * 16].permute(0, 3, 1, 2).double() / 255.0 - mean) / std
    ref = F.conv2d(t, wt.double(), bias.double(), stride=16).flatten(2).transpose(1, 2) + pos.double()
    got = xbig[1:1 + b, 1:].cpu().double()
    tol = (2e-3 if dtype == torch.float16 else 1.6e-2) * max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() <= tol
    # the three passes it replaces, same dtype: same quantity, their own roundings
    patches = hip.preprocess_patchify(img.to(DEV), 16, dtype)
    tok = F.linear(patches, wt.reshape(d, -1).to(DEV, dtype), bias.to(DEV, dtype)).float() + pos.to(DEV)
    assert (tok.cpu().double() - ref).abs().max().item() <= 2 * tol
    assert (got - tok.cpu().double()).abs().max().item() <= 2 * tol


def test_kfeatures_finalize_writes_into_caller_slices():
    """`out=`: several ViT forwards fill ONE step's buffers (bench.py step_fed) - same values as the allocating form, nothing
    outside the slices touched."""
    g = torch.Generator().manual_seed(2)
    kp = torch.randn(3, 11, 64, generator=g).to(DEV)
    bias = torch.randn(64, generator=g).to(DEV)
    want = hip.kfeatures_finalize(kp, bias)
    bufs = (torch.full((5, 10, 64), 7.0, device=DEV), torch.full((5, 10, 64), 7.0, dtype=torch.float16, device=DEV),
            torch.full((5, 10), 7.0, device=DEV))
    got = hip.kfeatures_finalize(kp, bias, out=tuple(b[1:4] for b in bufs))
    for w, gt, b in zip(want, got, bufs):
        assert torch.equal(w, gt) and torch.equal(b[1:4], w) and bool((b[0] == 7).all()) and bool((b[4] == 7).all())


# ----------------------------------------------------------------------------- attention
def _attention_ref(qkv, heads, scale):
    b, t, _ = qkv.shape
    q, k, v = qkv.double().reshape(b, t, 3, heads, 64).permute(2, 0, 3, 1, 4)
    a = ((q @ k.transpose(-1, -2)) * scale).softmax(-1)
    return (a @ v).transpose(1, 2).reshape(b, t, heads * 64)


@pytest.mark.parametrize("b,t,heads", [(2, 901, 6), (1, 197, 6), (1, 17, 2), (1, 64, 1), (1, 65, 1), (3, 130, 12),
                                        (1, 1, 1), (1, 33, 3), (1, 3601, 2), (2, 257, 2), (1, 320, 1), (1, 513, 2),
                                        (2, 545, 1), (1, 577, 3), (1, 96, 2), (1, 1025, 1), (1, 6401, 2),
                                        (1, 5136, 1)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_attention_matches_fp64_reference(b, t, heads, dtype):
    g = torch.Generator().manual_seed(b * 100 + t + heads)
    qkv = (torch.randn(b, t, 3 * heads * 64, generator=g) * 1.5).to(dtype)
    out = hip.attention(qkv.to(DEV), heads, 0.125).cpu()
    ref = _attention_ref(qkv, heads, 0.125)
    err = (out.double() - ref).abs().max().item()
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    assert err <= tol * max(1.0, ref.abs().max().item()), err
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("b,t,heads", [(2, 901, 6), (1, 197, 6), (3, 130, 12), (1, 65, 1)])
def test_attention_planar_qkv_is_bit_identical_to_interleaved(b, t, heads):
    g = torch.Generator().manual_seed(b + t)
    qkv = (torch.randn(b, t, 3 * heads * 64, generator=g) * 1.5).half()
    a = hip.attention(qkv.to(DEV), heads, 0.125)
    planar = _to_planar(qkv.reshape(b * t, -1)).to(DEV)          # [3*heads, B*T, 64]
    p = hip.attention(planar, heads, 0.125, planar_bt=(b, t))
    assert torch.equal(a, p)


def test_attention_peaked_rows_force_online_rescale():
    """One key far above the rest late in the sequence: the running max jumps, exercising the rescale."""
    b, t, heads = 1, 300, 2
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(b, t, 3, heads, 64, generator=g) * 0.3
    qkv[0, 250, 1] = qkv[0, 7, 0] * 40.0   # key 250 aligned with query 7 (both heads)
    qkv[0, 40, 1] = qkv[0, 9, 0] * 25.0
    qkv = qkv.reshape(b, t, -1).half()
    out = hip.attention(qkv.to(DEV), heads, 0.125).cpu()
    ref = _attention_ref(qkv, heads, 0.125)
    assert (out.double() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())


def test_attention_full_batch_is_deterministic_and_matches_fp64():
    """The bench shape (many workgroups, every XCD, ragged last query block and last key tile): repeated launches are
    bit-identical (no race between the LDS restaging and the fragment reads) and a sample of images matches fp64."""
    b, t, heads = 24, 901, 6
    g = torch.Generator().manual_seed(11)
    qkv = (torch.randn(b, t, 3 * heads * 64, generator=g) * 1.2).half()
    a = hip.attention(qkv.to(DEV), heads, 0.125)
    for _ in range(5):
        assert torch.equal(a, hip.attention(qkv.to(DEV), heads, 0.125))
    ref = _attention_ref(qkv[::8], heads, 0.125)
    assert (a[::8].cpu().double() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())


def test_attention_rescale_test_catches_slow_drift_and_late_spikes():
    """The kernel's rescale test is the row sum of the probabilities against the OLD running max (no max tree): rows
    whose maximum creeps up a little with every key tile (many small rescales never triggered individually), rows with
    a late outlier key of +-40 sigma, and all-equal scores - against fp64."""
    b, t, heads = 1, 700, 2
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(b, t, 3, heads, 64, generator=g) * 0.5
    ramp = torch.linspace(0.2, 3.0, t)[:, None]
    qkv[0, :, 1, 0] = qkv[0, 3, 0, 0] * ramp            # head 0: every key a bit more aligned with query 3 than the last
    qkv[0, 650, 1, 1] = qkv[0, 11, 0, 1] * 60.0          # head 1: one huge late key for query 11
    qkv[0, 100, 1, 1] = -qkv[0, 12, 0, 1] * 60.0         # and one hugely negative one for query 12
    qkv[0, 200:232, 0, 1] = 0.0                          # queries with all-equal (zero) scores
    qkv = qkv.reshape(b, t, -1).half()
    out = hip.attention(qkv.to(DEV), heads, 0.125).cpu()
    ref = _attention_ref(qkv, heads, 0.125)
    assert torch.isfinite(out.float()).all()
    assert (out.double() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())


# ----------------------------------------------------------------------------- normalise + affinity
def test_normalize_rows():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(50, 384, generator=g)
    x[3] = 0.0
    x[4] = 1e-20
    out = hip.normalize_rows(x.to(DEV)).cpu()
    ref = F.normalize(x, p=2, dim=-1)
    assert (out - ref).abs().max().item() < 1e-6 and torch.equal(out[3], torch.zeros(384))
    out2 = hip.normalize_rows(torch.randn(9, 37, generator=g).to(DEV))  # D not a multiple of 4
    assert abs(out2.norm(dim=-1).mean().item() - 1) < 1e-5


@pytest.mark.parametrize("b,n,d", [(2, 900, 384), (1, 713, 384), (1, 16, 32), (1, 64, 64), (1, 65, 32), (3, 129, 768),
                                    (1, 3600, 768), (2, 196, 384), (1, 80, 64), (1, 81, 64), (1, 784, 384)])
def test_affinity_matches_fp64(b, n, d):
    feats = torch.from_numpy(np.stack([synthetic.synthetic_features("blobs" if i % 2 else "random", n, d, 50 + i,
                                                                     None if int(n ** 0.5) ** 2 == n else (1, n))
                                       for i in range(b)]))
    fn = hip.normalize_rows(feats.to(DEV))
    wp = hip.affinity(fn)
    ld = hip.affinity_ld(n)
    nt, ntf, e4 = hip.wsym_layout(n)                 # full tile rows, then the edge strip's mini tiles in blocks of 16
    assert nt == ld // 64 and (e4 > 0) == (n > 64 and 1 <= n % 64 <= 16) and ntf == nt - (e4 > 0)
    assert tuple(wp.shape) == (b, (ntf * (ntf + 1) // 2 + ((ntf + 1) * e4 + 15) // 16) * 4096) and ld % 64 == 0 and ld >= n
    w = hip.affinity_to_dense(wp, n).cpu()           # [b, ld, ld] from the packed upper-triangular tiles
    assert torch.equal(w[:, :, n:], torch.zeros(b, ld, ld - n)) and torch.equal(w[:, n:, :], torch.zeros(b, ld - n, ld))
    x = F.normalize(feats.double(), dim=-1)
    ref = (x @ x.transpose(1, 2)).clamp_min(0)
    assert (w[:, :n, :n].double() - ref).abs().max().item() < 2e-6
    assert torch.equal(w, w.transpose(1, 2))         # diagonal tiles are stored in full and bit-symmetric
    wneg = hip.affinity_to_dense(hip.affinity(fn, threshold_at_zero=False), n).cpu()
    assert (wneg[:, :n, :n].double() - x @ x.transpose(1, 2)).abs().max().item() < 2e-6
    # split-f16 path (default of the spectral stage): same packed layout, fp32-class accuracy, fused normalisation
    ws = hip.affinity_to_dense(hip.affinity_split(feats.to(DEV)), n).cpu()
    assert torch.equal(ws[:, :, n:], torch.zeros(b, ld, ld - n)) and torch.equal(ws[:, n:, :], torch.zeros(b, ld - n, ld))
    assert (ws[:, :n, :n].double() - ref).abs().max().item() < 2e-6
    # symmetric to rounding only: inside a diagonal tile the two cross terms accumulate in the opposite order
    assert (ws - ws.transpose(1, 2)).abs().max().item() < 2e-7
    # 16-bit fixed-point storage (default of the spectral stage for the scale-invariant problem): round(65535 w)
    wq_packed = hip.affinity_split(feats.to(DEV), u16=True)
    assert wq_packed.dtype == torch.int16 and wq_packed.shape == wp.shape
    wq = hip.affinity_to_dense(wq_packed, n).cpu()
    assert torch.equal(wq[:, :, n:], torch.zeros(b, ld, ld - n)) and torch.equal(wq[:, n:, :], torch.zeros(b, ld - n, ld))
    assert (wq[:, :n, :n].double() - ref).abs().max().item() <= 0.5 / 65535 + 2e-6     # half a quantisation step
    assert wq.min().item() >= 0.0 and wq.max().item() <= 1.0
    assert abs(wq[0, 0, 0].item() - 1.0) < 1e-6                                         # w_ii = 1 -> 65535 exactly
    with pytest.raises(ValueError, match="16-bit"):
        hip.affinity_split(feats.to(DEV), threshold_at_zero=False, u16=True)
    wsn = hip.affinity_to_dense(hip.affinity_split(feats.to(DEV), threshold_at_zero=False), n).cpu()
    assert (wsn[:, :n, :n].double() - x @ x.transpose(1, 2)).abs().max().item() < 2e-6
    # fused build (default of the spectral stage): raw features -> packed 16-bit W in one kernel, f16 MFMA operands
    wf_packed = hip.affinity_fused_u16(feats.to(DEV))
    assert wf_packed.dtype == torch.int16 and wf_packed.shape == wp.shape
    wf = hip.affinity_to_dense(wf_packed, n).cpu()
    assert torch.equal(wf[:, :, n:], torch.zeros(b, ld, ld - n)) and torch.equal(wf[:, n:, :], torch.zeros(b, ld - n, ld))
    # f16 rounding of the features (2^-11 relative each): ~2^-11 sqrt(2 / D) on w - 3.5e-5 at the ViTs' D = 384
    assert (wf[:, :n, :n].double() - ref).abs().max().item() <= 1e-4 * max(1.0, (384 / d) ** 0.5)
    assert (wf[:, :n, :n].double() - ref).abs().mean().item() <= 1e-5 * max(1.0, (384 / d) ** 0.5)
    if n >= 256:   # round to nearest, not truncation: no bias of minus half a step (7.6e-6) over many entries
        assert abs((wf[:, :n, :n].double() - ref)[ref > 1e-3].mean().item()) <= 2e-6
    assert wf.min().item() >= 0.0 and wf.max().item() <= 1.0
    assert abs(torch.diagonal(wf[:, :n, :n], dim1=1, dim2=2).min().item() - 1.0) < 2e-5    # w_ii = 1 (norms of the rounded rows)
    assert (wf - wf.transpose(1, 2)).abs().max().item() <= 1.01 / 65535     # symmetric to one quantisation step
    scaled = hip.affinity_to_dense(hip.affinity_fused_u16(feats.to(DEV) * 32.0), n).cpu()  # normalises itself: a power-of-two
    assert (scaled - wf).abs().max().item() <= 1.01 / 65535                                 # scale changes no f16 rounding
    raw = feats.to(DEV) * 0.37   # un-normalised features (normalize=False path)
    wr = hip.affinity_to_dense(hip.affinity_split(raw, normalize=False), n).cpu()
    rr = (raw.double().cpu() @ raw.double().cpu().transpose(1, 2)).clamp_min(0)
    assert (wr[:, :n, :n].double() - rr).abs().max().item() < 2e-6 * max(1.0, rr.max().item())


@pytest.mark.parametrize("b,n,d", [(2, 900, 384), (1, 713, 384), (1, 16, 32), (1, 65, 32), (3, 129, 768), (9, 300, 384),
                                    (1, 3600, 768), (1, 6400, 768)])
def test_affinity_from_f16_handover(b, n, d):
    """The pipeline's affinity build: `kfeatures_finalize` (K-projection output + bias -> fp32 features without the CLS
    row, their f16 copy, inverse norms of the ROUNDED rows) followed by `affinity_f16_u16` (256 x 256 tiles, LDS-DMA
    panels) - against fp64 on the exact features and, entry by entry, against the one-kernel build from fp32."""
    g = torch.Generator().manual_seed(n + d)
    feats = torch.from_numpy(np.stack([synthetic.synthetic_features("blobs" if i % 2 else "random", n, d, 70 + i,
                                                                     None if int(n ** 0.5) ** 2 == n else (1, n))
                                       for i in range(b)]))
    bias = torch.randn(d, generator=g) * 0.1
    kproj = torch.cat((torch.randn(b, 1, d, generator=g), feats - bias), dim=1)        # token 0 = CLS, bias not yet added
    k32, k16, rn = hip.kfeatures_finalize(kproj.to(DEV), bias.to(DEV))
    want = (kproj[:, 1:] + bias)
    assert torch.equal(k32.cpu(), want) and torch.equal(k16.cpu(), want.half())
    assert (rn.cpu().double() - 1.0 / want.half().double().norm(dim=-1)).abs().max().item() <= 2e-6 * rn.max().item()
    wq_packed = hip.affinity_f16_u16(k16, rn)
    ld = hip.affinity_ld(n)
    assert wq_packed.dtype == torch.int16 and tuple(wq_packed.shape) == (b, hip.affinity_elems(n))
    wq = hip.affinity_to_dense(wq_packed, n).cpu()
    assert torch.equal(wq[:, :, n:], torch.zeros(b, ld, ld - n)) and torch.equal(wq[:, n:, :], torch.zeros(b, ld - n, ld))
    x = F.normalize(want.double(), dim=-1)
    ref = (x @ x.transpose(1, 2)).clamp_min(0)
    assert (wq[:, :n, :n].double() - ref).abs().max().item() <= 1e-4 * max(1.0, (384 / d) ** 0.5)
    assert (wq[:, :n, :n].double() - ref).abs().mean().item() <= 1e-5 * max(1.0, (384 / d) ** 0.5)
    assert wq.min().item() >= 0.0 and wq.max().item() <= 1.0
    assert abs(torch.diagonal(wq[:, :n, :n], dim1=1, dim2=2).min().item() - 1.0) < 2e-5
    assert (wq - wq.transpose(1, 2)).abs().max().item() <= 1.01 / 65535
    if d >= 256:   # the same operands as the one-kernel build from fp32 (it rounds to f16 itself): at most one step apart
        wf = hip.affinity_to_dense(hip.affinity_fused_u16(k32), n).cpu()
        assert (wq - wf).abs().max().item() <= 2.01 / 65535
    # zero row: eps keeps the inverse norm finite, its similarities are 0
    kz = kproj.clone()
    kz[0, 3] = -bias
    k32z, k16z, rnz = hip.kfeatures_finalize(kz.to(DEV), bias.to(DEV))
    wz = hip.affinity_to_dense(hip.affinity_f16_u16(k16z, rnz), n).cpu()
    assert torch.isfinite(rnz).all() and wz[0, 2].abs().max().item() == 0.0 and wz[0, :, 2].abs().max().item() == 0.0


# ----------------------------------------------------------------------------- eigen stage
EIG_FILES = sorted(glob.glob(str(HERE / "golden" / "eigs_*.npz")))


# storage of W x affinity build: fused f16 -> 16-bit fixed point (the default), split-f16 -> u16, split-f16 -> floats
@pytest.mark.parametrize("w_dtype,mode", [("u16", "fused"), ("u16", "split"), ("f32", "split")])
@pytest.mark.parametrize("path", EIG_FILES, ids=lambda p: p.split("eigs_")[-1][:-4])
def test_eigs_match_reference_goldens(path, w_dtype, mode):
    feats, K, ref_lam, ref_vec, g = golden_case(path)
    ev, vec, info = spectral.laplacian_eigs_from_features(torch.from_numpy(feats)[None].to(DEV), K, w_dtype=w_dtype,
                                                          affinity_mode=mode)
    assert info.item() > 0
    assert vec.dtype == torch.float32 and tuple(vec.shape) == (1, K, feats.shape[0]) and tuple(ev.shape) == (1, K)
    check_eigs(vec[0].cpu().numpy(), ev[0].cpu().numpy(), ref_vec, ref_lam, what=path, d=build_w64(feats)[1],
               ext=golden_ext(g))
    if feats.shape[0] <= 1600:
        _, d = build_w64(feats)
        assert d_orthonormality(vec[0].cpu().numpy(), d=d) < 1e-4  # v^T D v = 1 like ARPACK with M=D
    for k in range(K):
        assert not (0.5 < (vec[0, k] > 0).float().mean().item() < 1.0)  # sign-rule post-condition


def test_eigs_u16_storage_agrees_with_float_storage():
    """Same images through the storages / builds of W.  Split build, u16 vs float storage: eigenvalues within 3e-6,
    eigenvectors within 3e-6 in cosine (the quantisation step of 1/65535 is far below the 1e-4 parity budget), identical
    sign decisions.  Fused f16-operand build (the default) vs the float reference: within 2e-5 / 2e-5 - the measured cost
    of rounding the features to f16 (BASELINE config 5's "fp16 features"), still 5x inside the budget on the worst case
    (i.i.d. random features, eigenvalue gaps ~2e-3)."""
    n, d, K, b = 900, 384, 5, 6
    feats = torch.from_numpy(np.stack([synthetic.synthetic_features("blobs" if i % 2 else "random", n, d, 70 + i)
                                       for i in range(b)])).to(DEV)
    out = {}
    for key, (w_dtype, mode) in {"u16": ("u16", "split"), "f32": ("f32", "split"), "fused": ("u16", "fused")}.items():
        out[key] = spectral.laplacian_eigs_from_features(feats, K, w_dtype=w_dtype, affinity_mode=mode)
        assert (out[key][2] > 0).all()
    for key, tol in (("u16", 3e-6), ("fused", 2e-5)):
        assert (out[key][0] - out["f32"][0]).abs().max().item() < tol, key
        a, c = out[key][1].double(), out["f32"][1].double()
        cos = (a * c).sum(-1) / (a.norm(dim=-1) * c.norm(dim=-1))
        assert (1 - cos).max().item() < tol, (key, (1 - cos).max().item())   # signed cosine: same sign-rule decisions
        assert ((a.norm(dim=-1) / c.norm(dim=-1)) - 1).abs().max().item() < 1e-4   # same normalisation (v^T D v = 1)


@pytest.mark.parametrize("mode", ["fused", "split", "fp32"])
def test_eigs_batch_against_oracle(mode):
    """A batch of different images in ONE launch vs the scipy oracle image by image (both affinity builds)."""
    n, d, K, b = 400, 384, 5, 9
    feats = np.stack([synthetic.synthetic_features("blobs", n, d, 900 + i, (20, 20)) for i in range(b)])
    ev, vec, info = spectral.laplacian_eigs_from_features(torch.from_numpy(feats).to(DEV), K, affinity_mode=mode)
    assert (info > 0).all()
    for i in range(b):
        lam, v, ext, _ = oracle_target(torch.from_numpy(feats[i])[None], K)
        # fused build: the features are rounded to f16 on their way into the MFMAs - eigenvalues move by up to ~6e-5
        # (measured), eigenvectors stay inside the 1e-4 bound that check_eigs applies to every mode alike
        check_eigs(vec[i].cpu().numpy(), ev[i].cpu().numpy(), v.numpy(), lam.numpy(), what=f"img{i}",
                   d=build_w64(feats[i])[1], ext=ext, lam_tol=1e-4 if mode == "fused" else 1e-5)


@pytest.mark.parametrize("n,d,K", [(16, 32, 5), (12, 32, 3), (70, 64, 1), (70, 64, 2), (196, 384, 20), (333, 96, 7),
                                   (65, 32, 3), (80, 64, 4), (784, 384, 5), (128, 64, 4)])   # edge strips of 1 / 16 / 16 columns, none
def test_eigs_edge_shapes_against_fp64(n, d, K):
    feats = synthetic.synthetic_features("random", n, d, 4000 + n + K)
    lam64, v64 = spectral_ref.dense_f64_eigs(feats, min(K + 8, n))
    ev, vec, info = spectral.laplacian_eigs_from_features(torch.from_numpy(feats)[None].to(DEV), K)
    assert info.item() > 0
    check_eigs(vec[0].cpu().numpy(), ev[0].cpu().numpy(), v64[:K], lam64[:K], what=f"n{n}K{K}", d=build_w64(feats)[1],
               ext=(lam64, v64))


@pytest.mark.parametrize("n,K,ncv", [(15, 3, 0), (33, 4, 13), (70, 6, 21), (200, 5, 17), (130, 12, 33), (900, 5, 19)])
def test_eigs_odd_krylov_dimensions_against_fp64(n, K, ncv):
    """Odd projected dimensions: the dummy index of the Jacobi pairing, in the wave-scope solve that runs beside the W
    stream (convergence checks) and in the workgroup-scope one (end of a cycle, restarts with a small ncv); N = ncv is
    the breakdown case.  Against dense fp64 on the same features."""
    rng = np.random.default_rng(100 + n)
    base = rng.normal(size=(n, 32)).astype(np.float32)       # the affinity kernels take feature dims that are multiples of 32
    feats = base + 0.6 * rng.normal(size=(1, 32)).astype(np.float32)
    lam64, v64 = spectral_ref.dense_f64_eigs(feats, min(n - 1, K + 6))
    ev, vec, info = spectral.laplacian_eigs_from_features(torch.from_numpy(feats)[None].to(DEV), K, ncv=ncv,
                                                          affinity_mode="split", retry=False)
    assert info.item() > 0
    check_eigs(vec[0].cpu().numpy(), ev[0].cpu().numpy(), v64[:K], lam64[:K], what=f"n{n}K{K}ncv{ncv}",
               d=build_w64(feats)[1], ext=(lam64, v64))


def test_eigs_rejects_bad_arguments_and_reports_nonconvergence():
    feats = torch.from_numpy(synthetic.synthetic_features("random", 900, 384, 201, (30, 30)))[None].to(DEV)
    with pytest.raises(ValueError):
        spectral.laplacian_eigs_from_features(feats[:, :5], 5)
    with pytest.raises(spectral.EigsNotConverged):
        spectral.laplacian_eigs_from_features(feats, 5, max_restarts=1, retry=False)
    ev, vec, info = spectral.laplacian_eigs_from_features(feats, 5, max_restarts=1, strict=False, retry=False)
    assert info.item() < 0 and torch.isfinite(vec).all()
    # default behaviour: the starved image is re-solved with a bigger Krylov space and converges
    ev, vec, info = spectral.laplacian_eigs_from_features(feats, 5, max_restarts=1)
    assert info.item() > 0
    g = np.load(HERE / "golden" / "eigs_g2_random_900.npz")
    check_eigs(vec[0].cpu().numpy(), ev[0].cpu().numpy(), g["eigenvectors"], g["eigenvalues"], what="retry",
               ext=golden_ext(g))
    w = hip.affinity(hip.normalize_rows(feats))
    with pytest.raises(hip.HipLibraryError):
        hip.laplacian_eigs(w, 900, 40, ncv=30)  # Krylov dimension too small for K


def test_eigs_large_image_properties():
    """BASELINE config 5's largest shape (N = 6400, K = 20): too slow for scipy in a test; size-independent
    properties instead - generalized residual, D-orthonormality, ascending eigenvalues, constant v0."""
    n, d, K = 6400, 768, 20
    feats = torch.from_numpy(synthetic.synthetic_features("blobs", n, d, 77, (80, 80)))[None].to(DEV)
    ev, vec, info = spectral.laplacian_eigs_from_features(feats, K)
    assert info.item() > 0
    x = F.normalize(feats[0].double(), dim=-1)
    w = (x @ x.T).clamp_min(0)
    dd = w.sum(1)
    v, lam = vec[0].double(), ev[0].double()
    res = (dd[None] * v - v @ w) - lam[:, None] * dd[None] * v     # (D - W) v - lambda D v, row form
    assert (res.norm(dim=1) / dd.sqrt().norm()).max().item() < 1e-4
    gram = (v * dd[None]) @ v.T
    assert (gram - torch.eye(K, device=DEV, dtype=torch.float64)).abs().max().item() < 1e-4
    assert (lam[1:] >= lam[:-1] - 1e-6).all() and abs(lam[0].item()) < 1e-5
    assert (v[0].std() / v[0].mean().abs()).item() < 1e-3


def test_sign_rule_kernel():
    v = torch.tensor([[1.0, 1.0, 1.0, -1.0], [1.0, 1.0, -1.0, -1.0], [1.0, 1.0, 1.0, 1.0],
                      [-1.0, -1.0, -1.0, 1.0], [0.0, 1.0, 1.0, 1.0]])
    out = hip.sign_rule_(v.clone().to(DEV)).cpu()
    assert torch.equal(out, spectral_ref.ref_sign_rule(v.clone()))
    g = torch.Generator().manual_seed(0)
    big = torch.randn(6, 901, generator=g)
    assert torch.equal(hip.sign_rule_(big.clone().to(DEV)).cpu(), spectral_ref.ref_sign_rule(big.clone()))


# ----------------------------------------------------------------------------- the other _extract_eig branches
MODE_FILES = sorted(glob.glob(str(HERE / "golden" / "modes_*.npz")))


@pytest.mark.parametrize("path", MODE_FILES, ids=lambda p: p.split("modes_")[-1][:-4])
def test_other_branches_match_reference_goldens(path):
    """which_matrix='affinity' / 'affinity_svd' and lapnorm=False through the same Lanczos kernel, against the
    reference's own outputs (ordering quirks included: 'affinity' saves ascending values with DESCENDING vectors)."""
    g = np.load(path)
    feats = synthetic.synthetic_features(str(g["kind"]), int(g["n"]), int(g["d"]), int(g["seed"]), tuple(g["hw"]))
    kw = eval(str(g["kwargs"]))
    K = int(g["K"])
    problem = kw["which_matrix"] if kw["which_matrix"] != "laplacian" else (
        "laplacian" if kw.get("lapnorm", True) else "laplacian_unnormalized")
    up = None
    if "image_downsample_factor" in kw:  # P = 16 features resized to the (16 / f) x finer grid (extract.py:179-188)
        hp, wp = (int(v) for v in g["hw"])
        f = 16 // kw["image_downsample_factor"]
        up = ((hp, wp), (hp * f, wp * f))
    ev, vec, info = spectral.laplacian_eigs_from_features(torch.from_numpy(feats)[None].to(DEV), K, problem=problem,
                                                          threshold_at_zero=kw.get("threshold_at_zero", True),
                                                          normalize=kw.get("normalize", True), upsample=up)
    assert info.item() > 0
    lam, v = ev[0].cpu().numpy().astype(np.float64), vec[0].cpu().numpy()
    ref_lam, ref_v = np.asarray(g["eigenvalues"], np.float64), g["eigenvectors"]
    # MAGNITUDES, not just directions: the reference divides W by W.max() (extract.py:194), which rescales the
    # D-orthonormal vectors (lapnorm) or the eigenvalues (lapnorm=False) whenever the feature rows are not unit vectors
    ratio = np.linalg.norm(v, axis=1) / np.linalg.norm(ref_v, axis=1)
    assert np.abs(ratio - 1).max() < 2e-3, (path, ratio)
    scale = max(1.0, np.abs(ref_lam).max())
    tol = dict(lam_tol=2e-5 * scale, gap_tol=1e-4 * scale)
    if problem == "affinity":        # values ascending, vectors descending: flip the vectors to a common order
        check_eigs(v[::-1], lam, ref_v[::-1], ref_lam, what=path, **tol)
    elif problem == "affinity_svd":  # both descending: flip both for the ascending-order checker
        check_eigs(v[::-1], lam[::-1], ref_v[::-1], ref_lam[::-1], what=path, **tol)
    else:
        check_eigs(v, lam, ref_v, ref_lam, what=path, **tol)


# ----------------------------------------------------------------------------- exceptional dense path
@pytest.mark.parametrize("path", [p for p in EIG_FILES if "3600" not in p and "1600" not in p],
                         ids=lambda p: p.split("eigs_")[-1][:-4])
def test_dense_fallback_matches_reference_goldens(path):
    """spectral.dense_eigs - the per-image last resort behind the Lanczos kernel (the reference's second eigsh call,
    extract.py:228-229) and the K > 62 path - against the reference's own outputs, same bar."""
    feats, K, ref_lam, ref_vec, g = golden_case(path)
    ev, vec = spectral.dense_eigs(torch.from_numpy(feats)[None].to(DEV), K, True, True, "laplacian")
    check_eigs(vec[0].cpu().numpy(), ev[0].cpu().numpy(), ref_vec, ref_lam, what=path, d=build_w64(feats)[1],
               ext=golden_ext(g))
    for k in range(K):
        assert not (0.5 < (vec[0, k] > 0).float().mean().item() < 1.0)


def test_k_beyond_the_krylov_space_takes_the_dense_path(capsys):
    """The reference accepts any K < N; the Lanczos kernel holds at most 64 basis vectors (K <= 62)."""
    n, d, K = 196, 384, 70
    feats = synthetic.synthetic_features("blobs", n, d, 102, (14, 14))
    ev, vec, info = spectral.laplacian_eigs_from_features(torch.from_numpy(feats)[None].to(DEV), K)
    assert "dense fp64 solve" in capsys.readouterr().out and info.item() > 0 and tuple(vec.shape) == (1, K, n)
    lam64, v64 = spectral_ref.dense_f64_eigs(feats, K + 8)
    check_eigs(vec[0].cpu().numpy(), ev[0].cpu().numpy(), v64[:K], lam64[:K], what="K70", d=build_w64(feats)[1],
               ext=(lam64, v64))
    # the same K through the other branches keeps their conventions (ordering quirks, schema)
    for problem in ("affinity", "affinity_svd", "laplacian_unnormalized"):
        e, v, i = spectral.laplacian_eigs_from_features(torch.from_numpy(feats)[None].to(DEV), K, problem=problem)
        e5, v5, _ = spectral.laplacian_eigs_from_features(torch.from_numpy(feats)[None].to(DEV), 5, problem=problem)
        if problem == "affinity":      # ascending values: the 5 largest are the last 5; vectors are in descending order
            assert torch.allclose(e[0, -5:], e5[0], rtol=1e-4) and (1 - torch.nn.functional.cosine_similarity(
                v[0, :5], v5[0], dim=-1).abs()).max().item() < 1e-4
        else:
            assert torch.allclose(e[0, :5], e5[0], rtol=1e-4, atol=1e-4)
            assert (1 - torch.nn.functional.cosine_similarity(v[0, :4], v5[0, :4], dim=-1).abs()).max().item() < 1e-4


def test_starved_image_falls_back_per_image_without_aborting_the_batch(capsys, monkeypatch):
    """one unconverged image must not take the batch (or the other ranks) down.  With the restart budget
    forced to 1 and the retry's Krylov space capped the same way, the dense solve answers for that image only."""
    feats = np.stack([synthetic.synthetic_features("random", 900, 384, 201, (30, 30)),
                      synthetic.synthetic_features("blobs", 900, 384, 202, (30, 30))])
    real = hip.laplacian_eigs

    def starved(w, n, k, ncv=0, tol=0.0, max_restarts=0, **kw):   # every Lanczos launch: 12 basis vectors, one restart
        return real(w, n, k, ncv=12, tol=tol, max_restarts=1, **kw)

    monkeypatch.setattr(hip, "laplacian_eigs", starved)
    ev, vec, info = spectral.laplacian_eigs_from_features(torch.from_numpy(feats).to(DEV), 5, strict=True)
    out = capsys.readouterr().out
    assert "dense fp64 solve for them" in out and (info > 0).all()
    for i, name in enumerate(("g2_random_900", "g2_blobs_900")):
        g = np.load(HERE / "golden" / f"eigs_{name}.npz")
        check_eigs(vec[i].cpu().numpy(), ev[i].cpu().numpy(), g["eigenvectors"], g["eigenvalues"], what=name,
                   ext=golden_ext(g))


# ----------------------------------------------------------------------------- the reference-side binding of INTEGRATION.md
def test_integration_md_binding_stub_runs_against_the_library():
    """INTEGRATION.md section B shows the ctypes stub a maintainer of the reference would add (`extract/dss_binding.py`).
    Execute exactly that text against the in-tree library and hold its output to the reference's own golden."""
    import re
    from pathlib import Path
    from tests.util import check_eigs, golden_case, golden_ext, build_w64

    repo = Path(__file__).resolve().parents[1]
    text = (repo / "INTEGRATION.md").read_text()
    block = re.search(r"```python\n(# extract/dss_binding\.py.*?)```", text, re.S).group(1)
    assert 'ctypes.CDLL("libdss_hip.so")' in block
    block = block.replace('ctypes.CDLL("libdss_hip.so")', f'ctypes.CDLL("{hip.LIB_PATH}")')
    ns = {}
    exec(compile(block, "INTEGRATION.md#dss_binding", "exec"), ns)
    feats, K, lam, vec, g = golden_case(repo / "tests" / "golden" / "eigs_g2_blobs_900.npz")
    val, v = ns["laplacian_eigs"](torch.from_numpy(feats).to(DEV), K)
    assert val.device.type == "cpu" and tuple(v.shape) == (K, feats.shape[0])
    check_eigs(v.numpy(), val.numpy(), vec, lam, what="INTEGRATION.md stub", d=build_w64(feats)[1], ext=golden_ext(g))
