"""GPU: per-tensor plane scales of the split numerics (round 5; gp_vit_forward_split2, vit.py: calibrate_plane_scales).

The plane path keeps four activation tensors per ViT layer as f16 hi / lo planes of s x.  s was a compile-time 8 (|x| < 8190) and ONE
outlier anywhere moved the whole ViT to the two-accumulator 128 x 128 kernels (-43 %, BENCH_r04 `split128`).  Now s is a power of two per
(layer, tensor), picked from a calibration pass.  Checked here:
  * stage level: the plane epilogues 6 / 7 and the attention kernel with s != 8 produce the same VALUES (s undone) as with s = 8, to
    the f16 subnormal floor of the lo plane; the calibration launch records exactly max |x|;
  * ViT level: a ViT-L stand-in with planted DINOv2-like outliers (synthetic.plant_dinov2_outliers) TRIPS the default planes, is
    calibrated in one pass, then runs clean on the same kernels with features as close to the float64 forward as torch's own f32;
    clean weights calibrate to all-8 = bit-identical to the uncalibrated path;
  * GigaPose level: onboarding calibrates (no warning, no fallback), a crop beyond the calibrated range re-calibrates with a warning,
    NaN raises, and the wide-kernel fallback still exists behind it.
Arithmetic restated: HF modeling_dinov2.py:207-229, 272-299, 342-380 (the float64 oracle IS transformers.Dinov2Model in double)."""
import ctypes
import warnings

import numpy as np
import pytest
import torch

from gigapose_amd import _lib

from gigapose_testing import factory
from gigapose_testing import synthetic as syn
from gigapose_amd.vit import Dinov2ViT

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def clean_status():
    _lib.status_word(DEV).zero_()
    yield
    torch.cuda.synchronize()
    _lib.status_word(DEV).zero_()


def split_planes(x, scale):
    hi = torch.empty(x.shape, dtype=torch.float16, device=DEV)
    lo = torch.empty_like(hi)
    _lib.call("gp_split_planes", _lib.ptr(x), ctypes.c_size_t(x.numel()), _lib.f(scale), _lib.ptr(hi), _lib.ptr(lo), _lib.stream_ptr())
    return hi, lo


def scaled_gemm(A, Bm, epi, bias, b_scale, out_scale_planes, j_valid=None, amax=None):
    lib = _lib.lib()
    lib.gp_gemm_split256_workspace_bytes.restype = ctypes.c_size_t
    nb = lib.gp_gemm_split256_workspace_bytes()
    ws = torch.zeros(nb // 4, device=DEV)
    (ahi, alo), (bhi, blo) = split_planes(A, 64.0), split_planes(Bm, b_scale)
    I, J, K = A.shape[0], Bm.shape[0], A.shape[1]
    ohi = torch.zeros(J, I, dtype=torch.float16, device=DEV)
    olo = torch.zeros_like(ohi)
    _lib.call("gp_gemm_planes256_scaled", _lib.ptr(ahi), _lib.ptr(alo), _lib.ptr(bhi), _lib.ptr(blo), _lib.ptr(None), _lib.i(0), _lib.ptr(ohi),
              _lib.ptr(olo), _lib.i(I), _lib.i(I), _lib.i(J), _lib.i(J if j_valid is None else j_valid), _lib.i(K), _lib.i(epi), _lib.ptr(bias),
              _lib.ptr(None), _lib.ptr(None), _lib.i(0), _lib.f(1.0 / (64.0 * b_scale)), _lib.f(out_scale_planes), _lib.ptr(amax), _lib.ptr(ws),
              ctypes.c_size_t(nb), _lib.stream_ptr())
    torch.cuda.synchronize()
    return (ohi.double() + olo.double()) / out_scale_planes


@pytest.mark.parametrize("epi", [6, 7])
@pytest.mark.parametrize("s_in,s_out", [(8.0, 8.0), (2.0, 0.5), (0.25, 1.0)])
def test_plane_epilogues_with_other_scales(epi, s_in, s_out):
    """Tiles + ragged strip (J_valid = 4096 + 130): the values the planes hold do not depend on the scales beyond the lo plane's
    subnormal floor (2^-25 / s per element), the recorded amax is exactly max |x|, and s = 8 is the old entry point bit for bit."""
    torch.manual_seed(3)
    I, J, K, jv = 1024, 4352, 128, 4096 + 130
    A = torch.randn(I, K, device=DEV) * 0.05
    Bm = torch.randn(J, K, device=DEV)
    bias = torch.randn(I, device=DEV)
    amax = torch.zeros(1, device=DEV)
    got = scaled_gemm(A, Bm, epi, bias, s_in, s_out, jv, amax)[:jv]
    x = (A.double() @ Bm.double().t()).t()[:jv] + bias.double()[None, :]
    ref = torch.nn.functional.gelu(x) if epi == 6 else x
    mag = (A.double().abs() @ Bm.double().abs().t()).t()[:jv] + bias.double().abs()[None, :]
    err = ((got - ref).abs() / mag).max().item()
    _lib.check_status()
    # f32 epilogue arithmetic (one rounding of the affine step, GELU's polynomial) + the planes' 22 bits
    assert err < 3e-6, err
    assert abs(amax.item() - got.abs().max().item()) <= 2.0 ** -20 * amax.item()       # the record is max |x| of what was written
    if (s_in, s_out) == (8.0, 8.0):
        from test_gpu_split import planes256_gemm

        ohi, olo = planes256_gemm(A, Bm, epi, bias=bias, j_valid=jv)
        old = ((ohi.double() + olo.double()) / 8.0)[:jv]
        assert torch.equal(old, got), "plane scale 8 must be bit-identical to gp_gemm_planes256_ragged"


def test_plane_epilogue_range_guard_follows_the_scale():
    """3e4 in a bias: beyond the x 8 planes (2.4e5 > 65504: flagged), inside the x 1 planes (clean)."""
    torch.manual_seed(4)
    I, J, K = 1024, 4096, 128
    A = torch.randn(I, K, device=DEV) * 0.05
    Bm = torch.randn(J, K, device=DEV)
    bias = torch.zeros(I, device=DEV)
    bias[17] = 3.0e4
    scaled_gemm(A, Bm, 7, bias, 8.0, 8.0)
    with pytest.raises(_lib.GigaPoseHipError, match="range of the f16 planes"):
        _lib.check_status()
    got = scaled_gemm(A, Bm, 7, bias, 8.0, 1.0)
    _lib.check_status()
    assert abs(got[:, 17].mean().item() - 3.0e4) < 5.0


@pytest.mark.parametrize("scale", [8.0, 1.0, 0.125])
def test_attention_with_other_plane_scales(scale):
    """attention_split_kernel on q | k | v planes that carry `scale` (a V channel of 9e3 planted: beyond the x 8 planes): vs float64
    softmax attention on the plane values."""
    torch.manual_seed(11)
    B, H = 3, 6
    C = 64 * H
    M = B * 257
    Mpad = (M + 255) // 256 * 256
    qkv = torch.zeros(Mpad, 3 * C, device=DEV)
    qkv[:M] = torch.randn(M, 3 * C, device=DEV) * torch.tensor([1.5] * (2 * C) + [1.0] * C, device=DEV)
    if scale < 8.0:
        qkv[:M, 2 * C + 5] = 9.0e3 + qkv[:M, 2 * C + 5]
    hi, lo = split_planes(qkv, scale)
    ohi = torch.zeros(Mpad, C, dtype=torch.float16, device=DEV)
    olo = torch.zeros_like(ohi)
    _lib.call("gp_attention_split_scaled", _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(ohi), _lib.ptr(olo), _lib.i(B), _lib.i(H), _lib.i(C), _lib.i(Mpad),
              _lib.f(scale), _lib.stream_ptr())
    torch.cuda.synchronize()
    got = ((ohi.double() + olo.double()) / scale)[:M].view(B, 257, H, 64)
    x = ((hi.double() + lo.double()) / scale)[:M].view(B, 257, 3, H, 64)
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v).permute(0, 2, 1, 3)
    small = [c for c in range(64 * H) if c != 5]
    g2, r2 = got.reshape(B, 257, C)[..., small], ref.reshape(B, 257, C)[..., small]
    err = (g2 - r2).abs().max().item() / r2.abs().max().item()
    err5 = (got.reshape(B, 257, C)[..., 5] - ref.reshape(B, 257, C)[..., 5]).abs().max().item() / ref.reshape(B, 257, C)[..., 5].abs().max().item()
    print(f"split attention, plane scale {scale}: max |err| / max |ref| = {err:.2e} (the planted channel, relative to itself: {err5:.2e})")
    assert err < 4e-6 and err5 < 2e-6
    # run-to-run determinism (round 6: an inline-asm v_max3_f32 on accumulator registers fresh from the matrix core skipped hipcc's
    # MFMA -> VALU hazard padding and read stale values now and then): five more launches, all bit-equal to the first
    for _ in range(5):
        r2_ = torch.zeros_like(ohi), torch.zeros_like(olo)
        _lib.call("gp_attention_split_scaled", _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(r2_[0]), _lib.ptr(r2_[1]), _lib.i(B), _lib.i(H), _lib.i(C), _lib.i(Mpad),
                  _lib.f(scale), _lib.stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(r2_[0], ohi) and torch.equal(r2_[1], olo), "attention_split_kernel is not deterministic"
    if scale == 8.0:
        o2 = torch.zeros_like(ohi), torch.zeros_like(olo)
        _lib.call("gp_attention_split_scaled", _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(o2[0]), _lib.ptr(o2[1]), _lib.i(B), _lib.i(H), _lib.i(C), _lib.i(Mpad),
                  _lib.f(8.0), _lib.stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(o2[0], ohi) and torch.equal(o2[1], olo)


# ---------------------------------------------------------------------------------------------------------------- ViT level
def hf_model(dim, depth, heads, seed):
    from transformers import Dinov2Config, Dinov2Model

    hf = Dinov2Model(Dinov2Config(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads, image_size=224, patch_size=14)).eval()
    return syn.fill_state_dict(hf, seed)


def unit(h, B):
    return torch.nn.functional.normalize(h[:, 1:].permute(0, 2, 1), dim=1).reshape(B, h.shape[-1], 16, 16)


def test_clean_weights_calibrate_to_the_default_scales_bit_identically():
    vit = syn.fill_state_dict(Dinov2ViT(1024, 3, 16), 7).eval().to(DEV).set_numerics("split")
    x = torch.from_numpy(np.random.RandomState(3).standard_normal((64, 3, 224, 224)).astype(np.float32)).to(DEV)
    before = vit.patch_features(x).clone()
    assert vit.calibrate_plane_scales(x) is False and vit.plane_scales is None        # every tensor keeps x 8
    assert vit.plane_amax is not None and (vit.plane_amax > 0).all() and vit.plane_amax.max() < 2047.0
    assert torch.equal(vit.patch_features(x), before)
    vit.plane_scales = [8.0] * 12                                                       # explicit all-8 array: the same launches
    assert torch.equal(vit.patch_features(x), before)
    _lib.check_status()


def test_vit_large_with_planted_outliers_runs_on_the_plane_path_after_one_calibration_pass():
    """ViT-L/14 stand-in (24 layers, HF weights) with DINOv2-like planted outliers, 64 crops (the plane path the benchmark times).
    Default planes: the guard trips.  After ONE calibration pass over the same crops: clean status, only the offending tensors carry
    a smaller scale, and the unit-norm features are as close to the float64 forward of the same network as torch's own f32 forward."""
    B = 64
    hf = syn.plant_dinov2_outliers(hf_model(1024, 24, 16, seed=90))
    vit = Dinov2ViT.from_hf(hf).to(DEV).set_numerics("split")
    x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(91)).to(DEV)
    vit.patch_features(x)
    torch.cuda.synchronize()
    with pytest.raises(_lib.GigaPoseHipError, match="range of the f16 planes"):
        _lib.check_status()
    assert vit.calibrate_plane_scales(x) is True
    _lib.take_status()                                                                  # the calibration pass's own range bits (it measures them)
    rep = vit.plane_scale_report()
    print("plane scales after calibration (tensor: (max |x|, scale)):", rep)
    kinds = {k.split(".")[1] for k in rep}
    assert {"ln2", "qkv", "gelu"} <= kinds, kinds                                       # the three planted kinds were found
    assert len(rep) <= 24 and all(s < 8.0 and a * s * 4.0 <= 65504.0 for a, s in rep.values())
    assert {"L6.gelu", "L18.gelu", "L9.qkv"} <= set(rep)
    mine = vit.patch_features(x).double()
    torch.cuda.synchronize()
    _lib.check_status()                                                                 # clean: every GEMM stayed on the plane kernels
    assert vit.split_gemm == "256"
    hf = hf.to(DEV)
    with torch.no_grad():
        r32 = hf(pixel_values=x, output_hidden_states=True).hidden_states[-1]
        r64 = hf.double()(pixel_values=x.double(), output_hidden_states=True).hidden_states[-1]
    f64, f32 = unit(r64, B), unit(r32.double(), B)
    e_m, e_r = (mine - f64).abs(), (f32 - f64).abs()
    rms = lambda e: float((e ** 2).mean().sqrt())
    print(f"ViT-L with planted outliers, {B} crops, calibrated planes: unit-norm features vs float64 (rms {rms(f64):.3e}): ours max {float(e_m.max()):.2e} "
          f"rms {rms(e_m):.2e} | torch f32 (GPU) max {float(e_r.max()):.2e} rms {rms(e_r):.2e}")
    assert rms(e_m) < 2.0 * rms(e_r) + 1e-8 and float(e_m.max()) < 3.0 * float(e_r.max()) + 1e-6
    # a second pass over the same inputs changes nothing; new weights drop the calibration
    assert vit.calibrate_plane_scales(x) is False
    _lib.take_status()
    vit.invalidate()
    assert vit.plane_scales is None and vit.plane_amax is None


# ---------------------------------------------------------------------------------------------------------------- GigaPose level
def _gigapose(vit, k=3):
    from test_gpu_guards import _gigapose_with_vit

    return _gigapose_with_vit(vit, k)


def test_onboarding_calibrates_and_a_far_crop_recalibrates():
    from test_gpu_e2e import make_batch

    vit = syn.plant_dinov2_outliers(syn.fill_state_dict(Dinov2ViT(1024, 2, 16), 11).eval()).to(DEV)
    tset = factory.TemplateSet(1, 64, seed=80)                    # 64 templates: the chunk takes the plane path
    q = tset.crops(81, 64, "cpu")
    as_np = lambda d: {n: (v.numpy().copy() if torch.is_tensor(v) else v) for n, v in d.items()}
    model = _gigapose(vit)
    model.template_datasets = {"syn": tset}
    with warnings.catch_warnings():
        warnings.simplefilter("error")                            # calibrated BEFORE the first feature is computed: nothing trips
        model.eval_retrieval(make_batch(as_np(q)), 0, "syn")
    _lib.check_status()
    assert vit.split_gemm == "256" and vit.plane_scale_report(), "the planted tensors must carry their own scale"
    clean_report = dict(vit.plane_scale_report())
    # Inputs that drive an outlier beyond what the calibration set showed (simulated: forget the calibration, so that every tensor is
    # back at x 8 while the weights still hold the planted outliers): the guard trips inside eval_retrieval, the scales are
    # re-calibrated on the offending crops, the call completes on the same kernels -- with a warning that names the tensors
    vit.plane_scales, vit.plane_amax = None, np.ones_like(vit.plane_amax)
    with pytest.warns(RuntimeWarning, match="re-calibrated"):
        model.eval_retrieval(make_batch(as_np(q)), 1, "syn")
    _lib.check_status()
    assert vit.split_gemm == "256" and set(vit.plane_scale_report()) == set(clean_report)
    p = model.last_predictions
    assert torch.isfinite(p.pred_poses).all()
    # NaN is not a range problem: raises
    bad = as_np(q)
    bad["tar_img"][2, 0, 50, 50] = float("nan")
    with pytest.raises(_lib.GigaPoseHipError):
        model.eval_retrieval(make_batch(bad), 2, "syn")
    _lib.take_status()
