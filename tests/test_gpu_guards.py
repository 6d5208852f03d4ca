"""GPU guard rails: nothing on the product path may turn into silent garbage.

  * split numerics: an activation outside the f16 planes' range (|x| >= 8190) or a non-finite one raises at the next
    host synchronisation; large-but-legal activations (a token x 500) stay accurate;
  * a LOST stream-K accumulator hand-over (forced through the test hook) raises instead of returning a wrong tile;
  * labels outside the onboarded bank raise (host labels: IndexError as in the reference, gigaPose.py:520; device labels:
    the kernels clamp + flag, no out-of-bounds read).
"""
import numpy as np
import pytest
import torch

from gigapose_amd import _lib

from gigapose_testing import factory
from gigapose_testing import synthetic as syn
from gigapose_amd.vit import Dinov2ViT

pytestmark = pytest.mark.gpu
DEV = "cuda"


def small_vitl(seed=7, depth=2):
    """ViT-L width, two blocks: at B = 64 every GEMM takes the plane kernels (the benchmark path), in seconds."""
    return syn.fill_state_dict(Dinov2ViT(1024, depth, 16), seed).eval().to(DEV)


def images(B=64, seed=3):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal((B, 3, 224, 224)).astype(np.float32)).to(DEV)


@pytest.fixture(autouse=True)
def clean_status():
    _lib.status_word(DEV).zero_()
    yield
    torch.cuda.synchronize()
    _lib.status_word(DEV).zero_()


def test_one_channel_1e4_trips_the_range_guard():
    vit = small_vitl().set_numerics("split")
    x = images()
    vit.patch_features(x)
    torch.cuda.synchronize()
    _lib.check_status()                                   # clean weights: nothing flagged
    with torch.no_grad():
        vit.blocks[0].mlp.fc1.bias[7] = 1.0e4             # GELU(1e4) = 1e4 -> 8e4 in the x 8 planes: beyond f16
    vit.invalidate()
    vit.patch_features(x)
    torch.cuda.synchronize()
    with pytest.raises(_lib.GigaPoseHipError, match="range of the f16 planes"):
        _lib.check_status()
    _lib.check_status()                                   # reading clears the word
    # the chain mode has no such limit: same weights, finite features, no flag
    vit.set_numerics("chain")
    f = vit.patch_features(x)
    torch.cuda.synchronize()
    _lib.check_status()
    assert torch.isfinite(f).all()


def test_nan_input_trips_the_range_guard():
    vit = small_vitl().set_numerics("split")
    x = images()
    x[5, 1, 100, 100] = float("nan")
    vit.patch_features(x)
    torch.cuda.synchronize()
    with pytest.raises(_lib.GigaPoseHipError, match="not finite"):
        _lib.check_status()


def test_token_x500_stays_legal_and_accurate():
    """A massive-but-representable activation (one crop scaled by 500: patch-embedding outputs and the residual stream
    of ALL its tokens grow 500-fold) is inside the planes' range; split must still agree with chain."""
    x = images()
    x[9] *= 500.0
    vit = small_vitl()
    fc = vit.set_numerics("chain").patch_features(x).clone()
    fs = vit.set_numerics("split").patch_features(x)
    torch.cuda.synchronize()
    _lib.check_status()
    err = (fs - fc).abs().max().item()
    print("unit-norm features, crop 9 scaled x 500: max |split - chain| = %.2e (crop 9 alone %.2e)" % (err, (fs[9] - fc[9]).abs().max().item()))
    assert err < 2e-6


@pytest.mark.probes
def test_lost_handoff_raises():
    """260 tiles on 256 slots (a fully tiled J = 16640): half the slots hand accumulator fragments over.  With the
    publishes dropped (test hook) every waiter must time out, flag it, and the host must raise; afterwards a clean
    launch reproduces the reference result."""
    from test_gpu_split import planes256_gemm

    torch.manual_seed(5)
    A = torch.randn(1024, 64, device=DEV) * 0.05
    Bm = torch.randn(16640, 64, device=DEV)
    ref = planes256_gemm(A, Bm, 0)
    _lib.check_status()
    _lib.lib().gp_gemm_planes256_set_dp(1 | 2)            # test hook: head fragments are never published
    try:
        with pytest.raises(AssertionError):               # the per-scratch error word (planes256_gemm asserts it is clear)
            planes256_gemm(A, Bm, 0)
    finally:
        _lib.lib().gp_gemm_planes256_set_dp(1)
    with pytest.raises(_lib.GigaPoseHipError, match="hand-over"):
        _lib.check_status()                               # and the device status word the product path reads
    assert torch.equal(planes256_gemm(A, Bm, 0), ref)
    _lib.check_status()


def test_labels_outside_the_bank():
    model = factory.build_model("dinov2_vits14", k=3, device=DEV, seed=2)
    tset = factory.TemplateSet(2, 5, seed=70)
    model.template_datasets = {"syn": tset}
    model.set_template_data("syn")
    q = tset.crops(71, 4, DEV)
    good = model.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn")
    torch.cuda.synchronize()
    _lib.check_status()
    for bad in (0, 3):
        labels = q["labels"].clone()
        labels[1] = bad
        with pytest.raises(IndexError):                   # host labels: checked before anything is launched
            model.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], labels, "syn")
        out = model.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], labels.to(DEV), "syn")  # device labels
        torch.cuda.synchronize()
        with pytest.raises(_lib.GigaPoseHipError, match="outside the onboarded bank"):
            _lib.check_status()
        keep = [0, 2, 3]                                  # the other detections are untouched
        assert torch.equal(out.id_src[keep], good.id_src[keep]) and torch.equal(out.pred_poses[keep], good.pred_poses[keep])


def _gigapose_with_vit(vit, k=3):
    import tempfile

    from gigapose_amd.ae_net import AENet
    from gigapose_amd.gigaPose import GigaPose
    from gigapose_amd.ist_net import ISTNet, Regressor, ResNet
    from gigapose_amd.matching import LocalSimilarity

    ist = syn.fill_state_dict(ISTNet("resnet", ResNet(dict(factory.IST_CFG)), Regressor(256, 256, True, True), 64), 9)
    model = GigaPose("large", AENet("dinov2_vitl14", vit, 1024, 64), ist, None, LocalSimilarity(k=k, sim_threshold=0.5, patch_threshold=3),
                     None, 1000, tempfile.mkdtemp(), max_num_dets_per_forward=4).eval().to(DEV)
    return model.set_numerics("split")


def test_range_trip_falls_back_to_the_wide_kernels_automatically():
    """a checkpoint whose activations leave the x 8 planes' range (|x| >= 8190; planted: one fc1 bias
    of 1e4) must not make the drop-in fail -- and (round 5) must not cost the fast kernels either.
    (a) Default: onboarding calibrates the per-tensor plane scales on the templates (tests/test_gpu_plane_scales.py), the GELU tensor
        of that layer gets a smaller power of two, nothing trips, every GEMM stays on the 256 x 256 kernels; the predictions agree
        with a model built on the wide 128 x 128 kernels (both f32-class).
    (b) The remedy behind it still exists: with the calibration disabled the guard trips, GigaPose moves the ViT to the
        two-accumulator 128 x 128 kernels (range 65504), re-onboards and runs again -- the result equals a model that was told
        Dinov2ViT.set_split_gemm("128") from the start bit for bit, the status word is clean, a warning says what happened.
    A NaN still raises."""
    from test_gpu_e2e import make_batch

    def planted():
        vit = small_vitl(seed=11)
        with torch.no_grad():
            vit.blocks[0].mlp.fc1.bias[7] = 1.0e4
        vit.invalidate()
        return vit

    tset = factory.TemplateSet(1, 12, seed=80)
    q = tset.crops(81, 16, "cpu")
    batch = make_batch({n: (v.numpy() if torch.is_tensor(v) else v) for n, v in q.items()})
    want = _gigapose_with_vit(planted().set_split_gemm("128"))
    want.template_datasets = {"syn": tset}
    want.eval_retrieval(batch, 0, "syn")
    ref = {n: v.cpu() for n, v in want.last_predictions.tensors.items()}
    import warnings

    # (a) calibrated planes: no trip, no warning, the fast kernels
    model = _gigapose_with_vit(planted())
    model.template_datasets = {"syn": tset}
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        model.eval_retrieval(batch, 0, "syn")
    _lib.check_status()
    vit = model.ae_net.dinov2_model
    assert vit.split_gemm == "256" and "L0.gelu" in vit.plane_scale_report() and vit.plane_scale_report()["L0.gelu"][1] <= 1.0
    # the two f32-class kernel families agree on the features (the planted 1e4 makes every token's feature nearly the same vector, so
    # the template ranking of this fixture is a field of near-ties: the comparison that means something is the features themselves)
    x = q["tar_img"].to(DEV)
    f_cal, f_wide = model.ae_net(x), want.ae_net(x)
    torch.cuda.synchronize()
    _lib.check_status()
    d = (f_cal - f_wide).abs().max().item()
    print(f"planted fc1 bias 1e4: unit-norm ViT features, calibrated 256 x 256 planes vs 128 x 128 two-accumulator kernels: max |diff| {d:.2e}")
    assert d < 3e-6 and torch.isfinite(model.last_predictions.pred_poses).all()
    # (b) calibration disabled: the wide-kernel fallback
    model = _gigapose_with_vit(planted())
    model._needs_calibration = lambda: False
    model._calibrate_planes = lambda images: False
    assert model.ae_net.dinov2_model.split_gemm == "256"
    model.template_datasets = {"syn": tset}
    with pytest.warns(RuntimeWarning, match="falling back"):
        model.eval_retrieval(batch, 0, "syn")                      # onboarding trips the guard -> widen -> onboard again -> predict
    assert model.ae_net.dinov2_model.split_gemm == "128"
    _lib.check_status()                                            # clean afterwards
    got = {n: v.cpu() for n, v in model.last_predictions.tensors.items()}
    for n in ref:
        assert torch.equal(ref[n], got[n]), f"{n} differs from the model built with the wide kernels"
    # a second batch runs without another fallback
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        model.eval_retrieval(batch, 1, "syn")
    # the trip can also come from a crop (bank onboarded cleanly): a fresh model, clean onboarding, then a huge crop
    model = _gigapose_with_vit(small_vitl(seed=11))
    model.template_datasets = {"syn": tset}
    model.set_template_data("syn")
    big = {n: (v.numpy().copy() if torch.is_tensor(v) else v) for n, v in q.items()}
    big["tar_img"][3] *= 1.0e5
    with pytest.warns(RuntimeWarning, match="falling back"):
        model.eval_retrieval(make_batch(big), 2, "syn")
    _lib.check_status()
    # not a range problem: NaN pixels trip the guard again after the fallback -> raises
    bad = {n: (v.numpy().copy() if torch.is_tensor(v) else v) for n, v in q.items()}
    bad["tar_img"][2, 0, 50, 50] = float("nan")
    with pytest.raises(_lib.GigaPoseHipError):
        model.eval_retrieval(make_batch(bad), 3, "syn")
