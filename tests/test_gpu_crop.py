"""GPU parity: detection pre-processing kernels (gp_crop_resize_pad / gp_preprocess_detections) vs the reference
golden (tests/golden/crop.npz, written by the unmodified CropResizePad) and vs the numpy restatement on
randomised boxes.  Pixels and masks bit-exact; M to 1 ulp (the reference's 3x3 matmul order is BLAS-defined)."""
import os

import numpy as np
import pytest
import torch

from gigapose_testing import synthetic as syn
from oracle import crop_numpy

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.from_numpy(a).to(DEV)


def test_preprocess_matches_reference_golden(golden_dir):
    from gigapose_amd.crop import DetectionPreprocessor

    g = np.load(os.path.join(golden_dir, "crop.npz"))
    case = syn.detection_case(seed=int(g["seed"]))
    out = DetectionPreprocessor()(_t(case["rgb"]), _t(case["masks"]), _t(case["boxes"]), _t(case["im_id"]))
    np.testing.assert_array_equal(out["tar_mask"].cpu().numpy(), g["tar_mask"])
    np.testing.assert_array_equal(out["tar_img"].cpu().numpy().view(np.uint32), g["tar_img"].view(np.uint32))
    np.testing.assert_allclose(out["tar_M"].cpu().numpy(), g["M"], rtol=2e-7, atol=0)


@pytest.mark.parametrize("seed,H,W,D", [(5, 480, 640, 40), (6, 97, 131, 25), (7, 1080, 1920, 12)])
def test_crop_resize_pad_equals_restatement_on_random_boxes(seed, H, W, D):
    from gigapose_amd.crop import CropResizePad

    case = syn.detection_case(seed=seed, n_img=1, D=D, H=H, W=W)
    rs = np.random.RandomState(seed)
    images = rs.standard_normal((D, 5, H, W)).astype(np.float32)     # C = 5: not tied to RGBA
    ref_img, ref_M = crop_numpy.crop_resize_pad(images, case["boxes"])
    out = CropResizePad(target_size=224)(_t(case["boxes"]), _t(images))
    np.testing.assert_array_equal(out["images"].cpu().numpy().view(np.uint32), ref_img.view(np.uint32))
    np.testing.assert_allclose(out["M"].cpu().numpy(), ref_M, rtol=2e-7, atol=0)


def test_other_target_size_and_fused_equals_two_step():
    from gigapose_amd.crop import CLIP_MEAN, CLIP_STD, CropResizePad, DetectionPreprocessor

    case = syn.detection_case(seed=9, n_img=3, D=16, H=240, W=320)
    ref_img, ref_mask, ref_M = crop_numpy.preprocess_detections(case["rgb"], case["masks"], case["boxes"], case["im_id"],
                                                               target=112)
    out = DetectionPreprocessor(target_size=112)(_t(case["rgb"]), _t(case["masks"]), _t(case["boxes"]), _t(case["im_id"]))
    np.testing.assert_array_equal(out["tar_img"].cpu().numpy().view(np.uint32), ref_img.view(np.uint32))
    np.testing.assert_array_equal(out["tar_mask"].cpu().numpy(), ref_mask)
    # the reference's own composition on the GPU: (rgb/255 * mask, mask) -> CropResizePad -> normalise
    # (host float32 arithmetic: torch's GPU division is not correctly rounded on ROCm builds)
    rgb = _t(case["rgb"].astype(np.float32) / np.float32(255.0))
    m = _t(case["masks"])
    rgba = torch.cat([rgb[_t(case["im_id"]).long()] * m[:, None], m[:, None]], dim=1)
    two = CropResizePad(target_size=112)(_t(case["boxes"]), rgba)
    mean = np.asarray(CLIP_MEAN, np.float32).reshape(3, 1, 1)
    std = np.asarray(CLIP_STD, np.float32).reshape(3, 1, 1)
    np.testing.assert_array_equal((two["images"][:, :3].cpu().numpy() - mean) / std, out["tar_img"].cpu().numpy())
    assert torch.equal(two["images"][:, 3], out["tar_mask"])


def test_empty_batch_and_bad_box():
    from gigapose_amd.crop import CropResizePad

    out = CropResizePad()(torch.zeros(0, 4, dtype=torch.int64, device=DEV), torch.zeros(0, 3, 32, 32, device=DEV))
    assert out["images"].shape == (0, 3, 224, 224)
    with pytest.raises(ValueError):
        CropResizePad()(torch.tensor([[4, 4, 20, 20], [9, 9, 9, 12]], device=DEV), torch.zeros(2, 3, 32, 32, device=DEV))
