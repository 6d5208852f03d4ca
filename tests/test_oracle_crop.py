"""CPU: the numpy restatement of the detection pre-processing stage (oracle/crop_numpy.py) against the golden
written by the unmodified reference (CropResizePad + process_real arithmetic; oracle/make_goldens.py gen_crop)."""
import os

import numpy as np
import pytest

from gigapose_testing import synthetic as syn
from oracle import crop_numpy


def test_crop_restatement_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "crop.npz"))
    case = syn.detection_case(seed=int(g["seed"]))
    chk = case["rgb"].astype(np.float64).sum() + case["masks"].sum() + case["boxes"].sum()
    assert chk == float(g["input_checksum"]), "synthetic inputs drifted from the ones the golden was made with"
    img, mask, M = crop_numpy.preprocess_detections(case["rgb"], case["masks"], case["boxes"], case["im_id"])
    np.testing.assert_array_equal(mask, g["tar_mask"])
    np.testing.assert_array_equal(img.view(np.uint32), g["tar_img"].view(np.uint32))   # bit-exact pixels
    np.testing.assert_allclose(M, g["M"], rtol=2e-7, atol=0)                            # <= 1 ulp (matmul order)


def test_nearest_index_shortcuts_and_bounds():
    assert crop_numpy.nearest_index(224, 224).tolist() == list(range(224))
    assert crop_numpy.nearest_index(8, 4).tolist() == [0, 0, 1, 1, 2, 2, 3, 3]
    idx = crop_numpy.nearest_index(224, 223)
    assert idx[0] == 0 and idx[-1] == 222 and (np.diff(idx) >= 0).all()
    idx = crop_numpy.nearest_index(int(np.floor(300 * (224 / 300))), 300, float(np.float32(224) / np.float32(300)))
    assert idx.max() <= 299 and (np.diff(idx) >= 0).all()


def test_bad_boxes_raise():
    for box in [(5, 5, 5, 9), (-3, 0, 10, 10), (700, 10, 720, 30)]:
        with pytest.raises(ValueError):
            crop_numpy.crop_geometry(box, 480, 640)
