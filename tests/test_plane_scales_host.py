"""CPU: the host side of the per-tensor plane scales (gigapose_amd/vit.py, round 5) -- how a scale is picked from a calibrated max |x|,
what resets a calibration, and what the report lists.  The device side is tests/test_gpu_plane_scales.py."""
import numpy as np
import torch

from gigapose_amd.vit import Dinov2ViT


def test_scale_is_the_largest_power_of_two_with_headroom():
    f = Dinov2ViT.scale_for
    assert f(0.0, 4) == 8.0 and f(float("nan"), 4) == 8.0            # a tensor the calibration never saw on the plane path keeps x 8
    assert f(5.0, 4) == 8.0 and f(2047.0, 4) == 8.0                    # 2047 x 8 x 4 = 65504: the last value x 8 still covers with 4 x headroom
    assert f(2048.0, 4) == 4.0 and f(1.2e4, 4) == 1.0 and f(9.0e3, 4) == 1.0 and f(1.05e4, 4) == 1.0
    assert f(1.0e6, 4) == 2.0 ** -6 and f(1.0e12, 4) == 2.0 ** -10     # floor: the entry point accepts [2^-10, 64]
    for a in (3.0, 77.0, 4097.0, 3.3e4, 2.0e5):
        s = f(a, 4)
        assert a * s * 4 <= 65504.0 and (s == 8.0 or a * (2 * s) * 4 > 65504.0) and np.log2(s) == int(np.log2(s))
    assert f(1.2e4, 1) == 4.0                                          # plane_headroom = 1: no headroom


def test_calibration_state_and_report():
    vit = Dinov2ViT(384, 2, 6)
    assert vit.plane_scales is None and vit.plane_amax is None and vit.plane_scale_report() == {}
    # nothing to calibrate outside the split plane path: no launch is attempted (these would raise on a CPU tensor otherwise)
    x = torch.zeros(2, 3, 224, 224)
    assert vit.set_numerics("chain").calibrate_plane_scales(x) is False
    assert vit.set_numerics("split").set_split_gemm("128").calibrate_plane_scales(x) is False
    assert vit.set_split_gemm("256").calibrate_plane_scales(x[:0]) is False
    vit.plane_amax = np.array([[1.0, 2.0, 3.0, 1.2e4], [5.0, 9.0e3, 1.05e4, 0.5]])
    vit.plane_scales = [Dinov2ViT.scale_for(float(a), 4.0) for a in vit.plane_amax.reshape(-1)]
    assert vit.plane_scale_report() == {"L0.gelu": (1.2e4, 1.0), "L1.qkv": (9.0e3, 1.0), "L1.ln2": (1.05e4, 1.0)}
    vit.invalidate()                                                   # weights edited in place: the old calibration says nothing
    assert vit.plane_scales is None and vit.plane_amax is None
    vit.plane_scales = [8.0] * 8
    vit.load_state_dict(vit.state_dict())                              # so do newly loaded weights
    assert vit.plane_scales is None


def test_a_calibration_can_be_adopted_from_a_bank_file():
    vit = Dinov2ViT(384, 2, 6)
    assert vit.adopt_plane_amax(np.array([[1.0, 2.0, 3.0, 4.0], [5.0, 6.0, 7.0, 8.0]])) is False and vit.plane_scales is None
    assert vit.plane_amax is not None                                  # "calibrated": set_template_data will not calibrate again
    assert vit.adopt_plane_amax(np.array([[0.0, 0.0, 0.0, 1.2e4], [0.0, 0.0, 0.0, 0.0]])) is True
    assert vit.plane_scale_report() == {"L0.gelu": (1.2e4, 1.0)} and vit.plane_amax[1, 3] == 8.0     # running maximum
    import pytest

    with pytest.raises(ValueError):
        vit.adopt_plane_amax(np.full((2, 4), np.nan))
