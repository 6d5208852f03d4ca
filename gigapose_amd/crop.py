"""Detection pre-processing behind the reference interface.

`CropResizePad` mirrors src/utils/crop.py:11-61 (Hydra target configs/data/transform.yaml:9-11): same ctor
kwargs, `__call__(xyxy_boxes, images) -> {"M", "images"}`; on MI355X the per-detection Python loop (slice, two
nearest F.interpolate, F.pad) is ONE gather kernel for the whole batch (gp_crop_resize_pad).
`DetectionPreprocessor` fuses what GigaPoseTestSet.process_real + collate_fn do around it
(src/dataloader/train.py:80-123, src/dataloader/test.py:295-315): rgb/255 * mask, crop, CLIP normalisation ->
`tar_img`, `tar_mask`, `tar_M` straight from the uint8 frames, with no (D,4,H,W) float RGBA stack in between.
Results are bit-identical to the reference (tests/test_gpu_crop.py against tests/golden/crop.npz).
There is no CPU fallback: tensors must live on the GPU.
"""
import ctypes

import torch

from . import _lib

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # configs/data/transform.yaml:6-7
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _check(err, what):
    bad = int(err.item())  # one host sync per batch: the reference raises at the same place (empty slice)
    if bad:
        raise ValueError(f"{what}: detection {bad - 1} has an empty / out-of-frame box")


class CropResizePad:
    def __init__(self, target_size=224, patch_size=14):
        self.target_size = target_size
        self.patch_size = patch_size

    @torch.no_grad()
    def __call__(self, xyxy_boxes, images):
        """xyxy_boxes (D,4) integer tensor, images (D,C,H,W) float tensor -> dict(M (D,3,3), images (D,C,T,T))."""
        dev = images.device
        images = images.contiguous().float()
        boxes = xyxy_boxes.to(device=dev, dtype=torch.int64).contiguous()
        D, C, H, W = images.shape
        T = self.target_size
        out = torch.empty(D, C, T, T, device=dev)
        M = torch.empty(D, 3, 3, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.call("gp_crop_resize_pad", _lib.ptr(images), _lib.ptr(boxes), _lib.i(D), _lib.i(C), _lib.i(H), _lib.i(W),
                  _lib.i(T), _lib.ptr(out), _lib.ptr(M), _lib.ptr(err), _lib.stream_ptr())
        _check(err, "CropResizePad")
        return {"M": M, "images": out}

    def forward_image_wrap(self, images, M):
        raise NotImplementedError("cv2.warpAffine path (src/utils/crop.py:63-75): unused by the inference hot path")


class DetectionPreprocessor:
    """frames uint8 (n_img,3,H,W) + per-detection masks / boxes / frame ids -> what eval_retrieval consumes."""

    def __init__(self, target_size=224, mean=CLIP_MEAN, std=CLIP_STD):
        self.target_size = target_size
        self._mean = (ctypes.c_float * 3)(*mean)
        self._std = (ctypes.c_float * 3)(*std)

    @torch.no_grad()
    def __call__(self, rgb_u8, masks, xyxy_boxes, batch_im_id):
        dev = rgb_u8.device
        assert rgb_u8.dtype == torch.uint8 and rgb_u8.dim() == 4 and rgb_u8.shape[1] == 3
        rgb_u8 = rgb_u8.contiguous()
        masks = masks.to(device=dev, dtype=torch.float32).contiguous()
        boxes = xyxy_boxes.to(device=dev, dtype=torch.int64).contiguous()
        im_id = batch_im_id.to(device=dev, dtype=torch.int32).contiguous()
        n_img, _, H, W = rgb_u8.shape
        D, T = masks.shape[0], self.target_size
        assert masks.shape[1:] == (H, W) and boxes.shape == (D, 4) and im_id.shape == (D,)
        tar_img = torch.empty(D, 3, T, T, device=dev)
        tar_mask = torch.empty(D, T, T, device=dev)
        M = torch.empty(D, 3, 3, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.call("gp_preprocess_detections", _lib.ptr(rgb_u8), _lib.ptr(masks), _lib.ptr(boxes), _lib.ptr(im_id),
                  _lib.i(n_img), _lib.i(D), _lib.i(H), _lib.i(W), _lib.i(T), self._mean, self._std, _lib.ptr(tar_img),
                  _lib.ptr(tar_mask), _lib.ptr(M), _lib.ptr(err), _lib.stream_ptr())
        _check(err, "DetectionPreprocessor")
        return {"tar_img": tar_img, "tar_mask": tar_mask, "tar_M": M}
