"""numpy restatement of the detection pre-processing stage -- TEST INFRASTRUCTURE ONLY
(allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

Follows the reference line by line:
  * src/utils/crop.py:11-61     CropResizePad.__call__ (crop, nearest resize by scale_factor, pad, M, final resize)
  * src/dataloader/train.py:80-123  process_real (rgb / 255, * mask, RGBA stack, crop)
  * src/dataloader/test.py:295-315  collate_fn (normalize(real_data.rgb))
  * configs/data/transform.yaml:1-12  CLIP mean / std, target_size 224
Un-vendored third parties whose arithmetic is restated from their published behaviour (pinned by the golden
tests/golden/crop.npz, written by oracle/make_goldens.py from the reference itself):
  * torch.nn.functional.interpolate(mode="nearest") (torch 2.10, ATen UpSampleNearest: output size
    floor(in * scale_factor) in double; source index min(floorf(dst * scale), in - 1) with float
    scale = 1/scale_factor when a scale_factor is given and in/out otherwise; identity when in == out and
    dst >> 1 when out == 2 * in)
  * torchvision.transforms.Normalize: (x - mean) / std in float32.
Parity pinned: tests/test_oracle_crop.py (bit-exact images / masks, M to 1 ulp).
"""
import math

import numpy as np

CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], np.float32)
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], np.float32)


def nearest_index(out_size, in_size, scale_factor=None):
    """ATen nearest_idx for every output index (UpSample.h: nearest_idx / nearest_neighbor_compute_source_index)."""
    dst = np.arange(out_size, dtype=np.int64)
    if out_size == in_size:
        return dst
    if out_size == 2 * in_size:
        return dst >> 1
    if scale_factor is not None and scale_factor > 0:
        scale = np.float32(1.0 / scale_factor)           # static_cast<float>(1.0 / scale) with scale a double
    else:
        scale = np.float32(in_size) / np.float32(out_size)
    src = np.floor(dst.astype(np.float32) * scale).astype(np.int64)
    return np.minimum(src, in_size - 1)


def crop_geometry(box, H, W, target=224):
    """Everything CropResizePad derives from one xyxy box (crop.py:16-49).  Returns a dict with the crop window
    after Python slicing clamps it, the float32 scale, the sizes after the first resize, the pads and M."""
    x0, y0, x1, y1 = (int(v) for v in box)
    if not (0 <= x0 < x1 and 0 <= y0 < y1 and x0 < W and y0 < H):
        raise ValueError(f"box {box} is empty or starts outside the {W}x{H} frame")
    bw, bh = x1 - x0, y1 - y0                              # BoundingBox.get_box_size (bbox.py:52-70), int32
    scale32 = np.float32(target) / np.float32(max(bw, bh))  # crop.py:20, float32 tensor division
    scale = float(scale32)                                 # .item()
    cw, ch = min(x1, W) - x0, min(y1, H) - y0              # slicing clamps at the border (crop.py:31)
    h1, w1 = int(math.floor(float(ch) * scale)), int(math.floor(float(cw) * scale))
    if h1 <= 0 or w1 <= 0:
        raise ValueError(f"box {box}: scaled crop is empty ({w1}x{h1})")
    pad_l = pad_t = 0
    hp, wp = h1, w1
    if w1 / h1 != 1:                                       # crop.py:37-47
        pad_t = (target - h1) // 2
        pad_b = max(target - h1 - pad_t, 0)
        pad_l = max((target - w1) // 2, 0)
        pad_r = target - w1 - pad_l
        hp, wp = h1 + pad_t + pad_b, w1 + pad_l + pad_r
    M_crop = np.eye(3, dtype=np.float32)
    M_crop[:2, 2] = (-x0, -y0)
    M_rp = np.eye(3, dtype=np.float32)
    M_rp[:2, :2] *= scale32
    if w1 / h1 != 1:
        M_rp[:2, 2] = (pad_l, pad_t)
    return dict(x0=x0, y0=y0, cw=cw, ch=ch, scale=scale, h1=h1, w1=w1, pad_t=pad_t, pad_l=pad_l, hp=hp, wp=wp,
                M=(M_rp @ M_crop).astype(np.float32))


def crop_resize_pad(images, boxes, target=224):
    """CropResizePad.__call__(xyxy_boxes, images): images (D,C,H,W) f32, boxes (D,4) int -> (images (D,C,T,T), M (D,3,3))."""
    D, C, H, W = images.shape
    out = np.zeros((D, C, target, target), np.float32)
    Ms = np.zeros((D, 3, 3), np.float32)
    for d in range(D):
        g = crop_geometry(boxes[d], H, W, target)
        crop = images[d][:, g["y0"]:g["y0"] + g["ch"], g["x0"]:g["x0"] + g["cw"]]
        iy = nearest_index(g["h1"], g["ch"], g["scale"])
        ix = nearest_index(g["w1"], g["cw"], g["scale"])
        img = crop[:, iy][:, :, ix]
        padded = np.zeros((C, g["hp"], g["wp"]), np.float32)
        padded[:, g["pad_t"]:g["pad_t"] + g["h1"], g["pad_l"]:g["pad_l"] + g["w1"]] = img
        jy = nearest_index(target, g["hp"])
        jx = nearest_index(target, g["wp"])
        out[d] = padded[:, jy][:, :, jx]
        Ms[d] = g["M"]
    return out, Ms


def preprocess_detections(rgb_u8, masks, boxes, im_id, target=224, mean=CLIP_MEAN, std=CLIP_STD):
    """process_real + normalize: rgb_u8 (n_img,3,H,W) u8, masks (D,H,W) f32, boxes (D,4), im_id (D) ->
    tar_img (D,3,T,T) CLIP-normalised masked crops, tar_mask (D,T,T), M (D,3,3)."""
    rgb = rgb_u8.astype(np.float32) / np.float32(255.0)
    m_rgb = rgb[im_id] * masks[:, None]
    rgba, M = crop_resize_pad(np.concatenate([m_rgb, masks[:, None]], axis=1), boxes, target)
    img = (rgba[:, :3] - mean.reshape(3, 1, 1)) / std.reshape(3, 1, 1)
    return img.astype(np.float32), rgba[:, 3], M
