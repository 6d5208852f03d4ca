"""Deterministic synthetic inputs for the GigaPose hot path.

numpy RandomState streams only (bit-stable across machines), so the golden fixtures in
tests/golden/ store OUTPUTS plus an input checksum instead of megabytes of inputs.
Used by tests/, bench.py, __graft_entry__.smoke() and oracle/make_goldens.py.
"""
import hashlib

import numpy as np

P = 256
G = 16
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # reference configs/data/transform.yaml:6
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)    # reference configs/data/transform.yaml:7
TEMPLATE_K = np.array([[572.4114, 0.0, 320.0], [0.0, 573.57043, 240.0], [0.0, 0.0, 1.0]],
                      dtype=np.float32)             # reference template_dataset.py:194-196


def checksum(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()[:16]


def _unit(x, axis):
    n = np.sqrt((x.astype(np.float64) ** 2).sum(axis=axis, keepdims=True))
    return (x / np.maximum(n, 1e-12)).astype(np.float32)


def _smooth_field(rs, C, coarse=4, fine=0.6):
    """(C,16,16) field: bilinear-upsampled coarse noise + per-patch noise (neighbouring patches
    are correlated, so several template patches compete above the 0.5 threshold)."""
    c = rs.standard_normal((C, coarse, coarse)).astype(np.float32)
    xs = np.linspace(0, coarse - 1, G)
    i0 = np.floor(xs).astype(int).clip(0, coarse - 2)
    w = (xs - i0).astype(np.float32)
    rows = c[:, i0, :] * (1 - w)[None, :, None] + c[:, i0 + 1, :] * w[None, :, None]
    up = rows[:, :, i0] * (1 - w)[None, None, :] + rows[:, :, i0 + 1] * w[None, None, :]
    return up + fine * rs.standard_normal((C, G, G)).astype(np.float32)


def disc_mask(rs, size=224, include_origin=None):
    """Random disc / rectangle {0,1} mask with partially covered patches.  `include_origin`
    forces pixel (0,0) (= patch 0's sample point, matching.py nearest rule) in or out."""
    yy, xx = np.mgrid[0:size, 0:size]
    if rs.rand() < 0.5:
        cx, cy = rs.uniform(70, 154, 2)
        r = rs.uniform(60, 105)
        m = ((xx - cx) ** 2 + (yy - cy) ** 2) <= r * r
    else:
        x0, y0 = rs.randint(0, 60, 2)
        x1, y1 = rs.randint(150, 224, 2)
        m = (xx >= x0) & (xx < x1) & (yy >= y0) & (yy < y1)
    m = m.astype(np.float32)
    if include_origin is True:
        m[:20, :20] = 1.0
    elif include_origin is False:
        m[:14, :14] = 0.0
    return m


def matcher_case(seed, B, O, N, C, noise=0.35, shift=True, full_masks=False):
    """Feature-level inputs for LocalSimilarity.test with planted, graded matches.

    Template n of object o = unit(w_n * field_o rolled by (dy_n,dx_n) + (1-w_n) * noise_n) so
    templates score differently (no top-k ties); query b = unit(field_{o_b} + noise).
    Returns dict(src_feats (O,N,C,16,16), tar_feat (B,C,16,16), src_masks (O,N,224,224),
    tar_mask (B,224,224), labels (B,) 0-based int32).
    """
    rs = np.random.RandomState(seed)
    fields = np.stack([_smooth_field(rs, C) for _ in range(O)])
    src = np.empty((O, N, C, G, G), np.float32)
    for o in range(O):
        ws = np.linspace(1.0, 0.55, N)
        rs.shuffle(ws)
        for n in range(N):
            f = fields[o]
            if shift:
                dy, dx = rs.randint(-2, 3, 2)
                f = np.roll(f, (int(dy), int(dx)), axis=(1, 2))
            t = ws[n] * _unit(f, 0) + (1 - ws[n]) * 1.4 * _unit(rs.standard_normal((C, G, G)), 0)
            src[o, n] = _unit(t, 0)
    labels = rs.randint(0, O, B).astype(np.int32)
    tar = np.empty((B, C, G, G), np.float32)
    for b in range(B):
        t = _unit(fields[labels[b]], 0) + noise * _unit(rs.standard_normal((C, G, G)), 0)
        tar[b] = _unit(t, 0)
    if full_masks:
        src_masks = np.ones((O, N, 224, 224), np.float32)
        tar_mask = np.ones((B, 224, 224), np.float32)
    else:
        src_masks = np.stack([np.stack([disc_mask(rs, include_origin=[None, True, False][(o + n) % 3])
                                        for n in range(N)]) for o in range(O)])
        tar_mask = np.stack([disc_mask(rs, include_origin=[True, False, None][b % 3]) for b in range(B)])
    return dict(src_feats=src, tar_feat=tar, src_masks=src_masks, tar_mask=tar_mask, labels=labels)


def random_features(seed, B, O, N, C):
    """Unstructured bench-scale features (values only matter for timing, not for matching)."""
    rs = np.random.RandomState(seed)
    bank = _unit(rs.standard_normal((O, N, C, P)).astype(np.float32), 2)
    q = _unit(rs.standard_normal((B, C, P)).astype(np.float32), 1)
    return bank, q


def template_images(seed, N, size=224):
    """Templates: low-frequency noise (16x16 N(0,1) bilinear-upsampled) x disc mask radius 100."""
    rs = np.random.RandomState(seed)
    coarse = rs.standard_normal((N, 3, G, G)).astype(np.float32)
    xs = np.linspace(0, G - 1, size)
    i0 = np.floor(xs).astype(int).clip(0, G - 2)
    w = (xs - i0).astype(np.float32)
    rows = coarse[:, :, i0, :] * (1 - w)[None, None, :, None] + coarse[:, :, i0 + 1, :] * w[None, None, :, None]
    up = rows[:, :, :, i0] * (1 - w) + rows[:, :, :, i0 + 1] * w
    yy, xx = np.mgrid[0:size, 0:size]
    mask = (((xx - size / 2) ** 2 + (yy - size / 2) ** 2) <= 100.0 ** 2).astype(np.float32)
    masks = np.broadcast_to(mask, (N, size, size)).copy()
    return (up * masks[:, None]).astype(np.float32), masks


def query_crops(seed, templates, masks, choose, noise=0.1):
    """Crops = chosen template + noise, times its mask."""
    rs = np.random.RandomState(seed)
    imgs = templates[choose] + noise * rs.standard_normal(templates[choose].shape).astype(np.float32)
    return (imgs * masks[choose][:, None]).astype(np.float32), masks[choose].copy()


def crop_geometry(seed, B):
    """tar_K (f in [500,1200]) and tar_M = isotropic scale in [0.3,3] + translation
    (satisfies the asserts of reference lib3d/torch.py:54-55)."""
    rs = np.random.RandomState(seed)
    K = np.zeros((B, 3, 3), np.float32)
    f = rs.uniform(500, 1200, B)
    K[:, 0, 0] = f
    K[:, 1, 1] = f * rs.uniform(0.98, 1.02, B)
    K[:, 0, 2] = rs.uniform(280, 360, B)
    K[:, 1, 2] = rs.uniform(200, 280, B)
    K[:, 2, 2] = 1
    M = np.zeros((B, 3, 3), np.float32)
    s = rs.uniform(0.3, 3.0, B)
    M[:, 0, 0] = s
    M[:, 1, 1] = s
    M[:, 0, 2] = rs.uniform(-300, 50, B)
    M[:, 1, 2] = rs.uniform(-300, 50, B)
    M[:, 2, 2] = 1
    return K, M


def template_geometry(seed, O, N):
    """Template crop transforms M (O,N,3,3), intrinsics K (O,3,3) and poses (O,N,4,4):
    orthonormal R looking at the origin from an icosphere-like direction, |t| = 400
    (reference uses obj_poses_level1.npy * 0.4, render_bop_templates.py:69-70)."""
    rs = np.random.RandomState(seed)
    K = np.broadcast_to(TEMPLATE_K, (O, 3, 3)).copy()
    M = np.zeros((O, N, 3, 3), np.float32)
    s = rs.uniform(0.8, 2.5, (O, N))
    M[..., 0, 0] = s
    M[..., 1, 1] = s
    M[..., 0, 2] = rs.uniform(-400, -50, (O, N))
    M[..., 1, 2] = rs.uniform(-300, -20, (O, N))
    M[..., 2, 2] = 1
    poses = np.zeros((O, N, 4, 4), np.float32)
    for o in range(O):
        for n in range(N):
            q, _ = np.linalg.qr(rs.standard_normal((3, 3)))
            if np.linalg.det(q) < 0:
                q[:, 0] = -q[:, 0]
            poses[o, n, :3, :3] = q
            poses[o, n, :3, 3] = [rs.uniform(-20, 20), rs.uniform(-20, 20), rs.uniform(350, 450)]
            poses[o, n, 3, 3] = 1
    return K, M, poses


def fill_state_dict(module, seed):
    """Deterministic (numpy-stream) weights for any torch module, keyed by parameter NAME order, so
    the reference modules (golden generation) and gigapose_amd's mirrors (tests, bench) get identical
    weights wherever their state-dict names/shapes agree -- which also checks checkpoint-key parity.
    Scales: weights ~ N(0, 2/fan_in) (0.02 for tokens/pos-embed), norm scales / LayerScale in
    [0.8,1.2], running_var in [0.5,1.5], biases / running_mean ~ N(0, 0.05)."""
    import torch

    rs = np.random.RandomState(seed)
    sd = module.state_dict()
    new = {}
    for name in sorted(sd):
        t = sd[name]
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            new[name] = t.clone()
            continue
        if name.endswith("running_var"):
            v = rs.uniform(0.5, 1.5, shape)
        elif t.dim() >= 2 and not any(k in name for k in ("cls_token", "pos_embed", "position_embeddings", "mask_token")):
            fan_in = int(np.prod(shape[1:]))
            v = rs.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        elif t.dim() >= 2:
            v = rs.standard_normal(shape) * 0.02
        elif name.endswith("weight") or name.endswith("gamma") or name.endswith("lambda1"):
            v = rs.uniform(0.8, 1.2, shape)
        else:
            v = rs.standard_normal(shape) * 0.05
        new[name] = torch.from_numpy(np.asarray(v, dtype=np.float32)).reshape(shape)
    module.load_state_dict(new)
    return module


# DINOv2-like outliers for a random-init stand-in.  Trained ViTs without register tokens are known for a handful of
# "massive activations": a few residual channels two orders of magnitude above the rest, produced by one or two MLP layers whose
# hidden units fire at 1e3-1e4, and LayerNorm gains that blow single channels up.  The split numerics' f16 planes hold |s x| <= 65504
# (s = 8 by default): these are the tensors that need a smaller per-tensor scale.  Values chosen so that the default scale TRIPS the
# range guard in three different kinds of plane tensor: a LayerNorm output (norm2 gain on a residual outlier channel: ~1e4), q|k|v and
# the attention output (a value-projection bias of 9e3: the same for every token, so the softmax logits stay sane), and a GELU output
# (two hidden units at 1.2e4).  The outliers sit on the MLP / value side on purpose: a LayerNorm-1 outlier would put ~1e5 into the
# attention logits, where two correct f32 implementations already disagree.
OUTLIER_SPEC = dict(residual_channels=(5, 300, 911), residual_value=150.0,
                    ln2_layers=(4, 15), ln2_channels=(5, 700), ln2_gain=600.0,
                    v_layers=(9,), v_channels=(33, 520), v_bias=9.0e3,
                    fc1_layers=(6, 18), fc1_units=(7, 900), fc1_bias=1.2e4)


def plant_dinov2_outliers(model, spec=None):
    """Plant OUTLIER_SPEC into a DINOv2 ViT in place: a gigapose_amd.vit.Dinov2ViT (hub naming) or a transformers.Dinov2Model
    (HF naming: the float64 oracle of the tests).  Layer indices wrap around the model's depth.  Returns the model."""
    import torch

    sp = dict(OUTLIER_SPEC, **(spec or {}))
    hf = hasattr(model, "embeddings")
    with torch.no_grad():
        if hf:
            layers, pe_bias = model.encoder.layer, model.embeddings.patch_embeddings.projection.bias
        else:
            layers, pe_bias = model.blocks, model.patch_embed.proj.bias
        depth, dim = len(layers), pe_bias.shape[0]
        for c in sp["residual_channels"]:
            pe_bias[c % dim] += sp["residual_value"]
        for li in sp["ln2_layers"]:
            for c in sp["ln2_channels"]:
                layers[li % depth].norm2.weight[c % dim] *= sp["ln2_gain"]
        for li in sp["v_layers"]:
            for c in sp["v_channels"]:
                if hf:
                    layers[li % depth].attention.attention.value.bias[c % dim] = sp["v_bias"]
                else:
                    layers[li % depth].attn.qkv.bias[2 * dim + c % dim] = sp["v_bias"]
        for li in sp["fc1_layers"]:
            fc1 = layers[li % depth].mlp.fc1
            for u in sp["fc1_units"]:
                fc1.bias[u % fc1.bias.shape[0]] = sp["fc1_bias"]
    if hasattr(model, "invalidate"):
        model.invalidate()
    return model


def condition_ist(module):
    """Rescale a random-init ISTNet (reference or mirror; keyed by state-dict names) into the regime a TRAINED one
    works in: the Kaiming-initialised net emits |features| ~ 23, scales of 5 +- 15 (negative ones included) and fully
    saturated tanh outputs, so two correct f32 evaluations of it differ by ~3e-4 relative and RANSAC's 14-px test is
    decided by rounding.  Exact power-of-two rescalings of three weight tensors + fixed output biases give features
    of O(1), relScale = 1.2 +- 0.02 and (cos, sin) ~ (cos 0.3, sin 0.3) +- 0.03: |1 - s e^{i theta}| ~ 0.38, so the
    re-projection error of a correspondence d patches away from the proposing one is ~ 5.4 d px -- the integer
    lattice distances next to the 14-px boundary are sqrt(5) (12.0 px, inlier) and sqrt(8) (15.2 px, outlier)."""
    import torch

    sd = module.state_dict()
    with torch.no_grad():
        sd["backbone.layer4_outconv.weight"].mul_(1.0 / 32.0)
        sd["regressor.scale_predictor.4.weight"].mul_(1.0 / 32.0)
        sd["regressor.scale_predictor.4.bias"].fill_(1.2)
        sd["regressor.inplane_predictor.4.weight"].mul_(1.0 / 32.0)
        sd["regressor.inplane_predictor.4.bias"].copy_(torch.tensor([1.886, 0.3046]))
    module.load_state_dict(sd)
    return module


def many_to_one_case(seed, R):
    """RANSAC stress case: neighbouring query patches matched to the SAME template patch, so many
    errors are exactly 14 px in exact arithmetic and the `<= 14` test is decided by rounding (see
    gp_pose.hip).  Problem r has n_r valid correspondences, n_r straddling torch's bmm switch at 46."""
    rs = np.random.RandomState(seed)
    sizes = [2, 9, 44, 45, 46, 47, 90, 200, 256][:R] + [int(rs.randint(2, 257)) for _ in range(max(0, R - 9))]
    src_pts = -np.ones((R, P, 2), np.int64)
    tar_pts = -np.ones((R, P, 2), np.int64)
    rel_scale = np.full((R, P), -1000, np.float32)
    rel_inplane = np.full((R, P, 2), -1000, np.float32)
    for r, n in enumerate(sizes):
        pos = np.sort(rs.choice(P, n, replace=False))
        prev = None
        sat = r % 2 == 0  # saturated tanh outputs (+-1), as random-init nets produce
        for i, p in enumerate(pos):
            tar_pts[r, p] = (p % G, p // G)
            if prev is not None and i % 2 == 1:
                src_pts[r, p] = prev
            else:
                src_pts[r, p] = rs.randint(0, G, 2)
            prev = src_pts[r, p].copy()
            rel_scale[r, p] = rs.uniform(0.3, 2.0)
            ang = rs.uniform(-3, 3)
            rel_inplane[r, p] = (1.0, -1.0) if sat else (np.cos(ang), np.sin(ang))
    return dict(src_pts=src_pts, tar_pts=tar_pts, rel_scale=rel_scale, rel_inplane=rel_inplane)


def correspondences_case(seed, B, k):
    """Synthetic post-matcher state for the IST / RANSAC / recovery stages: (x,y) correspondences
    with -1 padding that mostly follow one similarity per (b,k) plus outliers; IST features random.
    Includes the edge cases N=0 (b=0,k=0) and N=1 (b=0,k=1)."""
    rs = np.random.RandomState(seed)
    src_pts = -np.ones((B, k, P, 2), np.int64)
    tar_pts = -np.ones((B, k, P, 2), np.int64)
    rel_scale = np.full((B, k, P), -1000, np.float32)
    rel_inplane = np.full((B, k, P, 2), -1000, np.float32)
    for b in range(B):
        for j in range(k):
            n = 0 if (b, j) == (0, 0) else 1 if (b, j) == (0, 1) else int(rs.randint(2, 120))
            pos = np.sort(rs.choice(P, n, replace=False))
            s = rs.uniform(0.6, 1.6)
            a = rs.uniform(-0.6, 0.6)
            for p in pos:
                tx, ty = p % G, p // G
                # template location = inverse similarity of the query location about the centre
                cx, cy = tx - 7.5, ty - 7.5
                sxf = (np.cos(a) * cx + np.sin(a) * cy) / s + 7.5
                syf = (-np.sin(a) * cx + np.cos(a) * cy) / s + 7.5
                if rs.rand() < 0.25:
                    sxf, syf = rs.uniform(0, 15, 2)
                tar_pts[b, j, p] = (tx, ty)
                src_pts[b, j, p] = (int(np.clip(round(sxf), 0, 15)), int(np.clip(round(syf), 0, 15)))
                rel_scale[b, j, p] = s * rs.uniform(0.9, 1.1)
                ang = a + rs.normal(0, 0.08)
                rel_inplane[b, j, p] = (np.cos(ang), np.sin(ang))
    return dict(src_pts=src_pts, tar_pts=tar_pts, rel_scale=rel_scale, rel_inplane=rel_inplane)


def detection_case(seed, n_img=2, D=10, H=480, W=640):
    """Full-frame uint8 images, per-detection binary masks and xyxy boxes for the crop pre-processing stage
    (reference src/dataloader/train.py:80-123 + src/utils/crop.py:11-61).  Boxes cover: wide, tall, square,
    exactly target-sized, tiny (up-scaling), odd sizes whose scaled side falls one pixel short of 224, a box
    that runs past the right/bottom border (the reference's slicing clamps it) and the full frame."""
    rs = np.random.RandomState(seed)
    rgb = rs.randint(0, 256, (n_img, 3, H, W)).astype(np.uint8)
    fixed = [(100, 50, 324, 274), (10, 20, 310, 140), (400, 100, 470, 420), (5, 7, 16, 14), (0, 0, W, H),
             (W - 120, H - 90, W + 30, H + 25), (33, 41, 33 + 223, 41 + 223), (200, 120, 200 + 225, 120 + 97)]
    if (H, W) != (480, 640):  # other frame sizes: keep the frame-relative cases only
        fixed = [(0, 0, W, H), (W - W // 5, H - H // 6, W + 30, H + 25), (W // 4, H // 4, W // 4 + 9, H // 4 + 6)]
    boxes = []
    for d in range(D):
        if d < len(fixed):
            boxes.append(fixed[d])
        else:
            w, h = rs.randint(8, W // 2), rs.randint(8, H // 2)
            x0, y0 = rs.randint(0, W - w), rs.randint(0, H - h)
            boxes.append((x0, y0, x0 + w, y0 + h))
    boxes = np.asarray(boxes, np.int64)
    masks = np.zeros((D, H, W), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for d, (x0, y0, x1, y1) in enumerate(boxes):
        cx, cy = 0.5 * (x0 + min(x1, W)), 0.5 * (y0 + min(y1, H))
        rx, ry = 0.55 * (min(x1, W) - x0), 0.45 * (min(y1, H) - y0)
        masks[d] = (((xx - cx) / max(rx, 1)) ** 2 + ((yy - cy) / max(ry, 1)) ** 2 <= 1.0).astype(np.float32)
    im_id = rs.randint(0, n_img, D).astype(np.int32)
    return dict(rgb=rgb, masks=masks, boxes=boxes, im_id=im_id)


def prediction_batches(seed, n_batches=3, k=4):
    """Per-batch npz payloads in the layout GigaPose.filter_and_save writes (reference gigaPose.py:436-447):
    several images, an image spread over two batches, repeated objects."""
    rs = np.random.RandomState(seed)
    out = []
    for b in range(n_batches):
        n = int(rs.randint(3, 7))
        im = np.sort(rs.randint(1 + b, 3 + b, n)).astype(np.int32)        # image ids overlap between batches
        poses = rs.standard_normal((n, k, 4, 4)).astype(np.float32)
        poses[..., 3, :] = (0, 0, 0, 1)
        det_t = {int(i): float(rs.uniform(0.05, 0.3)) for i in np.unique(im)}
        out.append(dict(scene_id=np.full(n, 2, np.int32), im_id=im, object_id=rs.randint(1, 9, n).astype(np.int32),
                        time=np.full(n, float(rs.uniform(0.02, 0.08))), detection_time=np.array([det_t[int(i)] for i in im]),
                        poses=poses, scores=np.sort(rs.rand(n, k).astype(np.float32), axis=1)[:, ::-1].copy()))
    return out
