"""TEST INFRASTRUCTURE ONLY -- numpy f32 restatement of the DINOv2 ViT forward that the reference
obtains from torch.hub (un-vendored; call site reference ae_net.py:44-47, :64-69).

Third-party dependency restated: facebookresearch/dinov2 `dinov2_vit{s,b,l}14` (no pinned commit,
reference configs/model/ae_net/dinov2_l.yaml:4-7).  Restated from the published architecture as
implemented in transformers 5.15.0 `models/dinov2/modeling_dinov2.py` (:97-112 embeddings,
:199-229 attention, :272-299 LayerScale/MLP, :342-380 block).  Parity for this stage is
"unpinned" by the reference itself (it has no tests); we pin the restatement against HF
Dinov2Model outputs (tests/test_oracle_vit.py).  Takes the hub-style state dict of
gigapose_amd.vit.Dinov2ViT as numpy arrays.
"""
import math

import numpy as np

try:
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf, otypes=[np.float64])


def _ln(x, g, b, eps=1e-6):
    x64 = x.astype(np.float64)
    mu = x64.mean(-1, keepdims=True)
    var = ((x64 - mu) ** 2).mean(-1, keepdims=True)
    return (((x64 - mu) / np.sqrt(var + eps)) * g + b).astype(np.float32)


def _gelu(x):
    x64 = x.astype(np.float64)
    return (0.5 * x64 * (1.0 + _erf(x64 / math.sqrt(2.0)))).astype(np.float32)


def forward_x_prenorm(sd, images, depth, heads, stop_after_layers=None):
    """images (B,3,224,224) -> x_prenorm (B,257,C) f32 (hidden state before the final LayerNorm)."""
    B = images.shape[0]
    W = sd["patch_embed.proj.weight"]
    C = W.shape[0]
    patches = images.reshape(B, 3, 16, 14, 16, 14).transpose(0, 2, 4, 1, 3, 5).reshape(B, 256, 588)
    x = patches @ W.reshape(C, 588).T + sd["patch_embed.proj.bias"]
    x = np.concatenate([np.broadcast_to(sd["cls_token"], (B, 1, C)), x], axis=1) + sd["pos_embed"]
    x = x.astype(np.float32)
    L = depth if stop_after_layers is None else stop_after_layers
    for i in range(L):
        p = f"blocks.{i}."
        h = _ln(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        qkv = h @ sd[p + "attn.qkv.weight"].T + sd[p + "attn.qkv.bias"]
        q, k, v = [t.reshape(B, 257, heads, 64).transpose(0, 2, 1, 3) for t in np.split(qkv, 3, axis=-1)]
        s = (q @ k.transpose(0, 1, 3, 2)) * np.float32(0.125)
        s = s - s.max(-1, keepdims=True)
        e = np.exp(s)
        a = (e / e.sum(-1, keepdims=True)).astype(np.float32)
        o = (a @ v).transpose(0, 2, 1, 3).reshape(B, 257, C)
        o = o @ sd[p + "attn.proj.weight"].T + sd[p + "attn.proj.bias"]
        x = x + sd[p + "ls1.gamma"] * o
        h = _ln(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
        m = _gelu(h @ sd[p + "mlp.fc1.weight"].T + sd[p + "mlp.fc1.bias"])
        m = m @ sd[p + "mlp.fc2.weight"].T + sd[p + "mlp.fc2.bias"]
        x = (x + sd[p + "ls2.gamma"] * m).astype(np.float32)
    return x


def patch_features(sd, images, depth, heads, normalize=True, stop_after_layers=None):
    """AENet output (reference ae_net.py:55-69): (B, C, 16, 16), L2-normalised over C."""
    x = forward_x_prenorm(sd, images, depth, heads, stop_after_layers)[:, 1:, :]
    B, _, C = x.shape
    f = np.ascontiguousarray(x.transpose(0, 2, 1)).reshape(B, C, 256)
    if normalize:
        from . import cpu
        f = cpu.l2norm_cp(f)
    return f.reshape(B, C, 16, 16)


# ---- position-table resampling: facebookresearch/dinov2 vision_transformer.py `interpolate_pos_encoding` ------------
# (un-vendored; the reference reaches it through forward_features, ae_net.py:44-47).  Restated from the published
# code: F.interpolate(grid (1,C,M,M), mode="bicubic", antialias=<interpolate_antialias>, scale_factor=(16+offset)/M)
# for interpolate_offset != 0 (released models: 0.1), or size=(16,16) for offset 0.  ATen semantics restated:
#   * output size = floor(M * scale_factor); the coordinate scale is 1/scale_factor when a scale factor is given
#     (recompute_scale_factor unset), M/out otherwise; align_corners=False: src = scale * (dst + 0.5) - 0.5;
#   * bicubic (UpSampleBicubic2d): Keys kernel A = -0.75 on taps floor(src)-1..+2, indices clamped to the border;
#   * antialias (UpSampleKernel.cpp, _compute_indices_min_size_weights_aa): Keys kernel a = -0.5 stretched by the scale
#     over support 2*scale around center = scale * (dst + 0.5), weights normalised to sum 1.
def _cubic(x, a):
    x = abs(x)
    if x <= 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return 0.0


def _resample_matrix(n_in, n_out, scale, antialias):
    """(n_out, n_in) float64 matrix of one separable pass."""
    W = np.zeros((n_out, n_in))
    for i in range(n_out):
        if not antialias:
            # ATen evaluates the source coordinate in f32 (area_pixel_compute_source_index<float>, scale =
            # float(1 / scale_factor)): with 37 / 16.1 that rounding moves values by ~1e-5, so it is restated too
            src = np.float32(np.float32(scale) * np.float32(i + 0.5) - np.float32(0.5))
            i0 = math.floor(src)
            t = float(np.float32(src - np.float32(i0)))
            for k in range(-1, 3):
                W[i, min(max(i0 + k, 0), n_in - 1)] += _cubic(k - t, -0.75)
        else:
            support = 2.0 * scale if scale >= 1.0 else 2.0
            inv = 1.0 / scale if scale >= 1.0 else 1.0
            center = scale * (i + 0.5)
            lo = max(int(center - support + 0.5), 0)
            hi = min(int(center + support + 0.5), n_in)
            w = np.array([_cubic((j - center + 0.5) * inv, -0.5) for j in range(lo, hi)])
            W[i, lo:hi] = w / w.sum()
    return W


def interpolate_pos_encoding(pos_embed, interpolate_offset=0.1, antialias=False, n_out=16):
    """pos_embed (1, 1 + M*M, C) -> (1, 1 + n_out*n_out, C), float64 arithmetic, returned as f32."""
    pe = np.asarray(pos_embed, np.float64)
    m = int(round(math.sqrt(pe.shape[1] - 1)))
    grid = pe[0, 1:].reshape(m, m, -1)
    if interpolate_offset:
        sf = float(n_out + interpolate_offset) / m
        assert math.floor(m * sf) == n_out
        scale = 1.0 / sf
    else:
        scale = m / n_out
    W = _resample_matrix(m, n_out, scale, antialias)
    out = np.einsum("ia,jb,abc->ijc", W, W, grid)
    return np.concatenate([pe[:, :1], out.reshape(1, n_out * n_out, -1)], axis=1).astype(np.float32)
