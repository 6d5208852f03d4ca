"""DINOv2 ViT backbone on MI355X (libgigapose_hip.so: gp_vit_forward).

Stands where the reference puts `torch.hub.load("facebookresearch/dinov2", "dinov2_vitl14")`
(configs/model/ae_net/dinov2_l.yaml:4-7; called at ae_net.py:44-47 as
`dinov2_model.forward_features(x)["x_prenorm"]`).  Parameters use the hub model's names
(`patch_embed.proj`, `blocks.{i}.attn.qkv`, `ls1.gamma`, ...) so that the reference checkpoint's
`ae_net.dinov2_model.*` keys load unchanged; `from_hf()` converts the HF `Dinov2Model` stand-in
used as CPU oracle.  All arithmetic is in the HIP library; this class owns weights + workspace.
"""
import ctypes
import math
import os

import torch
from torch import nn

from . import _lib

VARIANTS = {  # name: (dim, depth, heads)
    "dinov2_vits14": (384, 12, 6),
    "dinov2_vitb14": (768, 12, 12),
    "dinov2_vitl14": (1024, 24, 16),
}
T = 257
# Numerics of the linear layers:
#   "split" (default): each f32 operand split into two f16 halves, 3 f16 MFMAs per k-block with f32 accumulation
#            (gp_split.hip / gp_split256.hip): f32-equivalent accuracy (measured error vs f64 below the chain's), ~3x faster
#   "chain": f32-input MFMA, every dot product is the sequential fmaf chain the CPU oracle restates (bit-exact parity;
#            the verification mode)
NUMERICS = ("chain", "split")


def split_planes(w):
    """[out][in] f32 weight -> (hi, lo) f16 planes: w ~= hi + lo * 2^-11 (same rounding as the device's split1)."""
    w = w.detach().float().contiguous()
    hi = w.half()
    lo = ((w - hi.float()) * 2048.0).half()
    return hi.contiguous(), lo.contiguous()


def split_planes_x64(w):
    """[out][in] f32 weight -> (hi, lo) f16 planes of 64 w with the low half NOT rescaled: 64 w ~= hi + lo
    (gp_split256.hip's single-accumulator convention; gp_split256_weights on the device)."""
    w = w.detach().float().contiguous()
    hi = torch.empty(w.shape, dtype=torch.float16, device=w.device)
    lo = torch.empty_like(hi)
    _lib.call("gp_split256_weights", _lib.ptr(w), ctypes.c_size_t(w.numel()), _lib.ptr(hi), _lib.ptr(lo), _lib.stream_ptr())
    return hi, lo


class _Block(nn.Module):
    def __init__(self, dim, mlp_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(dim, 3 * dim)
        self.attn.proj = nn.Linear(dim, dim)
        self.ls1 = nn.Module()
        self.ls1.gamma = nn.Parameter(torch.ones(dim))
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(dim, mlp_dim)
        self.mlp.fc2 = nn.Linear(mlp_dim, dim)
        self.ls2 = nn.Module()
        self.ls2.gamma = nn.Parameter(torch.ones(dim))


class Dinov2ViT(nn.Module):
    """patch 14, 224x224 input (16x16 patches + CLS), head dim 64, MLP ratio 4, GELU(erf),
    LayerScale, pre-norm; forward stops BEFORE the final LayerNorm (x_prenorm)."""

    def __init__(self, dim=1024, depth=24, heads=16, mlp_ratio=4, num_pos=T):
        super().__init__()
        assert dim == heads * 64, "kernels are specialised for head dim 64"
        self.dim, self.depth, self.heads, self.mlp_dim = dim, depth, heads, dim * mlp_ratio
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_pos, dim))
        self.mask_token = nn.Parameter(torch.zeros(1, dim))  # unused at inference; keeps ckpt keys
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv2d(3, dim, kernel_size=14, stride=14)
        self.blocks = nn.ModuleList([_Block(dim, self.mlp_dim) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)  # final norm: present in checkpoints, not applied
        self._packed = None
        self._ws = None
        self.numerics = _lib.default_numerics()   # "split" unless GIGAPOSE_NUMERICS=chain (_lib.default_numerics)
        # split numerics: "256" = single-accumulator plane kernels (activations x 8 in f16 planes: |x| < 8190, guarded);
        # "128" = every GEMM on the two-accumulator 128 x 128 kernel (range 65504, slower).  GigaPose switches to "128" by itself
        # when the range guard trips (gigaPose.py: _widen_split_range) -- DINOv2 checkpoints are known for a few massive activations
        self.split_gemm = "256"
        # Per-tensor plane scales of the split plane path (gp_vit_forward_split2; round 5).  Four activation tensors per layer travel as
        # f16 hi / lo planes of s x -- LayerNorm-1 out, q|k|v (+ attention out), LayerNorm-2 out, GELU out -- with s a power of two
        # <= 8 chosen per (layer, tensor) from a calibration pass over real inputs (GigaPose onboarding: the templates) so that
        # max |x| * s * plane_headroom <= 65504.  None = uncalibrated = 8 everywhere (the stand-in weights never need less).  A
        # checkpoint with DINOv2's massive activations lowers s for the few tensors that hold them; every GEMM stays on the
        # single-accumulator 256 x 256 kernels (the old remedy moved the WHOLE ViT to the 128 x 128 kernels: -43 %).
        self.plane_scales = None      # list of depth * 4 floats, or None
        self.plane_amax = None        # running max |x| per (layer, tensor) over every calibration pass: numpy (depth, 4)
        self.plane_headroom = 4.0

    def set_split_gemm(self, mode):
        if mode not in ("256", "128"):
            raise ValueError("split_gemm must be '256' or '128'")
        if mode != self.split_gemm:
            self.split_gemm = mode
            self._packed = None
        return self

    def set_numerics(self, mode):
        if mode not in NUMERICS:
            raise ValueError(f"numerics must be one of {NUMERICS}")
        if mode != self.numerics:
            self.numerics = mode
            self._packed = None
        return self

    # ---------------------------------------------------------------- construction helpers
    @classmethod
    def from_name(cls, name):
        dim, depth, heads = VARIANTS[name]
        return cls(dim, depth, heads)

    @classmethod
    def from_hf(cls, hf_model):
        """Convert a transformers.Dinov2Model (separate q/k/v, `layer_scale1.lambda1` naming)."""
        cfg = hf_model.config
        m = cls(cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, int(cfg.mlp_ratio))
        sd = hf_model.state_dict()
        out = {
            "cls_token": sd["embeddings.cls_token"],
            "pos_embed": sd["embeddings.position_embeddings"],
            "mask_token": sd["embeddings.mask_token"],
            "patch_embed.proj.weight": sd["embeddings.patch_embeddings.projection.weight"],
            "patch_embed.proj.bias": sd["embeddings.patch_embeddings.projection.bias"],
            "norm.weight": sd["layernorm.weight"],
            "norm.bias": sd["layernorm.bias"],
        }
        for i in range(m.depth):
            p = f"encoder.layer.{i}."
            a = p + "attention.attention."
            out[f"blocks.{i}.norm1.weight"] = sd[p + "norm1.weight"]
            out[f"blocks.{i}.norm1.bias"] = sd[p + "norm1.bias"]
            out[f"blocks.{i}.attn.qkv.weight"] = torch.cat([sd[a + "query.weight"], sd[a + "key.weight"], sd[a + "value.weight"]])
            out[f"blocks.{i}.attn.qkv.bias"] = torch.cat([sd[a + "query.bias"], sd[a + "key.bias"], sd[a + "value.bias"]])
            out[f"blocks.{i}.attn.proj.weight"] = sd[p + "attention.output.dense.weight"]
            out[f"blocks.{i}.attn.proj.bias"] = sd[p + "attention.output.dense.bias"]
            out[f"blocks.{i}.ls1.gamma"] = sd[p + "layer_scale1.lambda1"]
            out[f"blocks.{i}.norm2.weight"] = sd[p + "norm2.weight"]
            out[f"blocks.{i}.norm2.bias"] = sd[p + "norm2.bias"]
            out[f"blocks.{i}.mlp.fc1.weight"] = sd[p + "mlp.fc1.weight"]
            out[f"blocks.{i}.mlp.fc1.bias"] = sd[p + "mlp.fc1.bias"]
            out[f"blocks.{i}.mlp.fc2.weight"] = sd[p + "mlp.fc2.weight"]
            out[f"blocks.{i}.mlp.fc2.bias"] = sd[p + "mlp.fc2.bias"]
            out[f"blocks.{i}.ls2.gamma"] = sd[p + "layer_scale2.lambda1"]
        m.load_state_dict(out)
        return m.eval()

    # How a 1 + M*M position table (hub checkpoints: M = 37, trained at 518 x 518) becomes the 1 + 16*16 one used at
    # 224 x 224.  The hub model does this at run time in `interpolate_pos_encoding` (facebookresearch/dinov2,
    # vision_transformer.py; reached from the reference through `forward_features`, ae_net.py:44-47): bicubic
    # F.interpolate of the (M, M) grid with scale_factor = (16 + interpolate_offset) / M and the released models'
    # interpolate_offset = 0.1, interpolate_antialias = False -- the OUTPUT size is floor(M * scale) = 16 but the
    # sampling positions follow 1/scale_factor = M / 16.1, not M / 16 (~0.2 source pixels apart at the grid edge).
    # interpolate_offset = 0.0 is the hub code's other branch (plain `size=(16, 16)`).  Done once on the host at load
    # time: weight preparation, not hot-path arithmetic.  Restated and tested in oracle/vit_numpy.py / tests/test_oracle_vit.py.
    interpolate_offset = 0.1
    interpolate_antialias = False

    def resample_pos_embed(self, pe):
        """(1, 1 + M*M, C) -> (1, 257, C), as the hub model's interpolate_pos_encoding does for a 224 x 224 input."""
        pe = pe.float()
        n = int(round(math.sqrt(pe.shape[1] - 1)))
        if n * n != pe.shape[1] - 1:
            raise ValueError(f"pos_embed with {pe.shape[1]} entries is not 1 + a square grid")
        grid = pe[:, 1:].reshape(1, n, n, -1).permute(0, 3, 1, 2)
        if self.interpolate_offset:
            sf = float(16 + self.interpolate_offset) / n
            grid = nn.functional.interpolate(grid, scale_factor=(sf, sf), mode="bicubic", antialias=self.interpolate_antialias)
        else:
            grid = nn.functional.interpolate(grid, size=(16, 16), mode="bicubic", antialias=self.interpolate_antialias)
        if tuple(grid.shape[-2:]) != (16, 16):
            raise ValueError(f"pos_embed resampling produced {tuple(grid.shape[-2:])}, expected (16, 16)")
        return torch.cat([pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, 256, -1)], dim=1)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        key = prefix + "pos_embed"
        if key in state_dict and state_dict[key].shape[1] != self.pos_embed.shape[1]:
            state_dict[key] = self.resample_pos_embed(state_dict[key])
        self._packed = None
        self.plane_scales = self.plane_amax = None   # new weights: the calibration of the old ones says nothing
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _apply(self, fn, *a, **k):   # .to() / .float(): packed weight copies follow the parameters
        self._packed = None
        return super()._apply(fn, *a, **k)

    # ---------------------------------------------------------------- weight packing
    def invalidate(self):
        """Call after editing parameters in place: packed copies and the plane-scale calibration are dropped."""
        self._packed = None
        self.plane_scales = self.plane_amax = None

    @torch.no_grad()
    def _pack(self, device):
        """Channel-major (pre-transposed) f32 copies of every weight + the host pointer table
        gp_vit_forward expects (order documented in include/gigapose_hip.h)."""
        def dev(t):
            return t.detach().to(device=device, dtype=torch.float32).contiguous()

        C = self.dim
        w = self.patch_embed.proj.weight.detach().float().reshape(C, 588).t()
        patch_wt = torch.zeros(592, C)
        patch_wt[:588] = w.cpu()
        tensors = [dev(patch_wt), dev(self.patch_embed.proj.bias),
                   dev((self.cls_token[0, 0] + self.pos_embed[0, 0])),
                   dev(self.pos_embed[0, 1:].t())]
        for blk in self.blocks:
            qkv_w, qkv_b = blk.attn.qkv.weight, blk.attn.qkv.bias
            qkv_bd = dev(qkv_b)   # ONE tensor: the q|k and v biases are views, so a fused q|k|v launch can read them as one (3C) bias
            tensors += [dev(blk.norm1.weight), dev(blk.norm1.bias),
                        dev(qkv_w[:2 * C].t()), qkv_bd[:2 * C],
                        dev(qkv_w[2 * C:].t()), qkv_bd[2 * C:],
                        dev(blk.attn.proj.weight.t()), dev(blk.attn.proj.bias), dev(blk.ls1.gamma),
                        dev(blk.norm2.weight), dev(blk.norm2.bias),
                        dev(blk.mlp.fc1.weight.t()), dev(blk.mlp.fc1.bias),
                        dev(blk.mlp.fc2.weight.t()), dev(blk.mlp.fc2.bias), dev(blk.ls2.gamma)]
        table = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        split, split_table = [], None
        if self.numerics == "split":  # pre-split weight planes, PyTorch-native [out][in] (k contiguous)
            for blk in self.blocks:
                qkv_w = blk.attn.qkv.weight.detach().to(device)
                ws = (qkv_w[:2 * C], qkv_w[2 * C:], blk.attn.proj.weight.to(device), blk.mlp.fc1.weight.to(device),
                      blk.mlp.fc2.weight.to(device))
                for w in ws:                       # entries 0..9: hi + lo * 2^-11 planes (128 x 128 kernel)
                    split += list(split_planes(w))
                if self.split_gemm != "128":
                    # entries 10..19: x64 single-accumulator planes (256 x 256 kernel; needs |activation| < 8190 --
                    # GIGAPOSE_SPLIT_GEMM=128 keeps every GEMM on the two-accumulator kernel, range 65504).  The q|k and v planes are
                    # views of ONE (3C, C) tensor: gp_vit_forward_split then runs q|k|v as a single launch (768 tiles at B = 64)
                    qkv_hi, qkv_lo = split_planes_x64(qkv_w)
                    split += [qkv_hi[:2 * C], qkv_lo[:2 * C], qkv_hi[2 * C:], qkv_lo[2 * C:]]
                    for w in ws[2:]:
                        split += list(split_planes_x64(w))
            split_table = (ctypes.c_void_p * len(split))(*[t.data_ptr() for t in split])
        self._packed = (device, tensors, table, split, split_table)

    def _workspace(self, B, device):
        lib = _lib.lib()
        lib.gp_vit_workspace_bytes.restype = ctypes.c_size_t
        need = lib.gp_vit_workspace_bytes(_lib.i(B), _lib.i(self.dim), _lib.i(self.mlp_dim))
        if self._ws is None or self._ws.numel() * 4 < need or self._ws.device != device:
            self._ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=device)
        return self._ws, need

    # ---------------------------------------------------------------- plane-scale calibration (split numerics, plane path)
    PLANE_TENSORS = ("ln1", "qkv", "ln2", "gelu")
    CALIBRATION_SCALE = 2.0 ** -4     # the calibration pass itself runs with range 65504 * 16 ~ 1e6 (precision does not matter there)

    def reset_plane_scales(self):
        self.plane_scales, self.plane_amax = None, None

    @staticmethod
    def scale_for(amax, headroom):
        """Largest power of two s <= 8 with amax * s * headroom <= 65504 (>= 2^-10); amax == 0 (tensor never on the plane path) -> 8."""
        if not amax > 0.0:
            return 8.0
        return float(min(8.0, max(2.0 ** -10, 2.0 ** math.floor(math.log2(65504.0 / (amax * headroom))))))

    @torch.no_grad()
    def calibrate_plane_scales(self, images, group=None, chunk=64, sync_ranks=False):
        """One forward over `images` (chunks of <= `chunk`) in calibration mode: every plane producer records max |x| of the tensor it
        writes (gp_vit_forward_split2: plane_amax); the running maximum over all calibration passes of this model picks the scales.
        Returns True if a scale changed.  `sync_ranks`: the ranks of `group` (None = the default group) calibrate together (sharded
        template bank: every rank must end up with the same scales) -- the maxima are all-reduced.  A non-finite activation raises."""
        active = self.numerics == "split" and self.split_gemm != "128"
        if not sync_ranks and (not active or images.shape[0] == 0):
            return False
        # with sync_ranks a rank that has nothing to measure (no image, or not on the plane path) still takes part in the
        # all-reduce below with zeros: its peers are inside that collective
        device = images.device
        amax = torch.zeros(self.depth * 4, dtype=torch.float32, device=device)
        keep = self.plane_scales
        self.plane_scales = [self.CALIBRATION_SCALE] * (self.depth * 4)
        try:
            for s0 in range(0, images.shape[0] if active else 0, chunk):
                self.patch_features(images[s0:s0 + chunk], plane_amax=amax)
        finally:
            self.plane_scales = keep
        if sync_ranks:
            import torch.distributed as dist

            if dist.is_initialized() and dist.get_world_size(group) > 1:
                if dist.get_backend(group) == "gloo":
                    host = amax.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
                    amax = host
                else:
                    dist.all_reduce(amax, op=dist.ReduceOp.MAX, group=group)
        if not active:
            return False
        seen = amax.double().cpu().numpy().reshape(self.depth, 4)      # a host synchronisation: calibration is outside every timed region
        if not bool((seen == seen).all()) or not bool((seen < float("inf")).all()):
            raise _lib.GigaPoseHipError("plane-scale calibration: a ViT activation is not finite (NaN / inf input or weights)")
        self.plane_amax = seen if self.plane_amax is None else __import__("numpy").maximum(self.plane_amax, seen)
        new = [self.scale_for(float(a), self.plane_headroom) for a in self.plane_amax.reshape(-1)]
        old = self.plane_scales or [8.0] * (self.depth * 4)
        changed = new != old
        self.plane_scales = None if all(v == 8.0 for v in new) else new
        return changed

    def adopt_plane_amax(self, amax):
        """Take over a calibration made elsewhere (bank_io: the onboarding process saved it with the bank): running maximum with what
        this model has seen itself, scales re-picked.  Returns True if a scale changed."""
        import numpy as np

        amax = np.asarray(amax, dtype=np.float64).reshape(self.depth, 4)
        if not np.isfinite(amax).all() or (amax < 0).any():
            raise ValueError("plane-scale calibration: the adopted maxima must be finite and non-negative")
        self.plane_amax = amax.copy() if self.plane_amax is None else np.maximum(self.plane_amax, amax)
        new = [self.scale_for(float(a), self.plane_headroom) for a in self.plane_amax.reshape(-1)]
        old = self.plane_scales or [8.0] * (self.depth * 4)
        self.plane_scales = None if all(v == 8.0 for v in new) else new
        return new != old

    def plane_scale_report(self):
        """{(layer, tensor): (amax, scale)} for every tensor whose scale is not the default 8 (diagnostics, bench.py)."""
        if self.plane_scales is None or self.plane_amax is None:
            return {}
        out = {}
        for l in range(self.depth):
            for t, name in enumerate(self.PLANE_TENSORS):
                if self.plane_scales[4 * l + t] != 8.0:
                    out[f"L{l}.{name}"] = (float(self.plane_amax[l, t]), self.plane_scales[4 * l + t])
        return out

    # ---------------------------------------------------------------- forward
    @torch.no_grad()
    def patch_features(self, images, normalize=True, stop_after_layers=-1, plane_amax=None):
        """images (B,3,224,224) f32 -> (B, C, 16, 16): x_prenorm[:, 1:] rearranged 'b (h w) c ->
        b c h w' and (optionally) L2-normalised over C -- i.e. AENet.forward_by_chunk's result."""
        if images.shape[1:] != (3, 224, 224):
            raise ValueError(f"expected (B,3,224,224) crops, got {tuple(images.shape)}")
        device = images.device
        if device.type == "cuda":
            _lib.status_word(device)   # the guard-rail word of THIS GPU (one per device, _lib.py)
        if self._packed is None or self._packed[0] != device:
            self._pack(device)
        B = images.shape[0]
        x = images.contiguous().float()
        out = torch.empty(B, self.dim, 16, 16, dtype=torch.float32, device=device)
        if B == 0:
            return out
        ws, need = self._workspace(B, device)
        _, tensors, table, split, split_table = self._packed
        scales = None
        if self.plane_scales is not None and self.numerics == "split":
            scales = (ctypes.c_float * len(self.plane_scales))(*self.plane_scales)   # host array, read at launch time
        _lib.call("gp_vit_forward_split2", _lib.ptr(x), _lib.i(B), _lib.i(self.dim), _lib.i(self.depth),
                  _lib.i(self.heads), _lib.i(self.mlp_dim), _lib.f(1e-6), table, _lib.i(len(tensors)),
                  split_table, _lib.i(len(split)), _lib.ptr(ws), ctypes.c_size_t(need), _lib.ptr(out),
                  _lib.i(1 if normalize else 0), _lib.i(stop_after_layers), scales, _lib.ptr(plane_amax), _lib.stream_ptr())
        return out

    @torch.no_grad()
    def forward_features(self, images):
        """hub-API compatibility: {"x_prenorm": (B, 257, C)} (token-major view of the workspace)."""
        B = images.shape[0]
        self.patch_features(images, normalize=False)
        mpad = (B * T + 255) // 256 * 256
        xt = self._ws[: self.dim * mpad].view(self.dim, mpad)[:, : B * T]
        return {"x_prenorm": xt.t().reshape(B, T, self.dim).contiguous()}
