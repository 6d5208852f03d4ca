"""IST network behind the reference interface (src/models/network/ist_net.py:11-162,
src/models/network/resnet.py:26-50, 318-381; Hydra targets in configs/model/ist_net/resnet.yaml).

* `ResNet`  -- LoFTR-style stride-16 CNN on the 256x256-resized crop -> (b,256,16,16).  Parameters /
  state-dict names mirror the reference; the forward is hand-written HIP (gp_resize_bilinear_cm +
  gp_conv2d_cm: implicit-GEMM f32-MFMA convolutions with folded eval-BN, residual and ReLU fused).
  It is evaluated ONCE per crop (the reference recomputes it k=5 times, gigaPose.py:553).
* `Regressor` / `ISTNet.inference*` -- gather + concat + two MLP heads run in libgigapose_hip.so
  (gp_ist_regress) for every (detection, hypothesis, patch) row at once.
"""
import ctypes
import os

import torch
from torch import nn

from . import _lib

P = 256


class BasicBlock(nn.Module):
    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, stride=stride, padding=1, bias=False)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1:
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes))

    def forward(self, x):
        # parameter container only: the arithmetic of resnet.py:40-50 runs inside ResNet.forward's HIP launches (gp_conv2d_*); the
        # plain-torch restatement used as checker lives in oracle/ist_torch.py (test infrastructure)
        raise _lib.GigaPoseHipError("BasicBlock has no torch forward in the product package: call ResNet.forward (HIP) "
                                    "or oracle.ist_torch.basic_block (test infrastructure)")


class ResNet(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config["n_heads"] > 0:
            raise NotImplementedError("attention branches are disabled in the released config (n_heads: 0)")
        self.input_size = config["input_size"]
        dims, c0 = list(config["block_dims"]), config["initial_dim"]
        self.conv1 = nn.Conv2d(config["input_dim"], c0, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(c0)
        self.relu = nn.ReLU(inplace=True)
        chans = [c0] + dims
        for i, stride in enumerate([1, 2, 2, 2]):
            setattr(self, f"layer{i + 1}", nn.Sequential(BasicBlock(chans[i], chans[i + 1], stride),
                                                         BasicBlock(chans[i + 1], chans[i + 1], 1)))
        self.layer4_outconv = nn.Conv2d(dims[3], config["descriptor_size"], 1, bias=False)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._packed = None
        self._bufs = None
        self._resized = None
        self._split = None
        self._planes = None
        self._conv_scratch = None
        self.numerics = _lib.default_numerics()  # "split" (default) | "chain" 
        # split numerics: "256" = conv_planes_kernel (gp_conv256.hip: 256-pixel tiles, single accumulator, plane GEMM loop;
        # needs B*OH*OW % 256 == 0, always true from the 16 x 16 output grid up); "128" = the first-generation 128 x 128 kernel
        self.conv_kernel = "256"

    def set_numerics(self, mode):
        if mode not in ("chain", "split"):
            raise ValueError("numerics must be 'chain' or 'split'")
        self.numerics = mode
        return self

    def invalidate(self):
        """Drop the packed (folded-BN / split-plane) weight copies; they are rebuilt at the next forward."""
        self._packed = None
        self._split = None
        self._planes = None

    def _load_from_state_dict(self, *a, **k):
        # nn.Module.load_state_dict on ANY ancestor (GigaPose, a Lightning checkpoint load) recurses through here
        self.invalidate()
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):   # .to() / .float() / .cuda(): the packed copies follow the parameters
        self.invalidate()
        return super()._apply(fn, *a, **k)

    # ------------------------------------------------------------------ HIP path
    @torch.no_grad()
    def _pack(self, device):
        """Per conv: W^T (Kpad, Cout) k-major with k = ci*KH*KW + dy*KW + dx, and eval-BatchNorm folded
        the way ATen folds it: alpha = gamma / sqrt(var + eps), beta = bias - mean * alpha."""
        def conv(c, bn):
            co, ci, kh, kw = c.weight.shape
            k = ci * kh * kw
            wt = torch.zeros((k + 15) // 16 * 16, co, dtype=torch.float32)
            wt[:k] = c.weight.detach().float().reshape(co, k).t().cpu()
            alpha = beta = None
            if bn is not None:
                invstd = 1.0 / torch.sqrt(bn.running_var.detach().float() + bn.eps)
                alpha = (invstd * bn.weight.detach().float()).to(device).contiguous()
                beta = (bn.bias.detach().float() - bn.running_mean.detach().float() * invstd * bn.weight.detach().float()).to(device).contiguous()
            return dict(wt=wt.to(device).contiguous(), alpha=alpha, beta=beta, cin=ci, cout=co, k=kh,
                        stride=c.stride[0], pad=c.padding[0])

        blocks = []
        for li in range(1, 5):
            for blk in getattr(self, f"layer{li}"):
                ds = None if blk.downsample is None else conv(blk.downsample[0], blk.downsample[1])
                blocks.append((conv(blk.conv1, blk.bn1), conv(blk.conv2, blk.bn2), ds))
        self._packed = dict(device=device, stem=conv(self.conv1, self.bn1), blocks=blocks,
                            out=conv(self.layer4_outconv, None))

    @torch.no_grad()
    def _pack_split(self, device):
        """Split numerics: weights as f16 planes (round_up(Cout,128), KH*KW*Cin), k = (dy, dx, ci) -- channel-last
        im2col order -- from the same folded-BN packing."""
        from .vit import split_planes

        def conv(c):
            co, ci, kh, kw = c.weight.shape
            w = torch.zeros((co + 127) // 128 * 128, kh * kw * ci, dtype=torch.float32, device=device)
            w[:co] = c.weight.detach().float().to(device).permute(0, 2, 3, 1).reshape(co, kh * kw * ci)
            return split_planes(w)

        blocks = []
        for li in range(1, 5):
            for blk in getattr(self, f"layer{li}"):
                blocks.append((conv(blk.conv1), conv(blk.conv2), None if blk.downsample is None else conv(blk.downsample[0])))
        self._split = dict(device=device, blocks=blocks, out=conv(self.layer4_outconv))

    @torch.no_grad()
    def _pack_planes(self, device):
        """conv_planes_kernel's weights: (Cout, KH*KW*Cin) planes of 64 w (hi + lo, unscaled lo), k = (dy, dx, ci)."""
        from .vit import split_planes_x64

        def conv(c):
            co, ci, kh, kw = c.weight.shape
            return split_planes_x64(c.weight.detach().float().to(device).permute(0, 2, 3, 1).reshape(co, kh * kw * ci).contiguous())

        blocks = []
        for li in range(1, 5):
            for blk in getattr(self, f"layer{li}"):
                blocks.append((conv(blk.conv1), conv(blk.conv2), None if blk.downsample is None else conv(blk.downsample[0])))
        # stem (7 x 7, 3 channels): k = dy * 32 + dx * 4 + ci over a kernel row of 8 taps x 4 channels (dx = 7 and ci = 3 are zeros)
        w = self.conv1.weight.detach().float().to(device)                     # (Cout, 3, 7, 7)
        ws = torch.zeros(w.shape[0], 7, 8, 4, dtype=torch.float32, device=device)
        ws[:, :, :7, :3] = w.permute(0, 2, 3, 1)
        stem = split_planes_x64(ws.reshape(w.shape[0], 224).contiguous()) if tuple(w.shape[1:]) == (3, 7, 7) else None
        self._planes = dict(device=device, blocks=blocks, out=conv(self.layer4_outconv), stem=stem)

    def _conv_planes(self, cv, w, x, y, B, H, W, residual=None, relu=True, out_f32=None):
        """x, y, residual: (hi, lo) f16 plane pairs, channel-last, x 8 (gp_conv2d_planes)."""
        dev = x[0].device
        lib = _lib.lib()
        lib.gp_conv2d_planes_workspace_bytes.restype = ctypes.c_size_t
        need = lib.gp_conv2d_planes_workspace_bytes()
        if self._conv_scratch is None or self._conv_scratch.device != dev:
            self._conv_scratch = torch.zeros((need + 3) // 4, dtype=torch.float32, device=dev)
        _lib.call("gp_conv2d_planes", _lib.ptr(x[0]), _lib.ptr(x[1]), _lib.ptr(w[0]), _lib.ptr(w[1]),
                  _lib.ptr(cv["alpha"]), _lib.ptr(cv["beta"]), _lib.ptr(None if residual is None else residual[0]),
                  _lib.ptr(None if residual is None else residual[1]), _lib.i(B), _lib.i(H), _lib.i(W), _lib.i(cv["cin"]),
                  _lib.i(cv["cout"]), _lib.i(cv["k"]), _lib.i(cv["k"]), _lib.i(cv["stride"]), _lib.i(cv["pad"]),
                  _lib.i(1 if relu else 0), _lib.ptr(None if y is None else y[0]), _lib.ptr(None if y is None else y[1]),
                  _lib.ptr(out_f32), _lib.ptr(self._conv_scratch), ctypes.c_size_t(need), _lib.stream_ptr())
        oh = (H + 2 * cv["pad"] - cv["k"]) // cv["stride"] + 1
        return oh, (W + 2 * cv["pad"] - cv["k"]) // cv["stride"] + 1

    def _conv_split(self, cv, w, x, y, B, H, W, residual=None, relu=True, out_f32=None):
        """x, y, residual: (hi, lo) f16 plane pairs, channel-last."""
        _lib.call("gp_conv2d_nhwc_split", _lib.ptr(x[0]), _lib.ptr(x[1]), _lib.ptr(w[0]), _lib.ptr(w[1]),
                  _lib.ptr(cv["alpha"]), _lib.ptr(cv["beta"]), _lib.ptr(None if residual is None else residual[0]),
                  _lib.ptr(None if residual is None else residual[1]), _lib.i(B), _lib.i(H), _lib.i(W), _lib.i(cv["cin"]),
                  _lib.i(cv["cout"]), _lib.i(cv["k"]), _lib.i(cv["k"]), _lib.i(cv["stride"]), _lib.i(cv["pad"]),
                  _lib.i(1 if relu else 0), _lib.ptr(None if y is None else y[0]), _lib.ptr(None if y is None else y[1]),
                  _lib.ptr(out_f32), _lib.stream_ptr())
        oh = (H + 2 * cv["pad"] - cv["k"]) // cv["stride"] + 1
        return oh, (W + 2 * cv["pad"] - cv["k"]) // cv["stride"] + 1

    def _conv(self, cv, x, y, B, H, W, residual=None, relu=True, nchw_out=False):
        _lib.call("gp_conv2d_cm", _lib.ptr(x), _lib.ptr(cv["wt"]), _lib.ptr(y), _lib.ptr(cv["alpha"]),
                  _lib.ptr(cv["beta"]), _lib.ptr(residual), _lib.i(cv["cin"]), _lib.i(B), _lib.i(H), _lib.i(W),
                  _lib.i(cv["cout"]), _lib.i(cv["k"]), _lib.i(cv["k"]), _lib.i(cv["stride"]), _lib.i(cv["pad"]),
                  _lib.i(1 if relu else 0), _lib.i(1 if nchw_out else 0), _lib.stream_ptr())
        oh = (H + 2 * cv["pad"] - cv["k"]) // cv["stride"] + 1
        return oh, (W + 2 * cv["pad"] - cv["k"]) // cv["stride"] + 1

    @torch.no_grad()
    def forward(self, x):
        """(b,3,224,224) crops -> (b,descriptor,16,16).  HIP only: bilinear resize, 21 implicit-GEMM
        convolutions with fused BN/residual/ReLU (gp_conv2d_cm), activations channel-major."""
        if not x.is_cuda:
            raise _lib.GigaPoseHipError("ResNet.forward runs on the GPU only (the plain-torch checker is oracle.ist_torch.resnet_forward)")
        dev = x.device
        if getattr(self, "_packed", None) is None or self._packed["device"] != dev:
            self._pack(dev)
        pk = self._packed
        B, S = x.shape[0], self.input_size
        cdesc = pk["out"]["cout"]
        out = torch.empty(B, cdesc, S // 16, S // 16, dtype=torch.float32, device=dev)
        if B == 0:
            return out
        need = pk["stem"]["cout"] * B * (S // 2) * (S // 2)
        if getattr(self, "_bufs", None) is None or self._bufs[0].numel() < need or self._bufs[0].device != dev:
            self._bufs = [torch.empty(need, dtype=torch.float32, device=dev) for _ in range(4)]
            self._resized = None
        if self._resized is None or self._resized.numel() < 3 * B * S * S:
            self._resized = torch.empty(3 * B * S * S, dtype=torch.float32, device=dev)
        xin = x.contiguous().float()
        cur, y1, sc, nxt = self._bufs
        if self.numerics == "split" and self.conv_kernel == "256" and self._stem_planes_ok(pk):
            return self._forward_split(pk, None, out, B, S // 2, S // 2, (y1, sc, nxt), images=xin)   # stem in split numerics too
        _lib.call("gp_resize_bilinear_cm", _lib.ptr(xin), _lib.ptr(self._resized), _lib.i(B), _lib.i(3),
                  _lib.i(x.shape[2]), _lib.i(x.shape[3]), _lib.i(S), _lib.stream_ptr())
        H, W = self._conv(pk["stem"], self._resized, cur, B, S, S)
        if self.numerics == "split":
            return self._forward_split(pk, cur, out, B, H, W, (y1, sc, nxt))
        for c1, c2, ds in pk["blocks"]:
            oh, ow = self._conv(c1, cur, y1, B, H, W)                       # relu(bn1(conv1(x)))
            short = cur
            if ds is not None:
                self._conv(ds, cur, sc, B, H, W, relu=False)                # bn(conv1x1(x))
                short = sc
            self._conv(c2, y1, nxt, B, oh, ow, residual=short)              # relu(shortcut + bn2(conv2(.)))
            cur, nxt = nxt, cur
            H, W = oh, ow
        self._conv(pk["out"], cur, out, B, H, W, relu=False, nchw_out=True)
        return out


    def _stem_planes_ok(self, pk):
        st = pk["stem"]
        return (st["cin"], st["k"], st["stride"], st["pad"]) == (3, 7, 2, 3) and st["cout"] in (128, 192, 256) and self.input_size % 2 == 0

    def _forward_split(self, pk, stem_out, out, B, H, W, f32_bufs, images=None):
        """The ResNet in split numerics on channel-last f16 planes.  conv_kernel "256" (default): gp_conv2d_planes for every layer (3 x 3 /
        stride 1 layers run its halo kernel) and, when `images` is given, the stem too (resize -> framed 4-channel planes ->
        gp_conv2d_stem_planes).  conv_kernel "128" (the wide-range fallback): gp_conv2d_nhwc_split after an f32 stem whose channel-major
        output `stem_out` is split + transposed once.  The plane buffers alias the f32 ping-pong buffers (2 planes x f16 = f32)."""
        dev = stem_out.device if stem_out is not None else images.device
        c0 = pk["stem"]["cout"]
        npix = B * H * W
        use256 = self.conv_kernel == "256"   # every later layer has B*OH*OW = B * 256 * 4^n pixels: multiples of 256
        if use256:
            if self._planes is None or self._planes["device"] != dev:
                self._pack_planes(dev)
            weights, conv = self._planes, self._conv_planes
        else:
            if self._split is None or self._split["device"] != dev:
                self._pack_split(dev)
            weights, conv = self._split, self._conv_split

        def planes(buf):  # one f32 buffer -> (hi, lo) f16 planes of the same total size
            h = buf.view(torch.float16)
            return h[: h.numel() // 2], h[h.numel() // 2:]

        y1, sc, nxt = (planes(b) for b in f32_bufs)
        if getattr(self, "_stem_planes", None) is None or self._stem_planes[0].numel() < c0 * npix or \
                self._stem_planes[0].device != dev:
            self._stem_planes = (torch.empty(c0 * npix, dtype=torch.float16, device=dev),
                                 torch.empty(c0 * npix, dtype=torch.float16, device=dev))
        cur = self._stem_planes
        if images is not None:   # resize -> zero-framed 4-channel planes -> the stem as conv_planes_kernel (one kernel row per k-step)
            S = self.input_size
            if getattr(self, "_framed", None) is None or self._framed[0].shape[0] != B or self._framed[0].device != dev:
                self._framed = (torch.zeros(B, S + 6, S + 8, 4, dtype=torch.float16, device=dev),
                                torch.zeros(B, S + 6, S + 8, 4, dtype=torch.float16, device=dev))   # the frame stays zero
            _lib.call("gp_resize_stem_planes", _lib.ptr(images), _lib.ptr(self._framed[0]), _lib.ptr(self._framed[1]), _lib.i(B),
                      _lib.i(images.shape[2]), _lib.i(images.shape[3]), _lib.i(S), _lib.stream_ptr())
            lib = _lib.lib()
            lib.gp_conv2d_planes_workspace_bytes.restype = ctypes.c_size_t
            need = lib.gp_conv2d_planes_workspace_bytes()
            if self._conv_scratch is None or self._conv_scratch.device != dev:
                self._conv_scratch = torch.zeros((need + 3) // 4, dtype=torch.float32, device=dev)
            st, sw = pk["stem"], weights["stem"]
            _lib.call("gp_conv2d_stem_planes", _lib.ptr(self._framed[0]), _lib.ptr(self._framed[1]), _lib.ptr(sw[0]), _lib.ptr(sw[1]),
                      _lib.ptr(st["alpha"]), _lib.ptr(st["beta"]), _lib.i(B), _lib.i(S), _lib.i(c0), _lib.i(1), _lib.ptr(cur[0]), _lib.ptr(cur[1]),
                      _lib.ptr(self._conv_scratch), ctypes.c_size_t(need), _lib.stream_ptr())
        elif use256:   # [C][npix] f32 -> [npix][C] planes of 8 x (single-accumulator convention)
            _lib.call("gp_planes_from_cm", _lib.ptr(stem_out), _lib.i(c0), _lib.i(npix), _lib.ptr(cur[0]), _lib.ptr(cur[1]), _lib.stream_ptr())
        else:        # ... of x with the low half scaled by 2^11 (two-accumulator convention)
            _lib.call("gp_split_weights", _lib.ptr(stem_out), _lib.i(c0), _lib.i(npix), _lib.i(npix), _lib.ptr(cur[0]),
                      _lib.ptr(cur[1]), _lib.stream_ptr())
        for (c1, c2, ds), (w1, w2, wd) in zip(pk["blocks"], weights["blocks"]):
            oh, ow = conv(c1, w1, cur, y1, B, H, W)                             # relu(bn1(conv1(x)))
            short = cur
            if ds is not None:
                conv(ds, wd, cur, sc, B, H, W, relu=False)                      # bn(conv1x1(x))
                short = sc
            conv(c2, w2, y1, nxt, B, oh, ow, residual=short)                    # relu(shortcut + bn2(conv2(.)))
            cur, nxt = nxt, cur
            H, W = oh, ow
        conv(pk["out"], weights["out"], cur, None, B, H, W, relu=False, out_f32=out)
        return out


class Regressor(nn.Module):
    def __init__(self, descriptor_size, hidden_dim, use_tanh_act, normalize_output):
        super().__init__()
        self.descriptor_size, self.hidden_dim = descriptor_size, hidden_dim
        self.normalize_output, self.use_tanh_act = normalize_output, use_tanh_act

        def mlp(nout, tail):
            return nn.Sequential(nn.Linear(descriptor_size * 2, hidden_dim * 2), nn.ReLU(inplace=True),
                                 nn.Linear(hidden_dim * 2, hidden_dim), nn.ReLU(inplace=True),
                                 nn.Linear(hidden_dim, nout), *tail)

        self.scale_predictor = mlp(1, [])
        self.inplane_predictor = mlp(2, [nn.Tanh() if use_tanh_act else nn.Identity()])
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class ISTNet(nn.Module):
    def __init__(self, model_name, backbone, regressor, max_batch_size, patch_size=14, **kwargs):
        super().__init__()
        self.model_name, self.patch_size = model_name, patch_size
        self.backbone, self.regressor = backbone, regressor
        self.max_batch_size = max_batch_size
        for m in self.modules():  # reference ist_net.py:33-42
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        self._packed = None
        self._ws = None

    def _load_from_state_dict(self, *a, **k):
        self._packed = None   # regressor weight table; reached from a parent module's load_state_dict too
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def forward_by_chunk(self, processed_rgbs):
        outs = [self.backbone(processed_rgbs[s:s + self.max_batch_size])
                for s in range(0, processed_rgbs.shape[0], self.max_batch_size)]
        if not outs:
            return torch.empty(0, self.regressor.descriptor_size, 16, 16, device=processed_rgbs.device)
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    # ------------------------------------------------------------------ HIP regressor
    head_numerics = None   # None: the MLP heads follow the backbone's numerics; "chain" / "split": on their own (tools/probe_parity_attribution.py)

    def _head_numerics(self):
        return self.head_numerics or getattr(self.backbone, "numerics", "chain")

    @torch.no_grad()
    def _pack(self, device):
        def dev(t):
            return t.detach().to(device=device, dtype=torch.float32).contiguous()

        tensors = []
        for seq in (self.regressor.scale_predictor, self.regressor.inplane_predictor):
            l1, l2, l3 = seq[0], seq[2], seq[4]
            tensors += [dev(l1.weight.t()), dev(l1.bias), dev(l2.weight.t()), dev(l2.bias), dev(l3.weight), dev(l3.bias)]
        numerics = self._head_numerics()
        if numerics == "split":   # the two hidden layers of each head as 3 x f16 MFMA (weights [out][in] as hi / lo planes)
            from .vit import split_planes

            for seq in (self.regressor.scale_predictor, self.regressor.inplane_predictor):
                tensors += list(split_planes(seq[0].weight.to(device))) + list(split_planes(seq[2].weight.to(device)))
        table = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        self._packed = (device, tensors, table, numerics)

    @torch.no_grad()
    def regress_bank(self, ist_bank, labels0, id_src, tar_feat, src_pts, tar_pts):
        """All hypotheses at once against a resident IST bank.
        ist_bank (O,N,D,16,16); labels0 (B) int32 0-based; id_src (B,k) int64; tar_feat (B,D,16,16);
        src_pts / tar_pts (B,k,256,2) int64 -> relScale (B,k,256), relInplane (B,k,256,2)."""
        dev = tar_feat.device
        if self._packed is None or self._packed[0] != dev or self._packed[3] != self._head_numerics():
            self._pack(dev)
        B, k = id_src.shape
        O, N, D = ist_bank.shape[:3]
        H = self.regressor.hidden_dim
        scales = torch.empty(B, k, P, dtype=torch.float32, device=dev)
        cos_sin = torch.empty(B, k, P, 2, dtype=torch.float32, device=dev)
        if B == 0:
            return scales, cos_sin
        lib = _lib.lib()
        lib.gp_ist_workspace_bytes.restype = ctypes.c_size_t
        need = lib.gp_ist_workspace_bytes(_lib.i(B), _lib.i(k), _lib.i(D), _lib.i(H))
        if self._ws is None or self._ws.numel() * 4 < need or self._ws.device != dev:
            # zeroed once: the regressor's GEMMs run over compacted rows and the last 128-column tile is only partly filled --
            # its stale columns (never read back) must at least be finite, whatever the allocator handed out
            self._ws = torch.zeros((need + 3) // 4, dtype=torch.float32, device=dev)
        _, tensors, table, _ = self._packed
        _lib.call("gp_ist_regress", _lib.ptr(tar_feat.contiguous().float()), _lib.ptr(ist_bank.contiguous()),
                  _lib.ptr(labels0.to(torch.int32).contiguous()), _lib.ptr(id_src.contiguous()),
                  _lib.ptr(tar_pts.contiguous()), _lib.ptr(src_pts.contiguous()), _lib.i(B), _lib.i(O), _lib.i(N),
                  _lib.i(k), _lib.i(D), _lib.i(H), table, _lib.i(len(tensors)), _lib.i(1 if self.regressor.use_tanh_act else 0),
                  _lib.ptr(self._ws), ctypes.c_size_t(need), _lib.ptr(scales), _lib.ptr(cos_sin), _lib.stream_ptr())
        return scales, cos_sin

    @torch.no_grad()
    def inference(self, src_feat, tar_feat, src_pts, tar_pts):
        """Reference signature (ist_net.py:97): src_feat/tar_feat (B,D,16,16), pts (B,P,2) ->
        (B,P) scales and (B,P,2) cos/sin, -1000 where the correspondence is invalid.  As in the
        reference, cos/sin are NOT re-normalised here although normalize_output is set."""
        B = src_feat.shape[0]
        dev = src_feat.device
        sv = (src_pts[..., 0] != -1) & (src_pts[..., 1] != -1)
        tv = (tar_pts[..., 0] != -1) & (tar_pts[..., 1] != -1)
        assert int(sv.sum()) == int(tv.sum())  # reference ist_net.py:116
        labels0 = torch.arange(B, dtype=torch.int32, device=dev)
        id_src = torch.zeros(B, 1, dtype=torch.int64, device=dev)
        s, c = self.regress_bank(src_feat.unsqueeze(1), labels0, id_src, tar_feat, src_pts.unsqueeze(1),
                                 tar_pts.unsqueeze(1))
        return s[:, 0], c[:, 0]

    def inference_by_chunk(self, src_feat, tar_feat, src_pts, tar_pts, max_batch_size):
        outs = [self.inference(src_feat[s:s + max_batch_size], tar_feat[s:s + max_batch_size],
                               src_pts[s:s + max_batch_size], tar_pts[s:s + max_batch_size])
                for s in range(0, src_feat.shape[0], max_batch_size)]
        return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
