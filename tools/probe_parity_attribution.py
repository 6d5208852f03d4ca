"""Which stage owns the config-2 parity gap?  

BASELINE config 2 end to end against the float64 run of the unmodified reference (tests/golden/e2e_cfg2_margins.npz, the checker of
tests/test_gpu_parity_big.py): the reference's own f32 run keeps 312 of 320 hypotheses on the float64 run's discrete path, `chain` 305,
`split` 301.  This tool re-runs the SPLIT model with ONE stage at a time replaced by float64 arithmetic on that stage's actual inputs
(its output rounded to the representation the next stage reads: 22-bit planes, or f32), and records `hyp_same_all` for each.

The ViT forward is rebuilt here from the C-ABI's stage entry points (gp_layernorm_planes, gp_gemm_planes256_ragged, gp_attention_split)
exactly as gp_vit_forward_split2 strings them together (q | k | v as one launch, in-place residual epilogues), so that `ideal=None`
reproduces the product's residual stream bit for bit (asserted) and a stage can be swapped from Python:
    ln     both LayerNorms in float64 (output still rounded to planes)
    qkv / attn / proj / fc1 / fc2     that stage in float64
    gemms  all four linear layers in float64
    ln+   LayerNorm AND its consumer GEMM fused in float64 (no 22-bit rounding of the LayerNorm output in between)
    all    every ViT stage in float64 (what is left is the representation between stages: 22-bit planes / f32 stream)
and, outside the ViT: the matcher and the IST network in chain numerics (f32 fmaf chains) next to a split ViT.
    python tools/probe_parity_attribution.py [variants ...]      (GPU box; ~1 min per variant)
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gigapose_amd import _lib  # noqa: E402
from gigapose_amd.vit import split_planes_x64  # noqa: E402

DEV = "cuda"
T = 257


def planes_of(x, scale=8.0):
    """f32 (rows, cols) -> hi / lo planes of scale * x, with the device's own split (gp_split_planes)."""
    x = x.float().contiguous()
    hi = torch.empty(x.shape, dtype=torch.float16, device=DEV)
    lo = torch.empty_like(hi)
    _lib.call("gp_split_planes", _lib.ptr(x), ctypes.c_size_t(x.numel()), _lib.f(scale), _lib.ptr(hi), _lib.ptr(lo), _lib.stream_ptr())
    return hi, lo


def val(hi, lo, scale=8.0):
    return (hi.double() + lo.double()) / scale


class HybridViT:
    """The split plane path of one Dinov2ViT, stage by stage (see module docstring)."""

    def __init__(self, vit):
        self.vit = vit
        lib = _lib.lib()
        lib.gp_gemm_split256_workspace_bytes.restype = ctypes.c_size_t
        self.nb = lib.gp_gemm_split256_workspace_bytes()
        self.ws = torch.zeros(self.nb // 4, device=DEV)
        f32 = lambda t: t.detach().float().to(DEV).contiguous()
        self.layers = []
        for blk in vit.blocks:
            L = dict(g1=f32(blk.norm1.weight), b1=f32(blk.norm1.bias), g2=f32(blk.norm2.weight), b2=f32(blk.norm2.bias),
                     Wqkv=f32(blk.attn.qkv.weight), bqkv=f32(blk.attn.qkv.bias), Wp=f32(blk.attn.proj.weight), bp=f32(blk.attn.proj.bias),
                     ls1=f32(blk.ls1.gamma), W1=f32(blk.mlp.fc1.weight), bf1=f32(blk.mlp.fc1.bias), W2=f32(blk.mlp.fc2.weight),
                     bf2=f32(blk.mlp.fc2.bias), ls2=f32(blk.ls2.gamma))
            for n in ("Wqkv", "Wp", "W1", "W2"):
                L[n + "_pl"] = split_planes_x64(L[n])
            self.layers.append(L)

    def gemm(self, wpl, bhi, blo, epi, bias, K, I, Mpad, Mtok, out_planes, scale=None, res=None):
        D = res if res is not None else None
        ohi = olo = None
        if out_planes:
            ohi = torch.zeros(Mpad, I, dtype=torch.float16, device=DEV)
            olo = torch.zeros_like(ohi)
        _lib.call("gp_gemm_planes256_scaled", _lib.ptr(wpl[0]), _lib.ptr(wpl[1]), _lib.ptr(bhi), _lib.ptr(blo), _lib.ptr(D), _lib.i(Mpad),
                  _lib.ptr(ohi), _lib.ptr(olo), _lib.i(I), _lib.i(I), _lib.i(Mpad), _lib.i(Mtok), _lib.i(K), _lib.i(epi), _lib.ptr(bias),
                  _lib.ptr(scale), _lib.ptr(D), _lib.i(Mpad if res is not None else 0), _lib.f(1.0 / 512.0), _lib.f(8.0), _lib.ptr(None), _lib.ptr(self.ws),
                  ctypes.c_size_t(self.nb), _lib.stream_ptr())
        return (ohi, olo) if out_planes else D

    def ln(self, X, g, b, Mpad):
        C = X.shape[0]
        hi = torch.zeros(Mpad, C, dtype=torch.float16, device=DEV)
        lo = torch.zeros_like(hi)
        _lib.call("gp_layernorm_planes", _lib.ptr(X), _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(g), _lib.ptr(b), _lib.i(C), _lib.i(Mpad), _lib.f(1e-6),
                  _lib.stream_ptr())
        return hi, lo

    @torch.no_grad()
    def residual_stream(self, images, ideal=None):
        """X [C][Mpad] f32 after all layers for a batch of exactly 64 images."""
        vit = self.vit
        B, C, H = images.shape[0], vit.dim, vit.heads
        Mtok, Mpad = B * T, (B * T + 255) // 256 * 256
        vit.patch_features(images, normalize=False, stop_after_layers=0)
        X = vit._ws[: C * Mpad].view(C, Mpad).clone()
        ideal = set(ideal or ())
        f64 = lambda t: t.double()
        LN = torch.nn.functional.layer_norm

        def attn64(qkv):
            q, k, v = [qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
            return (torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v).permute(0, 2, 1, 3).reshape(Mtok, C)

        def pad(t):   # (Mtok, n) -> (Mpad, n)
            out = torch.zeros(Mpad, t.shape[1], dtype=t.dtype, device=DEV)
            out[:Mtok] = t
            return out

        for L in self.layers:
            # LayerNorm 1 -> planes
            if "ln" in ideal or "all" in ideal:
                h1 = planes_of(pad(LN(X.t()[:Mtok].double(), (C,), f64(L["g1"]), f64(L["b1"]), 1e-6)))
            else:
                h1 = self.ln(X, L["g1"], L["b1"], Mpad)
            # q | k | v -> planes
            if "ln+" in ideal:
                a = planes_of(pad(LN(X.t()[:Mtok].double(), (C,), f64(L["g1"]), f64(L["b1"]), 1e-6) @ f64(L["Wqkv"]).t() + f64(L["bqkv"])))
            elif ideal & {"qkv", "gemms", "all"}:
                a = planes_of(pad(val(*h1)[:Mtok] @ f64(L["Wqkv"]).t() + f64(L["bqkv"])))
            else:
                a = self.gemm(L["Wqkv_pl"], h1[0], h1[1], 7, L["bqkv"], C, 3 * C, Mpad, Mtok, True)
            # attention -> planes
            if ideal & {"attn", "all"}:
                o = planes_of(pad(attn64(val(*a)[:Mtok].view(B, T, 3, H, 64))))
            else:
                o = (torch.zeros(Mpad, C, dtype=torch.float16, device=DEV), torch.zeros(Mpad, C, dtype=torch.float16, device=DEV))
                _lib.call("gp_attention_split_scaled", _lib.ptr(a[0]), _lib.ptr(a[1]), _lib.ptr(o[0]), _lib.ptr(o[1]), _lib.i(B), _lib.i(H), _lib.i(C), _lib.i(Mpad),
                          _lib.f(8.0), _lib.stream_ptr())
            # x += ls1 * proj(.)
            if ideal & {"proj", "gemms", "all"}:
                X1 = X.clone()
                X1.t()[:Mtok] = (X.t()[:Mtok].double() + f64(L["ls1"]) * (val(*o)[:Mtok] @ f64(L["Wp"]).t() + f64(L["bp"]))).float()
                X = X1
            else:
                X = self.gemm(L["Wp_pl"], o[0], o[1], 3, L["bp"], C, C, Mpad, Mtok, False, scale=L["ls1"], res=X)
            # LayerNorm 2 -> planes
            if "ln" in ideal or "all" in ideal:
                h2 = planes_of(pad(LN(X.t()[:Mtok].double(), (C,), f64(L["g2"]), f64(L["b2"]), 1e-6)))
            else:
                h2 = self.ln(X, L["g2"], L["b2"], Mpad)
            # gelu(fc1(.)) -> planes
            gelu = torch.nn.functional.gelu
            if "ln+" in ideal:
                f = planes_of(pad(gelu(LN(X.t()[:Mtok].double(), (C,), f64(L["g2"]), f64(L["b2"]), 1e-6) @ f64(L["W1"]).t() + f64(L["bf1"]))))
            elif ideal & {"fc1", "gemms", "all"}:
                f = planes_of(pad(gelu(val(*h2)[:Mtok] @ f64(L["W1"]).t() + f64(L["bf1"]))))
            else:
                f = self.gemm(L["W1_pl"], h2[0], h2[1], 6, L["bf1"], C, vit.mlp_dim, Mpad, Mtok, True)
            # x += ls2 * fc2(.)
            if ideal & {"fc2", "gemms", "all"}:
                X2 = X.clone()
                X2.t()[:Mtok] = (X.t()[:Mtok].double() + f64(L["ls2"]) * (val(*f)[:Mtok] @ f64(L["W2"]).t() + f64(L["bf2"]))).float()
                X = X2
            else:
                X = self.gemm(L["W2_pl"], f[0], f[1], 3, L["bf2"], vit.mlp_dim, C, Mpad, Mtok, False, scale=L["ls2"], res=X)
        return X

    @torch.no_grad()
    def features(self, images, ideal=None):
        """AENet.forward's result (B, C, 16, 16) through the hybrid forward, in batches of exactly 64 (the last one padded with its own
        first images, as the plane path needs the 64-crop tile partition)."""
        outs = []
        for s0 in range(0, images.shape[0], 64):
            x = images[s0:s0 + 64]
            n = x.shape[0]
            if n < 64:
                x = torch.cat([x, images[: 64 - n]])
            X = self.residual_stream(x.contiguous(), ideal)
            C = X.shape[0]
            tok = X[:, : 64 * T].view(C, 64, T)[:, :, 1:].permute(1, 0, 2).double()          # (64, C, 256)
            outs.append((tok / tok.norm(dim=1, keepdim=True).clamp_min(1e-12)).float().reshape(64, C, 16, 16)[:n])
        return torch.cat(outs)


def run_variant(name, golden_dir, report):
    import parity_explain as px
    from test_gpu_parity_big import E2E_CONFIGS, EPS_PX, EPS_SIM, build_e2e_model, ours_for_checker

    cfg = E2E_CONFIGS["e2e_cfg2"]
    m = dict(np.load(os.path.join(golden_dir, "e2e_cfg2_margins.npz")))
    t0 = time.time()
    vit_numerics = "chain" if name == "vit_chain" else "split"
    model, batch, q = build_e2e_model(cfg, "split")
    if name == "product_chain":
        model.set_numerics("chain")
    if name == "vit_chain":
        model.ae_net.dinov2_model.set_numerics("chain")
    if name == "matcher_chain":
        model.testing_metric.numerics = "chain"
    if name == "ist_chain":
        model.ist_net.backbone.set_numerics("chain")
    if name == "ist_backbone_chain":     # the ResNet in f32 chains, the two MLP heads in split numerics
        model.ist_net.backbone.set_numerics("chain")
        model.ist_net.head_numerics = "split"
    if name == "ist_heads_chain":        # the ResNet in split numerics, the heads in f32 chains
        model.ist_net.head_numerics = "chain"
    ideal = None
    if name.startswith("ideal:"):
        ideal = name.split(":", 1)[1].split(",")
    if name == "hybrid" or ideal is not None:
        hv = HybridViT(model.ae_net.dinov2_model)
        if name == "hybrid":   # the rebuilt forward IS the product's: same residual stream, bit for bit, on a real batch
            x = torch.from_numpy(q["tar_img"]).to(DEV)
            Xh = hv.residual_stream(x)
            model.ae_net.dinov2_model.patch_features(x, normalize=False)
            Xp = model.ae_net.dinov2_model._ws[: Xh.numel()].view_as(Xh)
            assert torch.equal(Xh, Xp), "the stage-by-stage forward differs from gp_vit_forward_split2"
        model.ae_net.forward = lambda imgs: hv.features(imgs, ideal)
        model.ae_net.forward_by_chunk = lambda imgs, patch_dim=(2, 3): hv.features(imgs, ideal)
    cap = {}
    match_tiles = model.testing_metric.match_tiles

    def spy(*a, **kw):
        cap["tiles"] = match_tiles(*a, **kw)
        return cap["tiles"]

    model.testing_metric.match_tiles = spy
    model.test_step(batch, 0)
    model.flush_pending()
    p = {n: v.cpu().numpy() for n, v in model.last_predictions.tensors.items()}
    geom = px.geometry(cfg["seed"], cfg["O"], cfg["N"], cfg["B"])
    rep = px.explain(m, ours_for_checker(model, p, cap["tiles"], m), eps_sim=EPS_SIM, eps_px=EPS_PX, geom=geom)
    # feature error against the float64 forward (2 templates + 2 crops, every 4th channel), in a batch of 64
    from test_gpu_e2e import e2e_inputs

    g64 = np.load(os.path.join(golden_dir, "e2e_cfg2_f64.npz"))
    items, qq = e2e_inputs(cfg["seed"], cfg["O"], cfg["N"], cfg["B"])
    x = torch.cat([items[0].rgb[:2], torch.from_numpy(qq["tar_img"][:2]), items[0].rgb[2:62]]).to(DEV)
    mine = model.ae_net(x).cpu().numpy()[:4, ::4].astype(np.float64)
    e = np.abs(mine - g64["feat_f64_templates01_crops01"])
    # IST regressions against the float64 run's, over the hypotheses both runs hold with identical correspondences
    o = ours_for_checker(model, p, cap["tiles"], m)
    ds, dc = [], []
    for b in range(o["id_src"].shape[0]):
        for jo, n in enumerate(o["id_src"][b]):
            jf = np.flatnonzero(m["id_src"][b] == n)
            if len(jf) and (o["src_pts"][b, jo] == m["src_pts"][b, jf[0]]).all() and (o["tar_pts"][b, jo] == m["tar_pts"][b, jf[0]]).all():
                ok = m["src_pts"][b, jf[0]][:, 0] != -1
                ds.append((o["relScale"][b, jo][ok] - m["relScale"][b, jf[0]][ok]).ravel())
                dc.append((o["relInplane"][b, jo][ok] - m["relInplane"][b, jf[0]][ok]).ravel())
    ds, dc = np.concatenate(ds), np.concatenate(dc)
    ist = f"IST vs f64: relScale rms {np.sqrt((ds ** 2).mean()):.2e} max {np.abs(ds).max():.1e}, cos/sin rms {np.sqrt((dc ** 2).mean()):.2e} max {np.abs(dc).max():.1e}"
    line = (f"{name:28s} hyp_same_all {rep['hyp_same_all']:3d} / {rep['hyp']}   unexplained {len(rep['unexplained']):2d}   "
            f"features vs f64: rms {np.sqrt((e ** 2).mean()):.2e} max {e.max():.1e}   {ist}   | {px.summary(rep)}   [{time.time() - t0:.0f} s]")
    print(line, flush=True)
    report.append(line)
    _lib.check_status()
    del model
    torch.cuda.empty_cache()


def main(variants):
    golden = os.path.join(ROOT, "tests", "golden")
    os.environ.setdefault("GIGAPOSE_NUMERICS", "split")
    report = []
    print("# BASELINE config 2 (ViT-L/14, 1 x 162 templates, 64 crops) vs the unmodified reference in float64; the reference's own f32 run: 312 / 320")
    for v in variants:
        run_variant(v, golden, report)


if __name__ == "__main__":
    default = ["product_split", "product_chain", "hybrid", "ideal:ln", "ideal:qkv", "ideal:attn", "ideal:proj", "ideal:fc1", "ideal:fc2", "ideal:gemms",
               "ideal:ln+", "ideal:all", "matcher_chain", "ist_chain", "vit_chain"]
    main(sys.argv[1:] or default)
