"""Where does the split path's feature error come from?  

ViT-L features on the plane path are rms 3.1e-8 from the float64 forward where the reference's own f32 forward is 2.2e-8, although
the split GEMM alone is CLOSER to float64 than a sequential f32 chain.  This tool runs ONE transformer layer of the plane path
stage by stage through the stage entry points of the C-ABI (gp_layernorm_planes, gp_gemm_planes256_ragged, gp_attention_split) on
real activations (the residual stream after `--layer` blocks of a B-crop forward) and compares every stage with
  * float64 arithmetic on the SAME inputs (the values our planes hold, promoted) and the f32 weights the reference holds -> our
    LOCAL error of the stage, rounding of its output to 22-bit planes included;
  * the same stage in plain torch f32 on the GPU (rocBLAS / ATen), also against float64 -> what "a second f32 implementation" does.
Errors are rms over the token rows, relative to the rms of the stage's float64 output.  Usage (GPU box):
    python tools/probe_stage_errors.py [--batch 64] [--layer 12]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapose_amd import _lib  # noqa: E402
from gigapose_testing import factory
from gigapose_amd.vit import split_planes_x64  # noqa: E402

DEV = "cuda"
F64 = torch.float64


def planes_empty(rows, cols):
    return torch.zeros(rows, cols, dtype=torch.float16, device=DEV), torch.zeros(rows, cols, dtype=torch.float16, device=DEV)


def val(hi, lo, scale=8.0):
    return (hi.double() + lo.double()) / scale


def rel(a, ref):
    return float(((a - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--layer", type=int, default=12)
    args = ap.parse_args()
    torch.manual_seed(0)
    model = factory.build_model("dinov2_vitl14", k=5, device=DEV, seed=0)
    model.set_numerics("split")
    vit = model.ae_net.dinov2_model
    tset = factory.TemplateSet(1, 16, seed=100)
    q = tset.crops(1000, args.batch, DEV)
    B, C, L = args.batch, vit.dim, args.layer
    Mtok, Mpad = B * 257, (B * 257 + 255) // 256 * 256
    vit.patch_features(q["tar_img"], normalize=False, stop_after_layers=L)
    torch.cuda.synchronize()
    X = vit._ws[: C * Mpad].view(C, Mpad).clone()                 # residual stream after L blocks, channel-major [C][Mpad]
    blk = vit.blocks[L]
    lib = _lib.lib()
    lib.gp_gemm_split256_workspace_bytes.restype = ctypes.c_size_t
    nb = lib.gp_gemm_split256_workspace_bytes()
    ws = torch.zeros(nb // 4, device=DEV)
    st = _lib.stream_ptr
    os_ = 1.0 / (8.0 * 64.0)
    rows = []

    def gemm_planes(W, bhi, blo, epi, bias, K, I, out_planes, scale=None, res=None):
        whi, wlo = split_planes_x64(W)
        D = res.clone() if res is not None else torch.zeros(1, device=DEV)
        ohi, olo = planes_empty(Mpad, I) if out_planes else (None, None)
        _lib.call("gp_gemm_planes256_scaled", _lib.ptr(whi), _lib.ptr(wlo), _lib.ptr(bhi), _lib.ptr(blo), _lib.ptr(D if not out_planes else None),
                  _lib.i(Mpad), _lib.ptr(ohi), _lib.ptr(olo), _lib.i(I), _lib.i(I), _lib.i(Mpad), _lib.i(Mtok), _lib.i(K), _lib.i(epi),
                  _lib.ptr(bias), _lib.ptr(scale), _lib.ptr(D if res is not None else None), _lib.i(Mpad if res is not None else 0), _lib.f(os_),
                  _lib.f(8.0), _lib.ptr(None), _lib.ptr(ws), ctypes.c_size_t(nb), st())
        torch.cuda.synchronize()
        return (ohi, olo) if out_planes else D

    def ln(Xcm, g, b):
        hi, lo = planes_empty(Mpad, C)
        _lib.call("gp_layernorm_planes", _lib.ptr(Xcm), _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(g), _lib.ptr(b), _lib.i(C), _lib.i(Mpad), _lib.f(1e-6), st())
        torch.cuda.synchronize()
        return hi, lo

    def report(name, ours, ref64, alt32):
        e_o, e_a = rel(ours[:Mtok], ref64[:Mtok]), rel(alt32[:Mtok].double(), ref64[:Mtok])
        rows.append((name, float((ref64[:Mtok] ** 2).mean().sqrt()), e_o, e_a))

    f32 = lambda t: t.detach().float().to(DEV)
    f64 = lambda t: t.detach().double().to(DEV)

    # ---- LayerNorm 1
    x_tok64 = X.t().double()                                      # [Mpad][C]
    h1hi, h1lo = ln(X, f32(blk.norm1.weight), f32(blk.norm1.bias))
    ref = torch.nn.functional.layer_norm(x_tok64, (C,), f64(blk.norm1.weight), f64(blk.norm1.bias), 1e-6)
    alt = torch.nn.functional.layer_norm(X.t().contiguous(), (C,), f32(blk.norm1.weight), f32(blk.norm1.bias), 1e-6)
    report("LayerNorm 1 -> planes", val(h1hi, h1lo), ref, alt)
    # ---- q | k | v
    h1 = val(h1hi, h1lo)
    Wqkv, bqkv = blk.attn.qkv.weight, blk.attn.qkv.bias
    ahi, alo = gemm_planes(f32(Wqkv), h1hi, h1lo, 7, f32(bqkv), C, 3 * C, True)
    ref = h1 @ f64(Wqkv).t() + f64(bqkv)
    alt = torch.nn.functional.linear(h1.float(), f32(Wqkv), f32(bqkv))
    report("q|k|v GEMM -> planes", val(ahi, alo), ref, alt)
    # ---- attention
    qkv = val(ahi, alo)[:Mtok].view(B, 257, 3, vit.heads, 64)
    ohi, olo = planes_empty(Mpad, C)
    _lib.call("gp_attention_split_scaled", _lib.ptr(ahi), _lib.ptr(alo), _lib.ptr(ohi), _lib.ptr(olo), _lib.i(B), _lib.i(vit.heads), _lib.i(C), _lib.i(Mpad), _lib.f(8.0), st())
    torch.cuda.synchronize()

    def attn(x):
        qq, kk, vv = [x[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
        return (torch.softmax(qq @ kk.transpose(-1, -2) * 0.125, dim=-1) @ vv).permute(0, 2, 1, 3).reshape(Mtok, C)

    ref, alt = attn(qkv), attn(qkv.float())
    report("attention -> planes", val(ohi, olo)[:Mtok], ref, alt)
    # ---- proj + LayerScale + residual
    a = val(ohi, olo)
    Wp, bp, ls1 = blk.attn.proj.weight, blk.attn.proj.bias, blk.ls1.gamma
    X1 = gemm_planes(f32(Wp), ohi, olo, 3, f32(bp), C, C, False, scale=f32(ls1), res=X)
    ref = x_tok64 + f64(ls1) * (a @ f64(Wp).t() + f64(bp))
    alt = X.t() + f32(ls1) * torch.nn.functional.linear(a.float(), f32(Wp), f32(bp))
    report("proj + residual (x1)", X1.t().double(), ref, alt)
    report("  proj branch alone", (X1 - X).t().double(), ref - x_tok64, alt - X.t())
    # ---- LayerNorm 2
    h2hi, h2lo = ln(X1, f32(blk.norm2.weight), f32(blk.norm2.bias))
    x1_64 = X1.t().double()
    ref = torch.nn.functional.layer_norm(x1_64, (C,), f64(blk.norm2.weight), f64(blk.norm2.bias), 1e-6)
    alt = torch.nn.functional.layer_norm(X1.t().contiguous(), (C,), f32(blk.norm2.weight), f32(blk.norm2.bias), 1e-6)
    report("LayerNorm 2 -> planes", val(h2hi, h2lo), ref, alt)
    # ---- fc1 + GELU
    h2 = val(h2hi, h2lo)
    W1, b1 = blk.mlp.fc1.weight, blk.mlp.fc1.bias
    fhi, flo = gemm_planes(f32(W1), h2hi, h2lo, 6, f32(b1), C, vit.mlp_dim, True)
    ref = torch.nn.functional.gelu(h2 @ f64(W1).t() + f64(b1))
    alt = torch.nn.functional.gelu(torch.nn.functional.linear(h2.float(), f32(W1), f32(b1)))
    report("fc1 + GELU -> planes", val(fhi, flo), ref, alt)
    # ---- fc2 + LayerScale + residual
    f = val(fhi, flo)
    W2, b2, ls2 = blk.mlp.fc2.weight, blk.mlp.fc2.bias, blk.ls2.gamma
    X2 = gemm_planes(f32(W2), fhi, flo, 3, f32(b2), vit.mlp_dim, C, False, scale=f32(ls2), res=X1)
    ref = x1_64 + f64(ls2) * (f @ f64(W2).t() + f64(b2))
    alt = X1.t() + f32(ls2) * torch.nn.functional.linear(f.float(), f32(W2), f32(b2))
    report("fc2 + residual (x2)", X2.t().double(), ref, alt)
    report("  fc2 branch alone", (X2 - X1).t().double(), ref - x1_64, alt - X1.t())
    _lib.check_status()

    print(f"ViT-L layer {L}, {B} crops ({Mtok} tokens): per-stage error vs float64 on the same inputs, rms relative to the stage output's rms")
    print(f"{'stage':28s} {'rms(out)':>10s} {'split path':>12s} {'torch f32':>12s} {'ratio':>7s}")
    for name, r, e_o, e_a in rows:
        print(f"{name:28s} {r:10.3e} {e_o:12.2e} {e_a:12.2e} {e_o / max(e_a, 1e-30):7.2f}")
    print("(planes hold 22 significand bits: rounding a stage output to planes alone costs ~2^-23 / sqrt(3) = 6.9e-8 relative rms per element "
          "against f32's 2^-25 / sqrt(3) = 1.7e-8)")


if __name__ == "__main__":
    main()
