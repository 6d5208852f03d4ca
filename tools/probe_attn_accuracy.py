"""GPU probe: where attention_split_kernel errs (per query tile / channel group) on uniform, v = 1 and random inputs."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigapose_amd import _lib
dev = "cuda"
torch.manual_seed(11)
B, H = 3, 6
C = 64 * H; M = B * 257; Mpad = (M + 255) // 256 * 256
def run(qkv):
    hi = torch.empty(Mpad, 3 * C, dtype=torch.float16, device=dev); lo = torch.empty_like(hi)
    _lib.call("gp_split_planes", _lib.ptr(qkv), ctypes.c_size_t(qkv.numel()), _lib.f(8.0), _lib.ptr(hi), _lib.ptr(lo), _lib.stream_ptr())
    ohi = torch.zeros(Mpad, C, dtype=torch.float16, device=dev); olo = torch.zeros_like(ohi)
    _lib.call("gp_attention_split_scaled", _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(ohi), _lib.ptr(olo), _lib.i(B), _lib.i(H), _lib.i(C), _lib.i(Mpad), _lib.f(8.0), _lib.stream_ptr())
    torch.cuda.synchronize()
    got = ((ohi.double() + olo.double()) / 8.0)[:M].view(B, 257, H, 64)
    x = ((hi.double() + lo.double()) / 8.0)[:M].view(B, 257, 3, H, 64)
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v).permute(0, 2, 1, 3)
    return got, ref
base = torch.zeros(Mpad, 3 * C, device=dev)
rnd = torch.randn(M, 3 * C, device=dev)
for name, qs, ks, vmode in [("uniform attention (q=k=0), random v", 0.0, 0.0, "rand"), ("random q,k (x1.5), v = 1", 1.5, 1.5, "one"),
                            ("random q,k (x0.3), random v", 0.3, 0.3, "rand"), ("random q,k (x1.5), random v", 1.5, 1.5, "rand")]:
    qkv = base.clone()
    qkv[:M, :C] = rnd[:, :C] * qs; qkv[:M, C:2 * C] = rnd[:, C:2 * C] * ks
    qkv[:M, 2 * C:] = rnd[:, 2 * C:] if vmode == "rand" else 1.0
    got, ref = run(qkv)
    err = (got - ref).abs()
    e = err.max().item() / ref.abs().max().item()
    idx = torch.nonzero(err == err.max())[0].tolist()
    per_tile = [err[:, 32 * t:32 * t + 32].max().item() for t in range(9)]
    per_d = [err[..., 8 * g:8 * g + 8].max().item() for g in range(8)]
    frac = (err > 1e-5 * ref.abs().max()).double().mean().item()
    print(f"{name}: max err/max|ref| {e:.2e} at (b,t,h,d)={idx}; frac(err>1e-5) {frac:.3f}\n   per query tile {['%.1e' % v for v in per_tile]}\n   per d-group {['%.1e' % v for v in per_d]}")
