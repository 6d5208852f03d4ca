"""Tiny workload for PMC passes over the split-f16 kernels (fc1 shape GEMM, a layer-1 conv, the matcher)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigapose_amd import _lib
from gigapose_amd.vit import split_planes
from gigapose_amd.matching import LocalSimilarity, MatchBank
dev = "cuda"
I, J, K = 4096, 16512, 1024
W = torch.randn(I, K, device=dev) * 0.05; X = torch.randn(K, J, device=dev)
hi, lo = split_planes(W)
D = torch.empty(I, J, device=dev)
for _ in range(3):
    _lib.call("gp_gemm_split", _lib.ptr(X), _lib.i(J), _lib.ptr(hi), _lib.ptr(lo), _lib.ptr(D), _lib.i(J), _lib.i(I), _lib.i(J),
              _lib.i(K), _lib.i(1), _lib.i(0), None, None, None, _lib.i(J), _lib.stream_ptr())
B, H, C = 32, 128, 128
xh, xl = split_planes(torch.randn(B, H, H, C, device=dev))
wh, wl = split_planes(torch.randn(128, 9 * C, device=dev) * 0.03)
oh, ol = torch.empty(B, H, H, C, dtype=torch.float16, device=dev), torch.empty(B, H, H, C, dtype=torch.float16, device=dev)
for _ in range(3):
    _lib.call("gp_conv2d_nhwc_split", _lib.ptr(xh), _lib.ptr(xl), _lib.ptr(wh), _lib.ptr(wl), None, None, None, None, _lib.i(B),
              _lib.i(H), _lib.i(H), _lib.i(C), _lib.i(C), _lib.i(3), _lib.i(3), _lib.i(1), _lib.i(1), _lib.i(1), _lib.ptr(oh), _lib.ptr(ol),
              None, _lib.stream_ptr())
Bq, N, Cf = 32, 64, 1024
m = LocalSimilarity(5, 0.5, 3); m.numerics = "split"
bank = MatchBank(torch.randn(1, N, Cf, 16, 16, device=dev), torch.ones(1, N, 224, 224, device=dev), "split")
q = m.normalize(torch.randn(Bq, Cf, 256, device=dev))
for _ in range(3):
    m.match_tiles(q, torch.ones(Bq, 256, device=dev), bank, torch.zeros(Bq, dtype=torch.int32, device=dev))
torch.cuda.synchronize()
print("done")
