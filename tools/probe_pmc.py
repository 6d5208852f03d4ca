"""Tiny workload for PMC passes: each hot kernel family a few times (ViT-L shapes at B=64)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigapose_amd import _lib
from gigapose_amd.matching import LocalSimilarity, MatchBank
from gigapose_testing import synthetic as syn
dev = "cuda"
J = 16512
for (I, K, epi) in [(1024, 1024, 3), (2048, 1024, 1), (4096, 1024, 2), (1024, 4096, 3)]:
    A = torch.randn(K, I, device=dev); Bm = torch.randn(K, J, device=dev); D = torch.randn(I, J, device=dev)
    bias = torch.randn(I, device=dev); sc = torch.randn(I, device=dev)
    for _ in range(3):
        _lib.call("gp_gemm_kmajor", _lib.ptr(A), _lib.i(I), _lib.ptr(Bm), _lib.i(J), _lib.ptr(D), _lib.i(J), _lib.i(I), _lib.i(J),
                  _lib.i(K), _lib.i(epi), _lib.ptr(bias), _lib.ptr(sc), _lib.ptr(D), _lib.i(J), _lib.stream_ptr())
B, N, C = 64, 162, 1024
bank_np, q_np = syn.random_features(1, B, 1, N, C)
metric = LocalSimilarity(5, 0.5, 3)
bank = MatchBank(torch.from_numpy(bank_np).view(1, N, C, 16, 16).to(dev), torch.ones(1, N, 224, 224, device=dev))
q = metric.normalize(torch.from_numpy(q_np).to(dev))
for _ in range(2):
    metric.match_tiles(q, torch.ones(B, 256, device=dev), bank, torch.zeros(B, dtype=torch.int32, device=dev))
torch.cuda.synchronize()
print("done")
