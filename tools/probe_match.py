"""GPU probe: time the fused matcher at BASELINE config-2 shape and print achieved TFLOP/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigapose_amd.matching import LocalSimilarity, MatchBank
from gigapose_testing import synthetic as syn

dev = "cuda"
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count, "CUs")
B, N, C = 64, 162, 1024
bank_np, q_np = syn.random_features(1, B, 1, N, C)
metric = LocalSimilarity(5, 0.5, 3)
bank = MatchBank(torch.from_numpy(bank_np).view(1, N, C, 16, 16).to(dev), torch.ones(1, N, 224, 224, device=dev))
q = metric.normalize(torch.from_numpy(q_np).to(dev))
qm = torch.ones(B, 256, device=dev)
lab = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(3):
    metric.match_tiles(q, qm, bank, lab)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 10
e0.record()
for _ in range(iters):
    metric.match_tiles(q, qm, bank, lab)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
flops = 2.0 * B * N * 256 * 256 * C
print(f"gp_match_tiles B={B} N={N} C={C}: {ms:.3f} ms  -> {flops / ms / 1e9:.1f} TFLOP/s f32 (peak 157.3) ; {B / ms * 1e3:.0f} crops/s matcher-only")
