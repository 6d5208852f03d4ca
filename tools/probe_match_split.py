"""GPU probe: template matcher, chain vs split numerics at BASELINE config-2 size: time and agreement."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gigapose_amd.matching import LocalSimilarity, MatchBank
from gigapose_testing import synthetic as syn
dev = "cuda"
def timeit(fn, iters=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B, N, C = 64, 162, 1024
from gigapose_amd.matching import patch_grid_mask
case = syn.matcher_case(seed=5, B=B, O=1, N=N, C=C)   # planted, graded matches + disc masks
feats = torch.from_numpy(case["src_feats"]).to(dev)
qf = torch.from_numpy(case["tar_feat"]).to(dev).view(B, C, 256)
masks = torch.from_numpy(case["src_masks"]).to(dev)
qmask = patch_grid_mask(torch.from_numpy(case["tar_mask"]).to(dev))
labels = torch.from_numpy(case["labels"]).to(dev)
res = {}
for mode in ["chain", "split"]:
    m = LocalSimilarity(5, 0.5, 3); m.numerics = mode
    bank = MatchBank(feats, masks, mode)
    q = m.normalize(qf)
    ms = timeit(lambda: m.match_tiles(q, qmask, bank, labels))
    res[mode] = m.match_tiles(q, qmask, bank, labels)
    print(f"{mode}: match_tiles {ms:.3f} ms -> {2.0*B*N*256*256*C/ms/1e9:.1f} TF-equivalent; normalize(query) {timeit(lambda: m.normalize(qf)):.3f} ms")
(i0, s0, m0, a0), (i1, s1, m1, a1) = res["chain"], res["split"]
print(f"idx_t2s equal: {(i0 == i1).float().mean().item()*100:.5f} %  ({(i0 != i1).sum().item()} of {i0.numel()} differ)")
print(f"mask_all equal: {(m0 == m1).float().mean().item()*100:.5f} %  score_t2s max |diff| {(s0 - s1).abs().max().item():.3e}  sim_avg max |diff| {(a0 - a1).abs().max().item():.3e}")
t0 = torch.topk(a0, 5, dim=1).indices; t1 = torch.topk(a1, 5, dim=1).indices
print("top-5 template ids equal:", torch.equal(t0, t1))
print("nonzero scores:", (s0 != 0).sum().item(), "of", s0.numel(), " max score", s0.max().item(), " mask_all sum", m0.sum().item(), " sim_avg max", a0.max().item())
nz = s0 != 0
d = (s0 - s1).abs()[nz | (s1 != 0)]
print("score diff over nonzero entries: max", d.max().item() if d.numel() else None, " count exact-equal", (d == 0).sum().item(), "of", d.numel())
