"""GPU probe: where a split-matcher tile spends its time.

  1. launch time at C = 32 (one k-step) .. C = 1024: the fixed per-tile cost (prologue + epilogue + dispatch);
  2. the probe build's per-tile stamps (gp_match_tiles_split_trace, 100 MHz wall clock): entry -> first slab staged ->
     k loop done -> maxima -> merge -> per-patch outputs -> end, and per CU the gap between one tile's end and the next
     tile's entry (workgroup dispatch: one 129 KB-LDS workgroup per CU, no overlap between tiles).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gigapose_amd import _lib

_lib.use_probe_library()   # hooks / traced builds / error words live in libgigapose_hip_probe.so (include/gigapose_hip_probe.h)
from gigapose_amd.matching import LocalSimilarity, MatchBank, patch_grid_mask

dev = "cuda"
B, N = 64, 162


def setup(C):
    torch.manual_seed(0)
    bank_f = torch.nn.functional.normalize(torch.randn(1, N, C, 16, 16, device=dev), dim=2)
    q_f = torch.nn.functional.normalize(torch.randn(B, C, 16, 16, device=dev), dim=1)
    m = LocalSimilarity(5, 0.5, 3)
    m.numerics = "split"
    bank = MatchBank(bank_f, torch.ones(1, N, 224, 224, device=dev), "split")
    return m, m.normalize(q_f), patch_grid_mask(torch.ones(B, 224, 224, device=dev)), bank, torch.zeros(B, dtype=torch.int32, device=dev)


for C in (32, 64, 256, 1024):
    m, q, qm, bank, lab = setup(C)
    for _ in range(2):
        m.match_tiles(q, qm, bank, lab)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        m.match_tiles(q, qm, bank, lab)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    print(f"C={C:5d}: {t*1e3:8.1f} us per launch = {t*1e3*256/(B*N):6.1f} us per tile per CU ({C//32} k-steps)")

def setup_disc(C):
    """The benchmark's kind of masks (disc-shaped: ~5 x 5 live 32 x 32 blocks per tile) with planted, graded matches."""
    from gigapose_testing import synthetic as syn

    case = syn.matcher_case(seed=5, B=B, O=1, N=N, C=C)
    m = LocalSimilarity(5, 0.5, 3)
    m.numerics = "split"
    bank = MatchBank(torch.from_numpy(case["src_feats"]).to(dev), torch.from_numpy(case["src_masks"]).to(dev), "split")
    q = m.normalize(torch.from_numpy(case["tar_feat"]).to(dev).view(B, C, 256))
    return m, q, patch_grid_mask(torch.from_numpy(case["tar_mask"]).to(dev)), bank, torch.from_numpy(case["labels"]).to(dev).int() - 0


for C, make in ((32, setup), (1024, setup), (1024, setup_disc)):
    m, q, qm, bank, lab = make(C)
    if make is setup_disc:
        lab = torch.zeros(B, dtype=torch.int32, device=dev)
        print("\n--- disc masks (the benchmark's kind): live-patch compaction at work")
    idx = torch.empty(B, N, 256, dtype=torch.uint8, device=dev)
    sc = torch.empty(B, N, 256, device=dev)
    ma = torch.empty(B, N, 256, device=dev)
    avg = torch.empty(B, N, device=dev)
    trace = torch.zeros(B * N, 8, dtype=torch.int64, device=dev)
    for _ in range(2):
        _lib.call("gp_match_tiles_split_trace", _lib.ptr(q[0]), _lib.ptr(q[1]), _lib.ptr(bank.hi), _lib.ptr(bank.lo), _lib.ptr(qm),
                  _lib.ptr(bank.masks), _lib.ptr(lab), _lib.i(B), _lib.i(1), _lib.i(N), _lib.i(C), _lib.f(0.5), _lib.f(3.0),
                  _lib.ptr(idx), _lib.ptr(sc), _lib.ptr(ma), _lib.ptr(avg), _lib.ptr(trace), _lib.stream_ptr())
    torch.cuda.synchronize()
    tr = trace.cpu().numpy()
    st = tr[:, :7].astype(np.float64) / 100.0                       # us
    names = ["entry->slab0 staged", "k loop", "mask/thr + maxima", "merge", "per-patch outputs", "ordered sum"]
    d = np.diff(st, axis=1)
    print(f"\nC={C}: per-tile stage times (us), median / mean over {B*N} tiles; launch span {st[:, 6].max() - st[:, 0].min():.1f} us")
    for i, nm in enumerate(names):
        print(f"  {nm:22s} {np.median(d[:, i]):7.2f} {d[:, i].mean():7.2f}")
    print(f"  {'tile total':22s} {np.median(st[:, 6] - st[:, 0]):7.2f} {(st[:, 6] - st[:, 0]).mean():7.2f}")
    cu = tr[:, 7]                                                   # XCC_ID << 32 | HW_ID: same value = same CU (cu, sh, se bits)
    key = ((cu >> 32) << 8) | ((cu >> 8) & 0xff)                    # xcc | se(3) sh(1) cu(4)
    gaps, busy = [], []
    for k in np.unique(key):
        rows = st[key == k]
        rows = rows[np.argsort(rows[:, 0])]
        gaps.extend(rows[1:, 0] - rows[:-1, 6])
        busy.append((rows[:, 6] - rows[:, 0]).sum() / (rows[-1, 6] - rows[0, 0]))
    gaps = np.array(gaps)
    print(f"  {len(np.unique(key))} CUs; end -> next entry on the same CU: median {np.median(gaps):.2f} us, mean {gaps.mean():.2f} us; "
          f"in-tile fraction of a CU's span {np.mean(busy):.3f}")
