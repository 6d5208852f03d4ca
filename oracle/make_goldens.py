"""Generate tests/golden/*.npz by running the UNMODIFIED reference (/root/reference) on CPU
through oracle/ref_shim.py.  Runs only in the build container (the reference cannot travel);
the fixtures it writes are committed.  Usage: python oracle/make_goldens.py [stage ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from gigapose_amd import synthetic as syn  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# (name, kwargs for synthetic.matcher_case, k)
MATCHER_CASES = [
    ("match_small", dict(seed=11, B=3, O=2, N=6, C=32), 5),
    ("match_vits", dict(seed=12, B=2, O=1, N=8, C=384), 5),
    ("match_kN", dict(seed=13, B=2, O=1, N=4, C=64), 4),           # k == N (config-1 shape)
    ("match_fullmask", dict(seed=14, B=2, O=2, N=7, C=48, full_masks=True), 5),
    ("match_noshift", dict(seed=15, B=4, O=3, N=9, C=96, shift=False, noise=0.2), 5),
]


def gen_matcher():
    ref_shim.install()
    from src.models.matching import LocalSimilarity

    for name, kw, k in MATCHER_CASES:
        case = syn.matcher_case(**kw)
        metric = LocalSimilarity(k=k, sim_threshold=0.5, patch_threshold=3)
        labels = torch.from_numpy(case["labels"]).long()
        src_feats = torch.from_numpy(case["src_feats"])[labels]          # gigaPose.py:520
        src_masks = torch.from_numpy(case["src_masks"])[labels]          # gigaPose.py:521
        out = metric.test(src_feats=src_feats, tar_feat=torch.from_numpy(case["tar_feat"]),
                          src_masks=src_masks, tar_mask=torch.from_numpy(case["tar_mask"]))
        np.savez_compressed(
            os.path.join(GOLD, name + ".npz"),
            input_checksum=syn.checksum(*[case[x] for x in sorted(case)]),
            case_kwargs=repr(kw), k=k,
            id_src=out.id_src.numpy(), score_src=out.score_src.numpy(),
            score_pts=out.score_pts.numpy(), tar_pts=out.tar_pts.numpy().astype(np.int16),
            src_pts=out.src_pts.numpy().astype(np.int16))
        print(name, "id_src[0] =", out.id_src[0].tolist(), "score_src[0] =",
              np.round(out.score_src[0].numpy(), 4).tolist(),
              "valid pts:", int((out.tar_pts[..., 0] >= 0).sum()))


STAGES = {"matcher": gen_matcher}

if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    todo = sys.argv[1:] or list(STAGES)
    for s in todo:
        STAGES[s]()
