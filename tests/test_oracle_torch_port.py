"""CPU: oracle/torch_port.py (the torch-CPU restatement of the reference's cost profile that bench.py times as
`cpu_baseline`) against the goldens written by the unmodified reference -- a baseline that computes something else
would be a strawman."""
import os

import numpy as np
import pytest
import torch

from oracle import ist_torch, torch_port
from test_oracle_matcher import CASES, load_case


@pytest.mark.parametrize("name", CASES)
def test_torch_matcher_port_vs_reference_golden(golden_dir, name):
    g, case, k = load_case(golden_dir, name)
    lab = torch.from_numpy(case["labels"]).long()
    out = torch_port.local_similarity_test(torch.from_numpy(case["src_feats"])[lab], torch.from_numpy(case["tar_feat"]),
                                           torch.from_numpy(case["src_masks"])[lab], torch.from_numpy(case["tar_mask"]), k, max_batch_size=2)
    np.testing.assert_array_equal(out["id_src"].numpy(), g["id_src"])
    np.testing.assert_array_equal(out["tar_pts"].numpy(), g["tar_pts"].astype(np.int64))
    np.testing.assert_array_equal(out["src_pts"].numpy(), g["src_pts"].astype(np.int64))
    np.testing.assert_allclose(out["score_src"].numpy(), g["score_src"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["score_pts"].numpy(), g["score_pts"], rtol=0, atol=2e-6)


def test_torch_eval_retrieval_port_vs_reference_golden(golden_dir):
    """The whole port on the config-1-shaped end-to-end golden (ViT-S/14 stand-in): ids, correspondences, poses."""
    from test_gpu_e2e import E2E, e2e_inputs
    from test_oracle_pose_ist import build_ist
    from transformers import Dinov2Config, Dinov2Model

    from gigapose_testing import synthetic as syn

    torch.set_num_threads(8)
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    dim, depth, heads = E2E["vit"]
    hf = Dinov2Model(Dinov2Config(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads, image_size=224, patch_size=14)).eval()
    syn.fill_state_dict(hf, 302)
    ist = build_ist(303, conditioned=True)
    items, q = e2e_inputs(E2E["seed"], E2E["O"], E2E["N"], E2E["B"])
    with torch.no_grad():
        bank_ae = torch.stack([torch_port.vit_features(hf, it.rgb) for it in items])
        bank_ist = torch.stack([ist_torch.resnet_forward(ist.backbone, it.rgb) for it in items])
    geom = tuple(np.stack([getattr(it, n).numpy() for it in items]) for n in ["K", "M", "poses"])
    crops = {n: torch.from_numpy(q[n]) for n in ["tar_img", "tar_mask", "tar_K", "tar_M", "labels"]}
    poses, pred = torch_port.eval_retrieval(hf, ist, bank_ae, bank_ist, torch.stack([it.mask for it in items]), geom, crops, E2E["k"])
    # the matcher outputs are unsorted; the golden is sorted by inlier score (stable): compare as sets per detection + poses
    assert (np.sort(pred["id_src"].numpy(), 1) == np.sort(g["id_src"].astype(np.int64), 1)).all()
    t = np.linalg.norm(poses[..., :3, 3] - g["all_poses"][..., :3, 3], axis=-1) / np.linalg.norm(g["all_poses"][..., :3, 3], axis=-1)
    assert t.max() < 1e-4 and np.abs(poses[..., :3, :3] - g["all_poses"][..., :3, :3]).max() < 1e-4
