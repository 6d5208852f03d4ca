"""CPU: oracle restatements of the IST regressor, RANSAC and pose recovery pinned against goldens
produced by the unmodified reference (oracle/make_goldens.py: gen_ist, gen_pose); plus state-dict
name parity of the gigapose_amd mirrors with the reference modules."""
import os

import numpy as np
import pytest
import torch

from gigapose_testing import synthetic as syn
from oracle import ist_torch
from oracle import cpu as oracle

IST_CFG = dict(n_heads=0, input_dim=3, input_size=256, initial_dim=128, block_dims=[128, 192, 256, 512],
               descriptor_size=256)


def build_ist(seed, conditioned=False):
    """conditioned: synthetic.condition_ist's rescaling (what the e2e goldens use, oracle/make_goldens.py: build_ref_ist)."""
    from gigapose_amd.ist_net import ISTNet, Regressor, ResNet

    net = ISTNet("resnet", ResNet(dict(IST_CFG)), Regressor(256, 256, True, True), 64).eval()
    syn.fill_state_dict(net, seed)
    return syn.condition_ist(net) if conditioned else net


def mlp_weights(net):
    out = {}
    for name, seq in (("scale", net.regressor.scale_predictor), ("inplane", net.regressor.inplane_predictor)):
        out[name] = [t.detach().numpy() for l in (seq[0], seq[2], seq[4]) for t in (l.weight, l.bias)]
    return out


def test_state_dict_names_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "ist.npz"))
    net = build_ist(101)
    assert "|".join(sorted(net.state_dict())) == str(g["state_names"])


def test_resnet_and_mlp_oracle_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "ist.npz"))
    net = build_ist(101)
    tmpl, _ = syn.template_images(102, 2)
    with torch.no_grad():
        feat = ist_torch.resnet_forward(net.backbone, torch.from_numpy(tmpl)).numpy()  # torch fp32 statement of the HIP path
    np.testing.assert_allclose(feat, g["resnet_feat"], rtol=1e-4, atol=1e-3)  # |feat| ~ 20
    rs = np.random.RandomState(103)
    src_feat = rs.standard_normal((3, 256, 16, 16)).astype(np.float32)
    tar_feat = rs.standard_normal((3, 256, 16, 16)).astype(np.float32)
    corr = syn.correspondences_case(104, 3, 1)
    sc, cs = oracle.ist_inference(tar_feat.reshape(3, 256, 256), src_feat.reshape(3, 1, 256, 256),
                                  corr["tar_pts"], corr["src_pts"], mlp_weights(net))
    assert ((sc[:, 0] == -1000) == (g["scales"] == -1000)).all()
    np.testing.assert_allclose(sc[:, 0], g["scales"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(cs[:, 0], g["cos_sin"], rtol=2e-5, atol=2e-6)


def pose_case():
    B, k, O, N = 4, 5, 2, 6
    corr = syn.correspondences_case(201, B, k)
    tK, tM, tP = syn.template_geometry(202, O, N)
    qK, qM = syn.crop_geometry(203, B)
    return corr, (tK, tM, tP), (qK, qM)


def test_ransac_oracle_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "pose.npz"))
    corr, _, _ = pose_case()
    M, failed, isrc, itar, isc = oracle.ransac(corr["src_pts"], corr["tar_pts"], corr["rel_scale"], corr["rel_inplane"])
    np.testing.assert_array_equal(failed, g["idx_failed"])
    np.testing.assert_array_equal(isc, g["ransac_scores"].astype(np.int64))
    np.testing.assert_array_equal(isrc, g["ransac_src_pts"].astype(np.int64))
    np.testing.assert_array_equal(itar, g["ransac_tar_pts"].astype(np.int64))
    np.testing.assert_allclose(M, g["M"], rtol=1e-5, atol=2e-4)  # translations are O(100) px
    # edge cases: N=0 -> identity, not failed; N=1 -> that candidate, failed
    assert (M[0, 0] == np.eye(3)).all() and not failed[0, 0] and failed[0, 1] and isc[0, 1].sum() == 0


def test_recovery_oracle_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "pose.npz"))
    _, (tK, tM, tP), (qK, qM) = pose_case()
    poses = oracle.recover(g["labels"] - 1, qK, qM, g["id_src"], g["M"], tK, tM, tP)
    ref = g["poses"]
    np.testing.assert_allclose(poses[..., :3, :3], ref[..., :3, :3], rtol=0, atol=2e-6)
    rel = np.linalg.norm(poses[..., :3, 3] - ref[..., :3, 3], axis=-1) / np.linalg.norm(ref[..., :3, 3], axis=-1)
    assert rel.max() < 1e-5, rel.max()
    np.testing.assert_array_equal(poses[..., 3, :], ref[..., 3, :])


def test_ransac_boundary_and_e2e_votes_reproduce_reference_bit_exactly(golden_dir):
    """Errors of exactly 14 px (two query patches on one template patch): the reference's inlier test
    is decided by the rounding of torch.bmm, whose path changes at n = 46 correspondences.  The
    restatement follows both paths, so counts, inlier lists AND M are bit-identical to the reference."""
    g = np.load(os.path.join(golden_dir, "pose_boundary.npz"))
    case = syn.many_to_one_case(211, 14)
    M, failed, isrc, itar, isc = oracle.ransac(case["src_pts"], case["tar_pts"], case["rel_scale"], case["rel_inplane"])
    np.testing.assert_array_equal(isc, g["scores"].astype(np.int64))
    np.testing.assert_array_equal(isrc, g["src_pts"].astype(np.int64))
    np.testing.assert_array_equal(itar, g["tar_pts"].astype(np.int64))
    np.testing.assert_array_equal(failed, g["idx_failed"])
    np.testing.assert_array_equal(M.view(np.uint32), g["M"].view(np.uint32))
    # the reference's own end-to-end run: feed its regressions back, get its votes and M back exactly
    e = np.load(os.path.join(golden_dir, "e2e.npz"))
    M, failed, isrc, itar, isc = oracle.ransac(e["src_pts"].astype(np.int64), e["tar_pts"].astype(np.int64),
                                               e["relScale"], e["relInplane"])
    np.testing.assert_array_equal(isc.sum(-1) / 256, e["all_scores"])
    np.testing.assert_array_equal(M.view(np.uint32), e["M"].view(np.uint32))


def _scored_cases():
    case, inv = syn.many_to_one_case(221, 14), syn.many_to_one_case(222, 14)
    return {"s2t_w": (case, True), "t2s": (inv, False), "t2s_w": (inv, True)}


def test_ransac_scores_and_tar2src_reproduce_the_reference(golden_dir):
    """RANSAC.forward's optional arguments (ransac.py:108-121) -- per-correspondence `scores` and direction="tar2src" (the `*_inv`
    fields) -- on the exact-14-px boundary case, against the unmodified reference (oracle/make_goldens.py: gen_pose_scored):
    weighted winners, failed flags (a problem whose weights are all zero fails), inlier lists, the weights cast
    to int64, and M bit for bit."""
    g = np.load(os.path.join(golden_dir, "pose_scored.npz"))
    for tag, (case, weighted) in _scored_cases().items():
        M, failed, isrc, itar, isc = oracle.ransac(case["src_pts"], case["tar_pts"], case["rel_scale"], case["rel_inplane"],
                                                   scores=g["weights"] if weighted else None)
        np.testing.assert_array_equal(isc, g[tag + "_scores"].astype(np.int64), err_msg=tag)
        np.testing.assert_array_equal(isrc, g[tag + "_src_pts"].astype(np.int64), err_msg=tag)
        np.testing.assert_array_equal(itar, g[tag + "_tar_pts"].astype(np.int64), err_msg=tag)
        np.testing.assert_array_equal(failed, g[tag + "_failed"], err_msg=tag)
        np.testing.assert_array_equal(M.view(np.uint32), g[tag + "_M"].view(np.uint32), err_msg=tag)
    assert g["s2t_w_failed"][3], "the all-zero-weights problem must fail (best weighted score 0), whatever its inliers"
