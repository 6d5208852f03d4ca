"""GPU parity at the BENCHMARK sizes: BASELINE config 2 (ViT-L width, 1 object x 162 templates,
64 crops) and config 3 (LM-O shape: 8 objects x 162 templates, mixed labels), both numerics modes, against

  * goldens written by the UNMODIFIED reference (oracle/make_goldens.py: gen_matcher_big -> LocalSimilarity.test,
    matching.py:188-316; gen_e2e("e2e_cfg2" / "e2e_cfg3") -> GigaPose.eval_retrieval, gigaPose.py:481-633), and
  * the CPU oracle over ALL 64 x 162 = 10 368 (detection, template) tiles of config 2.

Bars: template ids and patch correspondences bit-exact (no tie allowance: the assertion is plain equality); poses
within the north-star's 1e-4 relative.  Every test prints its disagreement counts so the log states them per mode.
"""
import ast
import os
import tempfile

import numpy as np
import pandas as pd
import pytest
import torch

from gigapose_amd import synthetic as syn
from oracle import cpu as oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"
_cases = {}
_oracle_tiles = {}


def big_case(golden_dir, name):
    if name not in _cases:
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        case = syn.matcher_case(**ast.literal_eval(str(g["case_kwargs"])))
        assert syn.checksum(*[case[x] for x in sorted(case)]) == str(g["input_checksum"]), "synthetic inputs drifted"
        _cases.clear()  # keep one (1.4 GB for config 3) alive at a time
        _cases[name] = (g, case)
    return _cases[name]


def run_bank(case, k, numerics):
    from gigapose_amd.matching import LocalSimilarity, MatchBank, patch_grid_mask

    metric = LocalSimilarity(k=k, sim_threshold=0.5, patch_threshold=3)
    metric.numerics = numerics
    t = lambda a: torch.from_numpy(a).to(DEV)
    bank = MatchBank(t(case["src_feats"]), t(case["src_masks"]), numerics)
    labels0 = t(case["labels"])
    out = metric.test_bank(bank, t(case["tar_feat"]), t(case["tar_mask"]), labels0)
    tiles = metric.match_tiles(metric.normalize(t(case["tar_feat"])), patch_grid_mask(t(case["tar_mask"])), bank, labels0)
    torch.cuda.synchronize()
    return {n: v.cpu().numpy() for n, v in out.tensors.items()}, [x.cpu().numpy() for x in tiles]


@pytest.mark.parametrize("numerics", ["chain", "split"])
@pytest.mark.parametrize("name", ["match_cfg2", "match_cfg3"])
def test_matcher_at_benchmark_size_vs_reference_golden(golden_dir, name, numerics):
    g, case = big_case(golden_dir, name)
    hip, _ = run_bank(case, int(g["k"]), numerics)
    d = {n: int((hip[n] != g[n].astype(np.int64)).sum()) for n in ["id_src", "src_pts", "tar_pts"]}
    es, ep = np.abs(hip["score_src"] - g["score_src"]).max(), np.abs(hip["score_pts"] - g["score_pts"]).max()
    print(f"{name} [{numerics}] vs reference: id_src differ {d['id_src']}/{g['id_src'].size}, src_pts {d['src_pts']}/"
          f"{g['src_pts'].size}, tar_pts {d['tar_pts']}/{g['tar_pts'].size}; score_src max err {es:.2e}, score_pts {ep:.2e}")
    assert d == dict(id_src=0, src_pts=0, tar_pts=0)
    assert es <= 1e-6 and ep <= 3e-6


@pytest.mark.parametrize("numerics", ["chain", "split"])
def test_matcher_config2_all_tiles_vs_oracle(golden_dir, numerics):
    """Every one of the 10 368 tiles: idx_tar2src, mask_all, score_tar2src, sim_avg.  chain: bit-exact incl. float bits
    (same fmaf chain).  split: a different f32-class arithmetic -- indices and masks must still be EQUAL; floats 4e-6."""
    g, case = big_case(golden_dir, "match_cfg2")
    _, (idx, sc, ma, avg) = run_bank(case, int(g["k"]), numerics)
    B, C = case["tar_feat"].shape[:2]
    O, N = case["src_feats"].shape[:2]
    if "cfg2" not in _oracle_tiles:   # ~1 min of host cores: once for both numerics
        qn = oracle.l2norm_cp(case["tar_feat"].reshape(B, C, 256))
        bn = oracle.l2norm_cp(case["src_feats"].reshape(O, N, C, 256))
        _oracle_tiles["cfg2"] = oracle.match(qn, bn, oracle.patch_mask(case["tar_mask"]), oracle.patch_mask(case["src_masks"]), case["labels"])
    oi, osc, oma, oavg = _oracle_tiles["cfg2"]
    n_idx, n_mask = int((idx != oi).sum()), int((ma != oma).sum())
    print(f"config 2, all {B * N} tiles [{numerics}] vs oracle: idx differ {n_idx}/{oi.size}, mask bits differ {n_mask}/{oma.size}; "
          f"score max err {np.abs(sc - osc).max():.2e}, sim_avg max err {np.abs(avg - oavg).max():.2e}")
    assert n_idx == 0 and n_mask == 0
    if numerics == "chain":
        np.testing.assert_array_equal(sc.view(np.uint32), osc.view(np.uint32))
        np.testing.assert_array_equal(avg.view(np.uint32), oavg.view(np.uint32))
    else:
        assert np.abs(sc - osc).max() <= 4e-6 and np.abs(avg - oavg).max() <= 1e-6


# ---------------------------------------------------------------------------------------------------------------- e2e
E2E_CONFIGS = {   # mirrors oracle/make_goldens.py: E2E_CONFIGS
    "e2e_cfg2": dict(seed=311, O=1, N=162, B=64, k=5, vit=(1024, 24, 16), name="dinov2_vitl14"),
    "e2e_cfg3": dict(seed=321, O=8, N=162, B=64, k=5, vit=(1024, 24, 16), name="dinov2_vitl14"),
}
_hf = {}


def hf_backbone(vit):
    """HF Dinov2Model with the golden generator's weights (synthetic.fill_state_dict seed 302), built once."""
    if vit not in _hf:
        from transformers import Dinov2Config, Dinov2Model

        dim, depth, heads = vit
        hf = Dinov2Model(Dinov2Config(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads,
                                      image_size=224, patch_size=14)).eval()
        _hf[vit] = syn.fill_state_dict(hf, 302)
    return _hf[vit]


def build_e2e_model(cfg, numerics):
    from test_gpu_e2e import FakeTemplates, e2e_inputs, make_batch
    from test_oracle_pose_ist import build_ist

    from gigapose_amd.ae_net import AENet
    from gigapose_amd.gigaPose import GigaPose
    from gigapose_amd.matching import LocalSimilarity
    from gigapose_amd.vit import Dinov2ViT

    dim = cfg["vit"][0]
    ae = AENet(cfg["name"], Dinov2ViT.from_hf(hf_backbone(cfg["vit"])), dim, 64)
    model = GigaPose("large", ae, build_ist(303, conditioned=True), None, LocalSimilarity(k=cfg["k"], sim_threshold=0.5, patch_threshold=3),
                     None, 1000, tempfile.mkdtemp(), max_num_dets_per_forward=4).eval().to(DEV)
    model.set_numerics(numerics)
    items, q = e2e_inputs(cfg["seed"], cfg["O"], cfg["N"], cfg["B"])
    model.template_datasets = {"syn": FakeTemplates(items)}
    model.test_dataset_name = "syn"
    return model, make_batch(q), q


@pytest.mark.parametrize("numerics", ["chain", "split"])
@pytest.mark.parametrize("which", ["e2e_cfg2", "e2e_cfg3"])
def test_eval_retrieval_at_benchmark_size_vs_reference_golden(golden_dir, which, numerics):
    from test_gpu_e2e import pose_rel_err

    g = np.load(os.path.join(golden_dir, which + ".npz"))
    cfg = E2E_CONFIGS[which]
    model, batch, q = build_e2e_model(cfg, numerics)
    assert model.test_step(batch, 0) == 0
    p = {n: v.cpu().numpy() for n, v in model.last_predictions.tensors.items()}
    np.testing.assert_allclose(model.template_datas["syn"].ae_features[0, 0].cpu().numpy(), g["tmpl_ae_feat_sample"], rtol=0, atol=3e-5)
    gid = g["id_src"].astype(np.int64)
    n_set = int((np.sort(p["id_src"], 1) != np.sort(gid, 1)).any(1).sum())
    n_order = int((p["id_src"] != gid).any(1).sum())
    d_src, d_tar = int((p["src_pts"] != g["src_pts"]).sum()), int((p["tar_pts"] != g["tar_pts"]).sum())
    d_cnt = np.abs(p["scores"] - g["all_scores"]) * 256
    valid = g["relScale"] > -999
    e_sc = np.abs(p["relScale"] - g["relScale"])[valid].max() if n_order == 0 and d_src == 0 else float("nan")
    print(f"{which} [{numerics}] vs reference: detections with a different top-k template SET {n_set}/{len(gid)}, different ORDER "
          f"{n_order}; src_pts differ {d_src}/{g['src_pts'].size}, tar_pts {d_tar}; inlier counts differ on "
          f"{int((d_cnt > 0).sum())}/{d_cnt.size} hypotheses; relScale max err {e_sc:.2e}")
    assert n_set == 0 and n_order == 0 and d_src == 0 and d_tar == 0
    np.testing.assert_allclose(p["score_src"], g["score_src"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(p["relScale"][valid], g["relScale"][valid], rtol=0, atol=2e-5)
    np.testing.assert_allclose(p["relInplane"][valid], g["relInplane"][valid], rtol=0, atol=2e-5)
    assert (d_cnt == 0).all(), "RANSAC inlier counts differ from the reference"
    np.testing.assert_array_equal(p["idx_failed"], g["idx_failed"])
    m_err = np.abs(p["M"] - g["M"]).max(axis=(-1, -2)) / np.abs(g["M"]).max(axis=(-1, -2))
    terr, rerr = pose_rel_err(p["pred_poses"], g["all_poses"])
    print(f"    M rel err max {m_err.max():.2e}; pose translation rel err max {terr.max():.2e}, rotation abs err max {rerr.max():.2e} "
          f"over all {terr.size} hypotheses")
    assert m_err.max() < 1e-4 and terr.max() < 1e-4 and rerr.max() < 1e-4   # the north-star tolerance, ALL hypotheses
    out = np.load(os.path.join(model.log_dir, "predictions", "0.npz"))
    np.testing.assert_array_equal(out["object_id"], g["object_id"])
    te, re_ = pose_rel_err(out["poses"], g["poses"])
    assert te.max() < 1e-4 and re_.max() < 1e-4
    np.testing.assert_array_equal(out["scores"], g["scores"])
