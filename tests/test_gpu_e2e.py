"""GPU end-to-end parity: gigapose_amd.GigaPose.eval_retrieval vs the golden written by the
UNMODIFIED reference GigaPose.eval_retrieval (oracle/make_goldens.py: gen_e2e; BASELINE config-1
shape: ViT-S/14 stand-in, few templates, CPU reference path)."""
import os
import sys
import tempfile

import numpy as np
import pandas as pd
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from gigapose_testing import synthetic as syn
from oracle import ist_torch
from test_oracle_pose_ist import build_ist

pytestmark = pytest.mark.gpu
DEV = "cuda"
E2E = dict(seed=301, O=2, N=6, B=3, k=4, vit=(384, 12, 6))


class FakeTemplates:
    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def e2e_inputs(seed, O, N, B):
    import types

    rs = np.random.RandomState(seed)
    tK, tM, tP = syn.template_geometry(seed + 1, O, N)
    items, all_t, all_m = [], [], []
    for o in range(O):
        t, m = syn.template_images(seed + 10 + o, N)
        all_t.append(t)
        all_m.append(m)
        items.append(types.SimpleNamespace(rgb=torch.from_numpy(t), mask=torch.from_numpy(m), K=torch.from_numpy(tK[o]),
                                           M=torch.from_numpy(tM[o]), poses=torch.from_numpy(tP[o])))
    labels = rs.randint(1, O + 1, B)
    views = rs.randint(0, N, B)
    imgs = np.stack([all_t[l - 1][v] for l, v in zip(labels, views)])
    msk = np.stack([all_m[l - 1][v] for l, v in zip(labels, views)])
    imgs = (imgs + 0.1 * rs.standard_normal(imgs.shape).astype(np.float32)) * msk[:, None]
    qK, qM = syn.crop_geometry(seed + 2, B)
    return items, dict(tar_img=imgs.astype(np.float32), tar_mask=msk, tar_K=qK, tar_M=qM, labels=labels, views=views)


def build_model(hf_features=False):
    """gigapose_amd model with the same deterministic weights the golden generator gave the
    reference (synthetic.fill_state_dict keyed by parameter names)."""
    from transformers import Dinov2Config, Dinov2Model

    from gigapose_amd.ae_net import AENet
    from gigapose_amd.gigaPose import GigaPose
    from gigapose_amd.matching import LocalSimilarity
    from gigapose_amd.vit import Dinov2ViT

    dim, depth, heads = E2E["vit"]
    hf = Dinov2Model(Dinov2Config(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads,
                                  image_size=224, patch_size=14)).eval()
    syn.fill_state_dict(hf, 302)
    ae = AENet("dinov2_vits14", Dinov2ViT.from_hf(hf), dim, 64)
    ist = build_ist(303, conditioned=True)   # as the golden generator (make_goldens.py: gen_e2e)
    metric = LocalSimilarity(k=E2E["k"], sim_threshold=0.5, patch_threshold=3)
    model = GigaPose("large", ae, ist, None, metric, None, 1000, tempfile.mkdtemp(), max_num_dets_per_forward=4)
    return model.eval().to(DEV), hf


def make_batch(q):
    from gigapose_amd.tensor_collection import PandasTensorCollection

    B = len(q["labels"])
    infos = pd.DataFrame(dict(label=[str(l) for l in q["labels"]], scene_id=[1] * B, view_id=[7] * B))
    batch = PandasTensorCollection(infos=infos, **{k: torch.from_numpy(q[k]).to(DEV) for k in ["tar_img", "tar_mask", "tar_K", "tar_M"]})
    objs = sorted(set(int(l) for l in q["labels"]))
    batch.test_list = PandasTensorCollection(infos=pd.DataFrame(dict(
        im_id=[7] * len(objs), scene_id=[1] * len(objs), obj_id=objs,
        inst_count=[int((q["labels"] == o).sum()) for o in objs], detection_time=[0.1] * len(objs))))
    return batch


def pose_rel_err(a, b):
    t = np.linalg.norm(a[..., :3, 3] - b[..., :3, 3], axis=-1) / np.linalg.norm(b[..., :3, 3], axis=-1)
    r = np.abs(a[..., :3, :3] - b[..., :3, :3]).max(axis=(-1, -2))
    return t, r


@pytest.mark.parametrize("numerics", ["chain", "split"])
def test_eval_retrieval_matches_reference_golden(golden_dir, numerics, monkeypatch):
    monkeypatch.setenv("GIGAPOSE_NUMERICS", numerics)   # read when the ViT / matcher / bank are constructed
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    model, _ = build_model()
    assert model.ae_net.dinov2_model.numerics == numerics and model.testing_metric.numerics == numerics
    items, q = e2e_inputs(E2E["seed"], E2E["O"], E2E["N"], E2E["B"])
    model.template_datasets = {"syn": FakeTemplates(items)}
    model.test_dataset_name = "syn"
    batch = make_batch(q)
    tiles, match_tiles = {}, model.testing_metric.match_tiles
    model.testing_metric.match_tiles = lambda *a, **kw: tiles.setdefault("out", match_tiles(*a, **kw))   # every tile's record, for the checker
    assert model.test_step(batch, 0) == 0
    model.flush_pending()   # test_step queues whole images (GigaPose.accumulate_crops, default 64): run + write what is pending
    p = model.last_predictions
    # onboarding produced the same features as the reference's (ViT-S on CPU)
    np.testing.assert_allclose(model.template_datas["syn"].ae_features[0, 0].cpu().numpy(), g["tmpl_ae_feat_sample"],
                               rtol=0, atol=3e-5)
    # Hypotheses are sorted by inlier count (gigaPose.py:588-595, ties stable in both): same ids in the same order
    def mine(name):
        return getattr(p, name).cpu().numpy()

    np.testing.assert_array_equal(mine("id_src"), g["id_src"].astype(np.int64))
    # (1) everything decided by the ViT + matcher: bit-exact template ids and patch correspondences
    np.testing.assert_array_equal(mine("src_pts"), g["src_pts"].astype(np.int64))
    np.testing.assert_array_equal(mine("tar_pts"), g["tar_pts"].astype(np.int64))
    np.testing.assert_allclose(mine("score_src"), g["score_src"], rtol=0, atol=2e-5)
    # (2) IST regression (conditioned init, synthetic.condition_ist: outputs of O(1) as a trained net's): f32 round-off
    valid = g["relScale"] > -999
    np.testing.assert_allclose(mine("relScale")[valid], g["relScale"][valid], rtol=0, atol=2e-5)
    np.testing.assert_allclose(mine("relInplane")[valid], g["relInplane"][valid], rtol=0, atol=2e-5)
    # (3) RANSAC: identical inlier counts, failure flags and winning candidates for ALL hypotheses
    np.testing.assert_array_equal(mine("scores"), g["all_scores"])
    np.testing.assert_array_equal(mine("idx_failed"), g["idx_failed"])
    # Equal inlier COUNTS do not pin the winning candidate: many-to-one matches put correspondences at exactly one patch (14 px) from
    # the proposing one, so candidates tie and rounding picks among them.  Every hypothesis must therefore either carry the
    # reference's similarity M and pose to the north-star's 1e-4, or be an EXPLAINED tie against the reference evaluated in float64
    # (tests/parity_explain.py with the margins golden of this configuration) -- no hypothesis is exempt.
    import parity_explain as px
    from test_gpu_parity_big import EPS_PX, EPS_SIM, ours_for_checker

    m = dict(np.load(os.path.join(golden_dir, "e2e_margins.npz")))
    p_np = {n: v.cpu().numpy() for n, v in p.tensors.items()}
    rep = px.explain(m, ours_for_checker(model, p_np, tiles["out"], m), eps_sim=EPS_SIM, eps_px=EPS_PX,
                     geom=px.geometry(E2E["seed"], E2E["O"], E2E["N"], E2E["B"]))
    assert rep["hyp_checked"] == rep["hyp"] == mine("id_src").size   # every hypothesis' pose checked, one way or the other
    m_err = np.abs(mine("M") - g["M"]).max(axis=(-1, -2)) / np.abs(g["M"]).max(axis=(-1, -2))
    same = m_err < 1e-4
    terr, rerr = pose_rel_err(mine("pred_poses")[same], g["all_poses"][same])
    print("e2e [%s] vs reference: %d of %d hypotheses with the reference's RANSAC winner; on them M rel err %.2e, translation rel %.2e, "
          "rotation abs %.2e | vs its float64 run: %s" % (numerics, same.sum(), same.size, m_err[same].max(), terr.max(), rerr.max(), px.summary(rep)))
    assert not rep["unexplained"], rep["unexplained"][:5]
    assert same.sum() >= same.size - 1               # at most one hypothesis on another (explained, 14.000 px) RANSAC tie than the f32 golden
    assert terr.max() < 1e-4 and rerr.max() < 1e-4   # the north-star tolerance
    # what filter_and_save wrote (the reference's on-disk contract, gigaPose.py:439-448)
    out = np.load(os.path.join(model.log_dir, "predictions", "0.npz"))
    np.testing.assert_array_equal(out["object_id"], g["object_id"])
    assert out["poses"].shape == g["poses"].shape and out["scores"].shape == g["scores"].shape
    assert out["poses"].dtype == np.float32 and out["scene_id"].dtype == np.int32


def test_voting_and_recovery_reproduce_reference_exactly_given_its_regressions(golden_dir):
    """Feed the REFERENCE's own relScale / relInplane (golden) into gp_ransac + gp_recover_poses: the
    discrete voting result and the poses must then equal the reference's to f32 round-off -- this
    isolates stages 5-6 from the conditioning of the random-init IST network."""
    from gigapose_amd.poses import ObjectPoseRecovery

    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    items, q = e2e_inputs(E2E["seed"], E2E["O"], E2E["N"], E2E["B"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    rec = ObjectPoseRecovery(torch.stack([it.K for it in items]).to(DEV), torch.stack([it.M for it in items]).to(DEV),
                             torch.stack([it.poses for it in items]).to(DEV))
    M, failed, isrc, itar, isc = rec.ransac.run(t(g["src_pts"].astype(np.int64)), t(g["tar_pts"].astype(np.int64)),
                                                t(g["relScale"]), t(g["relInplane"]))
    np.testing.assert_array_equal((isc.sum(-1) / 256).cpu().numpy(), g["all_scores"])
    np.testing.assert_array_equal(M.cpu().numpy().view(np.uint32), g["M"].view(np.uint32))  # bit-exact M
    poses = rec.forward_recovery(torch.from_numpy(q["labels"]), t(q["tar_K"]), t(q["tar_M"]), t(g["id_src"].astype(np.int64)), M).cpu().numpy()
    terr, rerr = pose_rel_err(poses, g["all_poses"])
    assert terr.max() < 1e-4 and rerr.max() < 1e-4, (terr.max(), rerr.max())  # north-star tolerance
    print("stages 5-6 vs reference: translation rel %.2e, rotation abs %.2e" % (terr.max(), rerr.max()))


def test_ist_outputs_as_close_to_f64_truth_as_the_reference(golden_dir):
    """relScale from the HIP path and from the reference (golden, torch-CPU f32) against a float64
    evaluation of the same network on the same correspondences: both f32 results sit at a similar
    distance from the exact answer, i.e. the HIP path is as accurate as the reference."""
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    model, _ = build_model()
    items, q = e2e_inputs(E2E["seed"], E2E["O"], E2E["N"], E2E["B"])
    model.template_datasets = {"syn": FakeTemplates(items)}
    batch = make_batch(q)
    model.eval_retrieval(batch, 0, "syn")
    p = model.last_predictions
    ist64 = build_ist(303, conditioned=True).double()
    with torch.no_grad():
        tar64 = ist_torch.resnet_forward(ist64.backbone, torch.from_numpy(q["tar_img"]).double()).reshape(E2E["B"], 256, 256)
        tmpl64 = [ist_torch.resnet_forward(ist64.backbone, it.rgb.double()).reshape(-1, 256, 256) for it in items]
    # evaluate the scale head in f64 for the golden's hypothesis order
    errs_mine, errs_ref, mags = [], [], []
    mine_ids = p.id_src.cpu().numpy()
    for b in range(E2E["B"]):
        for j in range(E2E["k"]):
            tid = int(g["id_src"][b, j])
            jm = int(np.flatnonzero(mine_ids[b] == tid)[0])
            sp, tp = g["src_pts"][b, j].astype(np.int64), g["tar_pts"][b, j].astype(np.int64)
            ok = sp[:, 0] >= 0
            if not ok.any():
                continue
            si, ti = sp[ok, 1] * 16 + sp[ok, 0], tp[ok, 1] * 16 + tp[ok, 0]
            feats = torch.cat([tar64[b][:, ti].t(), tmpl64[int(q["labels"][b]) - 1][tid][:, si].t()], dim=1)
            with torch.no_grad():
                truth = ist64.regressor.scale_predictor(feats)[:, 0].numpy()
            errs_ref.append(np.abs(g["relScale"][b, j][ok] - truth))
            errs_mine.append(np.abs(p.relScale[b, jm].cpu().numpy()[ok] - truth))
            mags.append(np.abs(truth))
    mag = np.median(np.concatenate(mags))
    em, er = np.concatenate(errs_mine) / mag, np.concatenate(errs_ref) / mag
    print("relScale error vs f64 truth (relative to median |scale| = %.3f): HIP median %.2e p99 %.2e | "
          "reference (f32 CPU) median %.2e p99 %.2e" % (mag, np.median(em), np.percentile(em, 99), np.median(er),
                                                        np.percentile(er, 99)))
    assert np.median(em) <= 3 * np.median(er) + 1e-6 and np.percentile(em, 99) <= 5 * np.percentile(er, 99) + 1e-5


@pytest.mark.parametrize("numerics", ["chain", "split"])
def test_template_sharded_path_over_rccl_equals_unsharded(monkeypatch, numerics):
    """The N>1 path on the 1-GPU box: torch.distributed 'nccl' (= RCCL) process group of ONE rank with the
    all-gathers forced (GIGAPOSE_FORCE_COLLECTIVES): exchange #1/#2, packing, merge and the hand-over to IST /
    RANSAC / recovery run on device buffers through RCCL and must reproduce the unsharded predict() exactly.
    (world_size 2 equality of the exchange + merge logic: tests/test_sharding_gloo.py on CPU.)"""
    import socket

    import torch.distributed as dist

    from gigapose_testing import factory

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(port))
    monkeypatch.setenv("GIGAPOSE_FORCE_COLLECTIVES", "1")
    monkeypatch.setenv("GIGAPOSE_NUMERICS", numerics)   # split: the exchange carries the f16 hi/lo query planes
    dev = torch.device("cuda", 0)
    tset = factory.TemplateSet(2, 9, seed=60)
    q = tset.crops(61, 5, dev)

    def run(sharded):
        model = factory.build_model("dinov2_vits14", k=4, device=dev, seed=5)
        if sharded:
            model.enable_template_sharding()
        model.template_datasets = {"syn": tset}
        model.set_template_data("syn")
        p = model.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn")
        torch.cuda.synchronize()
        return {n: v.cpu() for n, v in p.tensors.items()}

    plain = run(False)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        shard = run(True)
    finally:
        dist.destroy_process_group()
    assert set(plain) == set(shard)
    for n in plain:
        assert torch.equal(plain[n], shard[n]), f"{n} differs between the sharded and the unsharded path"


@pytest.mark.parametrize("numerics", ["chain", "split"])
def test_ist_backbone_on_a_second_stream_gives_the_same_predictions(monkeypatch, numerics):
    """`overlap_ist` (GIGAPOSE_OVERLAP_IST = 1 / auto): the IST backbone runs on a side stream next to ViT + matching -- below 64 crops the
    two chains together fill the chip (+7 % at 8 crops).  Same kernels with deterministic reductions: every tensor of predict() is equal."""
    from gigapose_testing import factory

    monkeypatch.setenv("GIGAPOSE_NUMERICS", numerics)
    dev = torch.device("cuda", 0)
    tset = factory.TemplateSet(2, 9, seed=70)
    model = factory.build_model("dinov2_vits14", k=4, device=dev, seed=6)
    model.template_datasets = {"syn": tset}
    model.set_template_data("syn")
    assert model.overlap_ist == "auto"   # the product default since round 5
    out = {}
    for B in (5, 40):
        q = tset.crops(71 + B, B, dev)
        for mode in (False, True, "auto"):
            model.overlap_ist = mode
            for _ in range(2):   # twice: the side stream is created on first use
                p = model.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn")
            torch.cuda.synchronize()
            out[(B, mode)] = {n: v.cpu() for n, v in p.tensors.items()}
        for mode in (True, "auto"):
            for n, v in out[(B, False)].items():
                assert torch.equal(v, out[(B, mode)][n]), f"{n} differs with overlap_ist = {mode!r} at {B} crops"
    model.overlap_ist = False
