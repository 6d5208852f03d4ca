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
from gigapose_testing import synthetic as syn  # noqa: E402

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


# The ctor arguments the released configs do not use (round 4): search_direction = "src2tar" (matching.py:242-244) and
# patch_threshold <= 0 = no cycle check (matching.py:256-257).  (name, matcher_case kwargs, k, search_direction, patch_threshold)
MATCHER_VARIANT_CASES = [
    ("match_s2t_small", dict(seed=11, B=3, O=2, N=6, C=32), 5, "src2tar", 3),
    ("match_s2t_noshift", dict(seed=15, B=4, O=3, N=9, C=96, shift=False, noise=0.2), 5, "src2tar", 3),
    ("match_nocycle_small", dict(seed=11, B=3, O=2, N=6, C=32), 5, "tar2src", 0),
    ("match_nocycle_fullmask", dict(seed=14, B=2, O=2, N=7, C=48, full_masks=True), 5, "tar2src", -1),
    ("match_s2t_nocycle_vits", dict(seed=12, B=2, O=1, N=8, C=384), 5, "src2tar", 0),
]


def gen_matcher_variants():
    ref_shim.install()
    from src.models.matching import LocalSimilarity

    for name, kw, k, direction, patch_thr in MATCHER_VARIANT_CASES:
        case = syn.matcher_case(**kw)
        metric = LocalSimilarity(k=k, sim_threshold=0.5, patch_threshold=patch_thr, search_direction=direction)
        labels = torch.from_numpy(case["labels"]).long()
        out = metric.test(src_feats=torch.from_numpy(case["src_feats"])[labels], tar_feat=torch.from_numpy(case["tar_feat"]),
                          src_masks=torch.from_numpy(case["src_masks"])[labels], tar_mask=torch.from_numpy(case["tar_mask"]))
        np.savez_compressed(
            os.path.join(GOLD, name + ".npz"), input_checksum=syn.checksum(*[case[x] for x in sorted(case)]),
            case_kwargs=repr(kw), k=k, search_direction=direction, patch_threshold=patch_thr,
            id_src=out.id_src.numpy(), score_src=out.score_src.numpy(), score_pts=out.score_pts.numpy(),
            tar_pts=out.tar_pts.numpy().astype(np.int16), src_pts=out.src_pts.numpy().astype(np.int16))
        print(name, direction, patch_thr, "id_src[0] =", out.id_src[0].tolist(), "valid pts:", int((out.tar_pts[..., 0] >= 0).sum()))


# BASELINE config-2 / config-3 sized feature-level cases: (name, matcher_case kwargs, k)
BIG_MATCHER_CASES = [
    ("match_cfg2", dict(seed=21, B=64, O=1, N=162, C=1024), 5),     # ViT-L width, 1 object x 162 templates, 64 crops
    ("match_cfg3", dict(seed=22, B=64, O=8, N=162, C=1024), 5),     # LM-O shape: 8 objects, labels mixed
    ("match_cfg5", dict(seed=23, B=64, O=40, N=162, C=1024), 5),    # BASELINE config 5: 40 objects x 162 templates (6.8 GB of f32 features)
]


def gen_matcher_big(only=None):
    """LocalSimilarity.test of the unmodified reference at the benchmark sizes.  The reference chunks detections
    itself (matching.py:201-214) and concatenates, so calling it on 8 detections at a time (to bound the
    gathered-bank copy, 1.36 GB per 8) gives exactly what one call would."""
    ref_shim.install()
    from src.models.matching import LocalSimilarity

    torch.set_num_threads(8)
    for name, kw, k in BIG_MATCHER_CASES:
        if only is not None and name not in only:
            continue
        case = syn.matcher_case(**kw)
        metric = LocalSimilarity(k=k, sim_threshold=0.5, patch_threshold=3)
        labels = torch.from_numpy(case["labels"]).long()
        bank, masks = torch.from_numpy(case["src_feats"]), torch.from_numpy(case["src_masks"])
        tar_feat, tar_mask = torch.from_numpy(case["tar_feat"]), torch.from_numpy(case["tar_mask"])
        outs = {n: [] for n in ["id_src", "score_src", "score_pts", "tar_pts", "src_pts"]}
        for s0 in range(0, kw["B"], 8):
            sl = slice(s0, s0 + 8)
            with torch.no_grad():
                o = metric.test(src_feats=bank[labels[sl]], tar_feat=tar_feat[sl], src_masks=masks[labels[sl]],
                                tar_mask=tar_mask[sl])
            for n in outs:
                outs[n].append(getattr(o, n).numpy())
        out = {n: np.concatenate(v) for n, v in outs.items()}
        np.savez_compressed(
            os.path.join(GOLD, name + ".npz"),
            input_checksum=syn.checksum(*[case[x] for x in sorted(case)]), case_kwargs=repr(kw), k=k,
            id_src=out["id_src"].astype(np.int16), score_src=out["score_src"], score_pts=out["score_pts"],
            tar_pts=out["tar_pts"].astype(np.int8), src_pts=out["src_pts"].astype(np.int8))
        print(name, "id_src[0] =", out["id_src"][0].tolist(), "score_src[0] =", np.round(out["score_src"][0], 4).tolist(),
              "valid pts:", int((out["tar_pts"][..., 0] >= 0).sum()), flush=True)


def gen_val():
    """LocalSimilarity.val (reference matching.py:115-186): one template per detection (the validation-time matcher)."""
    ref_shim.install()
    from src.models.matching import LocalSimilarity

    kw = dict(seed=16, B=6, O=3, N=1, C=64, noise=0.25, shift=False)
    case = syn.matcher_case(**kw)
    metric = LocalSimilarity(k=1, sim_threshold=0.5, patch_threshold=3)
    out = metric.val(src_feat=torch.from_numpy(case["src_feats"][case["labels"], 0]), tar_feat=torch.from_numpy(case["tar_feat"]),
                     src_mask=torch.from_numpy(case["src_masks"][case["labels"], 0]), tar_mask=torch.from_numpy(case["tar_mask"]))
    np.savez_compressed(os.path.join(GOLD, "match_val.npz"), input_checksum=syn.checksum(*[case[x] for x in sorted(case)]),
                        case_kwargs=repr(kw), src_pts=out.src_pts.numpy().astype(np.int16),
                        tar_pts=out.tar_pts.numpy().astype(np.int16), score=out.score.numpy())
    print("val: valid pts", int((out.tar_pts[..., 0] >= 0).sum()), "score max", float(out.score.max()))


IST_CFG = dict(n_heads=0, input_dim=3, input_size=256, initial_dim=128, block_dims=[128, 192, 256, 512],
               descriptor_size=256)


def build_ref_ist(seed, conditioned=False):
    ref_shim.install()
    from src.models.network.ist_net import ISTNet, Regressor
    from src.models.network.resnet import ResNet

    net = ISTNet("resnet", ResNet(dict(IST_CFG)), Regressor(256, 256, True, True), 64).eval()
    syn.fill_state_dict(net, seed)
    return syn.condition_ist(net) if conditioned else net


def gen_ist():
    """ISTNet.forward_by_chunk (ResNet, resnet.py:364-381) and ISTNet.inference (ist_net.py:97-120)."""
    torch.set_num_threads(8)
    net = build_ref_ist(seed=101)
    tmpl, masks = syn.template_images(102, 2)
    with torch.no_grad():
        feat = net.forward_by_chunk(torch.from_numpy(tmpl))           # (2,256,16,16)
    rs = np.random.RandomState(103)
    B = 3
    src_feat = rs.standard_normal((B, 256, 16, 16)).astype(np.float32)
    tar_feat = rs.standard_normal((B, 256, 16, 16)).astype(np.float32)
    corr = syn.correspondences_case(104, B, 1)
    with torch.no_grad():
        sc, cs = net.inference(torch.from_numpy(src_feat), torch.from_numpy(tar_feat),
                               torch.from_numpy(corr["src_pts"][:, 0]), torch.from_numpy(corr["tar_pts"][:, 0]))
    np.savez_compressed(os.path.join(GOLD, "ist.npz"), resnet_feat=feat.numpy(), scales=sc.numpy(),
                        cos_sin=cs.numpy(), n_state=len(net.state_dict()),
                        state_names="|".join(sorted(net.state_dict())))
    print("ist: resnet feat", tuple(feat.shape), float(feat.abs().mean()), "valid rows", int((sc > -999).sum()))


def gen_pose():
    """RANSAC.forward per hypothesis (ransac.py:108-172) via ObjectPoseRecovery.forward_ransac
    (poses.py:124-163) and forward_recovery (poses.py:103-122)."""
    ref_shim.install()
    import pandas as pd
    from src.megapose.utils.tensor_collection import PandasTensorCollection
    from src.models.poses import ObjectPoseRecovery

    B, k, O, N = 4, 5, 2, 6
    corr = syn.correspondences_case(201, B, k)
    tK, tM, tP = syn.template_geometry(202, O, N)
    qK, qM = syn.crop_geometry(203, B)
    rs = np.random.RandomState(204)
    labels = rs.randint(1, O + 1, B)
    id_src = rs.randint(0, N, (B, k))
    rec = ObjectPoseRecovery(torch.from_numpy(tK), torch.from_numpy(tM), torch.from_numpy(tP))
    pred = PandasTensorCollection(infos=pd.DataFrame(), src_pts=torch.from_numpy(corr["src_pts"]),
                                  tar_pts=torch.from_numpy(corr["tar_pts"]),
                                  relScale=torch.from_numpy(corr["rel_scale"]),
                                  relInplane=torch.from_numpy(corr["rel_inplane"]))
    pred = rec.forward_ransac(pred)
    poses = rec.forward_recovery(torch.from_numpy(labels), torch.from_numpy(qK), torch.from_numpy(qM),
                                 torch.from_numpy(id_src), pred.M.clone())
    np.savez_compressed(os.path.join(GOLD, "pose.npz"), labels=labels, id_src=id_src, M=pred.M.numpy(),
                        idx_failed=pred.idx_failed.numpy(), ransac_scores=pred.ransac_scores.numpy().astype(np.int8),
                        ransac_src_pts=pred.ransac_src_pts.numpy().astype(np.int16),
                        ransac_tar_pts=pred.ransac_tar_pts.numpy().astype(np.int16), poses=poses.numpy())
    print("pose: inlier counts", pred.ransac_scores.sum(-1).tolist(), "failed", pred.idx_failed.tolist())
    # boundary case: errors of exactly 14 px, problem sizes on both sides of torch's bmm switch (n = 46)
    from src.models.ransac import RANSAC

    case = syn.many_to_one_case(211, 14)
    batch = PandasTensorCollection(infos=pd.DataFrame(), src_pts=torch.from_numpy(case["src_pts"]),
                                   tar_pts=torch.from_numpy(case["tar_pts"]), relScale=torch.from_numpy(case["rel_scale"]),
                                   relInplane=torch.from_numpy(case["rel_inplane"]))
    Ms, failed, out = RANSAC(pixel_threshold=14)(batch)
    np.savez_compressed(os.path.join(GOLD, "pose_boundary.npz"), M=Ms.numpy(), idx_failed=failed.numpy(),
                        scores=out.scores.numpy().astype(np.int8), src_pts=out.src_pts.numpy().astype(np.int16),
                        tar_pts=out.tar_pts.numpy().astype(np.int16))
    print("pose_boundary: sizes", (case["src_pts"][..., 0] >= 0).sum(-1).tolist(), "inliers", out.scores.sum(-1).tolist())


class _FakeTemplates:
    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def e2e_inputs(seed, O, N, B):
    """Config-1-like end-to-end inputs: templates, crops = noisy templates, geometry."""
    import types

    rs = np.random.RandomState(seed)
    tK, tM, tP = syn.template_geometry(seed + 1, O, N)
    items, all_t, all_m = [], [], []
    for o in range(O):
        t, m = syn.template_images(seed + 10 + o, N)
        all_t.append(t)
        all_m.append(m)
        items.append(types.SimpleNamespace(rgb=torch.from_numpy(t), mask=torch.from_numpy(m),
                                           K=torch.from_numpy(tK[o]), M=torch.from_numpy(tM[o]),
                                           poses=torch.from_numpy(tP[o])))
    labels = rs.randint(1, O + 1, B)
    views = rs.randint(0, N, B)
    imgs = np.stack([all_t[l - 1][v] for l, v in zip(labels, views)])
    msk = np.stack([all_m[l - 1][v] for l, v in zip(labels, views)])
    imgs = (imgs + 0.1 * rs.standard_normal(imgs.shape).astype(np.float32)) * msk[:, None]
    qK, qM = syn.crop_geometry(seed + 2, B)
    return items, dict(tar_img=imgs.astype(np.float32), tar_mask=msk, tar_K=qK, tar_M=qM, labels=labels, views=views)


# name -> end-to-end configuration.  "e2e": BASELINE config-1 shape; "e2e_cfg2": config 2 (ViT-L/14, 1 object x 162
# templates, 64 crops); "e2e_cfg3": config 3 (LM-O shape: 8 objects x 162 templates, 64 crops with mixed labels).
# The IST net uses the conditioned init (synthetic.condition_ist) so the 1e-4 pose tolerance is assertable.
E2E_CONFIGS = {
    "e2e": dict(seed=301, O=2, N=6, B=3, k=4, vit=(384, 12, 6), name="dinov2_vits14"),
    "e2e_cfg2": dict(seed=311, O=1, N=162, B=64, k=5, vit=(1024, 24, 16), name="dinov2_vitl14"),
    "e2e_cfg3": dict(seed=321, O=8, N=162, B=64, k=5, vit=(1024, 24, 16), name="dinov2_vitl14"),
}
E2E = E2E_CONFIGS["e2e"]


def run_ref_e2e(which, double=False, dets_per_forward=4, model=None):
    """Whole GigaPose.eval_retrieval (gigaPose.py:481-633) of the unmodified reference: HF DINOv2 stand-in backbone,
    reference ISTNet/LocalSimilarity/ObjectPoseRecovery, CPU.  double: the SAME code with every module and input cast to
    float64 (the reference's arithmetic without its f32 rounding).  dets_per_forward: configs/test.yaml:21's
    max_num_dets_per_forward (a memory knob; changes the shapes of the reference's own GEMMs).  Returns (model, arrays)."""
    import tempfile

    import pandas as pd
    ref_shim.install()
    torch.set_num_threads(8)
    from src.megapose.utils.tensor_collection import PandasTensorCollection
    from src.models.gigaPose import GigaPose
    from src.models.matching import LocalSimilarity
    from src.models.network.ae_net import AENet

    cfg = E2E_CONFIGS[which]
    dim, depth, heads = cfg["vit"]
    dt = torch.float64 if double else torch.float32
    tmp = tempfile.mkdtemp()
    if model is None:
        backbone = ref_shim.HFDinov2Backbone.build(dim, depth, heads, seed=0)
        syn.fill_state_dict(backbone.m, 302)
        ae = AENet(cfg["name"], backbone, dim, 64)
        ist = build_ref_ist(seed=303, conditioned=True)
        metric = LocalSimilarity(k=cfg["k"], sim_threshold=0.5, patch_threshold=3)
        model = GigaPose("large", ae, ist, None, metric, None, 1000, tmp, max_num_dets_per_forward=dets_per_forward).eval().to(dt)
    else:
        model.log_dir = tmp
        os.makedirs(os.path.join(tmp, "predictions"), exist_ok=True)
    model.max_num_dets_per_forward = dets_per_forward
    items, q = e2e_inputs(cfg["seed"], cfg["O"], cfg["N"], cfg["B"])
    for it in items:
        for n in ["rgb", "mask", "K", "M", "poses"]:
            setattr(it, n, getattr(it, n).to(dt))
    model.template_datasets = {"syn": _FakeTemplates(items)}
    infos = pd.DataFrame(dict(label=[str(l) for l in q["labels"]], scene_id=[1] * cfg["B"], view_id=[7] * cfg["B"]))
    batch = PandasTensorCollection(infos=infos, tar_img=torch.from_numpy(q["tar_img"]).to(dt),
                                   tar_mask=torch.from_numpy(q["tar_mask"]).to(dt), tar_K=torch.from_numpy(q["tar_K"]).to(dt),
                                   tar_M=torch.from_numpy(q["tar_M"]).to(dt))
    objs = sorted(set(int(l) for l in q["labels"]))
    batch.test_list = PandasTensorCollection(infos=pd.DataFrame(dict(
        im_id=[7] * len(objs), scene_id=[1] * len(objs), obj_id=objs,
        inst_count=[int((q["labels"] == o).sum()) for o in objs], detection_time=[0.1] * len(objs))))
    # tie order of torch.argsort is unspecified (gigaPose.py:591): pin it to stable in the HARNESS only
    orig_argsort = torch.argsort
    torch.argsort = lambda x, dim=-1, descending=False, stable=False: orig_argsort(x, dim=dim, descending=descending, stable=True)
    captured = {}
    orig_fs = GigaPose.filter_and_save.__get__(model)

    def spy(predictions, **kw):
        captured["pred"] = predictions.clone()
        return orig_fs(predictions, **kw)

    model.filter_and_save = spy
    if double:  # the reference creates its temporaries with torch's default dtype (matching.py:275, ransac.py:139)
        torch.set_default_dtype(torch.float64)
    try:
        with torch.no_grad():
            model.eval_retrieval(batch, 0, "syn")
    finally:
        torch.argsort = orig_argsort
        torch.set_default_dtype(torch.float32)
        del model.filter_and_save
    out = np.load(os.path.join(tmp, "predictions", "0.npz"))
    p = captured["pred"]
    td = model.template_datas["syn"]
    arrays = dict(poses=out["poses"], scores=out["scores"],
                  object_id=out["object_id"], id_src=p.id_src.numpy().astype(np.int16), score_src=p.score_src.numpy(),
                  score_pts=p.score_pts.numpy(),
                  all_scores=p.scores.numpy(), all_poses=p.pred_poses.numpy(), M=p.M.numpy(),
                  relScale=p.relScale.numpy(), relInplane=p.relInplane.numpy(), idx_failed=p.idx_failed.numpy(),
                  src_pts=p.src_pts.numpy().astype(np.int8), tar_pts=p.tar_pts.numpy().astype(np.int8),
                  tmpl_ae_feat_sample=td.ae_features[0, 0].numpy(), ist_conditioned=True)
    print(which, "double" if double else "f32", "dets/forward", dets_per_forward, ": id_src", p.id_src[:4].tolist(), "scores",
          np.round(p.scores[:4].numpy(), 4).tolist())
    print("     valid corr", (p.src_pts[..., 0] >= 0).sum(-1)[:8].tolist(), "failed", int(p.idx_failed.sum()), flush=True)
    return model, arrays


def gen_pose_scored():
    """RANSAC.forward with its two optional arguments (ransac.py:108-121): `scores` (per-correspondence weights) and
    direction="tar2src" (the `*_inv` fields).  Unused at inference, present on the class: the unmodified reference's outputs on the
    many-to-one boundary case (exact 14 px ties) with integer-valued weights 0..3 (f32 sums of those are exact in any order,
    so torch.sum's unspecified order cannot matter) -> tests/golden/pose_scored.npz."""
    ref_shim.install()
    import pandas as pd
    from src.megapose.utils.tensor_collection import PandasTensorCollection
    from src.models.ransac import RANSAC

    case = syn.many_to_one_case(221, 14)
    inv = syn.many_to_one_case(222, 14)
    rs = np.random.RandomState(223)
    weights = rs.randint(0, 4, case["rel_scale"].shape).astype(np.float32)
    weights[3] = 0.0                                         # a problem whose every weight is zero: failed = True with inliers present
    batch = PandasTensorCollection(infos=pd.DataFrame(), src_pts=torch.from_numpy(case["src_pts"]), tar_pts=torch.from_numpy(case["tar_pts"]),
                                   relScale=torch.from_numpy(case["rel_scale"]), relInplane=torch.from_numpy(case["rel_inplane"]),
                                   src_pts_inv=torch.from_numpy(inv["src_pts"]), tar_pts_inv=torch.from_numpy(inv["tar_pts"]),
                                   relScale_inv=torch.from_numpy(inv["rel_scale"]), relInplane_inv=torch.from_numpy(inv["rel_inplane"]))
    out = {"weights": weights}
    ransac = RANSAC(pixel_threshold=14)
    for tag, kw in (("s2t_w", dict(scores=torch.from_numpy(weights))), ("t2s", dict(direction="tar2src")),
                    ("t2s_w", dict(scores=torch.from_numpy(weights), direction="tar2src"))):
        Ms, failed, o = ransac(batch, **kw)
        out.update({f"{tag}_M": Ms.numpy(), f"{tag}_failed": failed.numpy(), f"{tag}_scores": o.scores.numpy().astype(np.int8),
                    f"{tag}_src_pts": o.src_pts.numpy().astype(np.int16), f"{tag}_tar_pts": o.tar_pts.numpy().astype(np.int16)})
        print("pose_scored", tag, "failed", failed.tolist(), "weighted inliers", o.scores.sum(-1).tolist())
    np.savez_compressed(os.path.join(GOLD, "pose_scored.npz"), **out)


def gen_e2e(which="e2e"):
    _, arrays = run_ref_e2e(which)
    np.savez_compressed(os.path.join(GOLD, which + ".npz"), **arrays)


def disagreement(a, b):
    """Counts used by tests/test_gpu_parity_big.py: detections whose top-k template SET / ORDER differ, differing
    correspondence entries, hypotheses with a different inlier count."""
    ida, idb = a["id_src"].astype(np.int64), b["id_src"].astype(np.int64)
    return dict(set=int((np.sort(ida, 1) != np.sort(idb, 1)).any(1).sum()), order=int((ida != idb).any(1).sum()),
                src_pts=int((a["src_pts"] != b["src_pts"]).sum()), tar_pts=int((a["tar_pts"] != b["tar_pts"]).sum()),
                inliers=int((a["all_scores"] != b["all_scores"]).sum()))


def gen_e2e_f64(which="e2e_cfg2"):
    """The reference's own code in float64 (`model.double()`, double inputs): what the f32 runs -- the reference's and
    ours -- are approximations of.  Written to <which>_f64.npz (index / count fields + poses).  Also measured here: how
    far the reference's f32 golden is from it, and how far the reference is from ITSELF when only
    max_num_dets_per_forward (configs/test.yaml:21) changes from 4 to 8 -- the yardsticks for the GPU parity bar."""
    g32 = dict(np.load(os.path.join(GOLD, which + ".npz")))
    model, a8 = run_ref_e2e(which, dets_per_forward=8)
    d_self = disagreement(a8, g32)
    print(which, "reference f32, max_num_dets_per_forward 8 vs 4 (the golden):", d_self, flush=True)
    del model
    _, a64 = run_ref_e2e(which, double=True)
    d_64 = disagreement(g32, a64)
    print(which, "reference f32 golden vs reference in float64:", d_64, flush=True)
    keep = ["id_src", "src_pts", "tar_pts", "all_scores", "idx_failed", "score_src"]
    np.savez_compressed(os.path.join(GOLD, which + "_f64.npz"), **{k: a64[k] for k in keep}, all_poses=a64["all_poses"].astype(np.float64),
                        M=a64["M"].astype(np.float64), ref_f32_vs_f64=repr(d_64), ref_f32_dets8_vs_dets4=repr(d_self))


CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]   # reference configs/data/transform.yaml:6-7
CLIP_STD = [0.26862954, 0.26130258, 0.27577711]


def gen_crop():
    """Detection pre-processing as GigaPoseTestSet.process_real + collate_fn do it (reference
    src/dataloader/train.py:80-123, src/dataloader/test.py:295-315): rgb/255 * mask, CropResizePad on the RGBA
    stack, torchvision Normalize (un-vendored; restated here as its documented `(x - mean) / std`)."""
    ref_shim.install()
    from src.utils.crop import CropResizePad

    case = syn.detection_case(seed=401)
    rgb = torch.from_numpy(case["rgb"]) / 255.0
    masks = torch.from_numpy(case["masks"])
    boxes = torch.from_numpy(case["boxes"])
    m_rgb = rgb[torch.from_numpy(case["im_id"]).long()] * masks[:, None]
    m_rgba = torch.cat([m_rgb, masks[:, None]], dim=1)
    out = CropResizePad(target_size=224)(boxes, images=m_rgba)
    mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
    std = torch.tensor(CLIP_STD).view(3, 1, 1)
    tar_img = (out["images"][:, :3] - mean) / std
    np.savez_compressed(os.path.join(GOLD, "crop.npz"), seed=401, tar_img=tar_img.numpy(),
                        tar_mask=out["images"][:, -1].numpy(), M=out["M"].numpy(),
                        input_checksum=np.float64(case["rgb"].astype(np.float64).sum() + case["masks"].sum() + case["boxes"].sum()))
    print("crop: images", tuple(out["images"].shape), "M[0]", out["M"][0].tolist())


def gen_bop_csv():
    """The reference's BOP writer (src/utils/inout.py:278-367) on synthetic per-batch npz files, for an LM-O-named
    and a plain dataset: the csv texts are the golden."""
    import tempfile

    ref_shim.install()
    from src.utils.inout import save_predictions_from_batched_predictions

    gold = {"seed": 501}
    for ds in ["lmo", "ycbv"]:
        with tempfile.TemporaryDirectory() as tmp:
            for i, b in enumerate(syn.prediction_batches(501)):
                np.savez(os.path.join(tmp, f"{i}.npz"), **b)
            save_predictions_from_batched_predictions(tmp, dataset_name=ds, model_name="large", run_id="r0", is_refined=False)
            for f in sorted(os.listdir(tmp)):
                if f.endswith(".csv"):
                    gold[f"{ds}:{f}"] = np.frombuffer(open(os.path.join(tmp, f), "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "bop_csv.npz"), **gold)
    print("bop_csv:", [k for k in gold if k != "seed"])


STAGES = {"bop_csv": gen_bop_csv, "val": gen_val, "matcher": gen_matcher, "matcher_variants": gen_matcher_variants, "ist": gen_ist, "pose": gen_pose, "pose_scored": gen_pose_scored, "e2e": gen_e2e, "crop": gen_crop,
          "matcher_big": lambda: gen_matcher_big(["match_cfg2", "match_cfg3"]), "matcher_cfg5": lambda: gen_matcher_big(["match_cfg5"]), "e2e_cfg2": lambda: gen_e2e("e2e_cfg2"), "e2e_cfg3": lambda: gen_e2e("e2e_cfg3"),
          "e2e_cfg2_f64": lambda: gen_e2e_f64("e2e_cfg2"), "e2e_cfg3_f64": lambda: gen_e2e_f64("e2e_cfg3"), "e2e_f64": lambda: gen_e2e_f64("e2e")}

if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    todo = sys.argv[1:] or list(STAGES)
    for s in todo:
        STAGES[s]()
