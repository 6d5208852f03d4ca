"""GPU parity of the fused HIP matcher (through the C-ABI) against the CPU oracle and the
reference goldens.  Index outputs: bit-exact.  Float outputs: ALSO bit-exact against the
oracle (same fmaf order by construction); 1e-6 against the reference goldens."""
import os

import numpy as np
import pytest
import torch

from gigapose_testing import synthetic as syn
from oracle import cpu as oracle
from test_oracle_matcher import CASES, VARIANT_CASES, load_case

pytestmark = pytest.mark.gpu
DEV = "cuda"


def run_hip(case, k):
    from gigapose_amd.matching import LocalSimilarity, MatchBank

    metric = LocalSimilarity(k=k, sim_threshold=0.5, patch_threshold=3)
    bank = MatchBank(torch.from_numpy(case["src_feats"]).to(DEV), torch.from_numpy(case["src_masks"]).to(DEV))
    out = metric.test_bank(bank, torch.from_numpy(case["tar_feat"]).to(DEV),
                           torch.from_numpy(case["tar_mask"]).to(DEV),
                           torch.from_numpy(case["labels"]).to(DEV))
    torch.cuda.synchronize()
    return metric, bank, {k_: v.cpu().numpy() for k_, v in out.tensors.items()}


@pytest.mark.parametrize("name", CASES)
def test_matcher_vs_oracle_bit_exact(golden_dir, name):
    g, case, k = load_case(golden_dir, name)
    _, _, hip = run_hip(case, k)
    ref = oracle.local_similarity_test(case["src_feats"], case["tar_feat"], case["src_masks"],
                                       case["tar_mask"], case["labels"], k)
    for key in ["id_src", "tar_pts", "src_pts"]:
        np.testing.assert_array_equal(hip[key], ref[key], err_msg=key)
    for key in ["score_src", "score_pts"]:  # bit-exact floats: compare raw bits
        np.testing.assert_array_equal(hip[key].view(np.uint32), ref[key].view(np.uint32), err_msg=key)


@pytest.mark.parametrize("numerics", ["chain", "split"])
@pytest.mark.parametrize("name", CASES)
def test_matcher_vs_reference_golden(golden_dir, name, numerics, monkeypatch):
    """Both numerics modes against the outputs of the unmodified reference: indices bit-exact, scores 1e-6."""
    monkeypatch.setenv("GIGAPOSE_NUMERICS", numerics)
    g, case, k = load_case(golden_dir, name)
    _, _, hip = run_hip(case, k)
    np.testing.assert_array_equal(hip["id_src"], g["id_src"])
    np.testing.assert_array_equal(hip["tar_pts"], g["tar_pts"].astype(np.int64))
    np.testing.assert_array_equal(hip["src_pts"], g["src_pts"].astype(np.int64))
    np.testing.assert_allclose(hip["score_src"], g["score_src"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(hip["score_pts"], g["score_pts"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("numerics", ["chain", "split"])
@pytest.mark.parametrize("name", VARIANT_CASES)
def test_matcher_variants_vs_reference_golden_and_oracle(golden_dir, name, numerics, monkeypatch):
    """search_direction = "src2tar" (reference matching.py:242-244) and patch_threshold <= 0 (no cycle check, :256-257): both
    numerics against goldens of the unmodified reference built with those ctor arguments; chain also bit-exact vs the oracle."""
    from gigapose_amd.matching import LocalSimilarity, MatchBank

    monkeypatch.setenv("GIGAPOSE_NUMERICS", numerics)
    g, case, k = load_case(golden_dir, name)
    direction, pthr = str(g["search_direction"]), float(g["patch_threshold"])
    metric = LocalSimilarity(k=k, sim_threshold=0.5, patch_threshold=pthr, search_direction=direction)
    assert metric.numerics == numerics
    bank = MatchBank(torch.from_numpy(case["src_feats"]).to(DEV), torch.from_numpy(case["src_masks"]).to(DEV))
    out = metric.test_bank(bank, torch.from_numpy(case["tar_feat"]).to(DEV), torch.from_numpy(case["tar_mask"]).to(DEV),
                           torch.from_numpy(case["labels"]).to(DEV))
    hip = {k_: v.cpu().numpy() for k_, v in out.tensors.items()}
    np.testing.assert_array_equal(hip["id_src"], g["id_src"])
    np.testing.assert_array_equal(hip["tar_pts"], g["tar_pts"].astype(np.int64))
    np.testing.assert_array_equal(hip["src_pts"], g["src_pts"].astype(np.int64))
    np.testing.assert_allclose(hip["score_src"], g["score_src"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(hip["score_pts"], g["score_pts"], rtol=0, atol=2e-6)
    if numerics == "chain":
        ref = oracle.local_similarity_test(case["src_feats"], case["tar_feat"], case["src_masks"], case["tar_mask"], case["labels"], k,
                                           patch_thr=pthr, search_direction=direction)
        for key in ["score_src", "score_pts"]:
            np.testing.assert_array_equal(hip[key].view(np.uint32), ref[key].view(np.uint32), err_msg=key)
    if pthr > 0:   # the reference's val() refuses patch_threshold <= 0 (matching.py:158-159) and so does the mirror
        return
    with pytest.raises(ValueError):
        metric.val(torch.from_numpy(case["tar_feat"]).to(DEV), torch.from_numpy(case["tar_feat"]).to(DEV),
                   torch.from_numpy(case["tar_mask"]).to(DEV), torch.from_numpy(case["tar_mask"]).to(DEV))


def _negative_case():
    """matcher_case whose templates 0 and 1 of every object are ANTI-correlated with every query patch (similarities around -0.1):
    with sim_threshold = -0.25 those live values survive and the masked-out patches' exact zeros win the reference's maxima."""
    case = syn.matcher_case(seed=31, B=3, O=2, N=5, C=64, noise=0.6)
    rs = np.random.RandomState(7)
    u = rs.standard_normal(64).astype(np.float32)
    u /= np.linalg.norm(u)
    def around(mean, shape):
        v = rs.standard_normal(shape + (64, 16, 16)).astype(np.float32)
        v -= np.einsum("...chw,c->...hw", v, u)[..., None, :, :] * u[:, None, None]     # orthogonal to u
        v /= np.linalg.norm(v, axis=-3, keepdims=True)
        return (mean * u[:, None, None] + np.sqrt(1 - mean * mean) * v).astype(np.float32)
    case["tar_feat"] = around(0.98, (3,))
    case["src_feats"][:, :2] = around(-0.12, (2, 2))
    return case


def test_negative_sim_threshold_runs_uncompacted():
    """Live-patch compaction of the split matcher is bit-identical to the full tile only for sim_threshold >= 0 (a row maximum of 0
    then means an all-zero row); with a negative threshold the launcher must run the full tile: split == chain on the indices of
    every tile and both equal the oracle's (a compacted run would return the negative live maximum instead of a masked zero)."""
    from gigapose_amd.matching import LocalSimilarity, MatchBank

    case = _negative_case()
    outs = {}
    for numerics in ("chain", "split"):
        metric = LocalSimilarity(k=5, sim_threshold=-0.25, patch_threshold=3)
        metric.numerics = numerics
        bank = MatchBank(torch.from_numpy(case["src_feats"]).to(DEV), torch.from_numpy(case["src_masks"]).to(DEV), numerics)
        from gigapose_amd.matching import patch_grid_mask
        idx, sc, ma, avg = metric.match_tiles(metric.normalize(torch.from_numpy(case["tar_feat"]).to(DEV)),
                                              patch_grid_mask(torch.from_numpy(case["tar_mask"]).to(DEV)), bank,
                                              torch.from_numpy(case["labels"]).to(DEV).int())
        outs[numerics] = [t.cpu().numpy() for t in (idx, sc, ma, avg)]
    q = oracle.l2norm_cp(case["tar_feat"].reshape(3, 64, 256))
    bk = oracle.l2norm_cp(case["src_feats"].reshape(2, 5, 64, 256))
    ref = oracle.match(q, bk, oracle.patch_mask(case["tar_mask"]), oracle.patch_mask(case["src_masks"]), case["labels"], thr=-0.25)
    qm = oracle.patch_mask(case["tar_mask"])
    assert ((ref[1][:, :2] == 0) & (qm[:, None] > 0)).sum() > 100, "the case must hold live rows whose maximum is a masked-out zero"
    np.testing.assert_array_equal(outs["chain"][0], ref[0])
    np.testing.assert_array_equal(outs["chain"][1].view(np.uint32), ref[1].view(np.uint32))
    flips = int((outs["split"][0] != ref[0]).sum())
    if flips > 2:   # diagnostics: where, and what the kernel returned there
        per_tile = (outs["split"][0] != ref[0]).sum(-1)
        b, n = np.unravel_index(per_tile.argmax(), per_tile.shape)
        t = np.flatnonzero(outs["split"][0][b, n] != ref[0][b, n])[:8]
        print("flips per (det, template):", per_tile.tolist(), "| tile", (b, n), "patches", t.tolist(), "ours idx", outs["split"][0][b, n, t].tolist(),
              "ref idx", ref[0][b, n, t].tolist(), "ours score", outs["split"][1][b, n, t].tolist(), "ref score", ref[1][b, n, t].tolist())
    assert flips <= 2, f"{flips} patch ids differ between split and the oracle at a negative threshold"
    np.testing.assert_allclose(outs["split"][1], ref[1], rtol=0, atol=3e-6)
    assert (outs["split"][2] != ref[2]).sum() <= 2


def test_l2norm_bit_exact():
    rs = np.random.RandomState(5)
    x = (rs.standard_normal((7, 384, 256)) * rs.uniform(0.1, 30, (7, 1, 256))).astype(np.float32)
    x[0, :, 3] = 0.0  # zero vector -> eps clamp path
    from gigapose_amd.matching import LocalSimilarity

    y = LocalSimilarity(5, 0.5, 3).normalize(torch.from_numpy(x).to(DEV)).cpu().numpy()
    np.testing.assert_array_equal(y.view(np.uint32), oracle.l2norm_cp(x).view(np.uint32))
    np.testing.assert_allclose(y, torch.nn.functional.normalize(torch.from_numpy(x), dim=1).numpy(),
                               rtol=2e-6, atol=1e-12)


def test_reference_signature_equals_bank_path():
    """`test(src_feats[B,N,...])` (matching.py:188 signature) == resident-bank path."""
    case = syn.matcher_case(seed=21, B=5, O=2, N=6, C=32)
    metric, _, a = run_hip(case, 5)
    lab = torch.from_numpy(case["labels"]).long()
    out = metric.test(torch.from_numpy(case["src_feats"])[lab].to(DEV), torch.from_numpy(case["tar_feat"]).to(DEV),
                      torch.from_numpy(case["src_masks"])[lab].to(DEV), torch.from_numpy(case["tar_mask"]).to(DEV),
                      max_batch_size=2)  # ragged chunks 2,2,1
    for key, v in out.tensors.items():
        np.testing.assert_array_equal(v.cpu().numpy(), a[key], err_msg=key)
    assert out.id_src.dtype == torch.int64 and out.tar_pts.dtype == torch.int64
    assert out.score_src.dtype == torch.float32 and tuple(out.src_pts.shape) == (5, 5, 256, 2)


def test_errors_and_edges():
    from gigapose_amd import _lib
    from gigapose_amd.matching import LocalSimilarity, MatchBank

    case = syn.matcher_case(seed=22, B=2, O=1, N=3, C=32)
    metric = LocalSimilarity(k=5, sim_threshold=0.5, patch_threshold=3)
    bank = MatchBank(torch.from_numpy(case["src_feats"]).to(DEV), torch.from_numpy(case["src_masks"]).to(DEV))
    with pytest.raises(RuntimeError):  # N=3 < k=5: torch.topk raises in the reference too
        metric.test_bank(bank, torch.from_numpy(case["tar_feat"]).to(DEV), torch.from_numpy(case["tar_mask"]).to(DEV),
                         torch.zeros(2, dtype=torch.int32, device=DEV))
    with pytest.raises(_lib.GigaPoseHipError):  # C not a multiple of 16
        q = torch.zeros(1, 24, 256, device=DEV)
        metric.match_tiles(q, torch.ones(1, 256, device=DEV),
                           type("B", (), dict(features=torch.zeros(1, 1, 24, 256, device=DEV),
                                              masks=torch.ones(1, 1, 256, device=DEV), O=1, N=1, C=24))(),
                           torch.zeros(1, dtype=torch.int32, device=DEV))
    # empty batch
    m4 = LocalSimilarity(k=3, sim_threshold=0.5, patch_threshold=3)
    out = m4.test_bank(bank, torch.zeros(0, 32, 16, 16, device=DEV), torch.zeros(0, 224, 224, device=DEV),
                       torch.zeros(0, dtype=torch.int32, device=DEV))
    assert tuple(out.id_src.shape) == (0, 3) and tuple(out.src_pts.shape) == (0, 3, 256, 2)
    # all-zero masks: nothing matches, ids are the first k templates, points all -1
    out = m4.test_bank(bank, torch.from_numpy(case["tar_feat"]).to(DEV), torch.zeros(2, 224, 224, device=DEV),
                       torch.zeros(2, dtype=torch.int32, device=DEV))
    assert out.id_src.cpu().tolist() == [[0, 1, 2]] * 2 and (out.tar_pts == -1).all() and not out.score_src.any()


def test_full_size_config2_properties():
    """BASELINE config 2 shape (B=64, N=162, C=1024): oracle check on a sample of tiles plus
    size-independent properties (template-permutation equivariance, planted top-1)."""
    from gigapose_amd.matching import LocalSimilarity, MatchBank

    B, N, C = 64, 162, 1024
    rs = np.random.RandomState(31)
    case = syn.matcher_case(seed=31, B=B, O=1, N=N, C=C, full_masks=True)
    # plant: query b is a noisy copy of template (7*b) % N -> that template must win
    for b in range(B):
        t = case["src_feats"][0, (7 * b) % N] + 0.05 * syn._unit(rs.standard_normal((C, 16, 16)), 0)
        case["tar_feat"][b] = syn._unit(t, 0)
    metric, bank, hip = run_hip(case, 5)
    assert (hip["id_src"][:, 0] == (7 * np.arange(B)) % N).all()
    # oracle on a sample of (b, n) tiles
    q = metric.normalize(torch.from_numpy(case["tar_feat"]).to(DEV))
    from gigapose_amd.matching import patch_grid_mask
    idx, sc, ma, avg = metric.match_tiles(q, patch_grid_mask(torch.from_numpy(case["tar_mask"]).to(DEV)), bank,
                                          torch.zeros(B, dtype=torch.int32, device=DEV))
    bs, ns = [0, 17, 63], [0, 7, 119, 161]
    qn = oracle.l2norm_cp(case["tar_feat"].reshape(B, C, 256))[bs]
    bn = oracle.l2norm_cp(case["src_feats"].reshape(1, N, C, 256))[:, ns]
    oi, osc, oma, oavg = oracle.match(qn, bn, np.ones((len(bs), 256), np.float32),
                                      np.ones((1, len(ns), 256), np.float32), np.zeros(len(bs), np.int32))
    sel = np.ix_(bs, ns)
    np.testing.assert_array_equal(idx.cpu().numpy()[sel], oi)
    np.testing.assert_array_equal(sc.cpu().numpy()[sel].view(np.uint32), osc.view(np.uint32))
    np.testing.assert_array_equal(ma.cpu().numpy()[sel], oma)
    np.testing.assert_array_equal(avg.cpu().numpy()[sel].view(np.uint32), oavg.view(np.uint32))
    # permuting the templates permutes sim_avg and leaves the winners' records unchanged
    perm = rs.permutation(N)
    case2 = dict(case, src_feats=case["src_feats"][:, perm], src_masks=case["src_masks"][:, perm])
    _, _, hip2 = run_hip(case2, 5)
    np.testing.assert_array_equal(perm[hip2["id_src"][:, 0]], hip["id_src"][:, 0])
    np.testing.assert_array_equal(hip2["src_pts"][:, 0], hip["src_pts"][:, 0])
    np.testing.assert_array_equal(hip2["score_src"][:, 0].view(np.uint32), hip["score_src"][:, 0].view(np.uint32))


@pytest.mark.parametrize("numerics", ["chain", "split"])
def test_val_matches_reference_golden_and_oracle(golden_dir, numerics, monkeypatch):
    """LocalSimilarity.val through the HIP tile kernel: reference golden (indices bit-exact, scores 1e-6); chain mode
    additionally bit-exact against the oracle."""
    import ast

    from gigapose_amd.matching import LocalSimilarity

    monkeypatch.setenv("GIGAPOSE_NUMERICS", numerics)
    g = np.load(os.path.join(golden_dir, "match_val.npz"))
    case = syn.matcher_case(**ast.literal_eval(str(g["case_kwargs"])))
    t = lambda a: torch.from_numpy(a).to(DEV)
    out = LocalSimilarity(k=1, sim_threshold=0.5, patch_threshold=3).val(t(case["src_feats"][case["labels"], 0]), t(case["tar_feat"]),
                                                                         t(case["src_masks"][case["labels"], 0]), t(case["tar_mask"]))
    np.testing.assert_array_equal(out.src_pts.cpu().numpy(), g["src_pts"].astype(np.int64))
    np.testing.assert_array_equal(out.tar_pts.cpu().numpy(), g["tar_pts"].astype(np.int64))
    np.testing.assert_allclose(out.score.cpu().numpy(), g["score"], rtol=0, atol=1e-6)
    if numerics == "chain":
        ref = oracle.local_similarity_val(case["src_feats"][case["labels"], 0], case["tar_feat"], case["src_masks"][case["labels"], 0], case["tar_mask"])
        np.testing.assert_array_equal(out.score.cpu().numpy().view(np.uint32), ref["score"].view(np.uint32))


@pytest.mark.probes
@pytest.mark.parametrize("C", [64, 1024])
def test_split_matcher_live_patch_compaction_is_bit_identical(C):
    """The split matcher builds its tile from the live (mask != 0) patches only (gp_match.hip: 1..2 x 1..4 matrix tiles per wave
    instead of always 2 x 4).  Masks of every size -- empty, a single patch, patch 0 only, 31 / 32 / 33 / 64 / 65 / 127 / 128 / 129 /
    255 live patches, full, fractional values -- on both sides: every output must equal the uncompacted launch bit for bit
    (scores and sim_avg included)."""
    import ctypes  # noqa: F401

    from gigapose_amd import _lib
    from gigapose_amd.matching import LocalSimilarity, MatchBank, normalize_split

    rs = np.random.RandomState(5)
    counts = [0, 1, 31, 32, 33, 64, 65, 127, 128, 129, 160, 255, 256]
    O, N, B = 1, len(counts), len(counts) + 2
    feats = torch.from_numpy(syn._unit(rs.standard_normal((O * N + B, C, 256)).astype(np.float32), 1)).to(DEV)
    # planted similar patches so that thresholds pass and exact ties between columns exist (duplicated template patches)
    feats[O * N:] = feats[:1] * 0.9 + 0.1 * feats[O * N:]
    feats[1:O * N] = feats[:1] * 0.8 + 0.2 * feats[1:O * N]
    feats[:O * N, :, 7] = feats[:O * N, :, 200]

    def masks(rows):
        m = np.zeros((rows, 256), np.float32)
        for r in range(rows):
            c = counts[r % len(counts)]
            m[r, rs.permutation(256)[:c]] = 1.0
        return m

    bm, qm = masks(O * N), masks(B)
    bm[2] = 0; bm[2, 0] = 1.0                      # only patch 0 (the "no match" sentinel) live
    qm[B - 1] = rs.rand(256).astype(np.float32)    # fractional mask values multiply the similarity (matching.py:234-235)
    qm[B - 2] = 1.0
    metric = LocalSimilarity(k=5, sim_threshold=0.5, patch_threshold=3)
    metric.numerics = "split"
    bank = MatchBank.__new__(MatchBank)
    bank.numerics, bank.bank_dtype, bank.O, bank.N, bank.C, bank.features = "split", "f32", O, N, C, None
    hi, lo = normalize_split(feats[:O * N])
    bank.hi, bank.lo = hi.view(O, N, 256, -1), lo.view(O, N, 256, -1)
    bank.masks = torch.from_numpy(bm).to(DEV).view(O, N, 256)
    q = normalize_split(feats[O * N:])
    qmask = torch.from_numpy(qm).to(DEV)
    labels0 = torch.zeros(B, dtype=torch.int32, device=DEV)
    lib = _lib.lib()
    try:
        lib.gp_match_split_set_compact(0)
        full = [t.clone() for t in metric.match_tiles(q, qmask, bank, labels0)]
    finally:
        lib.gp_match_split_set_compact(1)
    comp = metric.match_tiles(q, qmask, bank, labels0)
    for name, a, b in zip(["idx", "score", "mask", "sim_avg"], full, comp):
        assert torch.equal(a, b), f"{name} differs between the compacted and the full tile"
    assert (full[2].sum() > 50) and (full[2].sum(-1) > 0).any(), "the case exercises no valid correspondence"
    print(f"compaction C={C}: {int(full[2].sum())} valid correspondences over {B * N} tiles, all outputs bit-identical")


def test_select_topk_equals_topk_gather_format():
    """gp_select_topk (one launch: what the hot loop calls since round 5) == gp_topk + gp_gather_records + gp_format_points +
    the int64 cast, on random tile records with EXACT ties in sim_avg (tie rule: higher score, then lower template index), k = N,
    k = 1, N not a multiple of 4 (the shared-memory layout pads the flags)."""
    from gigapose_amd.matching import LocalSimilarity

    rs = np.random.RandomState(12)
    for B, N, k in [(7, 162, 5), (3, 13, 13), (5, 6, 1), (64, 162, 5)]:
        metric = LocalSimilarity(k=k, sim_threshold=0.5, patch_threshold=3)
        avg = rs.randint(0, 12, (B, N)).astype(np.float32) / 16.0          # many exact ties
        idx = rs.randint(0, 256, (B, N, 256)).astype(np.uint8)
        sc = rs.rand(B, N, 256).astype(np.float32)
        ma = (rs.rand(B, N, 256) > 0.6).astype(np.float32)
        t = lambda a: torch.from_numpy(a).to(DEV)
        avg_d, idx_d, sc_d, ma_d = t(avg), t(idx), t(sc), t(ma)
        ids, score = metric.topk(avg_d)
        rec_idx, rec_score, rec_mask = metric.gather_records(ids, idx_d, sc_d, ma_d)
        tar, src = metric.format_points(rec_idx, rec_mask)
        ids2, score2, rec_score2, tar2, src2 = metric.select_topk(avg_d, idx_d, sc_d, ma_d)
        assert ids2.dtype == torch.int64 and torch.equal(ids2, ids.long()) and torch.equal(score2, score)
        assert torch.equal(rec_score2, rec_score) and torch.equal(tar2, tar) and torch.equal(src2, src)
    with pytest.raises(RuntimeError):
        LocalSimilarity(k=7, sim_threshold=0.5, patch_threshold=3).select_topk(avg_d[:, :6].contiguous(), idx_d, sc_d, ma_d)


def test_patch_masks_written_by_the_normalise_launch_equal_the_strided_copy():
    """gp_l2norm_split_mask (round 5): the query's 16 x 16 patch masks come out of the normalise + split launch; they must equal
    patch_grid_mask (= F.interpolate(mask, (16, 16)) nearest, matching.py:222, 227) for 224 x 224 and other 16-divisible sizes, and the
    planes must be the ones gp_l2norm_split writes.  Other dtypes / layouts take the strided copy."""
    from gigapose_amd.matching import normalize_split, patch_grid_mask

    rs = np.random.RandomState(3)
    for B, C, H, W in [(5, 64, 224, 224), (3, 384, 64, 96), (64, 1024, 224, 224)]:
        feats = torch.from_numpy(rs.standard_normal((B, C, 256)).astype(np.float32)).to(DEV)
        mask = torch.from_numpy((rs.rand(B, H, W) > 0.4).astype(np.float32) * rs.rand(B, H, W).astype(np.float32)).to(DEV)
        hi0, lo0 = normalize_split(feats)
        hi, lo, qm = normalize_split(feats, mask)
        assert torch.equal(hi, hi0) and torch.equal(lo, lo0) and torch.equal(qm, patch_grid_mask(mask))
        hi, lo, qm = normalize_split(feats, mask.double())                       # not f32: the fallback path, same values
        assert torch.equal(qm, patch_grid_mask(mask)) and torch.equal(hi, hi0)
        hi, lo, qm = normalize_split(feats, mask.transpose(1, 2).contiguous().transpose(1, 2))   # not contiguous
        assert torch.equal(qm, patch_grid_mask(mask))
