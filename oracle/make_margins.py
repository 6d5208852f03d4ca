"""Decision margins of the reference's float64 run -- the evidence an "equal or explained tie" parity assertion needs
(tests/parity_explain.py, tests/test_gpu_parity_big.py).  TEST INFRASTRUCTURE ONLY; runs in the build container (imports
the unmodified reference through oracle/ref_shim.py), writes tests/golden/<which>_margins.npz.

    python oracle/make_margins.py e2e_cfg2 [e2e_cfg3 ...]

What is stored, all from GigaPose.eval_retrieval of the reference evaluated in float64 (gigaPose.py:481-633):
  sim_avg (B,N) f64                   the tensor torch.topk ranks (matching.py:274-279), captured at the topk call
  tile_min_margin (B,N) f32           smallest decision margin inside each (detection, template) tile (see below)
  top_ids (B,K) i16                   the K = 12 best templates per detection by (sim_avg desc, id asc)
  tile_b, tile_n (T,) i16             the tiles whose per-patch records are stored: those K per detection + every tile with a
                                      decision margin < 4e-6 whose sim_avg is within 0.02 of the detection's k-th best
  for those T tiles, per patch (T,256) (matching.py:233-272 restated in numpy float64 on the reference's own float64
  features; the restatement is checked against the reference's outputs on its k winners before anything is written):
    idx_t2s u8                        row argmax of the thresholded similarity, first max (the reference's idx_tar2src)
    ridx_t2s / idx2_t2s u8            argmax and runner-up of the raw (masked, not thresholded) row
    row_margin f32                    top1 - top2 of the raw row, clipped to CLIP
    row_thr f32                       raw row max - sim_threshold, clipped to +-CLIP
    row_max f32                       raw row max (score_tar2src where it is >= the threshold)
    idx_s2t / ridx_s2t / idx2_s2t, col_margin, col_thr, col_max   the same per template patch (column)
    valid u8                          mask_all (matching.py:268)
  relScale / relInplane f64 (B,k,P[,2]) and the hypothesis tensors of the run (sorted order, as the goldens)
The same capture of the reference's own float32 run is written next to it (<which>_ref32_tiles.npz: valid / idx of the
T stored tiles and sim_avg of ALL tiles, via LocalSimilarity.test with k = N) -- the CPU test of the explanation checker uses it as
the "other implementation", which also measures what epsilon the reference's own rounding needs.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from oracle import make_goldens as mg  # noqa: E402
from gigapose_testing import synthetic as syn  # noqa: E402

P = 256
CLIP = 1e-3
K_STORE = 12
EXTRA_MARGIN = 4e-6
EXTRA_WINDOW = 0.02
SIM_THR = 0.5
PATCH_THR = 3.0


def patch_mask(m224):
    """F.interpolate(mask, size=(16,16)) nearest on a 224 grid = pixel (14 i, 14 j) (matching.py:222-227)."""
    return np.ascontiguousarray(m224[..., ::14, ::14]).reshape(*m224.shape[:-2], P).astype(np.float64)


def tile_records(q, s, qm, sm):
    """One (detection, template) tile in float64.  q, s: (C,256) features as AENet emits them; qm, sm: (256,) patch masks.
    Restates matching.py:229-272.  Returns dict of per-patch records (see module docstring) + sim_avg."""
    qn = q / np.maximum(np.linalg.norm(q, axis=0, keepdims=True), 1e-12)          # F.normalize(tar_feat, dim=1)
    sn = s / np.maximum(np.linalg.norm(s, axis=0, keepdims=True), 1e-12)          # F.normalize(src_feats, dim=2)
    raw = qn.T @ sn                                                               # (t, s)
    raw = raw * sm[None, :]
    raw = raw * qm[:, None]
    sim = np.where(raw < SIM_THR, 0.0, raw)
    ar = np.arange(P)
    # rows (query patch t -> template patch s)
    idx_t2s = sim.argmax(1)                                                       # first max
    score_t2s = sim[ar, idx_t2s]
    rmax_idx = raw.argmax(1)
    rmax = raw[ar, rmax_idx]
    tmp = raw.copy()
    tmp[ar, rmax_idx] = -np.inf
    idx2_t2s = tmp.argmax(1)
    row_margin = rmax - tmp[ar, idx2_t2s]
    # columns (template patch s -> query patch t)
    idx_s2t = sim.argmax(0)
    score_s2t = sim[idx_s2t, ar]
    cmax_idx = raw.argmax(0)
    cmax = raw[cmax_idx, ar]
    tmp = raw.copy()
    tmp[cmax_idx, ar] = -np.inf
    idx2_s2t = tmp.argmax(0)
    col_margin = cmax - tmp[idx2_s2t, ar]
    # masks (matching.py:246-268)
    mask_sim = score_t2s >= SIM_THR
    s2s = idx_s2t[idx_t2s]
    dist = np.hypot((s2s % 16) - (ar % 16), (s2s // 16) - (ar // 16))
    mask_cycle = (dist <= PATCH_THR) & (score_s2t[idx_t2s] >= SIM_THR)
    nonzero = qm * sm[idx_t2s] * (idx_s2t != 0) * (idx_t2s != 0)                  # (idx_src2tar != 0) indexed by t: matching.py:266
    valid = mask_sim & mask_cycle & (nonzero > 0)
    sim_avg = (score_t2s * valid).sum() / P if valid.any() else 0.0
    # a row / column whose raw maximum is 0 (fully masked) decides nothing: margins of masked rows are reported as CLIP
    row_live, col_live = qm > 0, sm > 0
    row_margin = np.where(row_live, row_margin, CLIP)
    col_margin = np.where(col_live, col_margin, CLIP)
    row_thr = np.where(row_live, rmax - SIM_THR, -CLIP)
    col_thr = np.where(col_live, cmax - SIM_THR, -CLIP)
    # below the threshold the argmax is over an all-zero row: the runner-up is irrelevant, only the threshold decides
    row_margin = np.where(rmax < SIM_THR - CLIP, CLIP, row_margin)
    col_margin = np.where(cmax < SIM_THR - CLIP, CLIP, col_margin)
    return dict(idx_t2s=idx_t2s, ridx_t2s=rmax_idx, idx2_t2s=idx2_t2s, row_margin=row_margin, row_thr=row_thr, row_max=rmax,
                idx_s2t=idx_s2t, ridx_s2t=cmax_idx, idx2_s2t=idx2_s2t, col_margin=col_margin, col_thr=col_thr, col_max=cmax,
                valid=valid, sim_avg=sim_avg)


NAMES_U8 = ["idx_t2s", "ridx_t2s", "idx2_t2s", "idx_s2t", "ridx_s2t", "idx2_s2t", "valid"]
NAMES_F32 = ["row_margin", "row_thr", "row_max", "col_margin", "col_thr", "col_max"]


def tile_min_margin(r):
    return float(min(r["row_margin"].min(), np.abs(r["row_thr"]).min(), r["col_margin"].min(), np.abs(r["col_thr"]).min()))


def run_captured(which, double, k_all=False):
    """run_ref_e2e with spies: AE features of the crops, sim_avg at the topk call.  k_all: LocalSimilarity with k = N (every
    template's record comes back; used for the float32 run only -- eval_retrieval's later stages then see N hypotheses, so
    that run stops after the matcher)."""
    ref_shim.install()
    from src.models import matching as ref_matching

    cap = dict(tar_feat=[], sim_avg=[], match_out=[])
    orig_test = ref_matching.LocalSimilarity.test
    orig_topk = torch.topk

    def topk_spy(x, k, dim=-1, **kw):
        cap["sim_avg"].append(x.detach().clone().numpy())
        return orig_topk(x, k, dim=dim, **kw)

    def test_spy(self, src_feats, tar_feat, src_masks, tar_mask, max_batch_size=None):
        cap["tar_feat"].append(tar_feat.detach().clone().numpy())
        torch.topk = topk_spy
        try:
            if k_all:
                k0, self.k = self.k, src_feats.shape[1]
                out = orig_test(self, src_feats, tar_feat, src_masks, tar_mask, max_batch_size)
                self.k = k0
                cap["match_out"].append({n: getattr(out, n).numpy() for n in ["id_src", "score_src", "score_pts", "tar_pts", "src_pts"]})
            return orig_test(self, src_feats, tar_feat, src_masks, tar_mask, max_batch_size)
        finally:
            torch.topk = orig_topk

    ref_matching.LocalSimilarity.test = test_spy
    try:
        model, arrays = mg.run_ref_e2e(which, double=double)
    finally:
        ref_matching.LocalSimilarity.test = orig_test
    cap["tar_feat"] = np.concatenate(cap["tar_feat"])
    cap["sim_avg"] = np.concatenate(cap["sim_avg"][1::2] if k_all else cap["sim_avg"])
    td = model.template_datas["syn"]
    cap["bank"] = td.ae_features.numpy()          # (O,N,C,16,16)
    cap["bank_mask"] = td.mask.numpy()            # (O,N,224,224)
    return cap, arrays


def run_phase(which, scratch="/tmp"):
    """Phase 1 (slow: two passes of the reference through a 24-layer ViT on CPU): everything phase 2 needs, to scratch."""
    cap, a64 = run_captured(which, double=True)
    np.savez(os.path.join(scratch, f"{which}_f64_run.npz"), tar_feat=cap["tar_feat"], bank=cap["bank"], bank_mask=cap["bank_mask"],
             sim_avg=cap["sim_avg"], **{"a_" + n: v for n, v in a64.items()})
    del cap, a64
    cap32, a32 = run_captured(which, double=False, k_all=True)
    mo = {n: np.concatenate([m[n] for m in cap32["match_out"]]) for n in cap32["match_out"][0]}
    np.savez(os.path.join(scratch, f"{which}_f32_run.npz"), tar_feat=cap32["tar_feat"], bank=cap32["bank"], sim_avg=cap32["sim_avg"],
             **{"m_" + n: v for n, v in mo.items()}, **{"a_" + n: v for n, v in a32.items()})


def records_phase(which, scratch="/tmp"):
    """Phase 2 (a minute): tile records of the float64 run + the float32 run's view of the same tiles -> tests/golden."""
    cfg = mg.E2E_CONFIGS[which]
    O, N, B, k = cfg["O"], cfg["N"], cfg["B"], cfg["k"]
    items, q = mg.e2e_inputs(cfg["seed"], O, N, B)
    labels0 = q["labels"].astype(np.int64) - 1
    qmask = patch_mask(q["tar_mask"])
    r64 = np.load(os.path.join(scratch, f"{which}_f64_run.npz"))
    a64 = {n[2:]: r64[n] for n in r64.files if n.startswith("a_")}
    C = r64["tar_feat"].shape[1]
    bank = r64["bank"].reshape(O, N, C, P)
    bmask = patch_mask(r64["bank_mask"])
    tq = r64["tar_feat"].reshape(B, C, P)
    sim_avg = r64["sim_avg"]
    assert sim_avg.shape == (B, N) and sim_avg.dtype == np.float64
    K = min(K_STORE, N)
    order = np.lexsort((np.arange(N)[None, :].repeat(B, 0), -sim_avg), axis=1)[:, :K]      # (score desc, id asc)
    kth = np.sort(sim_avg, axis=1)[:, -min(k, N)]                                           # the k-th best sim_avg per detection
    # Tiles whose per-patch records are stored: the K best templates of every detection, plus every tile that holds a near-tied
    # decision (margin < EXTRA_MARGIN) and whose sim_avg is within EXTRA_WINDOW of the k-th best -- a tie there can move a few
    # patches (one column decision gates every query patch matched to it) and lift the template into another run's top k.
    tile_b, tile_n, recs = [], [], {n: [] for n in NAMES_U8 + NAMES_F32}
    tmm = np.zeros((B, N), np.float32)
    my_avg = np.zeros((B, N))
    for b in range(B):
        o = labels0[b]
        for n in range(N):
            r = tile_records(tq[b], bank[o, n], qmask[b], bmask[o, n])
            my_avg[b, n] = r["sim_avg"]
            tmm[b, n] = tile_min_margin(r)
            if n in order[b] or (tmm[b, n] < EXTRA_MARGIN and sim_avg[b, n] >= kth[b] - EXTRA_WINDOW):
                tile_b.append(b)
                tile_n.append(n)
                for nm in NAMES_U8:
                    recs[nm].append(r[nm].astype(np.uint8))
                for nm in NAMES_F32:
                    recs[nm].append((r[nm] if nm.endswith("_max") else np.clip(r[nm], -CLIP, CLIP)).astype(np.float32))
        if b % 16 == 0:
            print(f"{which}: tiles of detection {b} done", flush=True)
    tile_b, tile_n = np.asarray(tile_b, np.int16), np.asarray(tile_n, np.int16)
    rec = {nm: np.stack(v) for nm, v in recs.items()}
    print(f"{which}: {len(tile_b)} tiles stored ({B * K} = the {K} best per detection + {len(tile_b) - B * K} near-tied ones near the top-{k} boundary)")
    # the restatement must BE the reference: sim_avg of every tile, and the correspondences of the k winners
    err = np.abs(my_avg - sim_avg).max()
    print(f"{which}: numpy float64 restatement vs the reference's float64 sim_avg: max |diff| {err:.3e}")
    assert err < 1e-12, err
    ids64 = a64["id_src"].astype(np.int64)
    where = {(int(b), int(n)): i for i, (b, n) in enumerate(zip(tile_b, tile_n))}
    for b in range(B):
        for j in range(k):
            i = where[(b, int(ids64[b, j]))]
            valid = rec["valid"][i].astype(bool)
            s = rec["idx_t2s"][i].astype(np.int64)
            src = np.where(valid[:, None], np.stack([s % 16, s // 16], -1), -1)
            tar = np.where(valid[:, None], np.stack([np.arange(P) % 16, np.arange(P) // 16], -1), -1)
            assert (src == a64["src_pts"][b, j]).all() and (tar == a64["tar_pts"][b, j]).all(), (b, j)
    print(f"{which}: restated tile records reproduce the reference's float64 correspondences of all {B * k} winners")
    out = dict(sim_avg=sim_avg, tile_min_margin=tmm, top_ids=order.astype(np.int16), tile_b=tile_b, tile_n=tile_n, clip=CLIP, **rec,
               id_src=a64["id_src"], src_pts=a64["src_pts"], tar_pts=a64["tar_pts"], all_scores=a64["all_scores"],
               idx_failed=a64["idx_failed"], score_src=a64["score_src"], relScale=a64["relScale"].astype(np.float64),
               relInplane=a64["relInplane"].astype(np.float64), M=a64["M"].astype(np.float64),
               all_poses=a64["all_poses"].astype(np.float64))
    path = os.path.join(mg.GOLD, which + "_margins.npz")
    np.savez_compressed(path, **out)
    print(which, "margins written:", os.path.getsize(path) / 1e6, "MB", flush=True)

    # ---- the reference's own float32 run seen on the same tiles (LocalSimilarity.test with k = N returned every record)
    r32 = np.load(os.path.join(scratch, f"{which}_f32_run.npz"))
    ids = r32["m_id_src"]                                                # (B,N) templates sorted by sim_avg
    bi = np.arange(B)[:, None]
    valid32 = np.zeros((B, N, P), bool)
    idx32 = np.zeros((B, N, P), np.uint8)
    valid32[bi, ids] = r32["m_src_pts"][..., 0] >= 0
    s = r32["m_src_pts"][..., 1] * 16 + r32["m_src_pts"][..., 0]
    idx32[bi, ids] = np.where(r32["m_src_pts"][..., 0] >= 0, s, 0).astype(np.uint8)
    g32 = np.load(os.path.join(mg.GOLD, which + ".npz"))
    assert (r32["a_id_src"] == g32["id_src"]).all() and (r32["a_src_pts"] == g32["src_pts"]).all(), "the float32 re-run is not the golden"
    path = os.path.join(mg.GOLD, which + "_ref32_tiles.npz")
    tb, tn = tile_b.astype(np.int64), tile_n.astype(np.int64)
    np.savez_compressed(path, valid=np.packbits(valid32[tb, tn], axis=-1), idx=idx32[tb, tn], sim_avg=r32["sim_avg"].astype(np.float32))
    print(which, "ref32 tiles written:", os.path.getsize(path) / 1e6, "MB", flush=True)


if __name__ == "__main__":
    args = sys.argv[1:] or ["e2e_cfg2"]
    phases = [a for a in args if a in ("run", "records")] or ["run", "records"]
    for w in [a for a in args if a not in ("run", "records")]:
        if "run" in phases:
            run_phase(w)
        if "records" in phases:
            records_phase(w)
