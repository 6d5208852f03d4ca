"""Template matching on MI355X behind the reference's `LocalSimilarity` interface.

Drop-in for `src.models.matching.LocalSimilarity` (reference matching.py:9-316; Hydra target
in configs/model/large.yaml:35-39): same constructor kwargs, `.k`, and
`test(src_feats, tar_feat, src_masks, tar_mask, max_batch_size=None)` returning a
PandasTensorCollection with id_src / score_src / score_pts / tar_pts / src_pts.

All arithmetic runs in libgigapose_hip.so (gp_l2norm_cp, gp_match_tiles, gp_topk,
gp_gather_records, gp_format_points); torch only owns the buffers.  The fast path,
`test_bank`, matches detections against a RESIDENT bank indexed by label inside the kernel,
so the reference's 170 MB/detection gather `ae_features[label-1]` (gigaPose.py:520) and its
per-batch re-normalisation of the bank (matching.py:229) disappear.
"""
import os

import pandas as pd
import torch

from . import _lib
from .tensor_collection import PandasTensorCollection

P = 256


def patch_grid_mask(mask224):
    """F.interpolate(mask, size=(16,16)) nearest (matching.py:222,227) == sample pixel (14i,14j)."""
    assert mask224.shape[-1] % 16 == 0 and mask224.shape[-2] % 16 == 0
    sy, sx = mask224.shape[-2] // 16, mask224.shape[-1] // 16
    m = mask224[..., ::sy, ::sx].to(torch.float32)
    return m.reshape(*m.shape[:-2], P).contiguous()


default_numerics = _lib.default_numerics   # "split" unless GIGAPOSE_NUMERICS=chain (the verification mode)


def _l2norm_split(x, hi, lo, rows, C):
    """gp_l2norm_split_mask without a mask image: the split normalisation alone."""
    _lib.call("gp_l2norm_split_mask", _lib.ptr(x), _lib.ptr(hi), _lib.ptr(lo), _lib.i(rows), _lib.i(C), _lib.ptr(None), _lib.i(0), _lib.i(0),
              _lib.ptr(None), _lib.stream_ptr())


def normalize_split(feats, mask224=None):
    """(rows, C, 256) f32 -> matcher-normalised, x32, split into f16 planes (hi, lo), each (rows, 256, Cp) with
    Cp = round_up(C, 32) (zero padded: the split matcher consumes 32 channels per step).  With mask224 (rows, H, W) f32 the same
    launch also writes the rows' patch masks (rows, 256) = patch_grid_mask(mask224), returned third."""
    rows, C = feats.shape[:2]
    x = feats.reshape(rows, C, P).contiguous().float()
    hi = torch.empty(rows, P, (C + 31) // 32 * 32, dtype=torch.float16, device=x.device)
    lo = torch.empty_like(hi)
    if mask224 is None:
        _l2norm_split(x, hi, lo, rows, C)
        return hi, lo
    fused = (mask224.dtype == torch.float32 and mask224.is_contiguous() and mask224.dim() == 3 and mask224.shape[0] == rows
             and mask224.shape[1] % 16 == 0 and mask224.shape[2] % 16 == 0)
    if not fused:   # another dtype / layout: the strided copy
        _l2norm_split(x, hi, lo, rows, C)
        return hi, lo, patch_grid_mask(mask224)
    qmask = torch.empty(rows, P, dtype=torch.float32, device=x.device)
    _lib.call("gp_l2norm_split_mask", _lib.ptr(x), _lib.ptr(hi), _lib.ptr(lo), _lib.i(rows), _lib.i(C), _lib.ptr(mask224),
              _lib.i(mask224.shape[1]), _lib.i(mask224.shape[2]), _lib.ptr(qmask), _lib.stream_ptr())
    return hi, lo, qmask


class MatchBank:
    """Matcher-ready template bank: twice-normalised features (AENet's F.normalize, then the
    matcher's own, matching.py:229) and patch masks (O,N,256).  numerics "chain": features (O,N,C,256) f32;
    "split": f16 planes hi / lo (O,N,256,C) of the same normalised values x 32 (same bytes per template)."""

    def __init__(self, ae_features, masks224, numerics=None, bank_dtype=None):
        """bank_dtype "f16" (split numerics only; env GIGAPOSE_BANK_DTYPE): keep only the f16 hi plane of the templates --
        BASELINE config 5's "fp16 feature bank resident in HBM": half the bytes (0.5 MB per template at C = 1024) and two of
        the three MFMA products; the template features are then f16-rounded (11 bits), which moves near-tied patch argmaxes
        (measured rate: DESIGN.md section 2).  The query keeps both planes."""
        self.numerics = numerics or default_numerics()
        self.bank_dtype = bank_dtype or os.environ.get("GIGAPOSE_BANK_DTYPE", "f32")
        if self.bank_dtype not in ("f32", "f16") or (self.bank_dtype == "f16" and self.numerics != "split"):
            raise ValueError("bank_dtype must be 'f32' or 'f16' ('f16' needs numerics='split')")
        O, N, C = ae_features.shape[:3]
        feats = ae_features.reshape(O * N, C, P).contiguous().float()
        self.features = self.hi = self.lo = None
        if self.numerics == "split":
            hi, lo = normalize_split(feats)
            self.hi = hi.view(O, N, P, -1)
            self.lo = lo.view(O, N, P, -1) if self.bank_dtype == "f32" else None
        else:
            self.features = torch.empty_like(feats)
            _lib.call("gp_l2norm_cp", _lib.ptr(feats), _lib.ptr(self.features), _lib.i(O * N), _lib.i(C),
                      _lib.stream_ptr())
            self.features = self.features.view(O, N, C, P)
        self.masks = patch_grid_mask(masks224)
        self.O, self.N, self.C = O, N, C


class LocalSimilarity(torch.nn.Module):
    def __init__(self, k, sim_threshold, patch_threshold, search_direction="tar2src",
                 image_size=224, patch_size=14, max_batch_size=32):
        super().__init__()
        if search_direction not in ("tar2src", "src2tar"):
            # the reference would fail later with an UnboundLocalError (matching.py:239-247: neither branch assigns)
            raise ValueError(f"search_direction must be 'tar2src' or 'src2tar', got {search_direction!r}")
        self.max_batch_size = max_batch_size
        self.k = k
        self.sim_threshold = sim_threshold
        self.patch_threshold = patch_threshold
        self.search_direction = search_direction
        self.num_patches = image_size // patch_size
        self.numerics = default_numerics()
        self.bank_dtype = None  # None: MatchBank's default (env GIGAPOSE_BANK_DTYPE or "f32"); "f16": hi-plane-only bank
        # patch_threshold <= 0: no cycle check (reference matching.py:256-257) -- the kernels take it as is
        if self.num_patches != 16:
            raise NotImplementedError("kernels are specialised for a 16x16 patch grid (224/14)")

    # ---- kernel-level stages (also used by the template-sharded multi-GPU path) -----------
    def normalize(self, feats):
        """(rows, C, 16, 16) or (rows, C, 256) -> matcher-normalised (rows, C, 256) f32 ("chain") or the
        (hi, lo) f16 planes (rows, 256, C) of the split numerics."""
        if self.numerics == "split":
            return normalize_split(feats.reshape(feats.shape[0], feats.shape[1], P))
        rows, C = feats.shape[:2]
        x = feats.reshape(rows, C, P).contiguous().float()
        out = torch.empty_like(x)
        _lib.call("gp_l2norm_cp", _lib.ptr(x), _lib.ptr(out), _lib.i(rows), _lib.i(C), _lib.stream_ptr())
        return out

    def match_tiles(self, query, qmask, bank, labels0, search_direction=None):
        """All (detection, template) tiles.  query (B,C,256) normalised, qmask (B,256),
        bank: MatchBank, labels0 (B,) int32 0-based.  Returns idx_t2s u8, score_t2s, mask_all
        (B,N,256) and sim_avg (B,N)."""
        split = isinstance(query, (tuple, list))
        if split != (getattr(bank, "numerics", "chain") == "split"):
            raise ValueError(f"query numerics and bank numerics ({bank.numerics}) differ")
        B, C = (query[0].shape[0], query[0].shape[2]) if split else query.shape[:2]
        N = bank.N
        dev = qmask.device
        idx = torch.empty(B, N, P, dtype=torch.uint8, device=dev)
        sc = torch.empty(B, N, P, dtype=torch.float32, device=dev)
        ma = torch.empty(B, N, P, dtype=torch.float32, device=dev)
        avg = torch.empty(B, N, dtype=torch.float32, device=dev)
        direction = 1 if (search_direction or self.search_direction) == "src2tar" else 0   # reference matching.py:239-244
        if split:
            _lib.call("gp_match_tiles_split_dir", _lib.ptr(query[0]), _lib.ptr(query[1]), _lib.ptr(bank.hi), _lib.ptr(bank.lo),
                      _lib.ptr(qmask), _lib.ptr(bank.masks), _lib.ptr(labels0), _lib.i(B), _lib.i(bank.O), _lib.i(N),
                      _lib.i(C), _lib.f(self.sim_threshold), _lib.f(self.patch_threshold), _lib.i(direction), _lib.ptr(idx),
                      _lib.ptr(sc), _lib.ptr(ma), _lib.ptr(avg), _lib.stream_ptr())
            return idx, sc, ma, avg
        _lib.call("gp_match_tiles_dir", _lib.ptr(query), _lib.ptr(bank.features), _lib.ptr(qmask),
                  _lib.ptr(bank.masks), _lib.ptr(labels0), _lib.i(B), _lib.i(bank.O), _lib.i(N), _lib.i(C),
                  _lib.f(self.sim_threshold), _lib.f(self.patch_threshold), _lib.i(direction), _lib.ptr(idx), _lib.ptr(sc),
                  _lib.ptr(ma), _lib.ptr(avg), _lib.stream_ptr())
        return idx, sc, ma, avg

    def topk(self, sim_avg, k=None):
        k = self.k if k is None else k
        B, N = sim_avg.shape
        if k > N:
            raise RuntimeError("selected index k out of range")  # torch.topk's error (matching.py:279)
        ids = torch.empty(B, k, dtype=torch.int32, device=sim_avg.device)
        scores = torch.empty(B, k, dtype=torch.float32, device=sim_avg.device)
        _lib.call("gp_topk", _lib.ptr(sim_avg), _lib.i(B), _lib.i(N), _lib.i(k), _lib.ptr(ids),
                  _lib.ptr(scores), _lib.stream_ptr())
        return ids, scores

    def gather_records(self, ids, idx, sc, ma):
        B, k = ids.shape
        N = idx.shape[1]
        dev = ids.device
        rec_idx = torch.empty(B, k, P, dtype=torch.uint8, device=dev)
        rec_score = torch.empty(B, k, P, dtype=torch.float32, device=dev)
        rec_mask = torch.empty(B, k, P, dtype=torch.float32, device=dev)
        _lib.call("gp_gather_records", _lib.ptr(ids), _lib.ptr(idx), _lib.ptr(sc), _lib.ptr(ma), _lib.i(B),
                  _lib.i(N), _lib.i(k), _lib.ptr(rec_idx), _lib.ptr(rec_score), _lib.ptr(rec_mask),
                  _lib.stream_ptr())
        return rec_idx, rec_score, rec_mask

    def format_points(self, rec_idx, rec_mask):
        B, k = rec_idx.shape[:2]
        tar_pts = torch.empty(B, k, P, 2, dtype=torch.int64, device=rec_idx.device)
        src_pts = torch.empty(B, k, P, 2, dtype=torch.int64, device=rec_idx.device)
        _lib.call("gp_format_points", _lib.ptr(rec_idx), _lib.ptr(rec_mask), _lib.i(B * k), _lib.ptr(tar_pts),
                  _lib.ptr(src_pts), _lib.stream_ptr())
        return tar_pts, src_pts

    def select_topk(self, sim_avg, idx, sc, ma, k=None):
        """topk + gather_records + format_points (reference matching.py:279-316) as ONE launch (gp_select_topk): ids (B,k) int64,
        score_src (B,k), score_pts (B,k,256), tar_pts / src_pts (B,k,256,2) int64.  Equal to the three stage calls (tests)."""
        k = self.k if k is None else k
        B, N = sim_avg.shape
        if k > N:
            raise RuntimeError("selected index k out of range")  # torch.topk's error (matching.py:279)
        dev = sim_avg.device
        ids = torch.empty(B, k, dtype=torch.int64, device=dev)
        scores = torch.empty(B, k, dtype=torch.float32, device=dev)
        rec_score = torch.empty(B, k, P, dtype=torch.float32, device=dev)
        tar_pts = torch.empty(B, k, P, 2, dtype=torch.int64, device=dev)
        src_pts = torch.empty(B, k, P, 2, dtype=torch.int64, device=dev)
        _lib.call("gp_select_topk", _lib.ptr(sim_avg), _lib.ptr(idx), _lib.ptr(sc), _lib.ptr(ma), _lib.i(B), _lib.i(N), _lib.i(k), _lib.ptr(ids),
                  _lib.ptr(scores), _lib.ptr(rec_score), _lib.ptr(tar_pts), _lib.ptr(src_pts), _lib.stream_ptr())
        return ids, scores, rec_score, tar_pts, src_pts

    # ---- resident-bank entry point (what GigaPose.eval_retrieval uses) --------------------
    def test_bank(self, bank, tar_feat, tar_mask, labels0):
        """tar_feat (B,C,16,16) AENet features, tar_mask (B,224,224), labels0 (B,) 0-based."""
        if self.numerics == "split":   # normalise + split + the patch masks in ONE launch
            hi, lo, qmask = normalize_split(tar_feat.reshape(tar_feat.shape[0], tar_feat.shape[1], P), tar_mask)
            query = (hi, lo)
        else:
            query = self.normalize(tar_feat)
            qmask = patch_grid_mask(tar_mask)
        idx, sc, ma, avg = self.match_tiles(query, qmask, bank, labels0.to(torch.int32).contiguous())
        ids, score_src, rec_score, tar_pts, src_pts = self.select_topk(avg, idx, sc, ma)
        return PandasTensorCollection(infos=pd.DataFrame(), id_src=ids, score_src=score_src,
                                      score_pts=rec_score, tar_pts=tar_pts, src_pts=src_pts)

    # ---- validation-time matcher (reference matching.py:115-186) ---------------------------
    def val(self, src_feat, tar_feat, src_mask, tar_mask):
        """src_feat / tar_feat (B,C,16,16), masks (B,224,224): detection b against ITS OWN template -- the same
        fused tile kernel with one template per 'object'.  Returns src_pts, tar_pts (B,256,2) int64 (-1 = invalid)
        and score (B,256) = the raw best similarity per query patch."""
        if self.patch_threshold <= 0:
            raise ValueError("patch_threshold must be greater than 0")                      # reference matching.py:158-159
        B = tar_feat.shape[0]
        dev = tar_feat.device
        bank = MatchBank(src_feat.unsqueeze(1), src_mask.unsqueeze(1), self.numerics)      # O = B objects x N = 1
        labels0 = torch.arange(B, dtype=torch.int32, device=dev)
        # `val` always searches tar2src (matching.py:147-148), whatever search_direction the metric was built with
        idx, sc, ma, _ = self.match_tiles(self.normalize(tar_feat), patch_grid_mask(tar_mask), bank, labels0, "tar2src")
        tar_pts, src_pts = self.format_points(idx.contiguous(), ma.contiguous())            # (B,1,256,2)
        return PandasTensorCollection(infos=pd.DataFrame(), src_pts=src_pts[:, 0], tar_pts=tar_pts[:, 0], score=sc[:, 0])

    # ---- reference-signature entry point ---------------------------------------------------
    def test(self, src_feats, tar_feat, src_masks, tar_mask, max_batch_size=None):
        """Reference signature (matching.py:188): src_feats (B,N,C,H,W) is a per-detection
        gathered bank.  Each detection becomes its own 'object' of a temporary bank; chunking by
        max_batch_size only bounds that temporary (results do not depend on it)."""
        if max_batch_size is None:
            max_batch_size = self.max_batch_size
        B = tar_feat.shape[0]
        outs = []
        for s in range(0, B, max_batch_size):
            e = min(B, s + max_batch_size)
            bank = MatchBank(src_feats[s:e], src_masks[s:e], self.numerics)
            labels0 = torch.arange(e - s, dtype=torch.int32, device=tar_feat.device)
            outs.append(self.test_bank(bank, tar_feat[s:e], tar_mask[s:e], labels0))
        out = outs[0]
        for o in outs[1:]:
            out.cat_df(o)
        return PandasTensorCollection(infos=pd.DataFrame(), **out.tensors)
