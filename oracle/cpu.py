"""ctypes/numpy front-end of the C oracle (oracle/gp_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Allowed importers: tests/, __graft_entry__.smoke(), bench.py (cpu_baseline leg).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgp_oracle.so")
_lib = None

P = 256


def build(force=False):
    """Compile the C oracle (gcc; seconds)."""
    src = os.path.join(_HERE, "gp_oracle.c")
    if force or (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        assert _lib.oracle_abi_version() >= 1
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def l2norm_cp(x):
    """F.normalize over C of (..., C, P) channel-major features (ae_net.py:69, matching.py:224)."""
    x = _f32(x)
    C = x.shape[-2]
    assert x.shape[-1] == P
    rows = int(np.prod(x.shape[:-2])) if x.ndim > 2 else 1
    out = np.empty_like(x)
    lib().oracle_l2norm_cp(_p(x), _p(out), ctypes.c_int(rows), ctypes.c_int(C))
    return out


def patch_mask(mask224):
    """F.interpolate(mask, size=(16,16)) nearest (matching.py:222,227): samples pixel (14i,14j)."""
    m = np.asarray(mask224, dtype=np.float32)
    return np.ascontiguousarray(m[..., ::14, ::14]).reshape(*m.shape[:-2], P)


def match(query, bank, qmask, bmask, labels, thr=0.5, patch_thr=3.0, search_direction="tar2src"):
    """LocalSimilarity.test steps 3-8 (matching.py:233-278) on matcher-normalised features; search_direction as the reference's
    ctor argument (matching.py:239-244), patch_thr <= 0 = no cycle check (matching.py:256-257).

    query (B,C,P), bank (O,N,C,P), qmask (B,P), bmask (O,N,P), labels (B,) 0-based.
    Returns idx_t2s u8 (B,N,P), score_t2s f32 (B,N,P), mask_all f32 (B,N,P), sim_avg f32 (B,N).
    """
    query, bank, qmask, bmask = _f32(query), _f32(bank), _f32(qmask), _f32(bmask)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    B, C, _ = query.shape
    O, N = bank.shape[:2]
    assert bank.shape == (O, N, C, P) and qmask.shape == (B, P) and bmask.shape == (O, N, P)
    assert labels.min() >= 0 and labels.max() < O
    idx = np.empty((B, N, P), np.uint8)
    sc = np.empty((B, N, P), np.float32)
    ma = np.empty((B, N, P), np.float32)
    avg = np.empty((B, N), np.float32)
    assert search_direction in ("tar2src", "src2tar")
    lib().oracle_match_dir(_p(query), _p(bank), _p(qmask), _p(bmask), _p(labels),
                           ctypes.c_int(B), ctypes.c_int(O), ctypes.c_int(N), ctypes.c_int(C),
                           ctypes.c_float(thr), ctypes.c_float(patch_thr), ctypes.c_int(int(search_direction == "src2tar")),
                           _p(idx), _p(sc), _p(ma), _p(avg))
    return idx, sc, ma, avg


def topk(sim_avg, k):
    sim_avg = _f32(sim_avg)
    B, N = sim_avg.shape
    assert k <= N, "topk requires N >= k (matching.py:279 raises too)"
    ids = np.empty((B, k), np.int32)
    sc = np.empty((B, k), np.float32)
    lib().oracle_topk(_p(sim_avg), ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(k), _p(ids), _p(sc))
    return ids, sc


def gather_format(ids, idx_t2s, score_t2s, mask_all):
    ids = np.ascontiguousarray(ids, np.int32)
    B, k = ids.shape
    N = idx_t2s.shape[1]
    score_pts = np.empty((B, k, P), np.float32)
    tar_pts = np.empty((B, k, P, 2), np.int64)
    src_pts = np.empty((B, k, P, 2), np.int64)
    lib().oracle_gather_format(_p(ids), _p(np.ascontiguousarray(idx_t2s)), _p(_f32(score_t2s)),
                               _p(_f32(mask_all)), ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(k),
                               _p(score_pts), _p(tar_pts), _p(src_pts))
    return score_pts, tar_pts, src_pts


def local_similarity_test(src_feats, tar_feat, src_masks224, tar_mask224, labels, k,
                          thr=0.5, patch_thr=3.0, search_direction="tar2src"):
    """Whole LocalSimilarity.test (matching.py:188-316) against a resident bank.

    src_feats (O,N,C,16,16) AENet-normalised bank; tar_feat (B,C,16,16); masks at 224x224.
    Equivalent to the reference call with src_feats[labels] gathered (gigaPose.py:520-531).
    """
    O, N, C = src_feats.shape[:3]
    B = tar_feat.shape[0]
    q = l2norm_cp(np.asarray(tar_feat).reshape(B, C, P))
    bank = l2norm_cp(np.asarray(src_feats).reshape(O, N, C, P))
    idx, sc, ma, avg = match(q, bank, patch_mask(tar_mask224), patch_mask(src_masks224), labels,
                             thr, patch_thr, search_direction)
    ids, score_src = topk(avg, k)
    score_pts, tar_pts, src_pts = gather_format(ids, idx, sc, ma)
    return dict(id_src=ids.astype(np.int64), score_src=score_src, score_pts=score_pts,
                tar_pts=tar_pts, src_pts=src_pts, sim_avg=avg, idx_t2s=idx, score_t2s=sc,
                mask_all=ma)


def local_similarity_val(src_feat, tar_feat, src_mask224, tar_mask224, thr=0.5, patch_thr=3.0):
    """LocalSimilarity.val (matching.py:115-186): detection b against ITS OWN template src_feat[b] -- the same tile
    arithmetic as `test` with N = 1; returns the reference's src_pts / tar_pts (B,256,2) and score (B,256)."""
    B, C = tar_feat.shape[:2]
    q = l2norm_cp(np.asarray(tar_feat).reshape(B, C, P))
    bank = l2norm_cp(np.asarray(src_feat).reshape(B, 1, C, P))
    idx, sc, ma, _ = match(q, bank, patch_mask(tar_mask224), patch_mask(src_mask224).reshape(B, 1, P),
                           np.arange(B, dtype=np.int32), thr, patch_thr)
    _, tar_pts, src_pts = gather_format(np.zeros((B, 1), np.int32), idx, sc, ma)
    return dict(src_pts=src_pts[:, 0], tar_pts=tar_pts[:, 0], score=sc[:, 0])


def gemm_kmajor(A, B, epi=0, bias=None, scale=None, res=None):
    """D[i][j] = epi(sum_k A[k][i] B[k][j]) with the sequential-fmaf order of gp_gemm.hip."""
    A, B = _f32(A), _f32(B)
    K, I = A.shape
    K2, J = B.shape
    assert K == K2
    D = np.empty((I, J), np.float32)
    bias = None if bias is None else _f32(bias)
    scale = None if scale is None else _f32(scale)
    res = None if res is None else _f32(res)
    lib().oracle_gemm_kmajor(_p(A), ctypes.c_int(I), _p(B), ctypes.c_int(J), _p(D), ctypes.c_int(J),
                             ctypes.c_int(I), ctypes.c_int(J), ctypes.c_int(K), ctypes.c_int(epi),
                             _p(bias) if bias is not None else None, _p(scale) if scale is not None else None,
                             _p(res) if res is not None else None, ctypes.c_int(J))
    return D


def gather_rows(feat, pts):
    """batch.py:46-73 `gather` WITHOUT the compaction: feat (B,D,256), pts (B,...,256,2) ->
    rows (B,...,256, D) and valid (B,...,256).  Invalid rows take index 0."""
    feat = _f32(feat)
    pts = np.asarray(pts)
    x, y = pts[..., 0], pts[..., 1]
    valid = (x != -1) & (y != -1)
    idx = np.where(valid, y * 16 + x, 0)
    return idx, valid


def ist_inference(tar_feat, src_feat_sel, tar_pts, src_pts, weights, use_tanh=True):
    """ISTNet.inference (ist_net.py:97-120) for all hypotheses at once.

    tar_feat (B,D,256); src_feat_sel (B,k,D,256) = ist bank rows of the selected templates;
    tar_pts/src_pts (B,k,256,2); weights: dict scale/inplane -> [W1,b1,W2,b2,W3,b3] torch layout.
    Returns scales (B,k,256), cos_sin (B,k,256,2)."""
    B, k = src_pts.shape[:2]
    D = tar_feat.shape[1]
    ti, tv = gather_rows(tar_feat, tar_pts)
    si, sv = gather_rows(tar_feat, src_pts)
    tar_feat, src_feat_sel = _f32(tar_feat), _f32(src_feat_sel)
    R = B * k * P
    feats = np.empty((B, k, P, 2 * D), np.float32)
    for b in range(B):
        for j in range(k):
            feats[b, j, :, :D] = tar_feat[b][:, ti[b, j]].T
            feats[b, j, :, D:] = src_feat_sel[b, j][:, si[b, j]].T
    valid = np.ascontiguousarray((tv & sv).reshape(R).astype(np.uint8))
    feats = feats.reshape(R, 2 * D)
    outs = []
    for name, nout, th in (("scale", 1, 0), ("inplane", 2, 1 if use_tanh else 0)):
        W1, b1, W2, b2, W3, b3 = [_f32(w) for w in weights[name]]
        out = np.empty((R, nout), np.float32)
        lib().oracle_ist_head(_p(feats), _p(valid), ctypes.c_int(R), ctypes.c_int(2 * D),
                              ctypes.c_int(W1.shape[0]), ctypes.c_int(W2.shape[0]), ctypes.c_int(nout),
                              _p(W1), _p(b1), _p(W2), _p(b2), _p(W3), _p(b3), ctypes.c_int(th), _p(out))
        outs.append(out)
    return outs[0].reshape(B, k, P), outs[1].reshape(B, k, P, 2)


def ransac(src_pts, tar_pts, rel_scale, rel_inplane, patch_size=14.0, thr=14.0, scores=None):
    """RANSAC.forward (ransac.py:108-172) over leading dims (...,256,2); scores (...,256) = its `scores` argument (None: ones)."""
    src_pts = np.ascontiguousarray(src_pts, np.int64)
    tar_pts = np.ascontiguousarray(tar_pts, np.int64)
    lead = src_pts.shape[:-2]
    R = int(np.prod(lead))
    rs, ri = _f32(rel_scale), _f32(rel_inplane)
    M = np.empty((R, 3, 3), np.float32)
    failed = np.empty(R, np.uint8)
    isrc = np.empty((R, P, 2), np.int64)
    itar = np.empty((R, P, 2), np.int64)
    isc = np.empty((R, P), np.int64)
    sc = _f32(scores) if scores is not None else None
    lib().oracle_ransac_scored(_p(src_pts), _p(tar_pts), _p(rs), _p(ri), _p(sc) if sc is not None else None, ctypes.c_int(R),
                               ctypes.c_float(patch_size), ctypes.c_float(thr), _p(M), _p(failed), _p(isrc), _p(itar), _p(isc))
    return (M.reshape(*lead, 3, 3), failed.reshape(lead).astype(bool), isrc.reshape(*lead, P, 2),
            itar.reshape(*lead, P, 2), isc.reshape(*lead, P))


def recover(labels, tar_K, tar_M, id_src, pred_M, tmpl_K, tmpl_M, tmpl_pose):
    """ObjectPoseRecovery.forward_recovery (poses.py:103-122); labels 0-based."""
    labels = np.ascontiguousarray(labels, np.int32)
    id_src = np.ascontiguousarray(id_src, np.int64)
    B, k = id_src.shape
    N = tmpl_M.shape[1]
    out = np.empty((B, k, 4, 4), np.float32)
    lib().oracle_recover(_p(labels), _p(_f32(tar_K)), _p(_f32(tar_M)), _p(id_src), _p(_f32(pred_M)),
                         _p(_f32(tmpl_K)), _p(_f32(tmpl_M)), _p(_f32(tmpl_pose)), ctypes.c_int(B),
                         ctypes.c_int(N), ctypes.c_int(k), _p(out))
    return out


def conv2d_cm(X, W, alpha=None, beta=None, res=None, stride=1, pad=0, relu=False):
    """Channel-major conv + folded BN + residual + ReLU (gp_conv.hip order). X (Cin,B,H,W), W torch layout."""
    X, W = _f32(X), _f32(W)
    Cin, B, H, Wd = X.shape
    Cout, _, KH, KW = W.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (Wd + 2 * pad - KW) // stride + 1
    Y = np.empty((Cout, B, OH, OW), np.float32)
    a = None if alpha is None else _f32(alpha)
    bt = None if beta is None else _f32(beta)
    r = None if res is None else _f32(res)
    lib().oracle_conv2d_cm(_p(X), _p(W), _p(Y), _p(a) if a is not None else None, _p(bt) if bt is not None else None,
                           _p(r) if r is not None else None, ctypes.c_int(Cin), ctypes.c_int(B), ctypes.c_int(H),
                           ctypes.c_int(Wd), ctypes.c_int(Cout), ctypes.c_int(KH), ctypes.c_int(KW), ctypes.c_int(stride),
                           ctypes.c_int(pad), ctypes.c_int(1 if relu else 0))
    return Y
