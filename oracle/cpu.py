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


def match(query, bank, qmask, bmask, labels, thr=0.5, patch_thr=3.0):
    """LocalSimilarity.test steps 3-8 (matching.py:233-278) on matcher-normalised features.

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
    lib().oracle_match(_p(query), _p(bank), _p(qmask), _p(bmask), _p(labels),
                       ctypes.c_int(B), ctypes.c_int(O), ctypes.c_int(N), ctypes.c_int(C),
                       ctypes.c_float(thr), ctypes.c_float(patch_thr),
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
                          thr=0.5, patch_thr=3.0):
    """Whole LocalSimilarity.test (matching.py:188-316) against a resident bank.

    src_feats (O,N,C,16,16) AENet-normalised bank; tar_feat (B,C,16,16); masks at 224x224.
    Equivalent to the reference call with src_feats[labels] gathered (gigaPose.py:520-531).
    """
    O, N, C = src_feats.shape[:3]
    B = tar_feat.shape[0]
    q = l2norm_cp(np.asarray(tar_feat).reshape(B, C, P))
    bank = l2norm_cp(np.asarray(src_feats).reshape(O, N, C, P))
    idx, sc, ma, avg = match(q, bank, patch_mask(tar_mask224), patch_mask(src_masks224), labels,
                             thr, patch_thr)
    ids, score_src = topk(avg, k)
    score_pts, tar_pts, src_pts = gather_format(ids, idx, sc, ma)
    return dict(id_src=ids.astype(np.int64), score_src=score_src, score_pts=score_pts,
                tar_pts=tar_pts, src_pts=src_pts, sim_avg=avg, idx_t2s=idx, score_t2s=sc,
                mask_all=ma)


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
