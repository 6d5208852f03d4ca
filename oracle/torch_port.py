"""TEST / BENCH INFRASTRUCTURE ONLY -- a torch-CPU restatement of what the reference's hot loop EXECUTES
(`GigaPose.eval_retrieval`, reference src/models/gigaPose.py:481-633), used by bench.py's `cpu_baseline` leg (kind
"port": the Python reference itself cannot travel to the GPU box) and pinned by tests/test_oracle_torch_port.py.

Unlike oracle/gp_oracle.c (which restates the ARITHMETIC in a fixed order, as the parity checker), this file restates
the reference's COST PROFILE: the same torch operators on the same tensor shapes in the same order, including the
work the reference does redundantly --
  * sub-batches of `max_num_dets_per_forward` = 4 detections (configs/test.yaml:21; gigaPose.py:501-509);
  * the 170 MB / detection bank gather `ae_features[label - 1]` (gigaPose.py:520-521);
  * three F.normalize passes, the materialised (b, N, 256, 256) similarity tensor and ~25 elementwise passes over it
    (matching.py:222-278);
  * the IST backbone recomputed k times (gigaPose.py:553);
RANSAC / recovery use the C oracle (faster than the reference's Python loops: it flatters the baseline slightly).
Allowed importers: tests/, bench.py (cpu_baseline).  Never imported by gigapose_amd/.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import cpu as oracle
from . import ist_torch


def vit_features(hf_model, images, chunk=64):
    """AENet.forward_by_chunk (ae_net.py:55-73): x_prenorm[:, 1:] -> (b, C, 16, 16), F.normalize(dim=1)."""
    outs = []
    for s in range(0, images.shape[0], chunk):
        h = hf_model(pixel_values=images[s:s + chunk], output_hidden_states=True).hidden_states[-1][:, 1:, :]
        b, p, c = h.shape
        outs.append(F.normalize(h.permute(0, 2, 1).reshape(b, c, 16, 16), dim=1))
    return torch.cat(outs)


def local_similarity_test(src_feats, tar_feat, src_masks, tar_mask, k, sim_threshold=0.5, patch_threshold=3, max_batch_size=32):
    """matching.py:188-316 operator for operator (search_direction 'tar2src').  src_feats (B,N,C,16,16) gathered bank."""
    outs = {n: [] for n in ["id_src", "score_src", "score_pts", "tar_pts", "src_pts"]}
    G = 16
    for s in range(0, tar_feat.shape[0], max_batch_size):
        sf, tf = src_feats[s:s + max_batch_size], tar_feat[s:s + max_batch_size]
        sm, tm = src_masks[s:s + max_batch_size], tar_mask[s:s + max_batch_size]
        B, N = sm.shape[:2]
        tm = F.interpolate(tm.unsqueeze(1), size=(G, G)).reshape(B, G * G)
        tf = F.normalize(tf, dim=1).reshape(B, -1, G * G)
        sm = F.interpolate(sm, size=(G, G)).reshape(B, N, G * G)
        sf = F.normalize(sf, dim=2).reshape(B, N, -1, G * G)
        sim = torch.einsum("b c t, b n c s -> b n t s", tf, sf)
        sim *= sm[:, :, None, :]
        sim *= tm[:, None, :, None]
        sim[sim < sim_threshold] = 0
        score_t2s, idx_t2s = torch.max(sim, dim=3)
        score_s2t, idx_s2t = torch.max(sim, dim=2)
        mask_sim = score_t2s >= sim_threshold
        # find_consistency_patches (matching.py:80-113)
        back = torch.gather(idx_s2t, 2, idx_t2s)
        loc = torch.stack([torch.arange(G * G) % G, torch.arange(G * G) // G], dim=1).float()
        dist = torch.norm(loc[back] - loc[None, None, :, :], dim=3)
        mask_cycle = (dist <= patch_threshold) & (torch.gather(score_s2t, 2, idx_t2s) >= sim_threshold)
        mask_nz = tm[:, None, :].expand(B, N, -1) * torch.gather(sm, 2, idx_t2s) * (idx_s2t != 0) * (idx_t2s != 0)
        mask_all = mask_sim * mask_cycle * mask_nz
        has = mask_all.sum(dim=2) > 0
        sim_avg = torch.zeros(B, N)
        sim_avg[has] = torch.sum(score_t2s * mask_all, dim=2)[has] / (G * G)
        score_src, id_src = torch.topk(sim_avg, k, dim=1)
        rows = torch.arange(B)[:, None].expand(B, k)
        m = mask_all[rows, id_src, :]
        idx = idx_t2s[rows, id_src, :]
        src_xy = torch.stack([idx % G, idx // G], dim=-1)
        tar_xy = torch.stack([torch.arange(G * G) % G, torch.arange(G * G) // G], dim=-1)[None, None].expand(B, k, -1, -1)
        valid = (m > 0)[..., None]
        outs["id_src"].append(id_src)
        outs["score_src"].append(score_src)
        outs["score_pts"].append(score_t2s[rows, id_src, :])
        outs["tar_pts"].append(torch.where(valid, tar_xy, torch.full_like(tar_xy, -1)))
        outs["src_pts"].append(torch.where(valid, src_xy, torch.full_like(src_xy, -1)))
    return {n: torch.cat(v) for n, v in outs.items()}


def ist_inference(ist_net, src_feat, tar_feat, src_pts, tar_pts):
    """ISTNet.inference (ist_net.py:97-120): gather + concat + the two torch MLP heads."""
    B, P = src_pts.shape[:2]
    valid = (src_pts[..., 0] != -1) & (src_pts[..., 1] != -1)
    si = (src_pts[..., 1] * 16 + src_pts[..., 0]).clamp(min=0)
    ti = (tar_pts[..., 1] * 16 + tar_pts[..., 0]).clamp(min=0)
    sfl, tfl = src_feat.reshape(B, 256, 256), tar_feat.reshape(B, 256, 256)
    s = torch.gather(sfl, 2, si[:, None, :].expand(B, 256, P)).permute(0, 2, 1)[valid]
    t = torch.gather(tfl, 2, ti[:, None, :].expand(B, 256, P)).permute(0, 2, 1)[valid]
    feats = torch.cat([t, s], dim=1)
    scales = torch.full((B, P), -1000.0)
    cos_sin = torch.full((B, P, 2), -1000.0)
    scales[valid] = ist_net.regressor.scale_predictor(feats).squeeze(1)
    cos_sin[valid] = ist_net.regressor.inplane_predictor(feats)
    return scales, cos_sin


@torch.no_grad()
def eval_retrieval(hf_vit, ist_net, bank_ae, bank_ist, bank_masks, tmpl_geom, crops, k, dets_per_forward=4):
    """One image's worth of the reference hot loop on CPU.  bank_ae (O,N,C,16,16), bank_ist (O,N,256,16,16), bank_masks
    (O,N,224,224), tmpl_geom = (K (O,3,3), M (O,N,3,3), poses (O,N,4,4)); crops: dict tar_img, tar_mask, tar_K, tar_M,
    labels (1-based).  Returns poses (B,k,4,4) and the matcher outputs."""
    tar_img, tar_mask, labels = crops["tar_img"], crops["tar_mask"], crops["labels"].long()
    B = tar_img.shape[0]
    parts = []
    for s in range(0, B, dets_per_forward):                                   # HOT LOOP 1 (gigaPose.py:511-536)
        sl = slice(s, s + dets_per_forward)
        tar_ae = vit_features(hf_vit, tar_img[sl])
        src_ae = bank_ae[labels[sl] - 1]                                      # the 170 MB / detection gather
        src_m = bank_masks[labels[sl] - 1]
        parts.append(local_similarity_test(src_ae, tar_ae, src_m, tar_mask[sl], k))
    pred = {n: torch.cat([p[n] for p in parts]) for n in parts[0]}
    rel_scale, rel_inplane = [], []
    for j in range(k):                                                        # HOT LOOP 2 (gigaPose.py:545-575)
        src_ist = bank_ist[labels - 1, pred["id_src"][:, j]]
        tar_ist = ist_torch.resnet_forward(ist_net.backbone, tar_img)                 # recomputed k times, as the reference does
        sc, cs = ist_inference(ist_net, src_ist, tar_ist, pred["src_pts"][:, j], pred["tar_pts"][:, j])
        rel_scale.append(sc)
        rel_inplane.append(cs)
    rel_scale, rel_inplane = torch.stack(rel_scale, 1), torch.stack(rel_inplane, 1)
    M, failed, isrc, itar, isc = oracle.ransac(pred["src_pts"].numpy(), pred["tar_pts"].numpy(), rel_scale.numpy(), rel_inplane.numpy())
    score = isc.sum(-1) / 256.0
    order = np.argsort(-score, axis=1, kind="stable")
    rows = np.arange(B)[:, None]
    tK, tM, tP = tmpl_geom
    poses = oracle.recover((labels - 1).numpy().astype(np.int32), crops["tar_K"].numpy(), crops["tar_M"].numpy(),
                           pred["id_src"].numpy()[rows, order], M[rows, order], tK, tM, tP)
    return poses, pred
