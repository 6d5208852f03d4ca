"""Equal-or-explained-tie comparison of one run of the hot path with the reference evaluated in float64.

TEST INFRASTRUCTURE (numpy only).  Inputs: the margins golden written by oracle/make_margins.py from the unmodified
reference's float64 run (every decision the path takes, with its float64 margin), and one implementation's results on the
same inputs ("ours": the HIP path on the GPU, or -- in the CPU test of this checker -- the reference's own float32 run).

A difference is EXPLAINED iff it sits on a float64 decision margin below epsilon:
  * a patch whose (valid, matched template patch) differs from the float64 run must depend on a decision -- its row argmax
    (matching.py:240), the 0.5 threshold on its row maximum (:236, :246), the column argmax / threshold of the template patch
    it is matched to in either run (cycle check, :249-255 -> find_consistency_patches), the column of the quirk term
    `idx_src2tar != 0` read at position t (:266) -- whose float64 margin is < eps_sim; if the matched patch differs it must be
    the float64 row's runner-up;
  * sim_avg of every stored tile must equal, within eps_sim, the float64 similarities summed over OUR valid patches; the
    templates we rank in the top k must be consistent with those sums (pairwise, within 2 eps_sim);
  * a hypothesis with identical correspondences must have the float64 run's inlier count and RANSAC winner (ransac.py:37-106)
    unless the candidates involved have correspondences whose error is within eps_px of the 14 px threshold (poses.py:18);
  * wherever every discrete choice equals the float64 run's, M and the pose must agree to the north-star's 1e-4;
  * NO hypothesis is exempt (round 4): one whose correspondences differ from the float64 run's (by explained patch flips), or whose
    template is not among the float64 run's top k at all, cannot be compared with that run downstream -- so RANSAC and the pose
    recovery are RESTATED in float64 on OUR correspondences and OUR IST regressions (ransac_f64 / recover_f64 below, both pinned
    to the reference's float64 goldens by tests/test_parity_explain.py) and our inlier count, winner, failed flag, M and pose must
    equal that restatement (correspondences within eps_px of the 14 px threshold allowed, as above); where the correspondences of
    such a hypothesis coincide with the float64 run's, the IST regressions must agree too.  report["hyp_checked"] counts the
    hypotheses whose pose was checked one way or the other; the tests assert it equals report["hyp"].
Anything else is reported as UNEXPLAINED; the tests assert there is none.
"""
import numpy as np

P = 256
PIXEL_THRESHOLD = 14.0
PATCH = 14.0


def ransac_f64(src_pts, tar_pts, rel_scale, rel_inplane, eps_px):
    """Candidates of one hypothesis in float64 (ransac.py:37-106 restated): returns dict(n, counts (n,), fragile (n,) = pairs
    within eps_px of the threshold, M (n,3,3), winner (first max), failed)."""
    ok = src_pts[:, 0] != -1
    n = int(ok.sum())
    if n == 0:
        return dict(n=0, counts=np.zeros(0, int), fragile=np.zeros(0, int), M=np.eye(3)[None], winner=-1, failed=False)
    s = src_pts[ok].astype(np.float64) * PATCH
    t = tar_pts[ok].astype(np.float64) * PATCH
    sc = rel_scale[ok].astype(np.float64)
    c, sn = rel_inplane[ok, 0].astype(np.float64), rel_inplane[ok, 1].astype(np.float64)
    A = np.zeros((n, 2, 2))
    A[:, 0, 0], A[:, 0, 1], A[:, 1, 0], A[:, 1, 1] = c * sc, -sn * sc, sn * sc, c * sc
    tr = t - np.einsum("nij,nj->ni", A, s)                       # candidate i maps its own source point onto its target
    M = np.tile(np.eye(3), (n, 1, 1))
    M[:, :2, :2], M[:, :2, 2] = A, tr
    proj = np.einsum("nij,mj->nmi", A, s) + tr[:, None, :]       # candidate i applied to every source point
    err = np.linalg.norm(t[None] - proj, axis=2)
    off = ~np.eye(n, dtype=bool)                                 # the proposing correspondence is not its own inlier
    counts = ((err <= PIXEL_THRESHOLD) & off).sum(1)
    fragile = ((np.abs(err - PIXEL_THRESHOLD) < eps_px) & off).sum(1)
    if n == 1:
        return dict(n=1, counts=counts, fragile=fragile, M=M, winner=0, failed=True)
    w = int(np.argmax(counts))
    return dict(n=n, counts=counts, fragile=fragile, M=M, winner=w, failed=bool(counts[w] == 0))


def pose_rel_err(a, b):
    t = np.linalg.norm(a[..., :3, 3] - b[..., :3, 3], axis=-1) / np.linalg.norm(b[..., :3, 3], axis=-1)
    r = np.abs(a[..., :3, :3] - b[..., :3, :3]).max(axis=(-1, -2))
    return t, r


def geometry(seed, O, N, B):
    """The camera / crop geometry of an end-to-end golden (oracle/make_goldens.py: e2e_inputs; tests/test_gpu_e2e.py: e2e_inputs),
    regenerated from its seed: labels (B,) 1-based, tar_K / tar_M (B,3,3), template K (O,3,3), M (O,N,3,3), poses (O,N,4,4)."""
    from gigapose_testing import synthetic as syn

    tK, tM, tP = syn.template_geometry(seed + 1, O, N)
    labels = np.random.RandomState(seed).randint(1, O + 1, B)
    qK, qM = syn.crop_geometry(seed + 2, B)
    return dict(labels=labels, tar_K=qK, tar_M=qM, tmpl_K=tK, tmpl_M=tM, tmpl_pose=tP)


def recover_f64(M, view, tmpl_K, tmpl_M, tmpl_pose, tar_K, tar_M):
    """ObjectPoseRecovery._forward_recovery (reference poses.py:26-100 with lib3d/torch.py: normalize_affine_transform :150-162,
    inverse_affine :44-62) restated in float64 for ONE hypothesis: M (3,3) the RANSAC transform, view the template id, tmpl_K (3,3),
    tmpl_M / tmpl_pose the (N,3,3) / (N,4,4) tables of the detection's object, tar_K / tar_M (3,3) of the crop.  Returns (4,4)."""
    f = np.float64
    M, tK, tMv, qK, qM = M.astype(f), tmpl_K.astype(f), tmpl_M[view].astype(f), tar_K.astype(f), tar_M.astype(f)
    pose = tmpl_pose[view].astype(f).copy()
    Rin = np.zeros((3, 3))
    Rin[2, 2] = 1.0
    Rin[:2, :2] = M[:2, :2] / np.linalg.norm(M[:2, 0])            # in-plane rotation = the transform without its scale
    pose[:3, :3] = Rin @ pose[:3, :3]
    temp_z = pose[2, 3]
    c = tK @ pose[:3, 3]
    c = c / c[2]                                                  # template object centre in template pixels
    sc = qM[0, 0]
    inv_qM = np.eye(3)
    inv_qM[0, 0] = inv_qM[1, 1] = 1.0 / sc
    inv_qM[:2, 2] = -qM[:2, 2] / sc
    aff = inv_qM @ M @ tMv                                        # template crop -> full query image
    qc = aff @ c
    qz = temp_z / np.linalg.norm(aff[:2, 0]) * (qK[0, 0] / tK[0, 0])
    tr = np.linalg.inv(qK) @ qc
    pose[:3, 3] = tr / tr[2] * qz
    return pose


def explain(m, ours, eps_sim=2e-6, eps_px=2e-3, tol_ist=1e-4, tol_pose=1e-4, verbose=False, geom=None):
    """m: dict of the margins golden.  ours: dict with
         tiles_valid (T,P) bool, tiles_idx (T,P) int       -- our records of the T tiles the golden stores (m["tile_b"], m["tile_n"])
         sim_avg (B,N)                                      -- our sim_avg of every tile
         id_src (B,k), src_pts / tar_pts (B,k,P,2), inliers (B,k) int, idx_failed (B,k), relScale (B,k,P), relInplane (B,k,P,2),
         M (B,k,3,3), poses (B,k,4,4)                       -- final hypotheses (sorted as eval_retrieval returns them)
    Returns a report dict; report["unexplained"] is a list of strings (empty = every difference sits on a float64 tie)."""
    clip = float(m["clip"])
    assert eps_sim < clip
    top = m["top_ids"].astype(np.int64)
    B, K = top.shape
    k = ours["id_src"].shape[1]
    A_f = m["sim_avg"]
    A_o = ours["sim_avg"].astype(np.float64)
    N = A_f.shape[1]
    tiles_of = [[] for _ in range(B)]
    for i, b_ in enumerate(m["tile_b"]):
        tiles_of[int(b_)].append(i)
    rep = dict(unexplained=[], tiles=len(m["tile_b"]), tiles_with_flips=0, patch_flips=0, set_diff=0, order_diff=0, hyp=B * k, hyp_common=0,
               hyp_same_corr=0, hyp_same_all=0, corr_flip_hyp=0, inlier_diff=0, winner_diff=0, max_avg_dev=0.0, max_ist_dev=0.0,
               max_M_err=0.0, max_t_err=0.0, max_r_err=0.0, unstored_dev=0, notes=[], hyp_checked=0, hyp_self_checked=0,
               self_winner_ties=0, max_self_M_err=0.0, max_self_t_err=0.0, max_self_r_err=0.0)
    bad = rep["unexplained"].append

    def pose_of(b, n, M):
        o = int(geom["labels"][b]) - 1
        return recover_f64(M, n, geom["tmpl_K"][o], geom["tmpl_M"][o], geom["tmpl_pose"][o], geom["tar_K"][b], geom["tar_M"][b])

    def check_pose_restated(b, jo, n, tag):
        """our pose == the float64 recovery of OUR M (whatever path led to M)"""
        if geom is None:
            return False
        want = pose_of(b, n, ours["M"][b, jo])
        te, re_ = pose_rel_err(ours["poses"][b, jo].astype(np.float64), want)
        rep["max_self_t_err"], rep["max_self_r_err"] = max(rep["max_self_t_err"], float(te)), max(rep["max_self_r_err"], float(re_))
        if not (te < tol_pose and re_ < tol_pose):
            bad(f"det {b} template {n} ({tag}): pose is not the float64 recovery of our own M (translation {te:.2e}, rotation {re_:.2e})")
        return True

    def check_self(b, jo, n, tag):
        """A hypothesis that cannot be aligned with the float64 run: RANSAC + recovery restated in float64 on OUR correspondences
        and OUR regressions; our count / winner / failed flag / M / pose must be that restatement's."""
        rep["hyp_self_checked"] += 1
        r = ransac_f64(ours["src_pts"][b, jo], ours["tar_pts"][b, jo], ours["relScale"][b, jo], ours["relInplane"][b, jo], eps_px)
        c_o, M_o = int(ours["inliers"][b, jo]), ours["M"][b, jo].astype(np.float64)
        failed_o = bool(ours["idx_failed"][b, jo])
        if r["n"] == 0:
            if c_o != 0 or np.abs(M_o - np.eye(3)).max() > 0 or failed_o:
                bad(f"det {b} template {n} ({tag}): no correspondences but count {c_o} / M not identity / failed {failed_o}")
            return check_pose_restated(b, jo, n, tag)
        e = np.abs(r["M"] - M_o[None]).max(axis=(1, 2)) / np.abs(r["M"]).max(axis=(1, 2))
        i_o = int(np.argmin(e))
        rep["max_self_M_err"] = max(rep["max_self_M_err"], float(e[i_o]))
        if not e[i_o] < tol_pose:
            bad(f"det {b} template {n} ({tag}): M is no candidate of the float64 RANSAC on our own correspondences (nearest differs by {e[i_o]:.2e})")
            return False
        cf, fg, w = r["counts"], r["fragile"], r["winner"]
        # candidates with the same transform (many-to-one matches propose identical M) are one candidate: compare by count
        if not (cf[i_o] + fg[i_o] >= (cf - fg).max()):
            bad(f"det {b} template {n} ({tag}): RANSAC winner {i_o} has {cf[i_o]} inliers (+-{fg[i_o]} within {eps_px:g} px of 14) but candidate "
                f"{int(np.argmax(cf - fg))} has at least {int((cf - fg).max())}")
        elif i_o != w and not (np.abs(r["M"][w] - r["M"][i_o]).max() == 0):
            rep["self_winner_ties"] += 1
            if fg[i_o] + fg[w] == 0 and not (cf[i_o] == cf[w] and e[w] < tol_pose):
                bad(f"det {b} template {n} ({tag}): RANSAC winner {i_o} ({cf[i_o]}) vs float64 first maximum {w} ({cf[w]}) with no 14 px tie")
        if abs(c_o - cf[i_o]) > fg[i_o]:
            bad(f"det {b} template {n} ({tag}): {c_o} inliers vs float64 {cf[i_o]} for our winner with {fg[i_o]} correspondences within {eps_px:g} px of 14")
        want_failed = (r["n"] == 1) or (c_o == 0)
        if failed_o != want_failed:
            bad(f"det {b} template {n} ({tag}): failed flag {failed_o} with {r['n']} correspondences and {c_o} inliers")
        return check_pose_restated(b, jo, n, tag)

    ids_f = m["id_src"].astype(np.int64)
    ar = np.arange(P)
    for b in range(B):
        E = {}            # template id -> float64 similarity summed over OUR valid patches / 256
        flips = {}        # template id -> number of (explained) patch differences
        for j in tiles_of[b]:
            n = int(m["tile_n"][j])
            vf, vo = m["valid"][j].astype(bool), ours["tiles_valid"][j].astype(bool)
            sf, rsf, s2 = m["idx_t2s"][j].astype(np.int64), m["ridx_t2s"][j].astype(np.int64), m["idx2_t2s"][j].astype(np.int64)
            so = ours["tiles_idx"][j].astype(np.int64)
            f_row = m["row_margin"][j] < eps_sim
            f_rthr = np.abs(m["row_thr"][j]) < eps_sim
            f_col = (m["col_margin"][j] < eps_sim) | (np.abs(m["col_thr"][j]) < eps_sim)
            differ = (vo != vf) | (vo & vf & (so != sf))
            nd = int(differ.sum())
            flips[n] = nd
            if nd:
                rep["tiles_with_flips"] += 1
                rep["patch_flips"] += nd
            for t in np.flatnonzero(differ):
                fr = f_row[t] or f_rthr[t] or f_col[rsf[t]] or f_col[t]
                if vo[t]:
                    fr = fr or f_col[so[t]]
                    if so[t] != rsf[t] and not (f_row[t] and so[t] == s2[t]):
                        bad(f"det {b} template {n} patch {t}: matched to {so[t]}, float64 best {rsf[t]} / runner-up {s2[t]} (row margin {m['row_margin'][j, t]:.2e})")
                        continue
                if not fr:
                    bad(f"det {b} template {n} patch {t}: valid {bool(vo[t])} vs float64 {bool(vf[t])}, match {so[t]} vs {sf[t]}; no decision margin "
                        f"below {eps_sim:g} (row {m['row_margin'][j, t]:.2e}, thr {m['row_thr'][j, t]:.2e}, col {m['col_margin'][j, rsf[t]]:.2e})")
            rm = m["row_max"][j].astype(np.float64)
            simf = np.where(so == rsf, rm, np.where(so == s2, rm - m["row_margin"][j].astype(np.float64), np.nan))
            e = np.where(vo, simf, 0.0).sum() / P
            E[n] = e
            dev = abs(A_o[b, n] - e)
            if np.isfinite(dev):
                rep["max_avg_dev"] = max(rep["max_avg_dev"], dev)
            if not dev < eps_sim:
                bad(f"det {b} template {n}: sim_avg {A_o[b, n]:.8f} vs the float64 similarities over our valid patches {e:.8f}")
        # tiles the golden does not store: no flips are visible, so our sim_avg must be the float64 one unless the tile holds a tie
        stored = np.zeros(N, bool)
        stored[[int(m["tile_n"][j]) for j in tiles_of[b]]] = True
        dev = np.abs(A_o[b] - A_f[b])
        loose = (~stored) & (dev >= eps_sim)
        rep["unstored_dev"] += int(loose.sum())
        for n in np.flatnonzero(loose & (m["tile_min_margin"][b] >= eps_sim)):
            bad(f"det {b} template {n} (not stored): sim_avg {A_o[b, n]:.8f} vs float64 {A_f[b, n]:.8f} with no decision margin below {eps_sim:g} in the tile")
        # ranking: our top-k (matcher order = (sim_avg desc, id asc)) must be consistent with E
        mine = [int(x) for x in ours["id_src"][b]]
        so_set, sf_set = set(mine), set(int(x) for x in ids_f[b])
        if so_set != sf_set:
            rep["set_diff"] += 1
        if mine != [int(x) for x in ids_f[b]]:
            rep["order_diff"] += 1
        own_rank = sorted(range(N), key=lambda n: (-A_o[b, n], n))[:k]
        if set(own_rank) != so_set:
            bad(f"det {b}: hypotheses {sorted(so_set)} are not the top-{k} of our own sim_avg {sorted(own_rank)}")
        for n in mine:
            if n not in E:
                bad(f"det {b}: template {n} in our top-{k} is not among the tiles the golden stores (float64 top-{K} + near-tied tiles near the boundary)")
        for n in [x for x in mine if x in E]:
            for mm in [x for x in E if x not in so_set]:
                if not (E[n] > E[mm] - 2 * eps_sim):
                    bad(f"det {b}: template {n} ranked above {mm} although the float64 similarities over our valid patches say {E[n]:.8f} < {E[mm]:.8f}")
        # hypotheses aligned by template id
        counts_o = {}
        for jo, n in enumerate(mine):
            counts_o[n] = int(ours["inliers"][b, jo])
            jf = np.flatnonzero(ids_f[b] == n)
            if not len(jf):   # a template the float64 run does not rank (its place in OUR top k is checked above): self-consistency
                rep["hyp_checked"] += bool(check_self(b, jo, n, "not in the float64 top k"))
                continue
            jf = int(jf[0])
            rep["hyp_common"] += 1
            same_corr = (ours["src_pts"][b, jo] == m["src_pts"][b, jf]).all() and (ours["tar_pts"][b, jo] == m["tar_pts"][b, jf]).all()
            if flips.get(n, 0) == 0 and not same_corr:
                bad(f"det {b} template {n}: tile records equal the float64 run's but the final correspondences do not")
            if not same_corr:
                rep["corr_flip_hyp"] += 1
                if flips.get(n, 0) == 0:
                    bad(f"det {b} template {n}: correspondences differ without a patch flip in the tile")
                # the patches both runs hold with the same match feed the IST heads the same inputs: regressions must agree there
                both = ((ours["src_pts"][b, jo] == m["src_pts"][b, jf]).all(-1) & (ours["tar_pts"][b, jo] == m["tar_pts"][b, jf]).all(-1)
                        & (m["src_pts"][b, jf][:, 0] != -1))
                if both.any():
                    d = max(np.abs(ours["relScale"][b, jo][both] - m["relScale"][b, jf][both]).max(),
                            np.abs(ours["relInplane"][b, jo][both] - m["relInplane"][b, jf][both]).max())
                    rep["max_ist_dev"] = max(rep["max_ist_dev"], float(d))
                    if not d < tol_ist:
                        bad(f"det {b} template {n}: IST regression off by {d:.2e} on the correspondences shared with the float64 run")
                rep["hyp_checked"] += bool(check_self(b, jo, n, "correspondences differ by explained flips"))
                continue
            rep["hyp_same_corr"] += 1
            ok = m["src_pts"][b, jf][:, 0] != -1
            if ok.any():
                d = max(np.abs(ours["relScale"][b, jo][ok] - m["relScale"][b, jf][ok]).max(),
                        np.abs(ours["relInplane"][b, jo][ok] - m["relInplane"][b, jf][ok]).max())
                rep["max_ist_dev"] = max(rep["max_ist_dev"], float(d))
                if not d < tol_ist:
                    bad(f"det {b} template {n}: IST regression off by {d:.2e}")
            r = ransac_f64(m["src_pts"][b, jf], m["tar_pts"][b, jf], m["relScale"][b, jf], m["relInplane"][b, jf], eps_px)
            c_o, M_o = counts_o[n], ours["M"][b, jo].astype(np.float64)
            if r["n"] == 0:
                if c_o != 0 or np.abs(M_o - np.eye(3)).max() > 0:
                    bad(f"det {b} template {n}: no correspondences but count {c_o}")
                te, re_ = pose_rel_err(ours["poses"][b, jo].astype(np.float64), m["all_poses"][b, jf])
                if not (te < tol_pose and re_ < tol_pose):
                    bad(f"det {b} template {n}: no correspondences, pose off by {te:.2e} / {re_:.2e}")
                rep["hyp_checked"] += 1
                continue
            # The float64 run's own winner and count are read from ITS outputs, not from this restatement: many-to-one matches put
            # correspondences at exactly one patch (14 px) from the proposing one, so in ANY precision some errors sit within an ulp of
            # the threshold and the count depends on the operation order inside torch's einsum (ransac.py:93-95).
            def nearest(Mx):
                e = np.abs(r["M"] - Mx[None]).max(axis=(1, 2)) / np.abs(r["M"]).max(axis=(1, 2))
                i = int(np.argmin(e))
                return i, float(e[i])
            (i_o, e_o), (w, e_w) = nearest(M_o), nearest(m["M"][b, jf])
            c_w = int(round(float(m["all_scores"][b, jf]) * P))
            cf, fg = r["counts"], r["fragile"]
            assert e_w < 1e-9 and abs(c_w - cf[w]) <= fg[w], f"det {b} template {n}: the RANSAC restatement does not reproduce the float64 golden"
            if e_o > 1e-3:
                bad(f"det {b} template {n}: M is no float64 candidate's (nearest differs by {e_o:.2e})")
                continue
            checked = False
            if i_o != w:
                rep["winner_diff"] += 1
                if not (fg[i_o] + fg[w] > 0 and cf[i_o] + fg[i_o] >= cf[w] - fg[w]):
                    bad(f"det {b} template {n}: RANSAC winner {i_o} ({cf[i_o]} inliers, {fg[i_o]} within {eps_px:g} px of 14) vs float64 winner {w} ({cf[w]}, {fg[w]})")
            if c_o != c_w:
                rep["inlier_diff"] += 1
            if abs(c_o - cf[i_o]) > fg[i_o]:
                bad(f"det {b} template {n}: {c_o} inliers vs float64 {cf[i_o]} for the same candidate with {fg[i_o]} correspondences within {eps_px:g} px of 14")
            if i_o == w and c_o == c_w:
                rep["hyp_same_all"] += 1
                if bool(ours["idx_failed"][b, jo]) != bool(m["idx_failed"][b, jf]):
                    bad(f"det {b} template {n}: failed flag {bool(ours['idx_failed'][b, jo])} vs {bool(m['idx_failed'][b, jf])}")
                me = float(np.abs(M_o - m["M"][b, jf]).max() / np.abs(m["M"][b, jf]).max())
                te, re_ = pose_rel_err(ours["poses"][b, jo].astype(np.float64), m["all_poses"][b, jf])
                rep["max_M_err"], rep["max_t_err"], rep["max_r_err"] = max(rep["max_M_err"], me), max(rep["max_t_err"], float(te)), max(rep["max_r_err"], float(re_))
                if not (me < tol_pose and te < tol_pose and re_ < tol_pose):
                    bad(f"det {b} template {n}: same discrete choices but M / translation / rotation off by {me:.2e} / {te:.2e} / {re_:.2e}")
                checked = True
            else:
                # same correspondences, another (explained: 14 px tie) winner or count: M must be that float64 candidate's and the pose
                # the float64 recovery of it
                me = float(np.abs(M_o - r["M"][i_o]).max() / np.abs(r["M"][i_o]).max())
                rep["max_self_M_err"] = max(rep["max_self_M_err"], me)
                if not me < tol_pose:
                    bad(f"det {b} template {n}: M off by {me:.2e} from the float64 candidate {i_o} it was matched to")
                want_failed = (r["n"] == 1) or (c_o == 0)
                if bool(ours["idx_failed"][b, jo]) != want_failed:
                    bad(f"det {b} template {n}: failed flag {bool(ours['idx_failed'][b, jo])} with {r['n']} correspondences and {c_o} inliers")
                checked = check_pose_restated(b, jo, n, "14 px tie: another winner / count than the float64 run")
            rep["hyp_checked"] += bool(checked)
        # final order: inlier count descending, ties in matcher order (gigaPose.py:588-594 with a stable sort)
        rank = {n: i for i, n in enumerate(sorted(mine, key=lambda n: (-A_o[b, n], n)))}
        want = sorted(mine, key=lambda n: (-counts_o[n], rank[n]))
        if want != mine:
            bad(f"det {b}: hypothesis order {mine} is not (inliers desc, sim_avg desc) = {want}")
    return rep


def summary(rep):
    keys = ["tiles", "tiles_with_flips", "patch_flips", "unstored_dev", "set_diff", "order_diff", "hyp", "hyp_common", "corr_flip_hyp", "hyp_same_corr",
            "winner_diff", "inlier_diff", "hyp_same_all", "hyp_self_checked", "hyp_checked"]
    s = ", ".join(f"{k2} {rep[k2]}" for k2 in keys)
    return (f"{s}; max |sim_avg - float64 over our patches| {rep['max_avg_dev']:.2e}, IST dev {rep['max_ist_dev']:.2e}, on identical discrete paths M "
            f"{rep['max_M_err']:.2e} / t {rep['max_t_err']:.2e} / R {rep['max_r_err']:.2e}; on the others vs the float64 restatement on our own "
            f"correspondences M {rep['max_self_M_err']:.2e} / t {rep['max_self_t_err']:.2e} / R {rep['max_self_r_err']:.2e}; "
            f"UNEXPLAINED {len(rep['unexplained'])}")
