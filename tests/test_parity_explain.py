"""CPU test of the explained-tie checker (tests/parity_explain.py) with the reference's OWN float32 run as the second
implementation: every difference between the reference's float32 goldens and its float64 run must be explained by a float64
decision margin below epsilon -- which also measures the epsilon the reference's own rounding needs (the yardstick for the GPU
assertion in tests/test_gpu_parity_big.py).  Goldens: oracle/make_goldens.py (<which>.npz), oracle/make_margins.py
(<which>_margins.npz, <which>_ref32_tiles.npz)."""
import os

import numpy as np
import pytest

import parity_explain as px


def ref32_as_ours(golden_dir, which):
    g = np.load(os.path.join(golden_dir, which + ".npz"))
    t = np.load(os.path.join(golden_dir, which + "_ref32_tiles.npz"))
    return dict(tiles_valid=np.unpackbits(t["valid"], axis=-1).astype(bool), tiles_idx=t["idx"], sim_avg=t["sim_avg"],
                id_src=g["id_src"].astype(np.int64), src_pts=g["src_pts"].astype(np.int64), tar_pts=g["tar_pts"].astype(np.int64),
                inliers=np.rint(g["all_scores"] * 256).astype(np.int64), idx_failed=g["idx_failed"], relScale=g["relScale"],
                relInplane=g["relInplane"], M=g["M"], poses=g["all_poses"])


@pytest.mark.parametrize("which", ["e2e", "e2e_cfg2", "e2e_cfg3"])
def test_reference_f32_vs_its_float64_run_is_explained(golden_dir, which):
    path = os.path.join(golden_dir, which + "_margins.npz")
    if not os.path.exists(path):
        pytest.skip(f"{which}_margins.npz not generated")
    m = dict(np.load(path))
    rep = px.explain(m, ref32_as_ours(golden_dir, which))
    print(f"{which}: the reference's float32 run vs its float64 run -- {px.summary(rep)}")
    for line in rep["unexplained"][:20]:
        print("   UNEXPLAINED:", line)
    assert not rep["unexplained"]


def test_checker_rejects_planted_errors(golden_dir):
    """A flipped patch away from any tie, a swapped template and a shifted pose must each be reported."""
    m = dict(np.load(os.path.join(golden_dir, "e2e_margins.npz")))
    base = ref32_as_ours(golden_dir, "e2e")
    assert not px.explain(m, base)["unexplained"]
    o = {k: v.copy() for k, v in base.items()}
    t = int(np.flatnonzero(o["tiles_valid"][0])[3])
    o["tiles_valid"][0, t] = False                                      # a robust patch dropped (stored tile 0)
    assert any("patch %d" % t in s for s in px.explain(m, o)["unexplained"])
    o = {k: v.copy() for k, v in base.items()}
    o["poses"][1, 0, :3, 3] *= 1.001                                    # a pose 1e-3 off
    assert any("translation" in s for s in px.explain(m, o)["unexplained"])
    o = {k: v.copy() for k, v in base.items()}
    o["inliers"][2, 0] -= 1                                             # an inlier count off by one with no 14 px tie
    assert px.explain(m, o)["unexplained"]
    o = {k: v.copy() for k, v in base.items()}
    o["sim_avg"][int(m["tile_b"][0]), int(m["tile_n"][0])] += 1e-4      # a wrong sim_avg
    assert any("sim_avg" in s for s in px.explain(m, o)["unexplained"])
