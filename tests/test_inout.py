"""CPU: the BOP csv writer (gigapose_amd/inout.py) produces byte-identical files to the unmodified reference
writer (golden tests/golden/bop_csv.npz from oracle/make_goldens.py gen_bop_csv) on the same per-batch npz files."""
import os

import numpy as np
import pytest

from gigapose_amd import inout
from gigapose_testing import synthetic as syn


@pytest.mark.parametrize("dataset", ["lmo", "ycbv"])
def test_bop_csv_is_byte_identical_to_reference(golden_dir, tmp_path, dataset):
    g = np.load(os.path.join(golden_dir, "bop_csv.npz"))
    for i, b in enumerate(syn.prediction_batches(int(g["seed"]))):
        np.savez(tmp_path / f"{i}.npz", **b)
    paths = inout.save_predictions_from_batched_predictions(str(tmp_path), dataset_name=dataset, model_name="large",
                                                            run_id="r0", is_refined=False)
    assert len(paths) == 2
    for p in paths:
        want = g[f"{dataset}:{os.path.basename(p)}"].tobytes()
        assert open(p, "rb").read() == want, f"{os.path.basename(p)} differs from the reference writer's output"


def test_top1_only_files_write_one_csv(tmp_path):
    b = syn.prediction_batches(3, n_batches=1)[0]
    b["poses"], b["scores"] = b["poses"][:, 0], b["scores"][:, 0]
    np.savez(tmp_path / "0.npz", **b)
    paths = inout.save_predictions_from_batched_predictions(str(tmp_path), "ycbv", "m", "x", is_refined=False)
    assert len(paths) == 1 and open(paths[0]).read().splitlines()[0] == "scene_id,im_id,obj_id,score,R,t,time"
