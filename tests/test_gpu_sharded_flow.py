"""GPU: `GigaPose.test_step` with a template-SHARDED bank and the REAL kernels (gigapose_amd/sharded_flow.py).

(a) two ranks on the one MI355X of the test box (gloo, host-staged collectives -- RCCL refuses two ranks on one device): rank 0 is fed
    images of 5 / 9 / 0 detections, rank 1 of 7 / 3 / 12, as the reference's test loop would (one image per test_step, reference
    test.py:55-60).  Nobody raises; every per-image npz equals, BYTE FOR BYTE in `chain` numerics (except `time`), the file the
    UNSHARDED per-image flow writes for the same image in the same process;
(b) one rank over RCCL with the collectives forced (GIGAPOSE_FORCE_COLLECTIVES=1): the same flow through the nccl backend's
    all_gather_into_tensor / all_to_all_single on device buffers, including the status-word all-gather and the done-word exchange.
"""
import os
import socket

import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu
SIZES = {0: [5, 9, 0], 1: [7, 3, 12]}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _image(tset, seed, n, view_id, dev):
    from gigapose_amd.tensor_collection import PandasTensorCollection

    q = tset.crops(seed, max(n, 1), dev)
    labels = q["labels"].numpy()[:n]
    infos = pd.DataFrame(dict(label=[str(l) for l in labels], scene_id=[2] * n, view_id=[view_id] * n))
    batch = PandasTensorCollection(infos=infos, **{k: q[k][:n].contiguous() for k in ["tar_img", "tar_mask", "tar_K", "tar_M"]})
    objs = sorted(set(int(l) for l in labels))
    batch.test_list = PandasTensorCollection(infos=pd.DataFrame(dict(
        im_id=[view_id] * len(objs), scene_id=[2] * len(objs), obj_id=objs,
        inst_count=[int((labels == o).sum()) for o in objs], detection_time=[0.05] * len(objs))))
    return batch


def _load(log_dir, idx):
    with np.load(os.path.join(log_dir, "predictions", f"{idx}.npz")) as z:
        return {k: z[k] for k in z.files}


def _run(tmp, sub, numerics, sizes, rank, dev, sharded, rows, n_templates=11):
    from gigapose_amd import _lib
    from gigapose_testing import factory

    log_dir = os.path.join(tmp, f"{sub}_r{rank}")
    tset = factory.TemplateSet(2, n_templates, seed=70)
    model = factory.build_model("dinov2_vits14", k=5, device=dev, seed=5)
    model.set_numerics(numerics)
    model.log_dir, model.test_dataset_name, model.run_id = log_dir, "syn", "r0"
    os.makedirs(os.path.join(log_dir, "predictions"), exist_ok=True)
    model.accumulate_crops = rows if sharded else 0
    if sharded:
        model.enable_template_sharding()
    model.template_datasets = {"syn": tset}
    model.set_template_data("syn")
    for i, n in enumerate(sizes):
        if n == 0 and not sharded:
            continue   # (the per-image flow has nothing to compare an empty image with)
        assert model.test_step(_image(tset, 300 + 10 * rank + i, n, 20 + 10 * rank + i, dev), i) == 0
    model.flush_pending()
    torch.cuda.synchronize()
    _lib.check_status()
    files = {i: _load(log_dir, i) for i, n in enumerate(sizes) if n > 0 or sharded}
    return files, model


def _compare(want, got, sizes, numerics, who):
    for i, n in enumerate(sizes):
        if n == 0:
            assert got[i]["poses"].shape == (0, 5, 4, 4) and len(got[i]["object_id"]) == 0
            continue
        assert sorted(want[i]) == sorted(got[i])
        for key in want[i]:
            assert want[i][key].dtype == got[i][key].dtype and want[i][key].shape == got[i][key].shape, (who, i, key)
            if key == "time":
                assert (got[i]["time"] > 0).all()
            elif numerics == "chain":
                assert want[i][key].tobytes() == got[i][key].tobytes(), f"{who} image {i}: {key} differs from the unsharded per-image flow"
        if numerics == "split":   # a crop's ViT round-off depends on its batch's GEMM partition (as in the unsharded accumulated flow)
            for key in ("scene_id", "im_id", "object_id", "detection_time"):
                np.testing.assert_array_equal(want[i][key], got[i][key])
            eq = want[i]["scores"] == got[i]["scores"]
            assert eq.mean() >= 0.8, f"{who} image {i}: only {eq.mean():.2f} of the hypothesis scores equal"


def _worker(rank, world, port, tmp):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for numerics in ("chain", "split"):
            want, _ = _run(tmp, "plain_" + numerics, numerics, SIZES[rank], rank, dev, sharded=False, rows=0)
            got, model = _run(tmp, "shard_" + numerics, numerics, SIZES[rank], rank, dev, sharded=True, rows=8)
            flow = model._flow()
            assert not flow.queue and flow.in_flight is None
            _compare(want, got, SIZES[rank], numerics, f"rank {rank} {numerics}")
    finally:
        dist.destroy_process_group()


def test_sharded_test_step_with_different_detection_counts_per_rank_two_ranks_on_one_gpu(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


@pytest.mark.parametrize("numerics", ["chain", "split"])
def test_sharded_test_step_over_rccl_with_forced_collectives(tmp_path, monkeypatch, numerics):
    import torch.distributed as dist

    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    monkeypatch.setenv("GIGAPOSE_FORCE_COLLECTIVES", "1")
    dev = torch.device("cuda", 0)
    sizes = [5, 9, 0, 7, 3, 12, 20]
    want, _ = _run(str(tmp_path), "plain", numerics, sizes, 0, dev, sharded=False, rows=0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        got, model = _run(str(tmp_path), "shard", numerics, sizes, 0, dev, sharded=True, rows=16)
    finally:
        dist.destroy_process_group()
    _compare(want, got, sizes, numerics, "rccl world 1")


def _replica_worker(rank, world, port, tmp):
    """Unsharded replicas + `image_ownership: round_robin`: both ranks are fed EVERY image (the reference's loader has no split_by_node),
    each computes idx % world == rank, rank 0 merges after the barrier of on_test_epoch_end."""
    import torch.distributed as dist

    from gigapose_testing import factory

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tset = factory.TemplateSet(2, 11, seed=70)
        model = factory.build_model("dinov2_vits14", k=5, device=dev, seed=5, log_dir=os.path.join(tmp, "owned"))
        model.set_numerics("chain")
        model.global_rank = rank
        model.image_ownership, model.accumulate_crops = "round_robin", 16
        model.test_dataset_name, model.run_id = "syn", "r0"
        model.template_datasets = {"syn": tset}
        model.set_template_data("syn")
        sizes = [5, 9, 3, 7, 12, 4]
        for i, n in enumerate(sizes):
            assert model.test_step(_image(tset, 300 + i, n, 20 + i, dev), i) == 0
        model.on_test_epoch_end()
        if rank == 0:   # every image's file is there when rank 0 merges (the barrier), each written once, and the csv holds them all
            pred = os.path.join(tmp, "owned", "predictions")
            assert sorted(f for f in os.listdir(pred) if f.endswith(".npz")) == sorted(f"{i}.npz" for i in range(len(sizes)))
            csv = [f for f in sorted(os.listdir(pred)) if f.endswith(".csv")][0]
            assert len(pd.read_csv(os.path.join(pred, csv))) == sum(sizes)
    finally:
        dist.destroy_process_group()


def test_round_robin_image_ownership_with_two_replicas_on_one_gpu(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_replica_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    # the same images through ONE process: identical files (chain numerics: a crop's result does not depend on its batch)
    from gigapose_testing import factory

    dev = torch.device("cuda", 0)
    tset = factory.TemplateSet(2, 11, seed=70)
    model = factory.build_model("dinov2_vits14", k=5, device=dev, seed=5, log_dir=str(tmp_path / "single"))
    model.set_numerics("chain")
    model.accumulate_crops, model.test_dataset_name, model.run_id = 0, "syn", "r0"
    model.template_datasets = {"syn": tset}
    model.set_template_data("syn")
    sizes = [5, 9, 3, 7, 12, 4]
    for i, n in enumerate(sizes):
        model.test_step(_image(tset, 300 + i, n, 20 + i, dev), i)
    for i in range(len(sizes)):
        a, b = _load(str(tmp_path / "owned"), i), _load(str(tmp_path / "single"), i)
        for key in a:
            if key != "time":
                assert a[key].tobytes() == b[key].tobytes(), f"image {i}: {key}"
