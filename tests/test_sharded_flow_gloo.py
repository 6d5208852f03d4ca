"""CPU, world_size 2 / 3 (gloo): `GigaPose.test_step` with a template-SHARDED bank (gigapose_amd/sharded_flow.py) -- host logic only.

The reference's test.py feeds one image per test_step (reference test.py:55-60) with as many detections as the image has; ranks
see different images.  The sharded step's exchanges are fixed-size collectives, so the product runs flushes of exactly
`accumulate_crops` crop rows on every rank and lets the ranks agree INSIDE the flushes on when everybody is done.  Here the device
half of a flush (`GigaPose._run_rows`: predict + downloads) is replaced by a CPU double that computes every crop's rows from the
crop alone and performs the flush's two cross-rank facts -- all ranks' "done" words (they travel with exchange #1) and all ranks'
status words -- as REAL gloo collectives.  A rank whose sequence of collectives differs from its peers' therefore hangs (the group
has a 60 s timeout) or reads words of the wrong shape: what runs as shipped is the crop-granular queue, the cut through images, the
padding, L(j+1)-then-F(j) pipelining, the collectively agreed end, the symmetric range recovery and the file writer."""
import os
import socket
from datetime import timedelta

import numpy as np
import pandas as pd
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gigapose_amd import _lib
from gigapose_amd.gigaPose import GigaPose
from gigapose_amd.tensor_collection import PandasTensorCollection

from test_accumulate_host import K_HYP, _Ev, crop_result, image, load


class _Metric:
    k = K_HYP


class _ShardedModel(GigaPose):
    def __init__(self, log_dir, rows, rank, world, ownership=None):
        torch.nn.Module.__init__(self)
        self.log_dir, self.test_setting, self.test_dataset_name = log_dir, "localization", "syn"
        os.makedirs(os.path.join(log_dir, "predictions"), exist_ok=True)
        self.template_datas, self.template_shard = {"syn": object()}, (rank, world, None)
        self.accumulate_crops, self._pending, self._pending_crops, self._in_flight = rows, [], 0, None
        self.image_ownership, self._sharded_flow = ownership, None
        self.testing_metric = _Metric()
        self.launched, self.live_rows, self.trip_on, self.recoveries, self.clock = 0, [], None, 0, 0.0
        self.model_name, self.run_id, self.global_rank = "large", "r0", rank

    def _flushable(self, batch):
        return getattr(batch, "test_list", None) is not None

    def _drain_device(self):
        pass

    def _recover_range(self, bits, images):
        # the product's remedy is collective (plane maxima all-reduced, once per tensor of `images`): the double all-reduces too,
        # so ranks that disagree on the number of calls do not pass
        assert len(images) == 2
        for x in images:
            t = torch.tensor([float(x.shape[0])])
            dist.all_reduce(t)
        self.recoveries += 1
        return "recalibrated" if self.recoveries <= 2 else False

    def _run_rows(self, inputs, labels_np, dataset_name, aux=None, live_rows=None):
        rank, world, _ = self.template_shard
        assert live_rows is not None and 0 <= live_rows <= len(labels_np)
        imgs = inputs["tar_img"]
        assert imgs.shape[0] == self.accumulate_crops == len(labels_np), "a sharded flush is exactly accumulate_crops rows"
        res = [crop_result(i) for i in imgs]
        status = 4 if (self.trip_on is not None and self.launched in self.trip_on) else 0
        self.launched += 1
        self.live_rows.append(int((imgs.reshape(imgs.shape[0], -1).abs().sum(1) > 0).sum()))
        assert self.live_rows[-1] == live_rows, "the flow tells the device half how many of its rows are real"
        words = torch.zeros(world, dtype=torch.int32)
        dist.all_gather_into_tensor(words, torch.tensor([status], dtype=torch.int32))                 # the flush's status all-gather
        aux_all = torch.zeros(world * len(labels_np), dtype=torch.int32)
        dist.all_gather_into_tensor(aux_all, torch.full((len(labels_np),), int(aux or 0), dtype=torch.int32))   # rides on exchange #1
        t0 = self.clock
        self.clock += 10.0
        host = dict(scores=torch.from_numpy(np.stack([r[0] for r in res])), pred_poses=torch.from_numpy(np.stack([r[1] for r in res])),
                    status=words, aux_all=aux_all, bad_crop_M=torch.zeros(1, dtype=torch.int32))
        return dict(inputs=inputs, labels=labels_np, pred=None, host=host, ev=(_Ev(t0), _Ev(self.clock)), dataset_name=dataset_name, device="cpu")


def full_image(seed, n, view_id):
    """test_accumulate_host.image with the other tensors the flow slices and pads."""
    b = image(seed, max(n, 1), view_id)
    if n == 0:
        b = PandasTensorCollection(infos=b.infos.iloc[:0], tar_img=b.tar_img[:0])
        b.test_list = PandasTensorCollection(infos=pd.DataFrame(dict(im_id=[], scene_id=[], obj_id=[], inst_count=[], detection_time=[])))
    tl = b.test_list
    b.register_tensor("tar_mask", torch.ones(n, 4, 4))
    b.register_tensor("tar_K", torch.eye(3).expand(n, 3, 3).contiguous())
    b.register_tensor("tar_M", torch.eye(3).expand(n, 3, 3).contiguous())
    b.test_list = tl
    return b


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, rows, sizes_by_rank, trip_on, ownership, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=60))
    try:
        m = _ShardedModel(os.path.join(tmp, "sharded" if ownership is None else "owned"), rows, rank, world, ownership)
        m.trip_on = trip_on.get(rank)
        if ownership is None:
            feed = [(1000 * rank + i, full_image(5000 + 100 * rank + i, n, view_id=100 * rank + i)) for i, n in enumerate(sizes_by_rank[rank])]
        else:   # every rank replays every image (the reference's loader has no split_by_node); ownership picks
            feed = [(i, full_image(7000 + i, n, view_id=i)) for i, n in enumerate(sizes_by_rank[0])]
        for idx, b in feed:
            assert m.test_step(b, idx) == 0
        m.on_test_epoch_end() if ownership is not None else m.flush_pending()
        flow = m._flow()
        assert not flow.queue and flow.queued == 0 and flow.in_flight is None
        ret[rank] = dict(launched=m.launched, live=m.live_rows, recoveries=m.recoveries)
    finally:
        dist.destroy_process_group()


def _spawn(world, tmp, rows, sizes_by_rank, trip_on=None, ownership=None):
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), str(tmp), rows, sizes_by_rank, trip_on or {}, ownership, ret), nprocs=world, join=True)
    return dict(ret)


def _expected(seed, n, view_id, tmp, idx):
    """The per-image flow's file for the same image (single process, accumulate_crops = 0: tests/test_accumulate_host.py)."""
    from test_accumulate_host import _Model

    m = _Model(str(tmp), 0)
    m.test_step(image(seed, n, view_id), idx)
    return load(tmp, idx)


def _check_files(tmp, sub, sizes_by_rank, ref_dir):
    for rank, sizes in enumerate(sizes_by_rank):
        for i, n in enumerate(sizes):
            idx = 1000 * rank + i
            got = load(os.path.join(tmp, sub), idx)
            if n == 0:
                assert got["poses"].shape == (0, K_HYP, 4, 4) and got["scores"].shape == (0, K_HYP) and len(got["object_id"]) == 0
                continue
            want = _expected(5000 + 100 * rank + i, n, 100 * rank + i, ref_dir, idx)
            assert sorted(got) == sorted(want)
            for key in want:
                assert got[key].dtype == want[key].dtype and got[key].shape == want[key].shape, (idx, key)
                if key != "time":
                    np.testing.assert_array_equal(got[key], want[key], err_msg=f"rank {rank} image {i}: {key}")
            assert (got["time"] > 0).all()


def test_ranks_with_different_detection_counts_flush_fixed_rows_and_agree_on_the_end(tmp_path):
    """The review's case: rank 0 sees images of 5 / 9 / 0 detections, rank 1 of 7 / 3 / 12; flushes of 8 rows.  Nobody raises, both ranks
    launch the same number of flushes, every file equals the per-image flow's."""
    sizes = [[5, 9, 0], [7, 3, 12]]
    ret = _spawn(2, tmp_path, 8, sizes)
    # rank 0: 14 crops -> 8 + 6(+2 dummies); rank 1: 22 -> 8 + 8 + 6(+2).  The end is seen in the third flush (both done), one
    # all-dummy flush was already queued behind it: 4 launches on BOTH ranks
    assert ret[0]["launched"] == ret[1]["launched"] == 4
    assert ret[0]["live"] == [8, 6, 0, 0] and ret[1]["live"] == [8, 8, 6, 0]
    _check_files(str(tmp_path), "sharded", sizes, tmp_path / "ref")


def test_a_rank_without_any_image_and_an_image_larger_than_a_flush(tmp_path):
    sizes = [[], [40, 3], [16]]
    ret = _spawn(3, tmp_path, 16, sizes)
    assert len({r["launched"] for r in ret.values()}) == 1
    assert ret[0]["live"] == [0] * ret[0]["launched"]                       # all-dummy flushes only
    assert ret[1]["live"][:3] == [16, 16, 11] and ret[2]["live"][0] == 16   # 40 crops cut 16 + 16 + 8, the 3-crop image joins the third flush
    _check_files(str(tmp_path), "sharded", sizes, tmp_path / "ref")


def test_a_range_trip_on_one_rank_is_recovered_by_all_ranks_together(tmp_path):
    """Flush 1 of rank 0 trips the split range guard.  The status words are shared, so BOTH ranks drop flushes 1 and 2, recover (a
    collective) and run the crops again; a later second trip (flush 4 of rank 1) is recovered the same way.  Files unchanged."""
    sizes = [[8, 8, 8, 5], [8, 3, 8, 8, 2]]
    clean = _spawn(2, tmp_path / "clean", 8, sizes)
    ret = _spawn(2, tmp_path / "trip", 8, sizes, trip_on={0: [1], 1: [4]})
    assert ret[0]["recoveries"] == ret[1]["recoveries"] == 2
    assert ret[0]["launched"] == ret[1]["launched"] > clean[0]["launched"] == clean[1]["launched"]
    _check_files(str(tmp_path / "trip"), "sharded", sizes, tmp_path / "ref")


def test_round_robin_image_ownership_splits_the_replayed_images(tmp_path):
    """Every rank is fed every image (the reference's loader does not split them); `image_ownership: round_robin` makes rank r
    compute idx % world == r only, and on_test_epoch_end merges after a barrier: all files exist exactly once, the csv holds them all."""
    sizes = [[5, 9, 3, 7, 12, 4, 8]]
    ret = _spawn(2, tmp_path, 8, sizes, ownership="round_robin")
    assert sum(sum(r["live"]) for r in ret.values()) == sum(sizes[0])        # every crop computed once over the two ranks
    pred = os.path.join(str(tmp_path), "owned", "predictions")
    assert sorted(f for f in os.listdir(pred) if f.endswith(".npz")) == sorted(f"{i}.npz" for i in range(len(sizes[0])))
    csvs = sorted(f for f in os.listdir(pred) if f.endswith(".csv"))
    assert len(csvs) == 2
    top1 = pd.read_csv(os.path.join(pred, csvs[0]))
    assert len(top1) == sum(sizes[0]) and set(top1.im_id) == set(range(len(sizes[0])))
