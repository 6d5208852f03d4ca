"""The template-sharded path at WORLD SIZE 2 with the REAL kernels (missing 1): two processes share the one
MI355X of the test box (RCCL refuses two ranks on one device, so the ranks talk over gloo with host-staged buffers --
gigapose_amd/sharding.py does that staging itself when the group's backend is gloo).  Each rank onboards its shard of every
object's templates (81 + 81 of 162, and the uneven 6 + 5 of 11), runs GigaPose.predict on ITS OWN crops -- exchange #1 gathers
both ranks' query rows, gp_match_tiles[_split] / gp_topk / gp_gather_records run on the rank-major gathered rows against a shard
with a non-zero template offset, exchange #2 returns each rank the candidates of its crops, merge, IST / RANSAC / recovery --
and every tensor of the result must equal, bit for bit, the unsharded predict of the same crops in the same process."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gigapose_amd import _lib
        from gigapose_testing import factory
        from gigapose_amd.sharding import ShardedMatcher, shard_bounds

        for n_templates in (162, 11):
            tset = factory.TemplateSet(2, n_templates, seed=70)
            q = tset.crops(71 + rank, 3, dev)                       # every rank its own three crops
            for numerics in ("chain", "split"):
                def run(sharded):
                    model = factory.build_model("dinov2_vits14", k=5, device=dev, seed=5)
                    model.set_numerics(numerics)
                    if sharded:
                        model.enable_template_sharding()
                    model.template_datasets = {"syn": tset}
                    model.set_template_data("syn")
                    if sharded:
                        bank = model.match_banks["syn"]
                        lo, hi = shard_bounds(n_templates, world, rank)
                        assert isinstance(bank, ShardedMatcher) and bank.lo == lo and bank.bank.N == hi - lo and (rank == 0 or lo > 0)
                    p = model.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn")
                    torch.cuda.synchronize()
                    _lib.check_status()
                    return {n: v.cpu() for n, v in p.tensors.items()}

                plain, shard = run(False), run(True)
                assert set(plain) == set(shard)
                for name in plain:
                    assert torch.equal(plain[name], shard[name]), f"rank {rank}, {n_templates} templates, {numerics}: {name} differs"
                assert (shard["id_src"] >= 0).all() and (shard["id_src"] < n_templates).all()
                # both shards contribute winners somewhere (otherwise the merge of two ranks' candidates was not exercised)
                lo1, _ = shard_bounds(n_templates, world, 1)
                assert (shard["id_src"] < lo1).any() and (shard["id_src"] >= lo1).any(), "all winners come from one shard"
    finally:
        dist.destroy_process_group()


def test_template_sharded_predict_with_two_ranks_on_one_gpu_equals_unsharded():
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(2, _free_port()), nprocs=2, join=True)
