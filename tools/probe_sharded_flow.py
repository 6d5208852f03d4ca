"""What does a template-sharded step cost over an unsharded one on ONE GPU?  (round 6)
predict() at 64 crops on the headline bank (ViT-L/14, 1 x 162 templates): MODE=plain -- the unsharded path; MODE=sharded -- a one-rank
RCCL group with the collectives forced (GIGAPOSE_FORCE_COLLECTIVES=1): exchange #1 / #2, packing, merge.  Prints ms per step (events);
under `rocprofv3 --kernel-trace --stats` the two kernel lists differ by exactly the sharded path's glue.
    MODE=sharded python tools/probe_sharded_flow.py [steps] [live_rows]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapose_testing import factory  # noqa: E402


def main(steps=10, live=None):
    mode = os.environ.get("MODE", "plain")
    dev = torch.device("cuda", 0)
    if mode == "sharded":
        import torch.distributed as dist

        os.environ["GIGAPOSE_FORCE_COLLECTIVES"] = "1"
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29631", rank=0, world_size=1, device_id=dev)
    model = factory.build_model("dinov2_vitl14", k=5, device=dev, seed=0)
    if mode == "sharded":
        model.enable_template_sharding()
    tset = factory.TemplateSet(1, 162, seed=100)
    model.template_datasets = {"syn": tset}
    model.set_template_data("syn")
    q = tset.crops(7, 64, dev)
    kw = {} if live is None else {"live_rows": live}
    for _ in range(3):
        model.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn", **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        model.predict(q["tar_img"], q["tar_mask"], q["tar_K"], q["tar_M"], q["labels"], "syn", **kw)
    e1.record()
    torch.cuda.synchronize()
    sys.stderr.write(f"mode {mode} live_rows {live}: {e0.elapsed_time(e1) / steps:.3f} ms per 64-crop step\n")
    if mode == "sharded":
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else None)
