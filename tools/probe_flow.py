"""Where does the accumulated drop-in flow (GigaPose.test_step with accumulate_crops = 64) spend its wall time?  (round 5)
64 images x 8 detections on the headline bank; per flush: host time to queue it (_run_flush), host wait for the previous flush's results
(event synchronize inside _finish_flush), host time to write its files, and the flush's device time (its own events).
    python tools/probe_flow.py [images] [detections]"""
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapose_testing import factory  # noqa: E402
from gigapose_amd.gigaPose import GigaPose  # noqa: E402
from gigapose_amd.tensor_collection import PandasTensorCollection  # noqa: E402


def main(n_img=64, n_det=8):
    dev = torch.device("cuda", 0)
    model = factory.build_model("dinov2_vitl14", k=5, device=dev, seed=0)
    tset = factory.TemplateSet(1, 162, seed=100)
    model.template_datasets = {"syn": tset}
    model.test_dataset_name = "syn"
    model.set_template_data("syn")
    images = []
    for im in range(n_img):
        q = tset.crops(5000 + im, n_det, dev)
        lab = q["labels"].numpy()
        b = PandasTensorCollection(infos=pd.DataFrame(dict(label=[str(l) for l in lab], scene_id=[1] * n_det, view_id=[im] * n_det)),
                                   **{k: q[k] for k in ["tar_img", "tar_mask", "tar_K", "tar_M"]})
        b.test_list = PandasTensorCollection(infos=pd.DataFrame(dict(im_id=[im], scene_id=[1], obj_id=[1], inst_count=[n_det], detection_time=[0.0])))
        images.append(b)
    log = []
    run0, fin0, save0 = GigaPose._run_flush, GigaPose._finish_flush, GigaPose._save_image
    acc = {"save": 0.0}

    def run(self, imgs, name):
        t = time.perf_counter()
        job = run0(self, imgs, name)
        job["t_queue"] = time.perf_counter() - t
        return job

    def save(*a, **k):
        t = time.perf_counter()
        r = save0(*a, **k)
        acc["save"] += time.perf_counter() - t
        return r

    def fin(self, job):
        t = time.perf_counter()
        job["ev"][1].synchronize()
        t_wait = time.perf_counter() - t
        acc["save"] = 0.0
        t = time.perf_counter()
        fin0(self, job)
        log.append(dict(crops=len(job["labels"]), queue_ms=1e3 * job["t_queue"], wait_ms=1e3 * t_wait, finish_ms=1e3 * (time.perf_counter() - t),
                        save_ms=1e3 * acc["save"], gpu_ms=job["ev"][0].elapsed_time(job["ev"][1])))

    GigaPose._run_flush, GigaPose._finish_flush, GigaPose._save_image = run, fin, staticmethod(save)
    for rep in range(2):
        model.log_dir = tempfile.mkdtemp(prefix="flow_")
        os.makedirs(os.path.join(model.log_dir, "predictions"), exist_ok=True)
        log.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_steps = 0.0
        for i, b in enumerate(images):
            t = time.perf_counter()
            model.test_step(b, i)
            t_steps += time.perf_counter() - t
        model.flush_pending()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print(f"pass {rep}: {n_img * n_det / wall:.1f} crops/s, wall {1e3 * wall:.1f} ms, sum of flush device times {sum(l['gpu_ms'] for l in log):.1f} ms")
        for l in log:
            print("   flush of %(crops)d crops: queue %(queue_ms).2f ms | wait for its results %(wait_ms).2f | finish %(finish_ms).2f (files %(save_ms).2f) | device %(gpu_ms).2f ms" % l)
    # the same 64-crop batches through predict() back to back (the bench's step) for comparison
    cat = lambda name, s: torch.cat([images[i].tensors[name] for i in range(s, s + 8)])
    batches = [(cat("tar_img", s), cat("tar_mask", s), cat("tar_K", s), cat("tar_M", s), torch.ones(64, dtype=torch.int64)) for s in range(0, n_img, 8)]
    model.pose_recovery["syn"].check_asserts = False
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for bt in batches:
            model.predict(*bt, "syn")
        torch.cuda.synchronize()
        print(f"predict() back to back over the same {len(batches)} batches: {1e3 * (time.perf_counter() - t0) / len(batches):.2f} ms per batch")


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:3]))
