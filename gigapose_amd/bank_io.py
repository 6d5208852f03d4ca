"""On-disk template bank: onboard once, then start from a file.

The reference re-runs the ViT and the ResNet over every template at start-up (`set_template_data`,
gigaPose.py:357-398; 57 s per object on 8 CPU threads) and only caches raw template pixels as `.npz`
(src/dataloader/template.py:55-81).  Here the ONBOARDED bank -- the matcher-ready features exactly as the kernels
consume them (f32 k-major in "chain" numerics, f16 hi/lo planes in "split"), patch masks, IST features, template
geometry -- is one flat file:

    bytes 0..7     magic  b"GPBANK01"
    bytes 8..15    little-endian u64: length L of the JSON header
    bytes 16..16+L JSON: {"numerics", "O", "N", "C", "sections": {name: {"offset", "shape", "dtype"}}}
    sections       raw little-endian arrays, each starting at a multiple of 4096

Every per-template section is laid out (O, N, ...), so the templates [lo, hi) a rank owns under template sharding
(sharding.shard_bounds) are O contiguous byte ranges: `load_bank(..., shard=(rank, world))` maps the file and uploads
only those (LM-O, 8 ranks: 21 of 162 templates per object; HANDAL-scale 40 objects: 0.85 GB of 6.8 GB per rank).
Loading replaces `set_template_data`; results are identical to onboarding in the same process (tests/test_gpu_bank_io.py).
"""
import json
import struct

import numpy as np
import pandas as pd
import torch

from .matching import MatchBank
from .poses import ObjectPoseRecovery
from .tensor_collection import PandasTensorCollection

MAGIC = b"GPBANK01"
ALIGN = 4096
_NP = {"float32": np.float32, "float64": np.float64, "float16": np.float16, "int64": np.int64, "uint8": np.uint8}


def write_sections(path, meta, arrays):
    """arrays: {name: np.ndarray}.  Returns the header dict written."""
    sections, off = {}, 0
    for name, a in arrays.items():
        a = np.ascontiguousarray(a)
        sections[name] = {"offset": off, "shape": list(a.shape), "dtype": str(a.dtype)}
        off += (a.nbytes + ALIGN - 1) // ALIGN * ALIGN
    header = dict(meta, sections=sections)
    blob = json.dumps(header).encode()
    base = (16 + len(blob) + ALIGN - 1) // ALIGN * ALIGN
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<Q", len(blob)) + blob)
        for name, a in arrays.items():
            f.seek(base + sections[name]["offset"])
            f.write(np.ascontiguousarray(a).tobytes())
        f.truncate(base + off)
    return header


def read_header(path):
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not a gigapose_amd bank file")
        (n,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(n).decode())
    header["_base"] = (16 + n + ALIGN - 1) // ALIGN * ALIGN
    return header


def map_section(path, header, name):
    s = header["sections"][name]
    return np.memmap(path, dtype=_NP[s["dtype"]], mode="r", offset=header["_base"] + s["offset"], shape=tuple(s["shape"]))


@torch.no_grad()
def save_bank(model, dataset_name, path):
    """Write the bank `model.set_template_data(dataset_name)` built (unsharded model)."""
    if model.template_shard is not None:
        raise ValueError("save the bank from an unsharded model; shards are cut at load time")
    td, bank = model.template_datas[dataset_name], model.match_banks[dataset_name]
    arrays = {"masks": bank.masks.cpu().numpy(), "mask224": td.mask.cpu().numpy().astype(np.uint8),
              "ist_features": td.ist_features.cpu().numpy(), "K": td.K.cpu().numpy(), "M": td.M.cpu().numpy(),
              "poses": td.poses.cpu().numpy()}
    if bank.numerics == "split":
        arrays["match_hi"] = bank.hi.cpu().numpy()
        if bank.lo is not None:            # bank_dtype "f16": the file holds the hi plane only (half the bytes)
            arrays["match_lo"] = bank.lo.cpu().numpy()
    else:
        arrays["match_f32"] = bank.features.cpu().numpy()
    # the ViT's plane-scale calibration (max |x| per (layer, tensor), vit.py) travels with the bank: a model started from the file
    # has never seen the templates, and would otherwise find its outlier tensors by tripping the range guard on the first query
    vit = getattr(getattr(model, "ae_net", None), "dinov2_model", None)
    if bank.numerics == "split" and getattr(vit, "plane_amax", None) is not None:
        arrays["vit_plane_amax"] = np.asarray(vit.plane_amax, dtype=np.float64)
    return write_sections(path, dict(numerics=bank.numerics, bank_dtype=getattr(bank, "bank_dtype", "f32"), O=bank.O, N=bank.N, C=bank.C), arrays)


@torch.no_grad()
def load_bank(model, dataset_name, path, shard=None, group=None):
    """Make `dataset_name` ready for predict() from a bank file.  shard=(rank, world): keep only this rank's template
    slice of the matcher features (the small per-template tensors -- masks, IST bank, geometry -- stay whole, as in
    set_template_data)."""
    from .sharding import ShardedMatcher, shard_bounds

    h = read_header(path)
    if h["numerics"] != model.testing_metric.numerics:
        raise ValueError(f"bank was saved in '{h['numerics']}' numerics, the model runs '{model.testing_metric.numerics}'")
    dev = model.device
    O, N = h["O"], h["N"]
    lo, hi = (0, N) if shard is None else shard_bounds(N, shard[1], shard[0])

    def up(name, sl=slice(None)):
        return torch.from_numpy(np.array(map_section(path, h, name)[:, sl])).to(dev)   # copy out of the read-only map

    bank = MatchBank.__new__(MatchBank)
    bank.numerics, bank.O, bank.N, bank.C = h["numerics"], O, hi - lo, h["C"]
    bank.features = bank.hi = bank.lo = None
    bank.bank_dtype = h.get("bank_dtype", "f32")
    if h["numerics"] == "split":
        bank.hi = up("match_hi", slice(lo, hi))
        bank.lo = up("match_lo", slice(lo, hi)) if "match_lo" in h["sections"] else None
    else:
        bank.features = up("match_f32", slice(lo, hi))
    bank.masks = up("masks", slice(lo, hi))
    data = {"mask": up("mask224").float(), "K": up("K"), "M": up("M"), "poses": up("poses"), "ist_features": up("ist_features")}
    model.template_datas[dataset_name] = PandasTensorCollection(infos=pd.DataFrame(), **data)
    if shard is None:
        model.match_banks[dataset_name] = bank
    else:
        model.template_shard = (shard[0], shard[1], group)
        model.match_banks[dataset_name] = ShardedMatcher(model.testing_metric, bank, lo, group)
    model.pose_recovery[dataset_name] = ObjectPoseRecovery(template_K=data["K"], template_Ms=data["M"],
                                                           template_poses=data["poses"])
    vit = getattr(getattr(model, "ae_net", None), "dinov2_model", None)
    if "vit_plane_amax" in h["sections"] and hasattr(vit, "adopt_plane_amax"):
        vit.adopt_plane_amax(np.array(map_section(path, h, "vit_plane_amax")))   # running maximum with what the model already knows
    return h
