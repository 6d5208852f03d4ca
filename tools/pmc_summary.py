"""Summarise rocprofv3 --pmc passes (csv): per kernel, per counter: dispatches, mean value per dispatch.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for
wide coalesced reads, so the corrected byte figure is 2 x (MI355X_MICROARCH.md, "HBM").
usage: python tools/pmc_summary.py <dir with <pass>/**/p_counter_collection.csv>"""
import collections
import csv
import glob
import os
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:64]


FAMILIES = [("gemm_kmajor", ("gemm_kmajor_kernel", "gemm_streamk_kernel")), ("match_tiles", ("match_tiles_kernel",)),
            ("attention", ("attention_kernel", "attention_planes_kernel")), ("attention_split", ("attention_split_kernel",)), ("layernorm", ("layernorm",)), ("conv", ("conv_kernel", "conv3x3_kernel")),
            ("gemm_split", ("gemm_split_kernel", "gemm_split256_kernel", "gemm_planes256_kernel")), ("gemm_planes", ("gemm_planes256_kernel",)),
            ("match_split", ("match_tiles_split_kernel",)), ("conv_split", ("conv_split_kernel", "conv_planes_kernel"))]


def family_json(acc, path):
    """Per kernel family: launches, mean corrected HBM bytes per launch (2 x FETCH_SIZE KiB + WRITE_SIZE KiB)."""
    import json

    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over one bench step; FETCH_SIZE (KiB) x2 = the "
                   "gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE (KiB) uncalibrated; "
                   "Infinity-Cache hits are counted as fetches"}
    for fam, pats in FAMILIES:
        fetch, write = [], []
        for (name, _grid), counters in acc.items():
            if any(pt in name for pt in pats):
                fetch += counters.get("FETCH_SIZE", [])
                write += counters.get("WRITE_SIZE", [])
        if fetch or write:
            f = (sum(fetch) / len(fetch)) * 1024 * 2 if fetch else None
            w = (sum(write) / len(write)) * 1024 if write else None
            out[fam] = {"launches": max(len(fetch), len(write)), "fetch_bytes_per_launch": f, "write_bytes_per_launch": w,
                        "hbm_bytes_per_launch": (f or 0) + (w or 0)}
    # when, and on which kernels: bench.py quotes both next to the (static) traffic figure
    import glob, hashlib, os, time
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gigapose_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        h.update(open(f, "rb").read())
    out["_recorded_at"] = time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime())
    out["_kernel_sources_sha1"] = h.hexdigest()
    json.dump(out, open(path, "w"), indent=1)


def main(root, json_path=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(path)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = (short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "")))
        for (d, c), v in per_dispatch.items():
            acc[names[d]][c].append(v)
    for k in sorted(acc):
        if not any(t in k[0] for t in ("gemm", "match", "conv", "attention", "layernorm", "split")):
            continue
        print(f"{k[0]}  grid={k[1]}")
        for c, v in sorted(acc[k].items()):
            extra = ""
            if c == "FETCH_SIZE":
                extra = f"  -> {sum(v)/len(v)*1024*2/1e6:.1f} MB/launch after the gfx950 x2 correction"
            if c == "WRITE_SIZE":
                extra = f"  -> {sum(v)/len(v)*1024/1e6:.1f} MB/launch (uncalibrated)"
            print(f"    {c:34s} n={len(v):3d} mean={sum(v)/len(v):.4g}{extra}")
    if json_path:
        family_json(acc, json_path)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--json" else None)
