#!/usr/bin/env python3
"""Summarise tools/profile_bench.sh output: kernel stats CSV + HBM bytes per launch of the dominant kernel."""
import csv
import glob
import json
import os
import shutil
import sys

KERNEL = "k_dc2015_async"    # the default plan's kernel (third-generation lean form; second: k_dc2015_spec, first: k_dc2015_run)
T, ALGO_PER_STEP = 250, 5_860_480          # bench.py: timesteps per launch, SURVEY 8(d) bytes per timestep (cfg2)


def counter_mean(folder, counter):
    vals = []
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if KERNEL in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def source_sha16():
    """sha256 (first 16 hex digits) of the sources the dominant kernel is built from: bench.py compares it with the tree it runs in, so a
    `roofline.traffic` taken from a profile of an OLDER kernel says so (the snapshot on the GPU box has no .git to name a commit)."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("snn_dc2015_async.hip", "snn_dc2015.hpp", "snn_dc2015_tile.hpp"):
        with open(os.path.join(root, "bindsnet_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main(out):
    for path in glob.glob(os.path.join(out, "ks", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(path, os.path.join(out, "kernel_stats.csv"))
    fetch, n = counter_mean(os.path.join(out, "fetch"), "FETCH_SIZE")
    write, _ = counter_mean(os.path.join(out, "write"), "WRITE_SIZE")
    res = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline",
           "kernel": KERNEL, "kernel_source_sha16": source_sha16(), "launches": n, "timesteps_per_launch": T,
           "FETCH_SIZE_KB_per_launch_mean": fetch, "WRITE_SIZE_KB_per_launch_mean": write}
    if fetch is not None and write is not None:
        raw = (fetch + write) * 1024.0
        corr = (2.0 * fetch + write) * 1024.0
        res.update({"hbm_bytes_per_launch_raw": int(raw), "hbm_bytes_per_launch_gfx950_corrected": int(corr),
                    "algorithmic_bytes_per_launch": ALGO_PER_STEP * T,
                    "note": "gfx950 rocprofv3 reports FETCH_SIZE at half the bytes of wide coalesced reads (MI355X_MICROARCH.md, "
                            "HBM section): corrected = 2*FETCH + WRITE; access widths are mixed here, so the true value lies "
                            "between raw and corrected."})
    with open(os.path.join(out, "pmc_hbm_traffic.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1])
