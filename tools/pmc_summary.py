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


def launch_durations(out, warmup=3, steps=20):
    """Per-launch durations of the dominant kernel from the kernel trace of `bench.py --steps 20 --warmup 3`, in launch order: rocprofv3's
    --stats average runs over EVERY launch of the process, and the first ones are slow for a reason that has nothing to do with the kernel
    (the untrained network crosses its thresholds far more often: 1.85, 1.31, 1.25 ms ...); bench.py's `roofline.avg_launch_us` is the HIP-event
    mean over the launches of its timed region, i.e. launches warmup .. warmup+steps-1 here."""
    for path in glob.glob(os.path.join(out, "ks", "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(path)) if KERNEL in r.get("Kernel_Name", "")]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        if len(d) >= warmup + steps:
            timed = d[warmup:warmup + steps]
            return {"launches": len(d), "all_launches_mean_us": round(sum(d) / len(d), 1), "warmup_launches_us": [round(x, 1) for x in d[:warmup]],
                    "timed_region_launches": [warmup, warmup + steps], "timed_region_mean_us": round(sum(timed) / len(timed), 1),
                    "timed_region_min_us": round(min(timed), 1), "timed_region_max_us": round(max(timed), 1),
                    "later_launches_mean_us": round(sum(d[warmup + steps:]) / max(1, len(d) - warmup - steps), 1)}
    return None


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
    res["kernel_trace_durations"] = launch_durations(out)
    with open(os.path.join(out, "pmc_hbm_traffic.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1])
