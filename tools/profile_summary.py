#!/usr/bin/env python3
"""Summarise tools/profile_round.sh: per config the dominant kernel's rocprofv3 average duration, the HBM bytes per launch
from the FETCH_SIZE / WRITE_SIZE passes, and the roofline fraction under SURVEY.md 8(d)'s algorithmic bytes per timestep."""
import csv
import glob
import json
import os
import shutil
import sys

# config -> (kernel name fragment, timesteps per launch, algorithmic bytes per timestep [SURVEY 8(d)])
CFG = {
    "cfg1": ("k_dc2015_async", 250, 1_030_000),
    "cfg2": ("k_dc2015_async", 250, 5_860_480),
    "cfg3_shard": ("k_two_run", 100, 4 * 3 * 784 * 1600 + 16 * (784 * 10 + 1600 * 26)),
    "cfg3_b32": ("k_two_run", 100, 4 * 3 * 784 * 1600 + 32 * (784 * 10 + 1600 * 26)),
    "cfg3": ("k_two_run", 100, 21_400_000),
    "cfg4": ("k_convlif_run", 250, 20_100_000),
    "cfg5": ("k_two_run", 100, 40_400_000),
    "f_hebbian": ("k_two_run", 100, 4 * 3 * 784 * 1600 + 32 * (784 * 10 + 1600 * 26)),      # (the bytes of cfg3 at B = 32: same graph, another rule)
    "f_wdpp": ("k_two_run", 100, 4 * 3 * 784 * 1600 + 32 * (784 * 10 + 1600 * 26)),
}
HBM_PEAK = 8000.0


def stats_row(folder, frag):
    best = None
    for path in glob.glob(os.path.join(folder, "ks", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if frag in row["Name"] and (best is None or float(row["TotalDurationNs"]) > float(best["TotalDurationNs"])):
                best = row
    return best


def dominant(folder):
    for path in glob.glob(os.path.join(folder, "ks", "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(path)))
        return max(rows, key=lambda r: float(r["TotalDurationNs"])) if rows else None
    return None


def counter_mean(folder, frag, counter):
    vals = []
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if frag in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals)) if vals else None


def main(out):
    summary = {}
    for cfg, (frag, T, algo) in CFG.items():
        folder = os.path.join(out, cfg)
        if not os.path.isdir(folder):
            continue
        row = stats_row(folder, frag)
        dom = dominant(folder)
        for path in glob.glob(os.path.join(folder, "ks", "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(path, os.path.join(out, f"kernel_stats_{cfg}.csv"))
        entry = {"kernel_fragment": frag, "dominant_kernel_by_total_time": dom["Name"][:80] if dom else None}
        if row is None:                                   # the plan launched per-operator kernels instead of the fused one
            entry["note"] = "fused kernel not launched (generic plan)"
            summary[cfg] = entry
            continue
        avg_us = float(row["AverageNs"]) / 1e3
        fetch = counter_mean(os.path.join(folder, "fetch"), frag, "FETCH_SIZE")
        write = counter_mean(os.path.join(folder, "write"), frag, "WRITE_SIZE")
        ach = algo * T / (avg_us * 1e-6) / 1e9
        entry.update({"kernel": row["Name"][:100], "launches": int(row["Calls"]), "avg_launch_us": round(avg_us, 2),
                      "timesteps_per_launch": T, "us_per_timestep": round(avg_us / T, 3),
                      "algorithmic_bytes_per_timestep": algo, "achieved_GBps": round(ach, 1),
                      "roofline_frac_of_8TBps": round(ach / HBM_PEAK, 4),
                      "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write})
        if fetch is not None and write is not None:
            entry["hbm_bytes_per_launch_raw"] = int((fetch + write) * 1024)
            entry["hbm_bytes_per_launch_gfx950_corrected"] = int((2 * fetch + write) * 1024)
            entry["traffic_over_algorithmic"] = round((2 * fetch + write) * 1024 / (algo * T), 4)
        summary[cfg] = entry
    summary["_note"] = ("rocprofv3 --kernel-trace --stats for durations; FETCH_SIZE / WRITE_SIZE from separate --pmc passes; gfx950 "
                        "reports FETCH_SIZE at half the bytes of wide coalesced reads, hence corrected = 2*FETCH + WRITE "
                        "(MI355X_MICROARCH.md, HBM section).  Algorithmic bytes: SURVEY.md 8(d) dense accounting.")
    with open(os.path.join(out, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
