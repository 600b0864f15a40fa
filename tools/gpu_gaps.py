#!/usr/bin/env python3
"""Device-side timeline of the bench loop from a rocprofv3 --kernel-trace --memory-copy-trace run (csv): per network.run() -- found by
its dominant kernel -- the time from the end of the previous dominant launch to the start of this one (what the device did and how long it
sat idle in between), averaged over the later runs.  python tools/gpu_gaps.py <dir with *_kernel_trace.csv> [dominant-kernel-substring]"""
import csv
import glob
import sys

d = sys.argv[1]
dom = sys.argv[2] if len(sys.argv) > 2 else "k_dc2015_async"
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", r.get("Kind", "?"))))
ev.sort()
idx = [i for i, e in enumerate(ev) if dom in e[2]]
print(f"{len(ev)} device operations, {len(idx)} launches of {dom}")
gaps, busy, per = [], [], {}
for a, b in zip(idx[len(idx) // 2:-1], idx[len(idx) // 2 + 1:]):
    g = ev[b][0] - ev[a][1]
    w = sum(e[1] - e[0] for e in ev[a + 1:b])
    gaps.append(g); busy.append(w)
    for e in ev[a + 1:b]:
        per.setdefault(e[2], []).append(e[1] - e[0])
n = max(1, len(gaps))
print(f"between two dominant launches: {sum(gaps) / n / 1e3:.1f} us, of which device operations {sum(busy) / n / 1e3:.1f} us, idle {(sum(gaps) - sum(busy)) / n / 1e3:.1f} us")
print(f"dominant kernel: {sum(ev[i][1] - ev[i][0] for i in idx[len(idx) // 2:]) / max(1, len(idx) - len(idx) // 2) / 1e3:.1f} us")
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:48s} x{len(v) / n:5.2f} per run, {sum(v) / n / 1e3:7.2f} us per run")
if len(idx) > 3:                         # one run in sequence
    a, b = idx[-3], idx[-2]
    t0 = ev[a][1]
    for e in ev[a:b + 1]:
        print(f"    +{(e[0] - t0) / 1e3:9.1f} .. +{(e[1] - t0) / 1e3:9.1f} us  {e[2]}")
