#!/bin/bash
# Developer aid (GPU box): kernel + memory-copy trace of a short bench run; prints the GPU timeline of one bench step.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_step
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT" -o tr -- python "$R/bench.py" --steps 6 --warmup 3 --no-cpu-baseline > "$OUT/run.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
ev = []
for path in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for path in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
runs = [i for i, e in enumerate(ev) if "k_dc2015_spec" in e[2] or "k_dc2015_run" in e[2]]
if len(runs) >= 4:
    a, b = runs[-3], runs[-2]
    t0 = ev[a][1]
    print("timeline between the ends of two consecutive resident launches (us from the first one's end):")
    for s, e, n in ev[a + 1:b + 1]:
        print(f"  start {(s - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f}  {n}")
PY
