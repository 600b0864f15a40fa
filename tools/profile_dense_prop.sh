#!/bin/bash
# GPU box: the MFMA-vs-event-driven decision for Connection.compute, with counters (run through gpurun):
#   bash tools/profile_dense_prop.sh   -> gpurun_out/dense_prop/{timing.jsonl, mfma_busy.json}
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/dense_prop
rm -rf "$OUT"; mkdir -p "$OUT"
python "$R/tools/bench_dense_prop.py" > "$OUT/timing.jsonl" 2> "$OUT/timing.err"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d "$OUT/pmc_mfma" -o p -- python "$R/tools/bench_dense_prop.py" --only-mfma --iters 20 > "$OUT/pmc_mfma.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- python "$R/tools/bench_dense_prop.py" --iters 20 > "$OUT/pmc_fetch.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = "mfma" if "k_prop_dense_mfma" in r["Kernel_Name"] else ("event" if "k_prop<" in r["Kernel_Name"] or "k_prop" in r["Kernel_Name"] else None)
        if k:
            acc[k + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in res.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("SQ_BUSY_CYCLES"):
        d["mfma_busy_fraction_of_SQ_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"]
json.dump(res, open(out + "/mfma_busy.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cat "$OUT/timing.jsonl"
