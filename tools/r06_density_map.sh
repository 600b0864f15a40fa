#!/bin/bash
# GPU box: Connection.compute on the matrix cores vs event-driven over the INPUT DENSITY (1 / 5 / 20 / 50 %), the two learning shapes (cfg3 whole batch
# 128 x 784 -> 1600, cfg5 16 x 6400 -> 500): HIP-event timing + rocprofv3 counters (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES, FETCH_SIZE in a pass of
# its own) per density -> gpurun_out/r06_density_map/density_map.json (copy to profiles/r06_dense_mfma_density_map.json)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_density_map
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for d in 0.01 0.05 0.2 0.5; do
  python "$R/tools/bench_dense_prop.py" --density $d > "$OUT/timing_$d.jsonl" 2> "$OUT/timing_$d.err"
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d "$OUT/pmc_mfma_$d" -o p -- python "$R/tools/bench_dense_prop.py" --density $d --iters 20 > "$OUT/pmc_mfma_$d.log" 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$d" -o p -- python "$R/tools/bench_dense_prop.py" --density $d --iters 20 > "$OUT/pmc_fetch_$d.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {"what": "Connection.compute (topology.py:332-346) on v_mfma_f32_16x16x4_f32 (snn_prop_dense_mfma_f32, one k-ordered chain per 16x16 tile) vs the event-driven kernel the plans use "
               "(snn_prop_dense_f32), both bit-identical to the canonical ordered sum, over the input density; counters: rocprofv3 --pmc in separate passes "
               "(FETCH_SIZE alone; KB as rocprofv3 prints it, x2 for wide loads on gfx950)", "by_density": {}}
for d in ("0.01", "0.05", "0.2", "0.5"):
    timing = [json.loads(l) for l in open(f"{out}/timing_{d}.jsonl") if l.startswith("{")]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(f"{out}/pmc_*_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = "mfma" if "k_prop_dense_mfma" in r["Kernel_Name"] else ("event" if "k_prop" in r["Kernel_Name"] else None)
            if k:
                acc[k + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    ctr = {k: {c: sum(v) / len(v) for c, v in dd.items()} for k, dd in acc.items()}
    for k, dd in ctr.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in dd and dd.get("SQ_BUSY_CYCLES"):
            dd["mfma_busy_fraction_of_SQ_busy"] = dd["SQ_VALU_MFMA_BUSY_CYCLES"] / dd["SQ_BUSY_CYCLES"]
    res["by_density"][d] = {"timing": timing, "counters_per_launch": ctr}
json.dump(res, open(out + "/density_map.json", "w"), indent=1)
for d, v in res["by_density"].items():
    for t in v["timing"]:
        print(d, t["shape"], "event", t["event_driven_us"], "us, mfma", t["mfma_us"], "us, identical", t["identical_bits"])
PY
