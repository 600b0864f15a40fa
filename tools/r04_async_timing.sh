#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp SNN_DC_ASYNC=1
O=gpurun_out/${1:-r04d}; mkdir -p $O
( SNN_DC_TIMING=${2:-10} timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep "dc2015 async" | tail -8 ) > $O/timing.log 2>&1
( timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>&1 | tail -2 ) > $O/bench.log 2>&1
cat $O/timing.log | cut -c1-900
BL=$O/bench.log python - <<'PY'
import json,sys
import os
l=open(os.environ.get("BL","gpurun_out/r04d/bench.log")).read().strip().split("\n")[-1]
try:
    d=json.loads(l); print(d["value"], d["roofline"].get("resident_form"), d["roofline"]["avg_launch_us"], d["parity"]["rasters_bit_exact"], d["parity"]["weights_bit_exact"], d["config"]["plan_retries(lean,resident)"])
except Exception as e: print("bench parse failed", e, l[-600:])
PY
