#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/${1:-r04i}; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -40 ) > $O/gpu_suite.log 2>&1
tail -4 $O/gpu_suite.log
( timeout 200 python tools/bench_configs.py --only cfg1 2>&1 | tail -3 ) > $O/cfg1.log 2>&1
tail -1 $O/cfg1.log | cut -c1-300
( timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench.log 2>&1
OO=${1:-r04i} python - <<'PY'
import json,os
l=open(os.path.join("gpurun_out", os.environ.get("OO","r04i"), "bench.log")).read().strip().split("\n")[-1]
try:
    d=json.loads(l); print(d["value"], d["ms_per_step"], d.get("host_enqueue_ms_per_step"), d["roofline"].get("resident_form"), d["roofline"]["avg_launch_us"], d["config"]["plan_retries(lean,resident)"])
except Exception as e: print("bench parse failed", e, l[-400:])
PY
