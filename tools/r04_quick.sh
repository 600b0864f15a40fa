#!/bin/bash
# quick check of a kernel change: D&C parity tests (default form), timing marks, bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/${1:-r04q}; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_async_form.py tests/test_gpu_fuzz.py tests/test_gpu_fused_stress.py tests/test_gpu_resident_safety.py -m gpu -q --no-header -x -k "dc or lean or resident or stress or learning or additive or one_sided or missing or arbitration or generation or short or excit or diagonal" 2>&1 | tail -15 ) > $O/tests.log 2>&1
tail -3 $O/tests.log | grep -v amdgpu
( SNN_DC_TIMING=${2:-10} timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep "dc2015 async" | tail -2 ) > $O/timing.log 2>&1
cat $O/timing.log | cut -c1-700
( timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 ) > $O/bench.log 2>&1
OO=$O python - <<'PY'
import json,os
l=open(os.path.join(os.environ["OO"], "bench.log")).read().strip().split("\n")[-1]
try:
    d=json.loads(l); print("bench:", d["value"], "ms/step", d["ms_per_step"], "host", d.get("host_enqueue_ms_per_step"), "form", d["roofline"].get("resident_form"), "kernel us", d["roofline"]["avg_launch_us"], d["config"]["plan_retries(lean,resident)"])
except Exception as e: print("bench parse failed", e, l[-400:])
PY
