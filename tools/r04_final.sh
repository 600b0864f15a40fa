#!/bin/bash
# Round 4 closing run on the GPU box: whole -m gpu suite, default bench line (with the CPU leg), profiles of the same command,
# cfg1 wall clock, literal eth_mnist.py timing
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_final; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q --no-header 2>&1 | tail -15 ) > $O/gpu_suite.log 2>&1
grep -a "passed\|failed" $O/gpu_suite.log | tail -2
( timeout 600 python bench.py 2>&1 | tail -1 ) > $O/bench_line.json 2>&1
python - <<'PY'
import json
l=open("gpurun_out/r04_final/bench_line.json").read().strip().split("\n")[-1]
try:
    d=json.loads(l); print("bench:", d["value"], "ms/step", d["ms_per_step"], "kernel us", d["roofline"]["avg_launch_us"], "frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"], "parity", d["parity"]["rasters_bit_exact"], d["parity"]["weights_bit_exact"])
except Exception as e: print("bench parse failed", e, l[-600:])
PY
bash tools/profile_bench.sh r04f > $O/profile_bench.log 2>&1
tail -1 $O/profile_bench.log | cut -c1-500
bash tools/pmc_issue_stats.sh > $O/issue_stats.txt 2>&1
grep "^SQ_" $O/issue_stats.txt | head -20
( timeout 200 python tools/bench_configs.py --only cfg1 2>&1 | tail -1 ) > $O/cfg1.log 2>&1; cat $O/cfg1.log | cut -c1-300
( timeout 400 python tools/eth_mnist_timing.py --impl amd --out $O/eth_mnist_literal_mi355x.json 2>&1 | tail -3 ) > $O/eth_timing.log 2>&1
tail -1 $O/eth_timing.log | cut -c1-900
