#!/bin/bash
# GPU box: cfg5 (dense MSTDP 6400 -> 500, B = 16) -- two-layer parity tests, wall clock, phase timing of the current build
O=gpurun_out/r06_few; mkdir -p $O
export TMPDIR=/tmp
line() { python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('$1', (d.get('config') or {}).get('workload', '')[:40], d.get('value'), (d.get('parity') or {}).get('rasters_bit_exact'))
"; }
(timeout 900 python -m pytest tests/test_gpu_twolayer.py tests/test_gpu_baseline_configs.py tests/test_gpu_fused_stress.py -m gpu -x -q --no-header 2>&1 | tail -5) > $O/tests.log; tail -3 $O/tests.log
for rep in 1 2; do
    timeout 600 python tools/bench_configs.py --runs 5 --only cfg5 --no-cpu-baseline 2>/dev/null | line "now"
done | tee $O/cfg5.log
SNN_TWO_TIMING=1 timeout 300 python tools/bench_configs.py --runs 2 --only cfg5 --no-cpu-baseline 2>&1 >/dev/null | grep "twolayer timing" | tail -4 | tee $O/timing.log
