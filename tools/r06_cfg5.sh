#!/bin/bash
# GPU box: cfg5 (dense MSTDP 6400 -> 500, B = 16) -- two-layer parity tests, wall clock, phase timing
O=gpurun_out/r06_cfg5; mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_twolayer.py tests/test_gpu_baseline_configs.py tests/test_gpu_fused_stress.py -m gpu -x -q --no-header 2>&1 | tail -3) > $O/tests.log; tail -2 $O/tests.log
for rep in 1 2; do timeout 600 python tools/bench_configs.py --runs 5 --only cfg5,cfg3_shard,cfg3_b32,cfg3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print((d.get('config') or {}).get('workload', '')[:40], d.get('value'), (d.get('parity') or {}).get('rasters_bit_exact'))
"; done
SNN_TWO_TIMING=1 timeout 300 python tools/bench_configs.py --runs 2 --only cfg5 --no-cpu-baseline 2>&1 >/dev/null | grep "twolayer timing" | tail -3
