#!/bin/bash
# GPU box: two-layer run kernel, tile width (columns per workgroup) -- cfg5 picks CW = 4 (125 workgroups on 256 CUs); CW = 2 gives 250
O=gpurun_out/r06_cw; mkdir -p $O
export TMPDIR=/tmp
line() { python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('$1', (d.get('config') or {}).get('workload', '')[:40], d.get('value'), (d.get('parity') or {}).get('rasters_bit_exact'))
"; }
for rep in 1 2; do
  for cw in 0 2 1; do
    SNN_TWO_CW=$cw timeout 600 python tools/bench_configs.py --runs 5 --only cfg5 --no-cpu-baseline 2>/dev/null | line "CW=$cw"
  done
done | tee $O/cfg5.log
for cw in 0 4; do
  SNN_TWO_CW=$cw timeout 600 python tools/bench_configs.py --runs 5 --only cfg3_shard,cfg3_b32,cfg3 --no-cpu-baseline 2>/dev/null | line "CW=$cw"
done | tee $O/cfg3.log
(SNN_TWO_CW=2 timeout 900 python -m pytest tests/test_gpu_twolayer.py tests/test_gpu_baseline_configs.py -m gpu -x -q --no-header 2>&1 | tail -3) > $O/tests_cw2.log; tail -2 $O/tests_cw2.log
SNN_TWO_CW=2 SNN_TWO_TIMING=1 timeout 300 python tools/bench_configs.py --runs 2 --only cfg5 --no-cpu-baseline 2>&1 >/dev/null | grep "twolayer timing" | tail -3 | tee $O/timing_cw2.log
