#!/bin/bash
# GPU box: the batch-constant instances of k_two_run against the general ones (SNN_TWO_BCONST=0), behind the two-layer parity tests
O=gpurun_out/r06_two_bconst; mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_twolayer.py tests/test_gpu_baseline_configs.py -m gpu -x -q --no-header 2>&1 | tail -3) > $O/tests.log; tail -2 $O/tests.log
for rep in 1 2; do for v in 0 1; do
  SNN_TWO_BCONST=$v timeout 600 python tools/bench_configs.py --runs 5 --only cfg3_shard,cfg3_b32,cfg3,cfg5 --no-cpu-baseline 2> $O/err_$v.txt | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('bconst $v rep $rep', (d.get('config') or {}).get('workload', '')[:40], d.get('value'), d.get('ms_per_step'))
"
done; done
