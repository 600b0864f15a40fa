#!/bin/bash
O=gpurun_out/r06_c16; mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_zz_exact_mode.py -m gpu -q --no-header 2>&1 | tail -4) | tee $O/ops_tests.log
bash tools/r06_density_map.sh 2>&1 | tail -10
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_k20.json 2> $O/bench_k20.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r06_c16/bench_k20.json").read().strip().splitlines()[-1]); r=d['roofline']
print('k20', d['value'], 'kernel us', r['avg_launch_us'], 'frac', r['frac'], 'sync', d['sync_runs'], 'matches', r['traffic_profile'].get('matches_current_source'), r['traffic'], r['traffic_raw'])
P
bash tools/r06_soak.sh
