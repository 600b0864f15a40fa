#!/bin/bash
# Round 6: the whole GPU tier + the driver's bench command (K=20, W=5) + K=200 on one box.   bash tools/r06_suite.sh <tag>
TAG=${1:-suite}; O=gpurun_out/r06_$TAG; mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q --no-header --durations=8 2>&1 | tail -25) > $O/gpu_suite.log; tail -3 $O/gpu_suite.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_k200.json 2> $O/bench_k200.err
for f in k20 k200; do python - $O/bench_$f.json $f <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'kernel us', r['avg_launch_us'], 'frac', r['frac'], 'sync', (d.get('sync_runs') or {}).get('value'), 'parity', (d.get('parity') or {}).get('rasters_bit_exact'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
P
done
