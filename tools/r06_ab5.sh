#!/bin/bash
# Round 6: same-box A/B of two step-loop experiments behind SNN_DC_SPECFLAGS (8: digest asked for two iterations ahead, 16: won-mask reads only behind a crossing)
TAG=${1:-ab5}; O=gpurun_out/r06_$TAG; mkdir -p $O
export TMPDIR=/tmp
(SNN_DC_SPECFLAGS=32 timeout 1200 python -m pytest tests/test_gpu_async_form.py tests/test_gpu_fullsize.py tests/test_gpu_baseline_configs.py tests/test_gpu_pipelined.py -m gpu -x -q --no-header 2>&1 | tail -5) > $O/dc_tests_flags32.log; tail -2 $O/dc_tests_flags32.log
run() {  # name, env...
  name=$1; shift
  for K in 20 200; do
    W=5; [ $K = 200 ] && W=10
    env "$@" timeout 200 python bench.py --steps $K --warmup $W --no-cpu-baseline > $O/bench_k${K}_$name.json 2> $O/bench_k${K}_$name.err
    python - $O/bench_k${K}_$name.json k$K $name <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[2], sys.argv[3], d['value'], 'ms/step', d['ms_per_step'], 'kernel us', r['avg_launch_us'], 'frac', r['frac'], 'sync', (d.get('sync_runs') or {}).get('value'))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'FAILED', e)
P
  done
}
for rep in 1 2; do
  for f in 0 32; do run flags${f}_$rep SNN_DC_SPECFLAGS=$f; done
done
