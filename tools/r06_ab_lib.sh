#!/bin/bash
# Round 6: same-box A/B of the in-tree library against bindsnet_amd/lib/libsnnhip_prev.so (the previous commit's build), behind the D&C parity tests
TAG=${1:-ablib}; O=gpurun_out/r06_$TAG; mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_async_form.py tests/test_gpu_fullsize.py tests/test_gpu_baseline_configs.py tests/test_gpu_pipelined.py tests/test_gpu_fused_stress.py tests/test_gpu_resident_safety.py tests/test_gpu_fuzz.py -m gpu -x -q --no-header 2>&1 | tail -5) > $O/dc_tests.log; tail -2 $O/dc_tests.log
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
  run prev_$rep SNN_DEVELOPER=1 SNN_LIB_OVERRIDE=$PWD/bindsnet_amd/lib/libsnnhip_prev.so
  run new_$rep SNN_DEVELOPER=0
done
