#!/bin/bash
# Round 6: same-box A/B of the step-loop switches.   bash tools/r06_ab3.sh <tag> [full]
TAG=${1:-ab3}; O=gpurun_out/r06_$TAG; mkdir -p $O
export TMPDIR=/tmp
TESTS="tests/test_gpu_async_form.py tests/test_gpu_fullsize.py tests/test_gpu_baseline_configs.py tests/test_gpu_pipelined.py"
[ "$2" = "full" ] && TESTS="$TESTS tests/test_gpu_fused_stress.py tests/test_gpu_resident_safety.py tests/test_gpu_fuzz.py tests/test_gpu_network.py"
(timeout 1200 python -m pytest $TESTS -m gpu -x -q --no-header 2>&1 | tail -15) > $O/dc_tests_form1.log; tail -3 $O/dc_tests_form1.log
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
  run form0_$rep SNN_DC_ASYNC_FORM=0
  run form1_$rep SNN_DC_ASYNC_FORM=1
  run form1_nodefer_$rep SNN_DC_ASYNC_FORM=1 SNN_DC_ASYNC_DEFER=0
  run form1_noldstrace_$rep SNN_DC_ASYNC_FORM=1 SNN_DC_ASYNC_LDSTRACE=0
done
for wg in 47 30; do
SNN_DC_ASYNC_FORM=1 timeout 300 python tools/timing_by_age.py --at 20,60 --wg $wg 2> $O/timing_form1_wg$wg.txt > /dev/null
grep "dc2015 async" $O/timing_form1_wg$wg.txt | cut -c1-700
done
