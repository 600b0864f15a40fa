#!/bin/bash
# Round 6: same-box A/B of the step-loop forms of k_dc2015_async (SNN_DC_ASYNC_FORM=0: round 5's, 1: round 6's) behind the D&C parity tests.
#   bash tools/r06_ab.sh <tag> [quick]
TAG=${1:-ab}; O=gpurun_out/r06_$TAG; mkdir -p $O
export TMPDIR=/tmp
TESTS="tests/test_gpu_async_form.py tests/test_gpu_baseline_configs.py tests/test_gpu_fullsize.py tests/test_gpu_fused_stress.py tests/test_gpu_pipelined.py tests/test_gpu_resident_safety.py"
[ "$2" = "quick" ] && TESTS="tests/test_gpu_async_form.py tests/test_gpu_fullsize.py"
(timeout 900 python -m pytest $TESTS -m gpu -x -q --no-header 2>&1 | tail -15) > $O/dc_tests_form1.log; tail -3 $O/dc_tests_form1.log
for rep in 1 2; do
for form in 0 1; do
  for K in 20 200; do
    W=5; [ $K = 200 ] && W=10
    SNN_DC_ASYNC_FORM=$form timeout 200 python bench.py --steps $K --warmup $W --no-cpu-baseline > $O/bench_k${K}_form${form}_$rep.json 2> $O/bench_k${K}_form${form}_$rep.err
    python - $O/bench_k${K}_form${form}_$rep.json k$K form$form <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[2], sys.argv[3], d['value'], 'ms/step', d['ms_per_step'], 'kernel us', r['avg_launch_us'], 'frac', r['frac'], 'sync', (d.get('sync_runs') or {}).get('value'))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'FAILED', e)
P
  done
done
done
SNN_DC_ASYNC_FORM=1 timeout 300 python tools/timing_by_age.py --at 20,60 --wg 47 2> $O/timing_form1_wg47.txt > /dev/null
SNN_DC_ASYNC_FORM=1 timeout 300 python tools/timing_by_age.py --at 20,60 --wg 3 2> $O/timing_form1_wg3.txt > /dev/null
grep "dc2015 async" $O/timing_form1_wg47.txt | cut -c1-700
grep "dc2015 async form" $O/timing_form1_wg3.txt | cut -c1-700
