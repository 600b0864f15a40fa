#!/bin/bash
# Round 6, first GPU call: the round-6 step loop of k_dc2015_async (SNN_DC_ASYNC_FORM=1) against round 5's (=0) on one box:
# parity tests of the D&C plans on the new form, then same-box A/B of the driver's command, then the TIMING instance's report.
O=gpurun_out/r06_c1; mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_async_form.py tests/test_gpu_baseline_configs.py tests/test_gpu_fullsize.py tests/test_gpu_fused_stress.py tests/test_gpu_pipelined.py tests/test_gpu_resident_safety.py tests/test_gpu_network.py -m gpu -x -q --no-header 2>&1 | tail -15) > $O/dc_tests_form1.log; tail -3 $O/dc_tests_form1.log
for rep in 1 2; do
for form in 0 1; do
  SNN_DC_ASYNC_FORM=$form timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_k20_form${form}_$rep.json 2> $O/bench_k20_form${form}_$rep.err
  SNN_DC_ASYNC_FORM=$form timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_k200_form${form}_$rep.json 2> $O/bench_k200_form${form}_$rep.err
  for f in k20 k200; do python - $O/bench_${f}_form${form}_$rep.json $f form$form <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[2], sys.argv[3], d['value'], 'ms/step', d['ms_per_step'], 'kernel us', r['avg_launch_us'], 'frac', r['frac'], 'parity', (d.get('parity') or {}).get('rasters_bit_exact'))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'FAILED', e)
P
  done
done
done
for form in 0 1; do
  SNN_DC_ASYNC_FORM=$form timeout 300 python tools/timing_by_age.py --at 5,20,60 --wg 3 2> $O/timing_form${form}_wg3.txt > /dev/null
  SNN_DC_ASYNC_FORM=$form timeout 300 python tools/timing_by_age.py --at 5,20,60 --wg 47 2> $O/timing_form${form}_wg47.txt > /dev/null
done
grep "dc2015 async" $O/timing_form1_wg3.txt | head -4 | cut -c1-600
(timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q --no-header 2>&1 | tail -5) > $O/fuzz_form1.log; tail -2 $O/fuzz_form1.log
