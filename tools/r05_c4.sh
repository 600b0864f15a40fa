#!/bin/bash
# round 5, call 4: producer workgroups inside the third-generation launch: parity, then same-box timing A/B against the two pre-pass launches
O=gpurun_out/r05_c4; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_async_form.py tests/test_gpu_pipelined.py tests/test_gpu_fused_stress.py tests/test_gpu_fuzz.py tests/test_gpu_baseline_configs.py tests/test_gpu_resident_safety.py -m gpu -q --no-header -x 2>&1 | tail -25) > $O/tests.log; tail -4 $O/tests.log
for v in "prod 128" "pre 0" "prod64 64" "prod 128" "pre 0"; do set -- $v
  SNN_DC_PRODUCERS=$2 timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err
  python - $O/bench_$1.json $1 <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'enqueue', d['host_enqueue_ms_per_step'], 'kernel us', d['roofline']['avg_launch_us'], d['config']['plan_retries(lean,resident)'])
P
done
