#!/bin/bash
# round 5, call 5: producers with relaxed polls + one fence: parity subset, N-rank bench tests, A/B, PMC traffic profile, full bench line
O=gpurun_out/r05_c5; mkdir -p $O
(timeout 400 python -m pytest tests/test_gpu_async_form.py tests/test_gpu_pipelined.py tests/test_gpu_baseline_configs.py -m gpu -q --no-header -x 2>&1 | tail -25) > $O/tests.log; tail -3 $O/tests.log
(timeout 500 python -m pytest tests/test_gpu_parallel.py -m gpu -q --no-header -x -k "bench" 2>&1 | tail -40) > $O/tests_bench.log; tail -5 $O/tests_bench.log
for v in "prod 128" "pre 0"; do set -- $v
  SNN_DC_PRODUCERS=$2 timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err
  python - $O/bench_$1.json $1 <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'enqueue', d['host_enqueue_ms_per_step'], 'kernel us', d['roofline']['avg_launch_us'], d['config']['plan_retries(lean,resident)'])
P
done
bash tools/profile_bench.sh r05 > $O/profile.log 2>&1; tail -2 $O/profile.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; tail -c 600 $O/bench_full.json
