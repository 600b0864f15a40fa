#!/bin/bash
# round 5, call 7: launch-mode A/B of a section's steady state, strokes fixture on all plans, bounce rate, slowest tests
O=gpurun_out/r05_c7; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_fuzz.py tests/test_gpu_pipelined.py tests/test_gpu_async_form.py tests/test_gpu_parallel.py -m gpu -q --no-header -x --durations=12 2>&1 | tail -30) > $O/tests.log; tail -22 $O/tests.log
for v in "coop_coop 1 1" "coop_plain 1 0" "plain_plain 0 0" "coop_coop 1 1" "coop_plain 1 0"; do set -- $v
  SNN_DC_COOP=$2 SNN_DC_GATED_COOP=$3 timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err
  python - $O/bench_$1.json $1 <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'enqueue', d['host_enqueue_ms_per_step'], 'kernel us', d['roofline']['avg_launch_us'], d['config']['plan_retries(lean,resident)'])
P
done
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_k20.json 2> $O/bench_k20.err; python -c "
import json; d=json.loads(open('$O/bench_k20.json').read().strip().splitlines()[-1]); print('k20', d['value'], d['ms_per_step'])"
timeout 200 python tools/bounce_rate.py --batches 8 > $O/bounce.json 2> $O/bounce.err; cat $O/bounce.json
