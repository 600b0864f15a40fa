#!/bin/bash
# round 5, call 3: pipelined sections + merged pre-pass: tests, then same-box timing A/B
O=gpurun_out/r05_c3; mkdir -p $O
(timeout 400 python -m pytest tests/test_gpu_pipelined.py tests/test_gpu_vector_thresh.py -m gpu -q --no-header -x 2>&1 | tail -25) > $O/tests_new.log; tail -4 $O/tests_new.log
(timeout 500 python -m pytest tests/test_gpu_async_form.py tests/test_gpu_fused_stress.py tests/test_gpu_fuzz.py tests/test_gpu_baseline_configs.py tests/test_gpu_resident_safety.py tests/test_gpu_ops.py -m gpu -q --no-header -x 2>&1 | tail -15) > $O/tests_dc.log; tail -3 $O/tests_dc.log
for v in "pipe" "sync --sync-runs" ; do set -- $v
  timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - $O/bench_$1.json $1 <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'enqueue', d['host_enqueue_ms_per_step'], 'kernel us', d['roofline']['avg_launch_us'], d['config']['plan_retries(lean,resident)'])
P
done
SNN_DC_PRE1=0 timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_pipe_pre0.json 2> $O/bench_pipe_pre0.err
python - $O/bench_pipe_pre0.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('pipe, two pre-pass launches', d['value'], 'ms/step', d['ms_per_step'], 'kernel us', d['roofline']['avg_launch_us'])
P
timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_pipe_200.json 2> $O/bench_pipe_200.err
python - $O/bench_pipe_200.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('pipe 200 steps', d['value'], 'ms/step', d['ms_per_step'], 'enqueue', d['host_enqueue_ms_per_step'])
P
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OLDPWD/$O/prof_bench.json 2> $OLDPWD/$O/prof.err
cd $OLDPWD; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {}
