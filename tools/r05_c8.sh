#!/bin/bash
# round 5, call 8: ordinary launches inside sections, row-major Hebbian / WDPP: parity, then timing; every BASELINE config to the 8(d) standard
O=gpurun_out/r05_c8; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_rules.py tests/test_gpu_twolayer.py tests/test_gpu_pipelined.py tests/test_gpu_async_form.py tests/test_gpu_baseline_configs.py tests/test_gpu_resident_safety.py -m gpu -q --no-header -x 2>&1 | tail -25) > $O/tests.log; tail -4 $O/tests.log
for v in "k200 200 10" "k20 20 5"; do set -- $v
  timeout 200 python bench.py --steps $2 --warmup $3 --no-cpu-baseline > $O/bench_$1.json 2> $O/bench_$1.err
  python - $O/bench_$1.json $1 <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'enqueue', d['host_enqueue_ms_per_step'], 'kernel us', d['roofline']['avg_launch_us'], d['config']['plan_retries(lean,resident)'])
P
done
timeout 300 python tools/bench_configs.py --runs 5 --only f_postpre_ref,f_hebbian,f_wdpp > $O/f_rules.jsonl 2> $O/f_rules.err; cat $O/f_rules.jsonl
SNN_TWO_ROWMAJOR=0 timeout 300 python tools/bench_configs.py --runs 5 --only f_hebbian,f_wdpp > $O/f_rules_rowmajor0.jsonl 2>> $O/f_rules.err; cat $O/f_rules_rowmajor0.jsonl
timeout 900 python tools/bench_configs.py --runs 5 --only cfg1,cfg3_shard,cfg3_b32,cfg3,cfg4,cfg5 > $O/bench_configs.jsonl 2> $O/bench_configs.err
python - $O/bench_configs.jsonl <<'P'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(d['config']['workload'][:14], d['value'], 'sync', d['sync_runs']['timesteps_per_s'], d['config']['plan'], 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', {k:(d.get('parity') or {}).get(k) for k in ('rasters_bit_exact','max_abs_dW','weights_bit_exact')})
P
tail -3 $O/bench_configs.err
