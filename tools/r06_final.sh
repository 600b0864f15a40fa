#!/bin/bash
# Round 6's artefacts in one call ON THE GPU BOX (gpurun -- 'bash tools/r06_final.sh'): the whole GPU test tier, the driver's bench command with
# its CPU legs, the same command under rocprofv3 (kernel stats, per-launch durations, PMC traffic), every BASELINE config (wall clock +
# reference leg + parity; rocprofv3 + PMC), the (f) rules, the bounce rate on digit-like images, the ramp probe.  Copy what should be judged
# from gpurun_out/r06_final/ and gpurun_out/prof_* into profiles/.
O=gpurun_out/r06_final; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -q --no-header --durations=8 2>&1 | tail -20) > $O/gpu_suite.log; tail -3 $O/gpu_suite.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_k200.json 2> $O/bench_k200.err
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sync-runs > $O/bench_k20_sync.json 2> $O/bench_k20_sync.err
for f in k20 k200 k20_sync; do python - $O/bench_$f.json $f <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'enqueue', d['host_enqueue_ms_per_step'], 'kernel us', r['avg_launch_us'], 'n', r['launches_timed'], 'frac', r['frac'], 'traffic profile matches', (r.get('traffic_profile') or {}).get('matches_current_source'), 'parity', (d.get('parity') or {}).get('rasters_bit_exact'))
P
done
bash tools/profile_bench.sh r06 > $O/profile_bench.log 2>&1; tail -1 $O/profile_bench.log | cut -c1-400
bash tools/profile_round.sh r06round > $O/profile_round.log 2>&1
timeout 900 python tools/bench_configs.py --runs 5 > $O/bench_configs.jsonl 2> $O/bench_configs.err
timeout 200 python tools/bounce_rate.py --batches 8 > $O/bounce.json 2> $O/bounce.err
timeout 120 python tools/ramp_probe.py --runs 100 > $O/ramp_identical.json 2> $O/ramp.err
bash tools/pmc_issue_stats.sh > $O/issue_stats_sq_counters.txt 2> $O/issue_stats.err
timeout 300 python bench.py --config cfg3 --gpus 1 --steps 20 --warmup 3 > $O/bench_cfg3_n1.json 2> $O/bench_cfg3_n1.err
timeout 300 python bench.py --config cfg3 --gpus 2 --backend gloo --steps 20 --warmup 3 > $O/bench_cfg3_n2_gloo_one_gpu.json 2> $O/bench_cfg3_n2.err
SNN_TWO_TIMING=1 timeout 300 python tools/bench_configs.py --runs 2 --only cfg5 --no-cpu-baseline > /dev/null 2> $O/two_timing_cfg5.txt
SNN_TWO_TIMING=1 timeout 300 python tools/bench_configs.py --runs 2 --only cfg3 --no-cpu-baseline > /dev/null 2> $O/two_timing_cfg3_b128.txt
