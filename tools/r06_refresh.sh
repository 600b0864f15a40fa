#!/bin/bash
# GPU box: the headline artefacts only (after a change of k_dc2015_async): driver's bench command + K=200 + sync leg, kernel stats + PMC traffic
O=gpurun_out/r06_refresh; mkdir -p $O
export TMPDIR=/tmp
bash tools/profile_bench.sh r06 > $O/profile_bench.log 2>&1; tail -1 $O/profile_bench.log | cut -c1-300
cp gpurun_out/prof_r06/pmc_hbm_traffic.json profiles/r06_lean_pmc_hbm_traffic.json     # (on the box: the bench lines below quote it; copy the same file into profiles/ at home)
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_k200.json 2> $O/bench_k200.err
for f in k20 k200; do python - $O/bench_$f.json $f <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'kernel us', r['avg_launch_us'], 'frac', r['frac'], 'sync', (d.get('sync_runs') or {}).get('value'), 'traffic profile matches', (r.get('traffic_profile') or {}).get('matches_current_source'), 'parity', (d.get('parity') or {}).get('rasters_bit_exact'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
P
done
