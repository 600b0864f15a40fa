#!/bin/bash
# Round 6: same-box comparison of library variants bindsnet_amd/lib/libsnnhip_<name>.so (SNN_LIB_OVERRIDE), K=20 and K=200, two rounds
TAG=${1:-abv}; shift; O=gpurun_out/r06_$TAG; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  for name in "$@"; do
    for K in 20 200; do
      W=5; [ $K = 200 ] && W=10
      SNN_DEVELOPER=1 SNN_LIB_OVERRIDE=$PWD/bindsnet_amd/lib/libsnnhip_$name.so timeout 200 python bench.py --steps $K --warmup $W --no-cpu-baseline > $O/bench_k${K}_${name}_$rep.json 2> $O/bench_k${K}_${name}_$rep.err
      python - $O/bench_k${K}_${name}_$rep.json k$K ${name}_$rep <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[2], sys.argv[3], d['value'], 'ms/step', d['ms_per_step'], 'kernel us', r['avg_launch_us'], 'parity', (d.get('parity') or {}).get('rasters_bit_exact'))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'FAILED', e)
P
    done
  done
done
