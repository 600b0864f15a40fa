#!/bin/bash
# GPU box: two-layer run kernel with the digest through LDS-DMA (all learning instances) -- parity tests, wall clock against the previous
# library (SNN_LIB_OVERRIDE), phase timing
O=gpurun_out/r06_dma; mkdir -p $O
export TMPDIR=/tmp
line() { python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    w = d.get('config'); w = w.get('workload', '') if isinstance(w, dict) else str(w)
    print('$1', w[:44], d.get('value') or d.get('timesteps_per_s'), (d.get('parity') or {}).get('rasters_bit_exact'))
"; }
(timeout 1200 python -m pytest tests/test_gpu_twolayer.py tests/test_gpu_baseline_configs.py tests/test_gpu_fused_stress.py tests/test_gpu_rules.py tests/test_gpu_network.py -m gpu -x -q --no-header 2>&1 | tail -5) > $O/tests.log; tail -3 $O/tests.log
CFGS=cfg3_shard,cfg3_b32,cfg3,cfg5,f_postpre_ref,f_hebbian,f_wdpp
for rep in 1 2; do
    timeout 900 python tools/bench_configs.py --runs 5 --only $CFGS --no-cpu-baseline 2>/dev/null | line "new"
    [ -f "$PREV" ] && SNN_DEVELOPER=1 SNN_LIB_OVERRIDE=$PREV timeout 900 python tools/bench_configs.py --runs 5 --only $CFGS --no-cpu-baseline 2>/dev/null | line "prev"
    SNN_TWO_ENT2=0 timeout 900 python tools/bench_configs.py --runs 5 --only cfg3_shard,cfg3_b32,cfg3 --no-cpu-baseline 2>/dev/null | line "new, one event area"
done | tee $O/wall.log
for c in cfg5 cfg3 cfg3_shard; do SNN_TWO_TIMING=1 timeout 300 python tools/bench_configs.py --runs 2 --only $c --no-cpu-baseline 2>&1 >/dev/null | grep "twolayer timing" | tail -4; done | tee $O/timing.log
