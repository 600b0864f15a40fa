#!/bin/bash
# GPU box: same-box A/B of the current library against another build of it (SNN_LIB_OVERRIDE), two-layer family
#   PREV=<path to .so> CFGS=cfg5,cfg3 bash tools/r06_ab_lib.sh
O=gpurun_out/r06_ab_lib; mkdir -p $O
export TMPDIR=/tmp
CFGS=${CFGS:-cfg5}
line() { python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    w = d.get('config'); w = w.get('workload', '') if isinstance(w, dict) else str(w)
    print('$1', w[:44], d.get('value') or d.get('timesteps_per_s'))
"; }
(timeout 1200 python -m pytest tests/test_gpu_twolayer.py tests/test_gpu_baseline_configs.py tests/test_gpu_fused_stress.py tests/test_gpu_rules.py -m gpu -x -q --no-header 2>&1 | tail -3) > $O/tests.log; tail -2 $O/tests.log
for rep in 1 2; do
    timeout 900 python tools/bench_configs.py --runs 5 --only $CFGS --no-cpu-baseline 2>/dev/null | line "new"
    [ -f "$PREV" ] && SNN_DEVELOPER=1 SNN_LIB_OVERRIDE=$PREV timeout 900 python tools/bench_configs.py --runs 5 --only $CFGS --no-cpu-baseline 2>/dev/null | line "prev"
done | tee $O/wall.log
SNN_TWO_TIMING=1 timeout 300 python tools/bench_configs.py --runs 2 --only cfg5 --no-cpu-baseline 2>&1 >/dev/null | grep "twolayer timing" | tail -4 | tee $O/timing.log
