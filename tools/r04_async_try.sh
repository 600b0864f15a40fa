#!/bin/bash
# Round 4: the third-generation lean kernel (SNN_DC_ASYNC=1) through the D&C parity tests + a bench line.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp SNN_DC_ASYNC=1
O=gpurun_out/${1:-r04c}; mkdir -p $O
( timeout 150 python __graft_entry__.py smoke 2>&1 | tail -15 ) > $O/smoke.log 2>&1
if ! grep -q "smoke ok" $O/smoke.log; then echo "smoke failed"; tail -5 $O/smoke.log; fi
( timeout 500 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q --no-header -x -k "dc2015 and auto" 2>&1 | tail -40 ) > $O/baseline_auto.log 2>&1
tail -3 $O/baseline_auto.log
( timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_fused_stress.py tests/test_gpu_resident_safety.py tests/test_gpu_network.py -m gpu -q --no-header -k "dc or lean or resident or stress or learning or additive or one_sided or missing or arbitration" 2>&1 | tail -60 ) > $O/dc_tests.log 2>&1
tail -3 $O/dc_tests.log
( timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>&1 | tail -2 ) > $O/bench.log 2>&1
cut -c1-400 $O/bench.log | tail -1
echo done
