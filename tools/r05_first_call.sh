#!/bin/bash
# first GPU call of round 5 (after tools/r05_build_variants.sh here): every variant in bindsnet_amd/lib/whatif/ against the D&C parity tests
# (a variant that fails them is not timed), then all surviving variants and the product build timed on this one box, two rounds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_first
GOOD=""
for f in $(ls bindsnet_amd/lib/whatif/libsnnhip_w*.so | sort -V); do
  ( SNN_LIB_OVERRIDE=$PWD/$f timeout 90 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_async_form.py tests/test_gpu_fused_stress.py -m gpu -q --no-header -x \
      -k "dc or lean or resident or stress or learning or additive or one_sided or missing or short or excit or diagonal" 2>&1 | tail -5 ) > gpurun_out/r05_first/tests_$(basename $f .so).log 2>&1
  if grep -q " passed" gpurun_out/r05_first/tests_$(basename $f .so).log && ! grep -q "failed\|error" gpurun_out/r05_first/tests_$(basename $f .so).log; then GOOD="$GOOD $f"; echo "$f: parity ok"; else echo "$f: PARITY FAILED"; tail -3 gpurun_out/r05_first/tests_$(basename $f .so).log; fi
done
for rep in 1 2; do
  timeout 60 python tools/time_run.py 30 2>&1 | tail -1
  for f in $GOOD; do SNN_LIB_OVERRIDE=$PWD/$f timeout 60 python tools/time_run.py 30 2>&1 | tail -1; done
done
