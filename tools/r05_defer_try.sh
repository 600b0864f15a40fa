#!/bin/bash
# the SNN_DEFER build (bindsnet_amd/lib/whatif/libsnnhip_w500.so: tools/r04_sensitivity_build.sh 500 with WHATIF_EXTRA=-DSNN_DEFER=1) on the GPU box:
# the D&C parity tests against it, then its time per launch beside the product build's
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=$PWD/bindsnet_amd/lib/whatif/libsnnhip_w500.so
[ -f "$L" ] || { echo "no $L"; exit 1; }
( SNN_LIB_OVERRIDE=$L timeout ${1:-60} python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_async_form.py tests/test_gpu_fused_stress.py -m gpu -q --no-header -x -k "dc or lean or resident or stress or learning or additive or one_sided or missing or short or excit or diagonal" 2>&1 | tail -12 ) | grep -v amdgpu | tail -6
for rep in 1; do
  timeout 40 python tools/time_run.py 30 2>&1 | tail -1
  SNN_LIB_OVERRIDE=$L timeout 40 python tools/time_run.py 30 2>&1 | tail -1
done
