#!/bin/bash
# A developer build of the library (tools/r04_sensitivity_build.sh <k> with WHATIF_EXTRA=-D<macro>=1 -> bindsnet_amd/lib/whatif/libsnnhip_w<k>.so) on
# the GPU box: the D&C parity tests and a short soak against it, then its time per launch beside the product build's.
#   built here (no GPU needed), e.g.:   WHATIF_EXTRA=-DSNN_LDS_XTRACE=1 bash tools/r04_sensitivity_build.sh 600
#                                       WHATIF_EXTRA=-DSNN_DEFER=1      bash tools/r04_sensitivity_build.sh 500
#                                       WHATIF_EXTRA=-DSNN_DIGEST_EARLY=1 bash tools/r04_sensitivity_build.sh 700
#                                       WHATIF_EXTRA="-DSNN_DIGEST_EARLY=1 -DSNN_LDS_XTRACE=1" bash tools/r04_sensitivity_build.sh 800
#                                       WHATIF_EXTRA=-DSNN_POLL2=1      bash tools/r04_sensitivity_build.sh 900
#   on the box:                         gpurun -- 'bash tools/r05_variant_try.sh 600 [test-timeout-s] [soak-cases]'
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=$PWD/bindsnet_amd/lib/whatif/libsnnhip_w${1:-600}.so
[ -f "$L" ] || { echo "no $L"; exit 1; }
( SNN_LIB_OVERRIDE=$L timeout ${2:-60} python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_async_form.py tests/test_gpu_fused_stress.py tests/test_gpu_fuzz.py -m gpu -q --no-header -x -k "dc or lean or resident or stress or learning or additive or one_sided or missing or short or excit or diagonal" 2>&1 | tail -12 ) | grep -v amdgpu | tail -6
if [ "${3:-0}" -gt 0 ]; then ( SNN_LIB_OVERRIDE=$L timeout 200 python tools/r04_soak.py $3 3000 2>&1 | grep -v amdgpu | tail -2 ); fi
for rep in 1 2; do
  timeout 40 python tools/time_run.py 30 2>&1 | tail -1
  SNN_LIB_OVERRIDE=$L timeout 40 python tools/time_run.py 30 2>&1 | tail -1
done
