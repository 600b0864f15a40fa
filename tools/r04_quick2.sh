#!/bin/bash
# D&C parity tests on the current build, then A/B against bindsnet_amd/lib/libsnnhip_base.so on the same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_async_form.py tests/test_gpu_fuzz.py tests/test_gpu_fused_stress.py tests/test_gpu_resident_safety.py -m gpu -q --no-header -x -k "dc or lean or resident or stress or learning or additive or one_sided or missing or arbitration or generation or short or excit or diagonal" 2>&1 | tail -12 ) | grep -v amdgpu | tail -3
bash tools/r04_ab_lib.sh
