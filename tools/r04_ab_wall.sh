#!/bin/bash
# A/B of two builds on one box by bench.py's WALL clock per run (pre-passes and launch gaps included)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_async_form.py tests/test_gpu_fuzz.py tests/test_gpu_fused_stress.py tests/test_gpu_resident_safety.py -m gpu -q --no-header -x -k "dc or lean or resident or stress or learning or additive or one_sided or missing or arbitration or generation or short or excit or diagonal" 2>&1 | tail -12 ) | grep -v amdgpu | tail -3
for rep in 1 2 3; do for lib in bindsnet_amd/lib/libsnnhip_base.so bindsnet_amd/lib/libsnnhip.so; do
  SNN_LIB_OVERRIDE=$PWD/$lib timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', d['value'], 'ms/step', d['ms_per_step'], 'kernel us', d['roofline']['avg_launch_us'])"
done; done
