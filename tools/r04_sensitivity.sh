#!/bin/bash
# times every bindsnet_amd/lib/whatif/libsnnhip_w*.so against the product build on one box (tools/time_run.py), two rounds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for rep in 1 2; do
  timeout 100 python tools/time_run.py 30 2>&1 | tail -1
  for f in $(ls bindsnet_amd/lib/whatif/libsnnhip_w*.so | sort -V); do
    SNN_LIB_OVERRIDE=$PWD/$f timeout 100 python tools/time_run.py 30 2>&1 | tail -1
  done
done
