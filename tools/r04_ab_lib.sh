#!/bin/bash
# A/B of two builds on one box: bindsnet_amd/lib/libsnnhip_base.so (copied before the change) vs libsnnhip.so
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  SNN_LIB_OVERRIDE=$PWD/bindsnet_amd/lib/libsnnhip_base.so timeout 100 python tools/time_run.py 40 2>&1 | tail -1
  timeout 100 python tools/time_run.py 40 2>&1 | tail -1
done
