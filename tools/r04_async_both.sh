#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
bash tools/r04_async_try.sh $1 2>&1 | grep -v amdgpu.ids
bash tools/r04_async_timing.sh $1t ${2:-10} 2>&1 | grep -v amdgpu.ids
