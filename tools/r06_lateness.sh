#!/bin/bash
# GPU box: the "lite" timing dump (publish times of every compute workgroup + the arbiter's marks) through tools/r04_lateness.py
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r06_late; mkdir -p $O
SNN_DC_TIMING=-1 SNN_DC_TIMING_DUMP=$O/lite.bin timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | grep "dc2015 async" | tail -1 | cut -c1-300
python tools/r04_lateness.py $O/lite.bin 250 100
