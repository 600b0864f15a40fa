#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04m; mkdir -p $O
for wg in ${@:-10 47 3}; do
  SNN_DC_TIMING=$wg SNN_DC_TIMING_DUMP=$O/marks_$wg.bin timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | grep "dc2015 async, us" | tail -1 | cut -c1-400
  python tools/r04_marks.py $O/marks_$wg.bin 250
done
