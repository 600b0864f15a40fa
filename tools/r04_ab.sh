#!/bin/bash
# A/B of a developer switch on the same box: SNN_DC_SPECFLAGS=<a> vs <b>, alternating, kernel time per launch from bench.py's HIP events
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for rep in 1 2 3; do for f in ${1:-0} ${2:-1}; do
  SNN_DC_SPECFLAGS=$f timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('flags $f:', d['value'], 'kernel us', d['roofline']['avg_launch_us'], 'parity skipped')"
done; done
