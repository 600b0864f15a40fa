#!/bin/bash
# round 5, first GPU call: (1) the developer variants round 4 left unmeasured (parity, then timing beside the product build),
# (2) the bench line at the start of the round, (3) host timeline and device timeline of the bench loop
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_call1
bash tools/r05_first_call.sh > gpurun_out/r05_call1/variants.log 2>&1
tail -20 gpurun_out/r05_call1/variants.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_call1/bench_start.json 2> gpurun_out/r05_call1/bench_start.err
tail -c 600 gpurun_out/r05_call1/bench_start.json
timeout 120 python tools/host_timeline.py 2>&1 | tail -3 | tee gpurun_out/r05_call1/host_timeline.txt
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OLDPWD/gpurun_out/r05_call1/trace -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OLDPWD/gpurun_out/r05_call1/trace_bench.json 2> $OLDPWD/gpurun_out/r05_call1/trace.err )
python tools/gpu_gaps.py gpurun_out/r05_call1/trace 2>&1 | tee gpurun_out/r05_call1/gpu_gaps.txt | head -40
