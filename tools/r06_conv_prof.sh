#!/bin/bash
# kernel-trace stats of the Conv2d PostPre graph (generic plan) -> which kernel its 113 us per timestep go to
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06_conv_prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks" -o ks -- python "$R/tools/bench_configs.py" --runs 3 --only f_conv_postpre > "$OUT/ks.log" 2>&1
f=$(find $OUT/ks -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-220; cp $f $OUT/kernel_stats.csv
