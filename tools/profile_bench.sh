#!/bin/bash
# Produces the profiles/ artefacts of a round ON THE GPU BOX (run through gpurun from the repo root):
#   kernel-trace stats of the default bench command, and HBM traffic of the dominant kernel from two
#   separate PMC passes (FETCH_SIZE, WRITE_SIZE -- never combined with tracing, MI355X_MICROARCH.md).
# usage: bash tools/profile_bench.sh [tag]      -> gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_hbm_traffic.json,*.log}
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks" -o ks -- python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/ks.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python "$R/bench.py" --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python "$R/bench.py" --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/write.log" 2>&1
python "$R/tools/pmc_summary.py" "$OUT"
