#!/bin/bash
# Round profile artefacts ON THE GPU BOX (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02
# For the headline bench (cfg2) and for every other BASELINE config (tools/bench_configs.py --only <cfg>):
#   rocprofv3 --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (never combined with
#   tracing: MI355X_MICROARCH.md), summarised by tools/profile_summary.py into gpurun_out/prof_<tag>/summary.json
#   (+ kernel_stats_<cfg>.csv).  Copy what should be judged into profiles/.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {   # name, command...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name/ks" -o ks -- "$@" > "$OUT/$name.ks.log" 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$name/fetch" -o p -- "$@" > "$OUT/$name.fetch.log" 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$name/write" -o p -- "$@" > "$OUT/$name.write.log" 2>&1
}
run cfg2 python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline
for cfg in cfg1 cfg3_shard cfg3_b32 cfg3 cfg4 cfg5 f_hebbian f_wdpp; do     # (f_*: SURVEY 8(f) rules in the one-launch plan, 784->1600, B=32)
  run $cfg python "$R/tools/bench_configs.py" --runs 5 --only $cfg --no-cpu-baseline
done
python "$R/tools/profile_summary.py" "$OUT"
