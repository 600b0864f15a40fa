#!/bin/bash
# Round-4 profile artefacts (run through gpurun): kernel stats + HBM traffic of the default bench command, SQ issue counters
cd "$GRAFT_REPO_ROOT"
bash tools/profile_bench.sh r04 > gpurun_out/prof_r04_bench.log 2>&1
tail -2 gpurun_out/prof_r04_bench.log | cut -c1-600
bash tools/pmc_issue_stats.sh > gpurun_out/prof_r04_issue.txt 2>&1
cat gpurun_out/prof_r04_issue.txt | tail -20
head -6 gpurun_out/prof_r04/kernel_stats.csv | cut -c1-160
