#!/bin/bash
# Round 4, first GPU call: everything round 3's second session prepared without a GPU.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
( SNN_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_gpu_zz_experimental.py -m gpu -q --no-header -rfEs 2>&1 | tail -150 ) > $O/experimental_all.log 2>&1
( timeout 200 python tools/bench_configs.py --only f_conv_postpre 2>&1 | tail -5 ) > $O/conv_pp_dense.log 2>&1
( SNN_CONV_PP_EVENTS=1 timeout 200 python tools/bench_configs.py --only f_conv_postpre 2>&1 | tail -5 ) > $O/conv_pp_events.log 2>&1
( timeout 300 python tools/bench_exact.py --device cuda --worlds 2 1 2>&1 | tail -5 ) > $O/exact_mode.jsonl 2>&1
( timeout 400 python tools/eth_mnist_timing.py --impl amd --out gpurun_out/r04a/eth_mnist_literal_mi355x.json 2>&1 | tail -30 ) > $O/eth_timing.log 2>&1
( timeout 400 python bench.py 2>&1 | tail -3 ) > $O/bench.log 2>&1
echo done
