#!/bin/bash
O=gpurun_out/r06_c13; mkdir -p $O
export TMPDIR=/tmp
python tools/r06_encode_timing.py > $O/encode_timing.json 2> $O/encode_timing.err; cat $O/encode_timing.json
SNN_EXACT_TIMING=1 timeout 600 python -m pytest tests/test_gpu_zz_exact_mode.py -m gpu -q --no-header -k gathered -s 2>&1 | grep "rank\|passed\|failed\|timesteps" | head -20
