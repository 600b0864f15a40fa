#!/bin/bash
O=gpurun_out/r06_c18; mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_extras.py tests/test_gpu_convlif.py tests/test_gpu_zz_experimental.py tests/test_gpu_network.py -m gpu -q --no-header 2>&1 | tail -5) | tee $O/conv_tests.log
timeout 300 python tools/bench_configs.py --runs 5 --only f_conv_postpre | tee $O/bench_conv_postpre.jsonl | cut -c1-300
bash tools/r06_conv_prof.sh
