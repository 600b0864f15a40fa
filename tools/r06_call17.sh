#!/bin/bash
O=gpurun_out/r06_c17; mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header 2>&1 | tail -3) | tee $O/ops_tests.log
bash tools/r06_density_map.sh 2>&1 | tail -10
bash tools/r06_conv_prof.sh
