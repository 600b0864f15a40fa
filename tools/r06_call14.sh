#!/bin/bash
O=gpurun_out/r06_c14; mkdir -p $O
export TMPDIR=/tmp
python tools/r06_encode_timing.py > $O/encode_timing.json 2> $O/encode_timing.err; cat $O/encode_timing.json
(SNN_EXACT_TIMING=1 timeout 600 python -m pytest tests/test_gpu_zz_exact_mode.py tests/test_gpu_encoding.py -m gpu -q --no-header -s 2>&1 | grep "^rank\|passed\|failed\|timesteps" | head -20) | tee $O/exact_and_encoding_tests.log
timeout 600 python tools/eth_mnist_timing.py --impl amd --n_train 40 --n_test 10 --encode-device cuda --out $O/eth_mnist_literal_device_encode.json > /dev/null 2> $O/eth_dev.err
timeout 600 python tools/eth_mnist_timing.py --impl amd --n_train 40 --n_test 10 --out $O/eth_mnist_literal_host_encode.json > /dev/null 2> $O/eth_host.err
python - <<'P'
import json
for f in ("host", "device"):
    try:
        d = json.load(open(f"gpurun_out/r06_c14/eth_mnist_literal_{f}_encode.json"))
        print(f, "encode ms/sample", d["ms_per_sample"]["encode"], "median", d["encode_ms_median"], "run", d["ms_per_sample"]["run"], "run median", d["run_ms_median"], "h2d", d["ms_per_sample"]["h2d"], "acc", d["accuracy"])
    except Exception as e:
        print(f, "FAILED", e)
P
bash tools/r06_ab5.sh c14ab
