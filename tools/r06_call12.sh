#!/bin/bash
O=gpurun_out/r06_c12; mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_encoding.py tests/test_gpu_zz_exact_mode.py tests/test_gpu_async_form.py tests/test_gpu_fullsize.py -m gpu -q --no-header 2>&1 | tail -15) > $O/tests.log; tail -4 $O/tests.log
timeout 600 python tools/eth_mnist_timing.py --impl amd --n_train 40 --n_test 10 --out $O/eth_mnist_literal_host_encode.json > /dev/null 2> $O/eth_host.err
timeout 600 python tools/eth_mnist_timing.py --impl amd --n_train 40 --n_test 10 --encode-device cuda --out $O/eth_mnist_literal_device_encode.json > /dev/null 2> $O/eth_dev.err
python - <<'P'
import json
for f in ("host", "device"):
    try:
        d = json.load(open(f"gpurun_out/r06_c12/eth_mnist_literal_{f}_encode.json"))
        print(f, "encode ms/sample", d["ms_per_sample"]["encode"], "median", d["encode_ms_median"], "run", d["ms_per_sample"]["run"], "run median", d["run_ms_median"], "h2d", d["ms_per_sample"]["h2d"], "acc", d["accuracy"])
    except Exception as e:
        print(f, "FAILED", e)
P
bash tools/r06_two512.sh
