#!/usr/bin/env python3
"""Fixture for the second end-to-end example: the UNMODIFIED reference script examples/mnist/conv_mnist.py on the reference's CPU path
(build container only) over the synthetic MNIST stand-in -- the Y raster of every input (sha256 + spike count), the convolution
weights and theta after training.

    python tests/golden/make_golden_conv_mnist.py"""
import hashlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REF = "/root/reference/bindsnet"
for name, path in (("bindsnet", REF), ("bindsnet.analysis", REF + "/analysis")):
    pkg = types.ModuleType(name)
    pkg.__path__ = [path]
    sys.modules[name] = pkg
sys.modules["cv2"] = types.ModuleType("cv2")
import tv_shim  # noqa: E402
tv_shim.install()
import bindsnet.network  # noqa: E402
import bindsnet.network.network as refnet  # noqa: E402
import conv_mnist_harness as H  # noqa: E402

SCRIPT = "/root/reference/examples/mnist/conv_mnist.py"
ARGV = ["--n_train", "4", "--time", "100"]

if __name__ == "__main__":
    r = H.run_script(SCRIPT, refnet, ARGV, seed=0)
    np.savez_compressed(os.path.join(HERE, "conv_mnist_literal.npz"), argv=np.array(ARGV), script_sha=hashlib.sha256(open(SCRIPT, "rb").read()).hexdigest(),
                        raster_sha=np.array(r["raster_sha"]), raster_sum=np.array(r["raster_sum"]), W=r["W"], theta=r["theta"])
    print("inputs run:", len(r["raster_sha"]), "Y spikes per input:", r["raster_sum"], "filter sums:", r["W"].reshape(25, -1).sum(1)[:3])
