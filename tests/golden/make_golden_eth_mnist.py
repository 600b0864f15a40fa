#!/usr/bin/env python3
"""Fixture for the end-to-end example: run the UNMODIFIED reference script examples/mnist/eth_mnist.py on the
reference's CPU path (build container only) over the synthetic MNIST stand-in and store what it produced -- the
excitatory raster of every input (sha256 + spike count), final weights / theta, label assignments, accuracies.

    python tests/golden/make_golden_eth_mnist.py

The GPU tests run the same flow through `bindsnet` = bindsnet_amd on an MI355X and must reproduce all of it."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REF = "/root/reference/bindsnet"
for name, path in (("bindsnet", REF), ("bindsnet.analysis", REF + "/analysis")):   # skip the two __init__ files that
    pkg = types.ModuleType(name)                                                    # need tensorboardX / gym / ...
    pkg.__path__ = [path]
    sys.modules[name] = pkg
sys.modules["cv2"] = types.ModuleType("cv2")          # bindsnet.datasets imports the video corpora, which import cv2
import tv_shim  # noqa: E402
tv_shim.install()
import bindsnet.network  # noqa: E402  (first: import cycle)
import bindsnet.network.network as refnet  # noqa: E402
import eth_mnist_harness as H  # noqa: E402

ARGV = ["--n_train", "8", "--n_test", "4", "--update_interval", "4", "--n_neurons", "100", "--time", "250"]

if __name__ == "__main__":
    r = H.run_script("/root/reference/examples/mnist/eth_mnist.py", refnet, ARGV, seed=0)
    out = dict(argv=np.array(ARGV), raster_sha=np.array(r["raster_sha"]), raster_sum=np.array(r["raster_sum"]),
               W_sha=H.sha(r["W"]), W_sample=r["W"].reshape(-1)[::97].copy(), theta=r["theta"], assignments=r["assignments"],
               proportions=r["proportions"], acc_all=np.float64(r["accuracy"]["all"]),
               acc_proportion=np.float64(r["accuracy"]["proportion"]))
    np.savez_compressed(os.path.join(HERE, "eth_mnist_flow.npz"), **out)
    print("inputs run:", len(r["raster_sha"]), "exc spikes per input:", r["raster_sum"], "accuracy:", r["accuracy"])
