#!/usr/bin/env python3
"""Fixture for a third end-to-end example: the UNMODIFIED reference script examples/mnist/reservoir.py (Input -> random dense Connection
-> LIFNodes with PER-NEURON thresholds and a random recurrent Connection, no learning; a torch read-out trained on the spike counts) on
the reference's CPU path (build container only) over the synthetic MNIST stand-in.

    python tests/golden/make_golden_reservoir.py"""
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

SCRIPT = "/root/reference/examples/mnist/reservoir.py"
ARGV = ["--n_neurons", "100", "--n_epochs", "2", "--examples", "5", "--time", "60", "--n_workers", "0"]


def result(g):
    return dict(thresh=g["network"].layers["O"].thresh.detach().cpu().numpy().copy(), accuracy=float(100 * g["correct"] / g["total"]))


if __name__ == "__main__":
    np.random.seed(0)                 # (the script draws its per-neuron thresholds from numpy's global generator and never seeds it)
    r = H.run_script(SCRIPT, refnet, ARGV, seed=0, monitor="O_spikes", result=result)
    np.savez_compressed(os.path.join(HERE, "reservoir_literal.npz"), argv=np.array(ARGV), script_sha=hashlib.sha256(open(SCRIPT, "rb").read()).hexdigest(),
                        raster_sha=np.array(r["raster_sha"]), raster_sum=np.array(r["raster_sum"]), thresh=r["thresh"], accuracy=np.float64(r["accuracy"]))
    print("inputs run:", len(r["raster_sha"]), "O spikes per input:", r["raster_sum"], "accuracy", r["accuracy"])
