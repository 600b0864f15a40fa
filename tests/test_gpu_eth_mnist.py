"""examples/mnist/eth_mnist.py end to end on the MI355X through the `bindsnet` import names.

 * test_eth_mnist_flow_matches_reference_cpu_run: tests/eth_mnist_flow.py (the script's workflow, call for call) on the
   GPU vs the fixture the UNMODIFIED reference script produced on the reference's CPU path with the same seeds and the
   same synthetic MNIST: the excitatory raster of every one of the 13 inputs (9 training, 4 test), final weights, theta,
   label assignments and both accuracies must be identical.
 * test_reference_script_itself_unmodified: where the reference checkout exists, its own script file is run as is."""
import os
import sys

import numpy as np
import pytest

import cases
import eth_mnist_harness as H

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_SCRIPT = "/root/reference/examples/mnist/eth_mnist.py"


def check(r):
    g = cases.gold("eth_mnist_flow")
    assert list(r["raster_sum"]) == list(g["raster_sum"]), "excitatory spike counts per input"
    assert list(r["raster_sha"]) == [str(x) for x in g["raster_sha"]], "excitatory rasters"
    assert H.sha(r["W"]) == str(g["W_sha"]), "learned weights"
    np.testing.assert_array_equal(r["theta"].view(np.uint32), g["theta"].view(np.uint32))
    np.testing.assert_array_equal(r["assignments"], g["assignments"])
    np.testing.assert_array_equal(r["proportions"].view(np.uint32), g["proportions"].view(np.uint32))
    assert r["accuracy"]["all"] == float(g["acc_all"]) and r["accuracy"]["proportion"] == float(g["acc_proportion"])


def _run(path):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bindsnet  # noqa: F401  (the alias package at the repository root)
    import bindsnet_amd.network.network as netmod
    g = cases.gold("eth_mnist_flow")
    return H.run_script(path, netmod, [str(x) for x in g["argv"]], seed=0)


def test_eth_mnist_flow_matches_reference_cpu_run():
    check(_run(os.path.join(HERE, "eth_mnist_flow.py")))


@pytest.mark.skipif(not os.path.exists(REF_SCRIPT), reason="the reference checkout is not on this machine")
def test_reference_script_itself_unmodified():
    check(_run(REF_SCRIPT))
