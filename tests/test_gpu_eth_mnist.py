"""examples/mnist/eth_mnist.py end to end on the MI355X through the `bindsnet` import names.

 * test_eth_mnist_flow_matches_reference_cpu_run: tests/eth_mnist_flow.py (the script's workflow, call for call) on the
   GPU vs the fixture the UNMODIFIED reference script produced on the reference's CPU path with the same seeds and the
   same synthetic MNIST: the excitatory raster of every one of the 13 inputs (9 training, 4 test), final weights, theta,
   label assignments and both accuracies must be identical.
 * test_reference_script_itself_unmodified: the reference's own script file, byte for byte (its sha256 is checked),
   run as is -- from the reference checkout where that exists, else from the staged copy build() ships to the GPU box."""
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
STAGED_SCRIPT = os.path.join(HERE, "_staged", "eth_mnist.py")     # byte copy made by __graft_entry__.build(), git-ignored
REF_SCRIPT_SHA256 = "ba4f875d097536bf1d935070112f1ad7fdffe50c190d2e9a9de1e0b278fefe4a"   # of the reference's file


def literal_script():
    """The reference's own file where the checkout exists, else the staged byte copy -- never an edited one."""
    import hashlib
    for path in (REF_SCRIPT, STAGED_SCRIPT):
        if os.path.exists(path):
            with open(path, "rb") as f:
                assert hashlib.sha256(f.read()).hexdigest() == REF_SCRIPT_SHA256, f"{path} is not the reference's eth_mnist.py"
            return path
    return None


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


@pytest.mark.skipif(literal_script() is None, reason="neither the reference checkout nor the staged copy of its script "
                                                     "(python __graft_entry__.py build, in the build container) is here")
def test_reference_script_itself_unmodified():
    """examples/mnist/eth_mnist.py, the literal file (sha256-checked), executed with `bindsnet` = this package on the
    MI355X: every input's excitatory raster, the learned weights, theta, assignments and accuracies equal what the same
    file produced on the reference's CPU path (tests/golden/eth_mnist_flow.npz)."""
    check(_run(literal_script()))
