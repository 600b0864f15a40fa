"""The reference's second MNIST example, examples/mnist/conv_mnist.py, ITSELF (sha256-checked byte copy or the checkout's file),
with `bindsnet` = this package: on the host here (network/host_path.py), on the MI355X in tests/test_gpu_zz_experimental.py.  It
must reproduce what the same file produced on the reference's CPU path (tests/golden/make_golden_conv_mnist.py): the Y raster of
every training input, theta, and the convolution weights (Conv2d + PostPre + per-filter normalisation) within the convolution's
tolerance."""
import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SCRIPT = "/root/reference/examples/mnist/conv_mnist.py"
STAGED = os.path.join(ROOT, "tests", "_staged", "conv_mnist.py")


def literal_script(g):
    path = REF_SCRIPT if os.path.exists(REF_SCRIPT) else STAGED
    if not os.path.exists(path):
        pytest.skip("no copy of examples/mnist/conv_mnist.py on this machine (python __graft_entry__.py build stages one where the reference checkout exists)")
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == str(g["script_sha"]), "not the reference's conv_mnist.py"
    return path


def run_and_check(expect_plan):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bindsnet  # noqa: F401  (alias package -> bindsnet_amd)
    import bindsnet_amd.network.network as netmod
    import conv_mnist_harness as H
    from cases import gold
    g = gold("conv_mnist_literal")
    r = H.run_script(literal_script(g), netmod, [str(a) for a in g["argv"]], seed=0)
    assert r["plan"] == expect_plan
    assert r["raster_sum"] == [int(v) for v in g["raster_sum"]] and r["raster_sha"] == [str(v) for v in g["raster_sha"]], "Y rasters"
    np.testing.assert_array_equal(r["theta"].view(np.uint32), g["theta"].view(np.uint32))
    np.testing.assert_allclose(r["W"], g["W"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(r["W"].reshape(r["W"].shape[0], -1).sum(1), 0.4 * 16 ** 2, rtol=1e-6)


def test_conv_mnist_script_itself_on_the_host():
    import torch
    if torch.cuda.is_available():
        pytest.skip("on a GPU box the script moves the network to the device: tests/test_gpu_zz_experimental.py")
    run_and_check("host-torch")


# ---------------------------------------------------------------------------------------------------- examples/mnist/reservoir.py
REF_RESERVOIR = "/root/reference/examples/mnist/reservoir.py"
STAGED_RESERVOIR = os.path.join(ROOT, "tests", "_staged", "reservoir.py")


def run_reservoir(extra_argv, plan):
    """examples/mnist/reservoir.py itself (see the host test below) with `extra_argv` added; the O raster of all twelve inputs against the
    reference's CPU run of the same file.  With --gpu the network (per-neuron LIF thresholds, dense Connections, no learning) runs on
    the MI355X's generic plan; the script seeds only the CUDA generator then, the harness seeds the CPU one, so weights, thresholds and
    the encoded inputs are the fixture's."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bindsnet  # noqa: F401
    import bindsnet_amd.network.network as netmod
    import conv_mnist_harness as H
    from cases import gold
    g = gold("reservoir_literal")
    path = REF_RESERVOIR if os.path.exists(REF_RESERVOIR) else STAGED_RESERVOIR
    if not os.path.exists(path):
        pytest.skip("no copy of examples/mnist/reservoir.py on this machine")
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == str(g["script_sha"])
    np.random.seed(0)
    r = H.run_script(path, netmod, [str(a) for a in g["argv"]] + list(extra_argv), seed=0, monitor="O_spikes",
                     result=lambda gl: dict(thresh=gl["network"].layers["O"].thresh.detach().cpu().numpy().copy(),
                                            accuracy=float(100 * gl["correct"] / gl["total"])))
    assert r["plan"] == plan
    np.testing.assert_array_equal(r["thresh"], g["thresh"])
    return r, g


def test_reservoir_script_itself_on_the_host():
    """examples/mnist/reservoir.py, unmodified (it runs on the CPU by default: parser.set_defaults(gpu=False)): Input -> random dense
    Connection -> LIFNodes with PER-NEURON thresholds (a numpy array handed to the constructor) and a random recurrent Connection, spike
    / voltage monitors with a `device=` argument, no learning, then a torch read-out trained on the recorded spikes.  The O raster of
    all twelve inputs (train + test pass) equals what the same file produced on the reference's CPU path."""
    r, g = run_reservoir([], "host-torch")
    assert r["raster_sum"] == [int(v) for v in g["raster_sum"]] and r["raster_sha"] == [str(v) for v in g["raster_sha"]], "O rasters"
    assert r["accuracy"] == float(g["accuracy"])
