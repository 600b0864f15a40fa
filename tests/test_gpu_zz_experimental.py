"""Kernels that compile and whose arithmetic is checked on the host (tests/test_conv_events_host.py) but that have not run on an
MI355X yet: opt-in in the library (environment switch), and their device tests only run when SNN_EXPERIMENTAL=1.

    SNN_EXPERIMENTAL=1 python -m pytest tests/test_gpu_zz_experimental.py -m gpu -q

conv PostPre from packed spike rows (csrc/snn_conv_events.hpp, k_conv_pp_partial_ev, SNN_CONV_PP_EVENTS=1): the existing parity
tests of Conv2d PostPre (bit-exact against the oracle, reference fixtures) are re-run in a process that has the switch on."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(os.environ.get("SNN_EXPERIMENTAL") != "1", reason="experimental kernels: set SNN_EXPERIMENTAL=1")
def test_event_driven_conv_postpre_passes_the_conv_postpre_parity_tests():
    env = dict(os.environ, SNN_CONV_PP_EVENTS="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_extras.py"), "-m", "gpu", "-q", "-k", "conv2d_postpre"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-3000:] + out.stderr[-1500:]


@pytest.mark.skipif(os.environ.get("SNN_EXPERIMENTAL") != "1", reason="not yet run on an MI355X: set SNN_EXPERIMENTAL=1")
def test_selftest_device_part():
    """bindsnet_amd.selftest's device checks: the propagation, normalisation and PostPre kernels against the torch expressions the
    reference evaluates (1 thread), on random data, bit for bit."""
    from bindsnet_amd import selftest
    msgs = []
    assert selftest.device_checks(msgs.append), msgs


@pytest.mark.skipif(os.environ.get("SNN_EXPERIMENTAL") != "1", reason="not yet run on an MI355X: set SNN_EXPERIMENTAL=1")
def test_exact_mode_through_the_native_rccl_communicator(tmp_path):
    """parallel.exact_run(comm=NativeComm): the per-timestep exchange through the C ABI's own collective snn_dist_allgather_step
    (world size 1 on a one-GPU box: RCCL refuses two ranks on one device) == the reference fixture."""
    import exact_harness as H
    res = H.launch(1, "run_dc_n400_b4", "cuda", tmp_path, timeout=240, native=True)
    H.check_against_reference(res, "run_dc_n400_b4")
