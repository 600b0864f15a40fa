"""Device paths that round 3's second session prepared and checked on the host only (they were gated behind SNN_EXPERIMENTAL=1
until they had run on an MI355X).  Round 4's first GPU call ran all eight green (profiles/r04_experimental_suite_mi355x.log);
the gate is gone and the file name is historical.

conv PostPre from packed spike rows (csrc/snn_conv_events.hpp, k_conv_pp_partial_ev) is now the library's default; the existing
parity tests of Conv2d PostPre (bit-exact against the oracle, reference fixtures) are re-run with the DENSE body selected
(SNN_CONV_PP_EVENTS=0) so that both bodies stay covered."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_dense_conv_postpre_body_still_passes_the_conv_postpre_parity_tests():
    env = dict(os.environ, SNN_CONV_PP_EVENTS="0")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_extras.py"), "-m", "gpu", "-q", "-k", "conv2d_postpre"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-3000:] + out.stderr[-1500:]


def test_selftest_device_part():
    """bindsnet_amd.selftest's device checks: the propagation, normalisation and PostPre kernels against the torch expressions the
    reference evaluates (1 thread), on random data, bit for bit."""
    from bindsnet_amd import selftest
    msgs = []
    assert selftest.device_checks(msgs.append), msgs


def test_exact_mode_through_the_native_rccl_communicator(tmp_path):
    """parallel.exact_run(comm=NativeComm): the per-timestep exchange through the C ABI's own collective snn_dist_allgather_step
    (world size 1 on a one-GPU box: RCCL refuses two ranks on one device) == the reference fixture."""
    import exact_harness as H
    res = H.launch(1, "run_dc_n400_b4", "cuda", tmp_path, timeout=240, native=True)
    H.check_against_reference(res, "run_dc_n400_b4")


def test_conv2d_normalize_on_the_device_matches_reference():
    """snn_normalize_conv2d (ABI 6; Conv2dConnection.normalize, topology.py:824-837): the op-level reference fixture bit for bit, and a
    conv_mnist.py style run (Conv2d + PostPre + norm, normalised after each of two inputs): rasters identical, weights <= 1e-5 * wmax."""
    import numpy as np
    import torch
    import synth
    from cases import gold
    from bindsnet_amd.learning import PostPre
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Conv2dConnection
    T_ = lambda a: torch.from_numpy(np.ascontiguousarray(a))       # noqa: E731
    g = gold("op_conv_normalize")
    for k, (Cout, Cin, K) in enumerate(g["cases"]):
        Cout, Cin, K = int(Cout), int(Cin), int(K)
        c = Conv2dConnection(Input(shape=(Cin, K + 3, K + 3)), LIFNodes(shape=(Cout, 4, 4)), kernel_size=K,
                             w=T_(synth.uniform_f32(3300 + k, (Cout, Cin, K, K), 0.05, 1.0)).clone(), norm=0.4 * K * K).to("cuda")
        c.normalize()
        np.testing.assert_array_equal(c.w.detach().cpu().numpy().view(np.uint32), g[f"w{k}"].view(np.uint32), err_msg=f"case {k}")
    B2, T2 = 2, 30
    net = Network(dt=1.0)
    net.add_layer(Input(shape=(1, 12, 12), traces=True), "X")
    net.add_layer(LIFNodes(shape=(4, 10, 10), traces=True), "Y")
    cc = Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=3, stride=1, w=T_(synth.uniform_f32(3390, (4, 1, 3, 3), 0.0, 3.0)).clone(),
                          update_rule=PostPre, nu=(1e-3, 1e-2), reduction=torch.sum, wmin=0.0, wmax=4.0, norm=9.0)
    net.add_connection(cc, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T2)
    net.add_monitor(mon, "s")
    net.to("cuda")
    for r in range(2):
        net.run({"X": T_(synth.dense_spikes(3391 + r, (T2, B2, 1, 12, 12), 0.2)).to("cuda")}, time=T2)
        np.testing.assert_array_equal(np.packbits(mon.get("s").cpu().numpy().astype(np.uint8)), g[f"run{r}_sY"], err_msg=f"run {r} raster")
        np.testing.assert_allclose(cc.w.detach().cpu().numpy(), g[f"run{r}_W"], rtol=0, atol=1e-5 * 4.0)
        net.reset_state_variables()


def test_conv_mnist_training_graph_on_the_device_matches_reference():
    """examples/mnist/conv_mnist.py's training graph (Conv2d 16x16 stride 4 + PostPre + norm -> D&C nodes (25, 4, 4) with lateral
    inhibition, batch 1) on the generic plan against the reference fixture: tests/test_host_path.py has the host twin."""
    from test_host_path import _conv_mnist_net, check_conv_mnist_graph
    net, mons, cc = _conv_mnist_net()
    net.to("cuda")
    check_conv_mnist_graph(net, mons, cc, dev="cuda")
    assert net.last_plan == "generic"


def test_conv_mnist_script_itself_on_the_device():
    """The LITERAL examples/mnist/conv_mnist.py (staged byte copy, sha256-checked) with `bindsnet` = this package on the MI355X (the
    script's default is --gpu): the reference's CPU run of the same file -- rasters, theta, weights <= 1e-5."""
    from test_example_scripts import run_and_check
    run_and_check("generic")


def test_reservoir_script_itself_on_the_device():
    """The LITERAL examples/mnist/reservoir.py with --gpu: its LIF layer has PER-NEURON thresholds (a numpy array handed to the
    constructor), which raised on the device until round 5 (snn_layer_desc.thresh_vec, ABI 8).  The reference's propagation goes through
    MKL here (dense Connections), so the comparison is the dense family's: the O rasters of the reference's CPU run, allowing a handful of
    threshold-edge flips."""
    from test_example_scripts import run_reservoir
    r, g = run_reservoir(["--gpu"], "generic")
    ref = [int(v) for v in g["raster_sum"]]
    assert len(r["raster_sum"]) == len(ref) and sum(ref) > 20
    exact = sum(a == b for a, b in zip(r["raster_sha"], [str(v) for v in g["raster_sha"]]))
    flips = sum(abs(a - b) for a, b in zip(r["raster_sum"], ref))
    assert exact >= len(ref) - 2 and flips <= 2, (exact, flips, r["raster_sum"], ref)


def test_index_tensor_clamps_on_the_device_match_reference():
    """supervised_mnist.py:201-207 clamps with an integer tensor of neuron INDICES (the reference's `s[:, clamp] = 1` takes masks and
    indices alike): D&C graph on the generic plan, index clamp / index unclamp / per-step index rows, against the reference fixture."""
    from test_host_path import clamp_index_runs
    clamp_index_runs("cuda")


def test_user_guide_example_on_the_device():
    """docs/source/guide/guide_part_i.rst's end-to-end example on the MI355X (generic plan: dense feed-forward + recurrent
    Connection, LIF, s / v monitors, a 2-D input): voltages within the dense family's tolerance of the reference, at most a
    handful of threshold-edge spike flips among 5 321 spikes."""
    from test_host_path import guide_example_check
    guide_example_check("cuda")
