"""Network.pipelined() where there is no device (CPU tier): a network on the host runs its synchronous plain-PyTorch loop inside a section too
(the section defers nothing there), sections do not nest, and the settle hook is removed again -- also when the body raises."""
import numpy as np
import pytest
import torch

from bindsnet_amd import rng, synth
from bindsnet_amd.models import DiehlAndCook2015
from bindsnet_amd.network.monitors import Monitor


def build(N=36, T=20):
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120.0, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    net.connections[("X", "Ae")].pipeline[0].value.data.copy_(torch.from_numpy(np.minimum(synth.uniform_f32(3, (784, N), 0.0, 1.0), 1.0)))
    mon = Monitor(net.layers["Ae"], ["s"], time=T)
    net.add_monitor(mon, "Ae")
    return net, mon


def run(net, mon, spikes, T, B, section):
    import contextlib
    torch.manual_seed(11)
    out = []
    with (net.pipelined() if section else contextlib.nullcontext()):
        for s in spikes:
            net.run({"X": torch.from_numpy(s.copy()).view(T, B, 1, 28, 28)}, time=T)     # (a copy: reset zeroes the last slice through the Input.s alias, as in the reference)
            out.append(mon.get("s").clone().numpy())
            net.reset_state_variables()
    return out, net.connections[("X", "Ae")].pipeline[0].value.detach().numpy().copy(), torch.rand(3).numpy()


def test_host_network_inside_a_section_is_the_synchronous_loop():
    T, B = 20, 3
    spikes = [synth.dense_spikes(40 + r, (T, B, 784), 0.1) for r in range(2)]
    a = run(*build(T=T), spikes, T, B, True)
    b = run(*build(T=T), spikes, T, B, False)
    assert sum(int(x.sum()) for x in b[0]) > 0
    for x, y in zip(a[0], b[0]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    np.testing.assert_array_equal(a[2], b[2])


def test_sections_do_not_nest_and_clean_up():
    net, _ = build()
    n0 = len(rng._PENDING)
    with net.pipelined():
        assert len(rng._PENDING) == n0 + 1
        with pytest.raises(RuntimeError):
            with net.pipelined():
                pass
        net.sync()                                         # nothing enqueued: a no-op
    assert len(rng._PENDING) == n0 and net.__dict__.get("_pipe") is None
    with pytest.raises(ZeroDivisionError):
        with net.pipelined():
            1 / 0
    assert len(rng._PENDING) == n0 and net.__dict__.get("_pipe") is None
    net.sync()


def test_vector_thresholds_on_the_host_path_match_the_reference_fixture():
    """(CPU tier twin of tests/test_gpu_vector_thresh.py::test_vector_threshold_run_matches_the_reference_fixture: the same fixture, the
    package's plain-PyTorch host path.)"""
    import test_gpu_vector_thresh as tv
    n = torch.get_num_threads()
    try:
        torch.set_num_threads(min(4, n))
        tv.check_against_reference_fixture("cpu", "host-torch")
    finally:
        torch.set_num_threads(n)
