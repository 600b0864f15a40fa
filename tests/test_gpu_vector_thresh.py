"""Per-neuron (tensor-valued) LIF thresholds on the device (bindsnet/network/nodes.py:425-498 accept `thresh` as a tensor;
examples/mnist/reservoir.py builds its output layer with one): snn_layer_desc.thresh_vec / snn_lif_step_vth (ABI 8).

Checker: the same network stepped by network/host_path.py on CPU tensors -- the plain-PyTorch restatement of the reference's step loop,
itself pinned to the reference by tests/test_host_path.py and the literal reservoir.py fixture (tests/test_example_scripts.py)."""
import numpy as np
import pytest
import torch

from bindsnet_amd import synth
from bindsnet_amd.network import Network
from bindsnet_amd.network.monitors import Monitor
from bindsnet_amd.network.nodes import Input, LIFNodes
from bindsnet_amd.network.topology import Connection

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def few_host_threads():
    """The checker is the plain-PyTorch host path: on the 256-core GPU box torch's default thread count makes every small operator of it
    take milliseconds (60 s for one test); the host path's reductions run at one thread anyway."""
    n = torch.get_num_threads()
    torch.set_num_threads(min(4, n))
    yield
    torch.set_num_threads(n)


def build(n_in, n_out, thresh, recurrent=True):
    torch.manual_seed(3)
    net = Network(dt=1.0)
    net.add_layer(Input(n=n_in), "I")
    net.add_layer(LIFNodes(n=n_out, thresh=thresh, refrac=2, tc_decay=50.0, traces=True), "O")
    net.add_connection(Connection(net.layers["I"], net.layers["O"], w=torch.from_numpy(synth.uniform_f32(5, (n_in, n_out), 0.0, 2.5))), "I", "O")
    if recurrent:
        net.add_connection(Connection(net.layers["O"], net.layers["O"], w=torch.from_numpy(synth.uniform_f32(6, (n_out, n_out), -0.5, 0.5))), "O", "O")
    return net


@pytest.mark.parametrize("B,as_numpy", [(1, True), (5, False)])
def test_vector_threshold_run_matches_the_host_path(B, as_numpy):
    n_in, n_out, T = 96, 70, 60
    th = (-60.0 + 12.0 * synth.uniform_f32(9, (n_out,), 0.0, 1.0)).astype(np.float32)
    spikes = synth.dense_spikes(17, (T, B, n_in), 0.08)
    out = {}
    for dev in ("cpu", "cuda"):
        net = build(n_in, n_out, th.copy() if as_numpy else torch.from_numpy(th.copy()))
        mons = {"s": Monitor(net.layers["O"], ["s"], time=T), "v": Monitor(net.layers["O"], ["v"], time=T)}
        for k, m in mons.items():
            net.add_monitor(m, k)
        net.to(dev)
        for rep in range(2):                                   # (second run: the kept descriptors)
            net.run({"I": torch.from_numpy(spikes).to(dev)}, time=T)
        out[dev] = dict(s=mons["s"].get("s").cpu().numpy().astype(np.uint8), v=mons["v"].get("v").cpu().numpy(),
                        x=net.layers["O"].x.cpu().numpy(), plan=net.last_plan)
    assert out["cpu"]["plan"] == "host-torch" and out["cuda"]["plan"] == "generic"
    assert int(out["cpu"]["s"].sum()) > 20, "vacuous: no spikes"
    per_neuron = out["cpu"]["s"].reshape(-1, n_out).sum(0)
    assert per_neuron.max() > per_neuron.min(), "thresholds should make the neurons differ"
    np.testing.assert_array_equal(out["cuda"]["s"], out["cpu"]["s"])
    np.testing.assert_array_equal(out["cuda"]["v"].view(np.uint32), out["cpu"]["v"].view(np.uint32))
    np.testing.assert_array_equal(out["cuda"]["x"].view(np.uint32), out["cpu"]["x"].view(np.uint32))


def test_vector_threshold_hand_stepped_layer():
    n, B = 50, 3
    th = torch.from_numpy((-58.0 + 6.0 * synth.uniform_f32(2, (n,), 0.0, 1.0)).astype(np.float32))
    cur = torch.from_numpy(synth.uniform_f32(4, (12, B, n), 0.0, 9.0))
    got = {}
    for dev in ("cpu", "cuda"):
        l = LIFNodes(n=n, thresh=th.clone(), traces=True)
        l.compute_decays(1.0)
        l.set_batch_size(B)
        l.to(dev)
        l.set_batch_size(B)
        ss = []
        for t in range(cur.shape[0]):
            l.forward(cur[t].clone().to(dev))
            ss.append(l.s.cpu().numpy().astype(np.uint8).copy())
        got[dev] = (np.stack(ss), l.v.cpu().numpy().copy())
    assert got["cpu"][0].sum() > 0
    np.testing.assert_array_equal(got["cuda"][0], got["cpu"][0])
    np.testing.assert_array_equal(got["cuda"][1].view(np.uint32), got["cpu"][1].view(np.uint32))


def _fixture_network(dev):
    """The network of tests/golden/make_golden_r5.py (run_lif_vector_thresh), built from this package's classes."""
    import make_golden_r5_params as P
    from bindsnet_amd.network.topology import MulticompartmentConnection
    from bindsnet_amd.network.topology_features import Weight
    net = Network(dt=1.0, learning=False)
    net.add_layer(Input(n=P.n_in), "I")
    net.add_layer(LIFNodes(n=P.n_out, thresh=torch.from_numpy(P.thresholds()), refrac=2, tc_decay=50.0, traces=True), "O")
    w_in, w_rec = P.weights()
    net.add_connection(MulticompartmentConnection(net.layers["I"], net.layers["O"], device="cpu", pipeline=[Weight("weight", torch.from_numpy(w_in).clone())]), "I", "O")
    net.add_connection(MulticompartmentConnection(net.layers["O"], net.layers["O"], device="cpu", pipeline=[Weight("weight", torch.from_numpy(w_rec).clone())]), "O", "O")
    mon = Monitor(net.layers["O"], ["s", "v"], time=P.T)
    net.add_monitor(mon, "O")
    net.to(dev)
    return net, mon, P


def check_against_reference_fixture(dev, plan):
    from cases import gold
    g = gold("run_lif_vector_thresh")
    net, mon, P = _fixture_network(dev)
    for r in range(2):
        sp = synth.dense_spikes(17 + r, (P.T, P.B, P.n_in), 0.08)
        net.run({"I": torch.from_numpy(sp).to(dev)}, time=P.T)
        assert net.last_plan == plan
        np.testing.assert_array_equal(np.packbits(mon.get("s").cpu().numpy().astype(np.uint8)), g[f"r{r}_s"], err_msg=f"run {r}: raster")
        np.testing.assert_array_equal(mon.get("v").cpu().numpy().view(np.uint32), g[f"r{r}_v"].view(np.uint32), err_msg=f"run {r}: voltages")
        np.testing.assert_array_equal(net.layers["O"].x.cpu().numpy().view(np.uint32), g[f"r{r}_x"].view(np.uint32), err_msg=f"run {r}: trace")


def test_vector_threshold_run_matches_the_reference_fixture():
    """The REFERENCE's run of a network with per-neuron LIF thresholds (tests/golden/make_golden_r5.py; MCC + Weight connections, so its sums
    are ATen's reproducible cascade): spike raster, voltage raster and trace of two consecutive runs, bit for bit, on the device (generic plan)."""
    check_against_reference_fixture("cuda", "generic")


def test_wrong_length_is_refused():
    net = build(16, 10, torch.zeros(7))
    net.to("cuda")
    with pytest.raises(ValueError):
        net.run({"I": torch.zeros(4, 1, 16, dtype=torch.uint8, device="cuda")}, time=4)
