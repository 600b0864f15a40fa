"""Network.run() keeps its descriptor arrays from one call to the next (network/network.py).  Whatever a user changes
between two calls must still take effect: every case below runs the same sequence twice -- once with the cache, once
with SNN_DESC_CACHE=0 semantics (descriptors rebuilt on every call) -- and compares all results bit for bit."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
T, B, N = 40, 4, 64


def build():
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05,
                           inpt_shape=(1, 28, 28))
    net.add_monitor(Monitor(net.layers["Ae"], ["s", "v"], time=T), "Ae")
    return net.to(DEV)


def x(seed, b=B):
    return torch.from_numpy(synth.spike_train(seed, T, b, 784)).view(T, b, 1, 28, 28).to(DEV)


def snapshot(net):
    out = {"W": net.connections[("X", "Ae")].pipeline[0].value.detach().cpu().numpy().copy(),
           "theta": net.layers["Ae"].theta.cpu().numpy().copy(), "v": net.layers["Ae"].v.cpu().numpy().copy()}
    for name, m in net.monitors.items():
        for var in m.state_vars:
            out[f"{name}.{var}"] = m.get(var).float().cpu().numpy().copy()
    return out


def change_thresh_in_place(net):
    net.layers["Ae"].thresh.fill_(-54.0)


def change_nu_element(net):
    rule = net.connections[("X", "Ae")].pipeline[0].learning_rule
    if isinstance(rule.nu, torch.Tensor):
        rule.nu[1] = 0.05
    else:
        rule.nu = [rule.nu[0], 0.05]


def stop_learning(net):
    net.train(False)


def add_monitor(net):
    from bindsnet_amd.network.monitors import Monitor
    net.add_monitor(Monitor(net.layers["Ai"], ["s"], time=T), "Ai")


def move_and_back(net):
    net.to("cpu")
    net.to(DEV)


def new_weights(net):
    feat = net.connections[("X", "Ae")].pipeline[0]
    feat.value = torch.nn.Parameter(0.2 * torch.ones_like(feat.value), requires_grad=False)


def change_rest_in_place(net):
    net.layers["Ai"].rest.fill_(-61.0)


def other_batch(net):
    pass            # (the next input has another batch size: see `sequence`)


def swap_weight_data(net):
    """`.data = other` re-homes the tensor without any attribute assignment on a network object."""
    feat = net.connections[("X", "Ae")].pipeline[0]
    feat.value.data = 0.2 * torch.ones_like(feat.value)


def swap_theta_data(net):
    th = net.layers["Ae"].theta
    th.data = torch.full_like(th, 0.3)


def load_state(net):
    """load_state_dict copies into the parameters / buffers IN PLACE."""
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    changed = 0
    for k in sd:
        if k.endswith("thresh"):
            sd[k] = sd[k] - 2.0
            changed += 1
    assert changed
    net.load_state_dict(sd)


CHANGES = [change_thresh_in_place, change_nu_element, stop_learning, add_monitor, move_and_back, new_weights,
           change_rest_in_place, other_batch, swap_weight_data, swap_theta_data, load_state]


def sequence(change, cached):
    from bindsnet_amd.network import network as netmod
    old = netmod._DESC_CACHE
    netmod._DESC_CACHE = cached
    try:
        net = build()
        torch.manual_seed(5)
        shots = []
        for k in range(2):                      # two calls so that the second one runs on kept descriptors
            net.run({"X": x(10 + k)}, time=T)
            shots.append(snapshot(net))
            net.reset_state_variables()
        if cached:
            assert net.__dict__.get("_run_cache") is not None
        change(net)
        for k in range(2):
            net.run({"X": x(20 + k, 2 if change is other_batch else B)}, time=T)
            shots.append(snapshot(net))
            net.reset_state_variables()
        shots.append({"probe": torch.rand(3).numpy()})
        return shots
    finally:
        netmod._DESC_CACHE = old


@pytest.mark.parametrize("change", CHANGES, ids=lambda f: f.__name__)
def test_change_between_calls_takes_effect(change):
    a, b = sequence(change, True), sequence(change, False)
    for k, (sa, sb) in enumerate(zip(a, b)):
        assert sa.keys() == sb.keys()
        for key in sa:
            np.testing.assert_array_equal(sa[key].view(np.uint32) if sa[key].dtype == np.float32 else sa[key],
                                          sb[key].view(np.uint32) if sb[key].dtype == np.float32 else sb[key],
                                          err_msg=f"{change.__name__}: call {k}, {key}")


def test_kept_descriptors_are_used_and_dropped():
    from bindsnet_amd import _lib
    net = build()
    net.run({"X": x(1)}, time=T)
    first = net.__dict__["_run_cache"]
    net.reset_state_variables()
    net.run({"X": x(2)}, time=T)
    assert net.__dict__["_run_cache"] is first, "second call of the same shape re-uses the arrays"
    net.layers["Ae"].one_spike = False          # any assignment to a network object
    net.run({"X": x(3)}, time=T)
    assert net.__dict__["_run_cache"] is not first
    second = net.__dict__["_run_cache"]
    net.run({"X": x(4)}, time=T // 2)           # another duration
    assert net.__dict__["_run_cache"] is not second
    net.run({"X": x(5)}, time=T // 2, clamp={"Ae": torch.zeros(N, dtype=torch.bool)})
    assert net.__dict__["_run_cache"] is None, "calls with keyword arguments are never kept"
    e = _lib.epoch()
    net.layers["Ae"].v.add_(1.0)                # state tensors are addressed, not copied: in-place edits need no rebuild
    assert e == _lib.epoch()


@pytest.mark.parametrize("edit", ["wmax_in_place", "w_data_swap"])
def test_dense_connection_edits_between_calls_take_effect(edit):
    """Dense Connection: its clamp bounds are Parameters (read into the descriptors as floats) and its weights can be
    re-homed through `.data`; both must reach a run that would otherwise re-use the kept descriptors."""
    from bindsnet_amd.models import TwoLayerNetwork
    from bindsnet_amd.network import network as netmod

    def seq(cached):
        old = netmod._DESC_CACHE
        netmod._DESC_CACHE = cached
        try:
            torch.manual_seed(0)
            net = TwoLayerNetwork(n_inpt=784, n_neurons=64, reduction=torch.sum).to(DEV)
            conn = net.connections[("X", "Y")]
            xs = [torch.from_numpy(synth.dense_spikes(30 + k, (T, B, 784), 0.05)).to(DEV) for k in range(4)]
            out = []
            for k in range(2):
                net.run({"X": xs[k]}, time=T)
                net.reset_state_variables()
            if edit == "wmax_in_place":
                with torch.no_grad():
                    conn.wmax.fill_(0.05)           # far below the learned weights: the clamp must bite in the next run
            else:
                conn.w.data = 0.1 * torch.ones_like(conn.w)
            for k in range(2, 4):
                net.run({"X": xs[k]}, time=T)
                out.append(conn.w.detach().cpu().numpy().copy())
                net.reset_state_variables()
            return out
        finally:
            netmod._DESC_CACHE = old

    a, b = seq(True), seq(False)
    for wa, wb in zip(a, b):
        np.testing.assert_array_equal(wa.view(np.uint32), wb.view(np.uint32))
