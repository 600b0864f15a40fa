"""Fused plan `convlif-fused` (csrc/snn_convlif.hip): whole run of Input -> Conv2dConnection -> LIFNodes in one
launch.  Must be bit-identical to the generic plan (k_conv2d + k_lif per step), which tests/test_gpu_network.py
checks against the oracle."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
u8 = np.uint8

#        name: (B, T, Cin, H, W, Cout, k, stride, pad, density, value_max, traces, voltage monitor)
CASES = {
    "cfg4_shape": (4, 40, 1, 28, 28, 32, 5, 1, 0, 0.25, 1, False, False),
    "stride2_pad1_traces": (3, 30, 1, 17, 19, 12, 3, 2, 1, 0.3, 1, True, True),
    "two_input_channels_odd_cout": (2, 25, 2, 12, 12, 5, 3, 1, 0, 0.3, 1, True, False),
    "multivalued_spike_bytes": (2, 20, 1, 16, 16, 8, 5, 1, 2, 0.2, 3, False, True),
    "big_image_many_tiles": (1, 10, 1, 40, 40, 9, 3, 1, 0, 0.2, 1, False, False),
}


def run(mode, case, n_runs=2):
    from bindsnet_amd import _lib
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Conv2dConnection
    B, T, Cin, H, W, Cout, k, stride, pad, dens, vmax, traces, vmon = case
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    _lib.lib().snn_set_plan_mode(mode)
    try:
        net = Network(dt=1.0, learning=False)
        net.add_layer(Input(shape=(Cin, H, W), traces=traces), "X")
        net.add_layer(LIFNodes(shape=(Cout, OH, OW), traces=traces), "Y")
        w = torch.from_numpy(synth.uniform_f32(7, (Cout, Cin, k, k), -0.1, 0.5))
        net.add_connection(Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=k, stride=stride, padding=pad, w=w), "X", "Y")
        mons = {"s": Monitor(net.layers["Y"], ["s"], time=T)}
        if vmon:
            mons["v"] = Monitor(net.layers["Y"], ["v"], time=T)
        for n_, m in mons.items():
            net.add_monitor(m, n_)
        net.to(DEV)
        out = []
        rs = np.random.RandomState(3)
        for r in range(n_runs):
            sp = synth.dense_spikes(60 + r, (T, B, Cin, H, W), dens)
            if vmax > 1:
                sp = (sp * rs.randint(1, vmax + 1, size=sp.shape)).astype(u8)
            net.run({"X": torch.from_numpy(sp).to(DEV)}, time=T)
            Y = net.layers["Y"]
            st = dict(s=mons["s"].get("s").cpu().numpy().copy(), v=Y.v.cpu().numpy().copy(), r=Y.refrac_count.cpu().numpy().copy(),
                      sl=Y.s.cpu().numpy().copy())
            if traces:
                st["xY"] = Y.x.cpu().numpy().copy(); st["xX"] = net.layers["X"].x.cpu().numpy().copy()
            if vmon:
                st["vm"] = mons["v"].get("v").cpu().numpy().copy()
            out.append(st)
            plan = net.last_plan
            if r == 0:
                pass                                   # second run continues from the first one's state
        return out, plan
    finally:
        _lib.lib().snn_set_plan_mode(0)


@pytest.mark.parametrize("name", list(CASES))
def test_convlif_fused_equals_generic(name):
    fused, plan = run(0, CASES[name])
    assert plan == "convlif-fused"
    generic, plan_g = run(1, CASES[name])
    assert plan_g == "generic"
    for r, (a, b) in enumerate(zip(fused, generic)):
        for k in a:
            np.testing.assert_array_equal(a[k].view(u8), b[k].view(u8), err_msg=f"run {r}: {k}")
    assert sum(int(x["s"].sum()) for x in fused) > 0, "no output spike: vacuous"
