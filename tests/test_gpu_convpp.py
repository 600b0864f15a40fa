"""Fused plan `convpp-fused` (csrc/snn_convlif.hip, round 6): the whole run of Input -> Conv2dConnection with PostPre (learning.py:457-497) ->
LIFNodes -- conv_mnist.py's training graph -- in ONE cooperative launch.  Must be bit-identical to the generic plan (five launches per
timestep: k_input, k_conv2d, k_lif, k_conv_pp_partial_ev, k_conv_pp_apply), which tests/test_gpu_extras.py / test_gpu_network.py check
against the oracle and the reference's fixtures: spike rasters, voltages, every state tensor, both traces and the weights, over two
consecutive runs."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
u8 = np.uint8

#        name: (B, T, Cin, H, W, Cout, k, stride, pad, density, value_max, nu, wmin, wmax, weight_decay, voltage monitor, learning)
CASES = {
    "conv_mnist_shape_b16": (16, 30, 1, 28, 28, 32, 5, 1, 0, 0.05, 1, (1e-4, 1e-2), None, None, 0.0, False, True),
    "dense_input_clamped": (4, 25, 1, 28, 28, 32, 5, 1, 0, 0.3, 1, (1e-3, 1e-2), -0.2, 0.6, 0.0, False, True),
    "stride2_pad1_b3": (3, 30, 1, 17, 19, 12, 3, 2, 1, 0.3, 1, (1e-3, 2e-2), 0.0, 1.0, 0.0, True, True),
    "two_input_channels_odd_cout": (5, 25, 2, 12, 12, 5, 3, 1, 0, 0.3, 1, (2e-3, 1e-2), None, None, 0.0, False, True),
    "multivalued_spike_bytes": (2, 20, 1, 16, 16, 8, 5, 1, 2, 0.2, 3, (1e-3, 1e-2), None, 0.8, 0.0, True, True),
    "weight_decay_b17": (17, 15, 1, 14, 14, 9, 3, 1, 0, 0.25, 1, (1e-3, 1e-2), None, None, 0.01, False, True),
    "only_pre_term": (4, 20, 1, 16, 16, 8, 5, 1, 0, 0.3, 1, (1e-3, 0.0), None, None, 0.0, False, True),
    "only_post_term": (4, 20, 1, 16, 16, 8, 5, 1, 0, 0.3, 1, (0.0, 1e-2), None, None, 0.0, False, True),
    "learning_off": (4, 20, 1, 16, 16, 8, 5, 1, 0, 0.3, 1, (1e-3, 1e-2), None, None, 0.0, False, False),
    "b33_tail_elements": (33, 12, 1, 12, 12, 3, 3, 1, 0, 0.3, 1, (1e-3, 1e-2), None, None, 0.0, False, True),
}


def run(mode, case, n_runs=2):
    from bindsnet_amd import _lib
    from bindsnet_amd.learning import PostPre
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Conv2dConnection
    B, T, Cin, H, W, Cout, k, stride, pad, dens, vmax, nu, wmin, wmax, wd, vmon, learning = case
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    _lib.lib().snn_set_plan_mode(mode)
    try:
        net = Network(dt=1.0, batch_size=B, learning=learning)
        net.add_layer(Input(shape=(Cin, H, W), traces=True), "X")
        net.add_layer(LIFNodes(shape=(Cout, OH, OW), traces=True), "Y")
        w = torch.from_numpy(synth.uniform_f32(7, (Cout, Cin, k, k), -0.1, 0.5))
        kw = dict(kernel_size=k, stride=stride, padding=pad, w=w, update_rule=PostPre, nu=nu, weight_decay=wd)
        if wmin is not None:
            kw["wmin"] = wmin
        if wmax is not None:
            kw["wmax"] = wmax
        net.add_connection(Conv2dConnection(net.layers["X"], net.layers["Y"], **kw), "X", "Y")
        mons = {"s": Monitor(net.layers["Y"], ["s"], time=T)}
        if vmon:
            mons["v"] = Monitor(net.layers["Y"], ["v"], time=T)
        for n_, m in mons.items():
            net.add_monitor(m, n_)
        net.to(DEV)
        out = []
        rs = np.random.RandomState(3)
        plan = None
        for r in range(n_runs):
            sp = synth.dense_spikes(60 + r, (T, B, Cin, H, W), dens)
            if vmax > 1:
                sp = (sp * rs.randint(1, vmax + 1, size=sp.shape)).astype(u8)
            net.run({"X": torch.from_numpy(sp).to(DEV)}, time=T)
            Y = net.layers["Y"]
            st = dict(s=mons["s"].get("s").cpu().numpy().copy(), v=Y.v.cpu().numpy().copy(), r=Y.refrac_count.cpu().numpy().copy(),
                      sl=Y.s.cpu().numpy().copy(), xY=Y.x.cpu().numpy().copy(), xX=net.layers["X"].x.cpu().numpy().copy(),
                      w=net.connections[("X", "Y")].w.detach().cpu().numpy().copy())
            if vmon:
                st["vm"] = mons["v"].get("v").cpu().numpy().copy()
            out.append(st)
            plan = net.last_plan
        return out, plan
    finally:
        _lib.lib().snn_set_plan_mode(0)


@pytest.mark.parametrize("name", list(CASES))
def test_convpp_fused_equals_generic(name):
    fused, plan = run(0, CASES[name])
    assert plan == "convpp-fused"
    generic, plan_g = run(1, CASES[name])
    assert plan_g == "generic"
    for r, (a, b) in enumerate(zip(fused, generic)):
        for k in a:
            np.testing.assert_array_equal(a[k].view(u8), b[k].view(u8), err_msg=f"run {r}: {k}")
    assert sum(int(x["s"].sum()) for x in fused) > 0, "no output spike: vacuous"
    if CASES[name][-1]:
        w0 = synth.uniform_f32(7, fused[0]["w"].shape, -0.1, 0.5)
        assert not np.array_equal(fused[-1]["w"], w0), "the weights never moved: vacuous"


@pytest.mark.parametrize("cc", [2, 4, 8])
@pytest.mark.parametrize("name", ["conv_mnist_shape_b16", "two_input_channels_odd_cout", "stride2_pad1_b3"])
def test_every_chunk_size_of_the_fused_plan_equals_generic(name, cc, monkeypatch):
    """The launch picks 2, 4 or 8 output channels per workgroup by what is co-resident; SNN_CONVPP_CC forces one: each instantiation against the
    generic plan."""
    generic, plan_g = run(1, CASES[name])
    assert plan_g == "generic"
    monkeypatch.setenv("SNN_CONVPP_CC", str(cc))
    fused, plan = run(0, CASES[name])
    assert plan == "convpp-fused"
    for r, (a, b) in enumerate(zip(fused, generic)):
        for k in a:
            np.testing.assert_array_equal(a[k].view(u8), b[k].view(u8), err_msg=f"cc {cc} run {r}: {k}")


def test_a_workgroup_that_never_arrives_leaves_every_state_untouched_and_the_run_is_repeated_on_the_generic_plan(monkeypatch):
    """SNN_CONVPP_TEST_STALL makes one workgroup of the cooperative grid return at once: its chunk's workgroups run out of their bounded polls, every
    OTHER chunk gets through its T steps -- and must not write anything back (the commit agreement at the end of the kernel); the status word says
    SNN_ERR_TIMEOUT, Network.run repeats the input on the per-operator plan: same bits as the generic plan from the start."""
    case = CASES["dense_input_clamped"]
    generic, plan_g = run(1, case, n_runs=1)
    monkeypatch.setenv("SNN_CONVPP_TEST_STALL", "3")
    fused, plan = run(0, case, n_runs=1)
    assert plan == "generic", plan                      # the plan of the attempt that succeeded
    for k in fused[0]:
        np.testing.assert_array_equal(fused[0][k].view(u8), generic[0][k].view(u8), err_msg=k)
