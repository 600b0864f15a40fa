"""BASELINE configs[1] at FULL size (D&C 784->400, batch 32, T = 250, PostPre on) on the GPU:
 * the fused plan and the generic per-operator plan are bit-identical (rasters, weights, theta, state,
   generator position) over three consecutive inputs -- two independent implementations of the same order;
 * (the comparison with the reference itself at this size lives in tests/test_gpu_baseline_configs.py);
 * size-independent properties of the domain hold: at most one excitatory spike per sample per step
   (one_spike), weights stay in [wmin, wmax] before normalisation and every column sums to `norm` after it,
   theta only grows by multiples of theta_plus, refractory counters stay in range."""
import numpy as np
import pytest
import torch

import cases
import oracle
import synth
from cases import u8
from test_oracle_golden import dc_params

pytestmark = pytest.mark.gpu
DEV = "cuda"
N, B, T = 400, 32, 250


def build(plan_generic):
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05,
                           inpt_shape=(1, 28, 28))
    net.connections[("X", "Ae")].pipeline[0].value.data.copy_(torch.from_numpy(synth.weights_q12(10, 784, N)))
    mons = {l: Monitor(net.layers[l], ["s"], time=T) for l in ("Ae", "Ai")}
    for l, m in mons.items():
        net.add_monitor(m, l)
    net.to(DEV)
    return net, mons


def run_plan(mode, n_inputs=3):
    from bindsnet_amd import _lib
    _lib.lib().snn_set_plan_mode(mode)                # 0 resident kernel (lean form), 1 generic, 2 one launch per timestep, 3 resident (general form)
    try:
        net, mons = build(mode)
        out = []
        for r in range(n_inputs):
            spikes = synth.spike_train(50 + r, T, B, 784)
            torch.manual_seed(7 + r)
            net.run({"X": torch.from_numpy(spikes).view(T, B, 1, 28, 28).to(DEV)}, time=T)
            probe = torch.rand(2)
            st = {k: v.detach().cpu().numpy().copy() for k, v in dict(
                W=net.connections[("X", "Ae")].pipeline[0].value, theta=net.layers["Ae"].theta, vE=net.layers["Ae"].v,
                rE=net.layers["Ae"].refrac_count, xE=net.layers["Ae"].x, xX=net.layers["X"].x, vI=net.layers["Ai"].v,
                rI=net.layers["Ai"].refrac_count).items()}
            st["sE"] = mons["Ae"].get("s").cpu().numpy().reshape(T, B, N).astype(u8)
            st["sI"] = mons["Ai"].get("s").cpu().numpy().reshape(T, B, N).astype(u8)
            st["probe"] = probe.numpy()
            out.append(st)
            assert net.last_plan == ("dc2015-resident-lean", "generic", "dc2015-fused", "dc2015-resident")[mode]
            net.reset_state_variables()
        return out
    finally:
        _lib.lib().snn_set_plan_mode(0)


def test_fused_equals_generic_and_properties_at_full_size():
    fused, generic, stepped, general = run_plan(0), run_plan(1), run_plan(2), run_plan(3)
    for other in (generic, stepped, general):
        for r, (a, b) in enumerate(zip(fused, other)):
            for k in a:
                np.testing.assert_array_equal(a[k].view(np.uint8), b[k].view(np.uint8), err_msg=f"input {r}: {k}")
    for r, a in enumerate(fused):
        assert a["sE"].sum(axis=2).max() <= 1, "one_spike violated"
        assert a["sE"].sum() > 50 and a["sI"].sum() > 50, "network is silent: test is vacuous"
        np.testing.assert_allclose(a["W"].sum(0), 78.4, rtol=1e-5)          # normalised columns
        assert a["W"].min() >= 0.0
        assert (a["rE"] <= 5).all() and (a["rI"] <= 2).all()
        assert (a["theta"] >= 0).all() and a["theta"].max() > 0
    # theta grows monotonically across inputs (tc_theta_decay = 1e7)
    assert (fused[2]["theta"] >= fused[0]["theta"] * 0.999).all()


def _final_state(net, mons):
    return {
        "W": net.connections[("X", "Ae")].pipeline[0].value.detach().cpu().numpy().copy(),
        "theta": net.layers["Ae"].theta.detach().cpu().numpy().copy(),
        "sE": mons["Ae"].get("s").cpu().numpy().reshape(T, B, N).astype(u8),
        "sI": mons["Ai"].get("s").cpu().numpy().reshape(T, B, N).astype(u8),
    }


def test_resident_kernel_soak_and_competing_load():
    """The resident plan hands spikes between workgroups INSIDE one launch (tagged granules).  Run it many
    times back to back, with another stream keeping the chip busy (uneven load, delayed workgroup
    start), and require bit-identical results to the one-launch-per-timestep form run on an idle chip."""
    from bindsnet_amd import _lib
    n_inputs = 12
    outs = {}
    for mode in (2, 0, 3):
        _lib.lib().snn_set_plan_mode(mode)
        try:
            net, mons = build(mode)
            side = torch.cuda.Stream()
            a = torch.randn(2048, 2048, device=DEV)
            for r in range(n_inputs):
                spikes = torch.from_numpy(synth.spike_train(300 + r, T, B, 784)).view(T, B, 1, 28, 28).to(DEV)
                torch.manual_seed(40 + r)
                if mode != 2:
                    with torch.cuda.stream(side):      # ~10 ms of GEMMs racing the resident kernel for CUs
                        for _ in range(40):
                            a = torch.tanh(a @ a * 1e-3)
                net.run({"X": spikes}, time=T)
                net.reset_state_variables() if r % 3 == 0 else None
            torch.cuda.synchronize()
            assert net.last_plan == ("dc2015-resident-lean", "generic", "dc2015-fused", "dc2015-resident")[mode]
            outs[mode] = _final_state(net, mons)
        finally:
            _lib.lib().snn_set_plan_mode(0)
    for k in outs[0]:
        np.testing.assert_array_equal(outs[0][k].view(np.uint8), outs[2][k].view(np.uint8), err_msg=k)
        np.testing.assert_array_equal(outs[3][k].view(np.uint8), outs[2][k].view(np.uint8), err_msg=k)
    assert outs[0]["sE"].sum() > 20
