"""Network.pipelined(): run() returns without waiting for the device, the host generator lives on the device between the runs of the
section, an input the lean kernel form gives up on is repeated on the general form ON THE DEVICE (snn_run_desc.status2, ABI 8).

Bar: everything a section leaves behind -- rasters of every input, weights, thresholds, membrane potentials, traces, the position of the
host generator -- equals what the same calls leave behind one synchronous run() at a time, bit for bit; and that in turn equals the CPU
oracle (oracle/snn_oracle.c) where the case is one it covers."""
import os

import numpy as np
import pytest
import torch

import synth
import test_gpu_fused_stress as dc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def go(pipelined, N, B, T, spikes, depth=64, w_scale=0.6, exc=22.5, reset=True, mode=0, seed=11):
    from bindsnet_amd import _lib
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    import contextlib
    _lib.lib().snn_set_plan_mode(int(mode))
    try:
        torch.manual_seed(0)
        net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=exc, inh=120.0, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
        net.connections[("X", "Ae")].pipeline[0].value.data.copy_(torch.from_numpy(np.minimum(synth.uniform_f32(3, (784, N), 0.0, w_scale), 1.0)))
        mons = {l: Monitor(net.layers[l], ["s"], time=T) for l in ("X", "Ae", "Ai")}
        for l, m in mons.items():
            net.add_monitor(m, l)
        net.to(DEV)
        xs = [torch.from_numpy(s).view(T, B, 1, 28, 28).to(DEV) for s in spikes]
        torch.manual_seed(seed)                              # ONCE: the section must not be interrupted by the caller's own draws
        dev_out, plans = [], []
        with (net.pipelined(depth=depth) if pipelined else contextlib.nullcontext()):
            for x in xs:
                net.run({"X": x}, time=T)
                plans.append(net.last_plan)
                dev_out.append(dict(sX=mons["X"].get("s").clone(), sE=mons["Ae"].get("s").clone(), sI=mons["Ai"].get("s").clone(),
                                    W=net.connections[("X", "Ae")].pipeline[0].value.detach().clone(), theta=net.layers["Ae"].theta.clone(),
                                    vE=net.layers["Ae"].v.clone(), xE=net.layers["Ae"].x.clone(), vI=net.layers["Ai"].v.clone()))
                if reset:
                    net.reset_state_variables()
        probe = torch.rand(3).numpy()                        # where the host generator stands afterwards
        out = [{k: v.cpu().numpy() for k, v in d.items()} for d in dev_out]
        for r, d in enumerate(out):                           # the Input layer's raster is a copy of the input (made by the launch's producer workgroups
            np.testing.assert_array_equal(d["sX"].reshape(T, B, 784).astype(np.uint8), spikes[r], err_msg=f"input {r}: X raster")   # or one device copy)
        out[-1]["probe"] = probe
        return out, plans, net
    finally:
        _lib.lib().snn_set_plan_mode(0)


def same(a, b, what):
    assert len(a) == len(b)
    for r, (x, y) in enumerate(zip(a, b)):
        for k in x:
            np.testing.assert_array_equal(np.ascontiguousarray(x[k]).reshape(-1).view(np.uint8), np.ascontiguousarray(y[k]).reshape(-1).view(np.uint8),
                                          err_msg=f"{what}: input {r}: {k}")


@pytest.mark.parametrize("depth", [64, 2])
def test_section_equals_synchronous_runs(depth):
    N, B, T = 100, 8, 40
    spikes = [synth.dense_spikes(500 + r, (T, B, 784), 0.02) for r in range(5)]
    p, plans_p, net = go(True, N, B, T, spikes, depth=depth)
    s, plans_s, _ = go(False, N, B, T, spikes)
    assert plans_p == plans_s == ["dc2015-resident-lean"] * 5
    assert sum(int(x["sE"].sum()) for x in s) > 0
    assert net.__dict__.get("lean_retries", 0) == 0
    same(p, s, f"pipelined (depth {depth}) vs synchronous")


def test_section_without_resets_carries_state_and_generator():
    """No reset between the inputs: run k+1 starts from run k's spikes, traces, refractory counters -- and from the generator position
    run k left on the device.  Third party: the CPU oracle (same schedule as test_gpu_fused_stress.oracle_run without its resets)."""
    import oracle
    N, B, T = 64, 4, 30
    spikes = [synth.dense_spikes(700 + r, (T, B, 784), 0.03) for r in range(3)]
    p, _, _ = go(True, N, B, T, spikes, reset=False, w_scale=1.0)
    s, _, _ = go(False, N, B, T, spikes, reset=False, w_scale=1.0)
    same(p, s, "pipelined vs synchronous, no resets")
    P = oracle.eth_mnist_dc_params(N, B, T)
    st = oracle.eth_mnist_dc_state(N, B, np.minimum(synth.uniform_f32(3, (784, N), 0.0, 1.0), 1.0).astype(np.float32))
    Q, cur = oracle.exp_noise(11, 3 * B * N * T + 16), np.zeros(1, np.int64)
    for r in range(3):
        sE, sI = oracle.run_dc2015(P, st, np.ascontiguousarray(spikes[r].reshape(T, B, 784)), Q, cur)
        np.testing.assert_array_equal(p[r]["sE"].reshape(T, B, N).astype(np.uint8), sE, err_msg=f"input {r}: Ae raster vs oracle")
        np.testing.assert_array_equal(p[r]["W"].view(np.uint32), st["W_xe"].view(np.uint32), err_msg=f"input {r}: W vs oracle")
        np.testing.assert_array_equal(p[r]["theta"].view(np.uint32), st["theta"].view(np.uint32), err_msg=f"input {r}: theta vs oracle")
    assert int(cur[0]) > 0, "vacuous: no arbitration draw"
    torch.manual_seed(11)
    torch.empty(int(cur[0])).exponential_(1)
    np.testing.assert_array_equal(p[-1]["probe"], torch.rand(3).numpy(), err_msg="host generator position after the section vs oracle")


def test_lean_give_up_is_repeated_on_the_device():
    """exc = 10: an Ai neuron does not follow its Ae partner -> the lean form gives up (SNN_ERR_RETRY, nothing written); inside a section the
    general form runs right behind it on the device.  Same results as the synchronous path (host-driven second attempt) and the oracle."""
    N, B, T = 100, 8, 40
    spikes = [synth.dense_spikes(4200 + r, (T, B, 784), 0.05) for r in range(3)]
    p, plans, net = go(True, N, B, T, spikes, w_scale=0.8, exc=10.0)
    assert net.lean_retries >= 1, "the lean form was expected to give up"
    s, _, net_s = go(False, N, B, T, spikes, w_scale=0.8, exc=10.0)
    assert net_s.lean_retries >= 1
    g, _, _ = go(False, N, B, T, spikes, w_scale=0.8, exc=10.0, mode=1)
    assert sum(int(x["sE"].sum()) for x in g) > 0
    same(p, s, "pipelined vs synchronous (give-up path)")
    same(p, g, "pipelined vs generic plan (give-up path)")


def test_generic_plan_in_a_section():
    N, B, T = 36, 4, 25
    spikes = [synth.dense_spikes(900 + r, (T, B, 784), 0.03) for r in range(3)]
    p, plans, _ = go(True, N, B, T, spikes, mode=1, w_scale=1.0)
    s, _, _ = go(False, N, B, T, spikes, mode=1, w_scale=1.0)
    assert plans == ["generic"] * 3
    same(p, s, "generic plan, pipelined vs synchronous")


def test_a_synchronous_user_of_the_host_generator_settles_the_section():
    """Anything in the package that reads the host generator (here: a hand-stepped DiehlAndCookNodes.forward through rng.NoiseStream)
    settles open sections first, so it sees the position the reference would."""
    from bindsnet_amd import rng
    N, B, T = 64, 4, 30
    spikes = [synth.dense_spikes(700 + r, (T, B, 784), 0.03) for r in range(2)]
    _, _, net = go(False, N, B, T, spikes, w_scale=1.0)
    torch.manual_seed(11)
    with net.pipelined():
        net.run({"X": torch.from_numpy(spikes[0]).view(T, B, 1, 28, 28).to(DEV)}, time=T)
        assert net._pipe.pending
        rng.flush_pending()
        assert not net._pipe.pending
        a = torch.rand(3).numpy()
    torch.manual_seed(11)
    net2 = go(False, N, B, T, spikes, w_scale=1.0)[2]
    torch.manual_seed(11)
    net2.run({"X": torch.from_numpy(spikes[0]).view(T, B, 1, 28, 28).to(DEV)}, time=T)
    # (both networks had run the same two inputs before: same weights, same state)
    np.testing.assert_array_equal(a, torch.rand(3).numpy())


def test_timeout_inside_a_section_raises(monkeypatch):
    """A resident grid that is not co-resident in time (test hook: one workgroup never starts) ends with SNN_ERR_TIMEOUT; the synchronous
    path repeats the input on the per-step plan, a section cannot (later runs are already enqueued) and says so."""
    from bindsnet_amd import _lib
    N, B, T = 100, 8, 10
    spikes = [synth.dense_spikes(500 + r, (T, B, 784), 0.02) for r in range(2)]
    monkeypatch.setenv("SNN_DC_TEST_STALL", "3")
    with pytest.raises(_lib.SnnError, match="pipelined run 1 of"):
        go(True, N, B, T, spikes)
    monkeypatch.delenv("SNN_DC_TEST_STALL")
    p, _, _ = go(True, N, B, T, spikes)                     # the process is fine afterwards
    s, _, _ = go(False, N, B, T, spikes)
    same(p, s, "after the refused section")


def _two_nets(N, B, T):
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    nets = []
    for k in range(2):
        torch.manual_seed(k)
        net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120.0, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
        net.connections[("X", "Ae")].pipeline[0].value.data.copy_(torch.from_numpy(np.minimum(synth.uniform_f32(3 + k, (784, N), 0.0, 1.0), 1.0)))
        net.add_monitor(Monitor(net.layers["Ae"], ["s"], time=T), "Ae")
        net.to(DEV)
        nets.append(net)
    return nets


def test_two_interleaved_sections_share_the_host_generator_like_synchronous_runs():
    """(round-5 advisor) Two networks with open sections, both drawing for one_spike, runs interleaved: each batch start must settle the OTHER
    network's section before it snapshots the host generator -- otherwise both consume the same mt19937 segment.  Against synchronous runs."""
    import contextlib
    N, B, T = 64, 4, 30
    spikes = [synth.dense_spikes(1300 + r, (T, B, 784), 0.03) for r in range(4)]

    def play(sections):
        a, b = _two_nets(N, B, T)
        xs = [torch.from_numpy(s).view(T, B, 1, 28, 28).to(DEV) for s in spikes]
        torch.manual_seed(5)
        out = []
        with (a.pipelined() if sections else contextlib.nullcontext()), (b.pipelined() if sections else contextlib.nullcontext()):
            for r, x in enumerate(xs):
                net = (a, b)[r & 1]
                net.run({"X": x}, time=T)
                out.append(net.monitors["Ae"].get("s").clone())
                net.reset_state_variables()
        probe = torch.rand(3).numpy()
        return [o.cpu().numpy() for o in out], [n.connections[("X", "Ae")].pipeline[0].value.detach().cpu().numpy() for n in (a, b)], probe

    sp, wp, pp = play(True)
    ss, ws, ps = play(False)
    assert sum(int(o.sum()) for o in ss) > 0
    for r in range(4):
        np.testing.assert_array_equal(sp[r], ss[r], err_msg=f"run {r}: Ae raster, interleaved sections vs synchronous")
    for k in range(2):
        np.testing.assert_array_equal(wp[k].view(np.uint32), ws[k].view(np.uint32), err_msg=f"network {k}: weights")
    np.testing.assert_array_equal(pp, ps, err_msg="host generator position after both sections")


def test_host_draw_inside_a_section_is_refused_and_host_encoders_settle_first():
    """(round-5 advisor) Inside a section the host generator is stale.  A caller's own torch.rand between two runs would be taken from the wrong
    position and silently discarded at settlement: sync() raises instead.  The package's own host encoders settle the section before they draw,
    so the canonical encode-then-run loop inside a section gives the synchronous loop's results."""
    import contextlib
    from bindsnet_amd.encoding import poisson
    N, B, T = 64, 4, 30
    spikes = [synth.dense_spikes(1500 + r, (T, B, 784), 0.03) for r in range(2)]
    (net, _) = _two_nets(N, B, T)
    xs = [torch.from_numpy(s).view(T, B, 1, 28, 28).to(DEV) for s in spikes]
    torch.manual_seed(5)
    with pytest.raises(RuntimeError, match="CPU generator was used inside the section"):
        with net.pipelined():
            net.run({"X": xs[0]}, time=T)
            torch.rand(2)
            net.run({"X": xs[1]}, time=T)

    def loop(section):
        (n2, _) = _two_nets(N, B, T)
        torch.manual_seed(7)
        img = 128.0 * torch.rand(B, 1, 28, 28) * (torch.rand(B, 1, 28, 28) < 0.19)
        outs = []
        with (n2.pipelined() if section else contextlib.nullcontext()):
            for _ in range(3):
                x = torch.stack([poisson(img[b], time=T, dt=1.0) for b in range(B)], 1).to(DEV)     # host encoder: draws from the CPU generator
                n2.run({"X": x}, time=T)
                outs.append(n2.monitors["Ae"].get("s").clone())
                n2.reset_state_variables()
        return [o.cpu().numpy() for o in outs], torch.rand(3).numpy()

    a, pa = loop(True)
    b, pb = loop(False)
    for r in range(3):
        np.testing.assert_array_equal(a[r], b[r], err_msg=f"input {r}: encode-then-run inside a section vs synchronous")
    np.testing.assert_array_equal(pa, pb)
