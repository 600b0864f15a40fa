"""Stress the rarely-taken paths of the fused D&C plan by comparing it bit-for-bit with the generic
per-operator plan (itself pinned to the reference):
  * dense inputs (hundreds of active sources per sample -> "busy" generic bit-scan path inside the kernel),
  * spike bytes other than 0/1 (value-weighted propagation and STDP),
  * very strong drive (every neuron of every sample crosses threshold at once: thousands of one_spike
    candidates, dozens of generator twists in one step -> serial fallback of the arbitration),
  * odd sizes (N not a multiple of 8 / 32, B < 32, tiny T), learning off, shapes the plan must refuse."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def run(mode, N, B, T, spikes, w_scale=0.3, n_inputs=2, learning=True, Nin=784, shape=(1, 28, 28), inh=120.0, additive=False, nu=(1e-4, 1e-2), exc=22.5,
        tweak=None, pipelined=False):
    """pipelined: the same calls inside ONE Network.pipelined() section (settled before every re-seed of the host generator, as a section asks
    of a caller that touches it): ordinary launches, per-run status pairs, the second attempt on the device, no memsets between the runs."""
    import contextlib
    from bindsnet_amd import _lib
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    _lib.lib().snn_set_plan_mode(int(mode))          # 0 auto (resident kernel), 1 generic, 2 one launch per timestep
    net = None
    try:
        torch.manual_seed(0)
        net = DiehlAndCook2015(n_inpt=Nin, n_neurons=N, exc=exc, inh=inh, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=shape, nu=nu)
        W0 = synth.uniform_f32(3, (Nin, N), 0.0, w_scale)
        net.connections[("X", "Ae")].pipeline[0].value.data.copy_(torch.from_numpy(np.minimum(W0, 1.0)))
        mons = {l: Monitor(net.layers[l], ["s"], time=T) for l in ("Ae", "Ai")}
        for l, m in mons.items():
            net.add_monitor(m, l)
        net.train(learning)
        if tweak is not None:                        # e.g. other recurrent weights
            tweak(net)
        if additive:                                 # nodes.py:96-103: x = x * decay + trace_scale * s instead of x <- trace_scale on a spike
            for l in net.layers.values():
                l.traces_additive = True
                if l.traces:                         # (the Ai layer of DiehlAndCook2015 records no trace)
                    l.trace_scale.fill_(0.5)
        net.to(DEV)
        out = []
        section = net.pipelined() if pipelined else contextlib.nullcontext()
        section.__enter__()
        for r in range(n_inputs):
            net.sync()
            torch.manual_seed(11 + r)
            net.run({"X": torch.from_numpy(spikes[r]).view(T, B, *shape).to(DEV)}, time=T)
            net.sync()
            probe = torch.rand(3).numpy()
            out.append(dict(
                sE=mons["Ae"].get("s").cpu().numpy().copy(), sI=mons["Ai"].get("s").cpu().numpy().copy(),
                W=net.connections[("X", "Ae")].pipeline[0].value.cpu().numpy().copy(),
                theta=net.layers["Ae"].theta.cpu().numpy().copy(), vE=net.layers["Ae"].v.cpu().numpy().copy(),
                xE=net.layers["Ae"].x.cpu().numpy().copy(), xX=net.layers["X"].x.cpu().numpy().copy(),
                vI=net.layers["Ai"].v.cpu().numpy().copy(), probe=probe))
            plan = net.last_plan
            run.last_net = net
            if r % 2 == 0:
                net.reset_state_variables()
        section.__exit__(None, None, None)
        return out, plan
    finally:
        if net is not None and net.__dict__.get("_pipe") is not None:     # (an exception inside the section: close it)
            try:
                section.__exit__(RuntimeError, RuntimeError("aborted"), None)
            except Exception:                                             # noqa: BLE001
                pass
        _lib.lib().snn_set_plan_mode(0)


def oracle_run(N, B, T, spikes, w_scale=0.3, n_inputs=2, learning=True, Nin=784, inh=120.0, nu=(1e-4, 1e-2), exc=22.5, w_ei=None):
    """The same sequence as run() through the CPU oracle (oracle/snn_oracle.c: the reference's algorithm restated, pinned to the reference
    fixtures by tests/test_oracle_golden.py / test_oracle_fullsize.py): the THIRD party of the HIP-vs-HIP comparisons in this file, the fuzz
    tests and tests/test_gpu_async_form.py.  Same outputs as run() (replacing traces only: the oracle's D&C run has no additive switch)."""
    import oracle
    P = oracle.eth_mnist_dc_params(N, B, T, Nin=Nin, learning=learning)
    P.nu0, P.nu1 = float(nu[0]), float(nu[1])
    W0 = np.minimum(synth.uniform_f32(3, (Nin, N), 0.0, w_scale), 1.0).astype(np.float32)
    st = oracle.eth_mnist_dc_state(N, B, W0, Nin=Nin, exc=exc, inh=inh)
    if w_ei is not None:
        st["W_ei"] = np.ascontiguousarray(w_ei, np.float32)
    out = []
    for r in range(n_inputs):
        Q = oracle.exp_noise(11 + r, B * N * T + 16)
        cur = np.zeros(1, np.int64)
        sE, sI = oracle.run_dc2015(P, st, np.ascontiguousarray(spikes[r].reshape(T, B, Nin)), Q, cur)
        torch.manual_seed(11 + r)
        if cur[0]:
            torch.empty(int(cur[0])).exponential_(1)
        probe = torch.rand(3).numpy()
        out.append(dict(sE=sE.astype(bool), sI=sI.astype(bool), W=st["W_xe"].copy(), theta=st["theta"].copy(), vE=st["vE"].copy(), xE=st["xE"].copy(),
                        xX=st["xX"].copy(), vI=st["vI"].copy(), probe=probe))
        if r % 2 == 0:
            import cases
            cases.dc_reset(st)
    return out


def same_as_oracle(a, orc):
    for r, (x, y) in enumerate(zip(a, orc)):
        for k in y:
            np.testing.assert_array_equal(np.ascontiguousarray(x[k]).reshape(-1).view(np.uint8), np.ascontiguousarray(y[k]).reshape(-1).view(np.uint8),
                                          err_msg=f"input {r}: {k} (device vs CPU oracle)")


def same(a, b):
    for r, (x, y) in enumerate(zip(a, b)):
        for k in x:
            np.testing.assert_array_equal(x[k].view(np.uint8), y[k].view(np.uint8), err_msg=f"input {r}: {k}")


CASES = {
    # name: (N, B, T, density, w_scale, value_max)
    "dense_inputs_busy_path": (400, 32, 12, 0.30, 0.02, 1),
    "multivalued_bytes": (100, 8, 25, 0.02, 0.3, 3),
    "all_cross_many_candidates": (400, 32, 8, 0.5, 1.0, 1),
    "n100_b32_many_rows": (100, 32, 20, 0.15, 0.6, 1),
    "odd_sizes_n37_b5": (37, 5, 40, 0.03, 0.5, 1),
    "n12_b1_tiny": (12, 1, 30, 0.05, 1.0, 1),
    "n1000_b16": (1000, 16, 10, 0.02, 0.3, 1),
}


@pytest.mark.parametrize("name", list(CASES))
def test_fused_equals_generic_under_stress(name):
    N, B, T, dens, wsc, vmax = CASES[name]
    rs = np.random.RandomState(5)
    spikes = []
    for r in range(2):
        s = synth.dense_spikes(70 + r, (T, B, 784), dens)
        if vmax > 1:
            s = (s * rs.randint(1, vmax + 1, size=s.shape)).astype(np.uint8)
        spikes.append(s)
    fused, plan = run(0, N, B, T, spikes, w_scale=wsc)
    assert plan.startswith("dc2015-resident")           # lean form where it applies, else / after a give-up the general one
    general, plan_r = run(3, N, B, T, spikes, w_scale=wsc)
    assert plan_r == "dc2015-resident"
    stepped, plan_s = run(2, N, B, T, spikes, w_scale=wsc)
    assert plan_s == "dc2015-fused"
    generic, plan_g = run(1, N, B, T, spikes, w_scale=wsc)
    assert plan_g == "generic"
    same(fused, generic)
    same(general, generic)
    same(stepped, generic)
    if vmax == 1 and N * B <= 4000:                      # third party: the CPU oracle (0/1 spike bytes; sizes it finishes in seconds)
        same_as_oracle(fused, oracle_run(N, B, T, spikes, w_scale=wsc))
    assert sum(int(x["sE"].sum()) for x in fused) > 0, "no excitatory spike at all: vacuous"
    assert all(x["sE"].reshape(T, B, N).sum(axis=2).max() <= 1 for x in fused)


def test_learning_off_and_weak_inhibition():
    spikes = [synth.dense_spikes(80 + r, (30, 6, 784), 0.03) for r in range(2)]
    f, _ = run(0, 100, 6, 30, spikes, learning=False, inh=17.5)
    r3, _ = run(3, 100, 6, 30, spikes, learning=False, inh=17.5)
    h, _ = run(2, 100, 6, 30, spikes, learning=False, inh=17.5)
    g, _ = run(1, 100, 6, 30, spikes, learning=False, inh=17.5)
    same(f, g)
    same(r3, g)
    same(h, g)
    same_as_oracle(f, oracle_run(100, 6, 30, spikes, learning=False, inh=17.5))


def test_plan_refuses_unsupported_shapes_and_falls_back():
    # Nin not a multiple of 16 and batch > 32 are outside the fused plan: the generic plan must take over
    spikes = [synth.dense_spikes(90, (6, 3, 100), 0.1)] * 2
    out, plan = run(0, 20, 3, 6, spikes, Nin=100, shape=(100,))
    assert plan == "generic"
    spikes = [synth.dense_spikes(91, (5, 40, 784), 0.02)] * 2
    out, plan = run(0, 64, 40, 5, spikes)
    assert plan == "generic"


@pytest.mark.parametrize("N,B", [(100, 8), (400, 32)])
def test_additive_traces_every_plan_equals_generic(N, B):
    """`traces_additive=True` on all three layers (nodes.py:96-103): the input-trace pre-pass, the Ae / Ai trace stages and
    PostPre of every D&C plan against the generic per-operator plan, bit for bit."""
    T = 25
    spikes = [synth.dense_spikes(700 + r, (T, B, 784), 0.03) for r in range(2)]
    gen, plan_g = run(1, N, B, T, spikes, w_scale=0.5, additive=True)
    assert plan_g == "generic" and sum(int(r["sE"].sum()) for r in gen) > 0
    plain, _ = run(1, N, B, T, spikes, w_scale=0.5)
    assert not np.array_equal(gen[0]["xX"], plain[0]["xX"])               # (the switch does change the traces)
    for mode, want in ((0, "dc2015-resident"), (3, "dc2015-resident"), (2, "dc2015-fused")):
        res, plan = run(mode, N, B, T, spikes, w_scale=0.5, additive=True)
        assert plan.startswith(want)
        same(res, gen)


@pytest.mark.parametrize("nu", [(0.0, 1e-2), (1e-3, 0.0)])
def test_one_sided_learning_rates(nu):
    """PostPre with only the post-synaptic (nu[0] = 0) or only the pre-synaptic term (MCC_learning.py:255, :279): the
    row-per-thread and per-(row, column) forms of the resident plans skip the other term entirely."""
    N, B, T = 100, 16, 25
    spikes = [synth.dense_spikes(720 + r, (T, B, 784), 0.03) for r in range(2)]
    gen, plan_g = run(1, N, B, T, spikes, w_scale=0.5, nu=nu)
    assert plan_g == "generic" and sum(int(r["sE"].sum()) for r in gen) > 0
    for mode, want in ((0, "dc2015-resident"), (3, "dc2015-resident"), (2, "dc2015-fused")):
        res, plan = run(mode, N, B, T, spikes, w_scale=0.5, nu=nu)
        assert plan.startswith(want)
        same(res, gen)
