"""Third-generation lean form of the resident D&C plan (csrc/snn_dc2015_async.hip: compute workgroups that do not wait for each other,
one arbiter workgroup, raster writers): what is specific to it.  The five full-size reference fixtures, the fuzz / stress / safety
tests and the literal eth_mnist.py run on it because it is the default; here:

 * it IS what runs (snn_dc2015_last_form() == 3), and the second generation (SNN_DC_ASYNC=0 -- still what a graph too wide for the
   third generation's grid gets) still reproduces the reference;
 * the conditions it gives up on, each ending in SNN_ERR_RETRY with nothing written and a repeat on the general resident form that
   equals the generic plan bit for bit: an Ai neuron that does not follow its Ae partner (weak Ae -> Ai weights), Ae -> Ai weights off
   the diagonal;
 * very short runs (the rings of granules and the one-step-early preparation have their corner cases at T = 1, 2, 3);
 * an arbiter / raster-writer workgroup that never shows up: bounded wait, state untouched, repeat on the per-step plan."""
import numpy as np
import pytest
import torch

import synth
import test_gpu_fused_stress as dc
import test_gpu_resident_safety as safety
from bindsnet_amd import _lib

pytestmark = pytest.mark.gpu
form = lambda: _lib.lib().snn_dc2015_last_form()       # noqa: E731


def test_third_generation_is_the_default_and_the_second_still_matches(monkeypatch):
    _, plans = safety.run_cfg2_inputs(2)
    assert plans == ["dc2015-resident-lean"] * 2 and form() == 3
    monkeypatch.setenv("SNN_DC_ASYNC", "0")
    _, plans = safety.run_cfg2_inputs(2)
    assert plans == ["dc2015-resident-lean"] * 2 and form() == 2


@pytest.mark.parametrize("T", [1, 2, 3, 5])
def test_very_short_runs(T):
    N, B = 400, 32
    spikes = [synth.dense_spikes(4100 + 3 * T + r, (T, B, 784), 0.05) for r in range(2)]      # ~39 events per sample (the lean forms take up to 63), strong weights: spikes from step 1 on
    res, plan = dc.run(0, N, B, T, spikes, w_scale=1.0)
    assert plan == "dc2015-resident-lean" and form() == 3
    gen, _ = dc.run(1, N, B, T, spikes, w_scale=1.0)
    assert T == 1 or sum(int(r["sE"].sum()) for r in gen) > 0          # (the first current arrives at step 1)
    dc.same(res, gen)
    dc.same_as_oracle(res, dc.oracle_run(N, B, T, spikes, w_scale=1.0))   # third party: the CPU oracle


def test_weak_excitation_gives_up_and_is_repeated_on_the_general_form():
    """exc = 10: one Ae spike does not take Ai_j over its threshold, so 'the Ai spikes of step t are the Ae winners of step t-1' --
    what every workgroup's inhibition rests on in this form -- fails at the first winner: the owner says so in its granule, the
    arbiter passes it on, everybody leaves, nothing is written, and the input is repeated on the general resident form."""
    N, B, T = 100, 8, 40
    spikes = [synth.dense_spikes(4200 + r, (T, B, 784), 0.05) for r in range(2)]
    res, plan = dc.run(0, N, B, T, spikes, w_scale=0.8, exc=10.0)
    net = dc.run.last_net
    assert plan == "dc2015-resident" and net.lean_retries >= 1
    gen, _ = dc.run(1, N, B, T, spikes, w_scale=0.8, exc=10.0)
    assert sum(int(r["sE"].sum()) for r in gen) > 0 and sum(int(r["sI"].sum()) for r in gen) < sum(int(r["sE"].sum()) for r in gen)
    dc.same(res, gen)
    dc.same_as_oracle(res, dc.oracle_run(N, B, T, spikes, w_scale=0.8, exc=10.0))


def test_give_up_in_the_LAST_iteration_writes_nothing():
    """The advisor's round-4 finding: a reason to give up that shows in iteration T-1 comes AFTER the workgroup's last publish; without the
    final report ("step T" crossing granule) the arbiter committed, every OTHER workgroup wrote its state back, and the host repeated the input
    on a half-advanced network.  exc = 10 (Ai never follows Ae) and T = (step of the first Ae winner) + 2: the first Ai mismatch is at step T-1."""
    N, B, T0 = 100, 8, 40
    long = [synth.dense_spikes(4200 + r, (T0, B, 784), 0.05) for r in range(2)]
    probe = dc.oracle_run(N, B, T0, long, w_scale=0.8, exc=10.0, n_inputs=1)
    first = int(np.nonzero(probe[0]["sE"].reshape(T0, -1).any(axis=1))[0][0])
    for T in (first + 2, first + 3):                                     # (the second: one ordinary iteration behind the mismatch)
        spikes = [x[:T].copy() for x in long]
        res, plan = dc.run(0, N, B, T, spikes, w_scale=0.8, exc=10.0)
        assert plan == "dc2015-resident" and dc.run.last_net.lean_retries >= 1
        orc = dc.oracle_run(N, B, T, spikes, w_scale=0.8, exc=10.0)
        assert int(orc[0]["sE"].sum()) > 0 and int(orc[0]["sI"].sum()) == 0
        dc.same_as_oracle(res, orc)
        gen, _ = dc.run(1, N, B, T, spikes, w_scale=0.8, exc=10.0)
        dc.same(res, gen)


def test_entry_spike_and_one_timestep():
    """T = 1 with an Ae spike in the entry state and weak Ae -> Ai weights: the mismatch is in the launch's only iteration."""
    N, B, T0 = 100, 8, 40
    long = [synth.dense_spikes(4200 + r, (T0, B, 784), 0.05) for r in range(2)]
    probe = dc.oracle_run(N, B, T0, long, w_scale=0.8, exc=10.0, n_inputs=1)
    first = int(np.nonzero(probe[0]["sE"].reshape(T0, -1).any(axis=1))[0][0])
    # input [0] ends at the first winner's step (the winner stays in Ae.s: no reset), input [1] is ONE step from that entry state
    spikes = [long[0][:first + 1].copy(), long[1][:1].copy()]

    def go(mode):
        from bindsnet_amd.models import DiehlAndCook2015
        from bindsnet_amd.network.monitors import Monitor
        _lib.lib().snn_set_plan_mode(mode)
        try:
            torch.manual_seed(0)
            net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=10.0, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
            net.connections[("X", "Ae")].pipeline[0].value.data.copy_(torch.from_numpy(np.minimum(synth.uniform_f32(3, (784, N), 0.0, 0.8), 1.0)))
            net.to("cuda")
            torch.manual_seed(5)
            T = first + 1
            net.run({"X": torch.from_numpy(spikes[0]).view(T, B, 1, 28, 28).to("cuda")}, time=T)
            assert bool(net.layers["Ae"].s.any()), "the entry state must hold an Ae spike"
            mons = {l: Monitor(net.layers[l], ["s"], time=1) for l in ("Ae", "Ai")}
            for l, m in mons.items():
                net.add_monitor(m, l)
            net.run({"X": torch.from_numpy(spikes[1]).view(1, B, 1, 28, 28).to("cuda")}, time=1)
            return dict(W=net.connections[("X", "Ae")].pipeline[0].value.cpu().numpy().copy(), theta=net.layers["Ae"].theta.cpu().numpy().copy(),
                        vE=net.layers["Ae"].v.cpu().numpy().copy(), vI=net.layers["Ai"].v.cpu().numpy().copy(), sE=net.layers["Ae"].s.cpu().numpy().copy(),
                        sI=net.layers["Ai"].s.cpu().numpy().copy(), rasE=mons["Ae"].get("s").cpu().numpy().copy(), probe=torch.rand(3).numpy()), net
        finally:
            _lib.lib().snn_set_plan_mode(0)
    res, net = go(0)
    assert net.lean_retries >= 1
    gen, _ = go(1)
    dc.same([res], [gen])


def test_off_diagonal_excitatory_weights_give_up_at_once():
    N, B, T = 64, 4, 30

    def dense_ei(net):
        w = net.connections[("Ae", "Ai")].pipeline[0].value
        w.data.copy_(torch.from_numpy(synth.uniform_f32(4300, tuple(w.shape), 0.0, 6.0)))
    spikes = [synth.dense_spikes(4310 + r, (T, B, 784), 0.06) for r in range(2)]
    res, plan = dc.run(0, N, B, T, spikes, w_scale=0.8, tweak=dense_ei)
    assert plan == "dc2015-resident" and dc.run.last_net.lean_retries >= 1
    gen, _ = dc.run(1, N, B, T, spikes, w_scale=0.8, tweak=dense_ei)
    assert sum(int(r["sE"].sum()) for r in gen) > 0
    dc.same(res, gen)
    dc.same_as_oracle(res, dc.oracle_run(N, B, T, spikes, w_scale=0.8, w_ei=synth.uniform_f32(4300, (N, N), 0.0, 6.0)))


@pytest.mark.parametrize("wg", [100, 102, 105, 140])
def test_missing_arbiter_or_raster_writer_times_out_state_untouched(monkeypatch, wg):
    """Workgroup 100 of the cfg2 grid is the arbiter, 101..104 write the rasters: without the arbiter no winners ever arrive (every
    compute workgroup gives up after its bounded poll); without a raster writer the arbiter stops at the first ring slot it may not
    overwrite and says so in the winners' granules.  105.. are the producer workgroups (digest entries and X traces inside the launch):
    without producer 0 the digest of step 0 never comes, without producer 35 that of entry 35 (and one chunk of the X traces)."""
    monkeypatch.setenv("SNN_DC_TEST_STALL", str(wg))
    net, plans = safety.run_cfg2_inputs(1)
    assert plans == ["dc2015-fused"] and net.resident_retries == 1
    monkeypatch.delenv("SNN_DC_TEST_STALL")
    net, plans = safety.run_cfg2_inputs(1)
    assert plans == ["dc2015-resident-lean"] and getattr(net, "resident_retries", 0) == 0 and form() == 3


def test_no_spike_monitors_means_no_raster_writers():
    """Without a spike monitor on Ae / Ai the launch has no raster-writer workgroups (and the arbiter nothing to wait for): state,
    weights, theta and the generator position still equal the generic plan's."""
    from bindsnet_amd.models import DiehlAndCook2015
    N, B, T = 100, 16, 60
    spikes = [synth.dense_spikes(4400 + r, (T, B, 784), 0.04) for r in range(2)]

    def go(mode):
        _lib.lib().snn_set_plan_mode(mode)
        try:
            torch.manual_seed(0)
            net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
            net.connections[("X", "Ae")].pipeline[0].value.data.copy_(torch.from_numpy(np.minimum(synth.uniform_f32(3, (784, N), 0.0, 0.6), 1.0)))
            net.to("cuda")
            out = []
            for r in range(2):
                torch.manual_seed(11 + r)
                net.run({"X": torch.from_numpy(spikes[r]).view(T, B, 1, 28, 28).to("cuda")}, time=T)
                out.append(dict(W=net.connections[("X", "Ae")].pipeline[0].value.cpu().numpy().copy(), theta=net.layers["Ae"].theta.cpu().numpy().copy(),
                                vE=net.layers["Ae"].v.cpu().numpy().copy(), vI=net.layers["Ai"].v.cpu().numpy().copy(),
                                sE=net.layers["Ae"].s.cpu().numpy().copy(), probe=torch.rand(3).numpy()))
            return out, net.last_plan
        finally:
            _lib.lib().snn_set_plan_mode(0)
    res, plan = go(0)
    assert plan == "dc2015-resident-lean" and form() == 3
    gen, _ = go(1)
    assert float(np.abs(gen[1]["theta"]).sum()) > 0, "the run must have spiked"
    dc.same(res, gen)
    orc = dc.oracle_run(N, B, T, spikes, w_scale=0.6)                    # (same weights: uniform_f32(3, ..., 0.6); dc.run's reset-after-input-0 schedule differs: compare input 0 only)
    for k in ("W", "theta", "vE", "vI", "probe"):
        np.testing.assert_array_equal(np.ascontiguousarray(res[0][k]).reshape(-1).view(np.uint8), np.ascontiguousarray(orc[0][k]).reshape(-1).view(np.uint8), err_msg=k)
