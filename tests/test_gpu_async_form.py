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


@pytest.mark.parametrize("wg", [100, 102])
def test_missing_arbiter_or_raster_writer_times_out_state_untouched(monkeypatch, wg):
    """Workgroup 100 of the cfg2 grid is the arbiter, 101..104 write the rasters: without the arbiter no winners ever arrive (every
    compute workgroup gives up after its bounded poll); without a raster writer the arbiter stops at the first ring slot it may not
    overwrite and says so in the winners' granules."""
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
