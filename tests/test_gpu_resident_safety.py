"""The resident D&C plan hands spikes between workgroups inside one launch, which is only correct when the whole
grid is running at once.  These tests cover what happens when that cannot be guaranteed:

 * the grid is sized against the device (CU count x occupancy) and launched cooperatively; a device that is too
   small takes the one-launch-per-timestep plan (SNN_DC_FAKE_CUS pretends to be one);
 * a workgroup that never shows up (SNN_DC_TEST_STALL) makes the others give up after a bounded wait; the kernel
   then returns WITHOUT having touched any state tensor, Network.run repeats the input on the per-step plan, and
   the results are still the reference's, bit for bit;
 * several processes sharing the GPU each get the reference's results.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import cases
import synth
from cases import gold, u8, unpack

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_cfg2_inputs(n_inputs=2):
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    g = gold("full_cfg2_dc_n400_b32")
    N, B, T = 400, 32, 250
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05,
                           inpt_shape=(1, 28, 28))
    mon = Monitor(net.layers["Ae"], ["s"], time=T)
    net.add_monitor(mon, "Ae_s")
    net.to(DEV)
    torch.manual_seed(2)
    plans = []
    for r in range(n_inputs):
        spikes = synth.spike_train(1000 + r, T, B, 784)
        net.run({"X": torch.from_numpy(spikes).view(T, B, 1, 28, 28).to(DEV)}, time=T)
        plans.append(net.last_plan)
        sE = mon.get("s").cpu().numpy().reshape(T, B, N).astype(u8)
        np.testing.assert_array_equal(sE, unpack(g[f"r{r}_sE"], (T, B, N)), err_msg=f"input {r}: Ae raster")
        W = net.connections[("X", "Ae")].pipeline[0].value.detach().cpu().numpy()
        assert cases.sha(W) == str(g[f"r{r}_W_sha"]), f"input {r}: weights"
        np.testing.assert_array_equal(net.layers["Ae"].theta.cpu().numpy().view(np.uint32), g[f"r{r}_theta"].view(np.uint32))
        cases.check_packed(g, f"r{r}_xX", net.layers["X"].x.cpu().numpy().reshape(B, 784))
        cases.check_packed(g, f"r{r}_vE", net.layers["Ae"].v.cpu().numpy())
        net.reset_state_variables()
    return net, plans


def test_device_too_small_for_the_grid_takes_the_per_step_plan(monkeypatch):
    monkeypatch.setenv("SNN_DC_FAKE_CUS", "16")          # 400 columns / 8 per workgroup = 50 workgroups > 16 "CUs"
    _, plans = run_cfg2_inputs(1)
    assert plans == ["dc2015-fused"]
    monkeypatch.setenv("SNN_DC_FAKE_CUS", "64")          # 4-column tiles (100 workgroups) do not fit, 8-column tiles (50) do
    _, plans = run_cfg2_inputs(1)
    assert plans == ["dc2015-resident"]                 # (the lean form exists for 4-column tiles only)


def test_missing_workgroup_times_out_state_untouched_and_run_is_repeated(monkeypatch):
    monkeypatch.setenv("SNN_DC_TEST_STALL", "7")
    net, plans = run_cfg2_inputs(2)
    assert plans == ["dc2015-fused", "dc2015-fused"], "the repeat runs on the one-launch-per-timestep plan"
    assert net.resident_retries == 2
    monkeypatch.delenv("SNN_DC_TEST_STALL")
    net, plans = run_cfg2_inputs(1)
    assert plans == ["dc2015-resident-lean"] and getattr(net, "resident_retries", 0) == 0


def test_lean_arbitration_exact_branch(monkeypatch):
    """The lean kernel picks the one_spike winner by comparing the raw 53-bit draws and evaluates the logarithms only
    when two draws are within 2^-19 of each other (practically never).  SNN_DC_TEST_ZONE=0 widens that margin to a factor
    of two, so the exact branch runs for most arbitrations -- and must give the same (reference) results."""
    monkeypatch.setenv("SNN_DC_TEST_ZONE", "0")
    net, plans = run_cfg2_inputs(2)
    assert plans == ["dc2015-resident-lean"] * 2 and getattr(net, "lean_retries", 0) == 0


def test_ordinary_launch_switch_still_matches(monkeypatch):
    """SNN_DC_COOP is read once per process; the non-cooperative launch is exercised in a child process."""
    code = ("import sys; sys.path[:0] = [%r, %r]; import test_gpu_resident_safety as t; _, p = t.run_cfg2_inputs(2); "
            "assert p == ['dc2015-resident-lean'] * 2; print('ok')" % (ROOT, os.path.join(ROOT, "tests")))
    env = dict(os.environ, SNN_DC_COOP="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_three_processes_sharing_the_gpu_all_get_the_reference_results():
    """3 x 100 workgroups of 1024 threads and ~150 KB LDS each cannot all be resident on 256 CUs: launches queue
    behind each other (cooperative launches are never partially resident), or time out and are repeated."""
    code = ("import sys; sys.path[:0] = [%r, %r]; import test_gpu_resident_safety as t\n"
            "for k in range(4):\n    net, p = t.run_cfg2_inputs(3)\n"
            "print('ok', p, getattr(net, 'resident_retries', 0))" % (ROOT, os.path.join(ROOT, "tests")))
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(3)]
    for p in procs:
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0 and out.startswith("ok"), err[-2000:]
        print(out.strip())
