"""What "the reference's summation order" means when the reference's own float sums depend on torch's thread count.

ATen's sum(dim=1) of [B, Nin, N] (MulticompartmentConnection.compute) and sum(dim=0) of [Nin, N] (Weight.normalize) take another
order in the last N mod 32 < 8 columns when the work is split over the columns and those columns end up alone in a thread's
range (tools/probe_aten_sum_threads.py: e.g. N = 100, B = 1 at 9 or >= 12 threads).  The package pins the SERIAL order -- the
oracle, the MI355X kernels, the host path.  This file pins (1) the model of when torch leaves that order, against torch itself,
(2) that the oracle is the serial order, (3) that the host path stays on it whatever the caller's thread setting."""
import importlib.util
import os

import numpy as np
import pytest
import torch

import oracle
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("probe_aten_sum_threads", os.path.join(ROOT, "tools", "probe_aten_sum_threads.py"))
probe = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(probe)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture
def threads():
    n0 = torch.get_num_threads()
    yield
    torch.set_num_threads(n0)


@pytest.mark.parametrize("B,Nin,N", [(1, 784, 100), (3, 784, 100), (2, 1000, 37), (4, 784, 400), (16, 784, 100), (1, 100, 100), (4, 100, 100),
                                     (1, 327, 100), (1, 328, 100), (36, 784, 36), (30, 784, 36)])
def test_torch_leaves_the_serial_order_exactly_where_the_model_says(B, Nin, N, threads):
    W = synth.uniform_f32(5100 + N, (Nin, N), -1.0, 1.0)
    s = synth.dense_spikes(5200 + B, (B, Nin), 0.3)
    want = oracle.prop_mcc(W, s)                                   # the oracle: serial order
    # the same columns inside a full group of 32 (weights padded with zero columns): the cascade order
    pad = (-N) % 32
    casc = oracle.prop_mcc(np.ascontiguousarray(np.concatenate([W, np.zeros((Nin, pad), np.float32)], 1)), s)[:, :N].copy() if pad else want
    tail = N % 32                                                 # scalar_outer_sum: groups of FOUR columns cascade, the rest row_sum
    if tail % 4:
        casc[:, N - tail % 4:] = want[:, N - tail % 4:]
    x = torch.from_numpy(s).view(B, Nin, 1).repeat(1, 1, N) * torch.from_numpy(W)
    seen_other = False
    for t in (1, 2, 8, 9, 16, 17, 40):
        torch.set_num_threads(t)
        got = x.sum(1).numpy()
        if probe.tail_isolated(B, N, t, Nin):
            np.testing.assert_array_equal(bits(got), bits(casc), err_msg=f"{t} threads: the isolated tail takes the cascade order")
            seen_other = seen_other or bool((bits(got) != bits(want)).any())
        else:
            np.testing.assert_array_equal(bits(got), bits(want), err_msg=f"{t} threads: serial order expected")
    if (B, Nin, N) in ((1, 784, 100), (3, 784, 100), (2, 1000, 37), (4, 100, 100), (1, 328, 100), (30, 784, 36)):
        assert seen_other, "the thread dependence this test documents did not show up (torch changed?)"
    if (B, Nin, N) in ((1, 100, 100), (1, 327, 100), (36, 784, 36), (4, 784, 400)):
        assert not seen_other


def test_host_path_stays_on_the_serial_order_at_any_thread_count(threads):
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import MulticompartmentConnection
    from bindsnet_amd.network.topology_features import Weight
    B, Nin, N = 1, 784, 100
    W = synth.uniform_f32(5300, (Nin, N), -1.0, 1.0)
    s = synth.dense_spikes(5301, (B, Nin), 0.3)
    feat = Weight("weight", torch.from_numpy(W).clone(), norm=78.4)
    conn = MulticompartmentConnection(Input(n=Nin), LIFNodes(n=N), device="cpu", pipeline=[feat])
    want = oracle.prop_mcc(W, s)
    Wn = W.copy()
    oracle.normalize(Wn, np.float32(78.4), use_abs=False)
    for t in (1, 8, 16):
        torch.set_num_threads(t)
        np.testing.assert_array_equal(bits(conn.compute(torch.from_numpy(s)).numpy()), bits(want), err_msg=f"compute at {t} threads")
        assert torch.get_num_threads() == t
    torch.set_num_threads(16)
    conn.normalize()
    np.testing.assert_array_equal(bits(feat.value.detach().numpy()), bits(Wn), err_msg="normalize at 16 threads")


def test_oracle_follows_torch_at_every_thread_count(threads):
    """oracle.reference_threads(t): propagation and normalisation as the reference computes them WITH t threads == torch at t
    threads, bit for bit (t = 1: the serial order, the default everything else is pinned to)."""
    for (B, Nin, N) in [(1, 784, 100), (3, 784, 100), (2, 1000, 37), (32, 784, 100), (4, 100, 100), (30, 784, 36), (4, 784, 400)]:
        W = synth.uniform_f32(5400 + N, (Nin, N), -1.0, 1.0)
        s = synth.dense_spikes(5500 + B, (B, Nin), 0.3)
        x = torch.from_numpy(s).view(B, Nin, 1).repeat(1, 1, N) * torch.from_numpy(W)
        Wn = synth.uniform_f32(5600 + N, (Nin, N), -0.2, 1.0)
        for t in (1, 8, 9, 16, 40, 255):
            torch.set_num_threads(t)
            with oracle.reference_threads(t):
                np.testing.assert_array_equal(bits(x.sum(1).numpy()), bits(oracle.prop_mcc(W, s)), err_msg=f"prop {(B, Nin, N)} at {t} threads")
                a = Wn.copy()
                oracle.normalize(a, np.float32(78.4), False)
            tw = torch.from_numpy(Wn.copy())
            cs = tw.sum(0).unsqueeze(0)
            cs[cs == 0] = 1.0
            tw *= 78.4 / cs
            np.testing.assert_array_equal(bits(tw.numpy()), bits(a), err_msg=f"normalize {(Nin, N)} at {t} threads")


def test_oracle_run_matches_the_reference_run_with_16_threads():
    """A whole cfg1 run (N = 100, batch 1, 3 x 250 timesteps of the stated input) by the UNMODIFIED reference with 16 torch threads
    (tests/golden/make_golden_full.py cfg1_poisson_t16): the oracle in reference_threads(16) mode reproduces it bit for bit --
    rasters, weights, theta, state --, and in its default (serial) mode it reproduces the 8-thread fixture instead: the two
    reference runs agree in every spike and differ in the last bits of the weights."""
    import cases
    from cases import check_packed, gold, unpack
    from test_oracle_fullsize import dc_params, ref_init_weights
    g16, g8 = gold("full_cfg1_dc_n100_b1_poisson_t16"), gold("full_cfg1_dc_n100_b1_poisson")
    assert int(g16["threads"]) == 16
    N, B, T, runs = int(g16["N"]), int(g16["B"]), int(g16["T"]), int(g16["runs"])
    assert [str(g16[f"r{r}_W_sha"]) for r in range(runs)] != [str(g8[f"r{r}_W_sha"]) for r in range(runs)], "the two reference runs should differ"
    for g, t in ((g16, 16), (g8, 1)):
        P = dc_params(g)
        st = cases.dc_state(N, B, inh=120.0)
        st["W_xe"] = ref_init_weights(784, N)
        Q = cases.exp_noise(2, sum(int(g[f"r{r}_consumed"]) for r in range(runs)) + B * N)
        cur = np.zeros(1, np.int64)
        with oracle.reference_threads(t):
            for r in range(runs):
                rasE, rasI = oracle.run_dc2015(P, st, cases.fixture_input(g, r, T, B), Q, cur)
                np.testing.assert_array_equal(rasE, unpack(g[f"r{r}_sE"], (T, B, N)), err_msg=f"{t} threads, run {r}: Ae raster")
                np.testing.assert_array_equal(rasI, unpack(g[f"r{r}_sI"], (T, B, N)), err_msg=f"{t} threads, run {r}: Ai raster")
                assert cases.sha(st["W_xe"]) == str(g[f"r{r}_W_sha"]), f"{t} threads, run {r}: weights"
                np.testing.assert_array_equal(bits(st["theta"]), bits(g[f"r{r}_theta"]))
                for key, a in (("vE", st["vE"]), ("xE", st["xE"]), ("vI", st["vI"])):
                    check_packed(g, f"r{r}_{key}", a)
                cases.dc_reset(st)
    for r in range(runs):
        np.testing.assert_array_equal(g16[f"r{r}_sE"], g8[f"r{r}_sE"])


def test_selftest_host_part(threads):
    """`python -m bindsnet_amd.selftest` ("ship a self-test that checks the summation rules against torch"):
    its host checks pass on this torch; the device checks run where there is an MI355X (tests/test_gpu_zz_experimental.py)."""
    from bindsnet_amd import selftest
    msgs = []
    assert selftest.host_checks(msgs.append), msgs
