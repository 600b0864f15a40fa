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


@pytest.mark.parametrize("B,Nin,N", [(1, 784, 100), (3, 784, 100), (2, 1000, 37), (4, 784, 400), (16, 784, 100)])
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
        if probe.tail_isolated(B, N, t):
            np.testing.assert_array_equal(bits(got), bits(casc), err_msg=f"{t} threads: the isolated tail takes the cascade order")
            seen_other = seen_other or bool((bits(got) != bits(want)).any())
        else:
            np.testing.assert_array_equal(bits(got), bits(want), err_msg=f"{t} threads: serial order expected")
    if N in (100, 37) and B < 16:
        assert seen_other, "the thread dependence this test documents did not show up (torch changed?)"


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
