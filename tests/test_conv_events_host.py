"""bindsnet_amd/csrc/snn_conv_events.hpp on the HOST: the event-driven body of the Conv2d-PostPre partial sums (set bits of packed
spike rows) against the dense body (every output position), element by element as the kernel's threads call them, and -- at
batch 1, where the batch reduction is the identity -- the resulting update against the oracle.  The bodies are
__host__ __device__; tests/hostcheck/conv_events_host.hip is compiled by hipcc (no GPU needed) and run on the CPU.  The device
kernel k_conv_pp_partial_ev is the default since it ran on an MI355X (round 4); SNN_CONV_PP_EVENTS=0 selects the dense body."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
f32, u8 = np.float32, np.uint8


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc on this machine")
    out = str(tmp_path_factory.mktemp("hostcheck") / "libconvhost.so")
    src = os.path.join(ROOT, "tests", "hostcheck", "conv_events_host.hip")
    subprocess.run([HIPCC, "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "--offload-arch=gfx950", src, "-o", out],
                   check=True, capture_output=True, timeout=600)
    lib = C.CDLL(out)
    lib.hostcheck_conv_pp_partials.argtypes = [C.c_void_p] * 4 + [C.c_int] * 10 + [C.c_void_p]
    lib.hostcheck_conv_pp_partials.restype = C.c_int
    return lib


def partials(lib, s_src, x_src, s_tgt, x_tgt, K, stride, pad, events):
    B, Cin, H, Wd = s_src.shape
    Cout = s_tgt.shape[1]
    part = np.full(2 * B * Cout * Cin * K * K, np.nan, f32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)               # noqa: E731
    rc = lib.hostcheck_conv_pp_partials(p(s_src), p(x_src), p(s_tgt), p(x_tgt), B, Cin, H, Wd, Cout, K, K, stride, pad, int(events), p(part))
    assert rc >= 0
    return part, rc


CASES = [  # B, Cin, H, W, Cout, K, stride, pad, p_src, p_tgt
    (2, 1, 28, 28, 32, 5, 1, 0, 0.05, 0.01), (3, 3, 12, 12, 4, 3, 1, 1, 0.2, 0.1), (2, 2, 13, 13, 5, 3, 2, 0, 0.3, 0.2),
    (1, 1, 16, 16, 6, 4, 2, 1, 0.5, 0.5), (2, 4, 9, 9, 3, 3, 1, 2, 0.15, 0.3), (1, 1, 32, 32, 2, 5, 3, 2, 0.1, 0.1), (1, 2, 8, 8, 4, 8, 1, 0, 0.4, 0.6),
]


@pytest.mark.parametrize("case", CASES)
def test_event_body_equals_dense_body_bit_for_bit(host, case):
    B, Cin, H, Wd, Cout, K, stride, pad, ps, pt = case
    OH, OW = (H + 2 * pad - K) // stride + 1, (Wd + 2 * pad - K) // stride + 1
    k = sum(case[:8])
    s_src, x_src = synth.dense_spikes(7000 + k, (B, Cin, H, Wd), ps), synth.uniform_f32(7100 + k, (B, Cin, H, Wd), 0.0, 1.0)
    s_tgt, x_tgt = synth.dense_spikes(7200 + k, (B, Cout, OH, OW), pt), synth.uniform_f32(7300 + k, (B, Cout, OH, OW), 0.0, 1.0)
    dense, _ = partials(host, s_src, x_src, s_tgt, x_tgt, K, stride, pad, False)
    ev, multi = partials(host, s_src, x_src, s_tgt, x_tgt, K, stride, pad, True)
    assert multi == 0 and not np.isnan(dense).any() and np.abs(dense).max() > 0
    np.testing.assert_array_equal(ev.view(np.uint32), dense.view(np.uint32))
    # and the sums are the sums: against float64 im2col products
    E = Cout * Cin * K * K
    cols = np.lib.stride_tricks.sliding_window_view(np.pad(s_src.astype(np.float64), ((0, 0), (0, 0), (pad, pad), (pad, pad))), (K, K), axis=(2, 3))
    cols = cols[:, :, ::stride, ::stride]                                      # [B, Cin, OH, OW, K, K]
    a64 = np.einsum("bolm,bclmyx->bocyx", x_tgt.astype(np.float64), cols).reshape(B, E)
    np.testing.assert_allclose(dense[:B * E].reshape(B, E), a64, rtol=0, atol=1e-4)


def test_multi_valued_spike_bytes_take_the_dense_body(host):
    B, Cin, H, Wd, Cout, K = 1, 1, 10, 10, 2, 3
    s_src = synth.dense_spikes(7400, (B, Cin, H, Wd), 0.3)
    s_src[0, 0, 4, 5] = 2
    x_src, s_tgt, x_tgt = synth.uniform_f32(7401, s_src.shape, 0, 1), synth.dense_spikes(7402, (B, Cout, 8, 8), 0.2), synth.uniform_f32(7403, (B, Cout, 8, 8), 0, 1)
    dense, _ = partials(host, s_src, x_src, s_tgt, x_tgt, K, 1, 0, False)
    ev, multi = partials(host, s_src, x_src, s_tgt, x_tgt, K, 1, 0, True)
    assert multi == 1
    np.testing.assert_array_equal(ev.view(np.uint32), dense.view(np.uint32))


@pytest.mark.parametrize("case", CASES[:5])
def test_batch_one_update_from_the_event_partials_equals_the_oracle(host, case):
    _, Cin, H, Wd, Cout, K, stride, pad, ps, pt = case
    OH, OW = (H + 2 * pad - K) // stride + 1, (Wd + 2 * pad - K) // stride + 1
    k = sum(case[1:8])
    W = synth.uniform_f32(7500 + k, (Cout, Cin, K, K), 0.0, 0.5)
    s_src, x_src = synth.dense_spikes(7600 + k, (1, Cin, H, Wd), ps), synth.uniform_f32(7700 + k, (1, Cin, H, Wd), 0.0, 1.0)
    s_tgt, x_tgt = synth.dense_spikes(7800 + k, (1, Cout, OH, OW), pt), synth.uniform_f32(7900 + k, (1, Cout, OH, OW), 0.0, 1.0)
    part, _ = partials(host, s_src, x_src, s_tgt, x_tgt, K, stride, pad, True)
    E = W.size
    nu0, nu1 = f32(1e-3), f32(1e-2)
    Wn = W.reshape(-1).copy()
    Wn = (Wn - nu0 * part[:E]).astype(f32)                                      # k_conv_pp_apply at B = 1: w - nu0 * a, + nu1 * p, clamp
    Wn = (Wn + nu1 * part[E:]).astype(f32)
    Wn = np.clip(Wn, f32(0.0), f32(1.0))
    Wo = W.copy()
    oracle.conv2d_postpre(Wo, s_src, x_src, s_tgt, x_tgt, stride=stride, pad=pad, nu0=nu0, nu1=nu1, wmin=0.0, wmax=1.0)
    np.testing.assert_array_equal(Wn.view(np.uint32), Wo.reshape(-1).view(np.uint32))


@pytest.mark.parametrize("case", [c for c in CASES if c[6] == 1] + [(3, 2, 28, 28, 4, 5, 1, 2, 0.3, 0.05), (1, 1, 32, 32, 3, 7, 1, 3, 0.9, 0.9)])
def test_list_form_of_the_fused_plan_equals_dense_body_bit_for_bit(host, case):
    """The list form k_convpp_run walks (round 6: one ascending event list per image, every weight element keeps the events inside its window,
    eight at a time) against the dense body, stride 1: same partial sums, bit for bit."""
    B, Cin, H, Wd, Cout, K, stride, pad, ps, pt = case
    OH, OW = (H + 2 * pad - K) // stride + 1, (Wd + 2 * pad - K) // stride + 1
    k = sum(case[:8])
    s_src, x_src = synth.dense_spikes(8000 + k, (B, Cin, H, Wd), ps), synth.uniform_f32(8100 + k, (B, Cin, H, Wd), 0.0, 1.0)
    s_tgt, x_tgt = synth.dense_spikes(8200 + k, (B, Cout, OH, OW), pt), synth.uniform_f32(8300 + k, (B, Cout, OH, OW), 0.0, 1.0)
    dense, _ = partials(host, s_src, x_src, s_tgt, x_tgt, K, stride, pad, 0)
    lists, multi = partials(host, s_src, x_src, s_tgt, x_tgt, K, stride, pad, 2)
    assert multi == 0 and np.abs(dense).max() > 0
    np.testing.assert_array_equal(lists.view(np.uint32), dense.view(np.uint32))
