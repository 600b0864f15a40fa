"""The kernels' ordered-sum accumulators (csrc/snn_order.hpp: Cascade, RowSum4, OuterSum, CascadeFlat, CascadeN, SeqSum) and the
device generator's arithmetic (csrc/snn_rng.hpp: mt19937 tempering / twist, the glibc log1p port, the Exp(1) draw) ON THE HOST:
they are __host__ __device__, tests/hostcheck/order_rng_host.hip drives them the way the kernels' threads do (non-zero terms
only, ascending index), and the results are compared bit for bit with the oracle (serial ATen order), with torch's own sums and
with torch's CPU generator.  The device code is unchanged by the attribute (the gfx950 ISA of every kernel file is identical
before and after, apart from the compilation-unit id)."""
import ctypes as C
import math
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

import oracle
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
f32, u8 = np.float32, np.uint8


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc on this machine")
    out = str(tmp_path_factory.mktemp("hostcheck") / "liborderhost.so")
    src = os.path.join(ROOT, "tests", "hostcheck", "order_rng_host.hip")
    subprocess.run([HIPCC, "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "--offload-arch=gfx950", src, "-o", out],
                   check=True, capture_output=True, timeout=600)
    lib = C.CDLL(out)
    lib.hostcheck_log1p.argtypes, lib.hostcheck_log1p.restype = [C.c_double], C.c_double
    return lib


def p_(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("B,Nin,N,dens", [(3, 784, 100, 0.3), (2, 784, 400, 0.012), (2, 1000, 37, 0.4), (1, 5000, 70, 0.6), (2, 33, 15, 0.5),
                                          (1, 70000, 33, 0.05), (2, 4099, 64, 0.5), (2, 400, 400, 0.01), (1, 20, 33, 0.9)])
def test_outer_sum_accumulators_equal_the_oracle_and_torch(host, B, Nin, N, dens):
    W = synth.uniform_f32(6000 + N, (Nin, N), -1.0, 1.0)
    s = synth.dense_spikes(6100 + Nin, (B, Nin), dens)
    out = np.empty((B, N), f32)
    host.hostcheck_prop(p_(W), p_(s), B, Nin, N, 0, p_(out))
    np.testing.assert_array_equal(bits(out), bits(oracle.prop_mcc(W, s)), err_msg="OuterSum vs the oracle")
    n0 = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        t = (torch.from_numpy(s).view(B, Nin, 1).repeat(1, 1, N) * torch.from_numpy(W)).sum(1).numpy()
    finally:
        torch.set_num_threads(n0)
    np.testing.assert_array_equal(bits(out), bits(t), err_msg="OuterSum vs torch (serial)")
    # the other forms of the same sums: CascadeN == the cascade class everywhere; CascadeFlat likewise below 4096 terms;
    # RowSum4 == the tail class everywhere; SeqSum == plain ascending order
    full = (N // 32) * 32
    host.hostcheck_prop(p_(W), p_(s), B, Nin, N, 3, p_(out))
    np.testing.assert_array_equal(bits(out[:, :full]), bits(t[:, :full]), err_msg="CascadeN")
    if Nin < 4096:
        host.hostcheck_prop(p_(W), p_(s), B, Nin, N, 1, p_(out))
        np.testing.assert_array_equal(bits(out[:, :full]), bits(t[:, :full]), err_msg="CascadeFlat")
    host.hostcheck_prop(p_(W), p_(s), B, Nin, N, 4, p_(out))
    np.testing.assert_array_equal(bits(out[:, full:]), bits(t[:, full:]), err_msg="RowSum4")
    host.hostcheck_prop(p_(W), p_(s), B, Nin, N, 2, p_(out))
    np.testing.assert_array_equal(bits(out), bits(oracle.prop_dense(W, s)), err_msg="SeqSum vs the oracle's canonical dense order")


@pytest.mark.parametrize("B,E", [(1, 64), (16, 960), (17, 960), (32, 2400), (48, 784 * 4), (128, 640), (37, 63), (200, 96)])
def test_batch_reduction_accumulator_equals_torch_sum_over_the_batch(host, B, E):
    terms = synth.uniform_f32(6200 + B, (B, E), -1.0, 1.0) * synth.dense_spikes(6300 + E, (B, E), 0.4)
    out = np.empty(E, f32)
    host.hostcheck_batch_sum(p_(np.ascontiguousarray(terms)), B, E, p_(out))
    n0 = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        want = torch.sum(torch.from_numpy(terms), dim=0).numpy()
    finally:
        torch.set_num_threads(n0)
    np.testing.assert_array_equal(bits(out), bits(want))


def test_device_generator_arithmetic_equals_torch_cpu_generator(host):
    from bindsnet_amd.rng import torch_state_to_words
    for seed, warm, n in ((0, 0, 5000), (7, 311, 3000), (123, 624 * 2 + 5, 2000)):
        torch.manual_seed(seed)
        if warm:
            torch.rand(warm)
        img = torch_state_to_words(torch.get_rng_state()).view(np.uint32).copy()
        mt, pos = np.ascontiguousarray(img[:624]), C.c_int(int(img[624]))
        want = torch.empty(n).exponential_(1).numpy()
        out = np.empty(n, f32)
        host.hostcheck_exponential(p_(mt), C.byref(pos), C.c_long(n), p_(out))
        np.testing.assert_array_equal(bits(out), bits(want), err_msg=f"seed {seed}")
        after = torch_state_to_words(torch.get_rng_state()).view(np.uint32)
        np.testing.assert_array_equal(mt, after[:624])
        assert pos.value == int(after[624])


def test_log1p_port_equals_the_host_libm(host):
    rs = np.random.RandomState(9)
    xs = np.concatenate([-rs.uniform(0, 1, 200000), -np.ldexp(rs.uniform(0.5, 1, 50000), -rs.randint(1, 60, 50000)), [-0.0, -1e-300, -0.5, -0.2928932188134525,
                                                                                                                         -(1 - 2.0 ** -53)]])
    for x in xs:
        got, want = host.hostcheck_log1p(float(x)), math.log1p(float(x))
        assert got == want or (got != got and want != want), (x, got, want)


def test_neuron_updates_equal_the_reference_fixture_and_the_oracle(host):
    """csrc/snn_common.hpp (lif_update, dc_update, trace_next) on the host: the LIF sequences of the reference fixture op_nodes
    (refractory masking, lower bound, overwrite and additive traces) bit for bit; the D&C membrane / adaptive-threshold sequence
    (one_spike off) against the oracle."""
    from cases import gold, unpack
    from bindsnet_amd._lib import DcParams, LifParams
    g = gold("op_nodes")
    B, N, T = int(g["B"]), int(g["N"]), int(g["T"])
    I = synth.uniform_f32(900, (T, B, N), -2.0, 6.0)
    n = B * N

    def lif(p, cur, v0):
        v = np.full((B, N), v0, f32); r = np.zeros((B, N), f32); s = np.zeros((B, N), u8); x = np.zeros((B, N), f32)
        ras = np.zeros((T, B, N), u8)
        host.hostcheck_lif_sequence(p_(v), p_(r), p_(s), p_(x), p_(np.ascontiguousarray(cur)), C.c_long(n), T, C.byref(p), p_(ras))
        return v, r, x, ras

    p = LifParams()
    p.decay, p.rest, p.reset, p.thresh, p.refrac, p.dt = float(g["lif_decay"]), -60.0, -45.0, -40.0, 2.0, 1.0
    p.has_lbound, p.lbound, p.traces, p.trace_decay, p.trace_scale, p.traces_additive = 1, -62.0, 1, float(g["lif_trace_decay"]), 1.0, 0
    v, r, x, ras = lif(p, I, -60.0)
    np.testing.assert_array_equal(ras, unpack(g["lif_s"], (T, B, N)))
    for a, key in ((v, "lif_v"), (x, "lif_x"), (r, "lif_r")):
        np.testing.assert_array_equal(bits(a), bits(g[key]), err_msg=key)
    p = LifParams()
    p.decay, p.rest, p.reset, p.thresh, p.refrac, p.dt = float(g["lifadd_decay"]), -65.0, -65.0, -52.0, 5.0, 1.0
    p.traces, p.trace_decay, p.trace_scale, p.traces_additive = 1, float(g["lifadd_trace_decay"]), 0.5, 1
    v, r, x, ras = lif(p, I * f32(3), -65.0)
    np.testing.assert_array_equal(bits(v), bits(g["lifadd_v"]))
    np.testing.assert_array_equal(bits(x), bits(g["lifadd_x"]))
    # D&C membrane + theta, one_spike off, against the oracle
    d = DcParams()
    d.lif.decay, d.lif.rest, d.lif.reset, d.lif.thresh, d.lif.refrac, d.lif.dt = float(g["dc_decay"]), -65.0, -60.0, -52.0, 5.0, 1.0
    d.lif.traces, d.lif.trace_decay, d.lif.trace_scale = 1, float(g["dc_trace_decay"]), 1.0
    d.theta_decay, d.theta_plus, d.learning, d.one_spike = float(g["dc_theta_decay"]), 0.05, 1, 0
    cur = np.ascontiguousarray(I * f32(2.0))
    v = np.full((B, N), -65.0, f32); r = np.zeros((B, N), f32); s = np.zeros((B, N), u8); x = np.zeros((B, N), f32); th = np.zeros(N, f32)
    ras = np.zeros((T, B, N), u8)
    host.hostcheck_dc_sequence(p_(v), p_(r), p_(s), p_(x), p_(th), p_(cur), B, N, T, C.byref(d), p_(ras))
    v2 = np.full((B, N), -65.0, f32); r2 = np.zeros((B, N), f32); s2 = np.zeros((B, N), u8); x2 = np.zeros((B, N), f32); th2 = np.zeros(N, f32)
    ras2 = np.zeros((T, B, N), u8)
    Q, c0 = np.zeros(1, f32), np.zeros(1, np.int64)
    for t in range(T):
        oracle.dc_step(v2, r2, s2, x2, th2, cur[t].copy(), Q, c0, decay=float(g["dc_decay"]), rest=-65.0, reset=-60.0, thresh=-52.0, refrac0=5.0,
                       theta_decay=float(g["dc_theta_decay"]), theta_plus=0.05, one_spike=False, trace_decay=float(g["dc_trace_decay"]))
        ras2[t] = s2
    assert ras.sum() > 50
    np.testing.assert_array_equal(ras, ras2)
    for a, b_, key in ((v, v2, "v"), (r, r2, "refrac"), (x, x2, "x"), (th, th2, "theta")):
        np.testing.assert_array_equal(bits(a), bits(b_), err_msg=key)


def test_the_draw_comparison_margin_of_the_lean_arbitration_holds():
    """The lean D&C kernels pick the one_spike winner by comparing the 53-bit draws m_j instead of evaluating fl32(1 / fl32(-log1p(-u_j)))
    for every candidate (csrc/snn_dc2015_resident.hip: `zone = mmin + (mmin >> 19) + 1`): outside the zone the smaller draw must win
    STRICTLY in the reference's arithmetic.  Checked here for draws right at the edge of the zone, over the whole range of m:
    val(m) = fl32(1 / fl32(-log1p(-m * 2^-53))) is strictly larger than val(zone(m) + 1)."""
    rs = np.random.RandomState(11)
    ms = np.concatenate([np.arange(0, 4096, dtype=np.uint64), (rs.uniform(0, 1, 150000) * 2.0 ** 53).astype(np.uint64),
                         np.exp2(rs.uniform(0, 53, 150000)).astype(np.uint64), (1 << 53) - 1 - np.exp2(rs.uniform(0, 50, 50000)).astype(np.uint64)])
    top = (1 << 53) - 1

    def val(m):
        u = float(int(m)) * 2.0 ** -53
        q = np.float32(-1.0 * math.log1p(-u))
        with np.errstate(divide="ignore"):
            return np.float32(1.0) / q

    bad = 0
    for m1 in ms:
        m1 = int(m1)
        m2 = m1 + (m1 >> 19) + 2                      # the first draw OUTSIDE the zone
        if m2 > top:
            continue
        if not val(m1) > val(m2):
            bad += 1
    assert bad == 0


def test_inner_sum_and_conv_normalize_equal_torch_the_oracle_and_the_reference(host):
    """csrc/snn_order.hpp inner_sum8 (ATen's vectorised inner sum) and the thread body of k_normalize_filters
    (Conv2dConnection.normalize, topology.py:824-837) on the host: == torch.sum of a contiguous row for 1 ... 9000 elements, == the
    oracle, and the normalised filters == the reference fixture op_conv_normalize (2x2 ... 23x23 taps), bit for bit."""
    from cases import gold
    host.hostcheck_inner_sum8.restype = C.c_float
    rs = np.random.RandomState(3)
    for n in list(range(1, 70)) + [100, 127, 128, 129, 255, 256, 257, 511, 512, 513, 529, 1000, 2047, 2048, 4099, 9000]:
        x = (rs.uniform(size=n) - 0.3).astype(f32)
        got = np.float32(host.hostcheck_inner_sum8(p_(x), n))
        assert got == np.float32(torch.from_numpy(x).sum(0).item()), n
        assert got == np.float32(oracle.inner_sum(x)), n
    g = gold("op_conv_normalize")
    for k, (Cout, Cin, K) in enumerate(g["cases"]):
        Cout, Cin, K = int(Cout), int(Cin), int(K)
        W = synth.uniform_f32(3300 + k, (Cout, Cin, K, K), 0.05, 1.0)
        host.hostcheck_normalize_filters(p_(W), Cout * Cin, K * K, C.c_float(0.4 * K * K))
        np.testing.assert_array_equal(bits(W), bits(g[f"w{k}"]), err_msg=f"case {k}")
