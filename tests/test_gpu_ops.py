"""GPU parity: every per-operator C-ABI entry point vs the CPU oracle AND the reference-generated
golden fixtures, bit-exact (uint32 compare of f32 bit patterns)."""
import numpy as np
import pytest
import torch

import cases
import oracle
import synth
from cases import f32, u8, gold, check_packed, unpack

pytestmark = pytest.mark.gpu

DEV = "cuda"


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def lifp(**kw):
    from bindsnet_amd._lib import LifParams
    d = dict(decay=0.0, rest=-65.0, reset=-65.0, thresh=-52.0, refrac=5.0, dt=1.0, has_lbound=0, lbound=0.0,
             traces=0, trace_decay=0.0, trace_scale=1.0, traces_additive=0)
    d.update(kw)
    p = LifParams()
    for k, v in d.items():
        setattr(p, k, v)
    return p


def test_library_loaded_and_device_visible():
    from bindsnet_amd import _lib
    assert _lib.lib().snn_device_count() >= 1


@pytest.mark.parametrize("order", ["cascade", "dense"])
def test_prop_vs_oracle_and_golden(order):
    from bindsnet_amd import ops
    g = gold("op_prop_mcc")
    shapes = [tuple(int(v) for v in c[:3]) + (float(c[3]),) for c in g["cases"]]
    shapes += [(32, 784, 400, 0.012), (5, 2500, 130, 0.2), (3, 16, 5, 0.7), (1, 1, 1, 1.0)]
    for k, (B, Nin, N, p) in enumerate(shapes):
        W = synth.uniform_f32(100 + k, (Nin, N), -1.0, 1.0)
        s = synth.dense_spikes(200 + k, (B, Nin), p)
        out = torch.full((B, N), 7.0, device=DEV)
        if order == "cascade":
            ops.prop_cascade(dev(W), dev(s), out)
            ref = oracle.prop_mcc(W, s)
            if k < len(g["cases"]):
                np.testing.assert_array_equal(bits(host(out)), bits(g[f"out{k}"]), err_msg=f"golden {k}")
        else:
            bias = synth.uniform_f32(50 + k, (N,), -1, 1)
            ops.prop_dense(dev(W), dev(s), out, bias=dev(bias))
            ref = oracle.prop_dense(W, s, bias=bias)
        np.testing.assert_array_equal(bits(host(out)), bits(ref), err_msg=f"{order} case {k} {(B, Nin, N)}")
        # accumulate: out = out + r, bool spikes
        acc0 = synth.uniform_f32(70 + k, (B, N), -3, 3)
        out2 = dev(acc0)
        fn = ops.prop_cascade if order == "cascade" else ops.prop_dense
        fn(dev(W), dev(s).bool(), out2, accumulate=True)
        ref2 = acc0.copy()
        (oracle.prop_mcc if order == "cascade" else oracle.prop_dense)(W, s, out=ref2, accumulate=True)
        np.testing.assert_array_equal(bits(host(out2)), bits(ref2))


def test_prop_empty_and_multivalued_spikes():
    from bindsnet_amd import ops
    W = synth.uniform_f32(1, (300, 70), -1, 1)
    s = np.zeros((4, 300), u8)
    out = torch.full((4, 70), 3.0, device=DEV)
    ops.prop_cascade(dev(W), dev(s), out)
    assert (host(out) == 0).all()
    s = (synth.dense_spikes(2, (4, 300), 0.3) * np.random.RandomState(0).randint(1, 5, (4, 300))).astype(u8)
    ops.prop_cascade(dev(W), dev(s), out)
    np.testing.assert_array_equal(bits(host(out)), bits(oracle.prop_mcc(W, s)))


@pytest.mark.parametrize("family", ["mcc", "dense"])
def test_postpre_vs_golden_and_oracle(family):
    from bindsnet_amd import ops
    g = gold("op_postpre")
    nu0, nu1 = 1e-4, 1e-2
    shapes = [tuple(int(v) for v in c) for c in g["cases"]] + [(128, 100, 330), (200, 20, 40), (256, 33, 65)]
    for k, (B, Nin, N) in enumerate(shapes):
        W = synth.uniform_f32(300 + k, (Nin, N), 0.0, 1.0)
        s_src = synth.dense_spikes(400 + k, (B, Nin), 0.3); s_tgt = synth.dense_spikes(500 + k, (B, N), 0.2)
        x_src = synth.uniform_f32(600 + k, (B, Nin), 0.0, 1.0); x_tgt = synth.uniform_f32(700 + k, (B, N), 0.0, 1.0)
        Wd = dev(W)
        ops.stdp_postpre(Wd, dev(s_src), dev(x_src), dev(s_tgt).bool(), dev(x_tgt), nu0, nu1, use_dt=(family == "mcc"),
                         wmin=0.0, wmax=1.0)
        Wo = W.copy()
        oracle.postpre(Wo, s_src, x_src, s_tgt, x_tgt, nu0=nu0, nu1=nu1, use_dt=(family == "mcc"), wmin=0.0, wmax=1.0)
        np.testing.assert_array_equal(bits(host(Wd)), bits(Wo), err_msg=f"case {k} {(B, Nin, N)}")
        if k < len(g["cases"]):
            check_packed(g, f"{family}{k}", host(Wd))


def test_postpre_sparse_skip_is_exact():
    """assume_clamped=1 may skip untouched elements: result must equal the dense evaluation."""
    from bindsnet_amd import ops
    B, Nin, N = 32, 784, 400
    W = synth.uniform_f32(1, (Nin, N), 0.0, 1.0)
    s_src = synth.dense_spikes(2, (B, Nin), 0.012); s_tgt = np.zeros((B, N), u8)
    s_tgt[3, 17] = 1; s_tgt[20, 399] = 1; s_tgt[21, 17] = 1
    x_src = synth.uniform_f32(3, (B, Nin), 0.0, 1.0); x_tgt = synth.uniform_f32(4, (B, N), 0.0, 1.0)
    Wo = W.copy()
    oracle.postpre(Wo, s_src, x_src, s_tgt, x_tgt, nu0=1e-4, nu1=1e-2, use_dt=True, wmin=0.0, wmax=1.0)
    for clamped in (False, True):
        Wd = dev(W)
        ops.stdp_postpre(Wd, dev(s_src), dev(x_src), dev(s_tgt), dev(x_tgt), 1e-4, 1e-2, use_dt=True, wmin=0.0,
                         wmax=1.0, assume_clamped=clamped)
        np.testing.assert_array_equal(bits(host(Wd)), bits(Wo))
    # all-silent step: nothing changes
    z_src = np.zeros_like(s_src); z_tgt = np.zeros_like(s_tgt)
    Wd = dev(W)
    ops.stdp_postpre(Wd, dev(z_src), dev(x_src), dev(z_tgt), dev(x_tgt), 1e-4, 1e-2, use_dt=True, wmin=0.0, wmax=1.0,
                     assume_clamped=True)
    np.testing.assert_array_equal(bits(host(Wd)), bits(W))


@pytest.mark.parametrize("family", ["mcc", "dense"])
def test_normalize_vs_golden(family):
    from bindsnet_amd import ops
    g = gold("op_normalize")
    for k, (Nin, N) in enumerate(g["cases"]):
        Nin, N = int(Nin), int(N)
        W = synth.uniform_f32(800 + k, (Nin, N), -0.2, 1.0)
        W[:, N // 2] = 0.0
        Wd = dev(W)
        ops.normalize(Wd, 78.4, use_abs=(family == "dense"))
        check_packed(g, f"{family}{k}", host(Wd))


def test_lif_and_input_vs_golden():
    from bindsnet_amd import ops
    g = gold("op_nodes")
    B, N, T = int(g["B"]), int(g["N"]), int(g["T"])
    I = synth.uniform_f32(900, (T, B, N), -2.0, 6.0)
    v = torch.full((B, N), -60.0, device=DEV); r = torch.zeros(B, N, device=DEV)
    s = torch.zeros(B, N, dtype=torch.bool, device=DEV); x = torch.zeros(B, N, device=DEV)
    ras = torch.zeros(T, B, N, dtype=torch.uint8, device=DEV); rv = torch.zeros(T, B, N, device=DEV)
    p = lifp(decay=float(g["lif_decay"]), rest=-60.0, reset=-45.0, thresh=-40.0, refrac=2.0, has_lbound=1,
             lbound=-62.0, traces=1, trace_decay=float(g["lif_trace_decay"]))
    Id = dev(I)
    for t in range(T):
        ops.lif_step(v, r, s, x, Id[t], p, raster_s=ras[t], raster_v=rv[t])
    np.testing.assert_array_equal(host(ras), unpack(g["lif_s"], (T, B, N)))
    for a, key in ((v, "lif_v"), (x, "lif_x"), (r, "lif_r")):
        np.testing.assert_array_equal(bits(host(a)), bits(g[key]), err_msg=key)
    np.testing.assert_array_equal(bits(host(rv[-1])), bits(g["lif_v"]))
    # additive traces
    v.fill_(-65.0); r.zero_(); s.zero_(); x.zero_()
    p = lifp(decay=float(g["lifadd_decay"]), traces=1, trace_decay=float(g["lifadd_trace_decay"]), trace_scale=0.5,
             traces_additive=1)
    I3 = dev(I * np.float32(3))
    for t in range(T):
        ops.lif_step(v, r, s, x, I3[t], p)
    np.testing.assert_array_equal(bits(host(v)), bits(g["lifadd_v"]))
    np.testing.assert_array_equal(bits(host(x)), bits(g["lifadd_x"]))
    # input layer trace vs oracle
    sp = synth.dense_spikes(5, (T, B, N), 0.1)
    xo = np.zeros((B, N), f32); xd = torch.zeros(B, N, device=DEV); rr = torch.zeros(B, N, dtype=torch.uint8, device=DEV)
    spd = dev(sp)
    for t in range(T):
        oracle.input_step(sp[t], xo, float(g["lif_trace_decay"]))
        ops.input_step(spd[t], xd, float(g["lif_trace_decay"]), raster=rr)
    np.testing.assert_array_equal(bits(host(xd)), bits(xo))
    np.testing.assert_array_equal(host(rr), sp[-1])


def test_dc_nodes_vs_golden_including_rng():
    from bindsnet_amd import ops
    from bindsnet_amd._lib import DcParams
    g = gold("op_nodes")
    B, N, T = int(g["B"]), int(g["N"]), int(g["T"])
    I = dev(synth.uniform_f32(900, (T, B, N), -2.0, 6.0) * np.float32(2.0))
    Q = dev(cases.exp_noise(77, B * N * T))
    v = torch.full((B, N), -65.0, device=DEV); r = torch.zeros(B, N, device=DEV)
    s = torch.zeros(B, N, dtype=torch.bool, device=DEV); x = torch.zeros(B, N, device=DEV)
    theta = torch.zeros(N, device=DEV)
    cursor = torch.zeros(2, dtype=torch.int64, device=DEV); status = torch.zeros(1, dtype=torch.int32, device=DEV)
    ras = torch.zeros(T, B, N, dtype=torch.uint8, device=DEV)
    p = DcParams()
    p.lif = lifp(decay=float(g["dc_decay"]), rest=-65.0, reset=-60.0, thresh=-52.0, refrac=5.0, traces=1,
                 trace_decay=float(g["dc_trace_decay"]))
    p.theta_decay, p.theta_plus, p.learning, p.one_spike = float(g["dc_theta_decay"]), 0.05, 1, 1
    for t in range(T):
        ops.dc_step(v, r, s, x, theta, I[t], p, Q, cursor, status, raster_s=ras[t])
    assert int(status.item()) == 0
    np.testing.assert_array_equal(host(ras), unpack(g["dc_s"], (T, B, N)))
    assert int(cursor[0].item()) == int(g["dc_consumed"])
    for a, key in ((v, "dc_v"), (x, "dc_x"), (r, "dc_r"), (theta, "dc_theta")):
        np.testing.assert_array_equal(bits(host(a)), bits(g[key]), err_msg=key)
    # noise exhaustion is reported, not silently ignored
    v.fill_(-65.0); r.zero_(); cursor.zero_()
    ops.dc_step(v, r, s, x, theta, torch.full((B, N), 50.0, device=DEV), p, Q[: N], cursor, status)
    assert int(status.item()) == -4


def test_conv2d_vs_golden_and_oracle():
    from bindsnet_amd import ops
    g = gold("op_conv2d")
    for k, (B, Cin, H, Wd, Cout, K, stride, pad) in enumerate(g["cases"]):
        B, Cin, H, Wd, Cout, K, stride, pad = (int(v) for v in (B, Cin, H, Wd, Cout, K, stride, pad))
        W = synth.uniform_f32(1000 + k, (Cout, Cin, K, K), 0.0, 0.3)
        s = synth.dense_spikes(1100 + k, (B, Cin, H, Wd), 0.2)
        ref = oracle.prop_conv2d(W, s, stride=stride, pad=pad)
        out = torch.empty(ref.shape, device=DEV)
        ops.prop_conv2d(dev(W), dev(s), out, stride=stride, pad=pad)
        np.testing.assert_array_equal(bits(host(out)), bits(ref))
        check_packed(g, f"out{k}", host(out))          # C_in = 1, 3, 4, 8, 16: the reference's own numbers, bit for bit


def test_mstdp_vs_oracle():
    from bindsnet_amd import ops
    B, Nin, N, T = 6, 90, 40, 12
    W = synth.uniform_f32(1, (Nin, N), 0.0, 1.0)
    Wd = dev(W)
    elig = np.zeros((B, Nin, N), f32); pp = np.zeros((B, Nin), f32); pm = np.zeros((B, N), f32)
    ppd = torch.zeros(B, Nin, device=DEV); pmd = torch.zeros(B, N, device=DEV)
    sp_prev = torch.zeros(B, Nin, dtype=torch.uint8, device=DEV); tp_prev = torch.zeros(B, N, dtype=torch.uint8, device=DEV)
    dp, dm = float(np.exp(np.float32(-1 / 20.0))), float(np.exp(np.float32(-1 / 20.0)))
    for t in range(T):
        s_src = synth.dense_spikes(10 + t, (B, Nin), 0.1); s_tgt = synth.dense_spikes(40 + t, (B, N), 0.1)
        oracle.mstdp(W, elig, pp, pm, s_src, s_tgt, reward=0.7, nu0=0.1, decay_plus=dp, decay_minus=dm, wmin=0.0, wmax=1.0)
        ops.mstdp_step(Wd, ppd, pmd, sp_prev, tp_prev, dev(s_src), dev(s_tgt), 0.7, 0.1, 1.0, -1.0, dp, dm, wmin=0.0, wmax=1.0)
        np.testing.assert_array_equal(bits(host(Wd)), bits(W), err_msg=f"step {t}")
    np.testing.assert_array_equal(bits(host(ppd)), bits(pp))
    np.testing.assert_array_equal(bits(host(pmd)), bits(pm))


@pytest.mark.parametrize("B,N,p_row,seed,warm", [(5, 70, 0.5, 1, 0), (32, 400, 0.3, 2, 1000), (32, 400, 1.0, 3, 311),
                                                 (3, 100, 0.7, 4, 623), (64, 312, 0.9, 5, 0), (8, 39, 0.5, 6, 12345)])
def test_device_generator_matches_torch_stream(B, N, p_row, seed, warm):
    """snn_rng_fill_exponential == the draws torch.multinomial would consume, and the generator
    state it leaves == torch's state after the same number of draws (several consecutive steps)."""
    from bindsnet_amd import ops, rng
    torch.manual_seed(seed)
    if warm:
        torch.rand(warm)          # start somewhere inside a 624-block
    st0 = torch.get_rng_state()
    state = torch.from_numpy(rng.torch_state_to_words(st0).copy()).to(DEV)
    qbuf = torch.zeros(B * N, device=DEV)
    cursor = torch.zeros(2, dtype=torch.int64, device=DEV)
    rs = np.random.RandomState(seed)
    total = 0
    for step in range(6):
        rows = rs.uniform(size=B) < p_row
        cr = (rs.uniform(size=(B, N)) < 0.1) & rows[:, None]
        if step == 3:
            cr[:] = False                                       # a silent step consumes nothing
        ops.rng_fill_exponential(state, dev(cr.astype(u8)), qbuf, cursor)
        anyrow = cr.any(axis=1)
        r = int(anyrow.sum())
        ref = torch.empty(r * N).exponential_(1).numpy().reshape(r, N) if r else np.zeros((0, N), f32)
        got = host(qbuf)[: r * N].reshape(r, N)
        mask = cr[anyrow]
        np.testing.assert_array_equal(bits(got[mask]), bits(ref[mask]), err_msg=f"step {step}")
        total += r * N
    img = host(state)
    assert int(img.view(np.int64)[(rng.RNG_STATE_BYTES - 8) // 8]) == total
    assert torch.equal(rng.words_to_torch_state(img, st0), torch.get_rng_state())


@pytest.mark.parametrize("B,Nin,N,dens,bias,acc", [(16, 784, 1600, 0.012, False, False), (128, 784, 1600, 0.012, False, False),
                                                    (16, 6400, 500, 0.05, False, False), (5, 37, 21, 0.5, True, True),
                                                    (33, 130, 70, 0.3, True, False), (1, 784, 100, 0.02, False, True),
                                                    (3, 7, 16, 0.5, False, False), (20, 2053, 130, 0.1, True, True), (16, 1072, 64, 0.1, False, False)])
def test_prop_dense_mfma_is_bit_identical_to_the_ordered_sum(B, Nin, N, dens, bias, acc):
    """snn_prop_dense_mfma_f32 (v_mfma_f32_16x16x4_f32, one k-ordered chain per tile) vs the oracle's ascending-source
    sequential sum and vs the event-driven kernel: bit for bit, for 0/1 spikes."""
    from bindsnet_amd import ops
    W = synth.uniform_f32(31, (Nin, N), -1.0, 1.0)
    s = synth.dense_spikes(32, (B, Nin), dens)
    b = synth.uniform_f32(33, (N,), -0.5, 0.5) if bias else None
    out0 = synth.uniform_f32(34, (B, N), -1.0, 1.0) if acc else np.zeros((B, N), np.float32)
    ref = out0.copy()
    oracle.prop_dense(W, s, bias=b, out=ref, accumulate=acc)
    dW, ds = torch.from_numpy(W).to(DEV), torch.from_numpy(s).to(DEV)
    db = None if b is None else torch.from_numpy(b).to(DEV)
    got = torch.from_numpy(out0.copy()).to(DEV)
    ops.prop_dense_mfma(dW, ds, got, bias=db, accumulate=acc)
    ev = torch.from_numpy(out0.copy()).to(DEV)
    ops.prop_dense(dW, ds, ev, bias=db, accumulate=acc)
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    np.testing.assert_array_equal(ev.cpu().numpy().view(np.uint32), ref.view(np.uint32))
