"""GPU parity of whole Network.run() calls through the BindsNET-compatible API:
DiehlAndCook2015 (MCC path) bit-exact against reference-generated fixtures (rasters, weights,
theta, membrane state, host-RNG position), the dense Connection family bit-exact against the
order-pinned oracle and raster-exact / 1e-5 against the reference fixtures."""
import numpy as np
import pytest
import torch

import cases
import oracle
import synth
from cases import f32, u8, gold, unpack
from test_oracle_golden import DC_RUNS, dc_params, two_params, two_state

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def host(t):
    return t.detach().cpu().numpy()


def build_dc(N, B, inh, dt=1.0):
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=inh, dt=dt, norm=78.4, theta_plus=0.05,
                           inpt_shape=(1, 28, 28))
    net.connections[("X", "Ae")].pipeline[0].value.data.copy_(torch.from_numpy(synth.weights_q12(10, 784, N)))
    return net


PLAN_MODE = {"auto": 0, "generic": 1, "per-step": 2, "resident": 3}
PLAN_NAME = {"auto": "dc2015-resident-lean", "generic": "generic", "per-step": "dc2015-fused", "resident": "dc2015-resident"}


@pytest.mark.parametrize("plan", ["auto", "resident", "per-step", "generic"])
@pytest.mark.parametrize("name", DC_RUNS)
def test_dc2015_network_run_matches_reference(name, plan):
    from bindsnet_amd import _lib
    from bindsnet_amd.network.monitors import Monitor
    g = gold(name)
    N, B, T, runs = int(g["N"]), int(g["B"]), int(g["T"]), int(g["runs"])
    dt = float(g["dt"]) if "dt" in g.files else 1.0      # (run_dc_n100_b3_dt05: T timesteps of half a millisecond)
    net = build_dc(N, B, float(g["inh"]), dt)
    mons = {}
    for l in ("X", "Ae", "Ai"):
        mons[l] = Monitor(net.layers[l], ["s"], time=T)
        net.add_monitor(mons[l], l + "_s")
    mv = Monitor(net.layers["Ae"], ["v"], time=T)
    net.add_monitor(mv, "Ae_v")
    net.to(DEV)
    _lib.lib().snn_set_plan_mode(PLAN_MODE[plan])
    try:
        for r in range(runs):
            spikes = synth.spike_train(20 + r, T, B, 784, max_rate=float(g["max_rate"]))
            torch.manual_seed(2 + r)
            net.run({"X": torch.from_numpy(spikes).view(T, B, 1, 28, 28).to(DEV)}, time=T * dt)
            # host generator left exactly where the reference leaves it
            probe = torch.rand(4)
            torch.manual_seed(2 + r)
            if int(g[f"r{r}_consumed"]):
                torch.empty(int(g[f"r{r}_consumed"])).exponential_(1)
            assert torch.equal(probe, torch.rand(4)), "host RNG position after run"
            sE = host(mons["Ae"].get("s")).reshape(T, B, N)
            sI = host(mons["Ai"].get("s")).reshape(T, B, N)
            assert mons["Ae"].get("s").dtype == torch.bool and tuple(mons["X"].get("s").shape) == (T, B, 1, 28, 28)
            np.testing.assert_array_equal(host(mons["X"].get("s")).reshape(T, B, 784), spikes)
            np.testing.assert_array_equal(sE.astype(u8), unpack(g[f"r{r}_sE"], (T, B, N)), err_msg=f"run {r} Ae raster")
            np.testing.assert_array_equal(sI.astype(u8), unpack(g[f"r{r}_sI"], (T, B, N)), err_msg=f"run {r} Ai raster")
            W = host(net.connections[("X", "Ae")].pipeline[0].value)
            assert cases.sha(W) == str(g[f"r{r}_W_sha"]), f"run {r} weights"
            Ae, Ai, X = net.layers["Ae"], net.layers["Ai"], net.layers["X"]
            for key, a in (("theta", Ae.theta), ("vE", Ae.v), ("rE", Ae.refrac_count), ("xE", Ae.x),
                           ("xX", X.x.reshape(B, 784)), ("vI", Ai.v), ("rI", Ai.refrac_count)):
                np.testing.assert_array_equal(bits(host(a)), bits(g[f"r{r}_{key}"]), err_msg=f"run {r} {key}")
            np.testing.assert_array_equal(bits(host(mv.get("v"))[-1]), bits(g[f"r{r}_vE"]))
            if r % 2 == 0:
                net.reset_state_variables()
        # (auto: the lean form, or -- after it gave up on a busy step -- the general resident kernel)
        assert net.last_plan == PLAN_NAME[plan] or (plan == "auto" and net.last_plan == "dc2015-resident")
    finally:
        _lib.lib().snn_set_plan_mode(0)


def test_dc2015_test_mode_no_learning():
    """network.train(False): no STDP, theta frozen -- vs the oracle."""
    g = gold("run_dc_n100_b3")
    N, B, T = 100, 3, 60
    net = build_dc(N, B, 120.0)
    from bindsnet_amd.network.monitors import Monitor
    m = Monitor(net.layers["Ae"], ["s"], time=T)
    net.add_monitor(m, "m")
    net.train(False)
    net.to(DEV)
    net.layers["Ae"].theta.fill_(0.3)
    spikes = synth.spike_train(20, T, B, 784)
    torch.manual_seed(5)
    net.run({"X": torch.from_numpy(spikes).view(T, B, 1, 28, 28).to(DEV)}, time=T)
    P = dc_params(g, learning=False)
    st = cases.dc_state(N, B)
    st["theta"][:] = 0.3
    W0 = st["W_xe"].copy()
    cur = np.zeros(1, np.int64)
    rasE, _ = oracle.run_dc2015(P, st, spikes, cases.exp_noise(5, B * N * T), cur)
    np.testing.assert_array_equal(host(m.get("s")).reshape(T, B, N).astype(u8), rasE)
    np.testing.assert_array_equal(bits(host(net.layers["Ae"].theta)), bits(st["theta"]))
    np.testing.assert_array_equal(bits(host(net.connections[("X", "Ae")].pipeline[0].value)), bits(st["W_xe"]))
    assert not np.array_equal(st["W_xe"], W0)   # normalisation still happens (network.py:464-465)


@pytest.mark.parametrize("name,rule", [("run_two_postpre_b4", "postpre"), ("run_two_postpre_b32", "postpre"),
                                       ("run_two_mstdp_b4", "mstdp")])
def test_dense_family_network_run(name, rule):
    from bindsnet_amd.learning import MSTDP
    from bindsnet_amd.models import TwoLayerNetwork
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    g = gold(name)
    P = two_params(g, rule)
    W0 = torch.from_numpy(synth.weights_q12(11, P.Nin, P.N))
    torch.manual_seed(0)
    if rule == "postpre":
        net = TwoLayerNetwork(n_inpt=P.Nin, n_neurons=P.N, reduction=torch.sum, norm=78.4 * P.Nin / 784)
        conn = net.connections[("X", "Y")]
        conn.w.data.copy_(W0)
    else:
        net = Network(dt=1.0)
        net.add_layer(Input(n=P.Nin, traces=True), "X")
        net.add_layer(LIFNodes(n=P.N, traces=True), "Y")
        conn = Connection(net.layers["X"], net.layers["Y"], w=W0.clone(), wmin=0, wmax=1, update_rule=MSTDP, nu=1e-1,
                          norm=0.1 * P.Nin, reduction=torch.sum)
        net.add_connection(conn, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=P.T)
    net.add_monitor(mon, "Y_s")
    net.to(DEV)
    spikes = synth.spike_train(30, P.T, P.B, P.Nin, active=0.3, max_rate=0.12)
    kw = {"reward": 1.0} if rule == "mstdp" else {}
    net.run({"X": torch.from_numpy(spikes).to(DEV)}, time=P.T, **kw)
    ras = host(mon.get("s")).astype(u8)
    # bit-exact vs the order-pinned oracle
    st = two_state(P)
    ras_o = oracle.run_two_layer(P, st, spikes)
    np.testing.assert_array_equal(ras, ras_o)
    np.testing.assert_array_equal(bits(host(conn.w)), bits(st["W"]))
    np.testing.assert_array_equal(bits(host(net.layers["Y"].v)), bits(st["vY"]))
    np.testing.assert_array_equal(bits(host(net.layers["Y"].x)), bits(st["xY"]))
    if rule == "mstdp":
        np.testing.assert_array_equal(bits(host(conn.update_rule.p_plus)), bits(st["p_plus"]))
        assert cases.sha(host(conn.update_rule.eligibility)) == str(g["elig_sha"])
    # vs the reference itself (MKL propagation): rasters identical, weights within 1e-5 (north star)
    np.testing.assert_array_equal(ras, unpack(g["sY"], (P.T, P.B, P.N)))
    np.testing.assert_allclose(host(conn.w), g["W"], rtol=0, atol=1e-5)


def test_conv_lif_network_no_learning():
    """cfg4 shape: Input(1,28,28) -> Conv2dConnection 5x5x32 -> LIF(32,24,24), learning off."""
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Conv2dConnection
    B, T = 4, 30
    W = synth.uniform_f32(7, (32, 1, 5, 5), 0.0, 0.3)
    net = Network(dt=1.0, learning=False)
    net.add_layer(Input(shape=(1, 28, 28)), "X")
    net.add_layer(LIFNodes(shape=(32, 24, 24)), "Y")
    net.add_connection(Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=5, stride=1,
                                        w=torch.from_numpy(W)), "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    net.to(DEV)
    spikes = synth.dense_spikes(8, (T, B, 1, 28, 28), 0.25)
    net.run({"X": torch.from_numpy(spikes).to(DEV)}, time=T)
    # oracle: conv + LIF step by step
    n = 32 * 24 * 24
    v = np.full((B, n), -65.0, f32); r = np.zeros((B, n), f32); s = np.zeros((B, n), u8)
    prev = np.zeros((B, 1, 28, 28), u8)
    decay = float(net.layers["Y"].decay)
    ras = np.zeros((T, B, n), u8)
    for t in range(T):
        I = oracle.prop_conv2d(W, prev, bias=np.zeros(32, f32)).reshape(B, n)
        oracle.lif_step(v, r, s, None, I, decay=decay, rest=-65.0, reset=-65.0, thresh=-52.0, refrac0=5.0)
        ras[t] = s
        prev = spikes[t]
    assert ras.sum() > 100
    np.testing.assert_array_equal(host(mon.get("s")).reshape(T, B, n).astype(u8), ras)
    np.testing.assert_array_equal(bits(host(net.layers["Y"].v).reshape(B, n)), bits(v))


def test_unsupported_features_fail_loudly():
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.nodes import Input, LIFNodes
    net = Network()
    net.add_layer(Input(n=10), "X")
    net.add_layer(LIFNodes(n=5), "Y")
    net.to(DEV)
    with pytest.raises(NotImplementedError):
        net.run({"X": torch.zeros(5, 1, 10, dtype=torch.uint8, device=DEV)}, time=5, clamp={"X": torch.zeros(10).bool()})   # Input layers
    with pytest.raises(NotImplementedError):
        net.run({"X": torch.zeros(5, 1, 10, device=DEV)}, time=5)   # float inputs


def test_sparse_monitor_and_network_monitor_match_reference():
    """monitors.py:30-329: `Monitor(sparse=True)` hands back a sparse COO recording; `NetworkMonitor` keeps float
    recordings of s / v for several layers in a rolling window.  Fixtures: tests/golden/make_golden_r2.py."""
    from bindsnet_amd.network.monitors import Monitor, NetworkMonitor
    g = gold("net_monitor")
    N, B, T = int(g["N"]), int(g["B"]), int(g["T"])
    net = build_dc(N, B, 120.0)
    sp = Monitor(net.layers["Ae"], ["s"], time=T, sparse=True, batch_size=B)
    dn = Monitor(net.layers["Ae"], ["s", "v"], time=T, batch_size=B)
    net.add_monitor(sp, "Ae_sparse"); net.add_monitor(dn, "Ae_dense")
    net.to(DEV)
    spikes = synth.spike_train(20, T, B, 784)
    torch.manual_seed(2)
    net.run({"X": torch.from_numpy(spikes).view(T, B, 1, 28, 28).to(DEV)}, time=T)
    s = sp.get("s")
    assert s.is_sparse == bool(g["sparse_is_sparse"]) and list(s.shape) == list(g["sparse_shape"])
    np.testing.assert_array_equal(np.packbits(host(s.to_dense()).astype(u8)), g["sparse_dense"])
    np.testing.assert_array_equal(np.packbits(host(dn.get("s")).astype(u8)), g["dense_s"])
    np.testing.assert_array_equal(bits(host(dn.get("v"))), bits(g["dense_v"]))
    # NetworkMonitor, batch 1 (the reference sizes its buffers from the layers' current state tensors)
    g = gold("net_monitor_b1")
    N, T = int(g["N"]), int(g["T"])
    net = build_dc(N, 1, 120.0)
    net.to(DEV)
    nm = NetworkMonitor(net, layers=["Ae", "Ai"], connections=[], state_vars=["s", "v"], time=T)
    net.add_monitor(nm, "all")
    spikes = synth.spike_train(21, T, 1, 784, max_rate=0.2)
    torch.manual_seed(3)
    net.run({"X": torch.from_numpy(spikes).view(T, 1, 1, 28, 28).to(DEV)}, time=T)
    rec = nm.get()
    for l in ("Ae", "Ai"):
        assert rec[l]["s"].dtype == torch.float32 and list(rec[l]["s"].shape) == list(g[f"{l}_s_shape"])
        np.testing.assert_array_equal(np.packbits(host(rec[l]["s"]).astype(u8)), g[f"{l}_s"], err_msg=l)
        np.testing.assert_array_equal(bits(host(rec[l]["v"])), bits(g[f"{l}_v"]), err_msg=l)
    # rolling window: a shorter second run shifts the first one's tail in front of it
    first_tail = rec["Ae"]["v"][10:].clone()
    net.run({"X": torch.from_numpy(spikes[:10]).view(10, 1, 1, 28, 28).to(DEV)}, time=10)
    rec = nm.get()
    assert rec["Ae"]["v"].shape[0] == T and torch.equal(rec["Ae"]["v"][:T - 10], first_tail)
    with pytest.raises(NotImplementedError):
        from bindsnet_amd.models import TwoLayerNetwork
        two = TwoLayerNetwork(n_inpt=16, n_neurons=4, reduction=torch.sum)
        NetworkMonitor(two, state_vars=["b"])              # (connections: 'w' only)


def test_connection_weight_monitors_match_reference_and_oracle():
    """Monitor(connection, ["w"]) and NetworkMonitor's default ("v", "s", "w") on Input -> Connection[PostPre] -> LIF
    (monitors.py:94-111, 222-262; recorded at the end of every timestep): the per-step weights equal the oracle's, stepped,
    bit for bit, and the reference fixture's within the dense family's MKL tolerance; rasters exactly."""
    from test_oracle_golden import conn_monitor_params, oracle_weight_snapshots, two_state
    from bindsnet_amd.learning import PostPre
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor, NetworkMonitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    g = gold("conn_monitor")
    Nin, N, B, T = int(g["Nin"]), int(g["N"]), int(g["B"]), int(g["T"])

    def build():
        net = Network(dt=1.0)
        X_, Y_ = Input(n=Nin, traces=True), LIFNodes(n=N, traces=True)
        conn = Connection(X_, Y_, w=torch.from_numpy(synth.weights_q12(12, Nin, N) * np.float32(2.0)), update_rule=PostPre,
                          nu=(1e-2, 5e-2), reduction=torch.sum, wmin=0.0, wmax=2.0, norm=0.4 * Nin)
        net.add_layer(X_, "X"); net.add_layer(Y_, "Y")
        net.add_connection(conn, "X", "Y")
        return net, conn

    net, conn = build()
    mw = Monitor(conn, ["w"], time=T)
    net.add_monitor(mw, "w")
    net.to(DEV)
    P = conn_monitor_params(g)
    st = two_state(P)
    st["W"] = synth.weights_q12(12, Nin, N) * np.float32(2.0)
    for r in range(2):
        spikes = synth.spike_train(40 + r, T, B, Nin, active=0.5, max_rate=0.3)
        net.run({"X": torch.from_numpy(spikes).to(DEV)}, time=T)
        assert net.last_plan == "generic"                  # (the fused plans keep the weights on chip for the whole run)
        got = host(mw.get("w"))
        assert got.shape == (T, Nin, N)
        snaps, _ = oracle_weight_snapshots(P, st, spikes)
        np.testing.assert_array_equal(bits(got), bits(snaps), err_msg=f"run {r}: per-step weights vs oracle")
        np.testing.assert_array_equal(bits(host(conn.w)), bits(st["W"]), err_msg=f"run {r}: final weights vs oracle")
        np.testing.assert_allclose(got, g[f"r{r}_mon_w"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(host(conn.w), g[f"r{r}_final_w"], rtol=0, atol=1e-5)
        net.reset_state_variables()
        fresh = two_state(P)
        for k in ("sX", "xX", "vY", "rY", "sY", "xY"):
            st[k] = fresh[k]
    # NetworkMonitor with its default state_vars, no window, not reset between two runs: the recording grows
    net2, conn2 = build()
    nm = NetworkMonitor(net2)
    net2.add_monitor(nm, "all")
    net2.to(DEV)
    for r in range(2):
        net2.run({"X": torch.from_numpy(synth.spike_train(40 + r, T, B, Nin, active=0.5, max_rate=0.3)).to(DEV)}, time=T)
        assert net2.last_plan == "generic"
    rec = nm.get()
    assert sorted(f"{k}:{v}" for k in rec for v in rec[k]) == [str(x) for x in g["nm_keys"]]
    assert list(rec["Y"]["s"].shape) == list(g["nm_Y_s_shape"]) and list(rec["X"]["s"].shape) == list(g["nm_X_s_shape"])
    np.testing.assert_array_equal(np.packbits(host(rec["Y"]["s"]).astype(u8)), g["nm_Y_s"])
    np.testing.assert_allclose(host(rec[("X", "Y")]["w"]), g["nm_w"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(host(rec["Y"]["v"]), g["nm_Y_v"], rtol=0, atol=1e-4)
    # a window shorter than the run keeps the tail
    net3, conn3 = build()
    short = Monitor(conn3, ["w"], time=5)
    net3.add_monitor(short, "w")
    net3.to(DEV)
    spikes = synth.spike_train(40, T, B, Nin, active=0.5, max_rate=0.3)
    net3.run({"X": torch.from_numpy(spikes).to(DEV)}, time=T)
    np.testing.assert_allclose(host(short.get("w")), g["r0_mon_w"][-5:], rtol=0, atol=1e-5)
    with pytest.raises(NotImplementedError):
        net3.add_monitor(Monitor(conn3, ["b"], time=5), "b")
        net3.run({"X": torch.from_numpy(spikes).to(DEV)}, time=T)


def test_dc2015_v2_network_run_matches_oracle_and_reference():
    """DiehlAndCook2015v2 (models.py:247-346) on the generic plan: dense input connection with PostPre, recurrent inhibitory
    Connection fed by the layer's own previous spikes, one_spike arbitration on the device generator.  Bit for bit against
    the hand-stepped oracle (itself pinned to the reference fixture on the CPU), rasters / draws exactly and weights within
    the MKL tolerance against the reference."""
    from test_oracle_golden import dc_v2_run_oracle
    from bindsnet_amd.models import DiehlAndCook2015v2
    from bindsnet_amd.network.monitors import Monitor
    g = gold("run_dc_v2_n64_b4")
    N, B, T = int(g["N"]), int(g["B"]), int(g["T"])
    torch.manual_seed(0)
    net = DiehlAndCook2015v2(n_inpt=784, n_neurons=N, inh=60.0, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28),
                             reduction=torch.sum)
    conn = net.connections[("X", "Y")]
    conn.w.data.copy_(torch.from_numpy(synth.weights_q12(10, 784, N)))
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    net.to(DEV)
    st = dict(W=synth.weights_q12(10, 784, N), W_yy=(-60.0 * (np.ones((N, N), f32) - np.eye(N, dtype=f32))).astype(f32),
              theta=np.zeros(N, f32))
    for r in range(2):
        spikes = synth.spike_train(20 + r, T, B, 784, max_rate=0.25)
        torch.manual_seed(2 + r)
        net.run({"X": torch.from_numpy(spikes).view(T, B, 1, 28, 28).to(DEV)}, time=T)
        assert net.last_plan == "generic"
        probe = torch.rand(4)
        torch.manual_seed(2 + r)
        torch.empty(int(g[f"r{r}_consumed"])).exponential_(1)
        assert torch.equal(probe, torch.rand(4)), "host RNG position after run"
        ras_o, v_o, _ = dc_v2_run_oracle(g, r, st)
        ras = host(mon.get("s")).reshape(T, B, N).astype(u8)
        np.testing.assert_array_equal(ras, ras_o, err_msg=f"run {r} raster vs oracle")
        np.testing.assert_array_equal(ras, unpack(g[f"r{r}_sY"], (T, B, N)), err_msg=f"run {r} raster vs reference")
        np.testing.assert_array_equal(bits(host(conn.w)), bits(st["W"]), err_msg=f"run {r} weights vs oracle")
        np.testing.assert_array_equal(bits(host(net.layers["Y"].theta)), bits(g[f"r{r}_theta"]))
        np.testing.assert_array_equal(bits(host(net.layers["Y"].v)), bits(v_o))
        np.testing.assert_allclose(host(conn.w)[::7], g[f"r{r}_W_rows7"], rtol=0, atol=1e-5)
        net.reset_state_variables()
