"""The package on a machine WITHOUT a GPU: Network.run() of a network whose tensors are CPU tensors takes the
plain-PyTorch step loop (bindsnet_amd/network/host_path.py) -- pinned here against the same reference-generated
fixtures the MI355X path is tested with (tests/golden/make_golden*.py), bit for bit: rasters, weights, theta, state,
and the position of the host generator.  Also: the external-current and one_step + clamp fixtures of round 3."""
import numpy as np
import pytest
import torch

import synth
from cases import gold, sha, unpack

u8 = np.uint8


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("name", ["run_dc_n100_b1", "run_dc_n100_b3", "run_dc_n100_b3_busy", "run_dc_n400_b4"])
def test_dc2015_on_the_host_matches_reference(name):
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    g = gold(name)
    N, B, T, runs = int(g["N"]), int(g["B"]), int(g["T"]), int(g["runs"])
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=float(g["inh"]), dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    feat = net.connections[("X", "Ae")].pipeline[0]
    feat.value.data.copy_(T_(synth.weights_q12(10, 784, N)))
    mons = {l: Monitor(net.layers[l], ["s"], time=T) for l in ("Ae", "Ai")}
    for l, m in mons.items():
        net.add_monitor(m, l + "_s")
    Ae, Ai, X = net.layers["Ae"], net.layers["Ai"], net.layers["X"]
    for r in range(runs):
        spikes = synth.spike_train(20 + r, T, B, 784, max_rate=float(g["max_rate"]))
        torch.manual_seed(2 + r)
        net.run({"X": T_(spikes).view(T, B, 1, 28, 28)}, time=T)
        np.testing.assert_array_equal(mons["Ae"].get("s").numpy().reshape(T, B, N).astype(u8), unpack(g[f"r{r}_sE"], (T, B, N)), err_msg=f"run {r} Ae")
        np.testing.assert_array_equal(mons["Ai"].get("s").numpy().reshape(T, B, N).astype(u8), unpack(g[f"r{r}_sI"], (T, B, N)), err_msg=f"run {r} Ai")
        assert sha(feat.value.detach().numpy()) == str(g[f"r{r}_W_sha"]), f"run {r} weights"
        for key, a in (("theta", Ae.theta), ("vE", Ae.v), ("rE", Ae.refrac_count), ("xE", Ae.x), ("xX", X.x.reshape(B, 784)), ("vI", Ai.v), ("rI", Ai.refrac_count)):
            np.testing.assert_array_equal(bits(a.numpy()), bits(g[f"r{r}_{key}"]), err_msg=f"run {r} {key}")
        if r % 2 == 0:
            net.reset_state_variables()


def test_two_layer_postpre_on_the_host_matches_reference():
    """TwoLayerNetwork (dense Connection + PostPre): rasters identical to the reference; weights / state within the dense
    family's tolerance (Connection.compute is an MKL sgemm whose summation order depends on the thread count of the
    machine the fixture was made on -- SURVEY.md finding 5)."""
    from bindsnet_amd.models import TwoLayerNetwork
    from bindsnet_amd.network.monitors import Monitor
    g = gold("run_two_postpre_b4")
    Nin, N, B, T = int(g["Nin"]), int(g["N"]), int(g["B"]), int(g["T"])
    torch.manual_seed(0)
    net = TwoLayerNetwork(n_inpt=Nin, n_neurons=N, reduction=torch.sum, norm=78.4 * Nin / 784)
    conn = net.connections[("X", "Y")]
    conn.w.data.copy_(T_(synth.weights_q12(11, Nin, N)))
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    spikes = synth.spike_train(30, T, B, Nin, active=0.3, max_rate=0.12)
    net.run({"X": T_(spikes)}, time=T)
    np.testing.assert_array_equal(mon.get("s").numpy().reshape(T, B, N).astype(u8), unpack(g["sY"], (T, B, N)))
    np.testing.assert_allclose(conn.w.detach().numpy(), g["W"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(net.layers["Y"].v.numpy(), g["vY"], rtol=0, atol=1e-4)
    np.testing.assert_array_equal(bits(net.layers["X"].x.numpy()), bits(g["xX"]))


def test_external_currents_and_one_step_clamp_on_the_host():
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import MulticompartmentConnection
    from bindsnet_amd.network.topology_features import Weight
    nX, nA, nB, B, T = 64, 40, 24, 3, 30

    def chain():
        net = Network(dt=1.0, learning=False)
        net.add_layer(Input(n=nX), "X"); net.add_layer(LIFNodes(n=nA, thresh=-60.0), "A"); net.add_layer(LIFNodes(n=nB, thresh=-61.0), "B")
        for k, (src, dst, ns, nd, sc) in enumerate((("X", "A", nX, nA, 2.0), ("A", "B", nA, nB, 0.35), ("B", "A", nB, nA, -1.0))):
            w = (synth.uniform_f32(3200 + k, (ns, nd), 0.0, abs(sc)) * np.sign(sc)).astype(np.float32)
            net.add_connection(MulticompartmentConnection(net.layers[src], net.layers[dst], device="cpu", pipeline=[Weight("weight", T_(w).clone())]), src, dst)
        mons = {l: Monitor(net.layers[l], ["s", "v"], time=T) for l in ("A", "B")}
        for l, m in mons.items():
            net.add_monitor(m, l)
        return net, mons

    g = gold("run_ext_current")
    net, mons = chain()
    sp = synth.dense_spikes(3210, (T, B, nX), 0.10)
    net.run({"X": T_(sp), "A": T_(synth.uniform_f32(3211, (T, B, nA), -1.0, 4.0)), "B": T_(synth.uniform_f32(3212, (T, B, nB), 0.0, 2.5))}, time=T)
    for l, n in (("A", nA), ("B", nB)):
        np.testing.assert_array_equal(mons[l].get("s").numpy().reshape(T, B, n).astype(u8), unpack(g[f"s_{l}"], (T, B, n)), err_msg=f"raster {l}")
        np.testing.assert_array_equal(bits(mons[l].get("v").numpy().reshape(T, B, n)), bits(g[f"v_{l}"].reshape(T, B, n)), err_msg=f"v {l}")

    g = gold("run_one_step_clamp")
    net, mons = chain()
    sp = synth.dense_spikes(3220, (T, B, nX), 0.15)
    clampA = torch.zeros(nA, dtype=torch.bool); clampA[::7] = True
    unclampA = torch.zeros(nA, dtype=torch.bool); unclampA[3::5] = True
    for tag, flag in (("one", True), ("sync", False)):
        net.reset_state_variables()
        net.run({"X": T_(sp)}, time=T, one_step=flag, clamp={"A": clampA}, unclamp={"A": unclampA})
        for l, n in (("A", nA), ("B", nB)):
            np.testing.assert_array_equal(mons[l].get("s").numpy().reshape(T, B, n).astype(u8), unpack(g[f"{tag}_{l}"], (T, B, n)), err_msg=f"{tag} {l}")


def test_mstdp_on_the_host_matches_reference():
    """Input -> Connection(MSTDP) -> LIF (the cfg5 graph, small): rasters and the rule's traces identical to the reference,
    weights within the dense family's tolerance (MKL propagation)."""
    from bindsnet_amd.learning import MSTDP
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    g = gold("run_two_mstdp_b4")
    Nin, N, B, T = int(g["Nin"]), int(g["N"]), int(g["B"]), int(g["T"])
    net = Network(dt=1.0)
    net.add_layer(Input(n=Nin, traces=True), "X")
    net.add_layer(LIFNodes(n=N, traces=True), "Y")
    conn = Connection(net.layers["X"], net.layers["Y"], w=T_(synth.weights_q12(11, Nin, N)).clone(), wmin=0, wmax=1, update_rule=MSTDP,
                      nu=1e-1, norm=0.1 * Nin, reduction=torch.sum)
    net.add_connection(conn, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    spikes = synth.spike_train(30, T, B, Nin, active=0.3, max_rate=0.12)
    net.run({"X": T_(spikes)}, time=T, reward=1.0)
    assert net.last_plan == "host-torch"
    np.testing.assert_array_equal(mon.get("s").numpy().reshape(T, B, N).astype(u8), unpack(g["sY"], (T, B, N)))
    np.testing.assert_allclose(conn.w.detach().numpy(), g["W"], rtol=0, atol=1e-5)
    np.testing.assert_array_equal(bits(conn.update_rule.p_plus.numpy()), bits(g["p_plus"]))
    np.testing.assert_array_equal(bits(conn.update_rule.p_minus.numpy()), bits(g["p_minus"]))


@pytest.mark.parametrize("rule", ["hebbian", "wdpp"])
def test_hebbian_and_wdpp_on_the_host_vs_oracle(rule):
    """The two outer-product rules on the host path against the order-pinned oracle run (the checker of the MI355X
    variants): rasters identical, weights within the dense family's tolerance."""
    import oracle
    from bindsnet_amd.learning import Hebbian, WeightDependentPostPre
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    Nin, N, B, T = 196, 48, 6, 40
    W0 = synth.weights_q12(11, Nin, N)
    net = Network(dt=1.0)
    net.add_layer(Input(n=Nin, traces=True), "X")
    net.add_layer(LIFNodes(n=N, traces=True), "Y")
    conn = Connection(net.layers["X"], net.layers["Y"], w=T_(W0).clone(), wmin=0.0, wmax=1.0,
                      update_rule={"hebbian": Hebbian, "wdpp": WeightDependentPostPre}[rule], nu=(1e-4, 1e-3), norm=0.1 * Nin, reduction=torch.sum)
    net.add_connection(conn, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    spikes = synth.spike_train(30, T, B, Nin, active=0.3, max_rate=0.12)
    net.run({"X": T_(spikes)}, time=T)
    P = oracle.TwoParams()
    P.B, P.Nin, P.N, P.T, P.dt = B, Nin, N, T, 1.0
    P.rule = {"hebbian": 3, "wdpp": 4}[rule]
    P.x_trace_decay = float(net.layers["X"].trace_decay); P.x_trace_scale = 1.0; P.x_traces = 1
    P.decay = float(net.layers["Y"].decay); P.rest, P.reset, P.thresh, P.refrac = -65.0, -65.0, -52.0, 5.0
    P.y_traces = 1; P.y_trace_decay = float(net.layers["Y"].trace_decay); P.y_trace_scale = 1.0
    P.nu0, P.nu1 = 1e-4, 1e-3
    P.has_min = P.has_max = 1; P.wmin, P.wmax = 0.0, 1.0; P.has_norm = 1; P.norm = 0.1 * Nin; P.learning = 1
    st = dict(W=W0.copy(), sX=np.zeros((B, Nin), u8), xX=np.zeros((B, Nin), np.float32), vY=np.full((B, N), -65.0, np.float32),
              rY=np.zeros((B, N), np.float32), sY=np.zeros((B, N), u8), xY=np.zeros((B, N), np.float32))
    ras = oracle.run_two_layer(P, st, spikes)
    assert ras.sum() > 20
    np.testing.assert_array_equal(mon.get("s").numpy().reshape(T, B, N).astype(u8), ras)
    np.testing.assert_allclose(conn.w.detach().numpy(), st["W"], rtol=0, atol=1e-5)


# ------------------------------------------------------------------------------------------------ the remaining rules
@pytest.mark.parametrize("name", ["run_two_mcc_mstdp_b4", "run_two_mcc_mstdp_b20", "run_two_mcc_mstdp_n208"])
def test_mcc_mstdp_on_the_host_matches_reference(name):
    """MulticompartmentConnection + Weight with MCC_learning.MSTDP: two consecutive runs (scalar reward, then per-sample
    rewards) bit for bit against the reference fixture the MI355X path is tested with."""
    import cases
    from bindsnet_amd.learning.MCC_learning import MSTDP
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import MulticompartmentConnection
    from bindsnet_amd.network.topology_features import Weight
    g = gold(name)
    Nin, N, B, T = int(g["Nin"]), int(g["N"]), int(g["B"]), int(g["T"])
    net = Network(dt=1.0)
    X_, Y_ = Input(n=Nin, traces=True), LIFNodes(n=N, traces=True)
    feat = Weight("weight", T_(synth.weights_q12(11, Nin, N)), range=[0.0, 1.0], norm=0.1 * Nin, nu=(1e-1, 1e-1), learning_rule=MSTDP)
    conn = MulticompartmentConnection(X_, Y_, device="cpu", pipeline=[feat])      # (before the layers get a batch size: reduction = sum)
    net.add_layer(X_, "X"); net.add_layer(Y_, "Y")
    net.add_connection(conn, "X", "Y")
    mon = Monitor(Y_, ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    for r in range(2):
        spikes = synth.spike_train(30 + r, T, B, Nin, active=0.3, max_rate=0.12)
        reward = 1.0 if r == 0 else T_(synth.uniform_f32(17, (B,), -1.0, 1.0)).view(B, 1, 1)
        net.run({"X": T_(spikes)}, time=T, reward=reward)
        assert net.last_plan == "host-torch"
        rule = feat.learning_rule
        np.testing.assert_array_equal(mon.get("s").numpy().reshape(T, B, N).astype(u8), unpack(g[f"r{r}_sY"], (T, B, N)))
        for got, key in ((feat.value, "W"), (Y_.v, "vY"), (rule.p_plus, "p_plus"), (rule.p_minus, "p_minus")):
            np.testing.assert_array_equal(bits(got.detach().numpy()), bits(g[f"r{r}_{key}"]), err_msg=f"run {r} {key}")
        assert cases.sha(rule.eligibility.numpy()) == str(g[f"r{r}_elig_sha"])
        net.reset_state_variables()


def test_mcc_mstdpet_on_the_host_matches_reference():
    """MCC_learning.MSTDPET (batch 1): two runs with different reward / a_plus, layers reset in between, the rule's state
    kept -- everything bit for bit against the reference fixture."""
    from bindsnet_amd.learning.MCC_learning import MSTDPET
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import MulticompartmentConnection
    from bindsnet_amd.network.topology_features import Weight
    g = gold("run_two_mcc_mstdpet_b1")
    Nin, N, T = int(g["Nin"]), int(g["N"]), int(g["T"])
    net = Network(dt=1.0)
    X_, Y_ = Input(n=Nin, traces=True), LIFNodes(n=N, traces=True)
    feat = Weight("weight", T_(synth.weights_q12(11, Nin, N)), range=[0.0, 1.0], norm=0.1 * Nin, nu=(1e-1, 1e-1), learning_rule=MSTDPET)
    conn = MulticompartmentConnection(X_, Y_, device="cpu", pipeline=[feat], tc_e_trace=25.0)
    net.add_layer(X_, "X"); net.add_layer(Y_, "Y")
    net.add_connection(conn, "X", "Y")
    mon = Monitor(Y_, ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    rule = feat.learning_rule
    for r in range(2):
        spikes = synth.spike_train(30 + r, T, 1, Nin, active=0.3, max_rate=0.12)
        net.run({"X": T_(spikes)}, time=T, reward=0.8 if r == 0 else -0.5, a_plus=1.0 if r == 0 else 0.75)
        np.testing.assert_array_equal(mon.get("s").numpy().reshape(T, 1, N).astype(u8), unpack(g[f"r{r}_sY"], (T, 1, N)))
        for got, key in ((feat.value, "W"), (Y_.v, "vY"), (rule.p_plus, "p_plus"), (rule.p_minus, "p_minus"), (rule.eligibility, "elig"),
                         (rule.eligibility_trace, "e_trace")):
            np.testing.assert_array_equal(bits(got.detach().numpy().reshape(-1)), bits(g[f"r{r}_{key}"].reshape(-1)), err_msg=f"run {r} {key}")
        net.reset_state_variables()


def test_dense_mstdpet_on_the_host_vs_oracle():
    import oracle
    from bindsnet_amd.learning import MSTDPET
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    Nin, N, B, T = 196, 48, 1, 40
    W0 = synth.weights_q12(11, Nin, N)
    net = Network(dt=1.0)
    net.add_layer(Input(n=Nin, traces=True), "X")
    net.add_layer(LIFNodes(n=N, traces=True), "Y")
    conn = Connection(net.layers["X"], net.layers["Y"], w=T_(W0).clone(), wmin=0.0, wmax=1.0, update_rule=MSTDPET, nu=(1e-1, 1e-1),
                      norm=0.1 * Nin, reduction=torch.sum)
    net.add_connection(conn, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    spikes = synth.spike_train(30, T, B, Nin, active=0.3, max_rate=0.12)
    net.run({"X": T_(spikes)}, time=T, reward=0.8)
    P = oracle.TwoParams()
    P.B, P.Nin, P.N, P.T, P.dt, P.rule = B, Nin, N, T, 1.0, 5
    P.x_trace_decay = float(net.layers["X"].trace_decay); P.x_trace_scale = 1.0; P.x_traces = 1
    P.decay = float(net.layers["Y"].decay); P.rest, P.reset, P.thresh, P.refrac = -65.0, -65.0, -52.0, 5.0
    P.y_traces = 1; P.y_trace_decay = float(net.layers["Y"].trace_decay); P.y_trace_scale = 1.0
    P.nu0, P.nu1 = 1e-1, 1e-1
    P.has_min = P.has_max = 1; P.wmin, P.wmax = 0.0, 1.0; P.has_norm = 1; P.norm = 0.1 * Nin; P.learning = 1
    dp, dm, de = conn.update_rule._decays()
    P.reward, P.a_plus, P.a_minus, P.decay_plus, P.decay_minus, P.decay_e, P.tc_e = 0.8, 1.0, -1.0, dp, dm, de, 25.0
    st = dict(W=W0.copy(), sX=np.zeros((B, Nin), u8), xX=np.zeros((B, Nin), np.float32), vY=np.full((B, N), -65.0, np.float32),
              rY=np.zeros((B, N), np.float32), sY=np.zeros((B, N), u8), xY=np.zeros((B, N), np.float32),
              elig=np.zeros((Nin, N), np.float32), e_trace=np.zeros((Nin, N), np.float32), p_plus=np.zeros(Nin, np.float32),
              p_minus=np.zeros(N, np.float32))
    ras = oracle.run_two_layer(P, st, spikes)
    assert ras.sum() > 20
    np.testing.assert_array_equal(mon.get("s").numpy().reshape(T, B, N).astype(u8), ras)
    np.testing.assert_allclose(conn.w.detach().numpy(), st["W"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(conn.update_rule.eligibility_trace.numpy(), st["e_trace"], rtol=0, atol=1e-6)


def test_local_connection_and_conv2d_learning_on_the_host_match_reference():
    """LocalConnection + PostPre (structural mask, signed normalisation), Conv2dConnection + PostPre and + MSTDP (batch 1):
    whole runs on the host against the reference fixtures of the MI355X tests -- rasters identical, weights within the
    dense family's tolerance (the reference's matmuls / bmms run in BLAS order)."""
    from bindsnet_amd.learning import MSTDP, PostPre
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Conv2dConnection, LocalConnection
    g = gold("run_extras")
    np.random.seed(7)
    T2 = 50
    net = Network(dt=1.0)
    X, Y = Input(n=144, traces=True), LIFNodes(n=4 * 16, traces=True)
    lc = LocalConnection(X, Y, kernel_size=6, stride=2, n_filters=4, update_rule=PostPre, nu=(1e-4, 1e-2), wmin=0.0, wmax=1.0, norm=0.2)
    net.add_layer(X, "X"); net.add_layer(Y, "Y"); net.add_connection(lc, "X", "Y")
    mon = Monitor(Y, ["s"], time=T2)
    net.add_monitor(mon, "s")
    net.run({"X": T_(synth.spike_train(31, T2, 1, 144, active=0.5, max_rate=0.25))}, time=T2)
    np.testing.assert_array_equal(mon.get("s").numpy().reshape(T2, 1, 64).astype(u8), unpack(g["lc_sY"], (T2, 1, 64)))
    np.testing.assert_allclose(lc.w.detach().numpy(), g["lc_W"], rtol=0, atol=1e-5)
    assert (lc.w.detach().numpy()[lc.mask.numpy()] == 0).all()

    B, T3 = 2, 30
    net = Network(dt=1.0)
    net.add_layer(Input(shape=(1, 12, 12), traces=True), "X")
    net.add_layer(LIFNodes(shape=(4, 10, 10), traces=True), "Y")
    cc = Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=3, stride=1, w=T_(synth.uniform_f32(1700, (4, 1, 3, 3), 0.0, 3.0)).clone(),
                          update_rule=PostPre, nu=(1e-3, 1e-2), reduction=torch.sum, wmin=0.0, wmax=4.0)
    net.add_connection(cc, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T3)
    net.add_monitor(mon, "s")
    net.run({"X": T_(synth.dense_spikes(1701, (T3, B, 1, 12, 12), 0.2))}, time=T3)
    np.testing.assert_array_equal(mon.get("s").numpy().reshape(T3, B, 400).astype(u8), unpack(g["crun_sY"], (T3, B, 400)))
    np.testing.assert_allclose(cc.w.detach().numpy(), g["crun_W"], rtol=0, atol=1e-5 * 4.0)

    g = gold("op_conv_mstdp")
    T4 = 40
    net = Network(dt=1.0)
    net.add_layer(Input(shape=(1, 12, 12), traces=True), "X")
    net.add_layer(LIFNodes(shape=(4, 10, 10), traces=True), "Y")
    cm = Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=3, stride=1, w=T_(synth.uniform_f32(2290, (4, 1, 3, 3), 0.0, 3.0)).clone(),
                          update_rule=MSTDP, nu=(2e-3, 1e-3), wmin=0.0, wmax=4.0)
    net.add_connection(cm, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T4)
    net.add_monitor(mon, "s")
    net.run({"X": T_(synth.dense_spikes(2291, (T4, 1, 1, 12, 12), 0.2))}, time=T4, reward=0.6)
    np.testing.assert_array_equal(mon.get("s").numpy().reshape(T4, 400).astype(u8), unpack(g["run_sY"], (T4, 400)))
    np.testing.assert_allclose(cm.w.detach().numpy(), g["run_W"], rtol=0, atol=1e-5 * 4.0)
    np.testing.assert_allclose(cm.update_rule.eligibility.numpy(), g["run_elig"], rtol=0, atol=1e-5 * max(1.0, float(np.abs(g["run_elig"]).max())))


@pytest.mark.parametrize("N,B,T,Nin,inh,rate,learning,threads", [
    (37, 2, 60, 196, 60.0, 0.35, True, 8), (68, 1, 80, 784, 120.0, 0.25, True, 16), (100, 3, 50, 784, 17.5, 0.3, True, 12),
    (100, 1, 60, 784, 120.0, 0.3, False, 9), (64, 5, 40, 400, 30.0, 0.4, True, 1), (132, 2, 40, 784, 120.0, 0.25, True, 16)])
def test_dc2015_on_the_host_vs_oracle_fuzz(N, B, T, Nin, inh, rate, learning, threads):
    """D&C runs on the host at sizes the reference fixtures do not cover -- column tails of 4 / 5 columns (the shapes where ATen's
    order depends on the thread count: the host path must stay on the serial one at any setting), test mode, small inputs --
    against the oracle, two consecutive inputs: rasters, weights, theta, state and the number of draws, bit for bit."""
    import oracle
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    side = int(round(Nin ** 0.5))
    n0 = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        torch.manual_seed(0)
        net = DiehlAndCook2015(n_inpt=Nin, n_neurons=N, exc=22.5, inh=inh, dt=1.0, norm=78.4 * Nin / 784, theta_plus=0.05, inpt_shape=(1, side, side))
        W0 = synth.weights_q12(10, Nin, N)
        feat = net.connections[("X", "Ae")].pipeline[0]
        feat.value.data.copy_(T_(W0))
        net.train(learning)
        mons = {l: Monitor(net.layers[l], ["s"], time=T) for l in ("Ae", "Ai")}
        for l, m in mons.items():
            net.add_monitor(m, l)
        P = oracle.eth_mnist_dc_params(N, B, T, Nin=Nin, learning=learning)
        P.norm = 78.4 * Nin / 784
        st = oracle.eth_mnist_dc_state(N, B, W0.copy(), Nin=Nin, inh=inh)
        for r in range(2):
            sp = synth.spike_train(60 + r, T, B, Nin, max_rate=rate)
            Q = oracle.exp_noise(9 + r, B * N * T + 16)
            cur = np.zeros(1, np.int64)
            rasE, rasI = oracle.run_dc2015(P, st, sp, Q, cur)
            torch.manual_seed(9 + r)
            net.run({"X": T_(sp).view(T, B, 1, side, side)}, time=T)
            got_probe = torch.rand(2)
            np.testing.assert_array_equal(mons["Ae"].get("s").numpy().reshape(T, B, N).astype(u8), rasE, err_msg=f"input {r} Ae")
            np.testing.assert_array_equal(mons["Ai"].get("s").numpy().reshape(T, B, N).astype(u8), rasI, err_msg=f"input {r} Ai")
            np.testing.assert_array_equal(bits(feat.value.detach().numpy()), bits(st["W_xe"]), err_msg=f"input {r} weights")
            for a, key in ((net.layers["Ae"].theta, "theta"), (net.layers["Ae"].v, "vE"), (net.layers["Ae"].x, "xE"), (net.layers["Ai"].v, "vI")):
                np.testing.assert_array_equal(bits(a.numpy()), bits(st[key]), err_msg=f"input {r} {key}")
            torch.manual_seed(9 + r)                                  # the host generator stands where `cur` draws leave it
            if int(cur[0]):
                torch.empty(int(cur[0])).exponential_(1)
            assert torch.equal(got_probe, torch.rand(2)), f"input {r}: host generator position"
            assert rasE.sum() > 3
            if r == 0:
                net.reset_state_variables()
                for k in ("sX", "xX", "sE", "xE", "rE", "rI", "sI"):
                    st[k][:] = 0
                st["vE"][:] = -65.0
                st["vI"][:] = -60.0
        assert torch.get_num_threads() == threads
    finally:
        torch.set_num_threads(n0)


def test_conv2d_normalize_on_the_host_matches_reference():
    """Conv2dConnection.normalize by hand and inside run() on the host: the op-level fixture bit for bit; a conv_mnist.py style run
    (Conv2d + PostPre + norm, two consecutive inputs, normalised after each): rasters identical, weights within the dense
    family's tolerance (the rule's bmm runs in BLAS order in the reference), every filter summing to norm."""
    from bindsnet_amd.learning import PostPre
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Conv2dConnection
    g = gold("op_conv_normalize")
    for k, (Cout, Cin, K) in enumerate(g["cases"]):
        Cout, Cin, K = int(Cout), int(Cin), int(K)
        c = Conv2dConnection(Input(shape=(Cin, K + 3, K + 3)), LIFNodes(shape=(Cout, 4, 4)), kernel_size=K,
                             w=T_(synth.uniform_f32(3300 + k, (Cout, Cin, K, K), 0.05, 1.0)).clone(), norm=0.4 * K * K)
        c.normalize()
        np.testing.assert_array_equal(bits(c.w.detach().numpy()), bits(g[f"w{k}"]), err_msg=f"case {k}")
    B2, T2 = 2, 30
    net = Network(dt=1.0)
    net.add_layer(Input(shape=(1, 12, 12), traces=True), "X")
    net.add_layer(LIFNodes(shape=(4, 10, 10), traces=True), "Y")
    cc = Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=3, stride=1, w=T_(synth.uniform_f32(3390, (4, 1, 3, 3), 0.0, 3.0)).clone(),
                          update_rule=PostPre, nu=(1e-3, 1e-2), reduction=torch.sum, wmin=0.0, wmax=4.0, norm=9.0)
    net.add_connection(cc, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T2)
    net.add_monitor(mon, "s")
    for r in range(2):
        net.run({"X": T_(synth.dense_spikes(3391 + r, (T2, B2, 1, 12, 12), 0.2))}, time=T2)
        np.testing.assert_array_equal(np.packbits(mon.get("s").numpy().astype(u8)), g[f"run{r}_sY"], err_msg=f"run {r} raster")
        np.testing.assert_allclose(cc.w.detach().numpy(), g[f"run{r}_W"], rtol=0, atol=1e-5 * 4.0)
        np.testing.assert_allclose(cc.w.detach().view(4, -1).sum(1).numpy(), 9.0, rtol=1e-6)
        net.reset_state_variables()


def _conv_mnist_net():
    """examples/mnist/conv_mnist.py:84-126 through the mirror (the graph builder the reference fixture was made with)."""
    import importlib.util
    import os
    import types
    from bindsnet_amd.learning import PostPre
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import DiehlAndCookNodes, Input
    from bindsnet_amd.network.topology import Connection, Conv2dConnection
    mod = types.SimpleNamespace(Network=Network, Input=Input, DiehlAndCookNodes=DiehlAndCookNodes, Conv2dConnection=Conv2dConnection,
                                Connection=Connection, PostPre=PostPre, Monitor=Monitor)
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden_r3.py")).read()
    body = src[src.index("def conv_mnist_graph("):src.index("def conv_mnist_case(")]
    ns = {"torch": torch}
    exec(body, ns)                                                  # (the generator module itself imports the reference at import time)
    return ns["conv_mnist_graph"](mod)


def check_conv_mnist_graph(net, mons, cc, dev="cpu"):
    g = gold("run_conv_mnist_graph")
    host = lambda t: t.detach().cpu().numpy()                      # noqa: E731
    np.testing.assert_array_equal(bits(host(cc.w)), bits(g["W0"]), err_msg="construction draws")
    T3 = 60
    torch.manual_seed(2)
    for r in range(2):
        sp = synth.spike_train(3400 + r, T3, 1, 784, active=0.5, max_rate=0.35)
        net.run({"X": T_(sp).view(T3, 1, 1, 28, 28).to(dev)}, time=T3)
        np.testing.assert_array_equal(np.packbits(host(mons["s"].get("s")).astype(u8)), g[f"r{r}_sY"], err_msg=f"run {r} raster")
        np.testing.assert_allclose(host(mons["v"].get("v")), g[f"r{r}_vY"], rtol=0, atol=2e-4, err_msg=f"run {r} voltages")
        np.testing.assert_allclose(host(cc.w), g[f"r{r}_W"], rtol=0, atol=1e-5, err_msg=f"run {r} weights")
        np.testing.assert_array_equal(bits(host(net.layers["Y"].theta)), bits(g[f"r{r}_theta"]), err_msg=f"run {r} theta")
        net.reset_state_variables()
    np.testing.assert_array_equal(torch.rand(4).numpy(), g["probe_after"], err_msg="host generator position")


def test_conv_mnist_training_graph_on_the_host_matches_reference():
    """The graph examples/mnist/conv_mnist.py trains -- Input -> Conv2dConnection(PostPre, norm, wmax) -> DiehlAndCookNodes(25, 4, 4)
    with lateral inhibition between the filters (dense recurrent Connection), one_spike arbitration from the host generator --
    at the script's sizes and batch 1, two consecutive inputs: rasters, theta and the generator position exact, voltages and
    weights within the convolution's tolerance (oneDNN / BLAS order in the reference)."""
    net, mons, cc = _conv_mnist_net()
    check_conv_mnist_graph(net, mons, cc)


def clamp_index_runs(dev="cpu"):
    """supervised_mnist.py's way of clamping -- an integer tensor of neuron INDICES -- against the reference fixture."""
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    from cases import check_packed
    g = gold("run_clamp_indices")
    N, B, T4 = 100, 2, 60
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    feat = net.connections[("X", "Ae")].pipeline[0]
    feat.value.data.copy_(T_(synth.weights_q12(10, 784, N)))
    mons = {l: Monitor(net.layers[l], ["s"], time=T4) for l in ("Ae", "Ai")}
    for l, m in mons.items():
        net.add_monitor(m, l + "_s")
    if dev != "cpu":
        net.to(dev)
    specs = [dict(clamp={"Ae": torch.tensor([37])}), dict(clamp={"Ae": torch.tensor([3, 64, 99])}, unclamp={"Ae": torch.tensor([5, 6, 7, 64])}),
             dict(clamp={"Ae": T_(np.stack([np.array([t % N, (7 * t + 3) % N]) for t in range(T4)]))})]
    for r, kw in enumerate(specs):
        sp = synth.spike_train(3500 + r, T4, B, 784, max_rate=0.25)
        torch.manual_seed(2 + r)
        net.run({"X": T_(sp).view(T4, B, 1, 28, 28).to(dev)}, time=T4, **kw)
        for l in ("Ae", "Ai"):
            got = np.packbits(mons[l].get("s").cpu().numpy().astype(u8))
            np.testing.assert_array_equal(got, g[f"r{r}_s{'E' if l == 'Ae' else 'I'}"], err_msg=f"run {r} {l} raster")
        check_packed(g, f"r{r}_W", feat.value.detach().cpu().numpy())
        np.testing.assert_array_equal(bits(net.layers["Ae"].theta.cpu().numpy()), bits(g[f"r{r}_theta"]), err_msg=f"run {r} theta")
        net.reset_state_variables()
    np.testing.assert_array_equal(torch.rand(4).numpy(), g["probe_after"], err_msg="host generator position")


def test_index_tensor_clamps_on_the_host_match_reference():
    clamp_index_runs()


def guide_example_check(dev="cpu"):
    """docs/source/guide/guide_part_i.rst, "Running Simulations": the user guide's end-to-end example (Input(100) -> Connection ->
    LIFNodes(1000) with a competitive recurrent Connection, spike and voltage monitors, a 2-D [time, n] input) through the mirror,
    against what the reference produced from the same statements."""
    import os
    import types
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    mod = types.SimpleNamespace(Network=Network, Input=Input, LIFNodes=LIFNodes, Connection=Connection, Monitor=Monitor)
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden_r3.py")).read()
    ns = {"torch": torch}
    exec(src[src.index("def guide_example("):src.index("def guide_case(")], ns)
    net, sm, tm, x = ns["guide_example"](mod)
    g = gold("run_guide_example")
    np.testing.assert_array_equal(np.packbits(x.numpy()), g["x"], err_msg="the example's own draws (torch.randn / torch.bernoulli)")
    if dev != "cpu":
        net.to(dev)
        x = x.to(dev)
    net.run(inputs={"A": x}, time=500)
    sA, sB, v = sm.get("s"), tm.get("s"), tm.get("v")
    assert [list(sA.shape), list(sB.shape), list(v.shape)] == g["shapes"].tolist()
    np.testing.assert_array_equal(np.packbits(sA.cpu().numpy().astype(u8)), g["sA"])
    got = np.packbits(sB.cpu().numpy().astype(u8))
    # the reference's propagation is an MKL gemv here (order not reproducible, SURVEY finding 5): voltages within its tolerance;
    # a spike flips only where a membrane potential sits within that tolerance of the threshold
    vv = v.cpu().numpy()
    np.testing.assert_allclose(vv.reshape(-1)[::997], g["v_sample"], rtol=0, atol=2e-3)
    flips = int(np.unpackbits(got ^ g["sB"]).sum())
    assert flips <= 4, flips
    return flips


def test_user_guide_example_on_the_host_matches_reference():
    assert guide_example_check() == 0          # same torch, same MKL on the host: identical here


def _r4_chain(dev="cpu"):
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import MulticompartmentConnection
    from bindsnet_amd.network.topology_features import Weight
    nX, nA, nB, nC, T = 64, 40, 24, 16, 30
    net = Network(dt=1.0, learning=False)
    net.add_layer(Input(n=nX), "X"); net.add_layer(LIFNodes(n=nA, thresh=-60.0), "A")
    net.add_layer(LIFNodes(n=nC, thresh=-58.0), "C"); net.add_layer(LIFNodes(n=nB, thresh=-61.0), "B")
    for k, (src, dst, ns, nd, sc) in enumerate((("X", "A", nX, nA, 0.6), ("A", "B", nA, nB, 0.5), ("B", "A", nB, nA, -1.0), ("C", "B", nC, nB, 1.0))):
        w = (synth.uniform_f32(3400 + k, (ns, nd), 0.0, abs(sc)) * np.sign(sc)).astype(np.float32)
        net.add_connection(MulticompartmentConnection(net.layers[src], net.layers[dst], device="cpu", pipeline=[Weight("weight", T_(w).clone())]), src, dst)
    mons = {l: Monitor(net.layers[l], ["s", "v"], time=T) for l in ("A", "B", "C")}
    for l, m in mons.items():
        net.add_monitor(m, l)
    if dev != "cpu":
        net.to(dev)
    return net, mons


def one_step_ext_current_runs(dev="cpu"):
    """run(..., one_step=True) with external currents into non-Input layers: the reference REPLACES the current of every layer a
    connection feeds by that layer's own `_get_inputs` (network.py:386-393): A's and B's are dropped, C's (no incoming connection)
    survives; the synchronous run of the same inputs adds all three.  Rasters and membrane potentials bit for bit
    (tests/golden/make_golden_r4.py).  Shared by the host test and the MI355X test."""
    g = gold("run_one_step_ext_current")
    nX, nA, nB, nC, B, T = 64, 40, 24, 16, 3, 30
    net, mons = _r4_chain(dev)
    sp = synth.dense_spikes(3410, (T, B, nX), 0.12)
    cur = {"A": synth.uniform_f32(3411, (T, B, nA), -1.0, 2.0), "B": synth.uniform_f32(3412, (T, B, nB), 0.0, 1.5),
           "C": synth.uniform_f32(3413, (T, B, nC), 0.0, 3.0)}
    for tag, flag in (("one", True), ("sync", False)):
        net.reset_state_variables()
        net.run({"X": T_(sp).to(dev), **{k: T_(v).to(dev) for k, v in cur.items()}}, time=T, one_step=flag)
        for l, n in (("A", nA), ("B", nB), ("C", nC)):
            np.testing.assert_array_equal(mons[l].get("s").cpu().numpy().reshape(T, B, n).astype(u8), unpack(g[f"{tag}_s_{l}"], (T, B, n)), err_msg=f"{tag} raster {l}")
            np.testing.assert_array_equal(bits(mons[l].get("v").cpu().numpy().reshape(T, B, n)), bits(g[f"{tag}_v_{l}"].reshape(T, B, n)), err_msg=f"{tag} v {l}")


def test_one_step_drops_the_external_current_of_fed_layers_like_the_reference():
    one_step_ext_current_runs("cpu")


def test_per_connection_a_plus_tables_are_refused_explicitly():
    """network.py:356-378 also accepts {connection: value} tables for a_plus / a_minus; neither path of this package does, and
    both say so (the host path used to fail inside torch.tensor(dict))."""
    net, _ = _r4_chain()
    with pytest.raises(NotImplementedError, match="a_plus/a_minus dicts"):
        net.run({"X": torch.zeros(2, 3, 64, dtype=torch.uint8)}, time=2, a_plus={("X", "A"): 1.0})


def test_encoder_thread_cap_is_reentrant_and_skips_worker_threads():
    """encodings._few_threads changes a PROCESS-global setting: nested calls restore what the outermost one found, and a call from a
    thread other than the main one leaves the setting alone."""
    import threading
    from bindsnet_amd.encoding import encodings as E
    n0 = torch.get_num_threads()
    try:
        torch.set_num_threads(8)
        with E._few_threads():
            inner = torch.get_num_threads()
            with E._few_threads():
                assert torch.get_num_threads() == inner
            assert torch.get_num_threads() == inner
        assert inner == min(8, E._ENCODER_THREADS) and torch.get_num_threads() == 8
        seen = []
        def worker():
            with E._few_threads():
                seen.append(torch.get_num_threads())
        th = threading.Thread(target=worker); th.start(); th.join()
        assert torch.get_num_threads() == 8 and E._cap_depth == 0
    finally:
        torch.set_num_threads(n0)
