"""Every BASELINE.json config at its STATED size, through the public API, on the plans a user gets by default,
against fixtures the unmodified reference produced (tests/golden/make_golden_full.py):

  cfg1  DiehlAndCook2015 784->100, B=1,  T=250, 3 inputs, reset between   bit-exact (rasters, W, theta, state, RNG)
  cfg2  DiehlAndCook2015 784->400, B=32, T=250, 3 inputs, reset between   bit-exact -- the bench.py workload itself
  cfg3  TwoLayerNetwork 784->1600 PostPre, T=100, B=16 / 32 / 128         rasters identical, weights <= 1e-5 (MKL order)
  cfg4  Conv2d 5x5x32 -> LIF, B=64, T=250, no learning                    bit-exact (raster sha256, v, refrac)
  cfg5  Input 6400 -> Connection(MSTDP) -> LIF 500, B=16, T=20            rasters identical, weights <= 1e-5, rule state exact

cfg3 / cfg5 are additionally compared bit for bit with the order-pinned oracle on the fixture's column subset.
"""
import numpy as np
import pytest
import torch

import cases
import oracle
import synth
from cases import check_packed, gold, u8, unpack
from test_oracle_fullsize import two_case, two_state_cols

pytestmark = pytest.mark.gpu
DEV = "cuda"
PLAN_MODE = {"auto": 0, "generic": 1, "per-step": 2, "resident": 3}


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("plan", ["auto", "resident", "per-step", "generic"])
@pytest.mark.parametrize("name", cases.DC_FULL)
def test_dc2015_full_size_matches_reference(name, plan):
    """`*_poisson`: the input BASELINE.md section 2 states (reference-encoded Poisson trains, ~1.17 % density) = what
    bench.py times.  `*_bold`: its second input is saturated digits (up to 101 events per sample-step, > the lean
    kernel's 32): the default plan must hand that input to the general resident form (SNN_ERR_RETRY, nothing written),
    stay there for the third input (cool-down) and still reproduce the REFERENCE bit for bit."""
    from bindsnet_amd import _lib
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    g = gold(name)
    N, B, T, runs = int(g["N"]), int(g["B"]), int(g["T"]), int(g["runs"])
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05,
                           inpt_shape=(1, 28, 28))
    feat = net.connections[("X", "Ae")].pipeline[0]
    assert cases.sha(host(feat.value)) == str(g["W0_sha"]), "construction draws differ from the reference"
    mons = {}
    for l in ("X", "Ae", "Ai"):
        mons[l] = Monitor(net.layers[l], ["s"], time=T)
        net.add_monitor(mons[l], l + "_s")
    net.to(DEV)
    _lib.lib().snn_set_plan_mode(PLAN_MODE[plan])
    try:
        torch.manual_seed(2)
        for r in range(runs):
            spikes = cases.fixture_input(g, r, T, B)
            net.run({"X": torch.from_numpy(spikes).view(T, B, 1, 28, 28).to(DEV)}, time=T)
            if name.endswith("_bold") and plan == "auto":
                assert net.last_plan == ("dc2015-resident-lean" if r == 0 else "dc2015-resident"), (r, net.last_plan)
                assert getattr(net, "lean_retries", 0) == (0 if r == 0 else 1)
            sE = host(mons["Ae"].get("s")).reshape(T, B, N).astype(u8)
            sI = host(mons["Ai"].get("s")).reshape(T, B, N).astype(u8)
            np.testing.assert_array_equal(sE, unpack(g[f"r{r}_sE"], (T, B, N)), err_msg=f"run {r} Ae raster")
            np.testing.assert_array_equal(sI, unpack(g[f"r{r}_sI"], (T, B, N)), err_msg=f"run {r} Ai raster")
            np.testing.assert_array_equal(host(mons["X"].get("s")).reshape(T, B, 784), spikes)
            assert cases.sha(host(feat.value)) == str(g[f"r{r}_W_sha"]), f"run {r} weights"
            Ae, Ai, X = net.layers["Ae"], net.layers["Ai"], net.layers["X"]
            np.testing.assert_array_equal(bits(host(Ae.theta)), bits(g[f"r{r}_theta"]), err_msg=f"run {r} theta")
            for key, a in (("vE", Ae.v), ("rE", Ae.refrac_count), ("xE", Ae.x), ("xX", X.x.reshape(B, 784)),
                           ("vI", Ai.v), ("rI", Ai.refrac_count)):
                check_packed(g, f"r{r}_{key}", host(a))
            net.reset_state_variables()
        np.testing.assert_array_equal(torch.rand(4).numpy(), g["probe_after"], err_msg="host generator position")
        bounced = name.endswith("_bold") and plan == "auto"
        assert net.last_plan == {"auto": "dc2015-resident" if bounced else "dc2015-resident-lean", "resident": "dc2015-resident",
                                 "per-step": "dc2015-fused", "generic": "generic"}[plan]
        assert getattr(net, "lean_retries", 0) == int(bounced) and getattr(net, "resident_retries", 0) == 0
    finally:
        _lib.lib().snn_set_plan_mode(0)


@pytest.mark.parametrize("name,rule", [("full_cfg3_two_b16", "postpre"), ("full_cfg3_two_b32", "postpre"),
                                       ("full_cfg3_two_b128", "postpre"), ("full_cfg5_mstdp_b16", "mstdp")])
def test_dense_family_full_size_matches_reference(name, rule):
    from bindsnet_amd.learning import MSTDP
    from bindsnet_amd.models import TwoLayerNetwork
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    g = gold(name)
    P, spikes, W0 = two_case(g, rule)
    Nin, N, B, T = P.Nin, P.N, P.B, P.T
    torch.manual_seed(0)
    if rule == "postpre":
        net = TwoLayerNetwork(n_inpt=Nin, n_neurons=N, reduction=torch.sum)
        conn = net.connections[("X", "Y")]
        x = torch.from_numpy(spikes)
        kw = {}
    else:
        net = Network(dt=1.0)
        net.add_layer(Input(n=Nin, shape=(1, 80, 80), traces=True), "X")
        net.add_layer(LIFNodes(n=N, traces=True), "Y")
        conn = Connection(net.layers["X"], net.layers["Y"], wmin=0, wmax=1, update_rule=MSTDP, nu=1e-1,
                          norm=0.5 * Nin, reduction=torch.sum)
        net.add_connection(conn, "X", "Y")
        x = torch.from_numpy(spikes).view(T, B, 1, 80, 80)
        kw = {"reward": 1.0}
    assert cases.sha(host(conn.w)) == str(g["W0_sha"]), "construction draws differ from the reference"
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    net.to(DEV)
    net.run({"X": x.to(DEV)}, time=T, **kw)
    ras = host(mon.get("s")).reshape(T, B, N).astype(u8)
    W = host(conn.w)
    # --- vs the reference (MKL propagation): rasters identical, weights within 1e-5 (north star)
    np.testing.assert_array_equal(ras, unpack(g["sY"], (T, B, N)))
    cols = g["cols"].astype(np.int64)
    np.testing.assert_allclose(W.reshape(-1)[::max(1, W.size // 8192)], g["W_sample"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(W[:, cols], g["W_cols"], rtol=0, atol=1e-5)
    check_packed(g, "xX", host(net.layers["X"].x).reshape(B, Nin))
    check_packed(g, "xY", host(net.layers["Y"].x))
    check_packed(g, "rY", host(net.layers["Y"].refrac_count))
    if rule == "mstdp":
        ur = conn.update_rule
        check_packed(g, "p_plus", host(ur.p_plus))
        np.testing.assert_array_equal(bits(host(ur.p_minus)), bits(g["p_minus"]))
        assert cases.sha(host(ur.eligibility)) == str(g["elig_sha"])
    # --- vs the order-pinned oracle on the column subset: bit for bit (target neurons are independent)
    P.N = len(cols)
    st = two_state_cols(P, W0[:, cols])
    ras_o = oracle.run_two_layer(P, st, spikes)
    np.testing.assert_array_equal(ras[:, :, cols], ras_o)
    np.testing.assert_array_equal(bits(W[:, cols]), bits(st["W"]))
    np.testing.assert_array_equal(bits(host(net.layers["Y"].v)[:, cols]), bits(st["vY"]))
    assert net.last_plan == "twolayer-fused", "every BASELINE dense-family config runs as one fused launch (batch <= 128)"
    print(f"{name}: plan {net.last_plan}; weights bit-identical to the reference: {cases.sha(W) == str(g['W_sha'])}")


def test_conv_lif_full_size_matches_reference():
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Conv2dConnection
    g = gold("full_cfg4_conv_b64")
    B, T = int(g["B"]), int(g["T"])
    torch.manual_seed(0)
    net = Network(dt=1.0, learning=False)
    net.add_layer(Input(shape=(1, 28, 28)), "X")
    net.add_layer(LIFNodes(shape=(32, 24, 24)), "Y")
    w = 0.3 * torch.rand(32, 1, 5, 5)
    assert cases.sha(w.numpy()) == str(g["W0_sha"])
    net.add_connection(Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=5, stride=1, w=w), "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    net.to(DEV)
    spikes = synth.dense_spikes(3, (T, B, 1, 28, 28), 0.05)
    net.run({"X": torch.from_numpy(spikes).to(DEV)}, time=T)
    assert net.last_plan == "convlif-fused"
    s = mon.get("s")
    assert tuple(s.shape) == (T, B, 32, 24, 24)
    np.testing.assert_array_equal(host(s.reshape(T, -1).sum(1)), g["sY_per_step"])
    s = host(s).astype(u8)
    np.testing.assert_array_equal(s.sum(axis=(0, 2, 3, 4)), g["sY_per_sample"])
    np.testing.assert_array_equal(s.sum(axis=(0, 1, 3, 4)), g["sY_per_channel"])
    assert cases.sha(np.packbits(s)) == str(g["sY_sha"]), "raster sha256"
    Y = net.layers["Y"]
    assert cases.sha(host(Y.v)) == str(g["v_sha"]) and cases.sha(host(Y.refrac_count)) == str(g["r_sha"])
