"""Fused plan "twolayer-fused" (one launch per run for Input -> Connection/MCC -> LIF) vs the generic
per-operator plan, bit for bit, over consecutive inputs; plus plan selection."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(kind, Nin, N, rule, bias):
    from bindsnet_amd.learning import PostPre
    from bindsnet_amd.learning.MCC_learning import PostPre as MCCPostPre
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection, MulticompartmentConnection
    from bindsnet_amd.network.topology_features import Weight
    torch.manual_seed(0)
    net = Network(dt=1.0)
    net.add_layer(Input(n=Nin, traces=True), "X")
    net.add_layer(LIFNodes(n=N, traces=True, lbound=-70.0 if bias else None), "Y")
    W0 = torch.from_numpy(synth.weights_q12(11, Nin, N))
    if kind == "dense":
        kw = dict(update_rule=PostPre, nu=(1e-4, 1e-2), reduction=torch.sum) if rule else {}
        b = torch.from_numpy(synth.uniform_f32(5, (N,), -0.5, 0.5)) if bias else None
        conn = Connection(net.layers["X"], net.layers["Y"], w=W0.clone(), b=b, wmin=0.0, wmax=1.0, norm=78.4 * Nin / 784, **kw)
    else:
        f = Weight("weight", W0.clone(), range=[0.0, 1.0], norm=78.4 * Nin / 784, nu=(1e-4, 1e-2) if rule else None,
                   learning_rule=MCCPostPre if rule else None, reduction=torch.sum if rule else None)
        conn = MulticompartmentConnection(net.layers["X"], net.layers["Y"], device="cpu", pipeline=[f])
    net.add_connection(conn, "X", "Y")
    return net


def run(generic, kind, Nin, N, B, T, rule=True, bias=False, n_inputs=2, dens=0.03, learning=True, additive=False):
    from bindsnet_amd import _lib
    from bindsnet_amd.network.monitors import Monitor
    _lib.lib().snn_set_plan_mode(1 if generic else 0)
    try:
        net = build(kind, Nin, N, rule, bias)
        ms, mv = Monitor(net.layers["Y"], ["s"], time=T), Monitor(net.layers["Y"], ["v"], time=T)
        net.add_monitor(ms, "s"); net.add_monitor(mv, "v")
        net.train(learning)
        if additive:                                 # nodes.py:96-103: additive instead of replacing traces
            for l in net.layers.values():
                l.traces_additive = True
                l.trace_scale.fill_(0.5)
        net.to(DEV)
        out = []
        for r in range(n_inputs):
            sp = synth.dense_spikes(40 + r, (T, B, Nin), dens)
            net.run({"X": torch.from_numpy(sp).to(DEV)}, time=T)
            conn = net.connections[("X", "Y")]
            W = (conn.w if kind == "dense" else conn.pipeline[0].value).detach().cpu().numpy().copy()
            out.append(dict(s=ms.get("s").cpu().numpy().copy(), v=mv.get("v").cpu().numpy().copy(), W=W,
                            vY=net.layers["Y"].v.cpu().numpy().copy(), rY=net.layers["Y"].refrac_count.cpu().numpy().copy(),
                            xY=net.layers["Y"].x.cpu().numpy().copy(), xX=net.layers["X"].x.cpu().numpy().copy(),
                            sY=net.layers["Y"].s.cpu().numpy().copy()))
            plan = net.last_plan
            if r == 0:
                net.reset_state_variables()
        return out, plan
    finally:
        _lib.lib().snn_set_plan_mode(0)


CASES = {
    "dense_postpre_b16": ("dense", 784, 200, 16, 40, True, False),
    "dense_postpre_b32_bias": ("dense", 784, 96, 32, 30, True, True),
    "dense_postpre_b1": ("dense", 256, 40, 1, 50, True, False),
    "dense_norule_b8": ("dense", 784, 64, 8, 30, False, True),
    "mcc_postpre_b5_tailcols": ("mcc", 784, 100, 5, 40, True, False),
    "mcc_postpre_b32_n37": ("mcc", 400, 37, 32, 25, True, False),
    "mcc_norule_b3": ("mcc", 784, 50, 3, 30, False, False),
    "dense_wide_input_b16": ("dense", 6400, 64, 16, 12, True, False),
    # batch > 32: sample masks several words wide (the reference's batch sums cross 16-sample cascade blocks)
    "dense_postpre_b48": ("dense", 784, 96, 48, 25, True, False),
    "dense_postpre_b128_cfg3_slice": ("dense", 784, 160, 128, 20, True, False),
    "dense_postpre_b33_bias": ("dense", 256, 40, 33, 30, True, True),
    "mcc_postpre_b64_tailcols": ("mcc", 784, 100, 64, 20, True, False),
    "mcc_postpre_b100_n37": ("mcc", 400, 37, 100, 15, True, False),
    "dense_norule_b96": ("dense", 784, 64, 96, 15, False, True),
}


@pytest.mark.parametrize("name", list(CASES))
def test_twolayer_fused_equals_generic(name):
    kind, Nin, N, B, T, rule, bias = CASES[name]
    dens = 0.05 if Nin <= 784 else 0.02
    f, plan = run(False, kind, Nin, N, B, T, rule, bias, dens=dens)
    assert plan == "twolayer-fused"
    g, plan_g = run(True, kind, Nin, N, B, T, rule, bias, dens=dens)
    assert plan_g == "generic"
    for r, (a, b) in enumerate(zip(f, g)):
        for k in a:
            np.testing.assert_array_equal(a[k].view(np.uint8), b[k].view(np.uint8), err_msg=f"{name} input {r}: {k}")
    assert sum(int(x["s"].sum()) for x in f) > 0, "silent network: vacuous"


def test_twolayer_learning_off_and_big_batch_fallback():
    f, plan = run(False, "dense", 784, 64, 8, 25, learning=False)
    g, _ = run(True, "dense", 784, 64, 8, 25, learning=False)
    assert plan == "twolayer-fused"
    for a, b in zip(f, g):
        for k in a:
            np.testing.assert_array_equal(a[k].view(np.uint8), b[k].view(np.uint8))
    _, plan = run(False, "dense", 784, 64, 130, 5)      # batch > 128: generic plan
    assert plan == "generic"


# ------------------------------------------------------------------------------------------------ MSTDP
def run_mstdp(generic, Nin, N, B, T, reward, n_inputs=3, dens=0.05, vmax=1, learning=True):
    """Input -> Connection(MSTDP) -> LIF (the cfg5 graph): fused plan vs generic plan, incl. the rule's state."""
    from bindsnet_amd import _lib
    from bindsnet_amd.learning import MSTDP
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    _lib.lib().snn_set_plan_mode(1 if generic else 0)
    try:
        torch.manual_seed(0)
        net = Network(dt=1.0)
        net.add_layer(Input(n=Nin, traces=True), "X")
        net.add_layer(LIFNodes(n=N, traces=True), "Y")
        W0 = torch.from_numpy(synth.weights_q12(11, Nin, N))
        conn = Connection(net.layers["X"], net.layers["Y"], w=W0.clone(), wmin=0.0, wmax=1.0, update_rule=MSTDP, nu=1e-1,
                          norm=0.5 * Nin * 0.1, reduction=torch.sum)
        net.add_connection(conn, "X", "Y")
        ms = Monitor(net.layers["Y"], ["s"], time=T)
        net.add_monitor(ms, "s")
        net.train(learning)
        net.to(DEV)
        out = []
        rs = np.random.RandomState(9)
        for r in range(n_inputs):
            sp = synth.dense_spikes(140 + r, (T, B, Nin), dens)
            if vmax > 1:
                sp = (sp * rs.randint(1, vmax + 1, size=sp.shape)).astype(np.uint8)
            rw = reward if not isinstance(reward, str) else torch.from_numpy(synth.uniform_f32(17 + r, (B,), -1.0, 1.0)).to(DEV)
            net.run({"X": torch.from_numpy(sp).to(DEV)}, time=T, reward=rw)
            rule = conn.update_rule
            st = dict(s=ms.get("s").cpu().numpy().copy(), W=conn.w.detach().cpu().numpy().copy(),
                      vY=net.layers["Y"].v.cpu().numpy().copy(), xX=net.layers["X"].x.cpu().numpy().copy(),
                      xY=net.layers["Y"].x.cpu().numpy().copy())
            if learning:
                st.update(pp=rule.p_plus.cpu().numpy().copy(), pm=rule.p_minus.cpu().numpy().copy(),
                          ss=rule._s_src_prev.cpu().numpy().copy(), st=rule._s_tgt_prev.cpu().numpy().copy())
            out.append(st)
            plan = net.last_plan
            if r == 0:
                net.reset_state_variables()           # X.s is zeroed, the rule's memory is not
        return out, plan
    finally:
        _lib.lib().snn_set_plan_mode(0)


MSTDP_CASES = {
    # name: (Nin, N, B, T, reward, density, value_max)
    "b16_scalar_reward": (784, 100, 16, 40, 1.0, 0.06, 1),
    "b32_negative_reward": (784, 64, 32, 30, -0.5, 0.06, 1),
    "b5_reward_vector_tailcols": (400, 37, 5, 40, "vec", 0.08, 1),
    "b1": (256, 40, 1, 50, 1.0, 0.08, 1),
    "multivalued_source_bytes": (784, 48, 8, 30, 1.0, 0.05, 3),
    "cfg5_shape_short": (6400, 500, 16, 8, 1.0, 0.05, 1),
    "b48_negative_reward": (784, 64, 48, 25, -0.5, 0.05, 1),
    "b100_reward_vector": (400, 40, 100, 20, "vec", 0.06, 1),
    # round 6: the sample-major burst update (batch <= 16, tiles of 2 / 4 columns) and the digest through LDS-DMA
    "wide_tile4_b12_reward_vector": (3200, 64, 12, 30, "vec", 0.05, 1),     # 4 columns per workgroup, 4 of 7 row slots, samples past the batch
    "wide_tile2_b16_long": (6400, 24, 16, 30, -0.5, 0.05, 1),               # 2 columns per workgroup, all 7 row slots, many burst steps
    "wide_tile4_b3": (4096, 20, 3, 40, 1.0, 0.04, 1),
}


@pytest.mark.parametrize("name", list(MSTDP_CASES))
def test_twolayer_mstdp_fused_equals_generic(name):
    Nin, N, B, T, reward, dens, vmax = MSTDP_CASES[name]
    f, plan = run_mstdp(False, Nin, N, B, T, reward, dens=dens, vmax=vmax)
    assert plan == "twolayer-fused"
    g, plan_g = run_mstdp(True, Nin, N, B, T, reward, dens=dens, vmax=vmax)
    assert plan_g == "generic"
    for r, (a, b) in enumerate(zip(f, g)):
        for k in a:
            np.testing.assert_array_equal(a[k].view(np.uint8), b[k].view(np.uint8), err_msg=f"{name} input {r}: {k}")
    assert sum(int(x["s"].sum()) for x in f) > 0, "silent network: vacuous"
    assert np.abs(f[-1]["pm"]).max() > 0 and not np.array_equal(f[0]["W"], f[-1]["W"])


@pytest.mark.parametrize("name", ["b16_scalar_reward", "wide_tile4_b12_reward_vector", "wide_tile2_b16_long", "cfg5_shape_short"])
def test_twolayer_mstdp_burst_forms_agree(name, monkeypatch):
    """The burst update (a step in which a column of the tile spiked) exists in three forms -- sample-major (default where it applies), the row
    walk in its packed forms, the row walk as round 5 left it (developer switch SNN_TWO_MSTDP_BURST = 2 / 1 / 0): same weights, same state."""
    Nin, N, B, T, reward, dens, vmax = MSTDP_CASES[name]
    outs = []
    for form in ("2", "1", "0"):
        monkeypatch.setenv("SNN_TWO_MSTDP_BURST", form)
        f, plan = run_mstdp(False, Nin, N, B, T, reward, n_inputs=2, dens=dens, vmax=vmax)
        assert plan == "twolayer-fused"
        outs.append(f)
    for other in outs[1:]:
        for r, (a, b) in enumerate(zip(outs[0], other)):
            for k in a:
                np.testing.assert_array_equal(a[k].view(np.uint8), b[k].view(np.uint8), err_msg=f"{name} input {r}: {k}")
    assert sum(int(x["s"].sum()) for x in outs[0]) > 0


def test_twolayer_mstdp_learning_off():
    f, plan = run_mstdp(False, 784, 64, 8, 25, 1.0, n_inputs=2, learning=False)
    g, _ = run_mstdp(True, 784, 64, 8, 25, 1.0, n_inputs=2, learning=False)
    assert plan == "twolayer-fused"
    for a, b in zip(f, g):
        for k in a:
            np.testing.assert_array_equal(a[k].view(np.uint8), b[k].view(np.uint8))


# ------------------------------------------------------------------------------------------------ MCC MSTDP
@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("name", ["run_two_mcc_mstdp_b4", "run_two_mcc_mstdp_b20", "run_two_mcc_mstdp_n208"])
def test_mcc_mstdp_matches_reference(name, generic):
    """MulticompartmentConnection + Weight with MCC_learning.MSTDP (MCC_learning.py:392-551): two consecutive runs
    (scalar reward, then per-sample rewards) bit for bit against the reference fixture, on both plans."""
    import cases
    from cases import gold, unpack
    from bindsnet_amd import _lib
    from bindsnet_amd.learning.MCC_learning import MSTDP
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import MulticompartmentConnection
    from bindsnet_amd.network.topology_features import Weight
    g = gold(name)
    Nin, N, B, T = int(g["Nin"]), int(g["N"]), int(g["B"]), int(g["T"])
    _lib.lib().snn_set_plan_mode(1 if generic else 0)
    try:
        net = Network(dt=1.0)
        X_, Y_ = Input(n=Nin, traces=True), LIFNodes(n=N, traces=True)
        feat = Weight("weight", torch.from_numpy(synth.weights_q12(11, Nin, N)), range=[0.0, 1.0], norm=0.1 * Nin, nu=(1e-1, 1e-1),
                      learning_rule=MSTDP)
        conn = MulticompartmentConnection(X_, Y_, device="cpu", pipeline=[feat])
        net.add_layer(X_, "X"); net.add_layer(Y_, "Y")
        net.add_connection(conn, "X", "Y")
        mon = Monitor(net.layers["Y"], ["s"], time=T)
        net.add_monitor(mon, "Y_s")
        net.to(DEV)
        for r in range(2):
            spikes = synth.spike_train(30 + r, T, B, Nin, active=0.3, max_rate=0.12)
            reward = 1.0 if r == 0 else torch.from_numpy(synth.uniform_f32(17, (B,), -1.0, 1.0)).view(B, 1, 1)
            net.run({"X": torch.from_numpy(spikes).to(DEV)}, time=T, reward=reward)
            assert net.last_plan == ("generic" if generic or Nin % 16 else "twolayer-fused")   # (the fused plan wants Nin % 16 == 0)
            rule = feat.learning_rule
            np.testing.assert_array_equal(mon.get("s").cpu().numpy().reshape(T, B, N).astype(np.uint8), unpack(g[f"r{r}_sY"], (T, B, N)))
            for got, key in ((feat.value, "W"), (net.layers["Y"].v, "vY"), (rule.p_plus, "p_plus"), (rule.p_minus, "p_minus")):
                np.testing.assert_array_equal(got.detach().cpu().numpy().view(np.uint32), g[f"r{r}_{key}"].view(np.uint32),
                                              err_msg=f"run {r} {key}")
            assert cases.sha(rule.eligibility.cpu().numpy()) == str(g[f"r{r}_elig_sha"])
            net.reset_state_variables()
    finally:
        _lib.lib().snn_set_plan_mode(0)


@pytest.mark.parametrize("kind", ["dense", "mcc"])
def test_additive_traces_fused_equals_generic(kind):
    """`traces_additive=True` on both layers: the fused plan's input-trace pre-pass, target trace and PostPre against the
    generic plan, bit for bit."""
    f, plan = run(False, kind, 784, 96, 16, 30, additive=True)
    g, plan_g = run(True, kind, 784, 96, 16, 30, additive=True)
    assert plan_g == "generic" and plan == "twolayer-fused"
    p, _ = run(True, kind, 784, 96, 16, 30)
    assert not np.array_equal(g[0]["xX"], p[0]["xX"]) and g[0]["s"].sum() > 0
    for a, b in zip(f, g):
        for k in a:
            np.testing.assert_array_equal(a[k].view(np.uint8), b[k].view(np.uint8), err_msg=k)
