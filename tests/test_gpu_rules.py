"""Dense learning rules beyond PostPre / MSTDP: Hebbian, WeightDependentPostPre, MSTDPET.
 * single update() calls through the rule classes against the reference's own results (tests/golden/op_rules.npz);
 * whole Network.run() calls (generic plan: snn_stdp_hebbian / snn_mstdpet_step every timestep) against the oracle's
   run driver, which tests/test_oracle_golden.py pins to the same fixtures."""
import numpy as np
import pytest
import torch

import oracle
import synth
from cases import f32, u8, gold
from test_oracle_golden import RULE_VARIANTS, rule_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def host(t):
    return t.detach().cpu().numpy()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def layers(B, Nin, N, s_src, x_src, s_tgt, x_tgt):
    from bindsnet_amd.network.nodes import Input, LIFNodes
    src, tgt = Input(n=Nin, traces=True), LIFNodes(n=N, traces=True)
    for l, s, x in ((src, s_src, x_src), (tgt, s_tgt, x_tgt)):
        l.batch_size = B
        l.s = dev(s)
        if x is not None:
            l.x = dev(x)
    return src, tgt


@pytest.mark.parametrize("tag", list(RULE_VARIANTS))
def test_hebbian_and_weight_dependent_postpre_update_vs_reference(tag):
    from bindsnet_amd.learning import Hebbian, WeightDependentPostPre
    from bindsnet_amd.network.topology import Connection
    g = gold("op_rules")
    wd, decay, lo, hi = RULE_VARIANTS[tag]
    for k, (B, Nin, N) in enumerate(g["cases"]):
        B, Nin, N = int(B), int(Nin), int(N)
        W0, s_src, s_tgt, x_src, x_tgt = rule_inputs(k, B, Nin, N)
        src, tgt = layers(B, Nin, N, s_src, x_src, s_tgt, x_tgt)
        kw = {} if lo is None else dict(wmin=lo, wmax=hi)
        if decay != 1.0:
            kw["weight_decay"] = 0.01
        conn = Connection(src, tgt, w=torch.from_numpy(W0).clone(), update_rule=WeightDependentPostPre if wd else Hebbian,
                          nu=(1e-2, 3e-2), reduction=torch.sum, **kw).to(DEV)
        conn.update(learning=True)
        np.testing.assert_array_equal(bits(host(conn.w)), bits(g[f"{tag}{k}"]), err_msg=f"{tag} case {k}")


def test_mstdpet_update_sequence_vs_reference():
    from bindsnet_amd.learning import MSTDPET
    from bindsnet_amd.network.topology import Connection
    g = gold("op_rules")
    Nin, N, T = 36, 20, 12
    src, tgt = layers(1, Nin, N, np.zeros((1, Nin), u8), None, np.zeros((1, N), u8), None)
    conn = Connection(src, tgt, w=torch.from_numpy(synth.uniform_f32(900, (Nin, N), 0.0, 1.0)), update_rule=MSTDPET, nu=(1e-1, 1e-1),
                      wmin=0.0, wmax=1.0, tc_e_trace=25.0).to(DEV)
    conn.dt = 1.0
    for t in range(T):
        src.s = dev(synth.dense_spikes(910 + t, (1, Nin), 0.2))
        tgt.s = dev(synth.dense_spikes(940 + t, (1, N), 0.2))
        conn.update(learning=True, reward=0.7 if t % 3 else -0.4)
    ur = conn.update_rule
    for got, key in ((conn.w, "et_w"), (ur.eligibility_trace, "et_trace"), (ur.eligibility, "et_elig"), (ur.p_plus, "et_p_plus"),
                     (ur.p_minus, "et_p_minus")):
        np.testing.assert_array_equal(bits(host(got)), bits(g[key]), err_msg=key)


@pytest.mark.parametrize("rule,Nin,N,B,plan", [
    ("hebbian", 196, 48, 6, "generic"), ("wdpp", 196, 48, 6, "generic"), ("mstdpet", 196, 48, 1, "generic"),
    # Nin % 16 == 0: the one-launch plan (k_two_run<false, HEBBIAN / WDPOSTPRE, .>), one and two batch-mask words
    ("hebbian", 208, 48, 6, "twolayer-fused"), ("wdpp", 208, 48, 6, "twolayer-fused"),
    ("hebbian", 208, 40, 40, "twolayer-fused"), ("wdpp", 208, 40, 40, "twolayer-fused")])
def test_network_run_with_the_rule_vs_oracle(rule, Nin, N, B, plan):
    _run_rule_vs_oracle(rule, Nin, N, B, plan)


@pytest.mark.parametrize("rule,N,B,nu", [("hebbian", 44, 6, (1e-4, 0.0)), ("hebbian", 44, 40, (0.0, 1e-3)), ("wdpp", 44, 6, (0.0, 1e-3)),
                                         ("wdpp", 44, 40, (1e-4, 0.0)), ("hebbian", 48, 70, (2e-4, 5e-4)), ("wdpp", 48, 70, (2e-4, 5e-4))])
def test_outer_rules_row_major_form_one_sided_rates_and_partial_tiles(rule, N, B, nu):
    """Round 5: Hebbian / WeightDependentPostPre take PostPre's row-major form in the one-launch plan (two_stdp_rowmajor<., RULE>).  One-sided
    learning rates (Hebbian visits the columns that spiked even with nu1 == 0; WeightDependentPostPre skips a side whose rate is zero), a last
    tile of 4 columns (N = 44), three batch-mask words (B = 70: the batch sums cross four 16-sample cascade blocks)."""
    _run_rule_vs_oracle(rule, 208, N, B, "twolayer-fused", nu=nu)


def _run_rule_vs_oracle(rule, Nin, N, B, plan, nu=None):
    from bindsnet_amd.learning import Hebbian, MSTDPET, WeightDependentPostPre
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    T = 40
    W0 = synth.weights_q12(11, Nin, N)
    net = Network(dt=1.0)
    net.add_layer(Input(n=Nin, traces=True), "X")
    net.add_layer(LIFNodes(n=N, traces=True), "Y")
    cls = {"hebbian": Hebbian, "wdpp": WeightDependentPostPre, "mstdpet": MSTDPET}[rule]
    if nu is None:
        nu = (1e-1, 1e-1) if rule == "mstdpet" else (1e-4, 1e-3)
    conn = Connection(net.layers["X"], net.layers["Y"], w=torch.from_numpy(W0).clone(), wmin=0.0, wmax=1.0, update_rule=cls, nu=nu,
                      norm=0.1 * Nin, reduction=torch.sum)
    net.add_connection(conn, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    net.to(DEV)
    spikes = synth.spike_train(30, T, B, Nin, active=0.3, max_rate=0.12)
    kw = {"reward": 0.8} if rule == "mstdpet" else {}
    net.run({"X": torch.from_numpy(spikes).to(DEV)}, time=T, **kw)
    assert net.last_plan == plan
    P = oracle.TwoParams()
    P.B, P.Nin, P.N, P.T, P.dt = B, Nin, N, T, 1.0
    P.rule = {"hebbian": 3, "wdpp": 4, "mstdpet": 5}[rule]
    P.x_trace_decay = float(net.layers["X"].trace_decay); P.x_trace_scale = 1.0; P.x_traces = 1
    P.decay = float(net.layers["Y"].decay); P.rest, P.reset, P.thresh, P.refrac = -65.0, -65.0, -52.0, 5.0
    P.y_traces = 1; P.y_trace_decay = float(net.layers["Y"].trace_decay); P.y_trace_scale = 1.0
    P.nu0, P.nu1 = nu
    P.has_min = P.has_max = 1; P.wmin, P.wmax = 0.0, 1.0; P.has_norm = 1; P.norm = 0.1 * Nin; P.learning = 1
    st = dict(W=W0.copy(), sX=np.zeros((B, Nin), u8), xX=np.zeros((B, Nin), f32), vY=np.full((B, N), -65.0, f32),
              rY=np.zeros((B, N), f32), sY=np.zeros((B, N), u8), xY=np.zeros((B, N), f32))
    if rule == "mstdpet":
        ur = conn.update_rule
        dp, dm, de = ur._decays()
        P.reward, P.a_plus, P.a_minus, P.decay_plus, P.decay_minus, P.decay_e, P.tc_e = 0.8, 1.0, -1.0, dp, dm, de, 25.0
        st.update(elig=np.zeros((Nin, N), f32), e_trace=np.zeros((Nin, N), f32), p_plus=np.zeros(Nin, f32), p_minus=np.zeros(N, f32))
    ras = oracle.run_two_layer(P, st, spikes)
    assert ras.sum() > 20
    np.testing.assert_array_equal(host(mon.get("s")).reshape(T, B, N).astype(u8), ras)
    np.testing.assert_array_equal(bits(host(conn.w)), bits(st["W"]))
    if rule == "mstdpet":
        np.testing.assert_array_equal(bits(host(conn.update_rule.eligibility_trace)), bits(st["e_trace"]))


def test_mcc_mstdpet_matches_reference():
    """MulticompartmentConnection + Weight with MCC_learning.MSTDPET (MCC_learning.py:554-733, batch 1): two consecutive
    Network.run() calls (different reward / a_plus, layers reset in between, the rule's state kept) bit for bit against
    the reference fixture, plus single update() calls through the rule object afterwards against the oracle."""
    from cases import unpack
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
    feat = Weight("weight", torch.from_numpy(synth.weights_q12(11, Nin, N)), range=[0.0, 1.0], norm=0.1 * Nin, nu=(1e-1, 1e-1),
                  learning_rule=MSTDPET)
    conn = MulticompartmentConnection(X_, Y_, device="cpu", pipeline=[feat], tc_e_trace=25.0)
    net.add_layer(X_, "X"); net.add_layer(Y_, "Y")
    net.add_connection(conn, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    net.to(DEV)
    rule = feat.learning_rule
    for r in range(2):
        spikes = synth.spike_train(30 + r, T, 1, Nin, active=0.3, max_rate=0.12)
        net.run({"X": torch.from_numpy(spikes).to(DEV)}, time=T, reward=0.8 if r == 0 else -0.5, a_plus=1.0 if r == 0 else 0.75)
        assert net.last_plan == "generic"
        np.testing.assert_array_equal(host(mon.get("s")).reshape(T, 1, N).astype(u8), unpack(g[f"r{r}_sY"], (T, 1, N)))
        for got, key in ((feat.value, "W"), (net.layers["Y"].v, "vY"), (rule.p_plus, "p_plus"), (rule.p_minus, "p_minus"),
                         (rule.eligibility, "elig"), (rule.eligibility_trace, "e_trace")):
            np.testing.assert_array_equal(bits(host(got).reshape(-1)), bits(g[f"r{r}_{key}"].reshape(-1)), err_msg=f"run {r} {key}")
        net.reset_state_variables()
    # the rule object driven by hand (connection.update -> feature.update -> rule.update) against the oracle
    W = host(feat.value).copy(); et = host(rule.eligibility_trace).copy(); el = host(rule.eligibility).copy()
    pp = host(rule.p_plus).copy(); pm = host(rule.p_minus).copy()
    dp, dm, de = rule._decays()
    for t in range(5):
        s_src, s_tgt = synth.dense_spikes(960 + t, (1, Nin), 0.2), synth.dense_spikes(970 + t, (1, N), 0.2)
        X_.s, Y_.s = dev(s_src), dev(s_tgt)
        rule.update(reward=0.3, a_minus=-0.5)
        oracle.mstdpet(W, el, et, pp, pm, s_src.reshape(-1), s_tgt.reshape(-1), reward=0.3, nu0=np.float32(1e-1), a_minus=-0.5,
                       decay_plus=dp, decay_minus=dm, decay_e=de, tc_e=25.0, wmin=0.0, wmax=1.0)
    for got, want, key in ((feat.value, W, "W"), (rule.eligibility_trace, et, "e_trace"), (rule.eligibility, el, "elig"),
                           (rule.p_plus, pp, "p_plus"), (rule.p_minus, pm, "p_minus")):
        np.testing.assert_array_equal(bits(host(got)), bits(want), err_msg=f"update() {key}")
    with pytest.raises(NotImplementedError):
        X_.batch_size = 2
        rule.update(reward=0.3)
