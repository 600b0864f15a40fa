#!/usr/bin/env python3
"""Round-2 fixtures from the UNMODIFIED reference (build container only):

    python tests/golden/make_golden_r2.py

  run_two_mcc_mstdp_*  Input -> MulticompartmentConnection[Weight, MCC_learning.MSTDP] -> LIFNodes, two consecutive runs
                       (reward +1, then a per-sample reward vector) -- everything on this path is ATen-ordered, so rasters,
                       weights and the rule's state are compared bit for bit.
  run_two_mcc_mstdpet_b1  the same graph with MCC_learning.MSTDPET (:554-733; batch 1 like the dense rule), two runs with
                       different rewards; the rule's dense eligibility trace is part of the fixture.
  run_dc_v2_n64_b4     DiehlAndCook2015v2 (dense input connection, recurrent inhibition), two runs.
  run_dc_n100_b3_dt05  DiehlAndCook2015 at dt = 0.5 (the generator of make_golden.py with another timestep).
  op_conv_mstdp        MSTDP on a Conv2dConnection (learning.py:1942-2015; batch 1): update sequences + a run.
  conn_monitor         Monitor / NetworkMonitor on a Connection's `w` (one snapshot per timestep).
  net_monitor          NetworkMonitor / sparse Monitor recordings of a DiehlAndCook2015 run (monitors.py:30-329).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402
import make_golden as mg  # noqa: E402
from make_golden import DiehlAndCook2015, Input, LIFNodes, Monitor, MulticompartmentConnection, Network, T_, Weight, save, sha  # noqa: E402
from bindsnet.learning.MCC_learning import MSTDP as MCC_MSTDP  # noqa: E402
from bindsnet.learning.MCC_learning import MSTDPET as MCC_MSTDPET  # noqa: E402
from bindsnet.network.monitors import NetworkMonitor  # noqa: E402

torch.set_num_threads(8)


def mcc_mstdp_case(name, Nin, N, B, T):
    W0 = synth.weights_q12(11, Nin, N)
    net = Network(dt=1.0)
    X_, Y_ = Input(n=Nin, traces=True), LIFNodes(n=N, traces=True)
    feat = Weight("weight", T_(W0).clone(), range=[0.0, 1.0], norm=0.1 * Nin, nu=(1e-1, 1e-1), learning_rule=MCC_MSTDP)
    # (reduction stays None: the reference's own type check of that argument raises for any callable, topology_features.py:117;
    #  None resolves to torch.sum here because the layers do not belong to a network yet, MCC_learning.py:75-81)
    conn = MulticompartmentConnection(X_, Y_, device="cpu", pipeline=[feat])
    net.add_layer(X_, "X")
    net.add_layer(Y_, "Y")
    net.add_connection(conn, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    out = {}
    for r in range(2):
        spikes = synth.spike_train(30 + r, T, B, Nin, active=0.3, max_rate=0.12)
        reward = 1.0 if r == 0 else T_(synth.uniform_f32(17, (B,), -1.0, 1.0)).view(B, 1, 1)   # (broadcasts over [B, Nin, N])
        net.run({"X": T_(spikes)}, time=T, reward=reward)
        rule = feat.learning_rule
        out[f"r{r}_sY"] = np.packbits(mon.get("s").numpy().astype(np.uint8))
        out[f"r{r}_W"] = feat.value.detach().numpy().copy()
        out[f"r{r}_vY"] = net.layers["Y"].v.numpy().copy()
        out[f"r{r}_p_plus"] = rule.p_plus.numpy().copy()
        out[f"r{r}_p_minus"] = rule.p_minus.numpy().copy()
        out[f"r{r}_elig_sha"] = sha(rule.eligibility.numpy())
        print(f"  {name} run {r}: Y spikes {int(mon.get('s').sum())}")
        net.reset_state_variables()
    Y, X = net.layers["Y"], net.layers["X"]
    out.update(decay=Y.decay.numpy(), y_trace_decay=Y.trace_decay.numpy(), x_trace_decay=X.trace_decay.numpy(),
               decay_plus=torch.exp(-torch.tensor(1.0) / rule.tc_plus).numpy(),
               decay_minus=torch.exp(-torch.tensor(1.0) / rule.tc_minus).numpy())
    save(name, Nin=Nin, N=N, B=B, T=T, **out)


def mcc_mstdpet_case(name, Nin, N, T):
    """Input -> MulticompartmentConnection[Weight, MCC_learning.MSTDPET] -> LIFNodes at batch 1 (the rule flattens the
    spikes, MCC_learning.py:665-666), two consecutive runs with different rewards, no reset of the rule's P+ / P- in
    between (MSTDPET.reset_state_variables clears the eligibilities only, :731-734)."""
    W0 = synth.weights_q12(11, Nin, N)
    net = Network(dt=1.0)
    X_, Y_ = Input(n=Nin, traces=True), LIFNodes(n=N, traces=True)
    feat = Weight("weight", T_(W0).clone(), range=[0.0, 1.0], norm=0.1 * Nin, nu=(1e-1, 1e-1), learning_rule=MCC_MSTDPET)
    conn = MulticompartmentConnection(X_, Y_, device="cpu", pipeline=[feat], tc_e_trace=25.0)
    net.add_layer(X_, "X")
    net.add_layer(Y_, "Y")
    net.add_connection(conn, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    out = {}
    rule = feat.learning_rule
    for r in range(2):
        spikes = synth.spike_train(30 + r, T, 1, Nin, active=0.3, max_rate=0.12)
        net.run({"X": T_(spikes)}, time=T, reward=0.8 if r == 0 else -0.5, a_plus=1.0 if r == 0 else 0.75)
        out[f"r{r}_sY"] = np.packbits(mon.get("s").numpy().astype(np.uint8))
        out[f"r{r}_W"] = feat.value.detach().numpy().copy()
        out[f"r{r}_vY"] = net.layers["Y"].v.numpy().copy()
        out[f"r{r}_p_plus"] = rule.p_plus.numpy().copy()
        out[f"r{r}_p_minus"] = rule.p_minus.numpy().copy()
        out[f"r{r}_elig"] = rule.eligibility.numpy().copy()
        out[f"r{r}_e_trace"] = rule.eligibility_trace.numpy().copy()
        print(f"  {name} run {r}: Y spikes {int(mon.get('s').sum())}, |e_trace| max {float(rule.eligibility_trace.abs().max()):.4g}")
        net.reset_state_variables()
    Y, X = net.layers["Y"], net.layers["X"]
    out.update(decay=Y.decay.numpy(), y_trace_decay=Y.trace_decay.numpy(), x_trace_decay=X.trace_decay.numpy(),
               decay_plus=torch.exp(-torch.tensor(1.0) / rule.tc_plus).numpy(),
               decay_minus=torch.exp(-torch.tensor(1.0) / rule.tc_minus).numpy(),
               decay_e=torch.exp(-torch.tensor(1.0) / rule.tc_e_trace).numpy(), tc_e=rule.tc_e_trace.numpy())
    save(name, Nin=Nin, N=N, B=1, T=T, **out)


def conn_monitor_case():
    """Monitors on a connection's weights (monitors.py:94-111 Monitor.record, :222-262 NetworkMonitor.record, called at the
    end of every timestep, network.py:456-458): Input -> Connection[PostPre] -> LIFNodes, batch 3, two consecutive runs;
    a Monitor on `w` with a window of T steps and a NetworkMonitor with its default state_vars ("v", "s", "w") and no
    window.  (`Connection.compute` is an MKL sgemm: rasters are compared exactly, weights within 1e-5 -- DESIGN.md 2.)"""
    from make_golden import Connection, PostPre
    Nin, N, B, T = 48, 16, 3, 24
    net = Network(dt=1.0)
    X_, Y_ = Input(n=Nin, traces=True), LIFNodes(n=N, traces=True)
    conn = Connection(X_, Y_, w=T_(synth.weights_q12(12, Nin, N) * np.float32(2.0)).clone(), update_rule=PostPre, nu=(1e-2, 5e-2),
                      reduction=torch.sum, wmin=0.0, wmax=2.0, norm=0.4 * Nin)
    net.add_layer(X_, "X"); net.add_layer(Y_, "Y")
    net.add_connection(conn, "X", "Y")
    mw = Monitor(conn, ["w"], time=T)
    nm = NetworkMonitor(net)
    net.add_monitor(mw, "w"); net.add_monitor(nm, "all")
    out = {}
    for r in range(2):
        spikes = synth.spike_train(40 + r, T, B, Nin, active=0.5, max_rate=0.3)
        net.run({"X": T_(spikes)}, time=T)
        out[f"r{r}_mon_w"] = mw.get("w").numpy().copy()
        out[f"r{r}_final_w"] = conn.w.detach().numpy().copy()
        net.reset_state_variables()          # (empties the Monitor; the NetworkMonitor is reset too, monitors.py:301-329)
        print(f"  conn monitor run {r}: mon_w {out[f'r{r}_mon_w'].shape}")
    # a NetworkMonitor that is NOT reset between two runs keeps growing
    net2 = Network(dt=1.0)
    X2, Y2 = Input(n=Nin, traces=True), LIFNodes(n=N, traces=True)
    conn2 = Connection(X2, Y2, w=T_(synth.weights_q12(12, Nin, N) * np.float32(2.0)).clone(), update_rule=PostPre, nu=(1e-2, 5e-2),
                       reduction=torch.sum, wmin=0.0, wmax=2.0, norm=0.4 * Nin)
    net2.add_layer(X2, "X"); net2.add_layer(Y2, "Y")
    net2.add_connection(conn2, "X", "Y")
    nm2 = NetworkMonitor(net2)
    net2.add_monitor(nm2, "all")
    for r in range(2):
        net2.run({"X": T_(synth.spike_train(40 + r, T, B, Nin, active=0.5, max_rate=0.3))}, time=T)
    rec = nm2.get()
    out["nm_keys"] = np.array(sorted(f"{k}:{v}" for k in rec for v in rec[k]))
    out["nm_w"] = rec[("X", "Y")]["w"].numpy().copy()
    out["nm_Y_s"] = np.packbits(rec["Y"]["s"].numpy().astype(np.uint8))
    out["nm_Y_s_shape"] = np.array(rec["Y"]["s"].shape)
    out["nm_Y_v"] = rec["Y"]["v"].numpy().copy()
    out["nm_X_s_shape"] = np.array(rec["X"]["s"].shape)
    print("  NetworkMonitor keys:", list(out["nm_keys"]), "w", out["nm_w"].shape, "Y.s", rec["Y"]["s"].shape, "spikes", int(rec["Y"]["s"].sum()))
    out.update(decay=Y_.decay.numpy(), y_trace_decay=Y_.trace_decay.numpy(), x_trace_decay=X_.trace_decay.numpy())
    save("conn_monitor", Nin=Nin, N=N, B=B, T=T, **out)


CONV_MSTDP_CASES = [(1, 12, 12, 4, 5, 1, 0), (2, 9, 9, 3, 3, 2, 1), (3, 8, 8, 40, 3, 1, 1)]      # (Cin, H, W, Cout, K, stride, pad)


def conv_mstdp_case():
    """MSTDP on a Conv2dConnection (learning.py:1942-2015), batch 1 -- the only batch size at which the reference's
    `eligibility.view(w.size())` works.  (a) sequences of update() calls on given spikes (three geometries; K = Cin*k*k is
    25 / 18 / 27 columns with 4 / 3 / 40 output channels summed by :1967); (b) a Network.run()."""
    from bindsnet.learning import MSTDP
    from make_golden import Conv2dConnection, _set_layer
    out = {"cases": np.array(CONV_MSTDP_CASES)}
    for k, (Cin, H, Wd, Cout, K, stride, pad) in enumerate(CONV_MSTDP_CASES):
        OH = (H + 2 * pad - K) // stride + 1
        src, tgt = Input(shape=(Cin, H, Wd), traces=True), LIFNodes(shape=(Cout, OH, OH), traces=True)
        c = Conv2dConnection(src, tgt, kernel_size=K, stride=stride, padding=pad, w=T_(synth.uniform_f32(2300 + k, (Cout, Cin, K, K), 0.0, 0.5)).clone(),
                             update_rule=MSTDP, nu=(2e-2, 1e-2), wmin=0.0, wmax=0.6, weight_decay=0.0 if k != 1 else 1e-3)
        c.dt = 1.0
        for t in range(10):
            _set_layer(src, 1, synth.dense_spikes(2310 + 20 * k + t, (1, Cin, H, Wd), 0.2), None)
            _set_layer(tgt, 1, synth.dense_spikes(2700 + 20 * k + t, (1, Cout, OH, OH), 0.15).astype(bool), None)
            c.update(learning=True, reward=0.7 if t % 3 else -0.4, a_plus=1.0, a_minus=-0.8)
        ur = c.update_rule
        out[f"w{k}"] = c.w.detach().numpy().copy()
        out[f"elig{k}"] = ur.eligibility.numpy().copy()
        out[f"p_plus{k}"] = ur.p_plus.numpy().copy()          # unfolded: [1, Cin*K*K, L]
        out[f"p_minus{k}"] = ur.p_minus.numpy().copy()        # [1, Cout, L]
    out.update(decay_plus=torch.exp(-torch.tensor(1.0) / ur.tc_plus).numpy(), decay_minus=torch.exp(-torch.tensor(1.0) / ur.tc_minus).numpy())
    # (b) a run
    T3 = 40
    net = Network(dt=1.0)
    net.add_layer(Input(shape=(1, 12, 12), traces=True), "X")
    net.add_layer(LIFNodes(shape=(4, 10, 10), traces=True), "Y")
    cc = Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=3, stride=1, w=T_(synth.uniform_f32(2290, (4, 1, 3, 3), 0.0, 3.0)).clone(),
                          update_rule=MSTDP, nu=(2e-3, 1e-3), wmin=0.0, wmax=4.0)
    net.add_connection(cc, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T3)
    net.add_monitor(mon, "s")
    sp = synth.dense_spikes(2291, (T3, 1, 1, 12, 12), 0.2)
    net.run({"X": T_(sp)}, time=T3, reward=0.6)
    out.update(run_sY=np.packbits(mon.get("s").numpy().astype(np.uint8)), run_W=cc.w.detach().numpy().copy(),
               run_elig=cc.update_rule.eligibility.numpy().copy(), run_Y_decay=net.layers["Y"].decay.numpy())
    print("  conv MSTDP: run spikes", int(mon.get("s").sum()), "| max |w - w0| in the update sequences:",
          [float(np.abs(out[f"w{k}"] - synth.uniform_f32(2300 + k, out[f"w{k}"].shape, 0.0, 0.5)).max()) for k in range(len(CONV_MSTDP_CASES))])
    save("op_conv_mstdp", **out)


def dc_v2_case(name="run_dc_v2_n64_b4", model="DiehlAndCook2015v2"):
    """DiehlAndCook2015v2 (models.py:247-346) / IncreasingInhibitionNetwork (:349-454): Input -> Connection[PostPre] -> D&C
    nodes with a recurrent Connection (uniform inhibition / distance-graded weights) -- dense propagation (MKL sgemm:
    rasters compared exactly, weights within 1e-5) into one_spike nodes that consume the host generator; two runs with a
    reset in between."""
    from bindsnet.models import DiehlAndCook2015v2, IncreasingInhibitionNetwork
    N, B, T = 64, 4, 80
    torch.manual_seed(0)
    if model == "DiehlAndCook2015v2":
        net = DiehlAndCook2015v2(n_inpt=784, n_neurons=N, inh=60.0, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28), reduction=torch.sum)
    else:       # start_inhib / max_inhib as examples/mnist/SOM_LM-SNNs.py:88-89 sets them
        net = IncreasingInhibitionNetwork(n_input=784, n_neurons=N, start_inhib=10, max_inhib=-40.0, dt=1.0, norm=78.4, theta_plus=0.05,
                                          inpt_shape=(1, 28, 28), reduction=torch.sum)
    conn = net.connections[("X", "Y")]
    conn.w.data.copy_(T_(synth.weights_q12(10, 784, N)))
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    counter = {"n": 0}
    orig = torch.multinomial

    def counting(p, n, *a, **k):
        counter["n"] += p.numel()
        return orig(p, n, *a, **k)

    torch.multinomial = counting
    out = {}
    Y, X = net.layers["Y"], net.layers["X"]
    for r in range(2):
        spikes = synth.spike_train(20 + r, T, B, 784, max_rate=0.25)
        torch.manual_seed(2 + r)
        counter["n"] = 0
        net.run({"X": T_(spikes).view(T, B, 1, 28, 28)}, time=T)
        out[f"r{r}_sY"] = np.packbits(mon.get("s").numpy().astype(np.uint8))
        out[f"r{r}_consumed"] = np.int64(counter["n"])
        out[f"r{r}_W_rows7"] = conn.w.detach().numpy()[::7].copy()          # every 7th source row (MKL propagation: compared within 1e-5)
        out[f"r{r}_W_colsum"] = conn.w.detach().numpy().sum(0)
        out[f"r{r}_theta"] = Y.theta.numpy().copy()
        out[f"r{r}_vY"] = Y.v.numpy().copy()
        print(f"  {model} run {r}: spikes {int(mon.get('s').sum())}, draws {counter['n']}")
        net.reset_state_variables()
    torch.multinomial = orig
    out.update(x_trace_decay=X.trace_decay.numpy(), decay=Y.decay.numpy(), theta_decay=Y.theta_decay.numpy(), trace_decay=Y.trace_decay.numpy())
    if model != "DiehlAndCook2015v2":
        out["W_yy"] = net.connections[("Y", "Y")].w.detach().numpy().copy()
    save(name, N=N, B=B, T=T, **out)


def net_monitor_case():
    """NetworkMonitor over layers + connections, and sparse spike Monitors, on a small D&C run."""
    N, B, T = 100, 3, 30
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    net.connections[("X", "Ae")].pipeline[0].value.data.copy_(T_(synth.weights_q12(10, 784, N)))
    # the reference's NetworkMonitor sizes its buffers for batch 1 (monitors.py:181-224): record sample by sample
    out = {}
    sp = Monitor(net.layers["Ae"], ["s"], time=T, sparse=True, batch_size=B)
    net.add_monitor(sp, "Ae_sparse")
    dn = Monitor(net.layers["Ae"], ["s", "v"], time=T, batch_size=B)
    net.add_monitor(dn, "Ae_dense")
    spikes = synth.spike_train(20, T, B, 784)
    torch.manual_seed(2)
    net.run({"X": T_(spikes).view(T, B, 1, 28, 28)}, time=T)
    s_sparse = sp.get("s")
    out.update(sparse_is_sparse=np.bool_(s_sparse.is_sparse), sparse_shape=np.array(s_sparse.shape),
               sparse_dense=np.packbits(s_sparse.to_dense().numpy().astype(np.uint8)),
               dense_s=np.packbits(dn.get("s").numpy().astype(np.uint8)), dense_v=dn.get("v").numpy().copy())
    save("net_monitor", N=N, B=B, T=T, **out)


def net_monitor_b1_case():
    N, T = 100, 40
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    net.connections[("X", "Ae")].pipeline[0].value.data.copy_(T_(synth.weights_q12(10, 784, N)))
    nm = NetworkMonitor(net, layers=["Ae", "Ai"], connections=[], state_vars=["s", "v"], time=T)
    net.add_monitor(nm, "all")
    spikes = synth.spike_train(21, T, 1, 784, max_rate=0.2)
    torch.manual_seed(3)
    net.run({"X": T_(spikes).view(T, 1, 1, 28, 28)}, time=T)
    rec = nm.get()
    out = {}
    for l in ("Ae", "Ai"):
        out[f"{l}_s"] = np.packbits(rec[l]["s"].numpy().astype(np.uint8))
        out[f"{l}_s_shape"] = np.array(rec[l]["s"].shape)
        out[f"{l}_v"] = rec[l]["v"].numpy().copy()
    save("net_monitor_b1", N=N, T=T, **out)


RULE_CASES = [(1, 40, 24), (3, 40, 24), (16, 40, 24), (32, 40, 24), (48, 40, 24), (5, 33, 31)]     # (B, Nin, N)


def rule_inputs(k, B, Nin, N):
    return (synth.uniform_f32(300 + k, (Nin, N), 0.0, 1.0), synth.dense_spikes(400 + k, (B, Nin), 0.3),
            synth.dense_spikes(500 + k, (B, N), 0.2), synth.uniform_f32(600 + k, (B, Nin), 0.0, 1.0),
            synth.uniform_f32(700 + k, (B, N), 0.0, 1.0))


def rules_case():
    """Single update() calls of the dense rules beyond PostPre / MSTDP (learning.py:562-653 WeightDependentPostPre,
    :1052-1135 Hebbian) and a 12-step sequence of MSTDPET (:2124-2248, batch 1)."""
    from bindsnet.learning import Hebbian, MSTDPET, WeightDependentPostPre
    from make_golden import Connection, _set_layer
    out = {}
    for k, (B, Nin, N) in enumerate(RULE_CASES):
        W0, s_src, s_tgt, x_src, x_tgt = rule_inputs(k, B, Nin, N)
        for tag, rule, kw in (("hebb", Hebbian, dict(wmin=0.0, wmax=1.0)), ("hebb_free", Hebbian, dict()),
                              ("wdpp", WeightDependentPostPre, dict(wmin=0.0, wmax=1.0)),
                              ("wdpp_decay", WeightDependentPostPre, dict(wmin=-0.5, wmax=1.5, weight_decay=0.01))):
            src, tgt = Input(n=Nin, traces=True), LIFNodes(n=N, traces=True)
            conn = Connection(src, tgt, w=T_(W0).clone(), update_rule=rule, nu=(1e-2, 3e-2), reduction=torch.sum, **kw)
            _set_layer(src, B, s_src, x_src)
            _set_layer(tgt, B, s_tgt.astype(bool), x_tgt)
            conn.update(learning=True)
            out[f"{tag}{k}"] = conn.w.detach().numpy().copy()
    # MSTDPET: batch 1, a sequence of updates (the eligibility trace integrates over steps)
    Nin, N, T = 36, 20, 12
    src, tgt = Input(n=Nin, traces=True), LIFNodes(n=N, traces=True)
    conn = Connection(src, tgt, w=T_(synth.uniform_f32(900, (Nin, N), 0.0, 1.0)).clone(), update_rule=MSTDPET, nu=(1e-1, 1e-1),
                      wmin=0.0, wmax=1.0, tc_e_trace=25.0)
    conn.dt = 1.0
    for t in range(T):
        _set_layer(src, 1, synth.dense_spikes(910 + t, (1, Nin), 0.2), None)
        _set_layer(tgt, 1, synth.dense_spikes(940 + t, (1, N), 0.2).astype(bool), None)
        conn.update(learning=True, reward=0.7 if t % 3 else -0.4)
    ur = conn.update_rule
    out.update(et_w=conn.w.detach().numpy().copy(), et_trace=ur.eligibility_trace.numpy().copy(), et_elig=ur.eligibility.numpy().copy(),
               et_p_plus=ur.p_plus.numpy().copy(), et_p_minus=ur.p_minus.numpy().copy(),
               et_decay_plus=torch.exp(-torch.tensor(1.0) / ur.tc_plus).numpy(), et_decay_minus=torch.exp(-torch.tensor(1.0) / ur.tc_minus).numpy(),
               et_decay_e=torch.exp(-torch.tensor(1.0) / ur.tc_e_trace).numpy(), et_tc_e=ur.tc_e_trace.numpy())
    save("op_rules", cases=np.array(RULE_CASES), **out)


def extras_case():
    """run() keyword arguments clamp / unclamp / injects_v / masks (network.py:395-449) on a dense two-layer network;
    LocalConnection with PostPre (topology.py:1304-1485; batch 1 like its compute()); PostPre on a Conv2dConnection
    (learning.py:457-497): single updates and a short run."""
    from bindsnet.learning import PostPre
    from make_golden import Connection, Conv2dConnection, TwoLayerNetwork, _set_layer
    from bindsnet.network.topology import LocalConnection
    out = {}
    # ---- kwargs
    Nin, N, B, T = 196, 48, 3, 40
    torch.manual_seed(0)
    net = TwoLayerNetwork(n_inpt=Nin, n_neurons=N, reduction=torch.sum, norm=78.4 * Nin / 784)
    conn = net.connections[("X", "Y")]
    conn.w.data.copy_(T_(synth.weights_q12(11, Nin, N)))
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    mv = Monitor(net.layers["Y"], ["v"], time=T)
    net.add_monitor(mon, "s"); net.add_monitor(mv, "v")
    spikes = synth.spike_train(30, T, B, Nin, active=0.3, max_rate=0.12)
    clamp = T_(synth.dense_spikes(51, (T, N), 0.03)).bool()          # per-step clamp rows
    unclamp = T_(synth.dense_spikes(52, (N,), 0.2)).bool()           # fixed unclamp mask
    inject = T_(synth.uniform_f32(53, (N,), 0.0, 0.6))
    mask = T_(synth.dense_spikes(54, (Nin, N), 0.3)).bool()
    net.run({"X": T_(spikes)}, time=T, clamp={"Y": clamp}, unclamp={"Y": unclamp}, injects_v={"Y": inject}, masks={("X", "Y"): mask})
    out.update(kw_sY=np.packbits(mon.get("s").numpy().astype(np.uint8)), kw_v=mv.get("v").numpy().copy(), kw_W=conn.w.detach().numpy().copy())
    # ---- LocalConnection + PostPre, batch 1
    np.random.seed(7)
    T2 = 50
    net = Network(dt=1.0)
    X, Y = Input(n=144, traces=True), LIFNodes(n=4 * 16, traces=True)
    lc = LocalConnection(X, Y, kernel_size=6, stride=2, n_filters=4, update_rule=PostPre, nu=(1e-4, 1e-2), wmin=0.0, wmax=1.0, norm=0.2)
    net.add_layer(X, "X"); net.add_layer(Y, "Y"); net.add_connection(lc, "X", "Y")
    out["lc_W0"] = lc.w.detach().numpy().copy()
    mon = Monitor(Y, ["s"], time=T2)
    net.add_monitor(mon, "s")
    sp = synth.spike_train(31, T2, 1, 144, active=0.5, max_rate=0.25)
    net.run({"X": T_(sp)}, time=T2)
    out.update(lc_sY=np.packbits(mon.get("s").numpy().astype(np.uint8)), lc_W=lc.w.detach().numpy().copy(), lc_norm=np.float64(lc.norm),
               lc_mask=np.packbits(lc.mask.numpy()))
    # ---- Conv2d PostPre: single updates
    cases = [(3, 1, 12, 12, 4, 3, 1, 0), (2, 3, 10, 10, 5, 3, 2, 1), (17, 2, 8, 8, 6, 5, 1, 2)]
    for k, (B, Cin, H, Wd, Cout, K, stride, pad) in enumerate(cases):
        OH = (H + 2 * pad - K) // stride + 1
        src, tgt = Input(shape=(Cin, H, Wd), traces=True), LIFNodes(shape=(Cout, OH, OH), traces=True)
        c = Conv2dConnection(src, tgt, kernel_size=K, stride=stride, padding=pad, w=T_(synth.uniform_f32(1200 + k, (Cout, Cin, K, K), 0.0, 0.5)).clone(),
                             update_rule=PostPre, nu=(1e-3, 1e-2), reduction=torch.sum, wmin=0.0, wmax=1.0)
        _set_layer(src, B, synth.dense_spikes(1300 + k, (B, Cin, H, Wd), 0.15), synth.uniform_f32(1400 + k, (B, Cin, H, Wd), 0.0, 1.0))
        _set_layer(tgt, B, synth.dense_spikes(1500 + k, (B, Cout, OH, OH), 0.1).astype(bool), synth.uniform_f32(1600 + k, (B, Cout, OH, OH), 0.0, 1.0))
        c.update(learning=True)
        out[f"cpp{k}"] = c.w.detach().numpy().copy()
    out["cpp_cases"] = np.array(cases)
    # ---- Conv2d PostPre: a run
    B, T3 = 2, 30
    net = Network(dt=1.0)
    net.add_layer(Input(shape=(1, 12, 12), traces=True), "X")
    net.add_layer(LIFNodes(shape=(4, 10, 10), traces=True), "Y")
    cc = Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=3, stride=1, w=T_(synth.uniform_f32(1700, (4, 1, 3, 3), 0.0, 3.0)).clone(),
                          update_rule=PostPre, nu=(1e-3, 1e-2), reduction=torch.sum, wmin=0.0, wmax=4.0)
    net.add_connection(cc, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T3)
    net.add_monitor(mon, "s")
    sp = synth.dense_spikes(1701, (T3, B, 1, 12, 12), 0.2)
    net.run({"X": T_(sp)}, time=T3)
    out.update(crun_sY=np.packbits(mon.get("s").numpy().astype(np.uint8)), crun_W=cc.w.detach().numpy().copy())
    print("  extras: kwargs run spikes", int(np.unpackbits(out["kw_sY"]).sum()), "| local", int(np.unpackbits(out["lc_sY"]).sum()),
          "| conv run", int(np.unpackbits(out["crun_sY"]).sum()))
    save("run_extras", **out)


def one_step_case():
    """run(..., one_step=True) (network.py:388-393) on Input -> A -> B (+ a feedback B -> A), MulticompartmentConnections."""
    nX, nA, nB, B, T = 64, 40, 24, 2, 30
    net = Network(dt=1.0, learning=False)
    X, A, Bl = Input(n=nX), LIFNodes(n=nA, thresh=-60.0), LIFNodes(n=nB, thresh=-61.0)
    net.add_layer(X, "X"); net.add_layer(A, "A"); net.add_layer(Bl, "B")
    for k, (src, dst, ns, nd, sc) in enumerate((("X", "A", nX, nA, 2.0), ("A", "B", nA, nB, 3.0), ("B", "A", nB, nA, -1.0))):
        w = synth.uniform_f32(2200 + k, (ns, nd), 0.0, abs(sc)) * np.sign(sc)
        c = MulticompartmentConnection(net.layers[src], net.layers[dst], device="cpu", pipeline=[Weight("weight", T_(w.astype(np.float32)).clone())])
        net.add_connection(c, src, dst)
    mons = {l: Monitor(net.layers[l], ["s"], time=T) for l in ("A", "B")}
    for l, m in mons.items():
        net.add_monitor(m, l)
    sp = synth.dense_spikes(2210, (T, B, nX), 0.15)
    out = {}
    for tag, flag in (("one", True), ("sync", False)):
        net.reset_state_variables()
        net.run({"X": T_(sp)}, time=T, one_step=flag)
        for l in ("A", "B"):
            out[f"{tag}_{l}"] = np.packbits(mons[l].get("s").numpy().astype(np.uint8))
        print(f"  one_step={flag}: A spikes {int(mons['A'].get('s').sum())}, B spikes {int(mons['B'].get('s').sum())}")
    save("run_one_step", **out)


if __name__ == "__main__":
    jobs = sys.argv[1:] or ["mstdp", "mstdpet", "dc_v2", "dc_dt", "conv_mstdp", "conn_monitor", "monitor", "rules", "extras", "one_step"]
    if "dc_v2" in jobs:
        dc_v2_case()
        dc_v2_case("run_iin_n64_b4", "IncreasingInhibitionNetwork")
    if "dc_dt" in jobs:      # a D&C run at dt = 0.5 ms: decays, refractory counters and MCC PostPre's `* dt` all depend on it
        mg.dc_case("run_dc_n100_b3_dt05", 100, 3, 80, 2, False, max_rate=0.125, dt=0.5)
    if "conv_mstdp" in jobs:
        conv_mstdp_case()
    if "conn_monitor" in jobs:
        conn_monitor_case()
    if "mstdpet" in jobs:
        mcc_mstdpet_case("run_two_mcc_mstdpet_b1", 196, 48, 60)
    if "one_step" in jobs:
        one_step_case()
    if "extras" in jobs:
        extras_case()
    if "rules" in jobs:
        torch.set_num_threads(1)
        rules_case()
        torch.set_num_threads(8)
    if "mstdp" in jobs:
        mcc_mstdp_case("run_two_mcc_mstdp_b4", 196, 48, 4, 40)
        mcc_mstdp_case("run_two_mcc_mstdp_b20", 196, 37, 20, 30)
        mcc_mstdp_case("run_two_mcc_mstdp_n208", 208, 40, 16, 30)      # Nin % 16 == 0: the fused plan's shape
    if "monitor" in jobs:
        net_monitor_case()
        net_monitor_b1_case()
