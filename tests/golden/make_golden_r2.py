#!/usr/bin/env python3
"""Round-2 fixtures from the UNMODIFIED reference (build container only):

    python tests/golden/make_golden_r2.py

  run_two_mcc_mstdp_*  Input -> MulticompartmentConnection[Weight, MCC_learning.MSTDP] -> LIFNodes, two consecutive runs
                       (reward +1, then a per-sample reward vector) -- everything on this path is ATen-ordered, so rasters,
                       weights and the rule's state are compared bit for bit.
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


if __name__ == "__main__":
    jobs = sys.argv[1:] or ["mstdp", "monitor"]
    if "mstdp" in jobs:
        mcc_mstdp_case("run_two_mcc_mstdp_b4", 196, 48, 4, 40)
        mcc_mstdp_case("run_two_mcc_mstdp_b20", 196, 37, 20, 30)
        mcc_mstdp_case("run_two_mcc_mstdp_n208", 208, 40, 16, 30)      # Nin % 16 == 0: the fused plan's shape
    if "monitor" in jobs:
        net_monitor_case()
        net_monitor_b1_case()
