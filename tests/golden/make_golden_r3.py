#!/usr/bin/env python3
"""Round-3 fixtures from the UNMODIFIED reference (build container only):

    python tests/golden/make_golden_r3.py

 run_ext_current.npz     run(inputs={"X": spikes, "A": currents, "B": currents}) -- external input currents into
                         non-Input layers (network.py:386-392) on Input -> A -> B with MulticompartmentConnections
 run_one_step_clamp.npz  run(..., one_step=True, clamp=..., unclamp=...) -- a clamped layer's spikes feed the layers
                         behind it in the same timestep (network.py:388-429)
 op_conv_normalize.npz   Conv2dConnection.normalize (topology.py:824-837): every [KH*KW] filter scaled to sum `norm` -- the 1-D
                         sums run in ATen's vectorised INNER-sum order -- for kernels of 2x2 ... 16x16 taps; and a conv_mnist.py
                         style run (Conv2d + PostPre + norm, two consecutive inputs: normalised after each)
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
from make_golden import Input, LIFNodes, Monitor, MulticompartmentConnection, Network, T_, Weight, save  # noqa: E402

nX, nA, nB, B, T = 64, 40, 24, 3, 30


def chain():
    net = Network(dt=1.0, learning=False)
    X, A, Bl = Input(n=nX), LIFNodes(n=nA, thresh=-60.0), LIFNodes(n=nB, thresh=-61.0)
    net.add_layer(X, "X"); net.add_layer(A, "A"); net.add_layer(Bl, "B")
    for k, (src, dst, ns, nd, sc) in enumerate((("X", "A", nX, nA, 2.0), ("A", "B", nA, nB, 0.35), ("B", "A", nB, nA, -1.0))):
        w = synth.uniform_f32(3200 + k, (ns, nd), 0.0, abs(sc)) * np.sign(sc)
        c = MulticompartmentConnection(net.layers[src], net.layers[dst], device="cpu", pipeline=[Weight("weight", T_(w.astype(np.float32)).clone())])
        net.add_connection(c, src, dst)
    mons = {l: Monitor(net.layers[l], ["s", "v"], time=T) for l in ("A", "B")}
    for l, m in mons.items():
        net.add_monitor(m, l)
    return net, mons


def ext_current_case():
    net, mons = chain()
    sp = synth.dense_spikes(3210, (T, B, nX), 0.10)
    cA = synth.uniform_f32(3211, (T, B, nA), -1.0, 4.0)
    cB = synth.uniform_f32(3212, (T, B, nB), 0.0, 2.5)
    out = {}
    net.run({"X": T_(sp), "A": T_(cA), "B": T_(cB)}, time=T)
    for l in ("A", "B"):
        out[f"s_{l}"] = np.packbits(mons[l].get("s").numpy().astype(np.uint8))
        out[f"v_{l}"] = mons[l].get("v").numpy().copy()
    print("  ext currents: A spikes", int(mons["A"].get("s").sum()), "B spikes", int(mons["B"].get("s").sum()))
    save("run_ext_current", **out)


def one_step_clamp_case():
    net, mons = chain()
    sp = synth.dense_spikes(3220, (T, B, nX), 0.15)
    clampA = torch.zeros(nA, dtype=torch.bool); clampA[::7] = True
    unclampA = torch.zeros(nA, dtype=torch.bool); unclampA[3::5] = True
    out = {}
    for tag, flag in (("one", True), ("sync", False)):
        net.reset_state_variables()
        net.run({"X": T_(sp)}, time=T, one_step=flag, clamp={"A": clampA}, unclamp={"A": unclampA})
        for l in ("A", "B"):
            out[f"{tag}_{l}"] = np.packbits(mons[l].get("s").numpy().astype(np.uint8))
        print(f"  one_step={flag} with clamp: A spikes {int(mons['A'].get('s').sum())}, B spikes {int(mons['B'].get('s').sum())}")
    assert not np.array_equal(out["one_B"], out["sync_B"]), "the fixture must tell the two modes apart"
    save("run_one_step_clamp", **out)


CONV_NORM_CASES = [(4, 1, 3), (8, 3, 5), (25, 1, 16), (3, 2, 2), (6, 4, 7), (2, 1, 23)]      # (Cout, Cin, K): K*K = 9 ... 529 taps, 4 (< one vector)


def conv_normalize_case():
    from make_golden import Conv2dConnection
    from bindsnet.learning import PostPre
    out = {"cases": np.array(CONV_NORM_CASES)}
    for k, (Cout, Cin, K) in enumerate(CONV_NORM_CASES):
        H = K + 3
        W = synth.uniform_f32(3300 + k, (Cout, Cin, K, K), 0.05, 1.0)
        c = Conv2dConnection(Input(shape=(Cin, H, H)), LIFNodes(shape=(Cout, 4, 4)), kernel_size=K, w=T_(W).clone(), norm=0.4 * K * K)
        c.normalize()
        out[f"w{k}"] = c.w.detach().numpy().copy()
    # a run: Input -> Conv2d(PostPre, norm) -> LIF, B = 2, two consecutive inputs (weights normalised after each run)
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
        out[f"run{r}_sY"] = np.packbits(mon.get("s").numpy().astype(np.uint8))
        out[f"run{r}_W"] = cc.w.detach().numpy().copy()
        print(f"  conv normalize run {r}: spikes {int(mon.get('s').sum())}, filter sums {cc.w.detach().view(4, -1).sum(1).numpy()}")
        net.reset_state_variables()
    save("op_conv_normalize", **out)


def conv_mnist_graph(mod, n_filters=25, kernel_size=16, stride=4, seed=0):
    """examples/mnist/conv_mnist.py:84-126 with the classes of `mod` (the reference here, the mirror in the tests): Input(1,28,28) ->
    Conv2dConnection(PostPre, norm = 0.4 k^2, wmax = 1) -> DiehlAndCookNodes(n_filters, c, c) with the script's lateral inhibition
    between filters at the same position (dense Connection, -100)."""
    conv_size = int((28 - kernel_size) / stride) + 1
    torch.manual_seed(seed)
    network = mod.Network()
    input_layer = mod.Input(n=784, shape=(1, 28, 28), traces=True)
    conv_layer = mod.DiehlAndCookNodes(n=n_filters * conv_size * conv_size, shape=(n_filters, conv_size, conv_size), traces=True)
    conv_conn = mod.Conv2dConnection(input_layer, conv_layer, kernel_size=kernel_size, stride=stride, update_rule=mod.PostPre,
                                     norm=0.4 * kernel_size ** 2, nu=[1e-4, 1e-2], wmax=1.0)
    w = torch.zeros(n_filters, conv_size, conv_size, n_filters, conv_size, conv_size)
    for f1 in range(n_filters):
        for f2 in range(n_filters):
            if f1 != f2:
                for i in range(conv_size):
                    for j in range(conv_size):
                        w[f1, i, j, f2, i, j] = -100.0
    w = w.view(n_filters * conv_size * conv_size, n_filters * conv_size * conv_size)
    recurrent_conn = mod.Connection(conv_layer, conv_layer, w=w)
    network.add_layer(input_layer, name="X")
    network.add_layer(conv_layer, name="Y")
    network.add_connection(conv_conn, source="X", target="Y")
    network.add_connection(recurrent_conn, source="Y", target="Y")
    mons = {"v": mod.Monitor(network.layers["Y"], ["v"], time=60), "s": mod.Monitor(network.layers["Y"], ["s"], time=60)}
    network.add_monitor(mons["v"], name="output_voltage")
    network.add_monitor(mons["s"], name="output_spikes")
    return network, mons, conv_conn


def conv_mnist_case():
    """The conv_mnist.py training graph, batch 1 (the script's), two consecutive inputs of 60 timesteps with a reset between."""
    import types
    from make_golden import Conv2dConnection, Connection, DiehlAndCookNodes
    from bindsnet.learning import PostPre
    mod = types.SimpleNamespace(Network=Network, Input=Input, DiehlAndCookNodes=DiehlAndCookNodes, Conv2dConnection=Conv2dConnection,
                                Connection=Connection, PostPre=PostPre, Monitor=Monitor)
    net, mons, cc = conv_mnist_graph(mod)
    out = {"W0": cc.w.detach().numpy().copy()}
    T3 = 60
    torch.manual_seed(2)
    for r in range(2):
        sp = synth.spike_train(3400 + r, T3, 1, 784, active=0.5, max_rate=0.35)
        net.run({"X": T_(sp).view(T3, 1, 1, 28, 28)}, time=T3)
        out[f"r{r}_sY"] = np.packbits(mons["s"].get("s").numpy().astype(np.uint8))
        out[f"r{r}_vY"] = mons["v"].get("v").numpy().copy()
        out[f"r{r}_W"] = cc.w.detach().numpy().copy()
        out[f"r{r}_theta"] = net.layers["Y"].theta.numpy().copy()
        print(f"  conv_mnist graph run {r}: spikes {int(mons['s'].get('s').sum())}, filter sums {cc.w.detach().view(25, -1).sum(1)[:3].numpy()}")
        net.reset_state_variables()
    out["probe_after"] = torch.rand(4).numpy()
    save("run_conv_mnist_graph", **out)


def clamp_index_case():
    """supervised_mnist.py:201-207: run(..., clamp={"Ae": LongTensor of neuron INDICES}) on DiehlAndCook2015 -- the reference's
    `s[:, clamp] = 1` takes an index tensor as well as a boolean mask (network.py:416-421); plus an index unclamp and a per-step
    [T, k] index clamp."""
    from make_golden import DiehlAndCook2015
    N, B, T4 = 100, 2, 60
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    feat = net.connections[("X", "Ae")].pipeline[0]
    feat.value.data.copy_(T_(synth.weights_q12(10, 784, N)))
    mons = {l: Monitor(net.layers[l], ["s"], time=T4) for l in ("Ae", "Ai")}
    for l, m in mons.items():
        net.add_monitor(m, l + "_s")
    out = {}
    specs = [dict(clamp={"Ae": torch.tensor([37])}), dict(clamp={"Ae": torch.tensor([3, 64, 99])}, unclamp={"Ae": torch.tensor([5, 6, 7, 64])}),
             dict(clamp={"Ae": T_(np.stack([np.array([t % N, (7 * t + 3) % N]) for t in range(T4)]))})]
    for r, kw in enumerate(specs):
        sp = synth.spike_train(3500 + r, T4, B, 784, max_rate=0.25)
        torch.manual_seed(2 + r)
        net.run({"X": T_(sp).view(T4, B, 1, 28, 28)}, time=T4, **kw)
        out[f"r{r}_sE"] = np.packbits(mons["Ae"].get("s").numpy().astype(np.uint8))
        out[f"r{r}_sI"] = np.packbits(mons["Ai"].get("s").numpy().astype(np.uint8))
        mg.pack(out, f"r{r}_W", feat.value.detach().numpy().copy())
        out[f"r{r}_theta"] = net.layers["Ae"].theta.numpy().copy()
        print(f"  index clamp run {r}: Ae spikes {int(mons['Ae'].get('s').sum())}, Ai {int(mons['Ai'].get('s').sum())}")
        net.reset_state_variables()
    out["probe_after"] = torch.rand(4).numpy()
    save("run_clamp_indices", **out)


def guide_example(mod, time=500, seed=0):
    """docs/source/guide/guide_part_i.rst, "Running Simulations": the guide's end-to-end example, statement for statement, with the
    classes of `mod`; returns (network, source_monitor, target_monitor, input_data)."""
    torch.manual_seed(seed)
    network = mod.Network()
    source_layer = mod.Input(n=100)
    target_layer = mod.LIFNodes(n=1000)
    network.add_layer(layer=source_layer, name="A")
    network.add_layer(layer=target_layer, name="B")
    forward_connection = mod.Connection(source=source_layer, target=target_layer, w=0.05 + 0.1 * torch.randn(source_layer.n, target_layer.n))
    network.add_connection(connection=forward_connection, source="A", target="B")
    recurrent_connection = mod.Connection(source=target_layer, target=target_layer, w=0.025 * (torch.eye(target_layer.n) - 1))
    network.add_connection(connection=recurrent_connection, source="B", target="B")
    source_monitor = mod.Monitor(obj=source_layer, state_vars=("s",), time=time)
    target_monitor = mod.Monitor(obj=target_layer, state_vars=("s", "v"), time=time)
    network.add_monitor(monitor=source_monitor, name="A")
    network.add_monitor(monitor=target_monitor, name="B")
    input_data = torch.bernoulli(0.1 * torch.ones(time, source_layer.n)).byte()
    return network, source_monitor, target_monitor, input_data


def guide_case():
    import types
    from make_golden import Connection
    mod = types.SimpleNamespace(Network=Network, Input=Input, LIFNodes=LIFNodes, Connection=Connection, Monitor=Monitor)
    net, sm, tm, x = guide_example(mod)
    net.run(inputs={"A": x}, time=500)
    s, v = tm.get("s").numpy(), tm.get("v").numpy()
    print("  guide example: B spikes", int(s.sum()), "shapes", sm.get("s").shape, s.shape, v.shape)
    save("run_guide_example", x=np.packbits(x.numpy()), sA=np.packbits(sm.get("s").numpy().astype(np.uint8)), sB=np.packbits(s.astype(np.uint8)),
         shapes=np.array([list(sm.get("s").shape), list(s.shape), list(v.shape)]), v_sample=v.reshape(-1)[::997].copy(), v_last=v[-1].copy())


if __name__ == "__main__":
    import sys as _sys
    jobs = {"guide": guide_case, "ext": ext_current_case, "clamp": one_step_clamp_case, "convnorm": conv_normalize_case, "convmnist": conv_mnist_case,
            "clampidx": clamp_index_case}
    for j in (_sys.argv[1:] or list(jobs)):
        jobs[j]()
