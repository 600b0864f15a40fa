"""BASELINE.md section 2's config table as code, written ONCE against the BindsNET class API so that the same builder constructs the
network from this package (tools/bench_configs.py, on the MI355X) and from the unmodified reference staged under oracle/_ref
(oracle/ref_cpu_leg.py --config, on the host cores): `ns` is a namespace holding the API's classes, wherever they come from.

Seeds as BASELINE.md states them: 0 = construction (torch.manual_seed(0) by the caller right before build()), inputs from numpy
generators (synth.*: independent of torch's generator), 2 = before the first run."""
import types

import numpy as np

# name -> (T on the GPU, T of the CPU leg [BASELINE.md: 20 for cfg3 / cfg5], batch, algorithmic bytes per timestep [SURVEY.md 8(d)], run kwargs)
CONFIGS = {
    "cfg1": dict(T=250, T_cpu=250, B=1, algo_bytes=1_030_000, kw={}, what="DiehlAndCook2015 784->100, B=1, T=250, PostPre, 3 spike monitors"),
    "cfg3_shard": dict(T=100, T_cpu=20, B=16, algo_bytes=4 * 3 * 784 * 1600 + 16 * (784 * 10 + 1600 * 26), kw={},
                       what="TwoLayerNetwork 784->1600, B=16 (one GPU's share of cfg3 over 8), T=100, PostPre"),
    "cfg3_b32": dict(T=100, T_cpu=20, B=32, algo_bytes=4 * 3 * 784 * 1600 + 32 * (784 * 10 + 1600 * 26), kw={},
                     what="TwoLayerNetwork 784->1600, B=32, T=100, PostPre"),
    "cfg3": dict(T=100, T_cpu=20, n_cpu=3, B=128, algo_bytes=21_400_000, kw={}, what="TwoLayerNetwork 784->1600, B=128, T=100, PostPre (whole batch on one GPU)"),
    "cfg4": dict(T=250, T_cpu=50, B=64, algo_bytes=20_100_000, kw={}, what="Conv2dConnection 28x28 -> 32 filters 5x5 -> LIF, B=64, T=250, no learning"),
    "cfg5": dict(T=100, T_cpu=20, n_cpu=3, B=16, algo_bytes=40_400_000, kw={"reward": 1.0}, what="Input 6400 -> Connection(MSTDP) -> 500 LIF, B=16, T=100, reward 1.0"),
}


def namespace(package: str):
    """The classes build() needs, from `package` ('bindsnet_amd', or 'bindsnet' = whatever the caller put into sys.modules)."""
    import importlib
    imp = lambda m: importlib.import_module(package + "." + m)          # noqa: E731
    importlib.import_module(package + ".network")                       # first: learning <-> topology_features import cycle
    nodes, topo, models = imp("network.nodes"), imp("network.topology"), imp("models")
    return types.SimpleNamespace(Network=imp("network").Network, Input=nodes.Input, LIFNodes=nodes.LIFNodes, Connection=topo.Connection,
                                 Conv2dConnection=topo.Conv2dConnection, MSTDP=imp("learning").MSTDP, Monitor=imp("network.monitors").Monitor,
                                 DiehlAndCook2015=models.DiehlAndCook2015, TwoLayerNetwork=models.TwoLayerNetwork)


def build(name: str, ns, T: int):
    """-> (network, {monitor name: (layer name, Monitor)}).  Call torch.manual_seed(0) first.  Every layer gets a spike monitor (the
    parity artefacts of BASELINE.md section 2 are the rasters of every layer; cfg1 / cfg2 state the three of eth_mnist.py:143-148)."""
    import torch
    if name == "cfg1":
        net = ns.DiehlAndCook2015(n_inpt=784, n_neurons=100, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    elif name in ("cfg3", "cfg3_b32", "cfg3_shard"):
        net = ns.TwoLayerNetwork(n_inpt=784, n_neurons=1600, reduction=torch.sum)
    elif name == "cfg4":
        net = ns.Network(dt=1.0, learning=False)
        net.add_layer(ns.Input(shape=(1, 28, 28)), "X")
        net.add_layer(ns.LIFNodes(shape=(32, 24, 24)), "Y")
        net.add_connection(ns.Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=5, stride=1, w=0.3 * torch.rand(32, 1, 5, 5)), "X", "Y")
    elif name == "cfg5":
        net = ns.Network(dt=1.0)
        net.add_layer(ns.Input(n=6400, shape=(1, 80, 80), traces=True), "X")
        net.add_layer(ns.LIFNodes(n=500, traces=True), "Y")
        net.add_connection(ns.Connection(net.layers["X"], net.layers["Y"], wmin=0, wmax=1, update_rule=ns.MSTDP, nu=1e-1, norm=0.5 * 6400,
                                         reduction=torch.sum), "X", "Y")
    else:
        raise KeyError(name)
    mons = {}
    for lname, layer in net.layers.items():
        m = ns.Monitor(layer, ["s"], time=T)
        net.add_monitor(m, lname + "_spikes")
        mons[lname] = m
    return net, mons


def inputs(name: str, n: int = 1):
    """`n` consecutive input batches u8 [T, B, ...] (numpy) and the Input layer's name."""
    from bindsnet_amd import synth
    c = CONFIGS[name]
    T, B = c["T"], c["B"]
    if name == "cfg1":
        return synth.poisson_mnist_like(1, 250, n, seed=1), "X"
    if name.startswith("cfg3"):
        return [synth.dense_spikes(2 + 10 * k, (T, B, 784), 0.012) for k in range(n)], "X"
    if name == "cfg4":
        return [synth.dense_spikes(3 + 10 * k, (T, B, 1, 28, 28), 0.05) for k in range(n)], "X"
    if name == "cfg5":
        return [synth.dense_spikes(4 + 10 * k, (T, B, 1, 80, 80), 0.05) for k in range(n)], "X"
    raise KeyError(name)


def learned_weights(net):
    """The connection's weight tensor whatever the connection class ((src, dst) -> tensor), for the parity record."""
    out = {}
    for key, conn in net.connections.items():
        if hasattr(conn, "pipeline"):
            out[key] = conn.pipeline[0].value
        else:
            out[key] = conn.w
    return out
