#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs (parity-test cases, not the headline bench line) on one
MI355X, through the public API.  Prints one JSON object per config.

    python tools/bench_configs.py [--runs 5]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bindsnet_amd import synth  # noqa: E402

DEV = "cuda"


def timed(net, inputs, T, runs, **kw):
    for _ in range(2):
        net.run(dict(inputs), time=T, **kw); net.reset_state_variables()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(runs):
        net.run(dict(inputs), time=T, **kw); net.reset_state_variables()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"timesteps_per_s": round(runs * T / dt, 1), "ms_per_timestep": round(dt / (runs * T) * 1e3, 4), "plan": net.last_plan}


def cfg1():
    from bindsnet_amd.models import DiehlAndCook2015
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=100, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28)).to(DEV)
    x = torch.from_numpy(synth.poisson_mnist_like(1, 250, 1, seed=1)[0]).view(250, 1, 1, 28, 28).to(DEV)   # BASELINE.md cfg1 input
    return "cfg1 D&C 784->100 B=1 T=250 PostPre", net, {"X": x}, 250, {}


def cfg3(B=128):
    from bindsnet_amd.models import TwoLayerNetwork
    torch.manual_seed(0)
    net = TwoLayerNetwork(n_inpt=784, n_neurons=1600, reduction=torch.sum).to(DEV)
    x = torch.from_numpy(synth.dense_spikes(2, (100, B, 784), 0.012)).to(DEV)
    return f"cfg3 TwoLayer 784->1600 B={B} T=100 PostPre (one GPU)", net, {"X": x}, 100, {}


def cfg3_shard():
    return cfg3(16)      # the per-GPU share of cfg3's batch of 128 over 8 GPUs


def cfg3_b32():
    return cfg3(32)


def cfg4():
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Conv2dConnection
    torch.manual_seed(0)
    net = Network(dt=1.0, learning=False)
    net.add_layer(Input(shape=(1, 28, 28)), "X")
    net.add_layer(LIFNodes(shape=(32, 24, 24)), "Y")
    net.add_connection(Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=5, stride=1, w=0.3 * torch.rand(32, 1, 5, 5)), "X", "Y")
    net.to(DEV)
    x = torch.from_numpy(synth.dense_spikes(3, (250, 64, 1, 28, 28), 0.05)).to(DEV)
    return "cfg4 Conv2d 5x5x32 -> LIF B=64 T=250 no learning", net, {"X": x}, 250, {}


def cfg5():
    from bindsnet_amd.learning import MSTDP
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    torch.manual_seed(0)
    net = Network(dt=1.0)
    net.add_layer(Input(n=6400, shape=(1, 80, 80), traces=True), "X")
    net.add_layer(LIFNodes(n=500, traces=True), "Y")
    net.add_connection(Connection(net.layers["X"], net.layers["Y"], wmin=0, wmax=1, update_rule=MSTDP, nu=1e-1, norm=0.5 * 6400,
                                  reduction=torch.sum), "X", "Y")
    net.to(DEV)
    x = torch.from_numpy(synth.dense_spikes(4, (100, 16, 1, 80, 80), 0.05)).to(DEV)
    return "cfg5 6400->500 LIF MSTDP B=16 T=100", net, {"X": x}, 100, {"reward": 1.0}


def _rule_two_layer(rule_name, B=32, Nin=784, N=1600, T=100):
    """Input -> Connection(rule) -> LIFNodes at cfg3's width (one GPU's share of the batch)."""
    from bindsnet_amd import learning
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    torch.manual_seed(0)
    net = Network(dt=1.0)
    net.add_layer(Input(n=Nin, traces=True), "X")
    net.add_layer(LIFNodes(n=N, traces=True), "Y")
    net.add_connection(Connection(net.layers["X"], net.layers["Y"], w=0.3 * torch.rand(Nin, N), wmin=0.0, wmax=1.0,
                                  update_rule=getattr(learning, rule_name), nu=(1e-4, 1e-2), norm=78.4, reduction=torch.sum), "X", "Y")
    net.to(DEV)
    x = torch.from_numpy(synth.dense_spikes(2, (T, B, Nin), 0.012)).to(DEV)
    return f"(f) {rule_name} 784->1600 B={B} T={T}", net, {"X": x}, T, {}


def f_hebbian():
    return _rule_two_layer("Hebbian")


def f_wdpp():
    return _rule_two_layer("WeightDependentPostPre")


def f_postpre_ref():
    return _rule_two_layer("PostPre")            # the same graph with PostPre, for comparison


def f_conv_postpre():
    """conv_mnist.py's training graph: Input -> Conv2dConnection(PostPre) -> LIFNodes (generic plan)."""
    from bindsnet_amd.learning import PostPre
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Conv2dConnection
    torch.manual_seed(0)
    net = Network(dt=1.0)
    net.add_layer(Input(shape=(1, 28, 28), traces=True), "X")
    net.add_layer(LIFNodes(shape=(32, 24, 24), traces=True), "Y")
    net.add_connection(Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=5, stride=1, update_rule=PostPre, nu=(1e-4, 1e-2),
                                        wmin=0.0, wmax=1.0, reduction=torch.sum, w=0.3 * torch.rand(32, 1, 5, 5)), "X", "Y")
    net.to(DEV)
    x = torch.from_numpy(synth.dense_spikes(3, (100, 16, 1, 28, 28), 0.05)).to(DEV)
    return "(f) Conv2d 5x5x32 PostPre -> LIF B=16 T=100", net, {"X": x}, 100, {}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    for make in (cfg1, cfg3_shard, cfg3_b32, cfg3, cfg4, cfg5, f_postpre_ref, f_hebbian, f_wdpp, f_conv_postpre):
        if a.only and make.__name__ not in a.only.split(','):
            continue
        name, net, inputs, T, kw = make()
        r = timed(net, inputs, T, a.runs, **kw)
        r["config"] = name
        print(json.dumps(r), flush=True)
