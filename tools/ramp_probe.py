#!/usr/bin/env python3
"""Does the duration of k_dc2015_async fall over the first ~100 launches of a process because the NETWORK changes (it learns from run to
run) or because the DEVICE does (clocks)?  Same cfg2 network and input as bench.py, but weights and thresholds are put back to their
initial values before every run, so every launch does identical work; wall clock per block of 10 runs inside one pipelined section.

    python tools/ramp_probe.py [--runs 200] [--idle-ms 0]      (--idle-ms: sleep that long, GPU idle, half-way through)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=200)
    ap.add_argument("--idle-ms", type=float, default=0.0)
    ap.add_argument("--learn", action="store_true", help="do NOT restore the weights: the bench's own sequence")
    a = ap.parse_args()
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    T, B, N = 250, 32, 400
    x = torch.from_numpy(synth.poisson_mnist_like(B, T, 1, seed=1)[0]).to("cuda")
    last = x[T - 1].clone()
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    for l in ("X", "Ae", "Ai"):
        net.add_monitor(Monitor(net.layers[l], ["s"], time=T), l + "_spikes")
    net.to("cuda")
    W = net.connections[("X", "Ae")].pipeline[0].value
    W0, th0 = W.detach().clone(), net.layers["Ae"].theta.clone()
    torch.manual_seed(2)
    blocks = []
    with net.pipelined():
        torch.cuda.synchronize()
        for blk in range(a.runs // 10):
            if a.idle_ms and blk == a.runs // 20:
                net.sync(); torch.cuda.synchronize(); time.sleep(a.idle_ms / 1e3)
            t0 = time.perf_counter()
            for _ in range(10):
                if not a.learn:
                    W.data.copy_(W0); net.layers["Ae"].theta.copy_(th0)
                net.run({"X": x}, time=T)
                net.reset_state_variables()
                x[T - 1].copy_(last)
            net.sync()
            torch.cuda.synchronize()
            blocks.append(round((time.perf_counter() - t0) / 10 * 1e3, 4))
    print(json.dumps({"what": "ms per run, blocks of 10 runs" + ("" if a.learn else ", identical work every run (weights and thresholds restored)"),
                      "idle_ms_before_block": [a.runs // 20, a.idle_ms] if a.idle_ms else None, "ms_per_run": blocks}))


if __name__ == "__main__":
    main()
