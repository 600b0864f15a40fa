#!/usr/bin/env python3
"""Host-side cost of Network.run() at eth_mnist.py's shape (D&C N = 100, batch 1, T = 250, its 7 monitors) as a function of
torch's intra-op thread count -- the script sets os.cpu_count() - 1 (eth_mnist.py:77).  python tools/run_overhead.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bindsnet_amd import synth  # noqa: E402
from bindsnet_amd.models import DiehlAndCook2015  # noqa: E402
from bindsnet_amd.network.monitors import Monitor  # noqa: E402

T = 250
torch.manual_seed(0)
net = DiehlAndCook2015(n_inpt=784, n_neurons=100, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
net.to("cuda")
for l in ("Ae", "Ai"):
    net.add_monitor(Monitor(net.layers[l], ["v"], time=T, device="cuda"), l + "_v0")
    net.add_monitor(Monitor(net.layers[l], ["v"], time=T, device="cuda"), l + "_voltages")
for l in ("X", "Ae", "Ai"):
    net.add_monitor(Monitor(net.layers[l], ["s"], time=T, device="cuda"), l + "_spikes")
xs = [torch.from_numpy(h).view(T, 1, 1, 28, 28) for h in synth.poisson_mnist_like(1, T, 8, seed=1)]
for nt in (8, 1, max(1, (os.cpu_count() or 2) - 1), 8):
    torch.set_num_threads(nt)
    for k in range(3):
        net.run({"X": xs[k].cuda()}, time=T); net.reset_state_variables()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 24
    for k in range(n):
        net.run({"X": xs[k % 8].cuda()}, time=T)
        net.monitors["Ae_spikes"].get("s")
        net.reset_state_variables()
    torch.cuda.synchronize()
    print(f"threads {nt:4d}: {1e3 * (time.perf_counter() - t0) / n:7.3f} ms per run()+reset  plan {net.last_plan} retries {getattr(net, 'lean_retries', 0)}", flush=True)
