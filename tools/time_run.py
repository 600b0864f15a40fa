#!/usr/bin/env python3
"""Kernel time per network.run() of the bench workload (cfg2, stated input) by HIP events -- no parity leg, no CPU leg: for A/B runs of
developer switches (SNN_DC_SPECFLAGS) on one box.  python tools/time_run.py [launches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bindsnet_amd import _lib  # noqa: E402

if os.environ.get("SNN_LIB_OVERRIDE"):          # A/B of two builds on one box (developer aid; _lib honours it only with SNN_DEVELOPER=1)
    _lib.LIB_PATH = os.environ["SNN_LIB_OVERRIDE"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
net = bench.build_network(dev)
from bindsnet_amd import synth  # noqa: E402
pool = [torch.from_numpy(h).view(bench.T, bench.BATCH, 1, 28, 28).to(dev) for h in synth.poisson_mnist_like(bench.BATCH, bench.T, 4, seed=1)]
torch.manual_seed(2)
for k in range(6):
    net.run({"X": pool[k % len(pool)]}, time=bench.T)
    net.reset_state_variables()
prof = _lib.profile_run(net, {"X": pool[0].clone()}, bench.T, repeats=n)
print(f"{os.path.basename(_lib.LIB_PATH)} flags {os.environ.get('SNN_DC_SPECFLAGS', '0')}: {prof['avg_ms'] * 1e3:.1f} us per launch over {prof['n']} launches, form {prof.get('resident_form')}, "
      f"{1e3 * prof['avg_ms'] / bench.T:.3f} us per timestep")
