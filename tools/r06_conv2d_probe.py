#!/usr/bin/env python3
"""snn_prop_conv2d_f32 alone at the conv_mnist.py shape (B=16, 1 -> 32 filters 5x5, 28x28 -> 24x24): HIP-event time per call over
the input density, back-to-back calls.  python tools/r06_conv2d_probe.py"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bindsnet_amd import ops, synth
DEV = "cuda"
W = torch.from_numpy(synth.uniform_f32(1, (32, 1, 5, 5), 0.0, 0.3)).to(DEV)
for B in (16, 64):
    for dens in (0.0, 0.05, 0.5):
        s = torch.from_numpy(synth.dense_spikes(2, (B, 1, 28, 28), dens)).to(DEV)
        out = torch.empty(B, 32, 24, 24, device=DEV)
        for _ in range(10):
            ops.prop_conv2d(W, s, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ops.prop_conv2d(W, s, out)
        e1.record(); torch.cuda.synchronize()
        print(json.dumps({"B": B, "density": dens, "us_per_call": round(e0.elapsed_time(e1) * 1e3 / 200, 2)}), flush=True)
