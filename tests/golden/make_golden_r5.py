#!/usr/bin/env python3
"""Round-5 fixtures from the UNMODIFIED reference (build container only):

    python tests/golden/make_golden_r5.py

 run_lif_vector_thresh.npz   LIFNodes with PER-NEURON thresholds (nodes.py:425-498 take `thresh` as a tensor; examples/mnist/reservoir.py builds its
                             layer that way): Input(96) -> MulticompartmentConnection + Weight -> LIFNodes(70, thresh = a [70] tensor, traces) with a
                             recurrent MCC connection; two consecutive runs (batch 5, T = 60, no reset between): spike raster, voltage raster,
                             final trace.  (MCC + Weight: ATen's cascade order, reproducible bit for bit -- a dense Connection goes through MKL.)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402
from make_golden import Input, LIFNodes, Monitor, MulticompartmentConnection, Network, T_, Weight, save  # noqa: E402

n_in, n_out, B, T = 96, 70, 5, 60


def thresholds():
    return (-60.0 + 12.0 * synth.uniform_f32(9, (n_out,), 0.0, 1.0)).astype(np.float32)


def weights():
    return synth.uniform_f32(5, (n_in, n_out), 0.0, 2.5), synth.uniform_f32(6, (n_out, n_out), -0.5, 0.5)


def main():
    net = Network(dt=1.0, learning=False)
    net.add_layer(Input(n=n_in), "I")
    net.add_layer(LIFNodes(n=n_out, thresh=T_(thresholds()), refrac=2, tc_decay=50.0, traces=True), "O")
    w_in, w_rec = weights()
    net.add_connection(MulticompartmentConnection(net.layers["I"], net.layers["O"], device="cpu", pipeline=[Weight("weight", T_(w_in).clone())]), "I", "O")
    net.add_connection(MulticompartmentConnection(net.layers["O"], net.layers["O"], device="cpu", pipeline=[Weight("weight", T_(w_rec).clone())]), "O", "O")
    mon = Monitor(net.layers["O"], ["s", "v"], time=T)
    net.add_monitor(mon, "O")
    out = {}
    for r in range(2):
        sp = synth.dense_spikes(17 + r, (T, B, n_in), 0.08)
        net.run({"I": T_(sp)}, time=T)
        out[f"r{r}_s"] = np.packbits(mon.get("s").numpy().astype(np.uint8))
        out[f"r{r}_v"] = mon.get("v").numpy().copy()
        out[f"r{r}_x"] = net.layers["O"].x.numpy().copy()
        per = mon.get("s").reshape(-1, n_out).sum(0)
        print(f"  run {r}: {int(mon.get('s').sum())} spikes, per neuron {int(per.min())}..{int(per.max())}")
    save("run_lif_vector_thresh", n_in=n_in, n_out=n_out, B=B, T=T, thresh=thresholds(), **out)


if __name__ == "__main__":
    main()
