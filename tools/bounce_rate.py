#!/usr/bin/env python3
"""How often does the lean form of the D&C plan hand an input to the general form on digit-like images?

The lean forms give up (SNN_ERR_RETRY, nothing written, the input repeated on the general resident kernel: 44.5 k instead of ~250 k
timesteps/s) on a batch in which ANY sample has more than 63 input events in one timestep.  BASELINE.md's stated generator scatters
128*U*Bernoulli(0.19) pixels (9 events per sample-step, max 24); real MNIST digits are connected, mostly saturated strokes.  This tool
runs `--batches` batches of 32 stroke images (bindsnet_amd/synth.stroke_digit: ~160 lit pixels, ~100 of them saturated -- MNIST's
statistics) Poisson-encoded at `--intensity` (eth_mnist.py: 128) through the cfg2 network inside one pipelined section and reports
the bounce rate and the event statistics.  One JSON line.

    python tools/bounce_rate.py [--batches 4] [--intensity 128]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bindsnet_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--intensity", type=float, default=128.0)
    a = ap.parse_args()
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    T, B, N = 250, 32, 400
    trains = synth.poisson_mnist_like(B, T, a.batches, seed=3, intensity=a.intensity, strokes=True)
    per = np.stack([t.reshape(T, B, 784).sum(2) for t in trains])                  # [batch, T, B] events per sample-step
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    for l in ("X", "Ae", "Ai"):
        net.add_monitor(Monitor(net.layers[l], ["s"], time=T), l + "_spikes")
    net.to("cuda")
    xs = [torch.from_numpy(t).to("cuda") for t in trains]
    torch.manual_seed(2)
    plans = []
    t0 = time.perf_counter()
    with net.pipelined():
        for x in xs:
            net.run({"X": x}, time=T)
            plans.append(net.last_plan)
            net.reset_state_variables()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bounced = int(getattr(net, "lean_retries", 0))
    over = [int((per[k] > 63).any()) for k in range(a.batches)]
    print(json.dumps({"what": "lean-form bounce rate on digit-like stroke images", "samples": a.batches * B, "batches": a.batches, "intensity": a.intensity,
                      "events_per_sample_timestep": {"mean": round(float(per.mean()), 2), "p99": int(np.percentile(per, 99)), "max": int(per.max())},
                      "lit_pixels_per_image_mean": round(float(np.mean([(synth.stroke_digit(7919 * 3 + 131 * k + b) > 0).sum() for k in range(a.batches) for b in range(B)])), 1),
                      "batches_with_a_sample_over_63_events": int(sum(over)), "batches_repeated_on_the_general_form": bounced,
                      "bounce_rate": round(bounced / a.batches, 4), "timesteps_per_s_incl_first_run_setup": round(a.batches * T / dt, 1), "plans_requested": plans}))


if __name__ == "__main__":
    main()
