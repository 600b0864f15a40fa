#!/usr/bin/env python3
"""Does the REFERENCE agree with itself across torch thread counts?  (build container only: imports /root/reference)

Runs the unmodified reference's DiehlAndCook2015 at cfg1's stated size and input (N = 100, batch 1, T = 250, three consecutive
reference-encoded Poisson inputs: tests/golden/full_cfg1_dc_n100_b1_poisson.npz) once per thread count and compares every run
with the committed fixture (made at 8 threads = the serial summation order).  ATen sums the last
N mod 32 = 4 columns of every propagation / normalisation in another order at 9 or >= 12 threads
(tools/probe_aten_sum_threads.py); this shows what that does to a whole run.

    python tools/ref_thread_divergence.py --threads 8 16 > profiles/r03_reference_thread_divergence_cfg1.json"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests"), ROOT]
import cases  # noqa: E402
import make_golden as mg  # noqa: E402  (the reference through the Appendix C recipe)


def run(threads, g):
    torch.set_num_threads(threads)
    N, B, T, runs = int(g["N"]), int(g["B"]), int(g["T"]), int(g["runs"])
    torch.manual_seed(0)
    net = mg.DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    feat = net.connections[("X", "Ae")].pipeline[0]
    mon = mg.Monitor(net.layers["Ae"], ["s"], time=T)
    net.add_monitor(mon, "Ae_s")
    torch.manual_seed(2)
    out = []
    for r in range(runs):
        spikes = cases.fixture_input(g, r, T, B)
        net.run({"X": mg.T_(spikes).view(T, B, 1, 28, 28)}, time=T)
        ras = mon.get("s").numpy().reshape(T, B, N).astype(np.uint8)
        ref = cases.unpack(g[f"r{r}_sE"], (T, B, N))
        W = feat.value.detach().numpy()
        diff = np.nonzero((ras != ref).reshape(T, -1).any(1))[0]
        sample = W.reshape(-1)[::97]
        out.append({"input": r, "exc_spikes": int(ras.sum()), "raster_equals_fixture": bool((ras == ref).all()),
                    "first_differing_timestep": int(diff[0]) if diff.size else None, "raster_bits_differing": int((ras != ref).sum()),
                    "weights_sha_equals_fixture": cases.sha(W) == str(g[f"r{r}_W_sha"]),
                    "weights_sample_max_abs_diff": float(np.abs(sample - g[f"r{r}_W_sample"]).max()),
                    "weights_sample_differing": int((sample.view(np.uint32) != g[f"r{r}_W_sample"].view(np.uint32)).sum()),
                    "theta_max_abs_diff": float(np.abs(net.layers["Ae"].theta.numpy() - g[f"r{r}_theta"]).max())})
        net.reset_state_variables()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, nargs="+", default=[8, 16])
    a = ap.parse_args()
    g = cases.gold("full_cfg1_dc_n100_b1_poisson")
    print(json.dumps({"what": "the unmodified reference, DiehlAndCook2015 784->100, batch 1, 3 x 250 timesteps of the stated input, per torch thread "
                              "count, against the committed fixture (made at 8 threads)", "host_cpus": os.cpu_count(),
                      "runs": {str(t): run(t, g) for t in a.threads}}, indent=1))


if __name__ == "__main__":
    main()
