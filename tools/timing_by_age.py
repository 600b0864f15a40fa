#!/usr/bin/env python3
"""Where does a step of k_dc2015_async go while the network is young (the driver's K=20 region: runs 5..24 of an untrained network) and
later?  Runs the bench's input sequence and switches the kernel's TIMING instance on (SNN_DC_TIMING=<workgroup>: its per-step report
goes to stderr) for the runs listed in --at.

    python tools/timing_by_age.py [--at 0,5,15,60] 2> report.txt
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bindsnet_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--at", default="0,5,15,60")
    ap.add_argument("--wg", default="3")
    a = ap.parse_args()
    at = sorted(int(x) for x in a.at.split(","))
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    T, B, N = 250, 32, 400
    pool = [torch.from_numpy(h).to("cuda") for h in synth.poisson_mnist_like(B, T, 4, seed=1)]
    last = [x[T - 1].clone() for x in pool]
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    mons = {l: Monitor(net.layers[l], ["s"], time=T) for l in ("X", "Ae", "Ai")}
    for l, m in mons.items():
        net.add_monitor(m, l + "_spikes")
    net.to("cuda")
    torch.manual_seed(2)
    for k in range(at[-1] + 1):
        if k in at:
            os.environ["SNN_DC_TIMING"] = a.wg
            sys.stderr.write(f"==== run {k}\n"); sys.stderr.flush()
        net.run({"X": pool[k % 4]}, time=T)
        torch.cuda.synchronize()
        if k in at:
            del os.environ["SNN_DC_TIMING"]
            sE = mons["Ae"].get("s")
            sys.stderr.write(f"     run {k}: {int(sE.sum())} Ae spikes ({float(sE.reshape(T, B, -1).any(2).float().sum(1).mean()):.2f} samples with a winner per step), "
                             f"theta mean {float(net.layers['Ae'].theta.mean()):.3f}\n"); sys.stderr.flush()
        net.reset_state_variables()
        pool[k % 4][T - 1].copy_(last[k % 4])


if __name__ == "__main__":
    main()
