#!/usr/bin/env python3
"""Developer aid: cProfile of the host side of bench steps (network.run + reset_state_variables) at the bench workload."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    net = bench.build_network(dev)
    from bindsnet_amd import synth
    pool = [torch.from_numpy(h).view(bench.T, bench.BATCH, 1, 28, 28).to(dev) for h in synth.poisson_mnist_like(bench.BATCH, bench.T, 2, seed=1)]
    torch.manual_seed(2)
    for k in range(5):
        net.run({"X": pool[k % 2]}, time=bench.T)
        net.reset_state_variables()
    pr = cProfile.Profile()
    pr.enable()
    for k in range(100):
        net.run({"X": pool[k % 2]}, time=bench.T)
        net.reset_state_variables()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
