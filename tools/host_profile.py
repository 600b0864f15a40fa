#!/usr/bin/env python3
"""Developer aid: where does the HOST time of one network.run() go?  (cProfile over the bench loop.)"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    net = bench.build_network(dev)
    pool = bench.make_inputs(1000, 2, dev)

    def one(k):
        torch.manual_seed(2 + k)
        net.run({"X": pool[k % 2]}, time=bench.T)
        net.reset_state_variables()

    for k in range(3):
        one(k)
    torch.cuda.synchronize()
    # wall-clock split: enqueue (python + launches) vs waiting for the device
    t0 = time.perf_counter()
    n = 20
    for k in range(n):
        one(k)
    torch.cuda.synchronize()
    print(f"wall per run: {(time.perf_counter() - t0) / n * 1e3:.3f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for k in range(n):
        one(k)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(18)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
