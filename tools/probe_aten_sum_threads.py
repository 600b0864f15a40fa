#!/usr/bin/env python3
"""The reference's float sums are NOT independent of torch's intra-op thread count -- where, and by how much.

`MulticompartmentConnection.compute` reduces [B, Nin, N] over Nin with torch.sum(dim=1) (topology.py:469-471); Weight.normalize
reduces [Nin, N] over Nin (topology_features.py:264).  ATen parallelises such a reduction over a NON-reduced dimension: the
outermost one that has at least `threads` entries -- the batch when B >= threads -- else the contiguous columns, cut into
`threads` ranges whose ends are rounded down to multiples of 32 columns (TensorIteratorReduce.cpp: parallel_dim_reduction /
round_columns).  A range that ends up holding ONLY the last N mod 32 columns, fewer than a vector (8), is summed by
scalar_outer_sum -- groups of four columns in the cascade order of a full group -- while the same columns at the end of a longer
range are vectorized_outer_sum's leftover: row_sum order (SumKernel.cpp).  Different roundings, different bits.

Model (this script checks it against torch for every thread count it is given):
    c = ceil(N / threads);   the tail [32*floor(N/32), N) is isolated  <=>  c * floor((N - 1) / c) >= 32 * floor(N / 32)
    (and it matters only when the reduction runs in parallel at all -- B * Nin * N >= 32768 --, when the columns are what is split
    -- B < threads, and N >= threads or N > B -- and when 0 < N mod 32 < 8)

    python tools/probe_aten_sum_threads.py [--threads 1 2 ... ] [--shapes B,Nin,N ...]"""
import argparse

import numpy as np
import torch


def tail_isolated(B: int, N: int, threads: int, Nin: int = 1 << 20) -> bool:
    """The model, as the host path uses it (bindsnet_amd/network/host_path.py::aten_sum_leaves_serial_order)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bindsnet_amd.network.host_path import aten_sum_leaves_serial_order
    return aten_sum_leaves_serial_order(B, Nin, N, threads)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, nargs="+", default=list(range(1, 21)) + [32, 64, 128, 255])
    ap.add_argument("--shapes", nargs="+", default=["1,784,100", "3,784,100", "16,784,100", "2,1000,37", "4,784,400", "1,784,68"])
    a = ap.parse_args()
    n0 = torch.get_num_threads()
    torch.manual_seed(0)
    for shape in a.shapes:
        B, Nin, N = (int(v) for v in shape.split(","))
        W = torch.rand(Nin, N) - 0.5
        s = (torch.rand(B, Nin) < 0.3).to(torch.uint8)
        x = s.view(B, Nin, 1).repeat(1, 1, N) * W
        torch.set_num_threads(1)
        serial = x.sum(1).numpy().view(np.uint32).copy()
        wrong = []
        changed = []
        for t in a.threads:
            torch.set_num_threads(t)
            got = x.sum(1).numpy().view(np.uint32)
            differs = bool((got != serial).any())
            cols = sorted(set(np.nonzero(got != serial)[1].tolist()))
            if differs:
                changed.append((t, cols))
            if differs != tail_isolated(B, N, t, Nin) and not (tail_isolated(B, N, t, Nin) and not differs):
                wrong.append(t)                      # (an isolated tail may still round to the same bits: not a model error)
        print(f"[B={B}, Nin={Nin}, N={N}] differs from the serial order at threads {[t for t, _ in changed]} "
              f"in columns {sorted(set(c for _, cs in changed for c in cs))}; model wrong at: {wrong or 'none'}")
    torch.set_num_threads(n0)


if __name__ == "__main__":
    main()
