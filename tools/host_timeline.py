#!/usr/bin/env python3
"""Developer aid: timeline of one bench step (network.run + reset_state_variables) on the host.

Wraps the C-ABI call and the generator hand-over with perf_counter stamps (no profiler overhead) and prints the
average microseconds spent in: python before the C call | snn_net_run (enqueue) | waiting for the device in the
blocking read-back | python after it | reset_state_variables.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bindsnet_amd import _lib, rng  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    net = bench.build_network(dev)
    from bindsnet_amd import synth
    pool = [torch.from_numpy(h).view(bench.T, bench.BATCH, 1, 28, 28).to(dev) for h in synth.poisson_mnist_like(bench.BATCH, bench.T, 2, seed=1)]
    last = [x[bench.T - 1].clone() for x in pool]
    L = _lib.lib()
    stamps = {}
    real_run = L.snn_net_run

    class Wrapped:
        argtypes, restype = real_run.argtypes, real_run.restype

        def __call__(self, *a):
            stamps["c0"] = time.perf_counter()
            r = real_run(*a)
            stamps["c1"] = time.perf_counter()
            return r

    L.snn_net_run = Wrapped()
    real_finish = rng.DeviceGenerator.finish

    def finish(self, *a, **k):
        stamps["f0"] = time.perf_counter()
        orig = self._host

        class Probe:                                       # stands in for the pinned staging tensor during the read-back
            def copy_(_, src, **kw):
                out = orig.copy_(src, **kw)
                stamps["f1"] = time.perf_counter()
                return out

            def numpy(_):
                return orig.numpy()
        self._host = Probe()
        try:
            return real_finish(self, *a, **k)
        finally:
            self._host = orig

    rng.DeviceGenerator.finish = finish
    torch.manual_seed(2)
    acc = {k: 0.0 for k in ("pre", "enqueue", "between", "wait", "post", "reset")}
    n = 50
    for k in range(n + 5):
        t0 = time.perf_counter()
        net.run({"X": pool[k % 2]}, time=bench.T)
        t1 = time.perf_counter()
        net.reset_state_variables()
        pool[k % 2][bench.T - 1].copy_(last[k % 2])
        t2 = time.perf_counter()
        if k >= 5:
            acc["pre"] += stamps["c0"] - t0
            acc["enqueue"] += stamps["c1"] - stamps["c0"]
            acc["between"] += stamps["f0"] - stamps["c1"]
            acc["wait"] += stamps["f1"] - stamps["f0"]
            acc["post"] += t1 - stamps["f1"]
            acc["reset"] += t2 - t1
    torch.cuda.synchronize()
    total = sum(acc.values())
    print("per bench step, us: " + " | ".join(f"{k} {v / n * 1e6:.1f}" for k, v in acc.items()) + f" | total {total / n * 1e6:.1f}")


if __name__ == "__main__":
    main()
