#!/usr/bin/env python3
"""Connection.compute two ways on one MI355X: the event-driven kernel the plans use (snn_prop_dense_f32: only the
rows that spiked are read) vs the f32-MFMA GEMM (snn_prop_dense_mfma_f32: one k-ordered chain per 16x16 tile), both
bit-identical to the canonical ordered sum.  Prints one JSON object per shape; with --only-mfma / --only-event it just
loops one kernel (for the rocprofv3 --pmc passes of tools/profile_dense_prop.sh).

    python tools/bench_dense_prop.py [--iters 200]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bindsnet_amd import synth  # noqa: E402
from bindsnet_amd import ops  # noqa: E402

DEV = "cuda"
SHAPES = [  # name, B, Nin, N, input density
    ("cfg3 per-GPU shard", 16, 784, 1600, 0.012),
    ("cfg3 batch 32", 32, 784, 1600, 0.012),
    ("cfg3 whole batch", 128, 784, 1600, 0.012),
    ("cfg5", 16, 6400, 500, 0.05),
    ("dense input, 50 % active", 128, 784, 1600, 0.5),
]


def timed(fn, iters):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters          # us per call


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--only-mfma", action="store_true")
    ap.add_argument("--only-event", action="store_true")
    ap.add_argument("--density", type=float, default=None, help="the two learning shapes (cfg3 whole batch, cfg5) at THIS input density instead of the stated ones "
                    "(tools/r06_density_map.sh: the MFMA-vs-event map over 1 / 5 / 20 / 50 %)")
    a = ap.parse_args()
    if a.density is not None:
        SHAPES = [("cfg3 whole batch", 128, 784, 1600, a.density), ("cfg5", 16, 6400, 500, a.density)]
    for name, B, Nin, N, dens in SHAPES:
        W = torch.from_numpy(synth.uniform_f32(1, (Nin, N), 0.0, 0.3)).to(DEV)
        s = torch.from_numpy(synth.dense_spikes(2, (B, Nin), dens)).to(DEV)
        o1, o2 = torch.empty(B, N, device=DEV), torch.empty(B, N, device=DEV)
        r = {"shape": name, "B": B, "Nin": Nin, "N": N, "input_density": dens}
        if not a.only_mfma:
            r["event_driven_us"] = round(timed(lambda: ops.prop_dense(W, s, o1), a.iters), 2)
        if not a.only_event:
            r["mfma_us"] = round(timed(lambda: ops.prop_dense_mfma(W, s, o2), a.iters), 2)
            flops = 2.0 * B * Nin * N
            r["mfma_TFLOPs"] = round(flops / (r["mfma_us"] * 1e-6) / 1e12, 3)
            r["mfma_chain_floor_us"] = round((Nin / 4) * 40 / 2400.0, 2)     # Nin/4 dependent MFMAs x 40 cycles @ 2.4 GHz
        if not a.only_mfma and not a.only_event:
            r["identical_bits"] = bool(torch.equal(o1.view(torch.int32), o2.view(torch.int32)))
            r["hbm_bytes_algorithmic"] = 4 * Nin * N + B * (Nin + 4 * N)
            r["event_driven_GBps"] = round(r["hbm_bytes_algorithmic"] / (r["event_driven_us"] * 1e-6) / 1e9, 1)
            r["mfma_GBps"] = round(r["hbm_bytes_algorithmic"] / (r["mfma_us"] * 1e-6) / 1e9, 1)
        print(json.dumps(r), flush=True)
