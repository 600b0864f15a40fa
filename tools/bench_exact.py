#!/usr/bin/env python3
"""Wall clock of the exact batch-sharded mode (bindsnet_amd.parallel.exact_run) on BASELINE cfg2's stated input: `--world`
processes (gloo between them) on ONE device -- the only multi-rank arrangement a 1-GPU box offers -- each with B / world rows
of the global batch of 32; every run is checked against the reference's single-process fixture before its time counts.

    python tools/bench_exact.py --device cuda --worlds 2 1 > gpurun_out/exact_mode.jsonl      (one JSON line per world size)"""
import argparse
import json
import os
import pathlib
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import exact_harness as H  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--worlds", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--fixture", default="full_cfg2_dc_n400_b32_poisson")
    ap.add_argument("--timeout", type=int, default=90, help="seconds per world size")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as d:
        for w in a.worlds:                                     # one JSON line per world size, flushed at once: a later size that hangs
            try:                                               # or times out does not take the earlier figures with it
                res = H.launch(w, a.fixture, a.device, pathlib.Path(d), timeout=a.timeout)
                H.check_against_reference(res, a.fixture)
            except BaseException as e:                         # noqa: BLE001
                print(json.dumps({"world": w, "device": a.device, "failed": f"{type(e).__name__}: {str(e)[-400:]}"}), flush=True)
                continue
            runs = len([k for k in res[0].files if k.endswith("_seconds")])
            secs = [max(float(x[f"r{r}_seconds"]) for x in res) for r in range(runs)]      # slowest rank per input
            T = 250
            print(json.dumps({"world": w, "device": a.device, "fixture": a.fixture, "seconds_per_input": [round(s, 4) for s in secs],
                              "timesteps_per_s_steady": round(T / min(secs[1:] or secs), 1),
                              "parity": "bit-exact vs the reference's global batch"}), flush=True)


if __name__ == "__main__":
    main()
