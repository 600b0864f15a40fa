#!/usr/bin/env python3
"""CPU-baseline leg of bench.py on the REAL reference (MEASUREMENT INFRASTRUCTURE ONLY; run as a subprocess by bench.py:cpu_baseline).

Imports the unmodified BindsNET packages staged under oracle/_ref/bindsnet (oracle/stage_ref.py; byte copies, sha256-checked against
oracle/ref_manifest.json before anything runs) through SURVEY.md Appendix C's stub recipe -- in a process of its own, because the repository's
`bindsnet` alias and the reference cannot both be `bindsnet` in one interpreter, and so that no HIP runtime thread competes for the cores.

The workload is bench.py's: examples/mnist/eth_mnist.py:91-100's DiehlAndCook2015(784 -> N, exc 22.5, inh 120, norm 78.4, theta_plus 0.05), its
three spike monitors (:143-148), network.run(inputs={"X": [T, B, 1, 28, 28]}, time=T) + reset_state_variables() per input
(bindsnet/network/network.py:380-465, :467-481), weights from torch.manual_seed(0), one_spike noise from torch.manual_seed(2).

  python oracle/ref_cpu_leg.py --inputs in.npy --out rec.npz [--n 400 --whole 5 --threads 8]

in.npy: uint8 [n_inputs, T, B, 784].  Prints ONE JSON object; rec.npz holds, per whole input of the 8-thread leg, the Ae / Ai rasters
(bit-packed), W and theta after it -- what bench.py's parity leg compares the GPU against.
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def load_reference():
    from oracle import stage_ref
    if not stage_ref.verify():
        raise SystemExit("oracle/_ref does not hold the reference recorded in oracle/ref_manifest.json")
    pkg = types.ModuleType("bindsnet")
    pkg.__path__ = [stage_ref.STAGED_ROOT]
    sys.modules["bindsnet"] = pkg                        # (skips bindsnet/__init__.py: torchvision, gymnasium ... are absent)
    import bindsnet.network                              # noqa: F401  first: learning <-> topology_features import cycle
    from bindsnet.models import DiehlAndCook2015
    from bindsnet.network.monitors import Monitor
    return DiehlAndCook2015, Monitor


def config_leg(args):
    """--config <name>: one of the OTHER BASELINE.md configs (tools/baseline_configs.py builds it from the reference's classes): `--whole`
    consecutive inputs of T_cpu timesteps each (20 for cfg3 / cfg5) through Network.run() + reset_state_variables(),
    8 threads pinned to one CCD; the record holds, per input, every non-input layer's raster (bit-packed) and the weights after it."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import baseline_configs as bc
    ncpu = os.cpu_count() or 2
    pin = None
    if hasattr(os, "sched_setaffinity"):
        try:
            pin = sorted(os.sched_getaffinity(0))[:8]
            torch.set_num_threads(min(args.threads, len(pin)))
            os.sched_setaffinity(0, pin)
        except OSError:
            pin = None
    load_reference()
    ns = bc.namespace("bindsnet")
    cfg = bc.CONFIGS[args.config]
    x = np.load(args.inputs)                                  # [n_inputs, T_cpu, B, ...]
    Tc = x.shape[1]
    th = min(args.threads, ncpu)
    torch.set_num_threads(th)
    torch.manual_seed(0)
    net, mons = bc.build(args.config, ns, Tc)
    torch.manual_seed(2)
    rates, recs = [], {}
    for k in range(min(args.whole, x.shape[0])):
        xin = torch.from_numpy(x[k]).clone()
        t0 = time.perf_counter()
        net.run(inputs={"X": xin}, time=Tc, **cfg["kw"])
        rates.append(Tc / (time.perf_counter() - t0))
        for lname, m in mons.items():
            if lname != "X":
                recs[f"r{k}_s_{lname}"] = np.packbits(m.get("s").reshape(Tc, -1).numpy().astype(np.uint8))
        for (src, dst), w in bc.learned_weights(net).items():
            recs[f"r{k}_w_{src}_{dst}"] = w.detach().numpy().copy()
        net.reset_state_variables()
    out = {"kind": "reference", "config": args.config, "threads": th, "timesteps_per_input": Tc, "inputs": len(rates),
           "per_input_timesteps_per_s": [round(v, 3) for v in rates], "min": round(min(rates), 3), "median": round(float(np.median(rates)), 3),
           "max": round(max(rates), 3), "host_cpus": ncpu, "torch": torch.__version__, "affinity": (f"pinned to CPUs {pin}" if pin else "not pinned")}
    np.savez(args.out, n_inputs=len(rates), **recs)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inputs", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--n", type=int, default=400)
    ap.add_argument("--whole", type=int, default=5)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--short-legs", type=int, default=1)
    ap.add_argument("--config", default="", help="one of tools/baseline_configs.py's names instead of the headline D&C workload")
    args = ap.parse_args()
    if args.config:
        return config_leg(args)

    ncpu = os.cpu_count() or 2
    aff_all = pin = None
    if hasattr(os, "sched_setaffinity"):
        try:
            aff_all = os.sched_getaffinity(0)
            pin = sorted(aff_all)[:8]
            torch.set_num_threads(min(args.threads, len(pin)))
            os.sched_setaffinity(0, pin)
        except OSError:
            aff_all = pin = None
    DiehlAndCook2015, Monitor = load_reference()
    x = np.load(args.inputs)
    n_in, T, B, Nin = x.shape
    spikes = [torch.from_numpy(x[k]).view(T, B, 1, 28, 28) for k in range(n_in)]

    def fresh():
        torch.manual_seed(0)
        net = DiehlAndCook2015(n_inpt=Nin, n_neurons=args.n, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
        mons = {}
        for l in ("X", "Ae", "Ai"):
            mons[l] = Monitor(net.layers[l], ["s"], time=T)
            net.add_monitor(mons[l], l + "_spikes")
        torch.manual_seed(2)
        return net, mons

    def leg(threads, n_steps, warm, n_inputs, keep):
        torch.set_num_threads(threads)
        net, mons = fresh()
        rates, recs = [], {}
        for k in range(n_inputs):
            sp = spikes[k % n_in]
            if warm:
                net.run(inputs={"X": sp[:warm].clone()}, time=warm)
            xin = sp[warm:warm + n_steps].clone()           # (Input.s aliases the last slice and reset zeroes it: keep the pool intact)
            t0 = time.perf_counter()
            net.run(inputs={"X": xin}, time=n_steps)
            rates.append(n_steps / (time.perf_counter() - t0))
            if keep:
                for l in ("Ae", "Ai"):
                    recs[f"r{k}_{l}"] = np.packbits(mons[l].get("s").reshape(T, B, args.n).numpy().astype(np.uint8))
                recs[f"r{k}_W"] = net.connections[("X", "Ae")].pipeline[0].value.detach().numpy().copy()
                recs[f"r{k}_theta"] = net.layers["Ae"].theta.numpy().copy()
            net.reset_state_variables()
        return rates, recs

    th = min(args.threads, ncpu)
    r8, recs = leg(th, T, 0, args.whole, True)
    out = {"kind": "reference", "threads": th, "per_input_timesteps_per_s": [round(v, 2) for v in r8],
           "min": round(min(r8), 2), "median": round(float(np.median(r8)), 2), "max": round(max(r8), 2),
           "inputs": args.whole, "host_cpus": ncpu, "torch": torch.__version__,
           "affinity": (f"pinned to CPUs {pin}" if pin else "not pinned")}
    if args.short_legs:
        r1, _ = leg(1, min(100, T), 0, 3, False)
        out["one_thread"] = {"value": round(float(np.median(r1)), 2), "per_sample": [round(v, 2) for v in r1], "cores": 1,
                             "sample": "median over the first 100 timesteps of 3 inputs"}
        if aff_all is not None:
            os.sched_setaffinity(0, aff_all)                 # eth_mnist.py's own setting is unpinned
        nd = max(1, ncpu - 1)
        rd, _ = leg(nd, 3, 1, 3, False)
        out["reference_default_threads"] = {"threads": nd, "value": round(float(np.median(rd)), 2), "per_sample": [round(v, 2) for v in rd],
                                            "note": "torch.set_num_threads(os.cpu_count() - 1), eth_mnist.py:77; median of 3 samples of 3 timesteps after 1 untimed one"}
    np.savez(args.out, n_inputs=args.whole, **recs)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
