#!/usr/bin/env python3
"""The other BASELINE.json configs (parity-test cases, not the headline bench line) on one MI355X through the public API, each to SURVEY.md
8(d)'s standard: one JSON object per config with

  value / ms_per_timestep   Network.run() + reset_state_variables() per input inside ONE pipelined section (Network.pipelined(); `sync`: the same
                            loop with a host sync per run), the monitors BASELINE.md section 2 states attached (a spike monitor on every layer);
  roofline                  SURVEY.md 8(d)'s algorithmic bytes per timestep / the dominant kernel's rocprofv3 duration of THIS command
                            (profiles/r05_roofline_summary_all_configs.json, written by tools/profile_round.sh) vs 8 TB/s, with the counter traffic;
  cpu_baseline              the UNMODIFIED reference (oracle/_ref, oracle/ref_cpu_leg.py --config) on 8 pinned host threads: median of 3 inputs
                            of T_cpu timesteps (20 for cfg3 / cfg5);
  parity                    the same 3 inputs on a fresh GPU network against that CPU run: rasters bit for bit, max |dW|.

    python tools/bench_configs.py [--runs 5] [--only cfg1,cfg3] [--no-cpu-baseline]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import baseline_configs as bc  # noqa: E402
from bindsnet_amd import synth  # noqa: E402

DEV = "cuda"


def timed(net, inputs, T, runs, **kw):
    """Wall clock of `runs` x (run + reset) -- inside one pipelined section, and with a host sync per run."""
    import contextlib
    out = {}
    for mode in ("pipelined", "sync"):
        with (net.pipelined() if mode == "pipelined" else contextlib.nullcontext()):
            for _ in range(2):
                net.run(dict(inputs), time=T, **kw); net.reset_state_variables()
            net.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(runs):
                net.run(dict(inputs), time=T, **kw); net.reset_state_variables()
            net.sync()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out[mode] = dt
    return {"timesteps_per_s": round(runs * T / out["pipelined"], 1), "ms_per_timestep": round(out["pipelined"] / (runs * T) * 1e3, 5),
            "sync": {"timesteps_per_s": round(runs * T / out["sync"], 1), "ms_per_timestep": round(out["sync"] / (runs * T) * 1e3, 5)},
            "plan": net.last_plan}


def reference_leg(name, xs, Tc, n):
    """oracle/ref_cpu_leg.py --config <name> in a process of its own -> (json, npz record) or (None / {"error"}, None)."""
    from oracle import stage_ref
    if not stage_ref.verify():
        return None, None
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.npy"), os.path.join(td, "rec.npz")
        np.save(fin, np.stack([x[:Tc] for x in xs[:n]]))
        try:
            res = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_cpu_leg.py"), "--inputs", fin, "--out", fout, "--config", name, "--whole", str(n)],
                                 capture_output=True, text=True, timeout=900, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
            out = json.loads(res.stdout.strip().splitlines()[-1])
        except Exception as e:                               # noqa: BLE001
            return {"error": f"reference leg failed: {str(e)[:300]}"}, None
        z = np.load(fout)
        return out, {k: z[k] for k in z.files}


def parity(name, xs, Tc, rec, kw):
    """The same inputs on a fresh GPU network, compared after every input with the reference's record."""
    ns = bc.namespace("bindsnet_amd")
    torch.manual_seed(0)
    net, mons = bc.build(name, ns, Tc)
    net.to(DEV)
    torch.manual_seed(2)
    ok, dW, wexact, spikes = True, 0.0, True, 0
    for k in range(int(rec["n_inputs"])):
        net.run({"X": torch.from_numpy(xs[k][:Tc].copy()).to(DEV)}, time=Tc, **kw)
        for lname, m in mons.items():
            if lname == "X":
                continue
            got = np.packbits(m.get("s").reshape(Tc, -1).cpu().numpy().astype(np.uint8))
            ok = ok and bool(np.array_equal(got, rec[f"r{k}_s_{lname}"]))
            spikes += int(np.unpackbits(rec[f"r{k}_s_{lname}"]).sum())
        for (src, dst), w in bc.learned_weights(net).items():
            ref = rec[f"r{k}_w_{src}_{dst}"]
            got = w.detach().cpu().numpy()
            dW = max(dW, float(np.abs(got - ref).max()))
            wexact = wexact and bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))
        net.reset_state_variables()
    return {"rasters_bit_exact": ok, "inputs": int(rec["n_inputs"]), "timesteps_per_input": Tc, "spikes": spikes, "max_abs_dW": dW, "weights_bit_exact": wexact,
            "plan": net.last_plan, "against": "the unmodified reference (oracle/_ref) on this host in this run, identical seeds; weights compared after every input"}


def roofline(name, T):
    """From the committed rocprofv3 summary of this same command (tools/profile_round.sh -> profiles/r05_roofline_summary_all_configs.json)."""
    for f in ("r06_roofline_summary_all_configs.json", "r05_roofline_summary_all_configs.json", "r04_roofline_summary_all_configs.json"):
        path = os.path.join(ROOT, "profiles", f)
        if os.path.exists(path):
            d = json.load(open(path)).get(name)
            if d and d.get("avg_launch_us"):
                ach = bc.CONFIGS[name]["algo_bytes"] * d["timesteps_per_launch"] / (d["avg_launch_us"] * 1e-6) / 1e9
                return {"bound": "hbm", "kernel": d.get("kernel"), "avg_launch_us": d["avg_launch_us"], "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(ach / 8000.0, 4), "traffic": d.get("hbm_bytes_per_launch_gfx950_corrected"), "source": "profiles/" + f}
    return None


def baseline_config(name, runs, cpu):
    c = bc.CONFIGS[name]
    ns = bc.namespace("bindsnet_amd")
    torch.manual_seed(0)
    net, _ = bc.build(name, ns, c["T"])
    net.to(DEV)
    xs, xname = bc.inputs(name, 3)
    x = torch.from_numpy(xs[0]).to(DEV)
    torch.manual_seed(2)
    r = timed(net, {xname: x}, c["T"], runs, **c["kw"])
    line = {"metric": "simulated timesteps/sec, " + c["what"], "value": r["timesteps_per_s"], "unit": "timesteps/s", "ms_per_timestep": r["ms_per_timestep"],
            "sync_runs": r["sync"], "n_gpus": 1, "runs": runs, "dtype": "f32", "data": "synthetic (BASELINE.md section 2's generator)", "config": {"workload": name + ": " + c["what"], "plan": r["plan"],
            "monitors": "a spike monitor on every layer", "host_sync": "pipelined section"}, "roofline": roofline(name, c["T"])}
    if cpu:
        ref, rec = reference_leg(name, xs, c["T_cpu"], c.get("n_cpu", 3))      # (cfg3 at B = 128 / cfg5: ~0.5 timesteps/s on the CPU -> one input of 20)
        if ref is not None and "error" not in ref and rec is not None:
            line["cpu_baseline"] = {"value": ref["median"], "unit": "timesteps/s", "cores": ref["threads"], "kind": "reference",
                                    "sample": f"median of {ref['inputs']} inputs of {ref['timesteps_per_input']} timesteps (batch {c['B']}) through the unmodified reference, Network.run() in a subprocess",
                                    "min": ref["min"], "max": ref["max"], "per_input_timesteps_per_s": ref["per_input_timesteps_per_s"], "affinity": ref["affinity"]}
            line["parity"] = parity(name, xs, c["T_cpu"], rec, c["kw"])
            line["speedup_vs_cpu_baseline"] = round(line["value"] / ref["median"], 1)
        else:
            line["cpu_baseline"] = ref
    return line


def cfg1(): return "cfg1"            # noqa: E704  (names kept for --only and tools/profile_round.sh)
def cfg3_shard(): return "cfg3_shard"   # noqa: E704
def cfg3_b32(): return "cfg3_b32"    # noqa: E704
def cfg3(): return "cfg3"            # noqa: E704
def cfg4(): return "cfg4"            # noqa: E704
def cfg5(): return "cfg5"            # noqa: E704


def _rule_two_layer(rule_name, B=32, Nin=784, N=1600, T=100):
    """Input -> Connection(rule) -> LIFNodes at cfg3's width (one GPU's share of the batch)."""
    from bindsnet_amd import learning
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    torch.manual_seed(0)
    net = Network(dt=1.0)
    net.add_layer(Input(n=Nin, traces=True), "X")
    net.add_layer(LIFNodes(n=N, traces=True), "Y")
    net.add_connection(Connection(net.layers["X"], net.layers["Y"], w=0.3 * torch.rand(Nin, N), wmin=0.0, wmax=1.0,
                                  update_rule=getattr(learning, rule_name), nu=(1e-4, 1e-2), norm=78.4, reduction=torch.sum), "X", "Y")
    net.to(DEV)
    x = torch.from_numpy(synth.dense_spikes(2, (T, B, Nin), 0.012)).to(DEV)
    return f"(f) {rule_name} 784->1600 B={B} T={T}", net, {"X": x}, T, {}


def f_hebbian():
    return _rule_two_layer("Hebbian")


def f_wdpp():
    return _rule_two_layer("WeightDependentPostPre")


def f_postpre_ref():
    return _rule_two_layer("PostPre")            # the same graph with PostPre, for comparison


def f_conv_postpre():
    """conv_mnist.py's training graph: Input -> Conv2dConnection(PostPre) -> LIFNodes (plan convpp-fused; SNN_CONVPP_FUSED=0: the generic plan)."""
    from bindsnet_amd.learning import PostPre
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Conv2dConnection
    torch.manual_seed(0)
    net = Network(dt=1.0)
    net.add_layer(Input(shape=(1, 28, 28), traces=True), "X")
    net.add_layer(LIFNodes(shape=(32, 24, 24), traces=True), "Y")
    net.add_connection(Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=5, stride=1, update_rule=PostPre, nu=(1e-4, 1e-2),
                                        wmin=0.0, wmax=1.0, reduction=torch.sum, w=0.3 * torch.rand(32, 1, 5, 5)), "X", "Y")
    net.to(DEV)
    x = torch.from_numpy(synth.dense_spikes(3, (100, 16, 1, 28, 28), 0.05)).to(DEV)
    return "(f) Conv2d 5x5x32 PostPre -> LIF B=16 T=100", net, {"X": x}, 100, {}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    for make in (cfg1, cfg3_shard, cfg3_b32, cfg3, cfg4, cfg5, f_postpre_ref, f_hebbian, f_wdpp, f_conv_postpre):
        if a.only and make.__name__ not in a.only.split(','):
            continue
        if make.__name__.startswith("cfg"):
            print(json.dumps(baseline_config(make(), a.runs, not a.no_cpu_baseline)), flush=True)
            continue
        name, net, inputs, T, kw = make()
        r = timed(net, inputs, T, a.runs, **kw)
        r["config"] = name
        print(json.dumps(r), flush=True)
