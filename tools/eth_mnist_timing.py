#!/usr/bin/env python3
"""Where the wall time of the LITERAL examples/mnist/eth_mnist.py goes, per sample.

    python tools/eth_mnist_timing.py --impl amd [--n_train 40 --n_test 10]     (MI355X box: `bindsnet` = this package)
    python tools/eth_mnist_timing.py --impl ref [...]                          (build container: the reference, CPU path)

The script file is executed unmodified through runpy (the reference checkout's file, or the byte copy
__graft_entry__.build() stages under tests/_staged/); MNIST is the synthetic stand-in of tests/tv_shim.py (no
network here), matplotlib runs on the Agg backend (the script ALWAYS plots: parser.set_defaults(plot=True),
eth_mnist.py:45).  Timers are wrappers around the callables the script uses -- nothing in the script changes:

  encode      PoissonEncoder.__call__ (host Poisson encoding inside the DataLoader, eth_mnist.py:103-112)
  h2d         Tensor.cuda() of the encoded sample (:183-184)
  run         Network.run (:243, :312), device-synchronised
  monitor_get Monitor.get (7 monitors registered, :131-153; 2 + 1 (+3 when plotting) reads per sample)
  evaluation  all_activity / proportion_weighting / assign_labels (:190-236, :320-340)
  plotting    the six bindsnet.analysis.plotting calls + get_square_* + plt.pause (:252-274)
  reset       Network.reset_state_variables (:276)

One JSON object on stdout (and --out FILE)."""
import argparse
import hashlib
import json
import os
import runpy
import sys
import time
import types
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_SCRIPT = "/root/reference/examples/mnist/eth_mnist.py"
STAGED = os.path.join(ROOT, "tests", "_staged", "eth_mnist.py")
REF_SHA = "ba4f875d097536bf1d935070112f1ad7fdffe50c190d2e9a9de1e0b278fefe4a"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impl", choices=["amd", "ref"], required=True)
    ap.add_argument("--n_train", type=int, default=40)
    ap.add_argument("--n_test", type=int, default=10)
    ap.add_argument("--n_neurons", type=int, default=100)
    ap.add_argument("--update_interval", type=int, default=20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--encode-device", default=None, help="amd: SNN_ENCODE_DEVICE for the run -- PoissonEncoder(time, dt), as the script builds it, "
                    "then encodes on that device (the specified Philox stream, not the host generator's)")
    args = ap.parse_args()
    if args.encode_device:
        os.environ["SNN_ENCODE_DEVICE"] = args.encode_device

    os.environ["MPLBACKEND"] = "Agg"
    import matplotlib
    matplotlib.use("Agg", force=True)
    import tv_shim
    if args.impl == "ref":
        REF = "/root/reference/bindsnet"
        for name, path in (("bindsnet", REF), ("bindsnet.analysis", REF + "/analysis")):
            pkg = types.ModuleType(name)
            pkg.__path__ = [path]
            sys.modules[name] = pkg
        sys.modules["cv2"] = types.ModuleType("cv2")
        tv_shim.install()
        import bindsnet.network  # noqa: F401
    else:
        sys.path.insert(0, ROOT)
        tv_shim.install()
        import bindsnet  # noqa: F401  (alias package -> bindsnet_amd)
    import torch
    import bindsnet.analysis.plotting as plotting
    import bindsnet.encoding as encoding
    import bindsnet.evaluation as evaluation
    import bindsnet.network.monitors as monitors
    import bindsnet.network.network as netmod
    import bindsnet.utils as utils
    import matplotlib.pyplot as plt

    path = REF_SCRIPT if os.path.exists(REF_SCRIPT) else STAGED
    with open(path, "rb") as f:
        assert hashlib.sha256(f.read()).hexdigest() == REF_SHA, "not the reference's eth_mnist.py"

    acc, per_call = {}, {}
    cuda = torch.cuda.is_available()

    def timed(key, fn, sync=False):
        def wrapper(*a, **k):
            if sync and cuda:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(*a, **k)
            if sync and cuda:
                torch.cuda.synchronize()
            dt_ = time.perf_counter() - t0
            e = acc.setdefault(key, [0.0, 0])
            e[0] += dt_
            e[1] += 1
            per_call.setdefault(key, []).append(dt_)
            return out
        return wrapper

    patches = []

    def patch(obj, name, key, sync=False):
        orig = getattr(obj, name)
        setattr(obj, name, timed(key, orig, sync))
        patches.append((obj, name, orig))

    patch(netmod.Network, "run", "run", sync=True)
    patch(netmod.Network, "reset_state_variables", "reset", sync=True)
    patch(monitors.Monitor, "get", "monitor_get", sync=True)
    patch(encoding.PoissonEncoder, "__call__", "encode")
    patch(torch.Tensor, "cuda", "h2d", sync=True)
    for mod in (evaluation, sys.modules.get("bindsnet.evaluation.evaluation")):
        if mod is not None:
            for fn in ("all_activity", "proportion_weighting", "assign_labels"):
                patch(mod, fn, "evaluation", sync=True)
    for fn in ("plot_assignments", "plot_input", "plot_performance", "plot_spikes", "plot_voltages", "plot_weights"):
        patch(plotting, fn, "plotting", sync=True)
    for fn in ("get_square_assignments", "get_square_weights"):
        patch(utils, fn, "plotting", sync=True)
    patch(plt, "pause", "plotting")

    argv = ["--n_train", str(args.n_train), "--n_test", str(args.n_test), "--update_interval", str(args.update_interval),
            "--n_neurons", str(args.n_neurons), "--time", "250"]
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = [path] + argv
    torch.manual_seed(0)
    t0 = time.perf_counter()
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            g = runpy.run_path(path, run_name="__main__")
    finally:
        if cuda:
            torch.cuda.synchronize()
        total = time.perf_counter() - t0
        sys.argv = old_argv
        os.chdir(old_cwd)
        for obj, name, orig in patches:
            setattr(obj, name, orig)
        plt.close("all")
    net = g["network"]
    n_samples = acc["run"][1]
    W = net.connections[("X", "Ae")].pipeline[0].value.detach().cpu().numpy()
    other = total - sum(v[0] for v in acc.values())
    out = {
        "script": "examples/mnist/eth_mnist.py (literal file, sha256 " + REF_SHA[:16] + "...)", "impl": args.impl,
        "argv": argv, "encode_device": os.environ.get("SNN_ENCODE_DEVICE", "cpu (the host generator's stream: the reference's)"), "device": str(next(iter(net.layers.values())).s.device) if hasattr(net.layers["Ae"], "s") else None,
        "host_cpus": os.cpu_count(), "torch_threads": torch.get_num_threads(),
        "samples_run": n_samples, "timesteps_per_sample": 250,
        "wall_s": round(total, 3), "samples_per_s": round(n_samples / total, 3),
        "ms_per_sample": {k: round(v[0] / n_samples * 1e3, 3) for k, v in sorted(acc.items())} | {
            "other (DataLoader, python, imports, first-call set-up)": round(other / n_samples * 1e3, 3),
            "total": round(total / n_samples * 1e3, 3)},
        "calls": {k: v[1] for k, v in sorted(acc.items())},
        "run_only_timesteps_per_s": round(n_samples * 250 / acc["run"][0], 1),
        "run_ms_first_call": round(per_call["run"][0] * 1e3, 3), "run_ms_median": round(sorted(per_call["run"])[len(per_call["run"]) // 2] * 1e3, 3),
        "run_median_timesteps_per_s": round(250 / sorted(per_call["run"])[len(per_call["run"]) // 2], 1),
        "encode_ms_median": round(sorted(per_call["encode"])[len(per_call["encode"]) // 2] * 1e3, 3),
        "accuracy": {k: float(v) for k, v in dict(g["accuracy"]).items()},
        "W_sha256": hashlib.sha256(W.tobytes()).hexdigest(),
        "plan": getattr(net, "last_plan", "reference torch CPU path"),
    }
    line = json.dumps(out)
    print(line)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
