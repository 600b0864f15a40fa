#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json: simulated timesteps/sec, DiehlAndCook2015
784->400 excitatory neurons, batch 32, T = 250 ms @ dt = 1 ms, PostPre STDP on (configs[1]).

A "step" (--steps K) is ONE network.run() of T = 250 timesteps over one resident synthetic input
batch [250, 32, 784] u8 followed by network.reset_state_variables() (what eth_mnist.py does per
sample); `value` = timesteps simulated per second summed over all ranks.

  python bench.py [--gpus N --steps K --warmup W]            (N > 1: launched by torch.distributed.run)

N > 1 (weak scaling): every rank simulates its own batch of 32 with replicated weights; after
each input the weight and threshold deltas are all-reduced over RCCL and re-normalised
(bindsnet_amd.parallel.sharded_run; BASELINE.json north-star schedule -- DESIGN.md section
"Multi-GPU" explains how it differs from a single global batch).

`--gpus N` without a torch.distributed environment re-executes this script under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, RCCL); the JSON line reports the
ranks RCCL actually formed (`rccl_ranks`) and the device every rank ran on.

Extra objects on the JSON line: "roofline" (dominant kernel, HIP-event-timed, vs HBM peak),
"cpu_baseline" (N = 1 only: oracle/torch_cpu_ref.py -- the reference's own ATen operator sequence on the
host CPU -- at cpu_count()-1 threads [what eth_mnist.py:77 sets] and at 1 thread, plus the scalar C port)
and "parity" (the first input re-run from a fresh network on the GPU and on the CPU restatement, same
host, same run: rasters compared bit for bit, weights by max |dW|).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_IN, N_EXC, BATCH, T = 784, 400, 32, 250
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)


def algorithmic_bytes_per_timestep(Nin=N_IN, N=N_EXC, B=BATCH):
    """SURVEY.md 8(d) dense accounting for the D&C graph."""
    return 4 * (3 * Nin * N + 2 * N * N) + B * (Nin * 10 + N * 26 + N * 18) + 8 * N


def make_inputs(seed, n_batches, device):
    import synth
    out = []
    for k in range(n_batches):
        sp = synth.spike_train(seed + k, T, BATCH, N_IN)
        out.append(torch.from_numpy(sp).view(T, BATCH, 1, 28, 28).to(device))
    return out


def build_network(device):
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=N_IN, n_neurons=N_EXC, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05,
                           inpt_shape=(1, 28, 28))
    for l in ("X", "Ae", "Ai"):                  # the three spike monitors eth_mnist.py:143-148 registers
        net.add_monitor(Monitor(net.layers[l], ["s"], time=T), l + "_spikes")
    net.to(device)
    return net


def pmc_traffic(plan):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same
    command (profiles/r01_pmc_hbm_traffic.json; PMC cannot be read from inside the process)."""
    name = {"dc2015-resident-lean": "r02_lean_pmc_hbm_traffic.json", "dc2015-resident": "r01_resident_pmc_hbm_traffic.json",
            "dc2015-fused": "r01_pmc_hbm_traffic.json"}.get(plan)
    path = os.path.join(ROOT, "profiles", name) if name else None
    if not path or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get("hbm_bytes_per_launch_gfx950_corrected")


def c_port_baseline(steps=100):
    """Scalar C port of the reference algorithm (oracle/snn_oracle.c) on one host core."""
    import cases
    import oracle
    import synth
    from test_oracle_golden import dc_params
    g = cases.gold("run_dc_n400_b32")
    P = dc_params(g)
    P.T, P.B, P.N = steps, BATCH, N_EXC
    st = cases.dc_state(N_EXC, BATCH)
    sp = np.ascontiguousarray(synth.spike_train(20, T, BATCH, N_IN)[:steps])
    Q = cases.exp_noise(2, 400_000)
    cur = np.zeros(1, np.int64)
    t0 = time.perf_counter()
    try:
        oracle.run_dc2015(P, st, sp, Q, cur, rasters=False)
    except RuntimeError:
        return None
    dt = time.perf_counter() - t0
    return {"value": round(steps / dt, 2), "unit": "timesteps/s", "cores": 1,
            "sample": f"{steps} timesteps of one input, oracle/snn_oracle.c ({dt:.1f} s)"}


def cpu_baseline_and_parity(dev, seed_inputs):
    """The reference's CPU path restated operator for operator (oracle/torch_cpu_ref.py), on this host's
    cores, in this run -- timed, and its outputs compared with a fresh GPU run of the same input."""
    import synth
    from oracle.torch_cpu_ref import DcTorchRef
    spikes = synth.spike_train(seed_inputs, T, BATCH, N_IN)
    ncpu = os.cpu_count() or 2
    threads0 = torch.get_num_threads()

    def fresh():
        torch.manual_seed(0)
        r = DcTorchRef(n_inpt=N_IN, n_neurons=N_EXC)
        r.set_batch(BATCH)
        torch.manual_seed(2)
        return r

    # --- thread-count probe on a bounded sample (6 timesteps after 1 untimed one, same start every time).  The
    # reference's own setting is cpu_count()-1 (examples/mnist/eth_mnist.py:77); on a many-core host that
    # oversubscribes these small operators badly, so the baseline VALUE is the best setting found, not that one.
    probe = {}
    for nt in sorted({1, 4, 8, 16, 32, max(1, ncpu - 1)}):
        if nt > max(1, ncpu - 1):
            continue
        torch.set_num_threads(nt)
        r = fresh()
        r.run(torch.from_numpy(spikes[:1]))
        t0 = time.perf_counter()
        r.run(torch.from_numpy(spikes[1:7]))
        probe[nt] = round(6 / (time.perf_counter() - t0), 2)
    best = max(probe, key=probe.get)
    # --- one whole input at the best thread count (also the parity witness)
    torch.set_num_threads(best)
    ref = fresh()
    t0 = time.perf_counter()
    rec = ref.run(torch.from_numpy(spikes))
    dt_best = time.perf_counter() - t0
    torch.set_num_threads(threads0)
    cpu = {"value": round(T / dt_best, 2), "unit": "timesteps/s", "cores": best, "kind": "port",
           "sample": f"1 input (T={T}, batch {BATCH}, 784->{N_EXC}, PostPre on, 3 monitors) through oracle/torch_cpu_ref.py = the "
                     f"reference's ATen operator sequence, {best} threads = the best of the probed settings ({dt_best:.1f} s); "
                     "/root/reference itself is absent on this box",
           "threads_probe_timesteps_per_s": {str(k): v for k, v in probe.items()},
           "reference_default_threads": {"threads": max(1, ncpu - 1), "value": probe[max(1, ncpu - 1)],
                                         "note": "torch.set_num_threads(os.cpu_count() - 1), eth_mnist.py:77; 6-timestep sample"},
           "one_thread": {"value": probe[1], "unit": "timesteps/s", "cores": 1, "sample": "6-timestep sample"},
           "c_port": c_port_baseline(), "host_cpus": ncpu}
    # --- parity: the same input from the same seeds on the GPU
    torch.manual_seed(0)
    net = build_network(dev)
    torch.manual_seed(2)
    net.run({"X": torch.from_numpy(spikes).view(T, BATCH, 1, 28, 28).to(dev)}, time=T)
    torch.cuda.synchronize()
    ok = {}
    for l in ("Ae", "Ai"):
        got = net.monitors[l + "_spikes"].get("s").reshape(T, BATCH, N_EXC).cpu()
        ok[l] = bool(torch.equal(got.bool(), rec[l].bool()))
    W = net.connections[("X", "Ae")].pipeline[0].value.detach().cpu()
    theta = net.layers["Ae"].theta.cpu()
    probe_gpu = torch.rand(4)                             # host generator position after the GPU run ...
    par = {"rasters_bit_exact": all(ok.values()), "exc_spikes": int(rec["Ae"].sum()), "inh_spikes": int(rec["Ai"].sum()),
           "max_abs_dW": float((W - ref.W_xe).abs().max()), "weights_bit_exact": bool(torch.equal(W, ref.W_xe)),
           "max_abs_dtheta": float((theta - ref.theta).abs().max()),
           "against": "oracle/torch_cpu_ref.py on this host in this run, one input from identical seeds"}
    return cpu, par


def respawn_under_launcher(args):
    """`python bench.py --gpus N` with no torch.distributed environment: start N ranks ourselves."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plan", default="auto", choices=["auto", "generic", "per-step"])
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if not torch.cuda.is_available() or torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
        respawn_under_launcher(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist = None
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)

    from bindsnet_amd import _lib, parallel
    _lib.lib().snn_set_plan_mode({"auto": 0, "generic": 1, "per-step": 2}[args.plan])
    net = build_network(dev)
    pool = make_inputs(1000 + 17 * rank, 4, dev)           # resident in HBM before the timed region

    def one(k):
        x = {"X": pool[k % len(pool)]}
        if world > 1:
            parallel.sharded_run(net, x, T)
        else:
            net.run(x, time=T)
        net.reset_state_variables()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    torch.manual_seed(2)                                   # host generator: feeds the one_spike arbitration (consumed on the device)
    for k in range(args.warmup):
        one(k)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one(args.warmup + k)
    fence()
    elapsed = time.perf_counter() - t0
    ranks_seen, devices = 1, [f"cuda:{dev.index} {torch.cuda.get_device_name(dev)}"]
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ranks_seen = dist.get_world_size()
        ids = [None] * ranks_seen
        dist.all_gather_object(ids, f"rank{rank}=cuda:{dev.index}")
        devices = ids

    # ---- roofline of the dominant kernel: HIP events around single launches, after the timed region
    roof = None
    if rank == 0:
        prof = _lib.profile_run(net, {"X": pool[0]}, T)
        if prof is not None:
            ab = algorithmic_bytes_per_timestep() * prof["timesteps_per_launch"]
            ach = ab / (prof["avg_ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": prof["kernel"], "avg_launch_us": round(prof["avg_ms"] * 1e3, 3),
                    "launches_timed": prof["n"], "algorithmic_bytes_per_launch": ab, "achieved": round(ach, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                    "traffic": pmc_traffic(net.last_plan)}

    if rank == 0:
        cpu = par = None
        if not (args.no_cpu_baseline or world > 1):        # N = 1 only
            cpu, par = cpu_baseline_and_parity(dev, 1000)
        steps_total = world * args.steps * T
        line = {
            "metric": "simulated timesteps/sec (whole node), DiehlAndCook2015 784->400 batch32",
            "value": round(steps_total / elapsed, 2), "unit": "timesteps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "rccl_ranks": ranks_seen, "devices": devices,
            "config": {"workload": "configs[1]: DiehlAndCook2015 784->400 exc, batch 32/GPU, 250 timesteps per "
                                   "network.run(), PostPre STDP on, 3 spike monitors (X, Ae, Ai), reset_state_variables() per input",
                       "timesteps_per_step": T, "batch_per_gpu": BATCH, "global_batch": BATCH * world,
                       "sample_timesteps_per_s": round(steps_total * BATCH / elapsed, 1),
                       "plan": net.last_plan, "plan_retries(lean,resident)": [getattr(net, "lean_retries", 0), getattr(net, "resident_retries", 0)], "graph_runs(plain,captured,replayed)": list(_lib.graph_stats()), "parallelism": f"batch-shard x{world}" if world > 1 else "single"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "parity": par,
        }
        if cpu is not None:
            line["speedup_vs_cpu_baseline"] = round(line["value"] / cpu["value"], 1)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
