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

Input: the generator BASELINE.md section 2 states for this config -- per sample a 28x28 image 128*U(0,1)*Bernoulli(0.19),
Poisson-encoded (bindsnet.encoding.poisson, time=250, dt=1) from torch.manual_seed(1): ~1.17 % spike density; the first
three batches are the trains tests/golden/full_cfg2_dc_n400_b32_poisson.npz holds from the REFERENCE encoder.

Order of the legs (N = 1): CPU baseline first (8-thread / 1-thread legs pinned to 8 CPUs = one CCD), then GPU warm-up +
timed region, parity, roofline profile last.
Extra objects on the JSON line: "roofline" (dominant kernel, HIP-event-timed, vs HBM peak), "cpu_baseline"
(oracle/torch_cpu_ref.py -- the reference's own ATen operator sequence on the host CPU -- 8 threads = `value`,
median of 3 whole inputs; 1 thread; cpu_count()-1 threads [eth_mnist.py:77]; the scalar C port) and "parity" (the
same three inputs from a fresh network on the GPU vs that CPU run, same host, same process: rasters bit for bit,
weights / theta after every input).
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

N_IN, N_EXC, BATCH, T = 784, 400, 32, 250
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)


def algorithmic_bytes_per_timestep(Nin=N_IN, N=N_EXC, B=BATCH):
    """SURVEY.md 8(d) dense accounting for the D&C graph."""
    return 4 * (3 * Nin * N + 2 * N * N) + B * (Nin * 10 + N * 26 + N * 18) + 8 * N


def sparse_effective_bytes_per_timestep(host_pool, Nin=N_IN, N=N_EXC, B=BATCH):
    """SURVEY.md 8(d)'s second figure, reported SEPARATELY from the dense accounting: the same terms, but a weight matrix counts
    only the rows whose source spiked in that timestep (rows of W actually read by an event-driven propagation: the union over the
    batch of the active input pixels; for the two recurrent matrices at most one row per excitatory / inhibitory spike, a handful
    per step, taken as B rows each as an upper bound).  Mean over the timesteps of the input pool."""
    rows = float(np.mean([(h.reshape(T, B, Nin).max(1) > 0).sum(1).mean() for h in host_pool]))      # active X rows per timestep
    return 4 * (rows * N + 2 * Nin * N + 2 * min(B, N) * N) + B * (Nin * 10 + N * 26 + N * 18) + 8 * N, rows


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
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/*_pmc_hbm_traffic.json; PMC cannot be read from inside the process) -> (bytes, where it came from).  The profile carries
    the sha256 of the kernel sources it was taken on (tools/pmc_summary.py): `matches_current_source` says whether that is the tree
    this process runs in."""
    names = {"dc2015-resident-lean": ["r06_lean_pmc_hbm_traffic.json", "r05_lean_pmc_hbm_traffic.json", "r04_lean_pmc_hbm_traffic.json", "r03_lean_pmc_hbm_traffic.json", "r02_lean_pmc_hbm_traffic.json"],
             "dc2015-resident": ["r01_resident_pmc_hbm_traffic.json"], "dc2015-fused": ["r01_pmc_hbm_traffic.json"]}.get(plan, [])
    for name in names:
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                d = json.load(f)
            meta = {"file": "profiles/" + name, "kernel_source_sha16": d.get("kernel_source_sha16")}
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import pmc_summary
                meta["matches_current_source"] = d.get("kernel_source_sha16") == pmc_summary.source_sha16()
            except Exception:                                    # noqa: BLE001
                meta["matches_current_source"] = None
            # raw = FETCH_SIZE + WRITE_SIZE as rocprofv3 prints them; corrected = 2 * FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section: gfx950
            # tallies the 128-byte requests of 16-byte-per-lane loads at 64 bytes).  The corrected value is the one to quote: this kernel's fetches are
            # the digest entries, the producers' input reads and the X-trace rows -- 16-byte-per-lane loads but for the X-trace walk's byte loads
            # of the input (6.3 MB per launch, uncalibrated width) and the 8-byte granule polls (a few hundred KB).
            meta["traffic_raw"] = d.get("hbm_bytes_per_launch_raw")
            meta["traffic_corrected"] = d.get("hbm_bytes_per_launch_gfx950_corrected")
            meta["quoted"] = "corrected (2 * FETCH_SIZE + WRITE_SIZE): the fetches are 16-byte-per-lane loads (digest entries, producers' input reads, X-trace rows)"
            return d.get("hbm_bytes_per_launch_gfx950_corrected"), meta
    return None, None


def c_port_baseline(spikes, steps=100):
    """Scalar C port of the reference algorithm (oracle/snn_oracle.c) on one host core."""
    import oracle
    P = oracle.eth_mnist_dc_params(N_EXC, BATCH, steps)
    torch.manual_seed(0)
    st = oracle.eth_mnist_dc_state(N_EXC, BATCH, (0.3 * torch.rand(N_IN, N_EXC)).numpy())
    Q = oracle.exp_noise(2, 400_000)
    cur = np.zeros(1, np.int64)
    t0 = time.perf_counter()
    try:
        oracle.run_dc2015(P, st, np.ascontiguousarray(spikes[:steps].reshape(steps, BATCH, N_IN)), Q, cur, rasters=False)
    except RuntimeError:
        return None
    dt = time.perf_counter() - t0
    return {"value": round(steps / dt, 2), "unit": "timesteps/s", "cores": 1,
            "sample": f"first {steps} timesteps of input 0, oracle/snn_oracle.c ({dt:.1f} s)"}


def reference_leg(host_inputs, n_whole=5):
    """The REAL reference (byte copies of /root/reference/bindsnet/{network,learning,models,encoding} + utils.py staged under oracle/_ref by
    __graft_entry__.build(), sha256-checked against oracle/ref_manifest.json) on this host's cores: oracle/ref_cpu_leg.py in a process of its
    own.  Returns (json object, records of the whole inputs) or (None, None) when nothing is staged on this box."""
    import subprocess
    import tempfile
    from oracle import stage_ref
    if not stage_ref.verify():
        return None, None
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.npy"), os.path.join(td, "rec.npz")
        np.save(fin, np.stack([h.reshape(T, BATCH, N_IN) for h in host_inputs[:n_whole]]))
        cmd = [sys.executable, os.path.join(ROOT, "oracle", "ref_cpu_leg.py"), "--inputs", fin, "--out", fout, "--n", str(N_EXC),
               "--whole", str(min(n_whole, len(host_inputs)))]
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
            out = json.loads(res.stdout.strip().splitlines()[-1])
        except Exception as e:                               # noqa: BLE001  (the bench line must survive a failed baseline leg)
            return {"error": f"reference leg failed: {str(e)[:300]}"}, None
        z = np.load(fout)
        recs = []
        for k in range(int(z["n_inputs"])):
            recs.append({"Ae": torch.from_numpy(np.unpackbits(z[f"r{k}_Ae"])[:T * BATCH * N_EXC].reshape(T, BATCH, N_EXC)),
                         "Ai": torch.from_numpy(np.unpackbits(z[f"r{k}_Ai"])[:T * BATCH * N_EXC].reshape(T, BATCH, N_EXC)),
                         "W": torch.from_numpy(z[f"r{k}_W"].copy()), "theta": torch.from_numpy(z[f"r{k}_theta"].copy())})
    return out, recs


def port_leg(host_inputs, full):
    """oracle/torch_cpu_ref.py (the reference's ATen operator sequence restated) in this process: 8 threads, median of 3 whole inputs;
    `full`: also its 1-thread and cpu_count()-1 legs (only when the real reference is not staged on this box)."""
    from oracle.torch_cpu_ref import DcTorchRef
    ncpu = os.cpu_count() or 2
    threads0 = torch.get_num_threads()
    spikes = [torch.from_numpy(h.reshape(T, BATCH, N_IN)) for h in host_inputs[:3]]

    def fresh():
        torch.manual_seed(0)
        r = DcTorchRef(n_inpt=N_IN, n_neurons=N_EXC)
        r.set_batch(BATCH)
        torch.manual_seed(2)
        return r

    def leg(threads, n_steps, warm):
        torch.set_num_threads(threads)
        r, rates, recs = fresh(), [], []
        for sp in spikes:
            if warm:
                r.run(sp[:warm])
            t0 = time.perf_counter()
            rec = r.run(sp[warm:warm + n_steps])
            rates.append(n_steps / (time.perf_counter() - t0))
            if n_steps == T:
                recs.append({"Ae": rec["Ae"], "Ai": rec["Ai"], "W": r.W_xe.clone(), "theta": r.theta.clone()})
            r.reset()
        return sorted(rates)[1], [round(x, 2) for x in rates], recs

    v8, all8, recs = leg(min(8, ncpu), T, 0)
    out = {"value": round(v8, 2), "unit": "timesteps/s", "cores": min(8, ncpu), "kind": "port", "per_input_timesteps_per_s": all8,
           "sample": f"median of 3 whole consecutive inputs through oracle/torch_cpu_ref.py (the reference's ATen operator sequence), {min(8, ncpu)} threads"}
    if full:
        v1, all1, _ = leg(1, 100, 0)
        out["one_thread"] = {"value": round(v1, 2), "unit": "timesteps/s", "cores": 1, "per_sample": all1,
                             "sample": "median over the first 100 timesteps of the same 3 inputs"}
    torch.set_num_threads(min(threads0, ncpu))
    return out, recs


def cpu_baseline(host_inputs, aff_all=None, pin=None):
    """The reference's CPU path on this host's cores, BEFORE the GPU leg.  `kind: "reference"` when the unmodified reference is staged on
    this box (oracle/_ref, see reference_leg): 8 threads (SURVEY.md / BASELINE.md section 3's setting) pinned to one CCD = `value`, the MEDIAN
    over 5 WHOLE consecutive inputs from the fixture's start state (weights and theta carry over, reset between), min / median / max
    reported because the figure swings with the box; its 1-thread and cpu_count()-1-thread (eth_mnist.py:77) legs; and, as secondary
    figures, the operator-for-operator port (oracle/torch_cpu_ref.py) and the scalar C port.  Without the staged reference the port is the
    baseline (`kind: "port"`).  Returns (json object, records of the whole inputs for the parity leg)."""
    ncpu = os.cpu_count() or 2
    ref, recs = reference_leg(host_inputs)
    have_ref = ref is not None and "error" not in ref and recs
    port, precs = port_leg(host_inputs, full=not have_ref)
    if aff_all is not None:
        os.sched_setaffinity(0, aff_all)
    cport = c_port_baseline(host_inputs[0])
    pinned = f"8-thread and 1-thread legs pinned to CPUs {pin}" if pin else "not pinned (sched_setaffinity unavailable)"
    if have_ref:
        cpu = {"value": ref["median"], "unit": "timesteps/s", "cores": ref["threads"], "kind": "reference",
               "sample": f"MEDIAN of {ref['inputs']} whole consecutive inputs (T={T}, batch {BATCH}, 784->{N_EXC}, PostPre on, 3 monitors, reset between) "
                         f"through the UNMODIFIED reference (oracle/_ref/bindsnet = byte copies of BindsNET's network/learning/models/encoding packages, "
                         f"sha256 == oracle/ref_manifest.json), Network.run() in a subprocess, {ref['threads']} threads",
               "min": ref["min"], "median": ref["median"], "max": ref["max"], "per_input_timesteps_per_s": ref["per_input_timesteps_per_s"],
               "one_thread": ref.get("one_thread"), "reference_default_threads": ref.get("reference_default_threads"),
               "port": port, "c_port": cport, "host_cpus": ncpu, "affinity": pinned, "torch": ref.get("torch")}
        return cpu, recs
    cpu = dict(port)
    cpu["sample"] += "; the reference itself is not staged on this box (oracle/_ref absent)"
    if ref is not None:
        cpu["reference_error"] = ref.get("error")
    cpu.update({"c_port": cport, "host_cpus": ncpu, "affinity": pinned})
    return cpu, precs


def parity_leg(dev, host_pool, recs, against, pipelined=True):
    """The same inputs from the same seeds on the GPU (fresh network), against the CPU run of this process -- in the mode the timed region
    ran in: inside a pipelined section the generator position of input k+1 is what input k left ON THE DEVICE."""
    import contextlib
    net = build_network(dev)
    torch.manual_seed(2)
    ok, dW, dth, wexact = True, 0.0, 0.0, True
    exc = inh = 0
    plans = []
    with (net.pipelined() if pipelined else contextlib.nullcontext()):
        for r, rec in enumerate(recs):
            net.run({"X": torch.from_numpy(host_pool[r]).view(T, BATCH, 1, 28, 28).to(dev)}, time=T)
            torch.cuda.synchronize()
            plans.append(net.last_plan)
            for l in ("Ae", "Ai"):
                got = net.monitors[l + "_spikes"].get("s").reshape(T, BATCH, N_EXC).cpu()
                ok = ok and bool(torch.equal(got.bool(), rec[l].bool()))
            W = net.connections[("X", "Ae")].pipeline[0].value.detach().cpu()
            dW = max(dW, float((W - rec["W"]).abs().max()))
            wexact = wexact and bool(torch.equal(W, rec["W"]))
            dth = max(dth, float((net.layers["Ae"].theta.cpu() - rec["theta"]).abs().max()))
            exc, inh = exc + int(rec["Ae"].sum()), inh + int(rec["Ai"].sum())
            net.reset_state_variables()
    return {"rasters_bit_exact": ok, "inputs": len(recs), "exc_spikes": exc, "inh_spikes": inh, "max_abs_dW": dW,
            "weights_bit_exact": wexact, "max_abs_dtheta": dth, "plan": plans,
            "plan_retries(lean,resident)": [getattr(net, "lean_retries", 0), getattr(net, "resident_retries", 0)],
            "mode": "pipelined section" if pipelined else "synchronous runs", "against": against}


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


def main_cfg3(args):
    """`bench.py --config cfg3 --gpus N`: BASELINE.json configs[2] -- TwoLayerNetwork 784 -> 1600, global batch 128, 100 timesteps per
    network.run(), PostPre (bindsnet/models/models.py:21-91, learning/learning.py:390-420) -- on N ranks, BOTH multi-GPU modes timed in
    one line (same schema as the headline's):
      value                 north_star's schedule: the batch shards (128 / N samples per rank), ONE flat RCCL all-reduce of the weight deltas per
                            input, clamp, normalise (parallel.sharded_run).  Not the reference's arithmetic for a global batch.
      exact_column_shard    the exact mode: every rank holds 32-aligned column slices and the WHOLE batch, no collective at all during the run
                            (parallel.column_shard): bit-identical to the single-process global batch.
    Strong scaling: the global batch is fixed at 128 as N grows."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    Tc, Bg, Nin, N = 100, 128, 784, 1600
    if Bg % world or (Bg // world) % 16:
        raise SystemExit(f"bench.py --config cfg3: 128 samples do not shard into 16-sample blocks over {world} ranks (SURVEY 8(e))")
    stage = {"name": "start", "t0": time.time()}
    metric = "simulated timesteps/sec (whole node), TwoLayerNetwork 784->1600 batch128"

    def error_line(msg):
        if rank == 0:
            print(json.dumps({"metric": metric, "value": None, "unit": "timesteps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "higher_is_better": True, "scaling": "strong", "error": msg, "stage": stage["name"], "backend": args.backend if world > 1 else None}), flush=True)

    def enter(name):
        stage["name"], stage["t0"] = name, time.time()

    if world > 1:
        import threading

        def bark():
            while stage["name"] != "done":
                time.sleep(1.0)
                if time.time() - stage["t0"] > args.watchdog:
                    error_line(f"stage '{stage['name']}' did not finish within {args.watchdog:.0f} s (rank {rank} of {world})")
                    sys.stdout.flush()
                    os._exit(3)
        threading.Thread(target=bark, daemon=True).start()
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ndev = torch.cuda.device_count()
        if args.backend == "nccl" and local >= ndev:
            error_line(f"rank {rank}: LOCAL_RANK {local} but only {ndev} GPU(s) visible (RCCL needs one GPU per rank)")
            raise SystemExit(2)
        local_dev = local % max(1, ndev)
        torch.cuda.set_device(local_dev)
        enter("init_process_group")
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev), timeout=datetime.timedelta(seconds=args.watchdog))
            else:
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=args.watchdog))
        except Exception as e:                                           # noqa: BLE001
            error_line(f"init_process_group({args.backend}) failed on rank {rank}: {str(e)[:400]}")
            raise
    else:
        dist, local_dev = None, 0
    dev = torch.device("cuda", local_dev)
    torch.cuda.set_device(dev)
    from bindsnet_amd import parallel, synth
    from bindsnet_amd.models import TwoLayerNetwork

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step):
        for k in range(args.warmup):
            step(k)
        fence()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(args.warmup + k)
        fence()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    host = [synth.dense_spikes(2 + 10 * k, (Tc, Bg, Nin), 0.012) for k in range(4)]          # tools/baseline_configs.py's cfg3 inputs
    try:
        # ---- north star: batch shards + one all-reduce of the deltas per input
        enter("sharded_run")
        Bs = Bg // world
        torch.manual_seed(0)
        net = TwoLayerNetwork(n_inpt=Nin, n_neurons=N, reduction=torch.sum)
        net.to(dev)
        shard_in = [torch.from_numpy(np.ascontiguousarray(h[:, rank * Bs:(rank + 1) * Bs])).to(dev) for h in host]
        last = [x[Tc - 1].clone() for x in shard_in]

        def step_a(k):
            x = shard_in[k % 4]
            parallel.sharded_run(net, {"X": x}, Tc)
            net.reset_state_variables()
            x[Tc - 1].copy_(last[k % 4])                    # (Input.s aliases the last slice; reset zeroes it in place)
        el_a = timed(step_a)
        plan_a = net.last_plan
        st = net.__dict__.get("_shard_state")
        coll = None
        if dist is not None and st is not None:
            buf = st["delta"].clone()
            fence()
            c0 = time.perf_counter()
            for _ in range(20):
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            fence()
            coll = {"op": "all_reduce(SUM) of the flat weight delta", "bytes": buf.numel() * buf.element_size(), "ms": round((time.perf_counter() - c0) / 20 * 1e3, 4)}
        # ---- exact: column slices, whole batch, no collective
        enter("column_shard")
        torch.manual_seed(0)
        full = TwoLayerNetwork(n_inpt=Nin, n_neurons=N, reduction=torch.sum)
        full.batch_size = Bg
        full.to(dev)
        shard, lo, hi = parallel.column_shard(full, rank, world)
        full_in = [torch.from_numpy(h).to(dev) for h in host]
        last_f = [x[Tc - 1].clone() for x in full_in]

        def step_b(k):
            x = full_in[k % 4]
            if shard is not None:
                shard.run({"X": x}, time=Tc)
                shard.reset_state_variables()
            x[Tc - 1].copy_(last_f[k % 4])
        el_b = timed(step_b)
        plan_b = shard.last_plan if shard is not None else None
    except Exception as e:                                     # noqa: BLE001
        error_line(f"rank {rank}, stage '{stage['name']}': {type(e).__name__}: {str(e)[:400]}")
        raise
    enter("post")
    if rank == 0:
        line = {"metric": metric, "value": round(args.steps * Tc / el_a, 2), "unit": "timesteps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(el_a / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "configs[2]: TwoLayerNetwork 784->1600 exc LIF, global batch 128, 100 timesteps per network.run(), PostPre STDP, "
                                       "reset_state_variables() per input; Bernoulli(0.012) spike trains, 4 resident batches cycled",
                           "timesteps_per_step": Tc, "global_batch": Bg, "batch_per_gpu": Bs, "parallelism": f"batch-shard x{world} + all-reduce of the weight deltas per input (north_star)",
                           "plan": plan_a, "backend": args.backend if world > 1 else None, "per_input_collective": coll,
                           "sample_timesteps_per_s": round(args.steps * Tc * Bg / el_a, 1)},
                "exact_column_shard": {"value": round(args.steps * Tc / el_b, 2), "unit": "timesteps/s", "ms_per_step": round(el_b / args.steps * 1e3, 4),
                                       "columns_of_rank0": [lo, hi], "batch_per_gpu": Bg, "collectives_during_the_run": 0, "plan": plan_b,
                                       "what": "parallel.column_shard: 32-aligned column slices, the whole batch on every rank: bit-identical to the single-process global batch"},
                "roofline": None, "cpu_baseline": None}
        print(json.dumps(line), flush=True)
    stage["name"] = "done"
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3"], help="cfg2: the headline (BASELINE.json configs[1]); cfg3: configs[2] on N ranks, both multi-GPU modes")
    ap.add_argument("--gpus", type=int, default=1)

    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plan", default="auto", choices=["auto", "generic", "per-step"])
    ap.add_argument("--sync-runs", action="store_true", help="every network.run() waits for the device (the reference's call-by-call "
                    "behaviour) instead of the pipelined section the timed region normally is")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="N > 1: nccl = RCCL over xGMI (one rank per GPU); gloo lets the "
                    "whole N-rank path run with N processes on ONE GPU (tests)")
    ap.add_argument("--watchdog", type=float, default=float(os.environ.get("SNN_BENCH_WATCHDOG", "240")),
                    help="seconds a stage (rendezvous, warm-up, timed region) may take before rank 0 prints a JSON line with `error` and the process exits")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if not torch.cuda.is_available() or (torch.cuda.device_count() < args.gpus and args.backend == "nccl"):
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
        respawn_under_launcher(args)
    if args.config == "cfg3":
        return main_cfg3(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # ---- inputs on the host, then the CPU leg -- BEFORE this process creates its HIP context (with the runtime's
    #      threads alive the same operators ran 3-4x slower on the GPU box's EPYC: 21 vs 80 timesteps/s).  The 8-thread and
    #      1-thread legs run PINNED to the first 8 CPUs this process may use (one CCD of the EPYC: 8 cores, one L3): unpinned,
    #      the 8-thread figure flipped between ~27 and ~170 timesteps/s from one input to the next on the same box.
    cpu_leg = rank == 0 and world == 1 and not args.no_cpu_baseline
    aff_all = pin = None
    if cpu_leg and hasattr(os, "sched_setaffinity"):
        try:
            aff_all = os.sched_getaffinity(0)
            pin = sorted(aff_all)[:8]
            torch.set_num_threads(min(8, len(pin)))        # (before the first parallel operator creates the worker threads)
            os.sched_setaffinity(0, pin)
        except OSError:
            aff_all = pin = None
    from bindsnet_amd import synth
    host_pool = synth.poisson_mnist_like(BATCH, T, 5 if cpu_leg else 4, seed=1 + 17 * rank)   # (the CPU leg runs 5 inputs; 4 are cycled on the GPU)
    per = np.stack([h.reshape(T, BATCH, N_IN).sum(2) for h in host_pool])
    input_stats = {"generator": "torch.manual_seed(1 + 17*rank); per sample img = 128*U(0,1)*Bernoulli(0.19); "
                                "bindsnet.encoding.poisson(img, time=250, dt=1.0) (BASELINE.md section 2)",
                   "density": round(float(np.mean([h.mean() for h in host_pool])), 5),
                   "events_per_sample_timestep": {"mean": round(float(per.mean()), 2), "max": int(per.max())},
                   "matches_reference_encoded_fixture": (
                       [synth.sha(h.reshape(T, BATCH, N_IN)) for h in host_pool[:3]] == synth.POISSON_CFG2_SHA) if rank == 0 else None}
    cpu = recs = None
    if cpu_leg:                                                 # N = 1 only
        cpu, recs = cpu_baseline(host_pool, aff_all, pin)
    # ---- N > 1: a stage that hangs (a rank that died, an RCCL ring that never forms) must not cost the line: the watchdog prints one with
    #      `error` from rank 0 and ends the process
    stage = {"name": "start", "t0": time.time()}

    def error_line(msg):
        if rank == 0:
            print(json.dumps({"metric": "simulated timesteps/sec (whole node), DiehlAndCook2015 784->400 batch32", "value": None, "unit": "timesteps/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
                              "error": msg, "stage": stage["name"], "backend": args.backend if world > 1 else None,
                              "env": {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "LOCAL_RANK", "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG")}}), flush=True)

    def watchdog():
        import threading

        def bark():
            while True:
                time.sleep(1.0)
                if stage["name"] == "done":
                    return
                if time.time() - stage["t0"] > args.watchdog:
                    error_line(f"stage '{stage['name']}' did not finish within {args.watchdog:.0f} s (rank {rank} of {world})")
                    sys.stdout.flush()
                    os._exit(3)
        threading.Thread(target=bark, daemon=True).start()

    def enter(name):
        stage["name"], stage["t0"] = name, time.time()

    if world > 1:
        watchdog()
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "WARN")                      # (RCCL's own complaints end up on stderr next to the line)
        ndev = torch.cuda.device_count()
        if args.backend == "nccl" and local >= ndev:
            error_line(f"rank {rank}: LOCAL_RANK {local} but only {ndev} GPU(s) visible (RCCL needs one GPU per rank)")
            raise SystemExit(2)
        local_dev = local % max(1, ndev)                                 # (gloo: several ranks may share a GPU)
        torch.cuda.set_device(local_dev)
        enter("init_process_group")
        try:
            import datetime
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev), timeout=datetime.timedelta(seconds=args.watchdog))
            else:
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=args.watchdog))
        except Exception as e:                                           # noqa: BLE001
            error_line(f"init_process_group({args.backend}) failed on rank {rank}: {str(e)[:400]}")
            raise
    else:
        dist = None
        local_dev = 0
    dev = torch.device("cuda", local_dev)
    torch.cuda.set_device(dev)

    from bindsnet_amd import _lib, parallel
    _lib.lib().snn_set_plan_mode({"auto": 0, "generic": 1, "per-step": 2}[args.plan])
    net = build_network(dev)
    pool = [torch.from_numpy(h).view(T, BATCH, 1, 28, 28).to(dev) for h in host_pool[:4]]   # resident in HBM before the timed region
    # Input.s aliases the last slice of the caller's input and reset_state_variables() zeroes it in place (the
    # reference does the same: nodes.py:219 `self.s = x`, :114 `self.s.zero_()`); eth_mnist.py encodes a fresh tensor
    # per sample and never notices, a cycled pool would lose its last timestep after the first pass.  The slice is
    # put back after every reset (one 25 KB device copy, INSIDE the timed region) so every pass simulates the stated input.
    last = [x[T - 1].clone() for x in pool]

    def one(k):
        x = {"X": pool[k % len(pool)]}
        if world > 1:
            parallel.sharded_run(net, x, T)
        else:
            net.run(x, time=T)
        net.reset_state_variables()
        pool[k % len(pool)][T - 1].copy_(last[k % len(pool)])

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    import contextlib
    # The timed region is ONE pipelined section (Network.pipelined(), an extension of the reference's API): run() returns without waiting
    # for the device, so the host prepares input k+1 while input k executes; the status words of all runs and the host generator are
    # settled by net.sync() INSIDE the timed region, before the closing fence.  --sync-runs: every run() waits, as a reference user's loop does.
    section = contextlib.nullcontext() if args.sync_runs else net.pipelined()
    torch.manual_seed(2)                                   # host generator: feeds the one_spike arbitration (consumed on the device)
    enter("warmup")
    try:
        with section:
            for k in range(args.warmup):
                one(k)
            net.sync()
            fence()
            enter("timed region")
            # HIP events (on the launch stream) around the dominant launch of every run of the timed region itself (the first 256 of them): the
            # `roofline` of the line is the kernel as it ran inside the number it stands beside (two event records per run)
            _lib.lib().snn_profile_enable(4)
            t0 = time.perf_counter()
            for k in range(args.steps):
                one(args.warmup + k)
            host_enqueue = time.perf_counter() - t0            # the host has issued every step: its own cost per step
            net.sync()                                         # status of every run checked, host generator written back
            fence()
            elapsed = time.perf_counter() - t0
            import ctypes as _C
            ev_sum, ev_n = _C.c_double(0), _C.c_int(0)
            _lib.check(_lib.lib().snn_profile_collect(_C.byref(ev_sum), _C.byref(ev_n)), "profile_collect")
            _lib.lib().snn_profile_enable(0)
    except Exception as e:                                     # noqa: BLE001
        error_line(f"rank {rank}, stage '{stage['name']}': {type(e).__name__}: {str(e)[:400]}")
        raise
    # ---- the same K steps as a reference user's loop runs them (examples/mnist/eth_mnist.py:243-250 reads its monitors after every
    #      network.run()): no section, every run() waits for the device.  A secondary figure of the same line (`sync_runs`), N = 1 only.
    sync_leg = None
    if world == 1 and not args.sync_runs:
        enter("sync runs")
        try:
            for k in range(2):                                 # (untimed: the process's FIRST cooperative launch creates the runtime's cooperative queue, several ms)
                one(args.warmup + args.steps + k)
            fence()
            s0 = time.perf_counter()
            for k in range(args.steps):
                one(args.warmup + args.steps + 2 + k)
            fence()
            s_el = time.perf_counter() - s0
            sync_leg = {"value": round(args.steps * T / s_el, 2), "unit": "timesteps/s", "ms_per_step": round(s_el / args.steps * 1e3, 4), "steps": args.steps,
                        "what": "the same steps with every network.run() waiting for the device (no Network.pipelined() section): the reference-shaped loop"}
        except Exception as e:                                 # noqa: BLE001  (a secondary figure must never cost the line)
            sync_leg = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    enter("reduce")
    ranks_seen, devices = 1, [f"cuda:{dev.index} {torch.cuda.get_device_name(dev)}"]
    collective = None
    if dist is not None:
        tdev = dev if args.backend == "nccl" else torch.device("cpu")
        own = elapsed
        tt = torch.tensor([elapsed], device=tdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ranks_seen = dist.get_world_size()
        ids = [None] * ranks_seen
        dist.all_gather_object(ids, f"rank{rank}=cuda:{dev.index} own_elapsed_s={own:.4f}")
        devices = ids
        # the per-input collective on its own: the flat delta buffer sharded_run all-reduces (weights + thresholds), 20 times
        st = net.__dict__.get("_shard_state")
        if st is not None and "delta" in st:
            buf = st["delta"].clone()
            fence()
            c0 = time.perf_counter()
            for _ in range(20):
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            fence()
            collective = {"op": "all_reduce(SUM) of the flat weight + threshold delta", "bytes": buf.numel() * buf.element_size(),
                          "ms": round((time.perf_counter() - c0) / 20 * 1e3, 4), "backend": args.backend}
    enter("post")

    if rank == 0:
        plan_timed = net.last_plan
        retries = [getattr(net, "lean_retries", 0), getattr(net, "resident_retries", 0)]
        par = parity_leg(dev, host_pool, recs, (
            f"the unmodified reference (oracle/_ref) on this host in this run, {len(recs)} consecutive inputs from identical seeds" if cpu.get("kind") == "reference"
            else "oracle/torch_cpu_ref.py on this host in this run, 3 consecutive inputs from identical seeds") + " (weights, theta compared after each)",
            pipelined=not args.sync_runs) if recs else None
        # ---- roofline of the dominant kernel: HIP events (on the launch stream) around single launches of further
        # runs of the same input pool, LAST, so the device is busy until the process prints its line
        roof = None
        prof = None
        if ev_n.value > 0 and plan_timed.startswith("dc2015-resident"):     # the timed region's own launches
            form = _lib.lib().snn_dc2015_last_form()
            prof = {"kernel": _lib.resident_kernel_name(form) + " (one launch per network.run(); HIP events around the launches of the timed region)",
                    "resident_form": form, "avg_ms": ev_sum.value / ev_n.value, "n": ev_n.value, "timesteps_per_launch": T}
        if prof is None:                                                    # per-step / generic plans: sampled launches of extra runs
            prof = _lib.profile_run(net, {"X": pool[0].clone()}, T, repeats=25, pipelined=not args.sync_runs)
        if prof is not None:
            ab = algorithmic_bytes_per_timestep() * prof["timesteps_per_launch"]
            ach = ab / (prof["avg_ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": prof["kernel"], "avg_launch_us": round(prof["avg_ms"] * 1e3, 3),
                    "launches_timed": prof["n"], "algorithmic_bytes_per_launch": ab, "achieved": round(ach, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                    "traffic": pmc_traffic(plan_timed)[0], "traffic_raw": (pmc_traffic(plan_timed)[1] or {}).get("traffic_raw"),
                    "traffic_profile": pmc_traffic(plan_timed)[1],
                    "frac_of_measured_copy_bandwidth_6290": round(ach / 6290.0, 5)}
            if "resident_form" in prof:
                roof["resident_form"] = prof["resident_form"]
            try:                                           # the sparse-effective variant, never mixed with the dense one
                sb, rows = sparse_effective_bytes_per_timestep(host_pool)
                sach = sb * prof["timesteps_per_launch"] / (prof["avg_ms"] * 1e-3) / 1e9
                roof["sparse_effective"] = {"bytes_per_timestep": int(sb), "active_input_rows_per_timestep": round(rows, 1),
                                            "achieved": round(sach, 2), "frac": round(sach / HBM_PEAK_GBS, 5)}
            except Exception as e:                         # noqa: BLE001  (an extra figure must never cost the bench line)
                roof["sparse_effective"] = {"error": str(e)[:200]}
        steps_total = world * args.steps * T
        line = {
            "metric": "simulated timesteps/sec (whole node), DiehlAndCook2015 784->400 batch32",
            "value": round(steps_total / elapsed, 2), "unit": "timesteps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "rccl_ranks": ranks_seen, "devices": devices,
            "config": {"workload": "configs[1]: DiehlAndCook2015 784->400 exc, batch 32/GPU, 250 timesteps per "
                                   "network.run(), PostPre STDP on, 3 spike monitors (X, Ae, Ai), reset_state_variables() per input; "
                                   "input = BASELINE.md's stated generator (Poisson-encoded 128*U*Bernoulli(0.19) images), "
                                   "4 resident batches cycled",
                       "input": input_stats,
                       "timesteps_per_step": T, "batch_per_gpu": BATCH, "global_batch": BATCH * world,
                       "sample_timesteps_per_s": round(steps_total * BATCH / elapsed, 1),
                       "plan": plan_timed, "plan_retries(lean,resident)": retries, "graph_runs(plain,captured,replayed)": list(_lib.graph_stats()), "parallelism": f"batch-shard x{world}" if world > 1 else "single",
                       "host_sync": "per run (--sync-runs)" if args.sync_runs else "pipelined section (Network.pipelined()): status words + host generator settled by net.sync() inside the timed region",
                       "backend": args.backend if world > 1 else None, "per_input_collective": collective},
            "roofline": roof,
            "cpu_baseline": cpu,
            "parity": par,
            "sync_runs": sync_leg,
        }
        if cpu is not None:
            line["speedup_vs_cpu_baseline"] = round(line["value"] / cpu["value"], 1)
        print(json.dumps(line), flush=True)
    stage["name"] = "done"
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
