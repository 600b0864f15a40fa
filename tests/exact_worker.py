#!/usr/bin/env python3
"""One rank of the exact batch-sharded mode (bindsnet_amd.parallel.exact_run): a process of its own with a torch.distributed
group (gloo), ITS rows of a reference fixture's global batch, on the host (network/host_path.py's operators) or on the GPU
(the C ABI's operators).  The parent puts the ranks' rows side by side and compares them with what the UNMODIFIED reference
produced for the global batch in one process (tests/golden/make_golden*.py).

    python tests/exact_worker.py --rank R --world W --port P --device cpu|cuda --fixture NAME --out FILE [--runs K]"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402
import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--fixture", required=True)
    ap.add_argument("--runs", type=int, default=0)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--native", action="store_true", help="exchange through the C ABI's own RCCL communicator (parallel.NativeComm)")
    ap.add_argument("--mode", default="per-step", choices=["per-step", "gathered", "auto"])
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    if a.world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port), RANK=str(a.rank), WORLD_SIZE=str(a.world))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    if a.device == "cuda":
        torch.cuda.set_device(0)
    from bindsnet_amd import parallel
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    g = cases.gold(a.fixture)
    N, B, T = int(g["N"]), int(g["B"]), int(g["T"])
    runs = a.runs or int(g["runs"])
    full = a.fixture.startswith("full_")
    per = B // a.world
    assert per * a.world == B
    lo, hi = a.rank * per, (a.rank + 1) * per
    torch.manual_seed(0)                                          # identical replicas
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120.0 if full else float(g["inh"]), dt=1.0, norm=78.4, theta_plus=0.05,
                           inpt_shape=(1, 28, 28))
    feat = net.connections[("X", "Ae")].pipeline[0]
    if not full:
        feat.value.data.copy_(torch.from_numpy(synth.weights_q12(10, 784, N)))
    mons = {l: Monitor(net.layers[l], ["s"], time=T) for l in ("X", "Ae", "Ai")}
    for l, m in mons.items():
        net.add_monitor(m, l + "_s")
    if a.device == "cuda":
        net.to("cuda")
    comm = parallel.NativeComm(a.rank, a.world) if a.native else None
    host = lambda t: t.detach().cpu().numpy()                    # noqa: E731
    out = {"lo": lo, "hi": hi}
    if full:
        torch.manual_seed(2)                                      # the SAME generator state on every rank
    for r in range(runs):
        if full:
            spikes = cases.fixture_input(g, r, T, B)
        else:
            spikes = synth.spike_train(20 + r, T, B, 784, max_rate=float(g["max_rate"]))
            torch.manual_seed(2 + r)
        shard = torch.from_numpy(np.ascontiguousarray(spikes[:, lo:hi])).view(T, per, 1, 28, 28).to(a.device)
        if a.device == "cuda":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        parallel.exact_run(net, {"X": shard}, T, comm=comm, mode=a.mode)
        if a.device == "cuda":
            torch.cuda.synchronize()
        out[f"r{r}_seconds"] = time.perf_counter() - t0
        assert net.last_plan == "exact-sharded" or (a.mode != "per-step" and net.last_plan.startswith("exact-gathered:")), net.last_plan
        out[f"r{r}_plan"] = net.last_plan
        if net.__dict__.get("_exact_timing"):
            print(f"rank {a.rank} run {r}: {out[f'r{r}_seconds'] * 1e3:.2f} ms", net.__dict__["_exact_timing"], flush=True)
        Ae, Ai, X = net.layers["Ae"], net.layers["Ai"], net.layers["X"]
        out[f"r{r}_sX"] = np.packbits(host(mons["X"].get("s")).astype(np.uint8))
        out[f"r{r}_sE"] = np.packbits(host(mons["Ae"].get("s")).astype(np.uint8))
        out[f"r{r}_sI"] = np.packbits(host(mons["Ai"].get("s")).astype(np.uint8))
        out[f"r{r}_W"] = host(feat.value).copy()
        out[f"r{r}_theta"] = host(Ae.theta).copy()
        for key, t in (("vE", Ae.v), ("rE", Ae.refrac_count), ("xE", Ae.x), ("xX", X.x.reshape(per, 784)), ("vI", Ai.v), ("rI", Ai.refrac_count)):
            out[f"r{r}_{key}"] = host(t).copy()
        if full or r % 2 == 0:
            net.reset_state_variables()
    out["probe_after"] = torch.rand(4).numpy()                    # where the host generator stands
    if a.world > 1:
        dist.barrier()
    np.savez_compressed(a.out, **out)
    if comm is not None:
        comm.close()
    if a.world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
