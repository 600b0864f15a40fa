#!/usr/bin/env python3
"""One rank of the real N > 1 test (tests/test_gpu_parallel.py::test_two_ranks_on_one_gpu_*): a process of its own with a
torch.distributed group, its OWN batch shard of a cfg2-shaped DiehlAndCook2015 on the GPU, bindsnet_amd.parallel.sharded_run
for `--inputs` consecutive inputs (reset between), results -> an .npz the parent compares with the oracle.

    python tests/parallel_worker.py --rank R --world W --port P --backend gloo|nccl --fixture NAME --out FILE"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--fixture", default="full_cfg2_dc_n400_b32_poisson")
    ap.add_argument("--inputs", type=int, default=2)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port), RANK=str(a.rank), WORLD_SIZE=str(a.world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    if a.backend == "nccl":
        dist.init_process_group("nccl", rank=a.rank, world_size=a.world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    from bindsnet_amd import parallel
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    g = cases.gold(a.fixture)
    N, B, T = int(g["N"]), int(g["B"]), int(g["T"])
    per = B // a.world
    lo, hi = a.rank * per, (a.rank + 1) * per
    torch.manual_seed(0)                                          # identical replicas
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    mons = {l: Monitor(net.layers[l], ["s"], time=T) for l in ("Ae", "Ai")}
    for l, m in mons.items():
        net.add_monitor(m, l + "_s")
    net.to("cuda")
    feat = net.connections[("X", "Ae")].pipeline[0]
    torch.manual_seed(2 + a.rank)                                 # this rank's own one_spike stream
    out = {"lo": lo, "hi": hi, "N": N, "T": T}
    for k in range(a.inputs):
        shard = np.ascontiguousarray(cases.fixture_input(g, k, T, B)[:, lo:hi])
        parallel.sharded_run(net, {"X": torch.from_numpy(shard).view(T, per, 1, 28, 28).cuda()}, T)
        torch.cuda.synchronize()
        out[f"i{k}_sE"] = np.packbits(mons["Ae"].get("s").cpu().numpy().astype(np.uint8))
        out[f"i{k}_sI"] = np.packbits(mons["Ai"].get("s").cpu().numpy().astype(np.uint8))
        out[f"i{k}_W"] = feat.value.detach().cpu().numpy().copy()
        out[f"i{k}_theta"] = net.layers["Ae"].theta.cpu().numpy().copy()
        out[f"i{k}_plan"] = net.last_plan
        net.reset_state_variables()
    dist.barrier()
    np.savez_compressed(a.out, **out)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
