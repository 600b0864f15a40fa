"""Multi-process (gloo, world size 2, CPU) coverage of the batch-shard merge used at N > 1 GPUs:
bindsnet_amd.parallel.merge_deltas / sharded_run.  The device run itself is stubbed (no GPU here); what
is tested is the distributed logic: normalisation postponed until after the merge, weight and theta
deltas summed over ranks with one all-reduce, clamp, then normalise -- identical on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bindsnet_amd import parallel
        from bindsnet_amd.models import DiehlAndCook2015
        torch.manual_seed(0)                              # identical replicas
        net = DiehlAndCook2015(n_inpt=64, n_neurons=16, exc=22.5, inh=120, dt=1.0, norm=7.0, theta_plus=0.05,
                               inpt_shape=(1, 8, 8))
        feat = net.connections[("X", "Ae")].pipeline[0]
        W0 = feat.value.data.clone()
        calls = []

        def fake_run(inputs, time, **kw):                 # stands in for the device run of this rank's shard
            assert feat.norm is None, "normalisation must be postponed until after the merge"
            g = torch.Generator().manual_seed(100 + rank)
            feat.value.data += 0.01 * torch.rand(W0.shape, generator=g) * (rank + 1)
            net.layers["Ae"].theta += 0.05 * (rank + 1)
            calls.append("run")

        def fake_normalize():                             # CPU stand-in for snn_normalize (signed column sums)
            calls.append("norm")
            cs = feat.value.data.sum(0, keepdim=True)
            cs[cs == 0] = 1.0
            feat.value.data *= feat.norm / cs

        net.run = fake_run
        feat.normalize = fake_normalize
        # 1) raw merge
        b = [torch.ones(3), torch.zeros(2, 2)]
        a = [torch.ones(3) + (rank + 1), torch.full((2, 2), float(rank))]
        parallel.merge_deltas(b, a)
        assert torch.equal(a[0], torch.ones(3) + 3.0) and torch.equal(a[1], torch.full((2, 2), 1.0))
        # 2) whole sharded step
        parallel.sharded_run(net, {"X": torch.zeros(2, 1, 1, 8, 8, dtype=torch.uint8)}, 2)
        assert calls == ["run", "norm"] and feat.norm == 7.0
        exp = W0.clone()
        for r in range(world):
            g = torch.Generator().manual_seed(100 + r)
            exp += 0.01 * torch.rand(W0.shape, generator=g) * (r + 1)
        exp.clamp_(0.0, 1.0)
        exp *= 7.0 / exp.sum(0, keepdim=True)
        torch.testing.assert_close(feat.value.data, exp, rtol=0, atol=1e-6)
        assert torch.allclose(net.layers["Ae"].theta, torch.full((16,), 0.05 * 3))
        out.put((rank, feat.value.data.numpy().tobytes()))
    finally:
        dist.destroy_process_group()


def test_sharded_merge_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == res[1], "replicas diverged after the merge"
