"""The N > 1 path of bench.py on ONE GPU: a real RCCL process group of world size 1.  Checks that
bindsnet_amd.parallel.sharded_run drives the device run (resident plan, normalisation postponed), the
all-reduce / barrier calls work on device tensors, and that with a single rank the result equals a plain
run() up to the rounding of `before + (after - before)`."""
import os
import socket

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def build():
    from bindsnet_amd.models import DiehlAndCook2015
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=100, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    net.connections[("X", "Ae")].pipeline[0].value.data.copy_(torch.from_numpy(synth.weights_q12(10, 784, 100)))
    net.to(DEV)
    return net


def test_sharded_run_single_rank_rccl():
    import torch.distributed as dist
    from bindsnet_amd import parallel
    T, B = 60, 8
    spikes = [torch.from_numpy(synth.spike_train(20 + r, T, B, 784)).view(T, B, 1, 28, 28).to(DEV) for r in range(2)]
    ref = build()
    for r in range(2):
        torch.manual_seed(3 + r)
        ref.run({"X": spikes[r]}, time=T)
    saved = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE")}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    except Exception as e:                                  # no usable RCCL on this box: an infrastructure matter
        pytest.skip(f"RCCL process group could not be created: {e}")
    try:
        net = build()
        for r in range(2):
            torch.manual_seed(3 + r)
            parallel.sharded_run(net, {"X": spikes[r]}, T)
            assert net.last_plan.startswith("dc2015-resident")
        dist.barrier()
        t = torch.ones(4, device=DEV)
        dist.all_reduce(t)
        assert float(t.sum()) == 4.0
    finally:
        dist.destroy_process_group()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    Wr = ref.connections[("X", "Ae")].pipeline[0].value.cpu().numpy()
    Ws = net.connections[("X", "Ae")].pipeline[0].value.cpu().numpy()
    np.testing.assert_allclose(Ws, Wr, rtol=0, atol=2e-6)
    np.testing.assert_allclose(Ws.sum(0), 78.4, rtol=1e-5)
    np.testing.assert_allclose(net.layers["Ae"].theta.cpu().numpy(), ref.layers["Ae"].theta.cpu().numpy(), rtol=0, atol=1e-6)
    assert net.connections[("X", "Ae")].pipeline[0].norm == 78.4
