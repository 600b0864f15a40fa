"""Run an eth_mnist-style script (the reference's examples/mnist/eth_mnist.py itself, or tests/eth_mnist_flow.py)
through runpy with: the torchvision stand-in installed, a non-interactive matplotlib backend, a seeded CPU generator,
and `Network.run` wrapped so that every input's excitatory raster is recorded.  Returns what a parity check needs."""
import hashlib
import os
import runpy
import sys
import warnings

import numpy as np
import torch

import tv_shim


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_script(path, network_module, argv, seed=0):
    """network_module: the module whose `Network.run` the script ends up calling (bindsnet.network.network of the
    reference, or bindsnet_amd.network.network)."""
    os.environ["MPLBACKEND"] = "Agg"
    import matplotlib
    matplotlib.use("Agg", force=True)
    tv_shim.install()
    records = []
    Network = network_module.Network
    orig_run = Network.run

    def recording_run(self, inputs, time, *a, **k):
        out = orig_run(self, inputs, time, *a, **k)
        mon = self.monitors.get("Ae_spikes")
        if mon is not None:
            s = mon.get("s").detach().cpu().numpy().astype(np.uint8)
            records.append((sha(np.packbits(s)), int(s.sum())))
        return out

    Network.run = recording_run
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = [path] + list(argv)
    torch.manual_seed(seed)          # in --gpu mode the script seeds only the CUDA generator (eth_mnist.py:69-72)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            g = runpy.run_path(path, run_name="__main__")
    finally:
        Network.run = orig_run
        sys.argv = old_argv
        os.chdir(old_cwd)
        import matplotlib.pyplot as plt
        plt.close("all")
    net = g["network"]
    W = net.connections[("X", "Ae")].pipeline[0].value.detach().cpu().numpy()
    return dict(raster_sha=[r[0] for r in records], raster_sum=[r[1] for r in records], W=W,
                theta=net.layers["Ae"].theta.detach().cpu().numpy(), accuracy=dict(g["accuracy"]),
                assignments=g["assignments"].detach().cpu().numpy(), proportions=g["proportions"].detach().cpu().numpy())
