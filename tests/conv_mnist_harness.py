"""Run one of the reference's example scripts (examples/mnist/conv_mnist.py, reservoir.py) ITSELF (unmodified, through runpy) with: the torchvision stand-in installed, a
non-interactive matplotlib backend whose plt.pause does not sleep, a seeded CPU generator, and `Network.run` wrapped so that every
input's Y raster is recorded.  `bindsnet` is whatever the caller put into sys.modules: the reference (fixture generator) or this
package's alias (tests)."""
import hashlib
import os
import runpy
import sys
import warnings

import numpy as np
import torch

import tv_shim

REF_SHA = "see tests/golden/conv_mnist_literal.npz (script_sha)"


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_script(path, network_module, argv, seed=0, monitor="Y_spikes", result=None):
    os.environ["MPLBACKEND"] = "Agg"
    import matplotlib
    matplotlib.use("Agg", force=True)
    import matplotlib.pyplot as plt
    tv_shim.install()
    records = []
    Network = network_module.Network
    orig_run, orig_pause = Network.run, plt.pause

    def recording_run(self, inputs, time, *a, **k):
        out = orig_run(self, inputs, time, *a, **k)
        mon = self.monitors.get(monitor)
        if mon is not None:
            s = mon.get("s").detach().cpu().numpy().astype(np.uint8)
            records.append((sha(np.packbits(s)), int(s.sum())))
        return out

    Network.run = recording_run
    plt.pause = lambda interval: None                    # (the script pauses one second per sample)
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = [path] + list(argv)
    torch.manual_seed(seed)                              # in --gpu mode the script seeds only the CUDA generator (conv_mnist.py:68-70)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            g = runpy.run_path(path, run_name="__main__")
    finally:
        Network.run, plt.pause = orig_run, orig_pause
        sys.argv = old_argv
        os.chdir(old_cwd)
        plt.close("all")
    net = g["network"]
    if result is not None:
        return dict(raster_sha=[r[0] for r in records], raster_sum=[r[1] for r in records], plan=getattr(net, "last_plan", None), **result(g))
    return dict(raster_sha=[r[0] for r in records], raster_sum=[r[1] for r in records], W=g["conv_conn"].w.detach().cpu().numpy().copy(),
                theta=net.layers["Y"].theta.detach().cpu().numpy().copy(), plan=getattr(net, "last_plan", None))
