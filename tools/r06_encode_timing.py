#!/usr/bin/env python3
"""Where a device-side Poisson encoding of one MNIST-sized sample goes: the kernel alone (HIP events), the package call, the literal script's route
(host tensor in, host tensor out).  python tools/r06_encode_timing.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bindsnet_amd import synth  # noqa: E402
from bindsnet_amd.encoding import PoissonEncoder, poisson, poisson_device  # noqa: E402
from bindsnet_amd.ops import encode_poisson  # noqa: E402

img = torch.from_numpy(synth.uniform_f32(9, (1, 28, 28), 0.0, 1.0)) * 128.0 * (torch.from_numpy(synth.uniform_f32(10, (1, 28, 28), 0.0, 1.0)) < 0.19)
xd = img.flatten().cuda()
out = {}
for name, fn in (("kernel only (device tensor in, HIP events)", None),
                 ("poisson_device(host tensor) -> device tensor", lambda: poisson_device(img, time=250, device="cuda")),
                 ("PoissonEncoder via SNN_ENCODE_DEVICE (host in, host out)", lambda: PoissonEncoder(time=250, dt=1.0)(img)),
                 ("host poisson()", lambda: poisson(img, time=250))):
    if fn is None:
        for _ in range(5):
            encode_poisson(xd, 250, 1.0, 3, "cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(50):
            encode_poisson(xd, 250, 1.0, 3 + k, "cuda")
        e1.record(); torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) / 50, 4)
        continue
    if "SNN_ENCODE_DEVICE" in name:
        os.environ["SNN_ENCODE_DEVICE"] = "cuda"
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    out[name] = round((time.perf_counter() - t0) / 50 * 1e3, 4)
    os.environ.pop("SNN_ENCODE_DEVICE", None)
print(json.dumps({"ms_per_sample": out, "sample": "1x28x28, 19 % of the pixels lit at up to 128 Hz, 250 timesteps"}))
