#!/usr/bin/env python3
"""Developer tool: find the first timestep at which a D&C plan's learned weights leave the oracle's (bisection over the
run length; both sides normalise at the end of a run).  python tools/debug_dc_parity.py [fixture] [plans...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
import oracle  # noqa: E402
from bindsnet_amd import _lib  # noqa: E402
from bindsnet_amd.models import DiehlAndCook2015  # noqa: E402
from bindsnet_amd.network.monitors import Monitor  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "full_cfg2_dc_n400_b32_poisson"
plans = sys.argv[2:] or ["resident", "per-step", "generic", "auto"]
MODE = {"auto": 0, "generic": 1, "per-step": 2, "resident": 3}
g = cases.gold(name)
N, B, T = int(g["N"]), int(g["B"]), int(g["T"])
spikes = cases.fixture_input(g, 0, T, B)
per = spikes.sum(2)
print("events per sample-step: mean %.2f max %d; steps with a sample > 16 events: %d" % (per.mean(), per.max(), int((per.max(1) > 16).sum())))


def gpu(plan, Tk):
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    mon = Monitor(net.layers["Ae"], ["s"], time=Tk)
    net.add_monitor(mon, "Ae_s")
    net.to("cuda")
    _lib.lib().snn_set_plan_mode(MODE[plan])
    torch.manual_seed(2)
    net.run({"X": torch.from_numpy(spikes[:Tk]).view(Tk, B, 1, 28, 28).cuda()}, time=Tk)
    torch.cuda.synchronize()
    _lib.lib().snn_set_plan_mode(0)
    return (net.connections[("X", "Ae")].pipeline[0].value.cpu().numpy(), mon.get("s").cpu().numpy().reshape(Tk, B, N).astype(np.uint8),
            net.last_plan, getattr(net, "lean_retries", 0))


def cpu(Tk):
    P = oracle.eth_mnist_dc_params(N, B, Tk)
    torch.manual_seed(0)
    st = oracle.eth_mnist_dc_state(N, B, (0.3 * torch.rand(784, N)).numpy())
    Q = oracle.exp_noise(2, B * N * (Tk + 1))
    cur = np.zeros(1, np.int64)
    rasE, _ = oracle.run_dc2015(P, st, np.ascontiguousarray(spikes[:Tk]), Q, cur)
    return st["W_xe"], rasE


cache = {}


def ref(Tk):
    if Tk not in cache:
        cache[Tk] = cpu(Tk)
    return cache[Tk]


for plan in plans:
    W, ras, used, retries = gpu(plan, T)
    Wr, rr = ref(T)
    same = np.array_equal(W.view(np.uint32), Wr.view(np.uint32))
    print(f"[{plan}] plan used {used}, lean retries {retries}: weights {'EXACT' if same else 'DIFFER max %.3g' % np.abs(W - Wr).max()}, "
          f"rasters {'exact' if np.array_equal(ras, rr) else 'DIFFER'}")
    if same:
        continue
    lo, hi = 1, T                       # weights exact at run length lo - 1 (vacuous), differ at hi
    while lo < hi:
        mid = (lo + hi) // 2
        Wm = gpu(plan, mid)[0]
        if np.array_equal(Wm.view(np.uint32), ref(mid)[0].view(np.uint32)):
            lo = mid + 1
        else:
            hi = mid
    Tk = lo
    Wm, rasm, _, _ = gpu(plan, Tk)
    Wk, rk = ref(Tk)
    bad = np.argwhere(Wm.view(np.uint32) != Wk.view(np.uint32))
    cols = sorted(set(bad[:, 1].tolist()))
    print(f"  first run length with a difference: {Tk} (timestep {Tk - 1}); {len(bad)} elements in columns {cols[:20]}")
    t = Tk - 1
    for j in cols[:6]:
        rows = bad[bad[:, 1] == j][:, 0]
        print(f"   column {j}: {len(rows)} rows differ (first {rows[:8].tolist()}), max |d| {np.abs(Wm[:, j] - Wk[:, j]).max():.3g}; "
              f"Ae spikes in this column at t={t}: samples {np.nonzero(rk[t, :, j])[0].tolist()}; at t-1: {np.nonzero(rk[t - 1, :, j])[0].tolist() if t else []}")
    print(f"   step {t}: events per sample {per[t].tolist()}")
    print(f"   exc spikes at step {t}: {[(int(b), int(j)) for b, j in np.argwhere(rk[t])]}")
