#!/usr/bin/env python3
"""Soak of the D&C resident plan (third-generation lean form where it applies) against the generic per-operator plan, bit for bit,
on seeded random sizes at input densities where the lean form stays in charge (longer runs than the default fuzz family: three
inputs of 40..250 timesteps with learning on, weights and thresholds carried over).  Prints one line per case with the resident form
that ran (3 = k_dc2015_async) and the retry counters.  python tools/r04_soak.py [cases] [first_seed] [sections]
(`sections`, round 5: every case a third time inside a Network.pipelined() section)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
import test_gpu_fused_stress as dc  # noqa: E402
from bindsnet_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
PIPE = len(sys.argv) > 3 and sys.argv[3] == "sections"      # python tools/r04_soak.py <cases> <first_seed> sections
bad = 0
forms = {}
t0 = time.time()
for seed in range(first, first + n):
    rs = np.random.RandomState(9000 + seed)
    N = int(rs.choice([16, 36, 64, 100, 128, 200, 256, 333, 400, 512, 625, 800, 1000, 1024]))
    B = int(rs.choice([1, 4, 8, 16, 24, 31, 32]))
    T = int(rs.choice([40, 64, 100, 150, 250]))
    Nin = int(rs.choice([784, 784, 784, 400, 1024, 196]))
    dens = float(rs.choice([0.006, 0.012, 0.02, 0.03]))
    wsc = float(rs.choice([0.3, 0.3, 0.6, 1.0]))
    inh = float(rs.choice([120.0, 17.5, 60.0]))
    exc = float(rs.choice([22.5, 22.5, 30.0]))
    additive = bool(rs.rand() < 0.15)
    shape = {784: (1, 28, 28), 400: (1, 20, 20), 1024: (1, 32, 32), 196: (1, 14, 14)}[Nin]
    spikes = [synth.dense_spikes(700 + 13 * seed + r, (T, B, Nin), dens) for r in range(3)]
    kw = dict(w_scale=wsc, n_inputs=3, learning=True, Nin=Nin, shape=shape, inh=inh, additive=additive, exc=exc)
    if seed >= 1000:                                             # second family: learning off now and then, one-sided / larger learning rates
        kw["learning"] = bool(rs.rand() < 0.85)
        kw["nu"] = [(1e-4, 1e-2), (1e-4, 1e-2), (0.0, 1e-2), (1e-3, 0.0), (5e-4, 5e-2)][int(rs.randint(5))]
    res, plan = dc.run(0, N, B, T, spikes, **kw)
    form = _lib.lib().snn_dc2015_last_form()
    net = dc.run.last_net
    retries = (getattr(net, "lean_retries", 0), getattr(net, "resident_retries", 0))
    gen, plan_g = dc.run(1, N, B, T, spikes, **kw)
    ok = True
    try:
        dc.same(res, gen)
        if PIPE:                                                 # round 5: the same case inside a Network.pipelined() section
            sec, _ = dc.run(0, N, B, T, spikes, pipelined=True, **kw)
            dc.same(sec, gen)
    except AssertionError as e:
        ok = False
        bad += 1
        print("MISMATCH:", str(e)[:300])
    forms[form] = forms.get(form, 0) + 1
    nsp = int(sum(r["sE"].sum() for r in res))
    lrn, nu_ = int(kw["learning"]), kw.get("nu", (1e-4, 1e-2))
    print(f"seed {seed}: N={N} B={B} T={T} Nin={Nin} dens={dens} w={wsc} inh={inh} exc={exc} additive={int(additive)} learning={lrn} nu={nu_} -> {plan} form {form} retries {retries} "
          f"Ae spikes {nsp} {'OK' if ok else 'DIFFERENT'}", flush=True)
print(f"{n} cases, {bad} different, forms {forms}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
