#!/usr/bin/env python3
"""Round-3 fixtures from the UNMODIFIED reference (build container only):

    python tests/golden/make_golden_r3.py

 run_ext_current.npz     run(inputs={"X": spikes, "A": currents, "B": currents}) -- external input currents into
                         non-Input layers (network.py:386-392) on Input -> A -> B with MulticompartmentConnections
 run_one_step_clamp.npz  run(..., one_step=True, clamp=..., unclamp=...) -- a clamped layer's spikes feed the layers
                         behind it in the same timestep (network.py:388-429)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402
import make_golden as mg  # noqa: E402
from make_golden import Input, LIFNodes, Monitor, MulticompartmentConnection, Network, T_, Weight, save  # noqa: E402

nX, nA, nB, B, T = 64, 40, 24, 3, 30


def chain():
    net = Network(dt=1.0, learning=False)
    X, A, Bl = Input(n=nX), LIFNodes(n=nA, thresh=-60.0), LIFNodes(n=nB, thresh=-61.0)
    net.add_layer(X, "X"); net.add_layer(A, "A"); net.add_layer(Bl, "B")
    for k, (src, dst, ns, nd, sc) in enumerate((("X", "A", nX, nA, 2.0), ("A", "B", nA, nB, 0.35), ("B", "A", nB, nA, -1.0))):
        w = synth.uniform_f32(3200 + k, (ns, nd), 0.0, abs(sc)) * np.sign(sc)
        c = MulticompartmentConnection(net.layers[src], net.layers[dst], device="cpu", pipeline=[Weight("weight", T_(w.astype(np.float32)).clone())])
        net.add_connection(c, src, dst)
    mons = {l: Monitor(net.layers[l], ["s", "v"], time=T) for l in ("A", "B")}
    for l, m in mons.items():
        net.add_monitor(m, l)
    return net, mons


def ext_current_case():
    net, mons = chain()
    sp = synth.dense_spikes(3210, (T, B, nX), 0.10)
    cA = synth.uniform_f32(3211, (T, B, nA), -1.0, 4.0)
    cB = synth.uniform_f32(3212, (T, B, nB), 0.0, 2.5)
    out = {}
    net.run({"X": T_(sp), "A": T_(cA), "B": T_(cB)}, time=T)
    for l in ("A", "B"):
        out[f"s_{l}"] = np.packbits(mons[l].get("s").numpy().astype(np.uint8))
        out[f"v_{l}"] = mons[l].get("v").numpy().copy()
    print("  ext currents: A spikes", int(mons["A"].get("s").sum()), "B spikes", int(mons["B"].get("s").sum()))
    save("run_ext_current", **out)


def one_step_clamp_case():
    net, mons = chain()
    sp = synth.dense_spikes(3220, (T, B, nX), 0.15)
    clampA = torch.zeros(nA, dtype=torch.bool); clampA[::7] = True
    unclampA = torch.zeros(nA, dtype=torch.bool); unclampA[3::5] = True
    out = {}
    for tag, flag in (("one", True), ("sync", False)):
        net.reset_state_variables()
        net.run({"X": T_(sp)}, time=T, one_step=flag, clamp={"A": clampA}, unclamp={"A": unclampA})
        for l in ("A", "B"):
            out[f"{tag}_{l}"] = np.packbits(mons[l].get("s").numpy().astype(np.uint8))
        print(f"  one_step={flag} with clamp: A spikes {int(mons['A'].get('s').sum())}, B spikes {int(mons['B'].get('s').sum())}")
    assert not np.array_equal(out["one_B"], out["sync_B"]), "the fixture must tell the two modes apart"
    save("run_one_step_clamp", **out)


if __name__ == "__main__":
    ext_current_case()
    one_step_clamp_case()
