#!/usr/bin/env python3
"""Round-4 fixtures from the UNMODIFIED reference (build container only):

    python tests/golden/make_golden_r4.py

 run_one_step_ext_current.npz  run(inputs={"X": spikes, "A": ..., "B": ..., "C": ...}, one_step=True): with one_step the reference
                               sets `current_inputs[l] = inputs[l][t]` and then REPLACES that entry by `self._get_inputs(layers=[l])`
                               for every layer a connection feeds (network.py:386-393) -- the external currents of A and B (both fed)
                               are dropped, the one of C (no incoming connection; C -> B) survives.  The synchronous run of the same
                               inputs (currents added to all three) is stored next to it.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402
from make_golden import Input, LIFNodes, Monitor, MulticompartmentConnection, Network, T_, Weight, save  # noqa: E402

nX, nA, nB, nC, B, T = 64, 40, 24, 16, 3, 30


def chain():
    net = Network(dt=1.0, learning=False)
    net.add_layer(Input(n=nX), "X"); net.add_layer(LIFNodes(n=nA, thresh=-60.0), "A")
    net.add_layer(LIFNodes(n=nC, thresh=-58.0), "C"); net.add_layer(LIFNodes(n=nB, thresh=-61.0), "B")
    for k, (src, dst, ns, nd, sc) in enumerate((("X", "A", nX, nA, 0.6), ("A", "B", nA, nB, 0.5), ("B", "A", nB, nA, -1.0), ("C", "B", nC, nB, 1.0))):
        w = synth.uniform_f32(3400 + k, (ns, nd), 0.0, abs(sc)) * np.sign(sc)
        net.add_connection(MulticompartmentConnection(net.layers[src], net.layers[dst], device="cpu",
                                                      pipeline=[Weight("weight", T_(w.astype(np.float32)).clone())]), src, dst)
    mons = {l: Monitor(net.layers[l], ["s", "v"], time=T) for l in ("A", "B", "C")}
    for l, m in mons.items():
        net.add_monitor(m, l)
    return net, mons


def main():
    net, mons = chain()
    sp = synth.dense_spikes(3410, (T, B, nX), 0.12)
    cur = {"A": synth.uniform_f32(3411, (T, B, nA), -1.0, 2.0), "B": synth.uniform_f32(3412, (T, B, nB), 0.0, 1.5),
           "C": synth.uniform_f32(3413, (T, B, nC), 0.0, 3.0)}
    out = {}
    for tag, flag in (("one", True), ("sync", False)):
        net.reset_state_variables()
        net.run({"X": T_(sp), **{k: T_(v) for k, v in cur.items()}}, time=T, one_step=flag)
        for l in ("A", "B", "C"):
            out[f"{tag}_s_{l}"] = np.packbits(mons[l].get("s").numpy().astype(np.uint8))
            out[f"{tag}_v_{l}"] = mons[l].get("v").numpy().copy()
        print(f"  {tag}:", {l: int(mons[l].get("s").sum()) for l in ("A", "B", "C")})
    save("run_one_step_ext_current", **out)


if __name__ == "__main__":
    main()
