#!/usr/bin/env python3
"""What sets the period of k_dc2015_async, from the publish times of EVERY compute workgroup (SNN_DC_TIMING_DUMP of the TIMING
instance: [T+1][256][4] per workgroup: [1] wall clock at its publish, [2] own crossings of tile wave 0; slot 255: the arbiter's
"all granules seen" / "winners out").  python tools/r04_lateness.py dump.bin [T] [G]"""
import sys

import numpy as np

path = sys.argv[1]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 250
G = int(sys.argv[3]) if len(sys.argv) > 3 else 100
raw = np.fromfile(path, dtype=np.int64)
hw = raw[24 * (T + 1):].reshape(T + 1, 256, 4)
pub = hw[:T, :G, 1].astype(float) / 100.0                      # us
seen = hw[:T, 255, 0].astype(float) / 100.0
out = hw[:T, 255, 1].astype(float) / 100.0
d = np.diff(pub, axis=0)                                       # publish-to-publish intervals
v = slice(5, T - 3)
P = np.diff(pub.max(1))[v].mean()
med = np.median(pub, axis=1)
late = pub - med[:, None]
longit = d > np.median(d[v]) + 1.2                             # an interval with an own crossing in it
print(f"period {P:.2f} us;  publish-to-publish: with an own crossing {d[v][longit[v]].mean():.2f} us ({100 * longit[v].mean():.1f} % of them, "
      f"{100 * (longit[v].sum(1) > 0).mean():.0f} % of the steps have one), without {d[v][~longit[v]].mean():.2f}")
print(f"the pack (median publisher) publishes step t {np.mean(med[2:][v] - out[:-2][v]):.2f} us behind the winners of t-2;  arbiter: last publish -> seen "
      f"{np.mean((seen - pub.max(1))[v]):.2f}, -> winners out +{np.mean((out - seen)[v]):.2f};  the last publisher is {late.max(1)[v].mean():.2f} us behind the pack")
print("   => 2 x period = (pack delay) + (lateness of the last publisher) + (arbiter): "
      f"{np.mean(med[2:][v] - out[:-2][v]):.2f} + {late.max(1)[v].mean():.2f} + {np.mean((out - pub.max(1))[v]):.2f} = "
      f"{np.mean(med[2:][v] - out[:-2][v]) + late.max(1)[v].mean() + np.mean((out - pub.max(1))[v]):.2f}")
sh = ~longit
print("interval without an own crossing, by the workgroup's lateness at its start (a late workgroup waits for nobody: its own iteration):")
for lo, hi in [(-9, -0.2), (-0.2, 0.2), (0.2, 1), (1, 2), (2, 3), (3, 9)]:
    m = sh & (late[:-1] >= lo) & (late[:-1] < hi)
    m[:5] = False
    m[T - 4:] = False
    if m.sum() > 20:
        print(f"   lateness [{lo:4.1f}, {hi:4.1f}): n = {m.sum():5d}   {d[m].mean():.2f} us")
L = pub.argmax(1)
n = c1 = c2 = 0
chain = []
for t in range(6, T - 3):
    g = L[t]
    n += 1
    if longit[t - 1, g]:
        c1 += 1
        chain.append(pub[t, g] - pub[t - 1].max())
    elif longit[t - 2, g]:
        c2 += 1
print(f"the last publisher of step t had its own crossing at t-1 in {100 * c1 / n:.0f} % of the steps (at t-2: {100 * c2 / n:.0f} %)")
chain = np.array(chain)
print(f"   then: its publish of t comes {chain.mean():.2f} us (10 / 50 / 90 %: {np.percentile(chain, [10, 50, 90]).round(2)}) behind the LAST publish of t-1 -- "
      "the step's granules seen, the resolution, the rest of the iteration, the next membrane stage")
x = []
for t in range(6, T - 3):
    for g in np.where(longit[t])[0]:
        x.append(pub[t].max() - pub[t, g])
x = np.array(x)
print(f"a workgroup that crosses at t published t {x.mean():.2f} us before the step's last publisher (10 / 50 / 90 %: {np.percentile(x, [10, 50, 90]).round(2)}): what it waits for")
fl = hw[:T, :G, 3]
nd = int(((fl == 1) | (fl == 2)).sum())
if nd:
    print(f"SNN_DEFER build: {nd} deferred iterations seen by tile wave 0, the two outcomes gave the same crossings in {int((fl == 2).sum())} of them")
