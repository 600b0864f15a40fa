#!/usr/bin/env python3
"""What sets the period of k_dc2015_async, from the publish times of EVERY compute workgroup (SNN_DC_TIMING_DUMP of the TIMING
instance: [T+1][256][4] per workgroup: [1] wall clock at its publish, [2] own crossings of tile wave 0; slot 255: the arbiter's
"all granules seen" / "winners out").  SNN_DC_TIMING=-1 writes the "lite" dump: no workgroup carries marks, the publish times are as close to the product's as the
instance gets.  python tools/r04_lateness.py dump.bin [T] [G]"""
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
longit = d > np.median(d[v]) + 1.2                             # a long interval: an own crossing in it, or the whole pack late
print(f"period {P:.2f} us;  publish-to-publish intervals: long ones {d[v][longit[v]].mean():.2f} us ({100 * longit[v].mean():.1f} % of them), the others {d[v][~longit[v]].mean():.2f}")
print(f"the pack (median publisher) publishes step t {np.mean(med[2:][v] - out[:-2][v]):.2f} us behind the winners of t-2;  arbiter: last publish -> seen "
      f"{np.mean((seen - pub.max(1))[v]):.2f}, -> winners out +{np.mean((out - seen)[v]):.2f};  the last publisher is {late.max(1)[v].mean():.2f} us behind the pack")
print("   => 2 x period = (pack delay) + (lateness of the last publisher) + (arbiter): "
      f"{np.mean(med[2:][v] - out[:-2][v]):.2f} + {late.max(1)[v].mean():.2f} + {np.mean((out - pub.max(1))[v]):.2f} = "
      f"{np.mean(med[2:][v] - out[:-2][v]) + late.max(1)[v].mean() + np.mean((out - pub.max(1))[v]):.2f}")
sh = ~longit
print("ordinary interval, by the workgroup's lateness at its start (a late workgroup waits for nobody: its own iteration):")
for lo, hi in [(-9, -0.2), (-0.2, 0.2), (0.2, 1), (1, 2), (2, 3), (3, 9)]:
    m = sh & (late[:-1] >= lo) & (late[:-1] < hi)
    m[:5] = False
    m[T - 4:] = False
    if m.sum() > 20:
        print(f"   lateness [{lo:4.1f}, {hi:4.1f}): n = {m.sum():5d}   {d[m].mean():.2f} us")
# ---- the chain, on the crossings that are KNOWN (tile wave 0 records its own; tile wave 1's are not in the dump).  (A long publish-to-publish
#      interval alone is no proof of an own crossing: when the arbiter is late the whole pack has one.)
s3 = hw[:T, :G, 3]
cr = (hw[:T, :G, 2] > 0) | ((s3 > 0x100) & (s3 < 0x200))      # tile wave 0's crossings; the "lite" dump (SNN_DC_TIMING=-1) carries tile wave 1's in slot 3
tt, gg = np.where(cr[6:T - 4])
tt = tt + 6
if len(tt):
    own = np.array([pub[t + 1, g] - pub[t, g] for t, g in zip(tt, gg)])
    c1 = np.array([pub[t + 1, g] - pub[t].max() for t, g in zip(tt, gg)])
    c2 = np.array([pub[t + 1].max() - pub[t + 1, g] for t, g in zip(tt, gg)])
    ahead = np.array([pub[t].max() - pub[t, g] for t, g in zip(tt, gg)])
    print(f"{len(tt)} crossings recorded by tile wave 0: the workgroup had published the step {ahead.mean():.2f} us before the step's LAST publisher "
          f"(10 / 50 / 90 %: {np.percentile(ahead, [10, 50, 90]).round(2)}); its own publish-to-publish interval {own.mean():.2f} us")
    print(f"   its NEXT publish comes {np.median(c1):.2f} us (10 / 90 %: {np.percentile(c1, [10, 90]).round(2)}) behind that last publish -- granules seen, resolution, rest of the "
          f"iteration, next membrane stage -- and is itself within {np.median(c2):.2f} us (75 %: {np.percentile(c2, 75):.2f}) of being the last publish of the next step:")
    print(f"   last publish(t+1) ~ last publish(t) + {np.median(c1):.2f}: the chain that sets the period ({P:.2f})")
fl = hw[:T, :G, 3]
nd = int(((fl == 1) | (fl == 2)).sum())
if nd:
    print(f"SNN_DEFER build: {nd} deferred iterations seen by tile wave 0, the two outcomes gave the same crossings in {int((fl == 2).sum())} of them")
