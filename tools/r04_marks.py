#!/usr/bin/env python3
"""Offline view of SNN_DC_TIMING_DUMP (k_dc2015_async<TIMING>): per-mark mean time from the iteration start of the instrumented compute
workgroup (wave 0, lane 0), split into iterations without / with an own crossing.  python tools/r04_marks.py dump.bin T"""
import sys
import numpy as np

path, T = sys.argv[1], int(sys.argv[2])
raw = np.fromfile(path, dtype=np.int64)
h = raw[: 24 * (T + 1)].reshape(T + 1, 24)
order = [0, 11, 18, 9, 10, 7, 12, 8, 13, 4, 14, 15, 5, 6, 1, 16, 2]
names = {0: "iteration start", 11: "(split build) older than the prefetch", 18: "(split build) the prefetch", 9: "abort word read", 10: "won mask read", 7: "winners(t-2) decoded", 12: "membrane update + ballot", 8: "published",
         13: "at barrier M", 4: "behind M", 14: "crossing state read", 15: "in front of the current pass", 5: "current pass done", 6: "untouched rows done",
         1: "resolution done", 16: "digest stored", 2: "behind B"}
rows = []
for t in range(3, T - 1):
    r = h[t]
    if r[0] == 0 or h[t + 1][0] == 0:
        continue
    rows.append([(r[k] - r[0]) / 100.0 for k in order] + [(h[t + 1][0] - r[0]) / 100.0])
a = np.array(rows)
gap = a[:, order.index(15)] - a[:, order.index(14)]            # a crossing workgroup prepares the won branch in between
plain = gap < np.median(gap) + 0.15
for label, sel in (("plain", plain), ("crossing", ~plain)):
    if sel.sum() == 0:
        continue
    m = a[sel].mean(0)
    print(f"--- {label} iterations: {sel.sum()}  (mean iteration {m[-1]:.2f} us)")
    prev = 0.0
    for k, v in zip(order, m[:-1]):
        print(f"  {names[k]:32s} {v:6.2f}  (+{v - prev:.2f})")
        prev = v
w2 = {19: "wave 2: crossing state read", 20: "wave 2: touched rows done (at P)", 21: "wave 2: current pass done", 22: "wave 2: untouched rows done", 23: "wave 2: digest stored (at B)"}
rows2 = []
for t in range(3, T - 2):
    r = h[t]
    if r[0] == 0 or r[23] == 0:
        continue
    rows2.append([(r[k] - r[0]) / 100.0 if r[k] else np.nan for k in sorted(w2)])
if rows2:
    a2 = np.array(rows2)
    sel = ~(a2[:, 1] > a2[:, 0])                                 # mark 20 is written by a crossing workgroup only: older than mark 19 = plain
    for label, ss in (("plain", sel), ("crossing", ~sel)):
        if ss.sum():
            m2 = np.nanmean(a2[ss], 0)
            print(f"--- wave 2, {label} iterations ({ss.sum()}; us from wave 0's iteration start): " + ", ".join(f"{w2[k].split(': ')[1]} {v:.2f}" for k, v in zip(sorted(w2), m2)))
st = h[3:T - 1, 17]
print("winners prefetch at use: fresh %d, stale %d, none %d" % ((st == 0).sum(), (st == 1).sum(), (st == 2).sum()))
