#!/usr/bin/env python3
"""Static view of a kernel's device code by SOURCE LINE: compiles one csrc/*.hip with line tables (no GPU needed) and prints, per
source-line range of the chosen kernel, how many instructions of each class the compiler emitted there.

    python tools/isa_by_line.py snn_dc2015_async.hip k_dc2015_asyncILb0 596 945 [bucket]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bindsnet_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "--cuda-device-only", "-S",
         "-gline-tables-only"]


def classify(op):
    if op.startswith("v_readlane") or op.startswith("v_writelane"):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def main():
    src, kern, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    bucket = int(sys.argv[5]) if len(sys.argv) > 5 else 10
    extra = sys.argv[6:]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [src, "-o", out], cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read().splitlines()
    files = {}
    inside = False
    cur = None
    main_file = None
    hist = collections.defaultdict(collections.Counter)
    for line in txt:
        m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", line)
        if m:
            name = m.group(3) or m.group(2)
            files[int(m.group(1))] = name
            if name.endswith(src):
                main_file = int(m.group(1))
            continue
        if re.match(r"^_Z\w+:", line):
            inside = kern in line
            continue
        if not inside:
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", line)
        if not m or line.strip().startswith("."):
            continue
        op = m.group(1)
        if cur is None:
            continue
        f, ln = cur
        key = (ln // bucket) * bucket if (f == main_file and lo <= ln <= hi) else (-1 if f == main_file else -2 - f)
        hist[key][classify(op)] += 1
    cols = ["valu", "salu", "lane", "lds", "vmem", "smem", "wait", "branch", "barrier", "other"]
    print("lines".ljust(14) + "".join(c.rjust(8) for c in cols) + "   total")
    for key in sorted(hist):
        if key >= 0:
            name = f"{key}-{key + bucket - 1}"
        elif key == -1:
            name = "(other lines)"
        else:
            name = os.path.basename(files.get(-2 - key, "?"))[:13]
        h = hist[key]
        print(name.ljust(14) + "".join(str(h[c]).rjust(8) for c in cols) + str(sum(h.values())).rjust(8))


if __name__ == "__main__":
    main()
