#!/usr/bin/env python3
"""Which gfx950 kernels does a source change actually change?

Compiles every csrc/*.hip to device assembly (hipcc --cuda-device-only -S: no GPU needed) at a git revision and in the working tree,
splits the assembly into kernels and compares their bodies with the basic-block labels normalised (their numbers shift when a
kernel is added or removed) and comments dropped.  A refactor that is meant to leave the device code alone -- adding `__host__` to
the accumulators so that tests/hostcheck can run them, moving a helper into a header -- must print "no existing kernel changed"; the
round's GPU minutes then need not be spent on re-testing it.

    python tools/isa_diff.py <git-rev> [file.hip ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bindsnet_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "--cuda-device-only", "-S"]


def kernels(asm_path):
    txt = open(asm_path).read()
    out = {}
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\n\s*\.Lfunc_end\d+:", txt, re.S | re.M):
        body = re.sub(r"BB\d+_", "BBn_", m.group(2))
        body = "\n".join(re.sub(r"\s*;.*$", "", line).rstrip() for line in body.splitlines())
        out[m.group(1)] = body
    return out


def compile_tree(src_dir, files, out_dir):
    procs = []
    for f in files:
        out = os.path.join(out_dir, f.replace(".hip", ".s"))
        procs.append((f, out, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + [f, "-o", out], cwd=src_dir, stdout=subprocess.DEVNULL,
                                               stderr=subprocess.PIPE)))
    res = {}
    for f, out, p in procs:
        err = p.communicate()[1].decode(errors="replace")
        if p.returncode != 0:
            raise SystemExit(f"{f}: hipcc failed\n{err[-2000:]}")
        res[f] = kernels(out)
    return res


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    rev = sys.argv[1]
    files = sys.argv[2:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    with tempfile.TemporaryDirectory() as d:
        old_root = os.path.join(d, "old")
        os.makedirs(old_root)
        subprocess.run(f"git -C {ROOT} archive {rev} bindsnet_amd/csrc include | tar -x -C {old_root}", shell=True, check=True)
        old_csrc = os.path.join(old_root, "bindsnet_amd", "csrc")
        old_files = [f for f in files if os.path.exists(os.path.join(old_csrc, f))]
        os.makedirs(os.path.join(d, "o"))
        os.makedirs(os.path.join(d, "n"))
        old = compile_tree(old_csrc, old_files, os.path.join(d, "o"))
        new = compile_tree(CSRC, files, os.path.join(d, "n"))
    changed = False
    for f in files:
        a, b = old.get(f, {}), new[f]
        for k in sorted(set(a) | set(b)):
            if k not in a:
                print(f"{f}: NEW      {k}")
            elif k not in b:
                print(f"{f}: REMOVED  {k}")
                changed = True
            elif a[k] != b[k]:
                print(f"{f}: CHANGED  {k}")
                changed = True
    print("no existing kernel changed" if not changed else "existing kernels changed (above)")
    return 1 if changed else 0


if __name__ == "__main__":
    sys.exit(main())
