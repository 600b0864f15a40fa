"""Launch the ranks of the exact batch-sharded mode (tests/exact_worker.py) and compare what they leave with the REFERENCE's
single-process run of the global batch (a D&C fixture of tests/golden): shared by the CPU (gloo, host operators) and the
GPU (gloo between two processes on the one device, C-ABI operators) tests."""
import os
import socket
import subprocess
import sys

import numpy as np

import cases
from cases import check_packed, unpack

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def launch(world, fixture, device, tmp_path, runs=0, timeout=600, threads=2, native=False, mode="per-step"):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    outs = [str(tmp_path / f"exact_{fixture}_w{world}_r{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "exact_worker.py"), "--rank", str(r), "--world", str(world), "--port", str(port),
                               "--device", device, "--fixture", fixture, "--runs", str(runs), "--threads", str(threads), "--out", outs[r], "--mode", mode] + (["--native"] if native else []),
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=timeout)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            p.kill()
            logs.append("timeout")
    assert [p.returncode for p in procs] == [0] * world, "\n".join(logs)[-4000:]
    if os.environ.get("SNN_EXACT_TIMING") == "1":
        print("\n".join(l for log in logs for l in log.splitlines() if l.startswith("rank ")))
    return [np.load(o) for o in outs]


def check_against_reference(res, fixture, runs=0):
    """Rows of all ranks side by side == the reference's global batch: rasters, weights, theta, membrane state, traces --
    bit for bit -- and every rank's host generator stands where the reference's does."""
    import torch
    g = cases.gold(fixture)
    N, B, T = int(g["N"]), int(g["B"]), int(g["T"])
    runs = runs or int(g["runs"])
    world = len(res)
    per = B // world
    cat = lambda key: np.concatenate([r[key] for r in res], axis=0)       # noqa: E731
    for r in range(runs):
        for key, n in (("sE", N), ("sI", N)):
            got = np.concatenate([unpack(x[f"r{r}_{key}"], (T, per, n)) for x in res], axis=1)
            np.testing.assert_array_equal(got, unpack(g[f"r{r}_{key}"], (T, B, n)), err_msg=f"run {r} raster {key}")
        if f"r{r}_in" in g.files:
            got = np.concatenate([unpack(x[f"r{r}_sX"], (T, per, 784)) for x in res], axis=1)
            np.testing.assert_array_equal(got, cases.fixture_input(g, r, T, B), err_msg=f"run {r} input raster")
        for x in res:                                              # replicated tensors: identical on every rank, equal to the reference's
            assert cases.sha(x[f"r{r}_W"]) == str(g[f"r{r}_W_sha"]), f"run {r} weights"
            np.testing.assert_array_equal(x[f"r{r}_theta"].view(np.uint32), g[f"r{r}_theta"].view(np.uint32), err_msg=f"run {r} theta")
        for key in ("vE", "rE", "xE", "xX", "vI", "rI"):
            check_packed(g, f"r{r}_{key}", cat(f"r{r}_{key}"))
    if "probe_after" in g.files:
        expect = g["probe_after"]
    else:                                                          # run_dc_* fixtures: manual_seed(2 + r) before run r, `consumed` draws in it
        torch.manual_seed(2 + runs - 1)
        n = int(g[f"r{runs - 1}_consumed"])
        if n:
            torch.empty(n).exponential_(1)
        expect = torch.rand(4).numpy()
    if runs == int(g["runs"]) or "probe_after" not in g.files:
        for x in res:
            np.testing.assert_array_equal(x["probe_after"], expect, err_msg="host generator position")
