"""bench.py's cpu_baseline leg on the REAL reference: the staging recipe (oracle/stage_ref.py) and the subprocess leg (oracle/ref_cpu_leg.py).
The repository holds only the sha256 manifest of the reference files; the byte copies live under the git-ignored oracle/_ref/."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import stage_ref  # noqa: E402


def test_manifest_is_committed_and_names_the_hot_path_packages():
    with open(stage_ref.MANIFEST) as f:
        m = json.load(f)["sha256"]
    for rel in ("network/network.py", "network/nodes.py", "network/topology.py", "network/topology_features.py", "learning/learning.py",
                "learning/MCC_learning.py", "models/models.py", "encoding/encodings.py", "utils.py"):
        assert rel in m and len(m[rel]) == 64


@pytest.mark.skipif(not os.path.isdir(stage_ref.REF_ROOT), reason="the reference checkout exists only in the build container")
def test_manifest_matches_the_reference_checkout():
    with open(stage_ref.MANIFEST) as f:
        assert json.load(f)["sha256"] == stage_ref.manifest()


@pytest.mark.skipif(not stage_ref.verify(), reason="oracle/_ref not staged (python __graft_entry__.py build stages it in the build container)")
def test_reference_leg_runs_the_staged_reference(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bindsnet_amd import synth
    T, B = 30, 2
    x = np.stack([h.reshape(T, B, 784) for h in synth.poisson_mnist_like(B, T, 2, seed=1)])
    np.save(tmp_path / "in.npy", x)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_cpu_leg.py"), "--inputs", str(tmp_path / "in.npy"),
                          "--out", str(tmp_path / "rec.npz"), "--n", "100", "--whole", "2", "--short-legs", "0"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["kind"] == "reference" and len(out["per_input_timesteps_per_s"]) == 2 and out["min"] <= out["median"] <= out["max"]
    z = np.load(tmp_path / "rec.npz")
    assert z["r1_W"].shape == (784, 100) and z["r0_theta"].shape == (100,)
    # the same two inputs through the CPU oracle's operator-for-operator port: identical rasters and weights
    import torch
    from oracle.torch_cpu_ref import DcTorchRef
    torch.manual_seed(0)
    r = DcTorchRef(n_inpt=784, n_neurons=100)
    r.set_batch(B)
    torch.manual_seed(2)
    for k in range(2):
        rec = r.run(torch.from_numpy(x[k]))
        got = np.unpackbits(z[f"r{k}_Ae"])[:T * B * 100].reshape(T, B, 100)
        assert np.array_equal(got, rec["Ae"].numpy().astype(np.uint8))
        assert np.array_equal(z[f"r{k}_W"], r.W_xe.numpy())
        r.reset()
