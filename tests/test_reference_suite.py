"""The reference's OWN test files, unmodified, with `bindsnet` resolving to this package (the alias of bindsnet/__init__.py) on a
machine without a GPU: the files whose imports lie inside the path's scope -- test/models/test_models.py (constructors of
TwoLayerNetwork / DiehlAndCook2015), test/encoding/test_encoding.py (bernoulli / poisson and their loaders),
test/network/test_network.py (Network.run on an empty network, add_*, save / load).  The other files of the reference's suite
import the out-of-scope zoo at module level (IFNodes, SRM0Nodes, Rmax, MaxPool2dConnection, SparseConnection, ...).

Runs where the reference checkout exists (the build container); nothing is written into it (no bytecode, no pytest cache)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TESTS = "/root/reference/test"
FILES = ["models/test_models.py", "encoding/test_encoding.py", "network/test_network.py"]


@pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="no reference checkout on this machine")
def test_reference_test_files_pass_against_the_alias_package(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT, PYTHONDONTWRITEBYTECODE="1")
    cmd = [sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", f"--rootdir={tmp_path}"] + [os.path.join(REF_TESTS, f) for f in FILES]
    out = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout and "error" not in out.stdout.lower(), out.stdout[-2000:]
