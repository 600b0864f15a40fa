"""Test-side alias of bindsnet_amd/synth.py (the generators live in the package so bench.py and tools/ do not
import the test tree).  Loaded by path: the golden-fixture generators import this with `bindsnet` bound to the
REFERENCE package and must not pull in bindsnet_amd."""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location(
    "_bindsnet_amd_synth", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bindsnet_amd", "synth.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("_")})
