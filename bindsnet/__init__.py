"""`bindsnet`: import alias of `bindsnet_amd`, the MI355X-native implementation of BindsNET's Network.run() hot path.

Scripts written against BindsNET -- `from bindsnet.network import Network`, `from bindsnet.models import
DiehlAndCook2015`, `from bindsnet.learning.MCC_learning import PostPre`, ... (examples/mnist/eth_mnist.py) -- resolve
to the `bindsnet_amd` modules: `bindsnet.X.Y` IS the module object `bindsnet_amd.X.Y` (one copy, so isinstance checks
and module state agree whichever name was used).  Sub-packages load on first use; `bindsnet.datasets` needs torchvision
and `bindsnet.analysis.plotting` matplotlib only when they are actually imported.
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import sys

import bindsnet_amd as _impl

__version__ = _impl.__version__
ROOT_DIR = _impl.ROOT_DIR
_PREFIX, _REAL = "bindsnet.", "bindsnet_amd."


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = _REAL + fullname[len(_PREFIX):]
        try:
            spec = importlib.util.find_spec(real)
        except (ImportError, AttributeError, ValueError):
            return None
        if spec is None:
            return None
        return importlib.machinery.ModuleSpec(fullname, self, is_package=spec.submodule_search_locations is not None)

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_PREFIX):])     # the existing module object itself

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())
__path__ = []            # a package without files of its own: every submodule comes from the finder above

_SUBPACKAGES = ("network", "learning", "models", "encoding", "evaluation", "datasets", "analysis", "utils", "parallel")


def __getattr__(name):
    if name in _SUBPACKAGES:
        return importlib.import_module(_PREFIX + name)
    raise AttributeError(f"module 'bindsnet' has no attribute {name!r}")
