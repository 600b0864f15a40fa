"""`bindsnet.analysis`: plotting helpers of the examples (imported lazily: matplotlib is only needed when they are)."""
import importlib

__all__ = ["plotting"]


def __getattr__(name):
    if name == "plotting":
        return importlib.import_module(__name__ + ".plotting")
    raise AttributeError(name)
