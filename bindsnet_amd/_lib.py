"""ctypes binding of libsnnhip.so (include/snnhip.h).

The library is the ONLY compute path of this package: if it is missing or a call fails, an
exception is raised -- there is no PyTorch/CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsnnhip.so")

SNN_OK, SNN_ERR_NOISE = 0, -4
ABI_VERSION = 1


class SnnError(RuntimeError):
    pass


class LifParams(C.Structure):
    _fields_ = [("decay", C.c_float), ("rest", C.c_float), ("reset", C.c_float), ("thresh", C.c_float),
                ("refrac", C.c_float), ("dt", C.c_float), ("has_lbound", C.c_int), ("lbound", C.c_float),
                ("traces", C.c_int), ("trace_decay", C.c_float), ("trace_scale", C.c_float),
                ("traces_additive", C.c_int)]


class DcParams(C.Structure):
    _fields_ = [("lif", LifParams), ("theta_decay", C.c_float), ("theta_plus", C.c_float),
                ("learning", C.c_int), ("one_spike", C.c_int)]


_lib = None

_vp, _i, _f, _l, _ll = C.c_void_p, C.c_int, C.c_float, C.c_long, C.c_longlong
_SIGS = {
    "snn_abi_version": ([], _i),
    "snn_error_string": ([_i], C.c_char_p),
    "snn_last_hip_error": ([], C.c_char_p),
    "snn_device_count": ([], _i),
    "snn_prop_cascade_f32": ([_vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "snn_prop_dense_f32": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "snn_prop_conv2d_f32": ([_vp, _vp, _vp, _vp] + [_i] * 10 + [_vp], _i),
    "snn_input_step": ([_vp, _vp, _l, _f, _f, _i, _vp, _vp], _i),
    "snn_lif_step": ([_vp, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(LifParams), _vp, _vp, _vp], _i),
    "snn_dc_step": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(DcParams), _vp, _ll, _vp, _vp, _vp, _vp, _vp], _i),
    "snn_stdp_postpre": ([_vp] * 5 + [_i, _i, _i, _f, _f, _i, _f, _f, _i, _f, _i, _f, _i, _vp], _i),
    "snn_mstdp_step": ([_vp] * 7 + [_i, _i, _i, _f, _vp, _f, _f, _f, _f, _f, _f, _i, _f, _i, _f, _vp], _i),
    "snn_normalize": ([_vp, _i, _i, _f, _i, _vp, _vp], _i),
}


def exported_symbols():
    """Every symbol include/snnhip.h declares (kept in sync by tests/test_abi.py)."""
    return sorted(_SIGS)


def lib():
    """Load libsnnhip.so (built by __graft_entry__.build() / csrc/Makefile). Fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SnnError(f"{LIB_PATH} is missing: build it with `make -C bindsnet_amd/csrc` "
                           "(python __graft_entry__.py build). There is no fallback path.")
        L = C.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(L, name)          # AttributeError if the .so is stale -> loud
            fn.argtypes, fn.restype = args, res
        if L.snn_abi_version() != ABI_VERSION:
            raise SnnError(f"libsnnhip ABI {L.snn_abi_version()} != binding {ABI_VERSION}")
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != SNN_OK:
        L = lib()
        msg = L.snn_error_string(rc).decode()
        if rc == -3:
            msg += ": " + L.snn_last_hip_error().decode()
        raise SnnError(f"libsnnhip {what}: {msg} (code {rc})")
