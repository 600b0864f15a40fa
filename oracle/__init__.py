"""ctypes loader for the CPU oracle (oracle/snn_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under bindsnet_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsnn_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "snn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


class DcParams(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("Nin", C.c_int), ("N", C.c_int), ("T", C.c_int),
        ("dt", C.c_float),
        ("x_trace_decay", C.c_float), ("x_trace_scale", C.c_float),
        ("e_decay", C.c_float), ("e_rest", C.c_float), ("e_reset", C.c_float),
        ("e_thresh", C.c_float), ("e_refrac", C.c_float), ("e_theta_decay", C.c_float),
        ("e_theta_plus", C.c_float), ("e_trace_decay", C.c_float), ("e_trace_scale", C.c_float),
        ("e_one_spike", C.c_int),
        ("i_decay", C.c_float), ("i_rest", C.c_float), ("i_reset", C.c_float),
        ("i_thresh", C.c_float), ("i_refrac", C.c_float),
        ("nu0", C.c_float), ("nu1", C.c_float), ("wmin", C.c_float), ("wmax", C.c_float),
        ("norm", C.c_float),
        ("learning", C.c_int),
    ]


class TwoParams(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("Nin", C.c_int), ("N", C.c_int), ("T", C.c_int), ("rule", C.c_int),
        ("dt", C.c_float),
        ("x_trace_decay", C.c_float), ("x_trace_scale", C.c_float), ("x_traces", C.c_int),
        ("decay", C.c_float), ("rest", C.c_float), ("reset", C.c_float), ("thresh", C.c_float),
        ("refrac", C.c_float), ("y_traces", C.c_int), ("y_trace_decay", C.c_float),
        ("y_trace_scale", C.c_float),
        ("nu0", C.c_float), ("nu1", C.c_float), ("has_min", C.c_int), ("has_max", C.c_int),
        ("wmin", C.c_float), ("wmax", C.c_float), ("has_norm", C.c_int), ("norm", C.c_float),
        ("reward", C.c_float), ("a_plus", C.c_float), ("a_minus", C.c_float),
        ("decay_plus", C.c_float), ("decay_minus", C.c_float),
        ("learning", C.c_int), ("decay_e", C.c_float), ("tc_e", C.c_float), ("mcc", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_dc_step.restype = C.c_int
        _lib.orc_run_dc2015.restype = C.c_int
    return _lib


def _p(a, dtype=None):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"], "oracle wants C-contiguous numpy"
    if dtype is not None:
        assert a.dtype == dtype, (a.dtype, dtype)
    return a.ctypes.data_as(C.c_void_p)


f32, u8 = np.float32, np.uint8
cf, ci, cl = C.c_float, C.c_int, C.c_long


class reference_threads:
    """`with oracle.reference_threads(16): ...` -- prop_mcc / normalize (and run_dc2015, run_two_layer's MCC form) follow the
    reference AS IT RUNS WITH that many torch intra-op threads (ATen's thread-dependent order in the last N mod 32 < 8
    columns: snn_oracle.c tail_isolated).  Outside the block: 1 = the serial order everything is pinned to."""

    def __init__(self, threads: int):
        self.threads = int(threads)

    def __enter__(self):
        self._old = int(lib().orc_get_reference_threads())
        lib().orc_set_reference_threads(ci(self.threads))
        return self

    def __exit__(self, *exc):
        lib().orc_set_reference_threads(ci(self._old))
        return False


def prop_mcc(W, s, out=None, accumulate=False):
    B, Nin = s.shape
    N = W.shape[1]
    if out is None:
        out = np.zeros((B, N), f32)
    lib().orc_prop_mcc(_p(W, f32), _p(s, u8), _p(out, f32), ci(B), ci(Nin), ci(N), ci(int(accumulate)))
    return out


def prop_dense(W, s, bias=None, out=None, accumulate=False):
    B, Nin = s.shape
    N = W.shape[1]
    if out is None:
        out = np.zeros((B, N), f32)
    lib().orc_prop_dense(_p(W, f32), _p(bias, f32), _p(s, u8), _p(out, f32), ci(B), ci(Nin), ci(N),
                         ci(int(accumulate)))
    return out


def prop_conv2d(W, s, bias=None, stride=1, pad=0):
    B, Cin, H, Wd = s.shape
    Cout, _, KH, KW = W.shape
    OH, OW = (H + 2 * pad - KH) // stride + 1, (Wd + 2 * pad - KW) // stride + 1
    out = np.zeros((B, Cout, OH, OW), f32)
    lib().orc_prop_conv2d(_p(W, f32), _p(bias, f32), _p(s, u8), _p(out, f32), ci(B), ci(Cin), ci(H),
                          ci(Wd), ci(Cout), ci(KH), ci(KW), ci(stride), ci(pad), ci(0))
    return out


def input_step(s, x, trace_decay, trace_scale=1.0, additive=False):
    lib().orc_input_step(_p(s, u8), _p(x, f32), cl(s.size), ci(1), cf(trace_decay), cf(trace_scale),
                         ci(int(additive)))


def lif_step(v, refrac, s, x, I, *, decay, rest, reset, thresh, refrac0, dt=1.0, lbound=None,
             trace_decay=0.0, trace_scale=1.0, additive=False):
    B = v.shape[0]
    N = v.size // B
    lib().orc_lif_step(_p(v, f32), _p(refrac, f32), _p(s, u8), _p(x, f32), _p(I, f32), ci(B), ci(N),
                       cf(decay), cf(rest), cf(reset), cf(thresh), cf(refrac0), cf(dt),
                       ci(lbound is not None), cf(lbound or 0.0), ci(x is not None), cf(trace_decay),
                       cf(trace_scale), ci(int(additive)))


def dc_step(v, refrac, s, x, theta, I, Q, cursor, *, decay, rest, reset, thresh, refrac0, dt=1.0,
            theta_decay, theta_plus, learning=True, one_spike=True, lbound=None, trace_decay=0.0,
            trace_scale=1.0, additive=False):
    """cursor: np.int64 array of shape [1] (advanced in place). Returns rows that consumed noise."""
    B, N = v.shape
    cur = C.c_long(int(cursor[0]))
    r = lib().orc_dc_step(_p(v, f32), _p(refrac, f32), _p(s, u8), _p(x, f32), _p(theta, f32), _p(I, f32),
                          ci(B), ci(N), cf(decay), cf(rest), cf(reset), cf(thresh), cf(refrac0), cf(dt),
                          cf(theta_decay), cf(theta_plus), ci(int(learning)), ci(int(one_spike)),
                          ci(lbound is not None), cf(lbound or 0.0), ci(x is not None), cf(trace_decay),
                          cf(trace_scale), ci(int(additive)), _p(Q, f32), cl(Q.size), C.byref(cur))
    cursor[0] = cur.value
    return r


def postpre(W, s_src, x_src, s_tgt, x_tgt, *, nu0, nu1, use_dt, dt=1.0, decay=1.0, wmin=None, wmax=None):
    B, Nin = s_src.shape
    N = s_tgt.shape[1]
    lib().orc_postpre(_p(W, f32), _p(s_src, u8), _p(x_src, f32), _p(s_tgt, u8), _p(x_tgt, f32),
                      ci(B), ci(Nin), ci(N), cf(nu0), cf(nu1), ci(int(use_dt)), cf(dt), cf(decay),
                      ci(wmin is not None), cf(wmin or 0.0), ci(wmax is not None), cf(wmax or 0.0))


def mstdp(W, elig, p_plus, p_minus, s_src, s_tgt, *, reward, nu0, a_plus=1.0, a_minus=-1.0,
          decay_plus, decay_minus, wdecay=1.0, wmin=None, wmax=None):
    B, Nin = s_src.shape
    N = s_tgt.shape[1]
    lib().orc_mstdp(_p(W, f32), _p(elig, f32), _p(p_plus, f32), _p(p_minus, f32), _p(s_src, u8),
                    _p(s_tgt, u8), ci(B), ci(Nin), ci(N), cf(reward), None, cf(nu0), cf(a_plus),
                    cf(a_minus), cf(decay_plus), cf(decay_minus), cf(wdecay),
                    ci(wmin is not None), cf(wmin or 0.0), ci(wmax is not None), cf(wmax or 0.0))


def hebbian_wdpp(W, s_src, x_src, s_tgt, x_tgt, *, nu0, nu1, weight_dependent, decay=1.0, wmin=None, wmax=None):
    B, Nin = s_src.shape
    N = s_tgt.shape[1]
    lib().orc_hebbian_wdpp(_p(W, f32), _p(s_src, u8), _p(x_src, f32), _p(s_tgt, u8), _p(x_tgt, f32), ci(B), ci(Nin), ci(N),
                           cf(nu0), cf(nu1), ci(int(weight_dependent)), cf(decay), ci(wmin is not None), cf(wmin or 0.0),
                           ci(wmax is not None), cf(wmax or 0.0))


def mstdpet(W, elig, e_trace, p_plus, p_minus, s_src, s_tgt, *, reward, nu0, dt=1.0, a_plus=1.0, a_minus=-1.0, decay_plus,
            decay_minus, decay_e, tc_e, wdecay=1.0, wmin=None, wmax=None):
    Nin, N = W.shape
    lib().orc_mstdpet(_p(W, f32), _p(elig, f32), _p(e_trace, f32), _p(p_plus, f32), _p(p_minus, f32), _p(s_src, u8), _p(s_tgt, u8),
                      ci(Nin), ci(N), cf(reward), cf(nu0), cf(dt), cf(a_plus), cf(a_minus), cf(decay_plus), cf(decay_minus),
                      cf(decay_e), cf(tc_e), cf(wdecay), ci(wmin is not None), cf(wmin or 0.0), ci(wmax is not None), cf(wmax or 0.0))


def conv2d_postpre(W, s_src, x_src, s_tgt, x_tgt, *, stride=1, pad=0, nu0, nu1, decay=1.0, wmin=None, wmax=None):
    B, Cin, H, Wd = s_src.shape
    Cout, _, KH, KW = W.shape
    lib().orc_conv2d_postpre(_p(W, f32), _p(s_src, u8), _p(x_src, f32), _p(s_tgt, u8), _p(x_tgt, f32), ci(B), ci(Cin), ci(H), ci(Wd),
                             ci(Cout), ci(KH), ci(KW), ci(stride), ci(pad), cf(nu0), cf(nu1), cf(decay), ci(wmin is not None),
                             cf(wmin or 0.0), ci(wmax is not None), cf(wmax or 0.0))


def conv2d_mstdp(W, E, P, Q, s_src, s_tgt, *, stride=1, pad=0, reward, nu0, a_plus=1.0, a_minus=-1.0, decay_plus, decay_minus,
                 wdecay=1.0, wmin=None, wmax=None):
    """W [Cout,Cin,KH,KW]; E like W; P [Cin,H,W]; Q [Cout,OH,OW]; s_src [Cin,H,W] u8; s_tgt [Cout,OH,OW] u8 (batch 1)."""
    Cin, H, Wd = s_src.shape
    Cout, _, KH, KW = W.shape
    lib().orc_conv2d_mstdp(_p(W, f32), _p(E, f32), _p(P, f32), _p(Q, f32), _p(s_src, u8), _p(s_tgt, u8), ci(Cin), ci(H), ci(Wd),
                           ci(Cout), ci(KH), ci(KW), ci(stride), ci(pad), cf(reward), cf(nu0), cf(a_plus), cf(a_minus),
                           cf(decay_plus), cf(decay_minus), cf(wdecay), ci(wmin is not None), cf(wmin or 0.0),
                           ci(wmax is not None), cf(wmax or 0.0))


def normalize(W, norm, use_abs):
    Nin, N = W.shape
    lib().orc_normalize(_p(W, f32), ci(Nin), ci(N), cf(norm), ci(int(use_abs)))


def normalize_conv2d(W, norm):
    """Conv2dConnection.normalize (topology.py:824-837) in place on W [Cout, Cin, KH, KW]."""
    Cout, Cin, KH, KW = W.shape
    lib().orc_normalize_conv2d(_p(W, f32), ci(Cout * Cin), ci(KH * KW), cf(norm))


def inner_sum(x):
    """ATen's vectorised inner sum of one contiguous f32 row."""
    f = lib().orc_inner_sum
    f.restype = C.c_float
    return float(f(_p(np.ascontiguousarray(x, f32), f32), cl(len(x))))


def run_dc2015(P: DcParams, st: dict, inputs, Q, cursor, rasters=True):
    """st: dict of numpy state arrays (W_xe, W_ei, W_ie, sX, xX, vE, rE, sE, xE, theta, vI, rI, sI),
    all updated in place. inputs u8 [T,B,Nin]. Returns (rasterE, rasterI)."""
    T, B, Nin = inputs.shape
    N = P.N
    rE = np.zeros((T, B, N), u8) if rasters else None
    rI = np.zeros((T, B, N), u8) if rasters else None
    cur = C.c_long(int(cursor[0]))
    rc = lib().orc_run_dc2015(C.byref(P), _p(st["W_xe"], f32), _p(st["W_ei"], f32), _p(st["W_ie"], f32),
                              _p(inputs, u8), _p(st["sX"], u8), _p(st["xX"], f32),
                              _p(st["vE"], f32), _p(st["rE"], f32), _p(st["sE"], u8), _p(st["xE"], f32),
                              _p(st["theta"], f32), _p(st["vI"], f32), _p(st["rI"], f32), _p(st["sI"], u8),
                              _p(Q, f32), cl(Q.size), C.byref(cur), _p(rE, u8), _p(rI, u8))
    if rc != 0:
        raise RuntimeError("oracle: noise buffer Q too short")
    cursor[0] = cur.value
    return rE, rI


def run_two_layer(P: TwoParams, st: dict, inputs, I_forced=None, bias=None):
    T, B, Nin = inputs.shape
    ras = np.zeros((T, B, P.N), u8)
    lib().orc_run_two_layer(C.byref(P), _p(st["W"], f32), _p(bias, f32), _p(inputs, u8), _p(st["sX"], u8),
                            _p(st["xX"], f32), _p(st["vY"], f32), _p(st["rY"], f32), _p(st["sY"], u8),
                            _p(st.get("xY"), f32), _p(st.get("elig"), f32), _p(st.get("p_plus"), f32),
                            _p(st.get("p_minus"), f32), _p(I_forced, f32), _p(ras, u8), _p(st.get("e_trace"), f32))
    return ras


def encode_poisson(datum: np.ndarray, steps: int, dt: float = 1.0, seed: int = 0) -> np.ndarray:
    """orc_encode_poisson: libsnnhip's snn_encode_poisson restated (specified stream: Philox-4x32-10 + fixed-series sampler) -> u8 [steps, n]."""
    x = np.ascontiguousarray(datum, f32).reshape(-1)
    out = np.zeros((steps, x.size), u8)
    fn = lib().orc_encode_poisson
    fn.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_uint64, C.c_void_p]
    fn.restype = None
    fn(_p(x, f32), x.size, int(steps), float(dt), int(seed) & ((1 << 64) - 1), _p(out, u8))
    return out


def mt_exponential(mt_state: np.ndarray, pos: int, n: int):
    """mt_state: uint32[624] (modified in place); pos in [0,624]. Returns (float32[n], new_pos)."""
    out = np.empty(n, f32)
    p = C.c_int(pos)
    lib().orc_mt_exponential(_p(mt_state, np.uint32), C.byref(p), _p(out, f32), cl(n))
    return out, p.value


# ----------------------------------------------------------------------------------------------------------------
# The eth_mnist.py D&C case (examples/mnist/eth_mnist.py:91-100 -> bindsnet/models/models.py:156-236) as oracle
# arguments; bench.py's cpu_baseline leg and the tests build the checker's inputs through these.
def eth_mnist_dc_params(N, B, T, Nin=784, learning=True) -> DcParams:
    """Decay constants as the reference computes them: torch.exp(-dt / tc) in f32 (nodes.py:122-131,540-548,1122-1133)."""
    import torch

    def dec(tc):
        return float(torch.exp(-torch.tensor(1.0) / torch.tensor(float(tc))))
    P = DcParams()
    P.B, P.Nin, P.N, P.T, P.dt = B, Nin, N, T, 1.0
    P.x_trace_decay, P.x_trace_scale = dec(20.0), 1.0
    P.e_decay, P.e_theta_decay, P.e_trace_decay, P.e_trace_scale, P.e_one_spike = dec(100.0), dec(1e7), dec(20.0), 1.0, 1
    P.i_decay = dec(10.0)
    P.e_rest, P.e_reset, P.e_thresh, P.e_refrac, P.e_theta_plus = -65.0, -60.0, -52.0, 5.0, 0.05
    P.i_rest, P.i_reset, P.i_thresh, P.i_refrac = -60.0, -45.0, -40.0, 2.0
    P.nu0, P.nu1, P.wmin, P.wmax, P.norm = 1e-4, 1e-2, 0.0, 1.0, 78.4
    P.learning = int(learning)
    return P


def eth_mnist_dc_state(N, B, W_xe, Nin=784, exc=22.5, inh=120.0) -> dict:
    return dict(
        W_xe=np.ascontiguousarray(W_xe, f32), W_ei=(exc * np.eye(N)).astype(f32),
        W_ie=(-inh * (np.ones((N, N)) - np.eye(N))).astype(f32),
        sX=np.zeros((B, Nin), u8), xX=np.zeros((B, Nin), f32),
        vE=np.full((B, N), -65.0, f32), rE=np.zeros((B, N), f32), sE=np.zeros((B, N), u8),
        xE=np.zeros((B, N), f32), theta=np.zeros(N, f32),
        vI=np.full((B, N), -60.0, f32), rI=np.zeros((B, N), f32), sI=np.zeros((B, N), u8))


def exp_noise(seed, n):
    """The Exp(1) stream torch.multinomial consumes after torch.manual_seed(seed)."""
    import torch
    torch.manual_seed(int(seed))
    return torch.empty(int(n)).exponential_(1).numpy()
