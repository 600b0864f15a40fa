"""Thin tensor-level wrappers over the C ABI (one per entry point of include/snnhip.h).

torch is used for device memory and the current HIP stream only.  Every tensor must be a
contiguous CUDA(HIP) tensor of the documented dtype; spikes may be uint8 or bool (same byte
layout, like BindsNET's Input.s / LIFNodes.s).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import DcParams, LifParams, check, lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, dtype=None, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError("tensor required")
    if not t.is_cuda:
        raise _lib.SnnError("bindsnet_amd runs on an MI355X only: tensor is on " + str(t.device))
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    if dtype == "spike":
        if t.dtype not in (torch.uint8, torch.bool):
            raise TypeError(f"spike tensor must be uint8/bool, got {t.dtype}")
    elif dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


F32 = torch.float32


def prop_cascade(W, s, out, accumulate=False):
    """a5: out[b,j] (+)= sum_i W[i,j]*s[b,i] in ATen sum(dim=1) order."""
    B = s.shape[0]
    Nin, N = W.shape
    assert s.numel() == B * Nin and out.numel() == B * N
    check(lib().snn_prop_cascade_f32(_ptr(W, F32), _ptr(s, "spike"), _ptr(out, F32), B, Nin, N, int(accumulate),
                                     _stream()), "prop_cascade")
    return out


def prop_dense(W, s, out, bias=None, accumulate=False):
    """a6: out (+)= s @ W (+ b), ascending-i sequential f32."""
    B = s.shape[0]
    Nin, N = W.shape
    assert s.numel() == B * Nin and out.numel() == B * N
    check(lib().snn_prop_dense_f32(_ptr(W, F32), _ptr(bias, F32, True), _ptr(s, "spike"), _ptr(out, F32), B, Nin, N,
                                   int(accumulate), _stream()), "prop_dense")
    return out


def prop_dense_mfma(W, s, out, bias=None, accumulate=False):
    """a6 on the f32 matrix cores: bit-identical to prop_dense for 0/1 spikes (one k-ordered MFMA chain per tile)."""
    B = s.shape[0]
    Nin, N = W.shape
    assert s.numel() == B * Nin and out.numel() == B * N
    check(lib().snn_prop_dense_mfma_f32(_ptr(W, F32), _ptr(bias, F32, True), _ptr(s, "spike"), _ptr(out, F32), B, Nin, N,
                                        int(accumulate), _stream()), "prop_dense_mfma")
    return out


def prop_conv2d(W, s, out, bias=None, stride=1, pad=0, accumulate=False):
    """a7: F.conv2d on spikes; s [B,Cin,H,W], W [Cout,Cin,KH,KW], out [B,Cout,OH,OW]."""
    B, Cin, H, Wd = s.shape
    Cout, Cin2, KH, KW = W.shape
    assert Cin == Cin2
    check(lib().snn_prop_conv2d_f32(_ptr(W, F32), _ptr(bias, F32, True), _ptr(s, "spike"), _ptr(out, F32), B, Cin, H,
                                    Wd, Cout, KH, KW, stride, pad, int(accumulate), _stream()), "prop_conv2d")
    return out


def input_step(s, x=None, trace_decay=0.0, trace_scale=1.0, additive=False, raster=None):
    check(lib().snn_input_step(_ptr(s, "spike"), _ptr(x, F32, True), s.numel(), trace_decay, trace_scale,
                               int(additive), _ptr(raster, "spike", True), _stream()), "input_step")


def lif_step(v, refrac, s, x, I, p: LifParams, raster_s=None, raster_v=None, thresh_vec=None):
    """thresh_vec: optional f32 [N] per-neuron thresholds (replace p.thresh; nodes.py:425-498 with a tensor-valued `thresh`)."""
    B = v.shape[0]
    N = v.numel() // B
    if thresh_vec is not None and thresh_vec.numel() != N:
        raise ValueError(f"thresh_vec has {thresh_vec.numel()} entries, the layer {N} neurons")
    check(lib().snn_lif_step_vth(_ptr(v, F32), _ptr(refrac, F32), _ptr(s, "spike"), _ptr(x, F32, True), _ptr(I, F32), B, N,
                                 C.byref(p), _ptr(thresh_vec, F32, True), _ptr(raster_s, "spike", True), _ptr(raster_v, F32, True), _stream()),
          "lif_step")


def dc_step(v, refrac, s, x, theta, I, p: DcParams, noise_q, cursor, status, raster_s=None, raster_v=None):
    """cursor: int64[2] device tensor ([0] running count, [1] scratch); status: int32[1]."""
    B = v.shape[0]
    N = v.numel() // B
    qlen = 0 if noise_q is None else noise_q.numel()
    check(lib().snn_dc_step(_ptr(v, F32), _ptr(refrac, F32), _ptr(s, "spike"), _ptr(x, F32, True), _ptr(theta, F32),
                            _ptr(I, F32), B, N, C.byref(p), _ptr(noise_q, F32, True), qlen,
                            _ptr(cursor, torch.int64, True), _ptr(status, torch.int32, True),
                            _ptr(raster_s, "spike", True), _ptr(raster_v, F32, True), _stream()), "dc_step")


def dc_arbitrate(s, x, p: DcParams, noise_q, cursor, status, raster_s=None):
    """The second half of dc_step on its own (nodes.py:1097-1111): one_spike winners on the crossings `s` [B,N] (in place),
    trace, raster.  cursor[1] = offset of this step's first draw in noise_q (rng_fill_exponential leaves 0 and the draws
    in its qbuf)."""
    B = s.shape[0]
    N = s.numel() // B
    qlen = 0 if noise_q is None else noise_q.numel()
    check(lib().snn_dc_arbitrate(_ptr(s, "spike"), _ptr(x, F32, True), B, N, C.byref(p), _ptr(noise_q, F32, True), qlen,
                                 _ptr(cursor, torch.int64, True), _ptr(status, torch.int32, True),
                                 _ptr(raster_s, "spike", True), _stream()), "dc_arbitrate")


def stdp_postpre(W, s_src, x_src, s_tgt, x_tgt, nu0, nu1, use_dt, dt=1.0, decay=1.0, wmin=None, wmax=None,
                 assume_clamped=False):
    B = s_src.shape[0]
    Nin, N = W.shape
    check(lib().snn_stdp_postpre(_ptr(W, F32), _ptr(s_src, "spike"), _ptr(x_src, F32), _ptr(s_tgt, "spike"),
                                 _ptr(x_tgt, F32), B, Nin, N, nu0, nu1, int(use_dt), dt, decay,
                                 int(wmin is not None), 0.0 if wmin is None else wmin,
                                 int(wmax is not None), 0.0 if wmax is None else wmax,
                                 int(assume_clamped), _stream()), "stdp_postpre")


def mstdp_step(W, p_plus, p_minus, s_src_prev, s_tgt_prev, s_src, s_tgt, reward, nu0, a_plus, a_minus,
               decay_plus, decay_minus, wdecay=1.0, wmin=None, wmax=None, reward_vec=None):
    B = s_src.shape[0]
    Nin, N = W.shape
    check(lib().snn_mstdp_step(_ptr(W, F32), _ptr(p_plus, F32), _ptr(p_minus, F32), _ptr(s_src_prev, "spike"),
                               _ptr(s_tgt_prev, "spike"), _ptr(s_src, "spike"), _ptr(s_tgt, "spike"), B, Nin, N,
                               reward, _ptr(reward_vec, F32, True), nu0, a_plus, a_minus, decay_plus, decay_minus,
                               wdecay, int(wmin is not None), 0.0 if wmin is None else wmin,
                               int(wmax is not None), 0.0 if wmax is None else wmax, _stream()), "mstdp_step")


def conv2d_postpre(W, s_src, x_src, s_tgt, x_tgt, nu0, nu1, stride=1, pad=0, decay=1.0, wmin=None, wmax=None, ws=None):
    """f4: PostPre on Conv2dConnection weights [Cout,Cin,KH,KW]; s_src / x_src [B,Cin,H,W], s_tgt / x_tgt [B,Cout,OH,OW]."""
    B, Cin, H, Wd = s_src.shape
    Cout, _, KH, KW = W.shape
    if ws is None:
        ws = torch.empty(2 * B * W.numel(), dtype=F32, device=W.device)
    check(lib().snn_conv2d_postpre(_ptr(W, F32), _ptr(s_src, "spike"), _ptr(x_src, F32), _ptr(s_tgt, "spike"), _ptr(x_tgt, F32),
                                   B, Cin, H, Wd, Cout, KH, KW, stride, pad, nu0, nu1, decay, int(wmin is not None),
                                   0.0 if wmin is None else wmin, int(wmax is not None), 0.0 if wmax is None else wmax,
                                   _ptr(ws, F32), _stream()), "conv2d_postpre")


def conv2d_mstdp_step(W, elig, p_plus, p_minus, s_src, s_tgt, reward, nu0, a_plus, a_minus, decay_plus, decay_minus, stride=1, pad=0,
                      wdecay=1.0, wmin=None, wmax=None):
    """f4: MSTDP on Conv2dConnection weights [Cout,Cin,KH,KW] at batch 1; elig like W, p_plus [Cin,H,W], p_minus [Cout,OH*OW],
    s_src [Cin,H,W], s_tgt [Cout,OH,OW] (a leading batch dimension of 1 is fine)."""
    Cin, H, Wd = s_src.shape[-3:]
    Cout, _, KH, KW = W.shape
    check(lib().snn_conv2d_mstdp_step(_ptr(W, F32), _ptr(elig, F32), _ptr(p_plus, F32), _ptr(p_minus, F32), _ptr(s_src, "spike"),
                                      _ptr(s_tgt, "spike"), Cin, H, Wd, Cout, KH, KW, stride, pad, reward, nu0, a_plus, a_minus,
                                      decay_plus, decay_minus, wdecay, int(wmin is not None), 0.0 if wmin is None else wmin,
                                      int(wmax is not None), 0.0 if wmax is None else wmax, _stream()), "conv2d_mstdp_step")


def stdp_hebbian(W, s_src, x_src, s_tgt, x_tgt, nu0, nu1, weight_dependent=False, decay=1.0, wmin=None, wmax=None):
    """f3: Hebbian (weight_dependent=False) / WeightDependentPostPre (True) on a dense weight matrix."""
    B = s_src.shape[0]
    Nin, N = W.shape
    check(lib().snn_stdp_hebbian(_ptr(W, F32), _ptr(s_src, "spike"), _ptr(x_src, F32), _ptr(s_tgt, "spike"), _ptr(x_tgt, F32),
                                 B, Nin, N, nu0, nu1, int(weight_dependent), decay, int(wmin is not None),
                                 0.0 if wmin is None else wmin, int(wmax is not None), 0.0 if wmax is None else wmax, _stream()),
          "stdp_hebbian")


def mstdpet_step(W, e_trace, p_plus, p_minus, s_src_prev, s_tgt_prev, s_src, s_tgt, reward, nu0, dt, a_plus, a_minus,
                 decay_plus, decay_minus, decay_e, tc_e, wdecay=1.0, wmin=None, wmax=None):
    Nin, N = W.shape
    check(lib().snn_mstdpet_step(_ptr(W, F32), _ptr(e_trace, F32), _ptr(p_plus, F32), _ptr(p_minus, F32), _ptr(s_src_prev, "spike"),
                                 _ptr(s_tgt_prev, "spike"), _ptr(s_src, "spike"), _ptr(s_tgt, "spike"), Nin, N, reward, nu0, dt,
                                 a_plus, a_minus, decay_plus, decay_minus, decay_e, tc_e, wdecay, int(wmin is not None),
                                 0.0 if wmin is None else wmin, int(wmax is not None), 0.0 if wmax is None else wmax, _stream()),
          "mstdpet_step")


def normalize(W, norm, use_abs, ws=None):
    Nin, N = W.shape
    if ws is None:
        ws = torch.empty(N, dtype=F32, device=W.device)
    check(lib().snn_normalize(_ptr(W, F32), Nin, N, norm, int(use_abs), _ptr(ws, F32), _stream()), "normalize")


def normalize_conv2d(W, norm):
    """Conv2dConnection.normalize (topology.py:824-837): every [KH*KW] filter of W [Cout, Cin, KH, KW] scaled to sum `norm`."""
    Cout, Cin, KH, KW = W.shape
    check(lib().snn_normalize_conv2d(_ptr(W, F32), Cout * Cin, KH * KW, float(norm), _stream()), "normalize_conv2d")


def rng_fill_exponential(rng_state, crossings, qbuf, cursor):
    """Device generator: draws for the rows of `crossings` [B,N] that have a non-zero entry."""
    B = crossings.shape[0]
    N = crossings.numel() // B
    check(lib().snn_rng_fill_exponential(_ptr(rng_state, torch.int32), _ptr(crossings, "spike"), B, N, _ptr(qbuf, F32),
                                         _ptr(cursor, torch.int64), _stream()), "rng_fill_exponential")


def encode_bernoulli(datum_flat, steps, max_prob, device):
    """encodings.bernoulli on the device, drawing from the HOST generator's stream (left advanced by steps * n outputs,
    exactly as torch.bernoulli on the CPU would leave it).  datum_flat: 1-D float tensor on any device."""
    from .rng import DeviceGenerator
    x = datum_flat.to(device=device, dtype=F32).contiguous()
    n = x.numel()
    out = torch.empty(steps * n, dtype=torch.uint8, device=x.device)
    with DeviceGenerator(x.device, 1) as g:
        check(lib().snn_encode_bernoulli(_ptr(g.state, torch.int32), _ptr(x, F32), n, steps, float(max_prob), _ptr(out, torch.uint8),
                                         _stream()), "encode_bernoulli")
        g.finish()
    return out


def encode_poisson(datum_flat, steps, dt, seed, device):
    """encodings.poisson_device: Poisson spike trains from a Philox stream keyed by (seed, element)."""
    x = datum_flat.to(device=device, dtype=F32).contiguous()
    n = x.numel()
    out = torch.empty(steps * n, dtype=torch.uint8, device=x.device)
    check(lib().snn_encode_poisson(_ptr(x, F32), n, steps, float(dt), C.c_ulonglong(seed & ((1 << 64) - 1)), _ptr(out, torch.uint8),
                                   _stream()), "encode_poisson")
    return out
