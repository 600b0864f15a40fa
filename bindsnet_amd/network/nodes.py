"""Neuron groups: API mirror of bindsnet/network/nodes.py for the layer types on the hot path
(`Nodes`, `Input`, `LIFNodes`, `DiehlAndCookNodes`); the arithmetic lives in libsnnhip
(snn_input_step / snn_lif_step / snn_dc_step).

State is held in the same attributes as the reference (`s`, `x`, `v`, `refrac_count`, `theta`,
`decay`, `trace_decay`, ...), so monitors, pickling and user code that pokes at them keep working.
"""
from functools import reduce
from operator import mul
from typing import Iterable, Optional, Union

import torch

from .. import _lib, ops

Scalar = Union[float, torch.Tensor]
_SCALARS = None          # collector of the parameter tensors read while run descriptors are being built


def _f(t) -> float:
    """Python float of a 0-dim parameter.  Per-neuron (tensor-valued) parameters are outside
    the accelerated path (SURVEY.md 8(b) fallback rule) and are rejected loudly."""
    if isinstance(t, torch.Tensor):
        # .item() on a device tensor is a blocking copy (~15 us each, ~20 per run()): remember the value on the
        # tensor object itself, keyed by its in-place version counter
        if _SCALARS is not None:
            _SCALARS.append((t, t._version))       # (Network._build_descriptors: what the descriptors were filled from)
        hit = getattr(t, "_snn_scalar", None)
        if hit is not None and hit[0] == t._version:
            return hit[1]
        if t.numel() != 1:
            raise NotImplementedError("bindsnet_amd: tensor-valued per-neuron parameters are not supported")
        val = float(t.reshape(()).item())
        try:
            t._snn_scalar = (t._version, val)
        except AttributeError:
            pass
        return val
    return float(t)


def _buf(value, dtype=None) -> torch.Tensor:
    """torch.tensor(value, dtype=...) as the reference writes it (nodes.py:459-472) -- for a tensor-valued parameter that is a detached copy,
    which torch.tensor() makes too but with a UserWarning into every user's log."""
    if isinstance(value, torch.Tensor):
        out = value.detach().clone()
        return out if dtype is None else out.to(dtype)
    return torch.tensor(value) if dtype is None else torch.tensor(value, dtype=dtype)


class Nodes(_lib.TouchingModule, torch.nn.Module):
    """Base class (reference: nodes.py:9-162): spikes `s`, optional trace `x`."""

    def __init__(self, n: Optional[int] = None, shape: Optional[Iterable[int]] = None, traces: bool = False,
                 traces_additive: bool = False, tc_trace: Scalar = 20.0, trace_scale: Scalar = 1.0,
                 sum_input: bool = False, learning: bool = True, **kwargs) -> None:
        super().__init__()
        if n is None and shape is None:
            raise AssertionError("Must provide either no. of neurons or shape of layer")
        self.n = reduce(mul, shape) if n is None else n
        self.shape = [self.n] if shape is None else shape
        assert self.n == reduce(mul, self.shape), "No. of neurons and shape do not match"
        if sum_input:
            raise NotImplementedError("bindsnet_amd: sum_input=True is outside the accelerated path")
        self.traces, self.traces_additive, self.sum_input = traces, traces_additive, sum_input
        self.register_buffer("s", torch.ByteTensor())
        if traces:
            self.register_buffer("x", torch.Tensor())
            self.register_buffer("tc_trace", _buf(tc_trace))
            self.register_buffer("trace_scale", _buf(trace_scale))
            self.register_buffer("trace_decay", torch.empty_like(self.tc_trace))
        self.dt = None
        self.batch_size = None
        self.learning = learning

    # -- lifecycle hooks called by Network.add_layer (network.py:130-132) ----------------------
    def compute_decays(self, dt) -> None:
        self.dt = torch.tensor(dt)
        if self.traces:  # same torch op as the reference so the constant is the same float
            self.trace_decay = torch.exp(-self.dt / self.tc_trace.cpu()).to(self.tc_trace.device)

    def set_batch_size(self, batch_size) -> None:
        self.batch_size = batch_size
        dev = self.s.device
        self.s = torch.zeros(batch_size, *self.shape, device=dev, dtype=torch.bool)
        if self.traces:
            self.x = torch.zeros(batch_size, *self.shape, device=dev)

    def reset_state_variables(self) -> None:
        self.s.zero_()
        if self.traces:
            self.x.zero_()

    def train(self, mode: bool = True) -> "Nodes":
        self.learning = mode
        return super().train(mode)

    # -- descriptor pieces for the run driver ---------------------------------------------------
    def _trace_fields(self, p: _lib.LifParams) -> None:
        p.traces = int(self.traces)
        if self.traces:
            p.trace_decay, p.trace_scale = _f(self.trace_decay), _f(self.trace_scale)
            p.traces_additive = int(self.traces_additive)

    def forward(self, x: torch.Tensor) -> None:
        raise NotImplementedError


class AbstractInput:
    pass


class Input(Nodes, AbstractInput):
    """User-driven spikes (reference: nodes.py:172-228): `s` aliases the input, trace optional."""

    def __init__(self, n=None, shape=None, traces=False, traces_additive=False, tc_trace=20.0, trace_scale=1.0,
                 sum_input=False, **kwargs) -> None:
        super().__init__(n=n, shape=shape, traces=traces, traces_additive=traces_additive, tc_trace=tc_trace,
                         trace_scale=trace_scale, sum_input=sum_input)

    def forward(self, x: torch.Tensor) -> None:
        if not x.is_cuda:                                 # a layer on the host: plain PyTorch (network/host_path.py)
            from . import host_path
            return host_path._step_input(self, x)
        self.s = x
        if self.traces:
            ops.input_step(x.contiguous(), self.x, _f(self.trace_decay), _f(self.trace_scale), self.traces_additive)


class LIFNodes(Nodes):
    """Leaky integrate-and-fire layer (reference: nodes.py:418-559)."""

    def __init__(self, n=None, shape=None, traces=False, traces_additive=False, tc_trace=20.0, trace_scale=1.0,
                 sum_input=False, thresh: Scalar = -52.0, rest: Scalar = -65.0, reset: Scalar = -65.0,
                 refrac: Union[int, torch.Tensor] = 5, tc_decay: Scalar = 100.0, lbound: float = None,
                 **kwargs) -> None:
        super().__init__(n=n, shape=shape, traces=traces, traces_additive=traces_additive, tc_trace=tc_trace,
                         trace_scale=trace_scale, sum_input=sum_input)
        self.register_buffer("rest", _buf(rest, torch.float))
        self.register_buffer("reset", _buf(reset, torch.float))
        self.register_buffer("thresh", _buf(thresh, torch.float))
        self.register_buffer("refrac", _buf(refrac))
        self.register_buffer("tc_decay", _buf(tc_decay, torch.float))
        self.register_buffer("decay", torch.zeros(*self.shape))
        self.register_buffer("v", torch.FloatTensor())
        self.register_buffer("refrac_count", torch.FloatTensor())
        self.lbound = None if lbound is None else torch.tensor(lbound, dtype=torch.float)

    def compute_decays(self, dt) -> None:
        super().compute_decays(dt=dt)
        self.decay = torch.exp(-self.dt / self.tc_decay.cpu()).to(self.tc_decay.device)

    def set_batch_size(self, batch_size) -> None:
        super().set_batch_size(batch_size=batch_size)
        dev = self.v.device
        self.v = self.rest.to(dev) * torch.ones(batch_size, *self.shape, device=dev)
        self.refrac_count = torch.zeros_like(self.v)

    def reset_state_variables(self) -> None:
        super().reset_state_variables()
        self.v.fill_(_f(self.rest))
        self.refrac_count.zero_()

    def _thresh_vec(self) -> Optional[torch.Tensor]:
        """Per-neuron thresholds (nodes.py:425-498 take `thresh` as a tensor; examples/mnist/reservoir.py passes one): the [n] f32
        device tensor the LIF kernel indexes by neuron, or None for the usual scalar."""
        t = self.thresh
        if not isinstance(t, torch.Tensor) or t.numel() == 1:
            return None
        if t.numel() != self.n:
            raise ValueError(f"LIFNodes.thresh has {t.numel()} entries, the layer {self.n} neurons")
        if t.device != self.v.device or t.dtype != torch.float32 or not t.is_contiguous():
            # (a buffer: .to() moves it with the layer; anything else is brought over once and kept -- an in-place change of the
            #  original is then not seen, which the version check below guards)
            cached = self.__dict__.get("_thresh_dev")
            if cached is None or cached[0] is not t or cached[1] != t._version or cached[2].device != self.v.device:
                cached = self.__dict__["_thresh_dev"] = (t, t._version, t.detach().to(self.v.device, torch.float32).contiguous())
            return cached[2]
        return t

    def _lif_params(self) -> _lib.LifParams:
        p = _lib.LifParams()
        p.decay, p.rest, p.reset = _f(self.decay), _f(self.rest), _f(self.reset)
        p.thresh = 0.0 if self._thresh_vec() is not None else _f(self.thresh)
        p.refrac, p.dt = _f(self.refrac), _f(self.dt)
        p.has_lbound = int(self.lbound is not None)
        p.lbound = _f(self.lbound) if self.lbound is not None else 0.0
        self._trace_fields(p)
        return p

    def forward(self, x: torch.Tensor) -> None:
        """One step (nodes.py:500-529); `x` is masked in place where refractory, as in the reference."""
        if not self.v.is_cuda:                            # a layer on the host: plain PyTorch (network/host_path.py)
            from . import host_path
            return host_path._step_lif(self, x)
        if self.s.dtype != torch.bool or self.s.shape != self.v.shape:
            self.s = torch.zeros_like(self.v, dtype=torch.bool)
        ops.lif_step(self.v, self.refrac_count, self.s, self.x if self.traces else None, x, self._lif_params(),
                     thresh_vec=self._thresh_vec())


class DiehlAndCookNodes(Nodes):
    """LIF with adaptive threshold and one-spike arbitration (reference: nodes.py:981-1144)."""

    def __init__(self, n=None, shape=None, traces=False, traces_additive=False, tc_trace=20.0, trace_scale=1.0,
                 sum_input=False, thresh: Scalar = -52.0, rest: Scalar = -65.0, reset: Scalar = -65.0,
                 refrac: Union[int, torch.Tensor] = 5, tc_decay: Scalar = 100.0, theta_plus: Scalar = 0.05,
                 tc_theta_decay: Scalar = 1e7, lbound: float = None, one_spike: bool = True, **kwargs) -> None:
        super().__init__(n=n, shape=shape, traces=traces, traces_additive=traces_additive, tc_trace=tc_trace,
                         trace_scale=trace_scale, sum_input=sum_input)
        self.register_buffer("rest", _buf(rest))
        self.register_buffer("reset", _buf(reset))
        self.register_buffer("thresh", _buf(thresh))
        self.register_buffer("refrac", _buf(refrac))
        self.register_buffer("tc_decay", _buf(tc_decay))
        self.register_buffer("decay", torch.empty_like(self.tc_decay))
        self.register_buffer("theta_plus", _buf(theta_plus))
        self.register_buffer("tc_theta_decay", _buf(tc_theta_decay))
        self.register_buffer("theta_decay", torch.empty_like(self.tc_theta_decay))
        self.register_buffer("v", torch.FloatTensor())
        self.register_buffer("theta", torch.zeros(*self.shape))
        self.register_buffer("refrac_count", torch.FloatTensor())
        self.lbound = lbound
        self.one_spike = one_spike

    def compute_decays(self, dt) -> None:
        super().compute_decays(dt=dt)
        dev = self.tc_decay.device
        self.decay = torch.exp(-self.dt / self.tc_decay.cpu()).to(dev)
        self.theta_decay = torch.exp(-self.dt / self.tc_theta_decay.cpu()).to(dev)

    def set_batch_size(self, batch_size) -> None:
        super().set_batch_size(batch_size=batch_size)
        dev = self.v.device
        self.v = self.rest.to(dev) * torch.ones(batch_size, *self.shape, device=dev)
        self.refrac_count = torch.zeros_like(self.v)

    def reset_state_variables(self) -> None:  # theta is NOT reset (nodes.py:1113-1120)
        super().reset_state_variables()
        self.v.fill_(_f(self.rest))
        self.refrac_count.zero_()

    def _dc_params(self) -> _lib.DcParams:
        p = _lib.DcParams()
        l = p.lif
        l.decay, l.rest, l.reset, l.thresh = _f(self.decay), _f(self.rest), _f(self.reset), _f(self.thresh)
        l.refrac, l.dt = _f(self.refrac), _f(self.dt)
        l.has_lbound = int(self.lbound is not None)
        l.lbound = (_f(self.lbound) if isinstance(self.lbound, torch.Tensor) else float(self.lbound)) if self.lbound is not None else 0.0
        self._trace_fields(l)
        p.theta_decay, p.theta_plus = _f(self.theta_decay), _f(self.theta_plus)
        p.learning, p.one_spike = int(self.learning), int(self.one_spike)
        return p

    def forward(self, x: torch.Tensor) -> None:
        """One step (nodes.py:1069-1111).  The winner draw consumes the global CPU generator
        exactly like torch.multinomial does in the reference (see bindsnet_amd/rng.py)."""
        if not self.v.is_cuda:                            # a layer on the host: plain PyTorch (network/host_path.py)
            from . import host_path
            return host_path._step_dc(self, x)
        from ..rng import NoiseStream
        if self.s.dtype != torch.bool or self.s.shape != self.v.shape:
            self.s = torch.zeros_like(self.v, dtype=torch.bool)
        B = self.v.shape[0]
        with NoiseStream(self.v.device, max_draws=B * self.n if self.one_spike else 0) as ns:
            ops.dc_step(self.v, self.refrac_count, self.s, self.x if self.traces else None, self.theta, x,
                        self._dc_params(), ns.q, ns.cursor, ns.status)
