"""Connections: API mirror of bindsnet/network/topology.py for the connection types on the hot
path (`Connection`, `MulticompartmentConnection`, `Conv2dConnection`).  `compute()` launches the
matching propagation kernel of libsnnhip; inside Network.run the same kernels are driven from C++.
"""
import warnings
from typing import Optional, Sequence, Tuple, Union

import numpy as np
import torch
from torch.nn import Module, Parameter
from torch.nn.modules.utils import _pair

from .. import _lib, ops
from .nodes import Nodes


class AbstractConnection(_lib.TouchingModule, Module):
    """Reference: topology.py:17-156 (wmin/wmax/norm/update_rule plumbing)."""

    def __init__(self, source: Nodes, target: Nodes, nu=None, reduction: Optional[callable] = None,
                 weight_decay: float = 0.0, **kwargs) -> None:
        super().__init__()
        assert isinstance(source, Nodes), "Source is not a Nodes object"
        assert isinstance(target, Nodes), "Target is not a Nodes object"
        self.source, self.target = source, target
        self.weight_decay, self.reduction = weight_decay, reduction
        from ..learning import NoOp
        self.wmin = Parameter(torch.as_tensor(kwargs.get("wmin", -np.inf), dtype=torch.float32), requires_grad=False)
        self.wmax = Parameter(torch.as_tensor(kwargs.get("wmax", np.inf), dtype=torch.float32), requires_grad=False)
        self.norm = kwargs.get("norm", None)
        self.decay = kwargs.get("decay", None)
        if kwargs.get("Dales_rule", None) is not None:
            raise NotImplementedError("bindsnet_amd: Dales_rule is outside the accelerated path")
        self.Dales_rule = None
        rule = kwargs.get("update_rule", None) or NoOp
        self.update_rule = rule(connection=self, nu=nu, reduction=reduction, weight_decay=weight_decay, **kwargs)

    def update(self, **kwargs) -> None:
        """Reference: topology.py:112-139."""
        if kwargs.get("learning", True):
            if not self.w.is_cuda:                        # a connection on the host: plain PyTorch (network/host_path.py)
                from . import host_path
                host_path._update_dense(self, kwargs, None)
            else:
                self.update_rule.update(**kwargs)
        mask = kwargs.get("mask", None)
        if mask is not None:                       # topology.py:129-133
            self.w.masked_fill_(torch.as_tensor(mask, device=self.w.device).bool(), 0)

    def reset_state_variables(self) -> None:
        pass

    @staticmethod
    def cast_dtype_if_needed(w, w_dtype):
        if w.dtype != w_dtype:
            warnings.warn(f"Provided w has data type {w.dtype} but parameter w_dtype is {w_dtype}")
            return w.to(dtype=w_dtype)
        return w


class Connection(AbstractConnection):
    """Dense all-to-all synapses (reference: topology.py:265-399)."""

    def __init__(self, source: Nodes, target: Nodes, nu=None, reduction=None, weight_decay: float = 0.0,
                 w_dtype: torch.dtype = torch.float32, **kwargs) -> None:
        super().__init__(source, target, nu, reduction, weight_decay, **kwargs)
        if w_dtype != torch.float32:
            raise NotImplementedError("bindsnet_amd computes in float32 only")
        w = kwargs.get("w", None)
        unbounded = bool((self.wmin == -np.inf).any() or (self.wmax == np.inf).any())
        if w is None:  # consumes the global generator exactly like the reference (topology.py:309-315)
            if unbounded:
                w = torch.clamp(torch.rand(source.n, target.n), self.wmin, self.wmax)
            else:
                w = self.wmin + torch.rand(source.n, target.n) * (self.wmax - self.wmin)
            w = w.to(dtype=w_dtype)
        else:
            if bool((self.wmin != -np.inf).any() or (self.wmax != np.inf).any()):
                w = torch.clamp(torch.as_tensor(w), self.wmin, self.wmax)
            w = self.cast_dtype_if_needed(w, w_dtype)
        self.w = Parameter(w, requires_grad=False)
        b = kwargs.get("b", None)
        self.b = Parameter(b, requires_grad=False) if b is not None else None

    def compute(self, s: torch.Tensor) -> torch.Tensor:
        """s.view(B,-1) @ w (+ b) in canonical ascending-source order (topology.py:332-346)."""
        if not self.w.is_cuda:
            from . import host_path
            return host_path._propagate(self, s)
        B = s.size(0)
        out = torch.empty(B, self.target.n, device=self.w.device)
        ops.prop_dense(self.w.data, s.reshape(B, -1).contiguous(), out, bias=None if self.b is None else self.b.data)
        return out.view(B, *self.target.shape)

    def normalize(self) -> None:
        """Reference: topology.py:383-392 (abs column sums)."""
        if self.norm is not None:
            if not self.w.is_cuda:
                from . import host_path
                return host_path.normalize_connection(self)
            ops.normalize(self.w.data, float(self.norm), use_abs=True)


class LocalConnection(AbstractConnection):
    """Locally connected synapses (reference: topology.py:1304-1485): a dense [source.n, target.n] matrix that is zero
    outside each target neuron's receptive field (`mask`), propagated like `Connection` (+ bias), learned with the dense
    rules, masked again after every update, normalised by the SIGNED column sums (norm scaled by the kernel size).  Like
    the reference it draws its initial weights from numpy's global generator, and its `compute` output carries no batch
    dimension, i.e. it is meant for batch size 1."""

    def __init__(self, source: Nodes, target: Nodes, kernel_size: Union[int, Tuple[int, int]],
                 stride: Union[int, Tuple[int, int]], n_filters: int, nu=None, reduction=None, weight_decay: float = 0.0,
                 w_dtype: torch.dtype = torch.float32, **kwargs) -> None:
        super().__init__(source, target, nu, reduction, weight_decay, **kwargs)
        if w_dtype != torch.float32:
            raise NotImplementedError("bindsnet_amd computes in float32 only")
        self.kernel_size, self.stride, self.n_filters = _pair(kernel_size), _pair(stride), n_filters
        shape = kwargs.get("input_shape", None)
        if shape is None:
            shape = _pair(int(np.sqrt(source.n)))
        kh, kw = self.kernel_size
        if self.kernel_size == tuple(shape):
            conv_size = [1, 1]
        else:
            conv_size = (int((shape[0] - kh) / self.stride[0]) + 1, int((shape[1] - kw) / self.stride[1]) + 1)
        self.conv_size = conv_size
        conv_prod, kernel_prod = int(np.prod(conv_size)), int(np.prod(self.kernel_size))
        assert target.n == n_filters * conv_prod, (
            f"Total neurons in target layer must be {n_filters * conv_prod}. Got {target.n}.")
        # source index of tap (k1, k2) of receptive field (c1, c2) -- with the reference's own row stride for k1
        # (shape[0], topology.py:1420-1427)
        c1, c2 = torch.arange(conv_size[0]).view(1, 1, -1, 1), torch.arange(conv_size[1]).view(1, 1, 1, -1)
        k1, k2 = torch.arange(kh).view(-1, 1, 1, 1), torch.arange(kw).view(1, -1, 1, 1)
        locations = (c1 * self.stride[0] * shape[1] + c2 * self.stride[1] + k1 * shape[0] + k2).long()
        self.register_buffer("locations", locations.reshape(kernel_prod, conv_prod))
        w = kwargs.get("w", None)
        unbounded = bool((self.wmin == -np.inf).any() or (self.wmax == np.inf).any())
        if w is None:
            w = torch.zeros(source.n, target.n)
            for f in range(n_filters):                       # same visiting order as the reference: the draws come
                for c in range(conv_prod):                   # from numpy's global generator one at a time
                    for k in range(kernel_prod):
                        w[self.locations[k, c], f * conv_prod + c] = np.random.rand()
            w = torch.clamp(w, self.wmin, self.wmax) if unbounded else self.wmin + w * (self.wmax - self.wmin)
            w = w.to(dtype=w_dtype)
        else:
            if not unbounded or bool((self.wmin != -np.inf).any() or (self.wmax != np.inf).any()):
                w = torch.clamp(torch.as_tensor(w), self.wmin, self.wmax)
            w = self.cast_dtype_if_needed(w, w_dtype)
        self.w = Parameter(w, requires_grad=False)
        self.register_buffer("mask", self.w == 0)
        self.b = Parameter(kwargs.get("b", torch.zeros(target.n)), requires_grad=False)
        if self.norm is not None:
            self.norm *= kernel_prod

    def compute(self, s: torch.Tensor) -> torch.Tensor:
        B = s.size(0)
        if not self.w.is_cuda:
            from . import host_path
            out = host_path._propagate(self, s)
            return out.view(*self.target.shape) if B == 1 else out
        out = torch.empty(B, self.target.n, device=self.w.device)
        ops.prop_dense(self.w.data, s.reshape(B, -1).contiguous(), out, bias=self.b.data)
        return out.view(*self.target.shape) if B == 1 else out.view(B, *self.target.shape)

    def update(self, **kwargs) -> None:
        if kwargs.get("mask", None) is None:
            kwargs["mask"] = self.mask
        super().update(**kwargs)

    def normalize(self) -> None:
        """Signed column sums (topology.py:1475-1482)."""
        if self.norm is not None:
            if not self.w.is_cuda:
                from . import host_path
                return host_path.normalize_connection(self)
            ops.normalize(self.w.data.view(self.source.n, self.target.n), float(self.norm), use_abs=False)


class Conv2dConnection(AbstractConnection):
    """2-D convolutional synapses (reference: topology.py:686-844): propagation, PostPre, and MSTDP at batch 1."""

    def __init__(self, source: Nodes, target: Nodes, kernel_size: Union[int, Tuple[int, int]],
                 stride: Union[int, Tuple[int, int]] = 1, padding: Union[int, Tuple[int, int]] = 0,
                 dilation: Union[int, Tuple[int, int]] = 1, nu=None, reduction=None, weight_decay: float = 0.0,
                 w_dtype: torch.dtype = torch.float32, **kwargs) -> None:
        super().__init__(source, target, nu, reduction, weight_decay, **kwargs)
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        if self.dilation != (1, 1) or self.stride[0] != self.stride[1] or self.padding[0] != self.padding[1]:
            raise NotImplementedError("bindsnet_amd: conv2d supports dilation 1 and symmetric stride/padding only")
        rule = kwargs.get("update_rule", None)
        if rule is not None and rule.__name__ not in ("PostPre", "MSTDP", "NoOp"):
            raise NotImplementedError(f"bindsnet_amd: {rule.__name__} on Conv2dConnection is not supported (PostPre and MSTDP are)")
        self.in_channels, ih, iw = source.shape[0], source.shape[1], source.shape[2]
        if self.in_channels > 16:
            raise NotImplementedError("bindsnet_amd: Conv2dConnection with more than 16 input channels is not supported (the "
                                      "reference's oneDNN accumulation order is only characterised up to 16)")
        self.out_channels = target.shape[0]
        oh = int((ih - self.kernel_size[0] + 2 * self.padding[0]) / self.stride[0] + 1)
        ow = int((iw - self.kernel_size[1] + 2 * self.padding[1]) / self.stride[1] + 1)
        assert target.shape[1] == oh and target.shape[2] == ow, (
            "Target dimensionality must be (out_channels, ?,"
            "(input_height - filter_height + 2 * padding_height) / stride_height + 1,"
            "(input_width - filter_width + 2 * padding_width) / stride_width + 1")
        w = kwargs.get("w", None)
        inf = torch.tensor(np.inf)
        unbounded = bool((self.wmin == -inf).any() or (self.wmax == inf).any())
        if w is None:
            r = torch.rand(self.out_channels, self.in_channels, *self.kernel_size)
            w = torch.clamp(r, self.wmin, self.wmax) if unbounded else (self.wmax - self.wmin) * r + self.wmin
            w = w.to(dtype=w_dtype)
        else:
            if unbounded:
                w = torch.clamp(w, self.wmin, self.wmax)
            w = self.cast_dtype_if_needed(w, w_dtype)
        self.w = Parameter(w, requires_grad=False)
        self.b = Parameter(kwargs.get("b", torch.zeros(self.out_channels)), requires_grad=False)

    def compute(self, s: torch.Tensor) -> torch.Tensor:
        B = s.size(0)
        if not self.w.is_cuda:
            from . import host_path
            return host_path._propagate(self, s)
        out = torch.empty(B, *self.target.shape, device=self.w.device)
        ops.prop_conv2d(self.w.data, s.contiguous(), out, bias=self.b.data, stride=self.stride[0], pad=self.padding[0])
        return out

    def normalize(self) -> None:
        """Every [KH*KW] filter scaled to sum `norm` (topology.py:824-837)."""
        if self.norm is not None:
            if not self.w.is_cuda:
                from . import host_path
                return host_path.normalize_connection(self)
            if isinstance(self.norm, torch.Tensor):
                raise NotImplementedError("bindsnet_amd: tensor norms are not supported")
            ops.normalize_conv2d(self.w.data, float(self.norm))


class AbstractMulticompartmentConnection(_lib.TouchingModule, Module):
    """Reference: topology.py:159-262 (feature pipeline bookkeeping)."""

    def __init__(self, source: Nodes, target: Nodes, device, pipeline: list = None, **kwargs) -> None:
        super().__init__()
        assert isinstance(source, Nodes), "Source is not a Nodes object"
        assert isinstance(target, Nodes), "Target is not a Nodes object"
        self.source, self.target, self.device = source, target, device
        self.pipeline = [] if pipeline is None else pipeline
        self.feature_index = {}
        for feature in self.pipeline:
            self.feature_index[feature.name] = feature
            feature.prime_feature(connection=self, device=self.device, **kwargs)

    def append_pipeline(self, feature) -> None:
        self.pipeline.append(feature)
        feature.prime_feature(connection=self, device=self.device)
        self.feature_index[feature.name] = feature

    def remove_pipeline(self, feature) -> None:
        self.pipeline.remove(feature)
        del self.feature_index[feature.name]

    def _apply(self, fn, *a, **k):
        """nn.Module.to()/cuda() hook: also move feature values, which live in a plain python list
        and which the reference leaves behind on the CPU (SURVEY.md finding 8)."""
        out = super()._apply(fn, *a, **k)
        probe = fn(torch.empty(0))
        for f in self.pipeline:
            f.to(probe.device)
        self.device = probe.device
        return out


class MulticompartmentConnection(AbstractMulticompartmentConnection):
    """Feature-pipeline connection (reference: topology.py:402-537).  The accelerated path is a
    pipeline of exactly one `Weight`."""

    def __init__(self, source: Nodes, target: Nodes, device, pipeline: list = [], manual_update: bool = False,
                 traces: bool = False, **kwargs) -> None:
        super().__init__(source, target, device, pipeline, **kwargs)
        if traces:
            raise NotImplementedError("bindsnet_amd: connection activity traces are outside the accelerated path")
        self.traces, self.manual_update = traces, manual_update

    def _weight(self):
        from .topology_features import Weight
        if len(self.pipeline) != 1 or not isinstance(self.pipeline[0], Weight):
            raise NotImplementedError("bindsnet_amd accelerates MulticompartmentConnection with a single Weight "
                                      f"feature; got {[type(f).__name__ for f in self.pipeline]}")
        return self.pipeline[0]

    def compute(self, s: torch.Tensor) -> torch.Tensor:
        """out[b,j] = sum_i value[i,j]*s[b,i] in the reference's ATen sum order (topology.py:437-479)."""
        w = self._weight().value
        B = s.size(0)
        if not w.is_cuda:
            from . import host_path
            return host_path._propagate(self, s)
        out = torch.empty(B, self.target.n, device=w.device)
        ops.prop_cascade(w.data, s.reshape(B, -1).contiguous(), out)
        return out.view(B, *self.target.shape)

    def update(self, **kwargs) -> None:
        """Reference: topology.py:509-518 (note the default learning=False)."""
        if kwargs.get("learning", False) and not self.manual_update:
            if len(self.pipeline) == 1 and isinstance(self.pipeline[0].value, torch.Tensor) and not self.pipeline[0].value.is_cuda:
                from . import host_path                 # a connection on the host: plain PyTorch (network/host_path.py)
                return host_path._update_mcc(self, float(self.dt), kwargs)
            for f in self.pipeline:
                f.update(**kwargs)

    def normalize(self) -> None:
        if len(self.pipeline) == 1 and isinstance(self.pipeline[0].value, torch.Tensor) and not self.pipeline[0].value.is_cuda:
            from . import host_path
            return host_path.normalize_connection(self)
        for f in self.pipeline:
            f.normalize()

    def reset_state_variables(self) -> None:
        for f in self.pipeline:
            f.reset_state_variables()
