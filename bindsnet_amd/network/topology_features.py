"""Connection features: API mirror of bindsnet/network/topology_features.py for the features on
the hot path (`AbstractFeature`, `Weight`).  A `Weight` only holds the [Nin, N] tensor and its
learning rule; the multiply-and-reduce it stands for runs in snn_prop_cascade_f32."""
import warnings
from typing import Optional, Sequence, Union

import torch
from torch.nn import Parameter

from .. import _lib


class AbstractFeature(_lib.Touching):
    """Reference: topology_features.py:15-362 (constructor contract, priming, normalisation)."""

    def __init__(self, name: str, value=None, value_dtype: torch.dtype = torch.float32, range=None,
                 clamp_frequency: Optional[int] = 1, norm=None, learning_rule=None, nu=None, reduction=None,
                 enforce_polarity: Optional[bool] = False, decay: float = 0.0, parent_feature=None,
                 sparse: Optional[bool] = False, batch_size: int = 1, **kwargs) -> None:
        from ..learning.MCC_learning import MSTDP, MSTDPET, NoOp, PostPre
        assert isinstance(name, str), f"Feature {name}'s name should be of type str"
        assert value is None or isinstance(value, (torch.Tensor, float, int)), (
            f"Feature {name} should be of type float, int, or torch.Tensor, not {type(value)}")
        assert norm is None or isinstance(norm, (torch.Tensor, float, int)), (
            f"Feature {name}'s norm should be of type float, int, or torch.Tensor, not {type(norm)}")
        assert learning_rule is None or learning_rule in (NoOp, PostPre, MSTDP, MSTDPET), (
            f"Feature {name}'s learning_rule should be of type bindsnet.LearningRule not {type(learning_rule)}")
        assert nu is None or isinstance(nu, (list, tuple)), (
            f"Feature {name}'s nu should be of type list or tuple, not {type(nu)}")
        assert reduction is None or callable(reduction), f"Feature {name}'s reduction should be callable"
        assert decay is None or isinstance(decay, float), f"Feature {name}'s decay should be of type float"
        if sparse or parent_feature is not None or enforce_polarity:
            raise NotImplementedError("bindsnet_amd: sparse / linked / polarity-enforcing features are outside "
                                      "the accelerated path")
        self.name, self.value = name, value
        self.range = [-1.0, 1.0] if range is None else range
        self.clamp_frequency, self.norm, self.learning_rule = clamp_frequency, norm, learning_rule
        self.nu, self.reduction, self.decay = nu, reduction, decay
        self.parent_feature, self.sparse, self.batch_size, self.kwargs = parent_feature, sparse, batch_size, kwargs
        self.is_primed = False
        r = self.range
        assert isinstance(r, (list, tuple)) and len(r) == 2, f"Invalid range for feature {name}"
        lo_hi_ok = (r[0] < r[1]).all() if isinstance(r[0], torch.Tensor) or isinstance(r[1], torch.Tensor) else r[0] < r[1]
        assert lo_hi_ok, f"Invalid range for feature {name}: the min value is larger than the max value"
        if value is None:
            return
        if isinstance(value, torch.Tensor):
            assert (value >= r[0]).all() and (value <= r[1]).all(), (
                f"Feature out of range for {name}: Features values not in [{r[0]}, {r[1]}]")
            if value.dtype != value_dtype:
                warnings.warn(f"Provided value has data type {value.dtype} but parameter w_dtype is {value_dtype}")
                self.value = value.to(dtype=value_dtype)
        else:
            assert r[0] <= value <= r[1], f"Feature out of range for {name}"

    def prime_feature(self, connection, device, **kwargs) -> None:
        """Reference: topology_features.py:173-240 -- shape checks, Parameter-isation, rule construction."""
        from ..learning.MCC_learning import NoOp
        if self.is_primed:
            return
        self.is_primed = True
        if isinstance(self.value, torch.Tensor):
            assert tuple(self.value.shape) == (connection.source.n, connection.target.n)
        if self.norm is not None and isinstance(self.norm, torch.Tensor):
            assert self.norm.shape[0] == connection.target.n
        if self.value is None:
            self.value = self.initialize_value()
        if isinstance(self.value, (int, float)):
            self.value = torch.Tensor([self.value])
        self.value = Parameter(self.value, requires_grad=False).to(device)
        rule = NoOp if self.learning_rule is None else self.learning_rule
        self.learning_rule = rule(connection=connection, feature_value=self.value, range=self.range, nu=self.nu,
                                  reduction=self.reduction, decay=self.decay, **kwargs)
        del self.nu, self.reduction, self.decay, self.range

    def update(self, **kwargs) -> None:
        self.learning_rule.update(**kwargs)

    def normalize(self) -> None:
        """Reference: topology_features.py:250-266 (signed column sums) -> snn_normalize(use_abs=0)."""
        if self.norm is None:
            return
        if isinstance(self.norm, torch.Tensor):
            raise NotImplementedError("bindsnet_amd: per-target tensor norms are not supported")
        from .. import ops
        ops.normalize(self.value.data, float(self.norm), use_abs=False)

    def reset_state_variables(self) -> None:
        if self.learning_rule:
            self.learning_rule.reset_state_variables()

    def to(self, device):
        """Move the value (the reference forgets to: SURVEY.md finding 8)."""
        if isinstance(self.value, torch.Tensor) and self.value.device != torch.device(device):
            self.value = Parameter(self.value.data.to(device), requires_grad=False)
            if hasattr(self.learning_rule, "feature_value"):
                self.learning_rule.feature_value = self.value
        return self


class Weight(AbstractFeature):
    """Per-synapse scalar gain (reference: topology_features.py:575-671)."""

    def __init__(self, name: str, value=None, value_dtype: torch.dtype = torch.float32,
                 range: Optional[Sequence[float]] = None, norm=None, norm_frequency: Optional[str] = "sample",
                 learning_rule=None, nu: Optional[Union[list, tuple]] = None, reduction=None,
                 enforce_polarity: Optional[bool] = False, decay: float = 0.0, sparse: Optional[bool] = False,
                 batch_size: int = 1) -> None:
        if norm_frequency != "sample":
            raise NotImplementedError("bindsnet_amd: norm_frequency='time step' is outside the accelerated path")
        self.norm_frequency, self.enforce_polarity = norm_frequency, enforce_polarity
        super().__init__(name=name, value=value, value_dtype=value_dtype,
                         range=[-torch.inf, +torch.inf] if range is None else range, norm=norm,
                         learning_rule=learning_rule, nu=nu, reduction=reduction, decay=decay, sparse=sparse,
                         batch_size=batch_size, enforce_polarity=enforce_polarity)

    def prime_feature(self, connection, device, **kwargs) -> None:
        if self.value is None:
            self.initialize_value = lambda: torch.rand(connection.source.n, connection.target.n)
        super().prime_feature(connection, device, enforce_polarity=self.enforce_polarity, **kwargs)

    def reset_state_variables(self) -> None:
        pass

    def compute(self, conn_spikes):
        raise NotImplementedError("Weight.compute is fused into MulticompartmentConnection.compute on the device")
