"""Learning rules for MulticompartmentConnection features: API mirror of
bindsnet/learning/MCC_learning.py for `MCC_LearningRule`, `NoOp`, `PostPre`, `MSTDP`, `MSTDPET`.
The updates themselves are snn_stdp_postpre (use_dt = 1) / snn_mstdp_step / snn_mstdpet_step."""
import warnings
from typing import Optional, Sequence, Union

import torch

from .. import _lib


class MCC_LearningRule(_lib.Touching):
    """Reference: MCC_learning.py:16-118 (nu parsing, reduction default, decay, clamp range)."""

    def __init__(self, connection, feature_value, range: Optional[Union[list, tuple]] = None,
                 nu: Optional[Union[float, Sequence[float]]] = None, reduction: Optional[callable] = None,
                 decay: float = 0.0, enforce_polarity: bool = False, **kwargs) -> None:
        self.connection = connection
        self.source, self.target = connection.source, connection.target
        self.feature_value = feature_value
        self.enforce_polarity = enforce_polarity
        if enforce_polarity:
            raise NotImplementedError("bindsnet_amd: enforce_polarity is outside the accelerated path")
        self.min, self.max = range
        if nu is None:
            nu = [0.2, 0.1]
        elif isinstance(nu, (float, int)):
            nu = [nu, nu]
        self.nu = torch.zeros(2, dtype=torch.float)
        self.nu[0], self.nu[1] = nu[0], nu[1]
        if (self.nu == torch.zeros(2)).all() and not isinstance(self, NoOp):
            warnings.warn(f"nu is set to [0., 0.] for {type(self).__name__} learning rule. "
                          "It will disable the learning process.")
        # MCC_learning.py:75-81: batch reduction defaults to sum unless the source already has batch 1
        if reduction is None:
            reduction = torch.squeeze if self.source.batch_size == 1 else torch.sum
        self.reduction = reduction
        self.decay = 1.0 - decay if decay else 1.0

    def _bounds(self):
        def one(v):
            if v is None:
                return None
            if isinstance(v, torch.Tensor):
                if v.numel() != 1:
                    raise NotImplementedError("bindsnet_amd: per-synapse clamp ranges are not supported")
                v = v.item()
            v = float(v)
            return None if v in (float("inf"), float("-inf")) else v
        return one(self.min), one(self.max)

    def update(self, **kwargs) -> None:
        raise NotImplementedError

    def reset_state_variables(self) -> None:
        pass


class NoOp(MCC_LearningRule):
    def __init__(self, **args) -> None:
        pass

    def update(self, **kwargs) -> None:
        pass

    def reset_state_variables(self) -> None:
        pass


class PostPre(MCC_LearningRule):
    """STDP with pre- (depressing) and post-synaptic (potentiating) terms.
    Reference: MCC_learning.py:149-305."""

    def __init__(self, connection, feature_value, range=None, nu=None, reduction=None, decay: float = 0.0,
                 enforce_polarity: bool = False, **kwargs) -> None:
        super().__init__(connection=connection, feature_value=feature_value,
                         range=[-1, +1] if range is None else range, nu=nu, reduction=reduction, decay=decay,
                         enforce_polarity=enforce_polarity, **kwargs)
        assert self.source.traces and self.target.traces, (
            "Both pre- and post-synaptic nodes must record spike traces "
            "(use traces='True' on source/target layers)")
        from ..network.topology import MulticompartmentConnection
        if not isinstance(connection, MulticompartmentConnection):
            raise NotImplementedError("This learning rule is not supported for this Connection type.")
        if kwargs.get("average_update", 0):
            raise NotImplementedError("bindsnet_amd: average_update buffers are outside the accelerated path")
        if self.reduction not in (torch.sum, torch.squeeze):
            raise NotImplementedError("bindsnet_amd: only reduction=torch.sum (or squeeze at batch 1) is supported")

    def update(self, **kwargs) -> None:
        """One step of MCC_learning.py:224-302 through the C ABI (used when a connection is
        updated by hand; Network.run drives the same kernel from C++)."""
        from .. import ops
        B = self.source.batch_size
        if self.reduction is torch.squeeze and B != 1:
            raise RuntimeError("reduction=torch.squeeze requires batch size 1 (as in the reference)")
        lo, hi = self._bounds()
        ops.stdp_postpre(self.feature_value.data, self.source.s.reshape(B, -1).contiguous(),
                         self.source.x.reshape(B, -1), self.target.s.reshape(B, -1), self.target.x.reshape(B, -1),
                         float(self.nu[0]), float(self.nu[1]), use_dt=True, dt=float(self.connection.dt),
                         decay=float(self.decay), wmin=lo, wmax=hi)


class MSTDP(MCC_LearningRule):
    """Reward-modulated STDP on a MulticompartmentConnection's Weight (reference: MCC_learning.py:392-551).  Same
    arithmetic as the dense rule (learning.py:1504-1574); the dense eligibility tensor of the reference is kept factored
    on the device (snn_mstdp_step), `p_plus` / `p_minus` keep their reference meaning."""

    def __init__(self, connection, feature_value, range=None, nu=None, reduction=None, decay: float = 0.0,
                 enforce_polarity: bool = False, **kwargs) -> None:
        super().__init__(connection=connection, feature_value=feature_value,
                         range=[-1, +1] if range is None else range, nu=nu, reduction=reduction, decay=decay,
                         enforce_polarity=enforce_polarity, **kwargs)
        from ..network.topology import MulticompartmentConnection
        if not isinstance(connection, MulticompartmentConnection):
            raise NotImplementedError("This learning rule is not supported for this Connection type.")
        if kwargs.get("average_update", 0):
            raise NotImplementedError("bindsnet_amd: average_update buffers are outside the accelerated path")
        if self.reduction not in (torch.sum, torch.squeeze):
            raise NotImplementedError("bindsnet_amd: only reduction=torch.sum (or squeeze at batch 1) is supported")
        self.tc_plus = torch.tensor(kwargs.get("tc_plus", 20.0))
        self.tc_minus = torch.tensor(kwargs.get("tc_minus", 20.0))

    def _ensure_state(self):
        B, dev = self.source.batch_size, self.feature_value.device
        if not hasattr(self, "p_plus") or self.p_plus.shape[0] != B or self.p_plus.device != dev:
            self.p_plus = torch.zeros(B, self.source.n, device=dev)
            self.p_minus = torch.zeros(B, self.target.n, device=dev)
            self._s_src_prev = torch.zeros(B, self.source.n, dtype=torch.uint8, device=dev)
            self._s_tgt_prev = torch.zeros(B, self.target.n, dtype=torch.uint8, device=dev)

    def _decays(self):
        dt = self.connection.dt
        return float(torch.exp(-dt / self.tc_plus)), float(torch.exp(-dt / self.tc_minus))   # MCC_learning.py:538,540

    def update(self, **kwargs) -> None:
        from .. import ops
        B = self.source.batch_size
        if self.reduction is torch.squeeze and B != 1:
            raise RuntimeError("reduction=torch.squeeze requires batch size 1 (as in the reference)")
        self._ensure_state()
        reward = kwargs["reward"]
        rvec = None
        if isinstance(reward, torch.Tensor) and reward.numel() > 1:
            rvec, reward = reward.to(self.feature_value.device, torch.float32).reshape(-1).contiguous(), 0.0
        dp, dm = self._decays()
        lo, hi = self._bounds()
        ops.mstdp_step(self.feature_value.data, self.p_plus, self.p_minus, self._s_src_prev, self._s_tgt_prev,
                       self.source.s.reshape(B, -1).contiguous(), self.target.s.reshape(B, -1), float(reward),
                       float(self.nu[0]), float(kwargs.get("a_plus", 1.0)), float(kwargs.get("a_minus", -1.0)), dp, dm,
                       wdecay=float(self.decay), wmin=lo, wmax=hi, reward_vec=rvec)

    @property
    def eligibility(self) -> torch.Tensor:
        """Dense view of the factored eligibility (for inspection only)."""
        self._ensure_state()
        return (self.p_plus.unsqueeze(2) * self._s_tgt_prev.float().unsqueeze(1)
                + self._s_src_prev.float().unsqueeze(2) * self.p_minus.unsqueeze(1))


class MSTDPET(MCC_LearningRule):
    """Reward-modulated STDP with an eligibility trace on a MulticompartmentConnection's Weight (reference:
    MCC_learning.py:554-733).  Same arithmetic as the dense rule (learning.py:2187-2248, snn_mstdpet_step) and, like it,
    defined for batch size 1 (the spikes are flattened, :665-666).  `eligibility_trace` is the rule's dense [Nin, N]
    state on the device; the point eligibility is kept as its two factors."""

    def __init__(self, connection, feature_value, range=None, nu=None, reduction=None, decay: float = 0.0,
                 enforce_polarity: bool = False, **kwargs) -> None:
        super().__init__(connection=connection, feature_value=feature_value,
                         range=[-1, +1] if range is None else range, nu=nu, reduction=reduction, decay=decay,
                         enforce_polarity=enforce_polarity, **kwargs)
        from ..network.topology import MulticompartmentConnection
        if not isinstance(connection, MulticompartmentConnection):
            raise NotImplementedError("This learning rule is not supported for this Connection type.")
        if kwargs.get("average_update", 0):
            raise NotImplementedError("bindsnet_amd: average_update buffers are outside the accelerated path")
        self.tc_plus = torch.tensor(kwargs.get("tc_plus", 20.0))
        self.tc_minus = torch.tensor(kwargs.get("tc_minus", 20.0))
        self.tc_e_trace = torch.tensor(kwargs.get("tc_e_trace", 25.0))

    def _ensure_state(self):
        dev = self.feature_value.device
        if not hasattr(self, "p_plus") or self.p_plus.device != dev:
            self.p_plus = torch.zeros(self.source.n, device=dev)
            self.p_minus = torch.zeros(self.target.n, device=dev)
            self.eligibility_trace = torch.zeros(*self.feature_value.shape, device=dev)
            self._s_src_prev = torch.zeros(self.source.n, dtype=torch.uint8, device=dev)
            self._s_tgt_prev = torch.zeros(self.target.n, dtype=torch.uint8, device=dev)

    def _decays(self):
        dt = torch.as_tensor(self.connection.dt, dtype=torch.float32)
        return (float(torch.exp(-dt / self.tc_plus)), float(torch.exp(-dt / self.tc_minus)),       # MCC_learning.py:711,713
                float(torch.exp(-dt / self.tc_e_trace)))                                           # :684-686

    @property
    def eligibility(self) -> torch.Tensor:
        """Dense view of the point eligibility (MCC_learning.py:724-726), for inspection only."""
        self._ensure_state()
        return torch.outer(self.p_plus, self._s_tgt_prev.float()) + torch.outer(self._s_src_prev.float(), self.p_minus)

    def update(self, **kwargs) -> None:
        from .. import ops
        if self.source.batch_size != 1:
            raise NotImplementedError("MCC MSTDPET is defined for batch size 1 (MCC_learning.py:665-666)")
        self._ensure_state()
        dp, dm, de = self._decays()
        lo, hi = self._bounds()
        ops.mstdpet_step(self.feature_value.data, self.eligibility_trace, self.p_plus, self.p_minus, self._s_src_prev,
                         self._s_tgt_prev, self.source.s.reshape(-1).contiguous(), self.target.s.reshape(-1), float(kwargs["reward"]),
                         float(self.nu[0]), float(self.connection.dt), float(kwargs.get("a_plus", 1.0)),
                         float(kwargs.get("a_minus", -1.0)), dp, dm, de, float(self.tc_e_trace),
                         wdecay=float(self.decay), wmin=lo, wmax=hi)

    def reset_state_variables(self) -> None:
        """MCC_learning.py:731-734: the point eligibility and its trace are cleared, P+ / P- are kept."""
        if hasattr(self, "p_plus"):
            self.eligibility_trace.zero_()
            self._s_src_prev.zero_()
            self._s_tgt_prev.zero_()
