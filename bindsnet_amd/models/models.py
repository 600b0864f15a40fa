"""Model constructors: API mirror of bindsnet/models/models.py for `TwoLayerNetwork`, `DiehlAndCook2015` and
`IncreasingInhibitionNetwork` (graph wiring only -- same layers, names, constants and the same draws from
the global generator for the initial weights, so seed-for-seed construction matches)."""
from typing import Iterable, Optional, Sequence, Union

import torch

from ..learning import PostPre
from ..learning.MCC_learning import PostPre as MCCPostPre
from ..network import Network
from ..network.nodes import DiehlAndCookNodes, Input, LIFNodes
from ..network.topology import Connection, MulticompartmentConnection
from ..network.topology_features import Weight


class TwoLayerNetwork(Network):
    """Input -> dense Connection (PostPre) -> LIFNodes.  Reference: models.py:21-91."""

    def __init__(self, n_inpt: int, n_neurons: int = 100, dt: float = 1.0, wmin: float = 0.0, wmax: float = 1.0,
                 nu: Optional[Union[float, Sequence[float]]] = (1e-4, 1e-2), reduction: Optional[callable] = None,
                 norm: float = 78.4) -> None:
        super().__init__(dt=dt)
        self.n_inpt, self.n_neurons, self.dt = n_inpt, n_neurons, dt
        self.add_layer(Input(n=n_inpt, traces=True, tc_trace=20.0), name="X")
        self.add_layer(LIFNodes(n=n_neurons, traces=True, rest=-65.0, reset=-65.0, thresh=-52.0, refrac=5,
                                tc_decay=100.0, tc_trace=20.0), name="Y")
        w = 0.3 * torch.rand(n_inpt, n_neurons)
        self.add_connection(Connection(source=self.layers["X"], target=self.layers["Y"], w=w, update_rule=PostPre,
                                       nu=nu, reduction=reduction, wmin=wmin, wmax=wmax, norm=norm),
                            source="X", target="Y")


class DiehlAndCook2015(Network):
    """Diehl & Cook (2015): Input -> exc (D&C nodes, learned STDP weights) <-> inh (LIF), one-to-one
    excitation of the inhibitory layer and all-to-all-but-self lateral inhibition.
    Reference: models.py:94-244."""

    def __init__(self, n_inpt: int, device: str = "cpu", batch_size: int = None, sparse: bool = False,
                 n_neurons: int = 100, exc: float = 22.5, inh: float = 17.5, dt: float = 1.0,
                 nu: Optional[Union[float, Sequence[float]]] = (1e-4, 1e-2), reduction: Optional[callable] = None,
                 wmin: float = 0.0, wmax: float = 1.0, w_dtype: torch.dtype = torch.float32, norm: float = 78.4,
                 theta_plus: float = 0.05, tc_theta_decay: float = 1e7, inpt_shape: Optional[Iterable[int]] = None,
                 inh_thresh: float = -40.0, exc_thresh: float = -52.0) -> None:
        super().__init__(dt=dt)
        if sparse:
            raise NotImplementedError("bindsnet_amd: sparse weights are outside the accelerated path")
        self.n_inpt, self.inpt_shape, self.n_neurons = n_inpt, inpt_shape, n_neurons
        self.exc, self.inh, self.dt = exc, inh, dt
        X = Input(n=n_inpt, shape=inpt_shape, traces=True, tc_trace=20.0)
        Ae = DiehlAndCookNodes(n=n_neurons, traces=True, rest=-65.0, reset=-60.0, thresh=exc_thresh, refrac=5,
                               tc_decay=100.0, tc_trace=20.0, theta_plus=theta_plus, tc_theta_decay=tc_theta_decay)
        Ai = LIFNodes(n=n_neurons, traces=False, rest=-60.0, reset=-45.0, thresh=inh_thresh, tc_decay=10.0, refrac=2,
                      tc_trace=20.0)
        w = 0.3 * torch.rand(n_inpt, n_neurons)                    # same generator draw as models.py:184
        x_e = MulticompartmentConnection(source=X, target=Ae, device=device, pipeline=[
            Weight("weight", w, value_dtype=w_dtype, range=[wmin, wmax], norm=norm, reduction=reduction, nu=nu,
                   learning_rule=MCCPostPre, sparse=sparse, batch_size=batch_size)])
        w = exc * torch.diag(torch.ones(n_neurons))
        e_i = MulticompartmentConnection(source=Ae, target=Ai, device=device, pipeline=[
            Weight("weight", w, value_dtype=w_dtype, range=[0, exc], sparse=sparse)])
        w = -inh * (torch.ones(n_neurons, n_neurons) - torch.diag(torch.ones(n_neurons)))
        i_e = MulticompartmentConnection(source=Ai, target=Ae, device=device, pipeline=[
            Weight("weight", w, value_dtype=w_dtype, range=[-inh, 0], sparse=sparse)])
        self.add_layer(X, name="X")
        self.add_layer(Ae, name="Ae")
        self.add_layer(Ai, name="Ai")
        self.add_connection(x_e, source="X", target="Ae")
        self.add_connection(e_i, source="Ae", target="Ai")
        self.add_connection(i_e, source="Ai", target="Ae")


class IncreasingInhibitionNetwork(Network):
    """Hazan et al. (2018): Input -> dense Connection (PostPre) -> D&C nodes on a sqrt(n) x sqrt(n) grid with a recurrent
    connection whose weights grow with the grid distance between two neurons.  Reference: models.py:349-454 (same
    layers, names and constants, the same draw for the input weights; the recurrent weights are
    start_inhib + max_inhib * sqrt(d_ij) / max sqrt(d), with start_inhib on the diagonal).  Runs on the generic plan."""

    def __init__(self, n_input: int, n_neurons: int = 100, start_inhib: float = 1.0, max_inhib: float = 100.0, dt: float = 1.0,
                 nu: Optional[Union[float, Sequence[float]]] = (1e-4, 1e-2), reduction: Optional[callable] = None,
                 wmin: float = 0.0, wmax: float = 1.0, norm: float = 78.4, theta_plus: float = 0.05,
                 tc_theta_decay: float = 1e7, inpt_shape: Optional[Iterable[int]] = None, exc_thresh: float = -52.0) -> None:
        super().__init__(dt=dt)
        self.n_input, self.n_neurons, self.n_sqrt = n_input, n_neurons, int(n_neurons ** 0.5)
        self.start_inhib, self.max_inhib, self.dt, self.inpt_shape = start_inhib, max_inhib, dt, inpt_shape
        self.add_layer(Input(n=n_input, shape=inpt_shape, traces=True, tc_trace=20.0), name="X")
        self.add_layer(DiehlAndCookNodes(n=n_neurons, traces=True, rest=-65.0, reset=-60.0, thresh=exc_thresh, refrac=5,
                                         tc_decay=100.0, tc_trace=20.0, theta_plus=theta_plus, tc_theta_decay=tc_theta_decay),
                       name="Y")
        w = 0.3 * torch.rand(n_input, n_neurons)                   # same generator draw as models.py:424
        self.add_connection(Connection(source=self.layers["X"], target=self.layers["Y"], w=w, update_rule=PostPre, nu=nu,
                                       reduction=reduction, wmin=wmin, wmax=wmax, norm=norm), source="X", target="Y")
        # models.py:439-450: sqrt of the Euclidean distance between the grid positions (f64), stored as f32, scaled
        idx = torch.arange(n_neurons)
        gx, gy = (idx // self.n_sqrt).double(), (idx % self.n_sqrt).double()
        dist = torch.sqrt((gx[:, None] - gx[None, :]) ** 2 + (gy[:, None] - gy[None, :]) ** 2)
        w = torch.sqrt(dist).float()
        w = (w / w.max()) * max_inhib + start_inhib
        self.add_connection(Connection(source=self.layers["Y"], target=self.layers["Y"], w=w), source="Y", target="Y")
