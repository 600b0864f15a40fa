"""Model constructors: API mirror of bindsnet/models/models.py -- `TwoLayerNetwork`, `DiehlAndCook2015`, `DiehlAndCook2015v2`,
`IncreasingInhibitionNetwork`, `LocallyConnectedNetwork` (graph wiring only -- same layers, names, constants and the same draws from
the global generator for the initial weights, so seed-for-seed construction matches)."""
from typing import Iterable, List, Optional, Sequence, Tuple, Union

import torch

from ..learning import PostPre
from ..learning.MCC_learning import PostPre as MCCPostPre
from ..network import Network
from ..network.nodes import DiehlAndCookNodes, Input, LIFNodes
from ..network.topology import Connection, LocalConnection, MulticompartmentConnection
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
    start_inhib + max_inhib * sqrt(d_ij) / max sqrt(d), with start_inhib on the diagonal).  Its layer and connection types are
    those of the generic plan; only the construction is pinned against the reference so far (tests/test_host_plumbing.py)."""

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


class DiehlAndCook2015v2(Network):
    """DiehlAndCook2015 without the inhibitory layer: Input -> dense Connection (PostPre) -> D&C nodes that inhibit each
    other directly through a recurrent all-to-all-but-self Connection.  Reference: models.py:247-346."""

    def __init__(self, n_inpt: int, n_neurons: int = 100, inh: float = 17.5, dt: float = 1.0,
                 nu: Optional[Union[float, Sequence[float]]] = (1e-4, 1e-2), reduction: Optional[callable] = None,
                 wmin: Optional[float] = 0.0, wmax: Optional[float] = 1.0, norm: float = 78.4, theta_plus: float = 0.05,
                 tc_theta_decay: float = 1e7, inpt_shape: Optional[Iterable[int]] = None, exc_thresh: float = -52.0) -> None:
        super().__init__(dt=dt)
        self.n_inpt, self.inpt_shape, self.n_neurons, self.inh, self.dt = n_inpt, inpt_shape, n_neurons, inh, dt
        self.add_layer(Input(n=n_inpt, shape=inpt_shape, traces=True, tc_trace=20.0), name="X")
        self.add_layer(DiehlAndCookNodes(n=n_neurons, traces=True, rest=-65.0, reset=-60.0, thresh=exc_thresh, refrac=5,
                                         tc_decay=100.0, tc_trace=20.0, theta_plus=theta_plus, tc_theta_decay=tc_theta_decay),
                       name="Y")
        w = 0.3 * torch.rand(n_inpt, n_neurons)                    # same generator draw as models.py:318
        self.add_connection(Connection(source=self.layers["X"], target=self.layers["Y"], w=w, update_rule=PostPre, nu=nu,
                                       reduction=reduction, wmin=wmin, wmax=wmax, norm=norm), source="X", target="Y")
        w = -inh * (torch.ones(n_neurons, n_neurons) - torch.diag(torch.ones(n_neurons)))
        self.add_connection(Connection(source=self.layers["Y"], target=self.layers["Y"], w=w, wmin=-inh, wmax=0),
                            source="Y", target="Y")


class LocallyConnectedNetwork(Network):
    """Input -> LocalConnection (PostPre) -> D&C nodes, one neuron per (filter, receptive field); neurons that look at the
    SAME receptive field through different filters inhibit each other (recurrent Connection).  Reference: models.py:457-600."""

    def __init__(self, n_inpt: int, input_shape: List[int], kernel_size: Union[int, Tuple[int, int]],
                 stride: Union[int, Tuple[int, int]], n_filters: int, inh: float = 25.0, dt: float = 1.0,
                 nu: Optional[Union[float, Sequence[float]]] = (1e-4, 1e-2), reduction: Optional[callable] = None,
                 theta_plus: float = 0.05, tc_theta_decay: float = 1e7, wmin: float = 0.0, wmax: float = 1.0,
                 norm: Optional[float] = 0.2, exc_thresh: float = -52.0) -> None:
        super().__init__(dt=dt)
        pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)
        kernel_size, stride = pair(kernel_size), pair(stride)
        self.n_inpt, self.input_shape, self.kernel_size, self.stride = n_inpt, input_shape, kernel_size, stride
        self.n_filters, self.inh, self.dt, self.theta_plus, self.tc_theta_decay = n_filters, inh, dt, theta_plus, tc_theta_decay
        self.wmin, self.wmax, self.norm = wmin, wmax, norm
        if kernel_size == tuple(input_shape):
            conv_size = (1, 1)
        else:
            conv_size = (int((input_shape[0] - kernel_size[0]) / stride[0]) + 1,
                         int((input_shape[1] - kernel_size[1]) / stride[1]) + 1)
        n_fields = conv_size[0] * conv_size[1]
        X = Input(n=n_inpt, traces=True, tc_trace=20.0)
        Y = DiehlAndCookNodes(n=n_filters * n_fields, traces=True, rest=-65.0, reset=-60.0, thresh=exc_thresh, refrac=5,
                              tc_decay=100.0, tc_trace=20.0, theta_plus=theta_plus, tc_theta_decay=tc_theta_decay)
        local = LocalConnection(X, Y, kernel_size=kernel_size, stride=stride, n_filters=n_filters, nu=nu, reduction=reduction,
                                update_rule=PostPre, wmin=wmin, wmax=wmax, norm=norm, input_shape=input_shape)
        # models.py:570-583: -inh between neuron (f1, field) and (f2, field) for f1 != f2, i.e. (1 - I_filters) (x) I_fields
        same_field = torch.kron(torch.ones(n_filters, n_filters) - torch.eye(n_filters), torch.eye(n_fields)) != 0
        w = torch.zeros(n_filters * n_fields, n_filters * n_fields).masked_fill_(same_field, -inh)
        recurrent = Connection(Y, Y, w=w)
        self.add_layer(X, name="X")
        self.add_layer(Y, name="Y")
        self.add_connection(local, source="X", target="Y")
        self.add_connection(recurrent, source="Y", target="Y")
