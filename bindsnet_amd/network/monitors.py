"""State-variable monitors: API mirror of bindsnet/network/monitors.py (`Monitor`).

The reference allocates and copies every monitored tensor once per timestep
(monitors.py:94-111).  Here Network.run hands the node kernels a [T, B, *shape] buffer and they
write the raster directly, so `get()` returns a tensor with the reference's shape and contents
without any per-step host work.  Supported variables: `s` of any layer, `v` of LIF / D&C layers.  `Monitor(sparse=True)` hands the recording back as a
sparse COO tensor like the reference; `NetworkMonitor` keeps float recordings of `s` / `v` of the chosen layers -- and
of `w` of the chosen connections -- in a rolling window.  A monitor on a connection's `w` makes the run take the generic
plan (the fused plans keep the weights on chip for the whole run): the weights are copied out at the end of every step.
"""
import os

import numpy as np
from typing import Iterable, Optional

import torch

from .. import _lib


class AbstractMonitor(_lib.Touching):
    pass


class Monitor(AbstractMonitor):
    def __init__(self, obj, state_vars: Iterable[str], time: Optional[int] = None, batch_size: int = 1,
                 device: str = "cpu", sparse: Optional[bool] = False):
        self.obj, self.state_vars, self.time = obj, list(state_vars), time
        self.batch_size, self.device, self.sparse = batch_size, device, sparse
        if self.time is None:
            self.device = "cpu"
        self.reset_state_variables()

    # -- reference API ---------------------------------------------------------------------------
    def get(self, var: str) -> torch.Tensor:
        """[time, batch, *shape] like torch.cat(recording[var], 0) in the reference (monitors.py:75-92)."""
        if self.clean:
            return torch.empty(0, device=self.device)
        chunks = self.recording[var]
        out = chunks[0] if len(chunks) == 1 else torch.cat(chunks, 0)
        if self.time is None:
            self.recording[var] = []
        return out.to_sparse() if self.sparse else out       # (monitors.py:105-107: sparse records concatenate to a sparse tensor)

    def record(self) -> None:
        """Single-step recording for code that steps layers by hand (reference: monitors.py:94-111)."""
        self.clean = False
        for v in self.state_vars:
            data = getattr(self.obj, v).unsqueeze(0).detach().clone().to(self.device)
            self._append(v, data)

    def reset_state_variables(self) -> None:
        self.recording = {v: [] for v in self.state_vars}
        self.clean = True

    # -- used by Network.run ----------------------------------------------------------------------
    def _append(self, var: str, chunk: torch.Tensor) -> None:
        """Append a [t, B, *shape] chunk, keeping only the last `time` steps (rolling window)."""
        self.clean = False
        rec = self.recording[var]
        rec.append(chunk)
        if self.time is not None:
            total = sum(c.shape[0] for c in rec)
            while total - rec[0].shape[0] >= self.time:
                total -= rec.pop(0).shape[0]
            if total > self.time:
                rec[0] = rec[0][total - self.time:]


class NetworkMonitor(AbstractMonitor):
    """State variables of several layers at once (reference: monitors.py:127-329): `get()` returns
    {layer: {var: float tensor [time, batch, *shape]}}; with `time` set the recording is a rolling window that starts
    out as zeros, without it the recording grows.  Supported: `s` and `v` of layers, `w` of connections."""

    def __init__(self, network, layers: Optional[Iterable[str]] = None, connections: Optional[Iterable] = None,
                 state_vars: Optional[Iterable[str]] = None, time: Optional[int] = None):
        self.network = network
        self.layers = list(layers) if layers is not None else list(network.layers.keys())
        self.connections = list(connections) if connections is not None else list(network.connections.keys())
        self.state_vars = tuple(state_vars) if state_vars is not None else ("v", "s", "w")
        self.time = time
        for c in self.connections:
            for v in self.state_vars:
                if v != "w" and hasattr(network.connections[c], v):
                    raise NotImplementedError(f"bindsnet_amd: NetworkMonitor records 'w' of connections; '{v}' of connection {c} "
                                              "is outside the accelerated path")
        self.reset_state_variables()

    def _wanted(self):
        """(layer name, var) pairs this monitor records."""
        return [(l, v) for v in self.state_vars for l in self.layers if hasattr(self.network.layers[l], v)]

    def _wanted_conns(self):
        """(connection key, var) pairs this monitor records (monitors.py:176-178: whatever of `state_vars` the connection
        has -- `w` of Connection / Conv2dConnection / LocalConnection; a MulticompartmentConnection has none)."""
        return [(c, v) for v in self.state_vars for c in self.connections if hasattr(self.network.connections[c], v)]

    def get(self):
        return self.recording

    def _append(self, layer, var: str, chunk: torch.Tensor) -> None:
        """`layer`: a layer name or a connection's (source, target) key."""
        data = chunk.float()
        old = self.recording[layer][var]
        if self.time is None:
            self.recording[layer][var] = data if old.numel() == 0 else torch.cat((old.to(data.device), data), 0)
        else:
            keep = torch.cat((old.to(data.device), data), 0) if old.shape[1:] == data.shape[1:] else data
            self.recording[layer][var] = keep[-self.time:]

    def record(self) -> None:
        """Single-step recording for code that steps layers by hand (monitors.py:222-262)."""
        for l, v in self._wanted():
            self._append(l, v, getattr(self.network.layers[l], v).unsqueeze(0))
        for c, v in self._wanted_conns():
            self._append(c, v, getattr(self.network.connections[c], v).detach().unsqueeze(0).clone())
        if self.time is not None:
            self.i += 1

    def save(self, path: str, fmt: str = "npz") -> None:
        """monitors.py:264-299."""
        d = os.path.dirname(path)
        if d and not os.path.exists(d):
            os.makedirs(d)
        if fmt == "npz":
            arrays = {}
            for o, rec in self.recording.items():
                key = "-".join(o) if isinstance(o, tuple) else o
                arrays.update({"_".join([key, v]): rec[v].cpu().numpy() for v in rec})
            np.savez_compressed(path, **arrays)
        elif fmt == "pickle":
            with open(path, "wb") as f:
                torch.save(self.recording, f)

    def reset_state_variables(self) -> None:
        self.recording = {k: {} for k in self.layers + self.connections}
        if self.time is not None:
            self.i = 0
        for l, v in self._wanted():
            t = getattr(self.network.layers[l], v)
            self.recording[l][v] = torch.Tensor() if self.time is None else torch.zeros(self.time, *t.size(), device=t.device)
        for c, v in self._wanted_conns():
            t = getattr(self.network.connections[c], v)
            self.recording[c][v] = torch.Tensor() if self.time is None else torch.zeros(self.time, *t.size(), device=t.device)
