"""State-variable monitors: API mirror of bindsnet/network/monitors.py (`Monitor`).

The reference allocates and copies every monitored tensor once per timestep
(monitors.py:94-111).  Here Network.run hands the node kernels a [T, B, *shape] buffer and they
write the raster directly, so `get()` returns a tensor with the reference's shape and contents
without any per-step host work.  Supported variables: `s` of any layer, `v` of LIF / D&C layers.
"""
from typing import Iterable, Optional

import torch


class AbstractMonitor:
    pass


class Monitor(AbstractMonitor):
    def __init__(self, obj, state_vars: Iterable[str], time: Optional[int] = None, batch_size: int = 1,
                 device: str = "cpu", sparse: Optional[bool] = False):
        if sparse:
            raise NotImplementedError("bindsnet_amd: sparse monitors are not supported")
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
        return out

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
