"""`Network`: API mirror of bindsnet/network/network.py.

`run()` does not loop over timesteps in Python.  It translates the network (layers and
connections in insertion order, monitors, keyword arguments) into the descriptor arrays of
include/snnhip.h and makes ONE call to snn_net_run, which executes all T timesteps on the
device (bindsnet/network/network.py:380-465 restated in bindsnet_amd/csrc/snn_run.hip).
Anything the kernels do not implement raises NotImplementedError -- there is no fallback.
"""
import ctypes as C
import os
import tempfile
from typing import Dict, Optional, Type

import torch

from .. import _lib
from ..rng import DeviceGenerator
from .monitors import AbstractMonitor, Monitor, NetworkMonitor
from .nodes import DiehlAndCookNodes, Input, LIFNodes, Nodes, _f
from .topology import AbstractConnection, Connection, Conv2dConnection, LocalConnection, MulticompartmentConnection


def load(file_name: str, map_location: str = "cpu", learning: bool = None) -> "Network":
    """Reference: network.py:12-28 (whole-object pickle)."""
    network = torch.load(open(file_name, "rb"), map_location=map_location, weights_only=False)
    if learning is not None and "learning" in vars(network):
        network.learning = learning
    return network


_DESC_CACHE = os.environ.get("SNN_DESC_CACHE", "1") != "0"      # developer switch: rebuild the descriptors on every call


class _Pipeline:
    """State of a Network.pipelined() section: the runs enqueued since the last settlement."""

    def __init__(self, depth: int):
        self.depth = max(1, int(depth))
        self.pending = []              # plan name of every unsettled run, in order
        self.status = None             # device int32 [2 * depth]: {first attempt, second attempt} status words per run
        self.block = self.host = None  # the generator block on the device / its page-locked host image
        self.host0 = None              # host generator state at the start of the batch (None: no run of the batch draws)
        self.enabled = False
        self.stream = None             # the stream the batch's runs were enqueued on (settlement waits for THAT stream)


def _dptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class Network(_lib.TouchingModule, torch.nn.Module):
    def __init__(self, dt: float = 1.0, batch_size: int = 1, learning: bool = True,
                 reward_fn: Optional[Type] = None) -> None:
        super().__init__()
        self.dt, self.batch_size = dt, batch_size
        self.layers, self.connections, self.monitors = {}, {}, {}
        self.train(learning)
        self.reward_fn = reward_fn() if reward_fn is not None else None
        self.last_plan = None          # name of the device plan the last run() used

    # ------------------------------------------------------------------ construction (network.py:119-161)
    def add_layer(self, layer: Nodes, name: str) -> None:
        self.layers[name] = layer
        self.add_module(name, layer)
        layer.train(self.learning)
        layer.compute_decays(self.dt)
        layer.set_batch_size(self.batch_size)

    def add_connection(self, connection, source: str, target: str) -> None:
        self.connections[(source, target)] = connection
        self.add_module(source + "_to_" + target, connection)
        connection.dt = self.dt
        connection.train(self.learning)

    def add_monitor(self, monitor: AbstractMonitor, name: str) -> None:
        self.monitors[name] = monitor
        monitor.network = self
        monitor.dt = self.dt

    def save(self, file_name: str) -> None:
        torch.save(self, open(file_name, "wb"))

    def clone(self) -> "Network":
        f = tempfile.SpooledTemporaryFile()
        torch.save(self, f)
        f.seek(0)
        return torch.load(f, weights_only=False)

    _TRANSIENT = ("_workspace", "_scratch_pool", "_keep", "last_plan", "resident_retries", "lean_retries", "_lean_cooldown",
                  "_run_cache", "_reset_cache", "_shard_state", "_defer_norm", "_pipe", "_ws_state")

    def __getstate__(self):
        """save() / clone() pickle the whole object like the reference (network.py:163-209); device scratch, the
        workspace of the fused plans and the references that keep the last inputs alive are not model state."""
        state = dict(self.__dict__)
        for k in self._TRANSIENT:
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self.last_plan = None

    def reset_state_variables(self) -> None:
        """Reference: network.py:467-481.  The state tensors of the stock layer classes (s, x, refrac_count <- 0,
        v <- rest; theta is NOT reset, nodes.py:1113-1120) are filled by ONE launch (snn_fill_segments) instead of
        four small fills per layer.  The segment table is kept while nothing has been assigned to a network object
        (_lib.epoch()); only Input.s, which aliases the last input slice of the previous run, is looked up again."""
        import struct
        plan = self.__dict__.get("_reset_cache")
        if plan is None or plan["epoch"] != _lib.epoch() or not _DESC_CACHE:
            segs, others, rebind = [], [], []
            for l in self.layers.values():
                if type(l) in (Input, LIFNodes, DiehlAndCookNodes) and l.s.is_cuda and l.s.is_contiguous():
                    if type(l) is Input:
                        rebind.append((len(segs), l))
                    segs.append((l.s, 0))
                    if l.traces:
                        segs.append((l.x, 0))
                    if type(l) is not Input:
                        segs += [(l.refrac_count, 0), (l.v, struct.unpack("<I", struct.pack("<f", _f(l.rest)))[0])]
                else:
                    others.append(l)
            arr = None
            if segs and all(t.is_contiguous() for t, _ in segs) and len(segs) <= _lib.MAX_FILL_SEGMENTS:
                arr = (_lib.FillSegment * len(segs))()
                for k, (t, pat) in enumerate(segs):
                    arr[k].ptr, arr[k].bytes, arr[k].pattern = t.data_ptr(), t.numel() * t.element_size(), pat
            plan = {"epoch": _lib.epoch(), "segs": segs, "arr": arr, "others": others, "rebind": rebind,
                    "rests": [(l.rest, l.rest._version) for l in self.layers.values()
                              if type(l) is not Input and isinstance(getattr(l, "rest", None), torch.Tensor)]}
            self.__dict__["_reset_cache"] = plan if all(t._version == v for t, v in plan["rests"]) else None
        elif any(t._version != v for t, v in plan["rests"]):     # layer.rest changed in place: rebuild next time, and now
            self.__dict__["_reset_cache"] = None
            return self.reset_state_variables()
        else:
            for k, l in plan["rebind"]:
                s = l.s
                if not (s.is_cuda and s.is_contiguous()):
                    self.__dict__["_reset_cache"] = None
                    return self.reset_state_variables()
                plan["segs"][k] = (s, 0)
                if plan["arr"] is not None:
                    plan["arr"][k].ptr, plan["arr"][k].bytes = s.data_ptr(), s.numel() * s.element_size()
        for l in plan["others"]:
            l.reset_state_variables()
        if plan["arr"] is not None:
            _lib.check(_lib.lib().snn_fill_segments(plan["arr"], len(plan["segs"]),
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), "snn_fill_segments")
        else:
            for t, pat in plan["segs"]:
                t.zero_() if pat == 0 else t.view(torch.int32).fill_(pat - (1 << 32) if pat >> 31 else pat)
        for c in self.connections.values():
            c.reset_state_variables()
        before = _lib.epoch()
        for m in self.monitors.values():
            m.reset_state_variables()
        # a monitor emptying its recording is an assignment, but not one the descriptors depend on
        for key in ("_run_cache", "_reset_cache"):
            kept = self.__dict__.get(key)
            if kept is not None and kept["epoch"] == before:
                kept["epoch"] = _lib.epoch()

    def train(self, mode: bool = True) -> "torch.nn.Module":
        self.learning = mode
        return super().train(mode)

    # ------------------------------------------------------------------ pipelined runs
    def pipelined(self, depth: int = 64):
        """Context manager (an extension of the reference's API): inside it run() RETURNS WITHOUT WAITING for the device --

            with network.pipelined():
                for batch in inputs:                     # tensors already on the device
                    network.run({"X": batch}, time=250)
                    network.reset_state_variables()

        The host prepares and enqueues run k+1 while run k executes (a synchronous run() costs ~0.2 ms of host time per call at
        cfg2, during which the MI355X idles).  Results are the same bit for bit; what changes is WHEN two things happen:
          * the host generator (the stream DiehlAndCookNodes' one_spike arbitration draws from, rng.py) lives on the device between
            the runs of the section and is written back when the section is settled: at sync(), at the end of the `with` block,
            every `depth` runs, and before anything in this package reads the host generator (a synchronous run, a device encoder).
            Code of the CALLER that draws from torch's CPU generator inside the section must call network.sync() first;
          * the device status word is read at settlement.  An input the lean kernel form gives up on (SNN_ERR_RETRY) is repeated on
            the general form ON THE DEVICE (snn_run_desc.status2: the second attempt is enqueued behind every first one and returns
            at once where it is not needed), so that case stays exact; a SNN_ERR_TIMEOUT (the GPU was shared and a resident grid
            was not co-resident in time) cannot be repaired after later runs have been enqueued and raises SnnError at settlement
            -- use synchronous runs on a shared GPU."""
        import contextlib

        @contextlib.contextmanager
        def section():
            if self.__dict__.get("_pipe") is not None:
                raise RuntimeError("Network.pipelined() sections do not nest")
            from .. import rng
            pipe = self.__dict__["_pipe"] = _Pipeline(depth)
            rng._PENDING.append(self.sync)
            try:
                yield self
                self.sync()
            except BaseException:
                # the body raised: what has been enqueued still runs on the device -- settle it (so that the host generator stands where
                # those runs left it) without letting a second error hide the first
                try:
                    self.sync()
                except Exception:                          # noqa: BLE001
                    pass
                raise
            finally:
                rng._PENDING.remove(self.sync)
                self.__dict__["_pipe"] = None
                pipe.pending.clear()
        return section()

    def sync(self) -> None:
        """Settle the runs of an open pipelined() section: wait for them, check their status words, bring the host generator
        up to date.  A no-op outside a section / with nothing enqueued."""
        pipe = self.__dict__.get("_pipe")
        if pipe is None or not pipe.pending:
            return
        from ..rng import RNG_STATE_BYTES, words_to_torch_state
        n = len(pipe.pending)
        if pipe.stream is not None and pipe.stream != torch.cuda.current_stream(pipe.block.device):
            pipe.stream.synchronize()                      # the caller changed streams inside the section: the read-back below is ordered behind the CURRENT one only
        pipe.host.copy_(pipe.block)                        # blocking: everything enqueued so far has run
        st = pipe.status[:2 * n].cpu().numpy().reshape(n, 2)
        pipe.pending.clear()
        bad = None
        for k in range(n):
            s1, s2 = int(st[k, 0]), int(st[k, 1])
            if s1 == _lib.SNN_ERR_RETRY and s2 == 0:       # repeated on the general form, on the device
                self.__dict__["lean_retries"] = self.__dict__.get("lean_retries", 0) + 1
                self.__dict__["_lean_cooldown"] = 16
            elif (s1 != 0 or s2 != 0) and bad is None:
                bad = (k, s1, s2)
        if bad is not None:
            if pipe.enabled:
                torch.set_rng_state(pipe.host0)
            raise _lib.SnnError(f"pipelined run {bad[0] + 1} of {n} reported status {bad[1]} (second attempt: {bad[2]}); the runs enqueued "
                                "behind it have executed on the state it left, so the section cannot be repaired -- "
                                "SNN_ERR_TIMEOUT (-6) means the GPU was shared: use synchronous runs there")
        if pipe.enabled:
            # Between the first run of the batch and now the host generator was stale (its state lived on the device).  Whoever drew
            # from it meanwhile -- torch.rand, a host encoder called without this package's flush -- took numbers from the wrong
            # position, and writing the device image back would silently discard those draws: refuse instead of diverging.
            drew = not torch.equal(torch.get_rng_state(), pipe.host0)
            torch.set_rng_state(words_to_torch_state(pipe.host.numpy()[:RNG_STATE_BYTES // 4], pipe.host0))
            if drew:
                raise RuntimeError("Network.pipelined(): torch's CPU generator was used inside the section while its state lived on the device "
                                   "(draws taken from a stale position); call network.sync() before drawing from the host generator inside a section")

    def _run_pipelined(self, pipe, built, lib, stream, dev):
        """Enqueue one run of a pipelined() section (see there)."""
        from ..rng import RNG_STATE_BYTES, torch_state_to_words
        L, Cn, R, names = built["L"], built["Cn"], built["R"], built["names"]
        block, qbuf, host = built["gen_bufs"]
        draws = built["max_draws"] > 0
        cur_stream = torch.cuda.current_stream(dev)
        if pipe.pending and (len(pipe.pending) >= pipe.depth or draws != pipe.enabled or pipe.block is not block or cur_stream != pipe.stream):
            self.sync()                                    # (a batch lives on ONE stream: its runs hand the generator block from one to the next in stream order)
        if not pipe.pending:                               # first run of a batch: the host generator goes to the device
            img = host.numpy()
            img[:] = 0
            pipe.enabled, pipe.host0 = draws, None
            if draws:
                # another network's open section holds the host generator on ITS device block: settle it first, or both would
                # consume the same mt19937 segment and the later sync() would overwrite the earlier one's position
                from .. import rng
                for settle in list(rng._PENDING):
                    if settle != self.sync:
                        settle()
                pipe.host0 = torch.get_rng_state()
                img[:RNG_STATE_BYTES // 4] = torch_state_to_words(pipe.host0)
            block.copy_(host, non_blocking=True)           # (the previous settlement's blocking read-back ordered us behind it)
            if pipe.status is None or pipe.status.device != dev:
                pipe.status = torch.zeros(2 * pipe.depth, dtype=torch.int32, device=dev)
            else:
                pipe.status.zero_()
            pipe.block, pipe.host, pipe.stream = block, host, cur_stream
        k = len(pipe.pending)
        R.rng, R.qbuf = _dptr(block[:RNG_STATE_BYTES // 4]), _dptr(qbuf)
        R.cursor = C.c_void_p(block.data_ptr() + 632 * 4)
        R.status = C.c_void_p(pipe.status.data_ptr() + 8 * k)
        R.status2 = C.c_void_p(pipe.status.data_ptr() + 8 * k + 4)
        cool = self.__dict__.get("_lean_cooldown", 0)
        R.plan = 3 if cool > 0 else 0
        if cool > 0:
            self.__dict__["_lean_cooldown"] = cool - 1
        try:
            _lib.check(lib.snn_net_run(L, len(names), Cn, len(self.connections), C.byref(R), stream), "snn_net_run")
        finally:
            R.status2 = None
        self.__dict__["last_plan"] = lib.snn_plan_name().decode()
        pipe.pending.append(self.last_plan)

    # ------------------------------------------------------------------ run
    def run(self, inputs: Dict[str, torch.Tensor], time: int, one_step=False, **kwargs) -> None:
        assert type(inputs) == dict, (
            "'inputs' must be a dict of names of layers (str) and relevant input tensors. "
            f"Got {type(inputs).__name__} instead.")
        clamps, unclamps = kwargs.get("clamp", {}) or {}, kwargs.get("unclamp", {}) or {}
        injects_v, masks = kwargs.get("injects_v", {}) or {}, kwargs.get("masks", {}) or {}
        if self.reward_fn is not None:
            kwargs["reward"] = self.reward_fn.compute(**kwargs)

        # shape normalisation + dynamic batch size (network.py:329-353)
        for key in inputs:
            if inputs[key].dim() == 1:
                inputs[key] = inputs[key].unsqueeze(0).unsqueeze(0)
            elif inputs[key].dim() == 2:
                inputs[key] = inputs[key].unsqueeze(1)
        for key in inputs:
            if inputs[key].size(1) != self.batch_size:
                self.batch_size = inputs[key].size(1)
                for l in self.layers.values():
                    l.set_batch_size(self.batch_size)
                for m in self.monitors.values():
                    m.reset_state_variables()
            break

        T, B = int(time / self.dt), self.batch_size
        dev = self._device()
        if dev.type != "cuda":
            # A network whose tensors live on the host: the plain-PyTorch step loop with the reference's semantics
            # (network/host_path.py).  Function, not speed -- the product is the MI355X path below; nothing is ever
            # silently moved between the two (a network on the GPU never falls back to this).
            if any(v.is_cuda for v in inputs.values()):
                raise _lib.SnnError("bindsnet_amd: the network is on the host but an input tensor is on the GPU: move the "
                                    "network with network.to('cuda')")
            from . import host_path
            if T <= 0:
                host_path.normalize(self)
                return
            host_path.run(self, inputs, T, one_step, kwargs)
            self.__dict__["last_plan"] = "host-torch"
            return
        if T <= 0:
            self._normalize_all()
            return
        # The descriptor arrays are kept from one call to the next: while nothing has been assigned to any object of the
        # network since they were built (_lib.epoch(), plus the in-place versions of the scalar parameter tensors they
        # were filled from) only the per-call pointers are rebound -- inputs, Input.s, monitor buffers.  Calls with
        # keyword arguments (clamp, masks, reward, ...) or one_step always build afresh.
        plain = not kwargs and not one_step and self.reward_fn is None and _DESC_CACHE
        built = self.__dict__.get("_run_cache") if plain else None
        if built is not None and not self._cache_valid(built, T, B, dev):
            built = None
        if built is None:
            built = self._build_descriptors(T, B, dev, one_step, kwargs, clamps, unclamps, injects_v, masks)
            self.__dict__["_run_cache"] = built if plain else None
        L, Cn, R, names = built["L"], built["Cn"], built["R"], built["names"]
        rasters, keep = self._bind_call(built, inputs, T, B, dev)
        gen_bufs, max_draws = built["gen_bufs"], built["max_draws"]
        lib = _lib.lib()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        # plan request of this run: automatic, unless the lean kernel form gave up on one of the last inputs (it is then
        # left alone for a while: an input it cannot finish costs a whole second run)
        pipe = self.__dict__.get("_pipe")
        if pipe is not None:
            self._run_pipelined(pipe, built, lib, stream, dev)
        cool = self.__dict__.get("_lean_cooldown", 0)
        R.plan = 3 if cool > 0 else 0
        if cool > 0 and pipe is None:
            self.__dict__["_lean_cooldown"] = cool - 1
        for attempt in (0, 1, 2) if pipe is None else ():
            with DeviceGenerator(dev, max_draws, gen_bufs) as ns:  # host generator <-> device, exact (rng.py)
                R.rng, R.qbuf = _dptr(ns.state), _dptr(ns.qbuf)
                R.cursor, R.status = _dptr(ns.cursor), _dptr(ns.status)
                _lib.check(lib.snn_net_run(L, len(names), Cn, len(self.connections), C.byref(R), stream), "snn_net_run")
                self.__dict__["last_plan"] = lib.snn_plan_name().decode()
                # plans that hand spikes between workgroups report a failed hand-off (or a step the lean form does not
                # handle) through the status word: it is read back after EVERY such run, with or without one_spike
                ns.always_read = self.last_plan.startswith("dc2015-resident") or self.last_plan == "convpp-fused"
                status = ns.finish(check_status=attempt == 2)
            if status == 0:
                break
            # The kernel returned without touching any state tensor or the generator, so the same input is simply run
            # again: SNN_ERR_RETRY (lean form, unsupported step) -> the general resident kernel; SNN_ERR_TIMEOUT (the GPU
            # was shared and the grid was not co-resident in time) -> the one-launch-per-timestep plan, which cannot wait
            # on another workgroup.
            if status == _lib.SNN_ERR_RETRY:
                self.__dict__["lean_retries"] = self.__dict__.get("lean_retries", 0) + 1
                self.__dict__["_lean_cooldown"] = 16
                R.plan = 3
            else:
                self.__dict__["resident_retries"] = self.__dict__.get("resident_retries", 0) + 1
                R.plan = 2
        if not self.__dict__.get("_defer_norm", False):     # network.py:463-465 for the connections snn_net_run does not normalise itself
            for conn in self.connections.values():
                if isinstance(conn, Conv2dConnection) and conn.norm is not None:
                    conn.normalize()
        # Input.s aliases the last input slice, as in the reference (nodes.py:219)
        before = _lib.epoch()
        for i, name, layer, _ in built["inputs"]:
            layer.s = inputs[name][T - 1]
        for mon, key, buf in rasters:
            if isinstance(key, tuple):
                mon._append(key[0], key[1], buf)       # NetworkMonitor: (layer name, variable)
            else:
                mon._append(key, buf)
        self.__dict__["_keep"] = keep
        for kept in (built, self.__dict__.get("_reset_cache")):    # the assignments just made are this call's own
            if kept is not None and kept["epoch"] == before:
                kept["epoch"] = _lib.epoch()

    # ------------------------------------------------------------------ descriptors
    def _cache_valid(self, built, T, B, dev) -> bool:
        if built["epoch"] != _lib.epoch() or built["T"] != T or built["B"] != B or built["dev"] != dev:
            return False
        if built["n_objects"] != (len(self.layers), len(self.connections), len(self.monitors)):
            return False
        if built["defer_norm"] != bool(self.__dict__.get("_defer_norm", False)):
            return False
        for t, ver in built["scalars"]:               # parameter tensors changed in place (layer.thresh.fill_(...))
            if t._version != ver:
                return False
        for seq, vals in built["lists"]:              # list-valued parameters changed element-wise (rule.nu[0] = ...)
            if (seq._version if isinstance(seq, torch.Tensor) else tuple(seq)) != vals:
                return False
        for obj, attr, ptr in built["ptrs"]:          # tensors re-homed without an assignment (t.data = ..., set_(), resize_())
            t = getattr(obj, attr, None)
            if not isinstance(t, torch.Tensor) or t.data_ptr() != ptr:
                return False
        return True

    def _build_descriptors(self, T, B, dev, one_step, kwargs, clamps, unclamps, injects_v, masks):
        """Everything of the snn_net_run arguments that does not change from call to call."""
        from . import nodes as _nodes
        epoch0 = _lib.epoch()
        keep = []                                 # tensors that must outlive the asynchronous launch
        names = list(self.layers)
        index = {n: i for i, n in enumerate(names)}
        L = (_lib.LayerDesc * len(names))()
        dyn_inputs, dyn_layers = [], []
        max_draws = 0
        scalars = _nodes._SCALARS = []
        try:
            for i, name in enumerate(names):
                layer, d = self.layers[name], L[i]
                d.n = layer.n
                if isinstance(layer, Input):
                    d.kind = _lib.LAYER_INPUT
                    layer._trace_fields(d.p.lif)
                    d.x = _dptr(layer.x) if layer.traces else None
                    wanted = []                        # (monitor, key) pairs recording this layer's spikes
                    for m in self.monitors.values():
                        if isinstance(m, Monitor) and m.obj is layer:
                            if list(m.state_vars) != ["s"]:
                                raise NotImplementedError("bindsnet_amd: Input layers can only be monitored for 's'")
                            wanted.append((m, "s"))
                        elif isinstance(m, NetworkMonitor) and (name, "s") in m._wanted():
                            wanted.append((m, (name, "s")))
                    dyn_inputs.append((i, name, layer, wanted))
                    continue
                self._check_state(layer, B, dev)
                requests = self._monitor_requests(layer, name)
                cur = self._scratch("cur_" + name, (B, layer.n), torch.float32, dev)
                d.v, d.refrac, d.s, d.current = _dptr(layer.v), _dptr(layer.refrac_count), _dptr(layer.s), _dptr(cur)
                d.x = _dptr(layer.x) if layer.traces else None
                dyn_layers.append((i, layer, requests))
                # clamp / unclamp / injects_v (network.py:395-429): [n] masks or values, or one slice per timestep
                for key, table in (("clamp", clamps), ("unclamp", unclamps)):
                    m = table.get(name)
                    if m is not None:
                        from .host_path import clamp_mask      # boolean masks, or neuron INDICES like supervised_mnist.py:201-207
                        m = clamp_mask(m, layer.n).to(dev)
                        per_step = m.dim() >= 2
                        if m.numel() != (T if per_step else 1) * layer.n and not (per_step and m.shape[0] >= T and m[0].numel() == layer.n):
                            raise ValueError(f"{key}['{name}'] must have {layer.n} entries (optionally one row per timestep)")
                        m = (m[:T] if per_step else m).ne(0).to(torch.uint8).contiguous()
                        keep.append(m)
                        setattr(d, key, _dptr(m))
                        setattr(d, key + "_per_step", int(per_step))
                inj = injects_v.get(name)
                if inj is not None:
                    inj = torch.as_tensor(inj).to(dev, torch.float32)
                    per_step = inj.dim() >= 2
                    one = inj[0] if per_step else inj
                    if one.numel() not in (layer.n, B * layer.n) or (per_step and inj.shape[0] < T):
                        raise ValueError(f"injects_v['{name}'] must have {layer.n} (or batch x {layer.n}) entries, optionally per timestep")
                    inj = (inj[:T] if per_step else inj).contiguous()
                    keep.append(inj)
                    d.inject_v, d.inject_per_step, d.inject_len = _dptr(inj), int(per_step), one.numel()
                if isinstance(layer, DiehlAndCookNodes):
                    d.kind, d.p, d.theta = _lib.LAYER_DC, layer._dc_params(), _dptr(layer.theta)
                    if layer.one_spike:
                        max_draws = max(max_draws, B * layer.n)
                elif isinstance(layer, LIFNodes):
                    d.kind = _lib.LAYER_LIF
                    d.p.lif = layer._lif_params()
                    tv = layer._thresh_vec()               # per-neuron thresholds (nodes.py:425-498; generic plan)
                    if tv is not None:
                        keep.append(tv)
                        d.thresh_vec = _dptr(tv)
                        scalars.append((layer.thresh, layer.thresh._version))      # (an in-place change rebuilds: tv may be a converted copy)
                else:
                    raise NotImplementedError(f"bindsnet_amd: layer type {type(layer).__name__} is outside the "
                                              "accelerated path (Input, LIFNodes, DiehlAndCookNodes)")
        except BaseException:
            _nodes._SCALARS = None
            raise

        try:           # ONE guard from here to the end of the collection: whatever raises in between must not leave a stale collector behind
            for table, what in ((clamps, "clamp"), (unclamps, "unclamp"), (injects_v, "injects_v")):
                for lname in table:
                    if lname not in self.layers or isinstance(self.layers[lname], Input):
                        raise NotImplementedError(f"bindsnet_amd: {what}['{lname}'] must name a non-Input layer of the network")
            Cn = (_lib.ConnDesc * max(1, len(self.connections)))()
            lists, dyn_conns = [], []
            for k, ((src, dst), conn) in enumerate(self.connections.items()):
                self._fill_conn(Cn[k], conn, index[src], index[dst], B, dev, keep, kwargs)   # (the collector stays on: clamp bounds are read through _f() as well)
                if self.__dict__.get("_defer_norm", False):    # parallel.sharded_run normalises the MERGED weights itself
                    Cn[k].has_norm = 0
                wanted = self._conn_monitor_requests(conn, (src, dst))
                if wanted:
                    dyn_conns.append((k, conn, wanted))
                rule = conn._weight().learning_rule if isinstance(conn, MulticompartmentConnection) else getattr(conn, "update_rule", None)
                nu = getattr(rule, "nu", None)
                if isinstance(nu, torch.Tensor):
                    lists.append((nu, nu._version))
                elif isinstance(nu, (list, tuple)):
                    lists.append((nu, tuple(nu)))
                mask = masks.get((src, dst))
                if mask is None:
                    mask = getattr(conn, "mask", None)         # LocalConnection's structural mask (topology.py:1468-1470)
                if mask is not None:
                    if not hasattr(conn, "w") or isinstance(conn, Conv2dConnection):
                        raise NotImplementedError("bindsnet_amd: weight masks are supported on dense connections")
                    m = torch.as_tensor(mask).to(dev).ne(0).to(torch.uint8).contiguous()
                    if m.numel() != conn.w.numel():
                        raise ValueError(f"mask of connection {(src, dst)} must have the shape of its weights")
                    keep.append(m)
                    Cn[k].mask = _dptr(m)
        except BaseException:
            _nodes._SCALARS = None
            raise

        _nodes._SCALARS = None
        # every tensor whose ADDRESS went into the descriptors: `t.data = other`, set_() or resize_() re-home a tensor
        # without any attribute assignment, so the kept arrays are only valid while these still are where they were
        ptrs = []
        for layer in self.layers.values():
            for attr in ("v", "refrac_count", "x", "theta") + (("thresh",) if isinstance(getattr(layer, "thresh", None), torch.Tensor) and layer.thresh.numel() > 1 else ()):
                t = getattr(layer, attr, None)
                if isinstance(t, torch.Tensor):
                    ptrs.append((layer, attr, t.data_ptr()))
            if not isinstance(layer, Input) and isinstance(getattr(layer, "s", None), torch.Tensor):
                ptrs.append((layer, "s", layer.s.data_ptr()))
        for conn in self.connections.values():
            if isinstance(conn, MulticompartmentConnection):
                ptrs.append((conn._weight(), "value", conn._weight().value.data_ptr()))
            else:
                for attr in ("w", "b"):
                    t = getattr(conn, attr, None)
                    if isinstance(t, torch.Tensor):
                        ptrs.append((conn, attr, t.data_ptr()))
        R = _lib.RunDesc()
        R.B, R.T, R.dt, R.learning = B, T, float(self.dt), int(self.learning)
        R.one_step = int(bool(one_step))                  # network.py:388-393 (generic plan)
        need = int(_lib.lib().snn_net_workspace_bytes(L, len(names), Cn, len(self.connections), C.byref(R)))
        if need:
            ws = self.__dict__.get("_workspace")
            if ws is None or ws.numel() < need or ws.device != dev:
                ws = self.__dict__["_workspace"] = torch.empty(need, dtype=torch.uint8, device=dev)
                self.__dict__["_ws_state"] = (C.c_ulonglong * 2)()       # snn_run_desc.host_state: what the library knows about the workspace's content
            R.workspace, R.workspace_bytes = _dptr(ws), need
            R.host_state = C.cast(self.__dict__.setdefault("_ws_state", (C.c_ulonglong * 2)()), C.c_void_p)
        pool = self.__dict__.setdefault("_scratch_pool", {})
        if "rng_host" not in pool:
            pool["rng_host"] = torch.zeros(640, dtype=torch.int32).pin_memory()
        gen_bufs = (self._scratch("rng_block", (640,), torch.int32, dev),
                    self._scratch("rng_qbuf", (max(max_draws, 1),), torch.float32, dev), pool["rng_host"])
        # (assignments made while building -- a state tensor re-created by _check_state, a rule's lazily allocated
        #  memory -- belong to this build: the descriptors already point at the new objects)
        return {"L": L, "Cn": Cn, "R": R, "names": names, "keep": keep, "inputs": dyn_inputs, "layers": dyn_layers, "conns": dyn_conns,
                "max_draws": max_draws, "gen_bufs": gen_bufs, "T": T, "B": B, "dev": dev, "scalars": scalars,
                "lists": lists, "epoch": _lib.epoch(), "n_objects": (len(self.layers), len(self.connections), len(self.monitors)),
                "epoch0": epoch0, "defer_norm": bool(self.__dict__.get("_defer_norm", False)), "ptrs": ptrs}

    def _bind_call(self, built, inputs, T, B, dev):
        """The per-call part of the descriptors: input spike trains, Input.s (it aliases the previous call's last
        slice), freshly allocated monitor buffers."""
        L = built["L"]
        keep = list(built["keep"])
        rasters = []                              # (monitor, key, tensor)
        for i, name, layer, wanted in built["inputs"]:
            if name not in inputs:
                raise NotImplementedError(f"bindsnet_amd: Input layer '{name}' needs an entry in `inputs`")
            x = inputs[name]
            if x.shape[0] < T:
                raise ValueError(f"inputs['{name}'] has {x.shape[0]} timesteps, run() needs {T}")
            if x.dtype not in (torch.uint8, torch.bool):
                raise NotImplementedError("bindsnet_amd: input spike trains must be uint8 or bool "
                                          f"(got {x.dtype}); encoders produce uint8")
            x = x.to(dev).contiguous()
            if x.numel() != x.shape[0] * B * layer.n:
                raise ValueError(f"inputs['{name}'] has shape {tuple(x.shape)}, expected [T, {B}, {layer.n}]")
            entry = layer.s
            if entry.dtype not in (torch.uint8, torch.bool) or entry.numel() != B * layer.n or entry.device != dev:
                entry = torch.zeros(B, layer.n, dtype=torch.uint8, device=dev)
            entry = entry.contiguous()
            keep += [x, entry]
            L[i].ext_spikes, L[i].s = _dptr(x), _dptr(entry)
            if wanted:                             # the raster of an input layer is a COPY of its input, like Monitor.record's
                # clone (monitors.py:94-111): the caller may refill its buffer in place.  The library makes the copy (snn_layer_desc.raster_s
                # of an INPUT layer): one device copy behind the plan, or -- third-generation D&C form -- by the launch's producer workgroups
                xcopy = torch.empty_like(x[:T])
                L[i].raster_s = _dptr(xcopy)
                rasters += [(m, key, xcopy.view(T, B, *layer.shape)) for m, key in wanted]
            else:
                L[i].raster_s = None
            inputs[name] = x
        for i, layer, requests in built["layers"]:
            # an entry of `inputs` for a non-Input layer is an external CURRENT (network.py:386-392): slice t is added to
            # the layer's summed input behind the connections' contributions (generic plan)
            ext = inputs.get(built["names"][i])
            if ext is not None:
                if ext.shape[0] < T or ext[0].numel() != B * layer.n:
                    raise ValueError(f"inputs['{built['names'][i]}'] has shape {tuple(ext.shape)}, expected [T >= {T}, {B}, {layer.n}]")
                ext = ext[:T].to(dev, torch.float32).contiguous()
                keep.append(ext)
                L[i].ext_current = _dptr(ext)
            else:
                L[i].ext_current = None
            mon_s = mon_v = None
            for m, key, var in requests:
                if var == "s":
                    if mon_s is None:              # bool like layer.s; the node kernels store a 0/1 byte for EVERY
                        # (step, sample, neuron), so the buffer needs no initialisation
                        mon_s = torch.empty(T, B, *layer.shape, dtype=torch.bool, device=dev)
                    rasters.append((m, key, mon_s))
                else:
                    if mon_v is None:
                        mon_v = torch.empty(T, B, *layer.shape, device=dev)
                    rasters.append((m, key, mon_v))
            L[i].raster_s, L[i].raster_v = _dptr(mon_s), _dptr(mon_v)
        for k, conn, wanted in built["conns"]:     # weights at the end of every timestep (monitors.py:94-111, 222-262)
            mon_w = torch.empty(T, *conn.w.shape, device=dev)
            built["Cn"][k].raster_w = _dptr(mon_w)
            rasters += [(m, key, mon_w) for m, key in wanted]
        return rasters, keep

    # ------------------------------------------------------------------ helpers
    def _scratch(self, key, shape, dtype, dev):
        """Persistent device scratch: the same addresses run after run keep the library's captured launch
        graphs valid (csrc/snn_dc2015.hip) and avoid allocator traffic."""
        pool = self.__dict__.setdefault("_scratch_pool", {})
        t = pool.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != dev:
            t = pool[key] = torch.empty(shape, dtype=dtype, device=dev)
        return t

    def _normalize_all(self):
        if self._device().type != "cuda":          # a network on the host: plain PyTorch (network/host_path.py)
            from . import host_path
            host_path.normalize(self)
            return
        for c in self.connections.values():
            c.normalize()

    def _device(self) -> torch.device:
        for l in self.layers.values():
            if hasattr(l, "v"):
                return l.v.device
        for l in self.layers.values():
            return l.s.device
        return torch.device("cpu")

    @staticmethod
    def _check_state(layer, B, dev):
        for nm in ("v", "refrac_count") + (("x",) if layer.traces else ()):
            t = getattr(layer, nm)
            if t.device != dev or t.dtype != torch.float32 or t.numel() != B * layer.n or not t.is_contiguous():
                raise ValueError(f"layer state '{nm}' must be a contiguous float32 [{B}, {layer.n}] tensor on {dev}")
        if layer.s.dtype not in (torch.bool, torch.uint8) or layer.s.numel() != B * layer.n or layer.s.device != dev \
                or not layer.s.is_contiguous():
            layer.s = torch.zeros(B, *layer.shape, dtype=torch.bool, device=dev)

    def _monitor_requests(self, layer, name):
        """(monitor, key handed back to its _append, variable) for every monitor recording this layer."""
        requests = []
        for m in self.monitors.values():
            if isinstance(m, NetworkMonitor):
                requests += [(m, (l, v), v) for l, v in m._wanted() if l == name]
            elif isinstance(m, Monitor):
                if m.obj is layer:
                    requests += [(m, v, v) for v in m.state_vars]
                elif not isinstance(m.obj, Nodes) and not any(m.obj is c for c in self.connections.values()):
                    raise NotImplementedError("bindsnet_amd: monitors on objects other than the network's layers and "
                                              "connections (features, ...) are not supported")
        for m, key, var in requests:
            if var != "s" and not (var == "v" and hasattr(layer, "v")):
                raise NotImplementedError(f"bindsnet_amd: monitoring '{var}' of {type(layer).__name__} is "
                                          "outside the accelerated path (supported: 's', 'v')")
        return requests

    def _conn_monitor_requests(self, conn, cname):
        """(monitor, key handed back to its _append) for every monitor recording this connection's weights."""
        requests = []
        for m in self.monitors.values():
            if isinstance(m, NetworkMonitor):
                requests += [(m, (c, v)) for c, v in m._wanted_conns() if c == cname]
            elif isinstance(m, Monitor) and m.obj is conn:
                if list(m.state_vars) != ["w"] or not hasattr(conn, "w"):
                    raise NotImplementedError(f"bindsnet_amd: monitoring {list(m.state_vars)} of {type(conn).__name__} is outside "
                                              "the accelerated path (supported on connections: 'w')")
                requests.append((m, "w"))
        if requests and (conn.w.dtype != torch.float32 or not conn.w.is_contiguous()):
            raise NotImplementedError("bindsnet_amd: monitored connection weights must be contiguous float32")
        return requests

    @staticmethod
    def _fill_mstdp(d, rule, kwargs, dev, keep):
        """Reward / a_plus / a_minus keyword arguments and the rule's device state (learning.py:1504-1574,
        MCC_learning.py:468-551)."""
        reward = kwargs["reward"]
        if isinstance(reward, torch.Tensor) and reward.numel() > 1:
            rv = reward.to(dev, torch.float32).reshape(-1).contiguous()
            keep.append(rv)
            d.reward_vec, reward = _dptr(rv), 0.0
        a_plus, a_minus = kwargs.get("a_plus", 1.0), kwargs.get("a_minus", -1.0)
        if isinstance(a_plus, dict) or isinstance(a_minus, dict):
            raise NotImplementedError("bindsnet_amd: per-connection a_plus/a_minus dicts are not supported")
        d.rule, d.reward, d.a_plus, d.a_minus = _lib.RULE_MSTDP, float(reward), float(a_plus), float(a_minus)
        d.decay_plus, d.decay_minus = rule._decays()
        d.p_plus, d.p_minus = _dptr(rule.p_plus), _dptr(rule.p_minus)
        d.s_src_prev, d.s_tgt_prev = _dptr(rule._s_src_prev), _dptr(rule._s_tgt_prev)

    @staticmethod
    def _fill_mstdpet(d, rule, wdecay, kwargs):
        """MSTDPET's keyword arguments and device state (learning.py:2187-2248, MCC_learning.py:652-729)."""
        lo, hi = rule._bounds()
        d.wdecay = wdecay
        d.has_min, d.wmin = int(lo is not None), lo or 0.0
        d.has_max, d.wmax = int(hi is not None), hi or 0.0
        d.nu0, d.nu1 = float(rule.nu[0]), float(rule.nu[1])
        dp, dm, de = rule._decays()
        d.rule, d.reward = _lib.RULE_MSTDPET, float(kwargs["reward"])
        d.a_plus, d.a_minus = float(kwargs.get("a_plus", 1.0)), float(kwargs.get("a_minus", -1.0))
        d.decay_plus, d.decay_minus, d.decay_e, d.tc_e = dp, dm, de, float(rule.tc_e_trace)
        d.p_plus, d.p_minus, d.e_trace = _dptr(rule.p_plus), _dptr(rule.p_minus), _dptr(rule.eligibility_trace)
        d.s_src_prev, d.s_tgt_prev = _dptr(rule._s_src_prev), _dptr(rule._s_tgt_prev)

    def _fill_conn(self, d, conn, src, dst, B, dev, keep, kwargs):
        from ..learning import learning as dense_rules
        from ..learning import MCC_learning as mcc_rules
        d.src, d.dst, d.rule, d.wdecay = src, dst, _lib.RULE_NONE, 1.0
        if isinstance(conn, MulticompartmentConnection):
            feat = conn._weight()
            if feat.value.device != dev:
                feat.to(dev)
            val = feat.value
            if val.dtype != torch.float32 or not val.is_contiguous() or tuple(val.shape) != (conn.source.n, conn.target.n):
                raise NotImplementedError(f"bindsnet_amd: Weight.value must be a contiguous float32 [{conn.source.n}, "
                                          f"{conn.target.n}] tensor (got {val.dtype}, shape {tuple(val.shape)}, "
                                          f"contiguous={val.is_contiguous()})")
            d.kind, d.w = _lib.CONN_MCC, _dptr(feat.value.data)
            rule = feat.learning_rule
            if isinstance(rule, mcc_rules.PostPre) and not conn.manual_update:
                if rule.reduction is torch.squeeze and B != 1:
                    raise RuntimeError("reduction=torch.squeeze requires batch size 1")
                lo, hi = rule._bounds()
                d.rule, d.use_dt, d.wdecay = _lib.RULE_POSTPRE, 1, float(rule.decay)
                d.nu0, d.nu1 = float(rule.nu[0]), float(rule.nu[1])
                d.has_min, d.wmin = int(lo is not None), lo or 0.0
                d.has_max, d.wmax = int(hi is not None), hi or 0.0
            elif isinstance(rule, mcc_rules.MSTDP) and not conn.manual_update:
                if rule.reduction is torch.squeeze and B != 1:
                    raise RuntimeError("reduction=torch.squeeze requires batch size 1")
                if "reward" not in kwargs:
                    raise KeyError("reward")
                rule._ensure_state()
                lo, hi = rule._bounds()
                d.wdecay = float(rule.decay)
                d.has_min, d.wmin = int(lo is not None), lo or 0.0
                d.has_max, d.wmax = int(hi is not None), hi or 0.0
                d.nu0, d.nu1 = float(rule.nu[0]), float(rule.nu[1])
                self._fill_mstdp(d, rule, kwargs, dev, keep)
            elif isinstance(rule, mcc_rules.MSTDPET) and not conn.manual_update:
                if B != 1:
                    raise NotImplementedError("MCC MSTDPET is defined for batch size 1 (MCC_learning.py:665-666)")
                if "reward" not in kwargs:
                    raise KeyError("reward")
                rule._ensure_state()
                self._fill_mstdpet(d, rule, float(rule.decay), kwargs)
            elif not isinstance(rule, mcc_rules.NoOp):
                raise NotImplementedError(f"bindsnet_amd: MCC rule {type(rule).__name__} is not supported")
            if feat.norm is not None:
                if isinstance(feat.norm, torch.Tensor):
                    raise NotImplementedError("bindsnet_amd: tensor norms are not supported")
                ws = self._scratch(f"norm_{src}_{dst}", (conn.target.n,), torch.float32, dev)
                d.has_norm, d.norm, d.norm_abs, d.norm_ws = 1, float(feat.norm), 0, _dptr(ws)
            return
        if not isinstance(conn, AbstractConnection):
            raise NotImplementedError(f"bindsnet_amd: connection type {type(conn).__name__} is not supported")
        if conn.w.device != dev:
            raise ValueError("connection weights are not on the network's device; call network.to('cuda')")
        d.w = _dptr(conn.w.data)
        d.bias = _dptr(conn.b.data) if getattr(conn, "b", None) is not None else None
        if isinstance(conn, Conv2dConnection):
            d.kind = _lib.CONN_CONV2D
            d.cin, d.h, d.wd = conn.in_channels, conn.source.shape[1], conn.source.shape[2]
            d.cout, d.kh, d.kw = conn.out_channels, conn.kernel_size[0], conn.kernel_size[1]
            d.stride, d.pad = conn.stride[0], conn.padding[0]
        elif isinstance(conn, (Connection, LocalConnection)):
            d.kind = _lib.CONN_DENSE
        else:
            raise NotImplementedError(f"bindsnet_amd: connection type {type(conn).__name__} is not supported")
        rule = conn.update_rule
        if isinstance(rule, (dense_rules.PostPre, dense_rules.MSTDP)):
            rule._check_reduction()
            lo, hi = rule._bounds()
            d.wdecay = float(rule.weight_decay)
            d.has_min, d.wmin = int(lo is not None), lo or 0.0
            d.has_max, d.wmax = int(hi is not None), hi or 0.0
            d.nu0, d.nu1 = float(rule.nu[0]), float(rule.nu[1])
            if isinstance(rule, dense_rules.PostPre):
                d.rule, d.use_dt = _lib.RULE_POSTPRE, 0
                if isinstance(conn, Conv2dConnection):       # learning.py:457-497: per-sample partial sums live in scratch
                    ws = self._scratch(f"convpp_{src}_{dst}", (2 * B * conn.w.numel(),), torch.float32, dev)
                    d.rule_ws = _dptr(ws)
            elif isinstance(conn, Conv2dConnection):         # learning.py:1942-2015, batch 1
                if B != 1:
                    raise NotImplementedError("MSTDP on a Conv2dConnection is defined for batch size 1 (learning.py:2013)")
                if "reward" not in kwargs:
                    raise KeyError("reward")
                reward = kwargs["reward"]
                if isinstance(reward, torch.Tensor):
                    if reward.numel() != 1:
                        raise NotImplementedError("bindsnet_amd: MSTDP on a Conv2dConnection takes a scalar reward")
                    reward = reward.item()
                a_plus, a_minus = kwargs.get("a_plus", 1.0), kwargs.get("a_minus", -1.0)
                if isinstance(a_plus, dict) or isinstance(a_minus, dict):
                    raise NotImplementedError("bindsnet_amd: per-connection a_plus/a_minus dicts are not supported")
                rule._ensure_state()
                d.rule, d.reward, d.a_plus, d.a_minus = _lib.RULE_MSTDP, float(reward), float(a_plus), float(a_minus)
                d.decay_plus, d.decay_minus = rule._decays()
                d.p_plus, d.p_minus, d.e_trace = _dptr(rule.p_plus), _dptr(rule.p_minus), _dptr(rule._elig)
            else:
                if "reward" not in kwargs:
                    raise KeyError("reward")
                rule._ensure_state()
                self._fill_mstdp(d, rule, kwargs, dev, keep)
        elif isinstance(rule, (dense_rules.Hebbian, dense_rules.WeightDependentPostPre)):
            rule._check_reduction()
            lo, hi = rule._bounds()
            d.rule = _lib.RULE_WDPOSTPRE if isinstance(rule, dense_rules.WeightDependentPostPre) else _lib.RULE_HEBBIAN
            d.wdecay = float(rule.weight_decay)
            d.has_min, d.wmin = int(lo is not None), lo or 0.0
            d.has_max, d.wmax = int(hi is not None), hi or 0.0
            d.nu0, d.nu1 = float(rule.nu[0]), float(rule.nu[1])
        elif isinstance(rule, dense_rules.MSTDPET):
            if B != 1:
                raise NotImplementedError("MSTDPET on a dense Connection is defined for batch size 1 (learning.py:2211-2212)")
            if "reward" not in kwargs:
                raise KeyError("reward")
            rule._ensure_state()
            self._fill_mstdpet(d, rule, float(rule.weight_decay), kwargs)
        elif not isinstance(rule, dense_rules.NoOp):
            raise NotImplementedError(f"bindsnet_amd: rule {type(rule).__name__} is not supported")
        elif rule.weight_decay != 1.0 and self.learning:
            # the reference's NoOp.update still decays w every step (learning.py:87-104); not on the accelerated path
            raise NotImplementedError("bindsnet_amd: weight_decay on a connection without a learning rule is not supported")
        if conn.w.dtype != torch.float32 or not conn.w.is_contiguous():
            raise NotImplementedError("bindsnet_amd: connection weights must be contiguous float32")
        if conn.norm is not None and isinstance(conn, Conv2dConnection):
            # Conv2dConnection.normalize (topology.py:824-837) scales every filter to sum `norm`: not a column normalisation, so
            # not snn_net_run's post-loop step -- run() calls the connection's own normalize() (snn_normalize_conv2d) behind it
            if isinstance(conn.norm, torch.Tensor):
                raise NotImplementedError("bindsnet_amd: tensor norms are not supported")
        elif conn.norm is not None:
            ws = self._scratch(f"norm_{src}_{dst}", (conn.target.n,), torch.float32, dev)
            # Connection.normalize sums |w| (topology.py:383-392), LocalConnection.normalize the signed weights (:1475-1482)
            d.has_norm, d.norm, d.norm_abs, d.norm_ws = 1, float(conn.norm), int(not isinstance(conn, LocalConnection)), _dptr(ws)
