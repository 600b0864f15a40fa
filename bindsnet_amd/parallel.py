"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed over RCCL).

Schedule (BASELINE.json north star): every rank holds a replica of the weights and simulates its
own shard of the batch for one input (`network.run`), then the weight and adaptive-threshold
DELTAS of that input are summed over ranks (all-reduce over xGMI), clamped, and the weights are
re-normalised -- one collective of Nin*N*4 + N*4 bytes per input, nothing per timestep.

This is NOT equivalent to the reference running the global batch in one process: there the batch
couples every timestep (theta += theta_plus * spikes summed over the batch, the STDP batch sum
feeds the next step's propagation, one_spike consumes RNG rows in global batch order) -- see
SURVEY.md 8(e) and DESIGN.md "Multi-GPU".  Each rank's rasters are bit-exact for ITS shard given
the merged weights at input boundaries.
"""
from typing import Dict

import torch
import torch.distributed as dist

from .network.topology import MulticompartmentConnection


def _learned(network):
    """(tensor, lo, hi, norm_holder) for every connection that learns."""
    out = []
    for conn in network.connections.values():
        if isinstance(conn, MulticompartmentConnection):
            feat = conn.pipeline[0]
            rule = feat.learning_rule
            if type(rule).__name__ in ("PostPre", "MSTDP", "MSTDPET"):     # every MCC rule this package implements
                lo, hi = rule._bounds()
                out.append((feat.value.data, lo, hi, feat))
        elif type(conn.update_rule).__name__ in ("PostPre", "MSTDP", "Hebbian", "WeightDependentPostPre", "MSTDPET"):
            lo, hi = conn.update_rule._bounds()
            out.append((conn.w.data, lo, hi, conn))
    return out


def merge_deltas(tensors_before, tensors_after, group=None):
    """after <- before + sum_over_ranks(after - before), in place; one flat all-reduce."""
    deltas = [a - b for a, b in zip(tensors_after, tensors_before)]
    flat = torch.cat([d.reshape(-1) for d in deltas])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for a, b in zip(tensors_after, tensors_before):
        n = a.numel()
        a.copy_(b + flat[off:off + n].view_as(a))
        off += n


def _shard_buffers(network, tensors):
    """Two flat f32 buffers (values before the run / deltas) with one view per merged tensor, kept on the network
    between calls (not model state: dropped by Network.__getstate__)."""
    key = tuple((t.data_ptr(), t.numel(), str(t.device)) for t in tensors)
    st = network.__dict__.get("_shard_state")
    if st is None or st["key"] != key:
        total = sum(t.numel() for t in tensors)
        dev = tensors[0].device if tensors else "cpu"
        before, delta = torch.empty(total, device=dev), torch.empty(total, device=dev)
        views, off = [], 0
        for t in tensors:
            n = t.numel()
            views.append((before[off:off + n].view_as(t), delta[off:off + n].view_as(t)))
            off += n
        st = network.__dict__["_shard_state"] = {"key": key, "before": before, "delta": delta, "views": views}
    return st


def sharded_run(network, inputs: Dict[str, torch.Tensor], time: int, group=None, **kwargs) -> None:
    """network.run on this rank's batch shard, then merge learning across ranks (see module doc): the weights and
    thresholds become  before + sum_over_ranks(after - before), are clamped, and only then normalised.  The post-run
    normalisation inside run() is switched off through the network's `_defer_norm` flag (part of the key of the kept
    descriptor arrays, so consecutive sharded runs re-use them like plain runs do)."""
    learned = _learned(network)
    thetas = [l.theta for l in network.layers.values() if hasattr(l, "theta")] if network.learning else []
    tensors = [t for t, _, _, _ in learned] + thetas
    st = _shard_buffers(network, tensors)
    for t, (b, _) in zip(tensors, st["views"]):
        b.copy_(t)
    network.__dict__["_defer_norm"] = True   # normalisation must see the MERGED weights: postponed
    try:
        network.run(inputs, time=time, **kwargs)
    finally:
        network.__dict__["_defer_norm"] = False
    for t, (b, d) in zip(tensors, st["views"]):
        torch.sub(t, b, out=d)
    if tensors and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(st["delta"], op=dist.ReduceOp.SUM, group=group)      # ONE flat collective per input
    for t, (b, d) in zip(tensors, st["views"]):
        torch.add(b, d, out=t)
    for t, lo, hi, _ in learned:
        if lo is not None or hi is not None:
            t.clamp_(min=lo, max=hi)
    # run() normalises EVERY connection after the loop (network.py:463-465), learning or not; with the in-run
    # normalisation switched off that is done here, on the merged weights
    network._normalize_all()


# =====================================================================================================
# Exact multi-GPU mode for coupling-free graphs: shard the TARGET COLUMNS, not the batch.
#
# Input -> one connection -> LIFNodes (TwoLayerNetwork / cfg3, the cfg5 graph) has no interaction between target
# neurons at all: the current, membrane state, traces, the learning rule's update of W[:, j] and the normalisation of
# column j depend on column j and on the inputs only.  So G ranks that each hold a contiguous slice of the columns
# and ALL samples of the batch compute, with no collective whatsoever, exactly the numbers the single-process
# reference computes for the global batch -- bit for bit, provided a shard's columns keep their ATen summation class:
# slices start at multiples of 32 columns (so the <32 "row_sum" tail columns, if any, sit at the end of the last
# slice, where they are again the tail) and Nin * N is a multiple of 32 (no flat-index tail in the batch reductions).
# This is the mode whose 2 / 4 / 8-GPU results meet the north star's parity bar; the batch-sharded `sharded_run` above
# is the north star's RCCL schedule, which does not.
# =====================================================================================================
def _scalar_bound(t) -> float:
    if isinstance(t, torch.Tensor) and t.numel() != 1:
        raise NotImplementedError("column sharding needs scalar wmin / wmax (per-synapse bounds are not supported)")
    return float(t)


def column_shard_bounds(n_columns: int, world: int, rank: int, align: int = 32):
    """[lo, hi) of rank's column slice: whole blocks of `align` columns dealt out as evenly as possible (the first
    ranks get the extra blocks); a trailing partial block belongs to the last non-empty slice."""
    nblocks = n_columns // align
    if nblocks == 0:
        return (0, n_columns) if rank == 0 else (n_columns, n_columns)
    per, extra = divmod(nblocks, world)
    first = rank * per + min(rank, extra)
    count = per + (1 if rank < extra else 0)
    if count == 0:
        return n_columns, n_columns
    lo, hi = first * align, (first + count) * align
    last_owner = min(world, nblocks) - 1
    if rank == last_owner:
        hi = n_columns
    return lo, min(hi, n_columns)


def column_shard(network, rank: int, world: int):
    """A new Network holding target columns [lo, hi) of a coupling-free two-layer `network` (same input layer, same
    rule and constants, weights / bias / layer state sliced).  Running it on the full batch gives exactly the
    full network's numbers for those columns.  Returns (shard_network, lo, hi)."""
    from .learning import learning as dense_rules
    from .learning import MCC_learning as mcc_rules
    from .network import Network
    from .network.nodes import Input, LIFNodes
    from .network.topology import Connection
    from .network.topology_features import Weight
    layers, conns = list(network.layers.items()), list(network.connections.items())
    if len(layers) != 2 or len(conns) != 1 or not isinstance(layers[0][1], Input) or type(layers[1][1]) is not LIFNodes:
        raise NotImplementedError("column sharding applies to Input -> one connection -> LIFNodes graphs (no coupling "
                                  "between target neurons); DiehlAndCook2015's lateral inhibition couples them")
    (xn, X), (yn, Y) = layers
    (key, conn) = conns[0]
    N, Nin = Y.n, X.n
    lo, hi = column_shard_bounds(N, world, rank)
    if (Nin * N) % 32 or (hi > lo and (Nin * (hi - lo)) % 32):
        raise NotImplementedError("exact column sharding needs Nin * N (and every slice) to be a multiple of 32 elements")
    n = hi - lo
    if n == 0:
        return None, lo, hi
    shard = Network(dt=network.dt, batch_size=network.batch_size, learning=network.learning)
    X2 = Input(n=X.n, shape=X.shape, traces=X.traces, traces_additive=X.traces_additive,
               tc_trace=float(X.tc_trace) if X.traces else 20.0, trace_scale=float(X.trace_scale) if X.traces else 1.0)
    Y2 = LIFNodes(n=n, traces=Y.traces, traces_additive=Y.traces_additive, tc_trace=float(Y.tc_trace) if Y.traces else 20.0,
                  trace_scale=float(Y.trace_scale) if Y.traces else 1.0, thresh=float(Y.thresh), rest=float(Y.rest),
                  reset=float(Y.reset), refrac=Y.refrac.item(), tc_decay=float(Y.tc_decay),
                  lbound=None if Y.lbound is None else float(Y.lbound))
    if isinstance(conn, MulticompartmentConnection):
        feat = conn._weight()
        rule = feat.learning_rule
        rule_cls = {mcc_rules.PostPre: mcc_rules.PostPre, mcc_rules.MSTDP: mcc_rules.MSTDP}.get(type(rule))
        lo_b, hi_b = (rule.min, rule.max) if rule_cls is not None else (-float("inf"), float("inf"))
        f2 = Weight(feat.name, feat.value.data[:, lo:hi].clone().cpu(), range=[lo_b, hi_b], norm=feat.norm,
                    nu=None if rule_cls is None else (float(rule.nu[0]), float(rule.nu[1])), learning_rule=rule_cls,
                    decay=0.0 if rule_cls is None or rule.decay == 1.0 else 1.0 - float(rule.decay))
        c2 = MulticompartmentConnection(X2, Y2, device="cpu", pipeline=[f2], manual_update=conn.manual_update)
        if rule_cls is not None:
            f2.learning_rule.reduction = rule.reduction
    elif isinstance(conn, Connection):
        rule = conn.update_rule
        rule_cls = type(rule) if isinstance(rule, (dense_rules.PostPre, dense_rules.MSTDP)) else None
        kw = {}
        if isinstance(rule, dense_rules.MSTDP):
            kw.update(tc_plus=float(rule.tc_plus), tc_minus=float(rule.tc_minus))
        c2 = Connection(X2, Y2, w=conn.w.data[:, lo:hi].clone().cpu(), b=None if conn.b is None else conn.b.data[lo:hi].clone().cpu(),
                        wmin=_scalar_bound(conn.wmin), wmax=_scalar_bound(conn.wmax), norm=conn.norm, update_rule=rule_cls,
                        nu=None if rule_cls is None else (float(rule.nu[0]), float(rule.nu[1])), reduction=rule.reduction,
                        weight_decay=0.0 if rule.weight_decay == 1.0 else 1.0 - float(rule.weight_decay), **kw)
    else:
        raise NotImplementedError(f"column sharding of {type(conn).__name__} is not supported")
    shard.add_layer(X2, xn)
    shard.add_layer(Y2, yn)
    shard.add_connection(c2, *key)
    dev = Y.v.device
    if dev.type != "cpu":
        shard.to(dev)
    # an MSTDP rule keeps p_plus / p_minus / the previous spikes across run() and reset_state_variables() (learning.py:
    # 1501-1574): a rule that has already run carries them into its shard -- p_plus and the source spikes whole, p_minus
    # and the target spikes by column -- so that the exact mode stays exact for a network sharded in mid-training
    old_rule = conn._weight().learning_rule if isinstance(conn, MulticompartmentConnection) else conn.update_rule
    new_rule = c2._weight().learning_rule if isinstance(c2, MulticompartmentConnection) else c2.update_rule
    if hasattr(old_rule, "p_plus") and hasattr(new_rule, "_ensure_state"):
        if old_rule.p_plus.dim() != 2 or old_rule.p_minus.shape[-1] != N:
            raise NotImplementedError("column sharding of a rule whose state is not [batch, n] (Conv2d MSTDP, MSTDPET) is not supported")
        new_rule.p_plus = old_rule.p_plus.clone()
        new_rule.p_minus = old_rule.p_minus[:, lo:hi].clone()
        new_rule._s_src_prev = old_rule._s_src_prev.clone()
        new_rule._s_tgt_prev = old_rule._s_tgt_prev[:, lo:hi].clone()
    # carry the full network's current state over (a freshly built network starts at rest like the reference)
    B = network.batch_size
    if Y.v.numel() == B * N:
        flat = lambda t: t.reshape(B, N)[:, lo:hi].clone().reshape(B, *Y2.shape)      # noqa: E731
        Y2.v, Y2.refrac_count, Y2.s = flat(Y.v), flat(Y.refrac_count), flat(Y.s)
        if Y.traces:
            Y2.x = flat(Y.x)
    if X.s.numel() == B * Nin:
        X2.s = X.s.clone()
        if X.traces:
            X2.x = X.x.clone()
    return shard, lo, hi


def gather_columns(local: torch.Tensor, n_columns: int, group=None) -> torch.Tensor:
    """Concatenate every rank's [..., hi - lo] column slice (all_gather over RCCL / gloo, ragged slices padded)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    widths = [b - a for a, b in (column_shard_bounds(n_columns, world, r) for r in range(world))]
    pad = max(widths)
    buf = torch.zeros(*local.shape[:-1], pad, dtype=local.dtype, device=local.device)
    buf[..., :local.shape[-1]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([p[..., :w] for p, w in zip(parts, widths)], dim=-1)


# =====================================================================================================
# The C ABI's own collectives (include/snnhip.h, snn_dist_*): RCCL without torch.distributed, for callers on the other
# side of the boundary.  `rendezvous` hands rank 0's 128-byte id to everybody (here: through a torch.distributed
# broadcast when a process group exists; any out-of-band channel does).
# =====================================================================================================
class NativeComm:
    def __init__(self, rank: int, world: int, unique_id: bytes = None):
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib, C
        L = _lib.lib()
        if unique_id is None:
            buf = C.create_string_buffer(128)
            if rank == 0:
                _lib.check(L.snn_dist_unique_id(buf), "snn_dist_unique_id")
            if world > 1:
                t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
                if dist.is_initialized():
                    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
                    t = t.to(dev)
                    dist.broadcast(t, src=0)
                    t = t.cpu()
                else:
                    raise ValueError("world > 1: pass rank 0's unique_id (128 bytes) to every rank")
                buf = C.create_string_buffer(bytes(t.numpy().tobytes()), 128)
            unique_id = buf.raw
        self.handle = C.c_void_p()
        _lib.check(L.snn_dist_init(rank, world, C.create_string_buffer(unique_id, 128), C.byref(self.handle)), "snn_dist_init")
        self.rank, self.world = rank, world

    def allreduce_(self, t: torch.Tensor) -> torch.Tensor:
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self._lib.check(self._lib.lib().snn_dist_allreduce_dw(self.handle, self._C.c_void_p(t.data_ptr()), t.numel(),
                                                              self._C.c_void_p(torch.cuda.current_stream().cuda_stream)), "snn_dist_allreduce_dw")
        return t

    def allgather(self, t: torch.Tensor) -> torch.Tensor:
        assert t.is_cuda and t.is_contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self._lib.check(self._lib.lib().snn_dist_allgather_step(self.handle, self._C.c_void_p(t.data_ptr()), self._C.c_void_p(out.data_ptr()),
                                                                t.numel() * t.element_size(),
                                                                self._C.c_void_p(torch.cuda.current_stream().cuda_stream)), "snn_dist_allgather_step")
        return out

    def close(self):
        if self.handle:
            self._lib.lib().snn_dist_destroy(self.handle)
            self.handle = None
