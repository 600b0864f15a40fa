"""Multi-GPU modes (one process per GPU, torch.distributed over RCCL; gloo on the host).  Three of them:

  sharded_run     the north-star schedule: batch shards, ONE all-reduce of the weight / threshold deltas per input.  Fast, and
                  NOT equivalent to the reference's single global batch (below).
  column_shard    exact for graphs without coupling between target neurons (Input -> connection -> LIFNodes): 32-aligned column
                  slices, no collective at all during the run.
  exact_run       exact for graphs whose batch IS coupled every timestep (DiehlAndCook2015): batch shards, one all-gather of the
                  spike bytes per timestep, the coupled operations on the global batch on every rank.  Bit-identical to the
                  single-process global batch at any world size; per-operator launches (parity and capacity, not speed).

Batch sharding across the GPUs of one node, the north-star schedule (BASELINE.json): every rank holds a replica of the weights and simulates its
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
# Exact multi-GPU mode for graphs whose batch IS coupled (DiehlAndCook2015): shard the batch, exchange the spikes of every
# timestep (SURVEY.md 8(e) "exact").
#
# What couples the samples of a batch in the reference's step: theta += theta_plus * (crossings summed
# over the batch, nodes.py:1093-1094), the one_spike draws consumed in batch-row order (nodes.py:1097-1105), and the
# learning rule's sum over the batch (MCC_learning.py:260-263,296-299), whose result feeds the next step's propagation.
# None of them needs more than the SPIKES of the other samples.  So per timestep every rank
#   (1) propagates and advances the membrane state of ITS rows (currents, decay, refractory, crossings: per-sample work),
#   (2) all-gathers the step's spike bytes of all non-input layers -- ONE collective of B_shard * sum(n) bytes,
#   (3) applies the coupled operations to the GLOBAL batch, identically on every rank (replicated, like the weights):
#       theta bump with the global crossing counts, the draws + arbitration of all rows (every rank's copy of the
#       generator advances identically), the traces, and the learning rule over the global factors in ATen's batch order.
# Every operation runs with exactly the operands the single-process global batch would hand it, so rasters, weights, theta,
# membrane state and the position of the host generator equal the single-process run bit for bit, at any world size that
# divides the batch.  The input spike trains are gathered once per run.  Cost: a collective and ~15 small launches per
# timestep (the generic operators of the C ABI, not a fused plan), and (3) is replicated work -- this mode buys batches
# beyond one device's, not speed (DESIGN.md section 6 has the numbers); `sharded_run` above is the fast, non-equivalent one.
# On the host (CPU tensors, gloo) the same schedule runs on network/host_path.py's operators: tests/test_parallel_gloo.py.
# =====================================================================================================
def _exact_check(network):
    from .learning import MCC_learning as mcc_rules
    from .learning import learning as dense_rules
    from .network.nodes import DiehlAndCookNodes, Input, LIFNodes
    from .network.topology import Connection
    for name, layer in network.layers.items():
        if type(layer) not in (Input, LIFNodes, DiehlAndCookNodes):
            raise NotImplementedError(f"exact_run: layer type {type(layer).__name__} ('{name}')")
    learned = []
    for key, conn in network.connections.items():
        if isinstance(conn, MulticompartmentConnection):
            rule = conn._weight().learning_rule
            if isinstance(rule, mcc_rules.PostPre) and not conn.manual_update:
                if not (conn.source.traces and conn.target.traces):
                    raise AssertionError("PostPre needs traces on both layers")
                learned.append(key)
            elif not isinstance(rule, mcc_rules.NoOp) and not conn.manual_update:
                raise NotImplementedError(f"exact_run: MCC rule {type(rule).__name__} (supported: PostPre)")
        elif type(conn) is Connection:
            if not isinstance(conn.update_rule, dense_rules.NoOp):
                raise NotImplementedError("exact_run: learning on a dense Connection (Input -> Connection -> LIFNodes graphs shard "
                                          "their columns exactly with column_shard, without any collective)")
        else:
            raise NotImplementedError(f"exact_run: connection type {type(conn).__name__}")
        if isinstance(network.layers[key[1]], Input):
            raise NotImplementedError("exact_run: a connection into an Input layer")
    return learned


class _ExactHost:
    """The per-rank and the replicated operators of exact_run as plain PyTorch (network/host_path.py's statements)."""

    def __init__(self, network, B, dev):
        from .network import host_path
        self.hp, self.net = host_path, network

    def begin(self):
        return self

    def prop(self, conn, s_rows, cur, acc):
        out = self.hp._propagate(conn, s_rows.view(s_rows.shape[0], *conn.source.shape))
        if acc:
            cur += out.view_as(cur)
        else:
            cur.copy_(torch.zeros_like(cur) + out.view_as(cur))          # zeros + c1 (network.py:240-248)

    def trace(self, layer, s, x):
        self.hp._trace_into(layer, s.view_as(x), x)

    def lif(self, layer, cur, rv):
        self.hp._lif_membrane(layer, cur.view(cur.shape[0], *layer.shape))
        if rv is not None:
            rv.copy_(layer.v)

    def dc_membrane(self, layer, cur, rv):
        self.hp._dc_membrane(layer, cur.view(cur.shape[0], *layer.shape))
        if layer.lbound is not None:
            layer.v.masked_fill_(layer.v < layer.lbound, layer.lbound)
        if rv is not None:
            rv.copy_(layer.v)

    def dc_couple(self, layer, s_g):
        """theta bump + one_spike on the global crossings `s_g` [B, n] u8, in place."""
        sb = s_g.bool()
        self.hp._dc_theta(layer, sb.view(sb.shape[0], *layer.shape))
        if layer.one_spike:
            self.hp._one_spike(sb)
            s_g.copy_(sb)

    def postpre(self, conn, s_src, x_src, s_tgt, x_tgt, t):
        feat = conn._weight()
        self.hp._postpre_mcc(feat.learning_rule, feat.value.data, s_src, x_src, s_tgt, x_tgt, float(self.net.dt))

    def end(self, ok):
        pass


class _ExactDevice:
    """The same on the MI355X: the C ABI's per-operator entry points (include/snnhip.h), torch for memory and streams."""

    def __init__(self, network, B, dev):
        from . import ops
        from .network.nodes import DiehlAndCookNodes
        from .rng import DeviceGenerator
        self.ops, self.net = ops, network
        draws = max([B * l.n for l in network.layers.values() if isinstance(l, DiehlAndCookNodes) and l.one_spike] or [0])
        self.gen = DeviceGenerator(dev, draws)
        self.params = {}

    def begin(self):
        self.gen.__enter__()
        return self

    def prop(self, conn, s_rows, cur, acc):
        if isinstance(conn, MulticompartmentConnection):
            self.ops.prop_cascade(conn._weight().value.data, s_rows, cur, accumulate=acc)
        else:
            b = getattr(conn, "b", None)
            self.ops.prop_dense(conn.w.data, s_rows, cur, bias=None if b is None else b.data, accumulate=acc)

    def trace(self, layer, s, x):
        from .network.nodes import _f
        self.ops.input_step(s, x, _f(layer.trace_decay), _f(layer.trace_scale), layer.traces_additive)

    def lif(self, layer, cur, rv):
        p = self.params.get(id(layer))
        if p is None:
            p = self.params[id(layer)] = layer._lif_params()
            p.traces = 0                                   # the trace is advanced on the global spikes (step 3)
        self.ops.lif_step(layer.v, layer.refrac_count, layer.s, None, cur, p, None, rv)

    def dc_membrane(self, layer, cur, rv):
        p = self.params.get(id(layer))
        if p is None:
            full = layer._dc_params()
            half = layer._dc_params()
            half.lif.traces, half.one_spike, half.learning = 0, 0, 0     # membrane half only; theta is handled around it
            full.lif.traces = 0
            p = self.params[id(layer)] = (half, full)
        if layer.learning:
            layer.theta.mul_(layer.theta_decay)            # nodes.py:1078-1079 (one f32 multiply per neuron, as in k_dc_membrane)
        self.ops.dc_step(layer.v, layer.refrac_count, layer.s, None, layer.theta, cur, p[0], None, None, None, None, rv)

    def dc_couple(self, layer, s_g):
        if layer.learning:                                 # nodes.py:1093-1094: counts are exact in f32; multiply, THEN add
            bump = s_g.sum(0, dtype=torch.float32)
            bump.mul_(layer.theta_plus)
            layer.theta.view(-1).add_(bump)
        if layer.one_spike:
            self.ops.rng_fill_exponential(self.gen.state, s_g, self.gen.qbuf, self.gen.cursor)
            self.ops.dc_arbitrate(s_g, None, self.params[id(layer)][1], self.gen.qbuf, self.gen.cursor, self.gen.status)

    def postpre(self, conn, s_src, x_src, s_tgt, x_tgt, t):
        rule = conn._weight().learning_rule
        lo, hi = rule._bounds()
        self.ops.stdp_postpre(conn._weight().value.data, s_src, x_src, s_tgt, x_tgt, float(rule.nu[0]), float(rule.nu[1]), True,
                              float(self.net.dt), float(rule.decay), lo, hi, assume_clamped=t > 0)

    def end(self, ok):
        if ok:
            self.gen.finish()
        else:
            self.gen.__exit__(RuntimeError, None, None)


def _exact_gathered(network, inputs, T, world, rank, gather, dev):
    """exact_run's `gathered` mode: ONE all-gather of the run's inputs and of the layers' per-sample state at entry, then every rank
    runs the GLOBAL batch through Network.run -- on the MI355X the resident D&C kernel, one launch for the whole run -- and keeps its
    own rows of the state and of the monitors' recordings.

    Why that is the right shape for this graph and not a cop-out: the coupled operations of DiehlAndCook2015 (theta, the one_spike draws in
    global row order, PostPre's batch sums, whose result feeds the next step's propagation: nodes.py:1093-1105,
    MCC_learning.py:260-263,296-299) need EVERY sample's factors on every rank at every timestep, and they are all of the step's work but
    the membrane update of the own rows (1 % of it) -- an exact mode replicates them whatever it exchanges.  Exchanging the inputs once
    instead of the spikes 250 times gives every rank the same numbers with no per-step traffic at all: the single-GPU kernel's speed at
    any world size (the per-step mode: 2.3 k timesteps/s at world 2), bit-identical to the single-process global batch by construction
    (same kernel, same operands, same draws from identically seeded generators).  What it does not buy is capacity: the global batch
    must fit one device's plan (B <= 32 for the resident kernel, 256 for the generic plan) -- beyond that `per-step` is the mode."""
    import os
    import time as _time
    from .network.monitors import Monitor
    Bs = network.batch_size
    B, lo = Bs * world, rank * Bs
    hi = lo + Bs
    u8 = torch.uint8
    timing = os.environ.get("SNN_EXACT_TIMING") == "1"          # developer aid: where a gathered run's wall time goes (device-synchronised marks)
    marks = []

    def mark(what):
        if timing:
            if dev.type == "cuda":
                torch.cuda.synchronize()
            marks.append((what, _time.perf_counter()))
    mark("start")
    # ---- ONE exchange per run: [bit-packed input spike trains | per-sample state of every layer] of the own rows as one byte string
    pieces, layout = [], []                                    # layout: (kind, key, dtype, shape of the own part)
    for name, x in inputs.items():
        x = x.to(dev).contiguous()
        n = x[0].numel() // Bs
        flat = (x.view(u8) if x.dtype == torch.bool else x)[:T].reshape(T, Bs, n)
        if flat.dtype != u8 or bool((flat > 1).any()):
            raise NotImplementedError("exact_run (gathered): input spike trains must be 0/1 bytes")
        npad = (-n) % 8
        bits = torch.nn.functional.pad(flat, (0, npad)) if npad else flat
        packed = (bits.view(T, Bs, -1, 8) * torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=u8, device=dev)).sum(-1, dtype=torch.int32).to(u8)
        pieces.append(packed.permute(1, 0, 2).reshape(Bs, -1))   # rows of a sample together: [Bs, T * ceil(n / 8)]
        layout.append(("in", name, x.shape[2:], n))
    state_names = ("s", "x", "v", "refrac_count")
    for name, layer in network.layers.items():                 # (identical replicas: every rank has the same tensors allocated, so the same layout)
        n = layer.n
        for nm in state_names:
            t = getattr(layer, nm, None)
            if not isinstance(t, torch.Tensor) or t.numel() != Bs * n:
                continue                                       # (a layer that has never run: Network.run starts it at rest)
            own = t.reshape(Bs, n).contiguous()
            pieces.append((own.view(u8) if own.dtype == torch.bool else own.view(u8)).reshape(Bs, -1))
            layout.append(("state", (name, nm), own.dtype, n))
    own_bytes = torch.cat(pieces, dim=1).contiguous()          # [Bs, bytes per sample]
    mark("packed")
    allb = gather(own_bytes) if world > 1 else own_bytes        # [B, bytes per sample], rows in rank order
    mark("exchanged")
    full_in, kept, off = {}, {}, 0
    for kind, key, a, n in layout:
        if kind == "in":
            w = T * ((n + 7) // 8)
            pk = allb[:, off:off + w].reshape(B, T, -1)
            off += w
            bits = ((pk.unsqueeze(-1) >> torch.arange(8, device=dev, dtype=u8)) & 1).reshape(B, T, -1)[:, :, :n]
            full_in[key] = bits.permute(1, 0, 2).contiguous().view(T, B, *a)
        else:
            name, nm = key
            w = n * (1 if a in (torch.bool, u8) else 4)
            kept[key] = allb[:, off:off + w].contiguous().view(a).reshape(B, *network.layers[name].shape).clone()
            off += w
    stash = {}
    for key, m in network.monitors.items():
        if not isinstance(m, Monitor):
            raise NotImplementedError("exact_run: Monitor objects on layers ('s', 'v') are supported")
        stash[key] = (m.recording, m.clean)
        m.recording, m.clean = {v: [] for v in m.state_vars}, True
    network.batch_size = B
    for name, layer in network.layers.items():
        layer.set_batch_size(B)
        for nm in state_names:
            if (name, nm) in kept:
                setattr(layer, nm, kept[(name, nm)])
    mark("state set")
    try:
        network.run(full_in, time=T * network.dt)
        mark("run")
    finally:
        inner = network.__dict__.get("last_plan")
        network.batch_size = Bs
        for name, layer in network.layers.items():
            cur = {nm: getattr(layer, nm, None) for nm in state_names}
            layer.set_batch_size(Bs)
            for nm, t in cur.items():
                if isinstance(t, torch.Tensor) and t.numel() == B * layer.n:
                    setattr(layer, nm, t.reshape(B, *layer.shape)[lo:hi].clone())
        for key, m in network.monitors.items():
            new, (old, clean) = m.recording, stash[key]
            m.recording, m.clean = old, clean
            for v in m.state_vars:
                for chunk in new[v]:
                    m._append(v, chunk[:, lo:hi].clone())
    network.__dict__["last_plan"] = "exact-gathered:" + str(inner)
    mark("restored")
    if timing:
        network.__dict__["_exact_timing"] = {b[0]: round((b[1] - a[1]) * 1e3, 3) for a, b in zip(marks, marks[1:])}


def exact_run(network, inputs: Dict[str, torch.Tensor], time: int, group=None, comm=None, mode: str = "auto") -> None:
    """network.run() for a batch that is sharded over the ranks of `group`, EXACTLY: `inputs` holds this rank's rows
    (rank r owns rows [r * B_shard, (r + 1) * B_shard) of the global batch), the network is this rank's replica with
    batch size B_shard, and after the call weights, theta, the state of the own rows, monitors and the host generator
    are what the single-process run of the global batch leaves (see the section comment).  Every rank must enter with the
    same weights / theta and the same state of the global generator (torch.manual_seed).  The exchange goes through
    torch.distributed (`group`; RCCL or gloo) or, with `comm` = a parallel.NativeComm, through the C ABI's own RCCL collective
    snn_dist_allgather_step -- what a caller on the other side of the boundary would use (device tensors only).

    mode: "per-step" = the schedule of the section comment (one all-gather of the spike bytes per timestep, per-operator launches: a global
    batch beyond one device's plan); "gathered" = one all-gather of the inputs and the state per run, then the global batch through
    Network.run on every rank (_exact_gathered: the resident kernel's speed, global batch <= what one device's plan takes); "auto" =
    gathered where the global batch is at most 32 samples (the resident kernel's), per-step otherwise."""
    from .network.monitors import Monitor
    from .network.nodes import DiehlAndCookNodes, Input
    assert type(inputs) == dict, "'inputs' must be a dict of names of layers (str) and relevant input tensors."
    if comm is not None:
        world, rank = comm.world, comm.rank
    else:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0

    def exchange(recv, own):
        """recv[r] <- rank r's `own` (recv: [world, *own.shape], contiguous)."""
        if comm is not None:
            recv.copy_(comm.allgather(own.contiguous()).view_as(recv))
        elif world > 1:
            dist.all_gather(list(recv.unbind(0)), own, group=group)
        else:
            recv[0].copy_(own)

    for key in inputs:                                        # network.py:329-353
        if inputs[key].dim() == 1:
            inputs[key] = inputs[key].unsqueeze(0).unsqueeze(0)
        elif inputs[key].dim() == 2:
            inputs[key] = inputs[key].unsqueeze(1)
    for key in inputs:
        if inputs[key].size(1) != network.batch_size:
            network.batch_size = inputs[key].size(1)
            for l in network.layers.values():
                l.set_batch_size(network.batch_size)
            for m in network.monitors.values():
                m.reset_state_variables()
        break
    learned = _exact_check(network)
    T, Bs = int(time / network.dt), network.batch_size
    B, lo = Bs * world, rank * Bs
    hi = lo + Bs
    dev = network._device()
    if T <= 0:
        network._normalize_all()
        return
    if mode not in ("auto", "gathered", "per-step"):
        raise ValueError(f"exact_run: mode {mode!r}")
    if mode == "gathered" or (mode == "auto" and B <= 32 and world > 1):

        def gather2(own):
            own = own.contiguous()
            out = torch.empty(world, *own.shape, dtype=own.dtype, device=own.device)
            exchange(out, own)
            return out.view(B, *own.shape[1:])
        return _exact_gathered(network, inputs, T, world, rank, gather2, dev)
    if B > 256 and learned:
        raise NotImplementedError("exact_run: the learning operators take global batches of up to 256 samples")
    u8 = torch.uint8

    def gather(own):
        """[Bs, k] (any dtype) of every rank -> [B, k], rows in rank order."""
        own = own.contiguous()
        out = torch.empty(world, *own.shape, dtype=own.dtype, device=own.device)
        exchange(out, own)
        return out.view(B, *own.shape[1:])

    def bytes_of(t, n):
        t = t.contiguous()
        return (t.view(u8) if t.dtype == torch.bool else t.to(u8)).view(-1, n)

    names = list(network.layers)
    s_g, x_g, in_own, in_g, cur = {}, {}, {}, {}, {}
    for name, layer in network.layers.items():
        n = layer.n
        if isinstance(layer, Input):
            if name not in inputs:
                raise NotImplementedError(f"bindsnet_amd: Input layer '{name}' needs an entry in `inputs`")
            x = inputs[name]
            if x.dtype not in (u8, torch.bool):
                raise NotImplementedError(f"bindsnet_amd: input spike trains must be uint8 or bool (got {x.dtype})")
            if x.shape[0] < T or x[0].numel() != Bs * n:
                raise ValueError(f"inputs['{name}'] has shape {tuple(x.shape)}, expected [T >= {T}, {Bs}, {n}]")
            x = inputs[name] = x.to(dev).contiguous()
            in_own[name] = bytes_of(x[:T], Bs * n).view(T, Bs, n)
            if world == 1 and comm is None:
                in_g[name] = in_own[name]
            else:                                            # the spike trains of all ranks, once per run: [T, B, n]
                out = torch.empty(world, T, Bs, n, dtype=u8, device=dev)
                exchange(out, in_own[name])
                in_g[name] = out.permute(1, 0, 2, 3).reshape(T, B, n).contiguous()
            entry = layer.s
            if entry.numel() != Bs * n or entry.device != dev:
                entry = torch.zeros(Bs, n, dtype=u8, device=dev)
        else:
            if name in inputs:
                raise NotImplementedError("exact_run: external currents into non-Input layers")
            network._check_state(layer, Bs, dev)
            cur[name] = torch.zeros(Bs, n, device=dev)
            entry = layer.s
        s_g[name] = gather(bytes_of(entry, n))              # the state the run starts from is the layers' own
        if layer.traces:
            x_g[name] = gather(layer.x.reshape(Bs, n).float())
    others = [nm for nm in names if not isinstance(network.layers[nm], Input)]
    width = sum(network.layers[nm].n for nm in others)
    send = torch.empty(Bs, width, dtype=u8, device=dev)
    recv = torch.empty(world, Bs, width, dtype=u8, device=dev)
    # monitors: spikes and voltages of the OWN rows, [T, B_shard, *shape] like run()'s
    ras_s, ras_v, appends = {}, {}, []
    for m in network.monitors.values():
        if not isinstance(m, Monitor) or not any(m.obj is l for l in network.layers.values()):
            raise NotImplementedError("exact_run: Monitor objects on layers ('s', 'v') are supported")
        name = next(nm for nm, l in network.layers.items() if l is m.obj)
        for var in m.state_vars:
            if var == "s":
                if name not in ras_s:
                    ras_s[name] = torch.empty(T, Bs, *m.obj.shape, dtype=torch.bool, device=dev)
                appends.append((m, var, ras_s[name]))
            elif var == "v" and hasattr(m.obj, "v"):
                if name not in ras_v:
                    ras_v[name] = torch.empty(T, Bs, *m.obj.shape, device=dev)
                appends.append((m, var, ras_v[name]))
            else:
                raise NotImplementedError(f"exact_run: monitoring '{var}' of {type(m.obj).__name__}")

    ops_ = (_ExactDevice if dev.type == "cuda" else _ExactHost)(network, B, dev).begin()
    ok = False
    try:
        for t in range(T):
            # (1a) network.py:384 _get_inputs(): previous-step spikes of the OWN rows through every connection, in order
            fed = set()
            for (src, dst), conn in network.connections.items():
                ops_.prop(conn, s_g[src][lo:hi], cur[dst], dst in fed)
                fed.add(dst)
            # (1b) layers in insertion order: the per-sample half of every step
            off = 0
            for name, layer in network.layers.items():
                if isinstance(layer, Input):
                    s_g[name] = in_g[name][t]
                    if layer.traces:
                        ops_.trace(layer, s_g[name], x_g[name])
                    continue
                if name not in fed:
                    cur[name].zero_()
                rv = ras_v[name][t] if name in ras_v else None
                if isinstance(layer, DiehlAndCookNodes):
                    ops_.dc_membrane(layer, cur[name], rv)
                else:
                    ops_.lif(layer, cur[name], rv)
                send[:, off:off + layer.n].copy_(bytes_of(layer.s, layer.n))
                off += layer.n
            # (2) the step's spikes of all ranks
            exchange(recv, send)
            # (3) the coupled half, on the global batch, identically on every rank
            off = 0
            for name in others:
                layer = network.layers[name]
                s_g[name].view(world, Bs, layer.n).copy_(recv[:, :, off:off + layer.n])
                off += layer.n
                if isinstance(layer, DiehlAndCookNodes):
                    ops_.dc_couple(layer, s_g[name])
                if layer.traces:
                    ops_.trace(layer, s_g[name], x_g[name])
                if name in ras_s:
                    ras_s[name][t].view(Bs, layer.n).copy_(s_g[name][lo:hi])
            if network.learning:
                for key in learned:
                    ops_.postpre(network.connections[key], s_g[key[0]], x_g[key[0]], s_g[key[1]], x_g[key[1]], t)
        ok = True
    finally:
        ops_.end(ok)
    # hand the own rows back to the layers (what run() leaves): final spikes, traces; Input.s aliases the last input slice
    for name, layer in network.layers.items():
        if isinstance(layer, Input):
            layer.s = inputs[name][T - 1]
            if name in ras_s:
                ras_s[name].view(T, Bs, layer.n).copy_(in_own[name])
        else:
            layer.s = s_g[name][lo:hi].bool().view(Bs, *layer.shape).clone()
        if layer.traces:
            layer.x = x_g[name][lo:hi].view(Bs, *layer.shape).clone()
    for m, var, buf in appends:
        m._append(var, buf)
    network.__dict__["last_plan"] = "exact-sharded"
    if not network.__dict__.get("_defer_norm", False):
        network._normalize_all()


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
