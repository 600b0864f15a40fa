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
            if type(rule).__name__ == "PostPre":
                lo, hi = rule._bounds()
                out.append((feat.value.data, lo, hi, feat))
        elif type(conn.update_rule).__name__ in ("PostPre", "MSTDP"):
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


def sharded_run(network, inputs: Dict[str, torch.Tensor], time: int, group=None, **kwargs) -> None:
    """network.run on this rank's batch shard, then merge learning across ranks (see module doc)."""
    learned = _learned(network)
    thetas = [l.theta for l in network.layers.values() if hasattr(l, "theta")] if network.learning else []
    before = [t.clone() for t, _, _, _ in learned] + [t.clone() for t in thetas]
    norms = [(h, h.norm) for _, _, _, h in learned]
    for h, _ in norms:              # normalisation must see the MERGED weights: postpone it
        h.norm = None
    try:
        network.run(inputs, time=time, **kwargs)
    finally:
        for h, n in norms:
            h.norm = n
    merge_deltas(before, [t for t, _, _, _ in learned] + thetas, group)
    for t, lo, hi, _ in learned:
        if lo is not None or hi is not None:
            t.clamp_(min=lo, max=hi)
    for _, _, _, h in learned:
        h.normalize()
