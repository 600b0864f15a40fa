"""Network.run() for networks that live on the HOST: a plain-PyTorch step loop with the reference's semantics.

The product of this package is the MI355X path (csrc/*.hip behind the C ABI); this module exists so that the drop-in
package still constructs and runs where there is no GPU (SURVEY.md 8(b) fallback rule) -- a laptop, a CI box -- and it is a
second, independent statement of the same semantics: tests/test_host_path.py pins it to the reference-generated fixtures
the GPU tests use.  It is selected by Network.run only when the network's tensors are CPU tensors; it never touches
libsnnhip and has nothing to do with oracle/ (test infrastructure).  Supported on this path: Input / LIFNodes /
DiehlAndCookNodes; MulticompartmentConnection + Weight (no rule / PostPre / MSTDP / MSTDPET), Connection and LocalConnection (no
rule / PostPre / MSTDP / Hebbian / WeightDependentPostPre / MSTDPET), Conv2dConnection (no rule / PostPre / MSTDP at batch 1);
clamp / unclamp / injects_v / masks / one_step / reward; Monitor / NetworkMonitor.

What each function states (paths inside BindsNET): network.py:211-250,380-465 (loop, `zeros + c1 + c2` accumulation,
normalise), nodes.py:96-107,211-221,500-529,1069-1111 (layers), topology.py:332-346,437-479,799-815 and
topology_features.py:633-645 (propagation), MCC_learning.py:224-302,86-110 / learning.py:390-420,1504-1574,87-104 (rules).
Per-step cost is the reference's (one ATen call per operation): this path is for function, not speed.
"""
from typing import Dict

import torch
import torch.nn.functional as F


def aten_sum_leaves_serial_order(outer: int, n_reduced: int, n_cols: int, threads: int) -> bool:
    """True when torch.sum over the middle dimension of a contiguous float [outer, n_reduced, n_cols] tensor (outer = 1: also
    sum(dim=0) of [n_reduced, n_cols]) at `threads` intra-op threads sums the last n_cols mod 32 columns in another order than it
    does serially.  The reference's own float sums are not independent of the thread count: reductions of at least 32768
    elements are split over the outermost non-reduced dimension that has `threads` entries (if neither has: the larger one,
    ties to the outer) -- and when that is the COLUMNS, cut into `threads` ranges of c = ceil(n_cols / threads) whose ends are
    rounded down to multiples of 32, a last range holding ONLY the n_cols mod 32 < 8 tail columns is summed by another kernel
    of SumKernel.cpp (scalar_outer_sum: groups of four columns in the cascade order of a full group) than the same columns at
    the end of a longer range (vectorized_outer_sum's row_sum leftover).  E.g. n_cols = 100, batch 1: columns 96-99 at 9 or
    >= 12 threads.  Pinned against torch itself by tests/test_aten_sum_threads.py (tools/probe_aten_sum_threads.py prints it)."""
    tail = n_cols % 32
    if threads <= 1 or not 0 < tail < 8:
        return False
    if outer * n_reduced * n_cols < 32768:       # at::internal::GRAIN_SIZE: small reductions run serially
        return False
    if outer >= threads:                         # the outer dimension is split: every slice is summed serially
        return False
    if n_cols < threads and n_cols <= outer:
        return False
    c = -(-n_cols // threads)
    return c * ((n_cols - 1) // c) >= n_cols - tail


def _sum(t, dim):
    """ATen's sum in its SERIAL order, whatever the caller's thread setting: this package pins the serial order (what the
    reference computes with up to 8 threads at the sizes of BASELINE.md, and at any thread count once the batch has at least
    `threads` samples) -- the MI355X kernels, the oracle and, through this helper, the host path.  The thread count is only
    taken down for the call when the model above says the order would change (or the shape is not one it covers)."""
    n = torch.get_num_threads()
    if n == 1:
        return t.sum(dim)
    if t.is_contiguous() and t.dim() in (2, 3) and dim in (0, 1):
        if dim == 0:                                 # [B, ...] over the batch: the trailing dimensions are one run of columns
            shape = (1, t.shape[0], t[0].numel())
        else:
            shape = (t.shape[0], t.shape[1], t.shape[2]) if t.dim() == 3 else None
        if shape is not None and not aten_sum_leaves_serial_order(*shape, n):
            return t.sum(dim)
    torch.set_num_threads(1)
    try:
        return t.sum(dim)
    finally:
        torch.set_num_threads(n)


def _trace_into(layer, s, x) -> None:
    """nodes.py:96-107 on explicit tensors: the trace `x` (in place) after the spikes `s`."""
    x *= layer.trace_decay
    if layer.traces_additive:
        x += layer.trace_scale * s.float()
    else:
        x.masked_fill_(s.bool(), float(layer.trace_scale))


def _trace(layer, s) -> None:
    if layer.traces:
        _trace_into(layer, s, layer.x)


def clamp_mask(spec, n: int) -> torch.Tensor:
    """run(..., clamp= / unclamp=): the reference assigns `s[:, spec] = 1` (network.py:416-429), so `spec` is whatever indexes the
    neuron dimension -- a boolean (or byte) MASK of n entries, or an integer tensor of neuron INDICES (supervised_mnist.py:201-207
    clamps `per_class * label + choice`) -- optionally one row per timestep.  Returns the boolean mask, [n] or [T, n]."""
    m = torch.as_tensor(spec)
    if m.dtype in (torch.bool, torch.uint8) or m.is_floating_point():
        return m.ne(0)
    idx = m.long()
    if idx.dim() <= 1:
        out = torch.zeros(n, dtype=torch.bool, device=idx.device)
        out[idx.reshape(-1)] = True
        return out
    out = torch.zeros(idx.shape[0], n, dtype=torch.bool, device=idx.device)
    out.scatter_(1, idx.reshape(idx.shape[0], -1), True)
    return out


def _step_input(layer, x) -> None:
    layer.s = x                                        # aliases the caller's tensor, like the reference
    _trace(layer, x)


def _lif_membrane(layer, x) -> None:
    """nodes.py:500-526: the LIF step without its trace."""
    layer.v = layer.decay * (layer.v - layer.rest) + layer.rest
    x.masked_fill_(layer.refrac_count > 0, 0.0)        # (in place on the summed input, as the reference does)
    layer.refrac_count -= layer.dt
    layer.v += x
    layer.s = layer.v >= layer.thresh
    layer.refrac_count.masked_fill_(layer.s, float(layer.refrac))
    layer.v.masked_fill_(layer.s, float(layer.reset))
    if layer.lbound is not None:
        layer.v.masked_fill_(layer.v < layer.lbound, float(layer.lbound))


def _step_lif(layer, x) -> None:
    _lif_membrane(layer, x)
    _trace(layer, layer.s)


def _dc_membrane(layer, x) -> None:
    """nodes.py:1069-1092: everything up to the threshold crossings (left in layer.s) and their reset."""
    layer.v = layer.decay * (layer.v - layer.rest) + layer.rest
    if layer.learning:
        layer.theta *= layer.theta_decay
    layer.v += (layer.refrac_count <= 0).float() * x
    layer.refrac_count -= layer.dt
    layer.s = layer.v >= layer.thresh + layer.theta
    layer.refrac_count.masked_fill_(layer.s, float(layer.refrac))
    layer.v.masked_fill_(layer.s, float(layer.reset))


def _dc_theta(layer, crossings) -> None:
    """nodes.py:1093-1094: every neuron that crossed bumps the shared threshold, summed over the batch `crossings` spans."""
    if layer.learning:
        layer.theta += layer.theta_plus * crossings.float().sum(0)


def _one_spike(s2d) -> None:
    """nodes.py:1097-1105 on a [B, N] bool matrix, in place: one winner per row with a crossing, drawn from the global
    generator (rows in batch order)."""
    if s2d.any():
        rows = s2d.any(1)
        ind = torch.multinomial(s2d.float()[rows], 1)
        rows = rows.nonzero()
        s2d.zero_()
        s2d[rows, ind] = 1


def _step_dc(layer, x) -> None:
    B = x.shape[0]
    _dc_membrane(layer, x)
    _dc_theta(layer, layer.s)
    if layer.one_spike:
        _one_spike(layer.s.view(B, -1))
    if layer.lbound is not None:
        layer.v.masked_fill_(layer.v < layer.lbound, layer.lbound)
    _trace(layer, layer.s)


def _propagate(conn, s):
    from .topology import Connection, Conv2dConnection, LocalConnection, MulticompartmentConnection
    B = s.shape[0]
    if isinstance(conn, MulticompartmentConnection):
        value = conn._weight().value
        spikes = s.reshape(B, conn.source.n, 1).repeat(1, 1, conn.target.n)
        return _sum(value * spikes, 1).view(B, *conn.target.shape)
    if isinstance(conn, Conv2dConnection):
        return F.conv2d(s.float(), conn.w, conn.b, stride=conn.stride, padding=conn.padding, dilation=conn.dilation)
    if isinstance(conn, (Connection, LocalConnection)):
        post = s.reshape(B, -1).float() @ conn.w.view(conn.source.n, conn.target.n)
        if getattr(conn, "b", None) is not None:
            post = post + conn.b
        return post.view(B, *conn.target.shape)
    raise NotImplementedError(f"bindsnet_amd host path: connection type {type(conn).__name__}")


def _reduce(rule, t):
    if rule.reduction is torch.squeeze:
        return t.squeeze(0)
    return _sum(t, 0) if rule.reduction is torch.sum else rule.reduction(t, dim=0)


def _mstdp(rule, W, src_s, tgt_s, kwargs) -> None:
    """learning.py:1504-1574 / MCC_learning.py:468-551 (the same arithmetic): the update uses the PREVIOUS step's eligibility
    (kept as its two factors, like on the device: elig[b] = p_plus[b] (x) s_tgt_prev[b] + s_src_prev[b] (x) p_minus[b]),
    then the traces move on.  src_s / tgt_s: [B, n] float spikes of this step."""
    rule._ensure_state()
    reward = kwargs["reward"]
    elig = torch.bmm(rule.p_plus.unsqueeze(2), rule._s_tgt_prev.float().unsqueeze(1)) + \
        torch.bmm(rule._s_src_prev.float().unsqueeze(2), rule.p_minus.unsqueeze(1))
    if isinstance(reward, torch.Tensor) and reward.numel() > 1:
        reward = reward.view(-1, 1, 1).float()
    W += float(rule.nu[0]) * _reduce(rule, reward * elig)
    dp, dm = rule._decays()
    rule.p_plus *= dp
    rule.p_plus += torch.tensor(kwargs.get("a_plus", 1.0)) * src_s
    rule.p_minus *= dm
    rule.p_minus += torch.tensor(kwargs.get("a_minus", -1.0)) * tgt_s
    rule._s_src_prev, rule._s_tgt_prev = src_s.to(torch.uint8), tgt_s.to(torch.uint8)


def _mstdpet(rule, W, dt, src_s, tgt_s, kwargs) -> None:
    """learning.py:2187-2248 / MCC_learning.py:652-729 (batch 1; src_s / tgt_s flat float spikes): the eligibility TRACE
    takes the previous step's point eligibility, the update is ((nu0 * dt) * reward) * trace, then P+ / P- move on."""
    if rule.source.batch_size != 1:
        raise NotImplementedError("MSTDPET is defined for batch size 1 (learning.py:2211-2212, MCC_learning.py:665-666)")
    rule._ensure_state()
    dp, dm, de = rule._decays()
    rule.eligibility_trace *= de
    rule.eligibility_trace += rule.eligibility / rule.tc_e_trace
    W += rule.nu[0] * dt * kwargs["reward"] * rule.eligibility_trace
    rule.p_plus *= dp
    rule.p_plus += torch.tensor(kwargs.get("a_plus", 1.0)) * src_s
    rule.p_minus *= dm
    rule.p_minus += torch.tensor(kwargs.get("a_minus", -1.0)) * tgt_s
    rule._s_src_prev, rule._s_tgt_prev = src_s.to(torch.uint8), tgt_s.to(torch.uint8)


def _conv_postpre(conn, rule) -> None:
    """learning.py:457-497: im2col of the source's spikes / traces, one bmm per term, reduced over the batch."""
    from ..utils import im2col_indices
    W = conn.w.data
    Cout, _, kh, kw = W.shape
    B = conn.source.batch_size
    src_x = im2col_indices(conn.source.x.view(B, *conn.source.shape), kh, kw, padding=conn.padding, stride=conn.stride)
    src_s = im2col_indices(conn.source.s.view(B, *conn.source.shape).float(), kh, kw, padding=conn.padding, stride=conn.stride)
    tgt_x, tgt_s = conn.target.x.view(B, Cout, -1), conn.target.s.view(B, Cout, -1).float()
    if rule.nu[0].any():
        W -= rule.nu[0] * _reduce(rule, torch.bmm(tgt_x, src_s.permute(0, 2, 1))).view(W.shape)
    if rule.nu[1].any():
        W += rule.nu[1] * _reduce(rule, torch.bmm(tgt_s, src_x.permute(0, 2, 1))).view(W.shape)


def _conv_mstdp(conn, rule, kwargs) -> None:
    """learning.py:1942-2015 (batch 1).  From its second call on the reference's eligibility has the weight's shape, so its
    torch.sum(update, dim=0) runs over the OUTPUT CHANNELS and is broadcast back (in the first call the eligibility is all
    zeros either way); P+ is kept in input space here -- its im2col goes through the same elementwise operations."""
    from ..utils import im2col_indices
    if conn.source.batch_size != 1:
        raise NotImplementedError("MSTDP on a Conv2dConnection is defined for batch size 1 (learning.py:2013)")
    rule._ensure_state()
    W = conn.w.data
    Cout, _, kh, kw = W.shape
    W += rule.nu[0] * _sum(kwargs["reward"] * rule._elig, 0)
    dp, dm = rule._decays()
    src_s = conn.source.s.view(1, *conn.source.shape).float()
    tgt_s = conn.target.s.view(1, Cout, -1).float()
    rule.p_plus *= dp
    rule.p_plus += torch.tensor(kwargs.get("a_plus", 1.0)) * src_s[0]
    rule.p_minus *= dm
    rule.p_minus += torch.tensor(kwargs.get("a_minus", -1.0)) * tgt_s[0]
    unf = lambda t: im2col_indices(t, kh, kw, padding=conn.padding, stride=conn.stride)      # noqa: E731
    rule._elig = (torch.bmm(tgt_s, unf(rule.p_plus[None]).permute(0, 2, 1)) +
                  torch.bmm(rule.p_minus[None], unf(src_s).permute(0, 2, 1))).view(W.shape)


def _postpre_mcc(rule, W, s_src, x_src, s_tgt, x_tgt, dt) -> None:
    """MCC_learning.py:224-302 + :86-110 on explicit [B, n] factors (parallel.exact_run hands the GLOBAL batch's)."""
    nu0, nu1 = float(rule.nu[0]), float(rule.nu[1])
    if nu0:
        pre = torch.bmm(s_src.unsqueeze(2).float(), x_tgt.unsqueeze(1) * nu0)
        W -= _reduce(rule, pre) * dt
    if nu1:
        post = torch.bmm(x_src.unsqueeze(2), s_tgt.unsqueeze(1).float() * nu1)
        W += _reduce(rule, post) * dt
    W *= float(rule.decay)
    lo, hi = rule._bounds()
    if lo is not None or hi is not None:
        W.clamp_(lo, hi)


def _update_mcc(conn, dt, kwargs) -> None:
    from ..learning import MCC_learning as rules
    feat = conn._weight()
    rule = feat.learning_rule
    if isinstance(rule, rules.NoOp) or conn.manual_update:
        return
    B = conn.source.batch_size
    W = feat.value.data
    if rule.reduction is torch.squeeze and B != 1 and not isinstance(rule, rules.MSTDPET):
        raise RuntimeError("reduction=torch.squeeze requires batch size 1")      # (the reference fails with a broadcast error here)
    if isinstance(rule, rules.PostPre):
        _postpre_mcc(rule, W, conn.source.s.view(B, -1), conn.source.x.view(B, -1), conn.target.s.view(B, -1),
                     conn.target.x.view(B, -1), dt)
        return
    if isinstance(rule, rules.MSTDP):
        _mstdp(rule, W, conn.source.s.view(B, -1).float(), conn.target.s.view(B, -1).float(), kwargs)
    elif isinstance(rule, rules.MSTDPET):
        _mstdpet(rule, W, dt, conn.source.s.view(-1).float(), conn.target.s.view(-1).float(), kwargs)
    else:
        raise NotImplementedError(f"bindsnet_amd host path: MCC rule {type(rule).__name__} (supported: PostPre, MSTDP, MSTDPET)")
    W *= float(rule.decay)                               # MCC_learning.py:86-110
    lo, hi = rule._bounds()
    if lo is not None or hi is not None:
        W.clamp_(lo, hi)


def _update_dense(conn, kwargs, mask) -> None:
    from ..learning import learning as rules
    from .topology import Conv2dConnection
    rule = conn.update_rule
    if rule is None or isinstance(rule, rules.NoOp):
        return
    B = conn.source.batch_size
    W = conn.w.data
    conv = isinstance(conn, Conv2dConnection)
    if W.dim() != 2 and not conv:
        raise NotImplementedError(f"bindsnet_amd host path: rule {type(rule).__name__} on {type(conn).__name__}")
    nu0, nu1 = float(rule.nu[0]), float(rule.nu[1])
    if conv:
        if isinstance(rule, rules.PostPre):
            rule._check_reduction()
            _conv_postpre(conn, rule)
        elif isinstance(rule, rules.MSTDP):
            _conv_mstdp(conn, rule, kwargs)
        else:
            raise NotImplementedError(f"bindsnet_amd host path: rule {type(rule).__name__} on a Conv2dConnection (supported: PostPre, MSTDP)")
    elif isinstance(rule, rules.MSTDPET):
        _mstdpet(rule, W, conn.dt, conn.source.s.view(-1).float(), conn.target.s.view(-1).float(), kwargs)
    else:
        rule._check_reduction()
        src_s, tgt_s = conn.source.s.view(B, -1).float(), conn.target.s.view(B, -1).float()
        if isinstance(rule, rules.MSTDP):
            _mstdp(rule, W, src_s, tgt_s, kwargs)
        elif isinstance(rule, (rules.Hebbian, rules.WeightDependentPostPre)):
            # learning.py:1052-1135 / 562-653: raw outer products reduced over the batch, THEN scaled by nu
            u1 = _reduce(rule, torch.bmm(src_s.unsqueeze(2), conn.target.x.view(B, -1).unsqueeze(1)))
            u2 = _reduce(rule, torch.bmm(conn.source.x.view(B, -1).unsqueeze(2), tgt_s.unsqueeze(1)))
            if isinstance(rule, rules.Hebbian):
                W += nu0 * u1
                W += nu1 * u2
            else:
                update = 0
                if nu0:
                    update = update - nu0 * u1 * (W - conn.wmin)
                if nu1:
                    update = update + nu1 * u2 * (conn.wmax - W)
                W += update
        elif isinstance(rule, rules.PostPre):               # learning.py:390-420
            if nu0:
                W -= _reduce(rule, torch.bmm(src_s.unsqueeze(2), conn.target.x.view(B, -1).unsqueeze(1) * nu0))
            if nu1:
                W += _reduce(rule, torch.bmm(conn.source.x.view(B, -1).unsqueeze(2), tgt_s.unsqueeze(1) * nu1))
        else:
            raise NotImplementedError(f"bindsnet_amd host path: rule {type(rule).__name__} (supported: PostPre, MSTDP, Hebbian, "
                                      "WeightDependentPostPre, MSTDPET)")
    W *= float(rule.weight_decay)                          # learning.py:87-104
    lo, hi = rule._bounds()
    if lo is not None or hi is not None:
        W.clamp_(lo, hi)
    if mask is not None and not conv:
        W.masked_fill_(torch.as_tensor(mask).bool().view_as(W), 0.0)


def run(network, inputs: Dict[str, torch.Tensor], T: int, one_step: bool, kwargs) -> None:
    from .nodes import DiehlAndCookNodes, Input, LIFNodes
    from .topology import MulticompartmentConnection
    clamps, unclamps = kwargs.get("clamp", {}) or {}, kwargs.get("unclamp", {}) or {}
    injects_v, masks = kwargs.get("injects_v", {}) or {}, kwargs.get("masks", {}) or {}
    dt = float(network.dt)
    if isinstance(kwargs.get("a_plus"), dict) or isinstance(kwargs.get("a_minus"), dict):
        # network.py:356-378, 432-447 also takes {connection key: value} tables; like the MI355X path (network.py::_fill_conn)
        # this one does not, and says so instead of failing inside torch.tensor(dict)
        raise NotImplementedError("bindsnet_amd: per-connection a_plus/a_minus dicts are not supported")
    for name, layer in network.layers.items():
        if not isinstance(layer, (Input, LIFNodes, DiehlAndCookNodes)):
            raise NotImplementedError(f"bindsnet_amd host path: layer type {type(layer).__name__}")
        if isinstance(layer, Input) and name not in inputs:
            raise NotImplementedError(f"bindsnet_amd: Input layer '{name}' needs an entry in `inputs`")

    def currents(only=None):
        """Summed input per target layer from the sources' current `s` (network.py:211-250), connection order."""
        cur = {}
        for (src, dst), conn in network.connections.items():
            if only is not None and dst != only:
                continue
            tgt = network.layers[dst]
            if dst not in cur:
                cur[dst] = torch.zeros(network.batch_size, *tgt.shape)
            cur[dst] += _propagate(conn, network.layers[src].s)
        return cur

    for t in range(T):
        cur = {} if one_step else currents()
        for name, layer in network.layers.items():
            if isinstance(layer, Input):
                _step_input(layer, inputs[name][t])
            else:
                fed_now = False
                if one_step:
                    own = currents(name)
                    fed_now = name in own
                    cur.update(own)
                x = cur.get(name)
                # an external current for a non-Input layer (network.py:386-392).  With one_step the reference's
                # `current_inputs.update(self._get_inputs(layers=[l]))` (:388-393) REPLACES the entry it has just set for a
                # layer with an incoming connection: the current survives only where no connection feeds the layer
                if name in inputs and not fed_now:
                    ext = inputs[name][t].float().view(network.batch_size, *layer.shape)
                    x = ext.clone() if x is None else x + ext
                if x is None:
                    x = torch.zeros(network.batch_size, *layer.shape)
                inj = injects_v.get(name)
                if inj is not None:
                    inj = torch.as_tensor(inj)
                    layer.v += inj[t] if inj.dim() >= 2 else inj
                (_step_dc if isinstance(layer, DiehlAndCookNodes) else _step_lif)(layer, x)
                for table, value in ((clamps, 1), (unclamps, 0)):
                    m = table.get(name)
                    if m is not None:
                        m = clamp_mask(m, layer.n)
                        m = m[t] if m.dim() >= 2 else m
                        layer.s[:, m.view(*layer.shape)] = bool(value)
        if network.learning:
            for key, conn in network.connections.items():
                if isinstance(conn, MulticompartmentConnection):
                    _update_mcc(conn, dt, kwargs)
                else:
                    _update_dense(conn, kwargs, _mask_of(conn, masks.get(key)))
        for key, conn in network.connections.items():                  # masks apply every step, learning or not
            mask = _mask_of(conn, masks.get(key))
            if mask is not None and hasattr(conn, "w") and conn.w.dim() == 2:
                conn.w.data.masked_fill_(torch.as_tensor(mask).bool().view_as(conn.w), 0.0)
        for m in network.monitors.values():
            m.record()
    if not network.__dict__.get("_defer_norm", False):     # (parallel.sharded_run normalises the MERGED weights itself)
        normalize(network)


def _mask_of(conn, mask):
    """run(..., masks={...}) entry of the connection, else a LocalConnection's structural mask (topology.py:1468-1470)."""
    return mask if mask is not None else getattr(conn, "mask", None)


def normalize_connection(conn) -> None:
    """One connection's normalisation -- Weight features by their SIGNED column sums (topology_features.py:250-266), dense
    connections by the absolute ones (topology.py:383-392), a LocalConnection by the signed ones again (topology.py:1475-1482),
    a Conv2dConnection filter by filter (topology.py:824-837)."""
    from .topology import Conv2dConnection, LocalConnection, MulticompartmentConnection
    if isinstance(conn, Conv2dConnection):                       # topology.py:824-837: every [KH*KW] filter to sum `norm`
        if conn.norm is not None:
            w = conn.w.data.view(conn.w.shape[0] * conn.w.shape[1], conn.w.shape[2] * conn.w.shape[3])
            for fltr in range(w.shape[0]):
                w[fltr] *= conn.norm / w[fltr].sum(0)
        return
    if isinstance(conn, MulticompartmentConnection):
        feat = conn._weight()
        if feat.norm is not None:
            colsum = _sum(feat.value.data, 0).unsqueeze(0)
            colsum[colsum == 0] = 1.0
            feat.value.data *= feat.norm / colsum
    elif getattr(conn, "norm", None) is not None and hasattr(conn, "w") and conn.w.dim() == 2:
        colsum = _sum(conn.w.data if isinstance(conn, LocalConnection) else conn.w.data.abs(), 0).unsqueeze(0)
        colsum[colsum == 0] = 1.0
        conn.w.data *= conn.norm / colsum


def normalize(network) -> None:
    """network.py:463-465: every connection's normalisation."""
    for conn in network.connections.values():
        normalize_connection(conn)
