"""Spike encodings: API mirror of bindsnet/encoding/encodings.py (`single`, `repeat`, `bernoulli`, `poisson`,
`rank_order`).

Host path (device="cpu", the default -- what the DataLoader-driven examples use): the same torch sampling
primitives in the same order as the reference, so for a given state of the global CPU generator the spike trains
are identical and the generator is left in the same state (tests/test_encoding.py pins this against
reference-generated fixtures).

Device path (device="cuda"): `bernoulli` is produced by libsnnhip (snn_encode_bernoulli) from the HOST generator's
stream, bit for bit what the host path would give -- torch.bernoulli(p) on the CPU draws one 24-bit mt19937 value
per element -- without materialising the [time, n] probability tensor; `poisson` cannot be parallelised exactly
(ATen's sampler consumes a data-dependent number of draws per element), so the device variant draws from an
explicitly seeded counter-based stream instead: same distribution, NOT the reference's stream (see
`poisson_device`).
"""
import contextlib
import threading
from typing import Optional

import os

import torch

_ENCODER_THREADS = 4


_cap_lock = threading.Lock()
_cap_depth = 0          # encoder calls currently inside the cap (main thread only)
_cap_saved = None       # the caller's setting, taken by the outermost call


@contextlib.contextmanager
def _few_threads():
    """The host encoders work on [time, n] tensors of ~200 k elements through a dozen small ATen calls: with the
    intra-op pool a script like eth_mnist.py asks for (`torch.set_num_threads(os.cpu_count() - 1)`, 255 on the GPU box) every
    one of them pays a 255-way fork/join and one `poisson()` call took 1.6 s there against ~4 ms with a handful of threads
    (measured on the MI355X box in round 4: 5.6 ms per sample).  The samplers themselves are serial in
    the generator (ATen's cpu_serial_kernel), so the thread count changes nothing in the result: the encoders cap it for their
    own duration and put it back.

    `torch.set_num_threads` is PROCESS-global, so the cap is only applied from the main thread (a DataLoader worker thread or
    any other thread that encodes concurrently with ATen work elsewhere leaves the setting alone and just runs at the caller's
    count), and nested / re-entrant calls share one save / restore through a depth counter under a lock: the value put back is
    always the one the OUTERMOST call found."""
    global _cap_depth, _cap_saved
    if threading.current_thread() is not threading.main_thread():
        yield
        return
    with _cap_lock:
        if _cap_depth == 0:
            n0 = torch.get_num_threads()
            _cap_saved = n0 if n0 > _ENCODER_THREADS else None
            if _cap_saved is not None:
                torch.set_num_threads(_ENCODER_THREADS)
        _cap_depth += 1
    try:
        yield
    finally:
        with _cap_lock:
            _cap_depth -= 1
            if _cap_depth == 0 and _cap_saved is not None:
                torch.set_num_threads(_cap_saved)
                _cap_saved = None


def single(datum: torch.Tensor, time: int, dt: float = 1.0, sparsity: float = 0.5, device="cpu", **kwargs) -> torch.Tensor:
    """One spike at t = 0 for the features above the (1 - sparsity) quantile (encodings.py:6-33)."""
    steps = int(time / dt)
    shape = list(datum.shape)
    datum = torch.as_tensor(datum, device=device)
    quantile = torch.quantile(datum, 1 - sparsity)
    s = torch.zeros([steps, *shape], device=device)
    s[0] = torch.where(datum > quantile, torch.ones(shape, device=device), torch.zeros(shape, device=device))
    return s.byte()


def repeat(datum: torch.Tensor, time: int, dt: float = 1.0, **kwargs) -> torch.Tensor:
    """The datum repeated along a new leading time dimension (encodings.py:36-48)."""
    steps = int(time / dt)
    return datum.repeat([steps, *([1] * datum.dim())])


def _settle_sections() -> None:
    """The host encoders draw from torch's CPU generator: a network with an open pipelined() section holds that generator's state on the
    device meanwhile (rng.py) -- settle it first, so the draws come from where the reference's would."""
    from .. import rng
    rng.flush_pending()


def bernoulli(datum: torch.Tensor, time: Optional[int] = None, dt: float = 1.0, device="cpu", **kwargs) -> torch.Tensor:
    """Bernoulli spike trains, success probability = max_prob * (datum scaled into [0, 1]) (encodings.py:51-98).
    Like the reference, a datum whose maximum exceeds 1 is divided by it IN PLACE when it is contiguous."""
    max_prob = kwargs.get("max_prob", 1.0)
    assert 0 <= max_prob <= 1, "Maximum firing probability must be in range [0, 1]"
    assert (datum >= 0).all(), "Inputs must be non-negative"
    shape = datum.shape
    # the reference normalises `datum.flatten().to(device)`: the caller's tensor is divided in place exactly when that
    # expression is a view of it (contiguous datum already on `device`), a copy otherwise
    flat = datum.flatten().to(device)
    steps = None if time is None else int(time / dt)
    if flat.max() > 1.0:
        flat /= flat.max()
    if torch.device(device).type == "cuda":
        from ..ops import encode_bernoulli
        return encode_bernoulli(flat, 1 if steps is None else steps, max_prob, device).view(*(() if steps is None else (steps,)), *shape)
    _settle_sections()
    with _few_threads():
        if steps is None:
            return torch.bernoulli(max_prob * flat).view(*shape).byte()
        return torch.bernoulli(max_prob * flat.repeat([steps, 1])).view(steps, *shape).byte()


def poisson(datum: torch.Tensor, time: int, dt: float = 1.0, device="cpu", approx=False, **kwargs) -> torch.Tensor:
    """Poisson spike trains with rate = datum in Hz: inter-spike intervals ~ Poisson(1000 / (datum * dt)) (zero
    intervals bumped to one), cumulated into spike times (encodings.py:101-152)."""
    assert (datum >= 0).all(), "Inputs must be non-negative"
    shape, size = datum.shape, datum.numel()
    if torch.device(device).type == "cpu" and os.environ.get("SNN_ENCODE_DEVICE") and not approx:
        # A script that builds PoissonEncoder(time, dt) without a device (examples/mnist/eth_mnist.py:103-112 encodes every sample on the host
        # inside the DataLoader) gets the device encoder without being edited: SNN_ENCODE_DEVICE=cuda.  Opt-in: the device stream is a
        # specified one of its own (poisson_device), not the reference's CPU generator's.  The result comes back where the caller asked for it
        # (the host: the script's DataLoader pins it and the script moves it with .cuda() -- two 196 KB copies against 3.7 ms of host encoding).
        return poisson_device(datum, time, dt=dt, device=os.environ["SNN_ENCODE_DEVICE"], **kwargs).to(device)
    if torch.device(device).type == "cuda":
        return poisson_device(datum, time, dt=dt, device=device, **kwargs)
    with _few_threads():
        return _poisson_host(datum, time, dt, device, approx, shape, size)


def _poisson_host(datum, time, dt, device, approx, shape, size):
    _settle_sections()
    flat = datum.flatten().to(device)
    steps = int(time / dt)
    if approx:      # the reference's "fast, less accurate" variant: |N(0,1)| ^ ((x * 0.11 + 5) / 50) < 0.6
        x = torch.randn((steps, size), device=device).abs()
        x = torch.pow(x, (flat * 0.11 + 5) / 50)
        return (x < 0.6).view(steps, *shape).byte()
    nz = flat != 0
    rate = torch.zeros(size, device=device)
    rate[nz] = 1 / flat[nz] * (1000 / dt)
    # torch.distributions.Poisson(rate).sample([steps + 1]) is torch.poisson on the expanded rate
    intervals = torch.poisson(rate.expand(steps + 1, size))
    intervals[:, nz] += (intervals[:, nz] == 0).float()
    times = torch.cumsum(intervals, dim=0).long()
    times[times >= steps + 1] = 0
    spikes = torch.zeros(steps + 1, size, device=device).byte()
    spikes[times, torch.arange(size)] = 1
    return spikes[1:].view(steps, *shape)


def poisson_device(datum: torch.Tensor, time: int, dt: float = 1.0, device="cuda", seed: Optional[int] = None, **kwargs) -> torch.Tensor:
    """Poisson encoding generated on the MI355X (snn_encode_poisson): the same construction as `poisson` -- intervals
    ~ Poisson(1000 / (x dt)), zeros bumped to one, cumulated -- one thread per input element walking its own spike
    times, from a counter-based stream keyed by (seed, element).  NOT stream-compatible with the reference's CPU
    generator: `seed` defaults to one draw from the global CPU generator (so torch.manual_seed still makes runs
    repeatable).  The stream is specified operation by operation (csrc/snn_encode.hip) and restated on the CPU
    (oracle/snn_oracle.c: orc_encode_poisson): tests/test_gpu_encoding.py compares the two bit for bit, and both with
    the host path's distribution."""
    from ..ops import encode_poisson
    assert (datum >= 0).all(), "Inputs must be non-negative"
    if seed is None:
        _settle_sections()                                    # (one draw from the host generator: an open pipelined() section holds its state)
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
    steps = int(time / dt)
    return encode_poisson(datum.flatten(), steps, dt, seed, device).view(steps, *datum.shape)


def rank_order(datum: torch.Tensor, time: int, dt: float = 1.0, device="cpu", **kwargs) -> torch.Tensor:
    """One spike per non-zero feature, earlier for larger values (encodings.py:155-189).  Like the reference, the
    datum is normalised in place when it is contiguous."""
    assert (datum >= 0).all(), "Inputs must be non-negative"
    shape, size = datum.shape, datum.numel()
    flat = datum.flatten().to(device)
    steps = int(time / dt)
    flat /= flat.max()
    nz = flat != 0
    times = torch.zeros(size, device=flat.device)
    times[nz] = 1 / flat[nz]
    times *= steps / times.max()
    times = torch.ceil(times).long()
    spikes = torch.zeros(steps, size, device=device).byte()
    idx = torch.arange(size, device=times.device)
    ok = (times > 0) & (times < steps)
    spikes[times[ok] - 1, idx[ok]] = 1
    return spikes.reshape(steps, *shape)
