"""Host-RNG parity for DiehlAndCookNodes' one_spike arbitration.

The reference calls torch.multinomial(p, 1) on the *global CPU generator* once per timestep
that has a threshold crossing; on CPU that is exactly `argmax(p / q)` with
`q = torch.empty_like(p).exponential_(1)`, one draw per element, and consecutive calls
concatenate into one stream.  NoiseStream hands that stream to the device and, when the run
is over, leaves the global generator exactly where the reference would have left it
(state at entry advanced by the number of draws the device consumed), so everything
downstream that shares the generator (Poisson encoders, shuffling) sees the same numbers.
"""
import torch

# Pipelined networks (Network.pipelined()) keep the host generator's state ON THE DEVICE between their runs; whoever is about to read or
# replace the host generator inside this package (a synchronous run, a device encoder, a hand-stepped layer) settles them first.
_PENDING = []          # callables: Network.sync of every network with a pipelined section open


def flush_pending() -> None:
    for settle in list(_PENDING):
        settle()


class NoiseStream:
    """Pre-drawn Exp(1) stream + device cursor/status words.

    max_draws: upper bound of draws the run can consume (rows * N * T); 0 disables the stream.
    """

    def __init__(self, device, max_draws: int):
        self.device = torch.device(device)
        self.max_draws = int(max_draws)
        self.q = None
        self.cursor = torch.zeros(2, dtype=torch.int64, device=self.device)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._state0 = None
        self.consumed = 0

    def __enter__(self):
        flush_pending()
        if self.max_draws > 0:
            self._state0 = torch.get_rng_state()
            q = torch.empty(self.max_draws).exponential_(1)      # the reference's own stream
            self.q = q.to(self.device, non_blocking=False)
        return self

    def finish(self) -> int:
        """Synchronise, check the device status word, restore + advance the host generator."""
        from ._lib import SnnError
        if self.max_draws <= 0:
            return 0
        cur, st = int(self.cursor[0].item()), int(self.status.item())   # .item() synchronises
        torch.set_rng_state(self._state0)
        if st != 0:
            raise SnnError(f"one_spike noise stream exhausted (status {st}, consumed {cur} of {self.max_draws})")
        if cur:
            torch.empty(cur).exponential_(1)                   # advance by exactly `cur` draws
        self.consumed = cur
        return cur

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.finish()
        elif self._state0 is not None:
            torch.set_rng_state(self._state0)
        return False


# ---------------------------------------------------------------------------------------------
# Device generator: the same stream, produced on the MI355X (csrc/snn_rng.hpp), so a run costs no
# host draws at all.  Layout of torch.get_rng_state() for the CPU generator (5056 bytes,
# at::CPUGeneratorImplState): u64 seed | i32 left | i32 seeded | u64 next | u64 state[624] | normal-
# distribution cache.  at::mt19937 outputs state[next++] after `if (--left == 0) twist`, so
# left == 1 means "twist before the next output" (our pos == 624) and otherwise left == 625 - next.
# ---------------------------------------------------------------------------------------------
import struct

import numpy as np

_STATE_OFF, _N = 24, 624
RNG_STATE_BYTES = _N * 4 + 4 + 4 + 8          # sizeof(snn_rng_state)


def torch_state_to_words(state: torch.Tensor) -> np.ndarray:
    """torch CPU generator state -> int32[630] image of snn_rng_state (mt[624], pos, reserved, consumed)."""
    raw = state.numpy().tobytes()
    if len(raw) != 5056:
        raise RuntimeError(f"unexpected CPU generator state size {len(raw)} (torch {torch.__version__})")
    _seed, left, _seeded, nxt = struct.unpack_from("<QiiQ", raw, 0)
    mt = np.frombuffer(raw, dtype=np.uint64, count=_N, offset=_STATE_OFF).astype(np.uint32)
    pos = _N if left == 1 else int(nxt)
    if left != 1 and left != _N + 1 - nxt:
        raise RuntimeError("inconsistent mt19937 state (left/next)")
    img = np.zeros(RNG_STATE_BYTES // 4, dtype=np.uint32)
    img[:_N] = mt
    img[_N] = pos
    return img.view(np.int32)


def words_to_torch_state(img: np.ndarray, template: torch.Tensor) -> torch.Tensor:
    """Inverse of torch_state_to_words: patch mt / left / next into a copy of `template` (numpy views on one 5056-byte buffer:
    this runs once per synchronous run(), and the bytearray / struct version of it cost 32 us)."""
    img = np.ascontiguousarray(img).view(np.uint32)
    raw = template.numpy().copy()
    if raw.size != 5056:
        raise RuntimeError(f"unexpected CPU generator state size {raw.size} (torch {torch.__version__})")
    pos = int(img[_N])
    raw[8:12].view(np.int32)[0] = _N + 1 - pos             # left
    raw[16:24].view(np.uint64)[0] = pos                    # next
    raw[_STATE_OFF:_STATE_OFF + _N * 8].view(np.uint64)[:] = img[:_N]
    return torch.from_numpy(raw)


_BLOCK_WORDS = 640          # int32 words of the run's host<->device block: state[628] | status | pad[3] | cursor (i64[2]) | pad
_STATUS_AT, _CURSOR_AT = 628, 632


class DeviceGenerator:
    """Context manager: upload the host generator at entry, write it back (advanced by what the
    device consumed) at exit.  `consumed` is the number of Exp(1) draws used.

    State image, status word and cursor live in ONE device block, so a run costs one host->device copy at entry
    (which also zeroes status and cursor) and one blocking device->host copy at exit."""

    def __init__(self, device, qbuf_floats: int, buffers=None):
        """buffers: optional persistent (block int32[640], qbuf f32[n], pinned host int32[640]) tensors."""
        self.device = torch.device(device)
        self.enabled = qbuf_floats > 0
        block = buffers[0] if buffers is not None else torch.zeros(_BLOCK_WORDS, dtype=torch.int32, device=self.device)
        self._block = block
        self.state = block[:RNG_STATE_BYTES // 4]
        self.status = block[_STATUS_AT:_STATUS_AT + 1]
        self.cursor = block[_CURSOR_AT:_CURSOR_AT + 4].view(torch.int64)
        self.qbuf = buffers[1] if buffers is not None else None
        # page-locked staging block: the upload is an asynchronous copy and the read-back lands without a bounce
        # buffer (a pageable copy costs a blocking driver round trip each way, every run)
        self._host = buffers[2] if buffers is not None and len(buffers) > 2 else torch.zeros(_BLOCK_WORDS, dtype=torch.int32).pin_memory()
        self.consumed = 0
        self._n = qbuf_floats
        self.always_read = False       # set by Network.run for plans that report through the status word

    def __enter__(self):
        flush_pending()
        img = self._host.numpy()
        img[:] = 0                                         # status = 0, cursor = 0
        if self.enabled:
            self._host0 = torch.get_rng_state()
            img[:RNG_STATE_BYTES // 4] = torch_state_to_words(self._host0)
            if self.qbuf is None:
                self.qbuf = torch.empty(self._n, dtype=torch.float32, device=self.device)
        self._block.copy_(self._host, non_blocking=True)   # (the previous run's blocking read-back ordered us behind it)
        return self

    def finish(self, check_status: bool = True) -> int:
        """Synchronise with the run, read the block back, leave the host generator where the reference would.
        Returns the device status word: 0, or -- when `check_status` is False -- SNN_ERR_TIMEOUT (a resident run gave
        up on a workgroup hand-off) / SNN_ERR_RETRY (the lean kernel form met a step it does not handle).  The caller
        then repeats the run on another plan; the device left every state tensor and the generator state untouched
        in that case, so the host generator is restored to its entry state."""
        from ._lib import SNN_ERR_RETRY, SNN_ERR_TIMEOUT, SnnError
        if not self.enabled and not check_status and not self.always_read:
            return 0
        self._host.copy_(self._block)                      # blocking: synchronises with the run
        blk = self._host.numpy()
        st = int(blk[_STATUS_AT])
        if st != 0:
            if self.enabled:
                torch.set_rng_state(self._host0)
            if st in (SNN_ERR_TIMEOUT, SNN_ERR_RETRY) and not check_status:
                return st
            raise SnnError(f"device run reported status {st}")
        if self.enabled:
            img = blk[:RNG_STATE_BYTES // 4]
            self.consumed = int(np.ascontiguousarray(img).view(np.int64)[(RNG_STATE_BYTES - 8) // 8])
            torch.set_rng_state(words_to_torch_state(img, self._host0))
        return 0

    def __exit__(self, exc_type, exc, tb):
        if exc_type is not None and self.enabled:
            torch.set_rng_state(self._host0)
        return False
