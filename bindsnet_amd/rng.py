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
