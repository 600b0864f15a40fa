"""Callable encoder objects: API mirror of bindsnet/encoding/encoders.py (used as dataset transforms)."""
from . import encodings


class Encoder:
    """Stores the encoding's arguments; calling it applies `self.enc` to a datum (encoders.py:4-18)."""

    def __init__(self, *args, **kwargs) -> None:
        self.enc_args = args
        self.enc_kwargs = kwargs

    def __call__(self, img):
        return self.enc(img, *self.enc_args, **self.enc_kwargs)


class NullEncoder(Encoder):
    """Pass-through (encoders.py:21-35)."""

    def __init__(self):
        super().__init__()

    def __call__(self, img):
        return img


def _make(name, fn, **defaults):
    def __init__(self, time: int, dt: float = 1.0, **kwargs):
        merged = {**defaults, **kwargs}
        Encoder.__init__(self, time, dt=dt, **merged)
        self.enc = fn
    return type(name, (Encoder,), {"__init__": __init__, "__doc__": f"`bindsnet.encoding.{fn.__name__}` as a callable."})


SingleEncoder = _make("SingleEncoder", encodings.single, sparsity=0.5)
RepeatEncoder = _make("RepeatEncoder", encodings.repeat)
BernoulliEncoder = _make("BernoulliEncoder", encodings.bernoulli)
PoissonEncoder = _make("PoissonEncoder", encodings.poisson, approx=False)
RankOrderEncoder = _make("RankOrderEncoder", encodings.rank_order)
