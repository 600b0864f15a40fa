"""Lazy generators over a dataset tensor / iterable: API mirror of bindsnet/encoding/loaders.py."""
from typing import Iterable, Iterator, Optional, Union

import torch

from .encodings import bernoulli, poisson, rank_order


def _loader(enc, data, time, dt, kwargs) -> Iterator[torch.Tensor]:
    for i in range(len(data)):
        yield enc(datum=data[i], time=time, dt=dt, **kwargs)


def bernoulli_loader(data: Union[torch.Tensor, Iterable[torch.Tensor]], time: Optional[int] = None, dt: float = 1.0,
                     **kwargs) -> Iterator[torch.Tensor]:
    """loaders.py:8-33 (keyword `max_prob` is forwarded)."""
    return _loader(bernoulli, data, time, dt, {"max_prob": kwargs.get("max_prob", 1.0)})


def poisson_loader(data: Union[torch.Tensor, Iterable[torch.Tensor]], time: int, dt: float = 1.0,
                   **kwargs) -> Iterator[torch.Tensor]:
    """loaders.py:36-54."""
    return _loader(poisson, data, time, dt, {})


def rank_order_loader(data: Union[torch.Tensor, Iterable[torch.Tensor]], time: int, dt: float = 1.0,
                      **kwargs) -> Iterator[torch.Tensor]:
    """loaders.py:57-75."""
    return _loader(rank_order, data, time, dt, {})
