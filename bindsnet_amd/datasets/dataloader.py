"""`bindsnet.datasets.DataLoader` / `time_aware_collate` (reference: datasets/dataloader.py, datasets/collate.py): a
torch DataLoader whose batches are TIME-major -- every tensor field of a sample [time, n_0, ...] is stacked along a new
dimension 1, giving [time, batch, n_0, ...], the layout Network.run() takes (examples/mnist/batch_eth_mnist.py)."""
import collections.abc

import numpy as np
import torch


def time_aware_collate(batch):
    """Collate a list of samples field by field (collate.py:27-85).  Tensors: 0-d -> [1, batch], 1-d [time] -> [time, batch],
    otherwise [time, n_0, ...] -> [time, batch, n_0, ...]; numpy arrays are treated as tensors, numpy scalars / ints become
    a 1-d tensor, floats a float64 one; mappings, named tuples and sequences are collated member by member."""
    first = batch[0]
    if isinstance(first, torch.Tensor):
        if first.dim() == 0:
            batch = [x.view(1, 1) for x in batch]
        elif first.dim() == 1:
            batch = [x.view(x.shape[0], 1) for x in batch]
        return torch.stack(batch, 1)
    if isinstance(first, np.ndarray):
        if first.dtype.kind in "SUO":
            raise TypeError(f"time_aware_collate: arrays of strings / objects cannot be batched (dtype {first.dtype})")
        return time_aware_collate([torch.as_tensor(b) for b in batch])
    if isinstance(first, np.generic) and not isinstance(first, (np.str_, np.bytes_)):
        return torch.as_tensor(np.asarray(batch))
    if isinstance(first, float):
        return torch.tensor(batch, dtype=torch.float64)
    if isinstance(first, int):
        return torch.tensor(batch)
    if isinstance(first, collections.abc.Mapping):
        return {key: time_aware_collate([sample[key] for sample in batch]) for key in first}
    if isinstance(first, tuple) and hasattr(first, "_fields"):
        return type(first)(*(time_aware_collate(members) for members in zip(*batch)))
    if isinstance(first, collections.abc.Sequence) and not isinstance(first, (str, bytes)):
        return [time_aware_collate(members) for members in zip(*batch)]
    raise TypeError(f"time_aware_collate: cannot batch elements of type {type(first)}")


class DataLoader(torch.utils.data.DataLoader):
    """torch.utils.data.DataLoader with `time_aware_collate` as the default collate function (dataloader.py:6-33)."""

    def __init__(self, dataset, batch_size=1, shuffle=False, sampler=None, batch_sampler=None, num_workers=0,
                 collate_fn=time_aware_collate, pin_memory=False, drop_last=False, timeout=0, worker_init_fn=None):
        super().__init__(dataset, batch_size=batch_size, shuffle=shuffle, sampler=sampler, batch_sampler=batch_sampler,
                         num_workers=num_workers, collate_fn=collate_fn, pin_memory=pin_memory, drop_last=drop_last,
                         timeout=timeout, worker_init_fn=worker_init_fn)
