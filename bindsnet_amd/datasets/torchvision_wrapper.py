"""Spike-encoding wrappers around torchvision datasets: API mirror of bindsnet/datasets/torchvision_wrapper.py.
`__getitem__` returns {"image", "label", "encoded_image", "encoded_label"} instead of (image, label)."""
from typing import Dict, Optional

import torch

from ..encoding import Encoder, NullEncoder


def create_torchvision_dataset_wrapper(ds_type):
    """ds_type: a torchvision dataset class, or its name inside torchvision.datasets."""
    if isinstance(ds_type, str):
        from torchvision import datasets as torch_db          # only needed once a wrapper is actually requested
        ds_type = getattr(torch_db, ds_type)

    class TorchvisionDatasetWrapper(ds_type):
        __doc__ = "BindsNET-style wrapper (dict items with encoded image / label) around:\n\n" + str(ds_type.__doc__ or ds_type)

        def __init__(self, image_encoder: Optional[Encoder] = None, label_encoder: Optional[Encoder] = None, *args, **kwargs):
            super().__init__(*args, **kwargs)
            self.args, self.kwargs = args, kwargs
            self.image_encoder = NullEncoder() if image_encoder is None else image_encoder
            self.label_encoder = NullEncoder() if label_encoder is None else label_encoder

        def __getitem__(self, ind: int) -> Dict[str, torch.Tensor]:
            image, label = super().__getitem__(ind)
            return {"image": image, "label": label, "encoded_image": self.image_encoder(image),
                    "encoded_label": self.label_encoder(label)}

        def __len__(self):
            return super().__len__()

    return TorchvisionDatasetWrapper
