"""A stand-in for `torchvision` (absent from this image, and there is no network for MNIST anyway): just enough of
`torchvision.datasets` / `torchvision.transforms` for examples/mnist/eth_mnist.py -- a deterministic synthetic
"MNIST" (class-dependent 28x28 blobs, numpy RandomState) and the three transforms the script composes.
Test infrastructure: installed into sys.modules by the eth_mnist tests and by the fixture generator only."""
import sys
import types

import numpy as np
import torch

N_TRAIN, N_TEST = 64, 32


class SyntheticMNIST(torch.utils.data.Dataset):
    """(image, label) items like torchvision.datasets.MNIST: image = HxW uint8 array run through `transform`."""

    def __init__(self, root=None, train=True, transform=None, target_transform=None, download=False):
        rs = np.random.RandomState(20240 + (0 if train else 1))
        proto = np.random.RandomState(777).uniform(0, 255, size=(10, 28, 28)) * (np.random.RandomState(778).uniform(size=(10, 28, 28)) < 0.19)
        n = N_TRAIN if train else N_TEST
        self.targets = rs.randint(0, 10, size=n)
        noise = rs.uniform(0, 40, size=(n, 28, 28)) * (rs.uniform(size=(n, 28, 28)) < 0.05)
        self.data = np.clip(proto[self.targets] * rs.uniform(0.6, 1.0, size=(n, 1, 1)) + noise, 0, 255).astype(np.uint8)
        self.transform, self.target_transform = transform, target_transform

    def __getitem__(self, index):
        img, target = self.data[index], int(self.targets[index])
        if self.transform is not None:
            img = self.transform(img)
        if self.target_transform is not None:
            target = self.target_transform(target)
        return img, target

    def __len__(self):
        return len(self.data)


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class ToTensor:
    """HxW uint8 -> float32 [1, H, W] in [0, 1] (torchvision semantics for a single-channel image)."""

    def __call__(self, pic):
        return torch.from_numpy(np.ascontiguousarray(pic)).unsqueeze(0).to(torch.float32).div(255)


class Lambda:
    def __init__(self, lambd):
        self.lambd = lambd

    def __call__(self, x):
        return self.lambd(x)


class _Unavailable(torch.utils.data.Dataset):
    def __init__(self, *a, **k):
        raise RuntimeError("dataset not available in the torchvision stand-in")


def install():
    """Register the stand-in as `torchvision` (no-op when a torchvision module is already imported)."""
    if "torchvision" in sys.modules:
        return sys.modules["torchvision"]
    tv = types.ModuleType("torchvision")
    ds = types.ModuleType("torchvision.datasets")
    tf = types.ModuleType("torchvision.transforms")
    ds.MNIST = SyntheticMNIST
    for name in ("CIFAR10", "CIFAR100", "Cityscapes", "CocoCaptions", "CocoDetection", "DatasetFolder", "EMNIST", "FakeData",
                 "FashionMNIST", "Flickr30k", "Flickr8k", "ImageFolder", "KMNIST", "LSUN", "LSUNClass", "Omniglot", "PhotoTour",
                 "SBU", "SEMEION", "STL10", "SVHN", "VOCDetection", "VOCSegmentation"):
        setattr(ds, name, type(name, (_Unavailable,), {}))
    tf.Compose, tf.ToTensor, tf.Lambda = Compose, ToTensor, Lambda
    tv.datasets, tv.transforms = ds, tf
    tv.__version__ = "0.0-standin"
    sys.modules.update({"torchvision": tv, "torchvision.datasets": ds, "torchvision.transforms": tf})
    return tv
