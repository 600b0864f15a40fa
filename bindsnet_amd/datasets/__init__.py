"""`bindsnet.datasets`: the torchvision wrappers (MNIST, CIFAR10, ...) are created on first attribute access, so
importing this package does not need torchvision.  `DataLoader` / `time_aware_collate` give the time-major batches
examples/mnist/batch_eth_mnist.py feeds to Network.run().  The reference's own corpora (SpokenMNIST, Davis, ALOV300) are
host-side data plumbing outside the accelerated path and are not provided."""
from .dataloader import DataLoader, time_aware_collate
from .torchvision_wrapper import create_torchvision_dataset_wrapper

_TORCHVISION = ("CIFAR10", "CIFAR100", "Cityscapes", "CocoCaptions", "CocoDetection", "DatasetFolder", "EMNIST",
                "FakeData", "FashionMNIST", "Flickr30k", "Flickr8k", "ImageFolder", "KMNIST", "LSUN", "LSUNClass",
                "MNIST", "Omniglot", "PhotoTour", "SBU", "SEMEION", "STL10", "SVHN", "VOCDetection", "VOCSegmentation")
__all__ = ["create_torchvision_dataset_wrapper", "DataLoader", "time_aware_collate", *_TORCHVISION]
_made = {}


def __getattr__(name):
    if name in _TORCHVISION:
        if name not in _made:
            _made[name] = create_torchvision_dataset_wrapper(name)
        return _made[name]
    raise AttributeError(f"module 'bindsnet.datasets' has no attribute {name!r}")
