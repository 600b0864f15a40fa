from .encodings import bernoulli, poisson, poisson_device, rank_order, repeat, single
from .loaders import bernoulli_loader, poisson_loader, rank_order_loader
from .encoders import (BernoulliEncoder, Encoder, NullEncoder, PoissonEncoder, RankOrderEncoder, RepeatEncoder,
                       SingleEncoder)

__all__ = ["encodings", "single", "repeat", "bernoulli", "poisson", "poisson_device", "rank_order", "loaders",
           "bernoulli_loader", "poisson_loader", "rank_order_loader", "encoders", "Encoder", "NullEncoder",
           "SingleEncoder", "RepeatEncoder", "BernoulliEncoder", "PoissonEncoder", "RankOrderEncoder"]
