from .models import DiehlAndCook2015, IncreasingInhibitionNetwork, TwoLayerNetwork

__all__ = ["TwoLayerNetwork", "DiehlAndCook2015", "IncreasingInhibitionNetwork"]
