from .models import DiehlAndCook2015, TwoLayerNetwork

__all__ = ["TwoLayerNetwork", "DiehlAndCook2015"]
