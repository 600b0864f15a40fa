from .models import (DiehlAndCook2015, DiehlAndCook2015v2, IncreasingInhibitionNetwork, LocallyConnectedNetwork,
                     TwoLayerNetwork)

__all__ = ["TwoLayerNetwork", "DiehlAndCook2015", "DiehlAndCook2015v2", "IncreasingInhibitionNetwork", "LocallyConnectedNetwork"]
