from .network import Network, load
from . import monitors, nodes, topology, topology_features

__all__ = ["Network", "load", "nodes", "topology", "topology_features", "monitors"]
