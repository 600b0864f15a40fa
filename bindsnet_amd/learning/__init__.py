from .learning import Hebbian, LearningRule, MSTDP, MSTDPET, NoOp, PostPre, WeightDependentPostPre

__all__ = ["LearningRule", "NoOp", "PostPre", "WeightDependentPostPre", "Hebbian", "MSTDP", "MSTDPET"]
