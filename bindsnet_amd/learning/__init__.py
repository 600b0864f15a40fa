from .learning import LearningRule, MSTDP, NoOp, PostPre

__all__ = ["LearningRule", "NoOp", "PostPre", "MSTDP"]
