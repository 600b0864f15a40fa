"""Reward signals for the reward-modulated rules (MSTDP / MSTDPET): API mirror of bindsnet/learning/reward.py.
Host-side bookkeeping only -- `Network(reward_fn=MovingAvgRPE)` calls `compute()` once per `run()` (network.py:318-320)
and hands the result to the rules as their `reward` keyword; `update()` is called by the caller once per episode."""
from abc import ABC, abstractmethod

import torch


class AbstractReward(ABC):
    """Interface of a reward function (reward.py:6-26)."""

    @abstractmethod
    def compute(self, **kwargs):
        """The (possibly modified) reward for this run."""

    @abstractmethod
    def update(self, **kwargs) -> None:
        """Advance whatever the modification depends on; usually once per episode."""


class MovingAvgRPE(AbstractReward):
    """Reward prediction error against exponential moving averages of past rewards (reward.py:29-87)."""

    def __init__(self, **kwargs) -> None:
        self.reward_predict = torch.tensor(0.0)            # predicted reward per step
        self.reward_predict_episode = torch.tensor(0.0)    # predicted reward per episode
        self.rewards_predict_episode = []                  # its history, one entry per update()

    def compute(self, **kwargs) -> torch.Tensor:
        """reward - (predicted reward per step)."""
        return kwargs["reward"] - self.reward_predict

    def update(self, **kwargs) -> None:
        """Keyword arguments: `accumulated_reward` of the episode, its number of `steps`, `ema_window` (default 10)."""
        total = kwargs["accumulated_reward"]
        steps = torch.tensor(kwargs["steps"]).float()
        window = torch.tensor(kwargs.get("ema_window", 10.0))
        keep, gain = 1 - 1 / window, 1 / window
        self.reward_predict = keep * self.reward_predict + gain * (total / steps)
        self.reward_predict_episode = keep * self.reward_predict_episode + gain * total
        self.rewards_predict_episode.append(self.reward_predict_episode.item())
