"""Deterministic synthetic inputs shared by the golden-fixture generator and the tests.

Everything here is numpy ``RandomState`` (frozen stream guarantee), so a fixture generated in
the build container can be regenerated bit-identically on the GPU box without storing it.
Sizes/statistics follow SURVEY.md §8(d): MNIST-like images with ~19 % active pixels whose
Poisson rates give ~1-2 % spike density per timestep.
"""
import numpy as np


def weights_q12(seed: int, n_in: int, n_out: int, scale: float = 0.3) -> np.ndarray:
    """[n_in, n_out] f32 in [0, scale) on a 2^-12 grid (exactly representable, arbitrary)."""
    rs = np.random.RandomState(seed)
    k = rs.randint(0, int(scale * 4096), size=(n_in, n_out))
    return (k.astype(np.float32) / np.float32(4096.0)).astype(np.float32)


def spike_train(seed: int, T: int, B: int, n: int, active: float = 0.19, max_rate: float = 0.0625) -> np.ndarray:
    """u8 [T, B, n] Bernoulli spikes: each sample has `active` fraction of pixels with a
    per-pixel rate U(0, max_rate); others silent (mean density ~0.6 %..1.5 %)."""
    rs = np.random.RandomState(seed)
    rate = rs.uniform(0.0, max_rate, size=(B, n)) * (rs.uniform(size=(B, n)) < active)
    u = rs.uniform(size=(T, B, n))
    return (u < rate[None]).astype(np.uint8)


def dense_spikes(seed: int, shape, p: float) -> np.ndarray:
    rs = np.random.RandomState(seed)
    return (rs.uniform(size=shape) < p).astype(np.uint8)


def uniform_f32(seed: int, shape, lo: float, hi: float) -> np.ndarray:
    rs = np.random.RandomState(seed)
    return rs.uniform(lo, hi, size=shape).astype(np.float32)
