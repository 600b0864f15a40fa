"""Deterministic synthetic inputs shared by bench.py, tools/, the golden-fixture generators and the tests.

Two families:

* numpy ``RandomState`` generators (frozen stream guarantee): a fixture generated in the build container is
  regenerated bit-identically on the GPU box without storing it.
* ``poisson_mnist_like``: the input BASELINE.md section 2 / SURVEY.md 8(d) STATE for cfg1 / cfg2 --
  ``img[b] = 128 * U(0,1) * Bernoulli(0.19)`` per sample, encoded by ``poisson(img[b], time=250, dt=1.0)``
  (bindsnet/encoding/encodings.py:101-152) from torch's global CPU generator after ``torch.manual_seed(1)``.
  `encoder` is the reference's function in the fixture generator and this package's host-stream-exact mirror
  everywhere else; the fixtures keep the reference-made trains (bit-packed) and their sha256, so the mirror is
  checked against them wherever the trains are regenerated.
"""
import hashlib

import numpy as np


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


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


def stroke_digit(seed: int, size: int = 28) -> np.ndarray:
    """A digit-like image f32 [size, size] in [0, 1]: three to five thick pen strokes (a random polyline through the middle 20 x 20
    pixels, pen radius 1.1-1.7 px with a soft edge) -- the statistics of a real MNIST digit that matter to the kernels: 100-200 lit
    pixels (MNIST: ~150 of 784), most of them SATURATED (MNIST strokes are mostly 255), in connected runs rather than scattered."""
    rs = np.random.RandomState(seed)
    n = int(rs.randint(3, 6))
    pts = rs.uniform(4.0, size - 4.0, size=(n + 1, 2))
    rad = float(rs.uniform(1.1, 1.7))
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    d2 = np.full((size, size), 1e9)
    for a, b in zip(pts[:-1], pts[1:]):
        ab = b - a
        t = np.clip(((xx - a[0]) * ab[0] + (yy - a[1]) * ab[1]) / max(float(ab @ ab), 1e-9), 0.0, 1.0)
        d2 = np.minimum(d2, (xx - (a[0] + t * ab[0])) ** 2 + (yy - (a[1] + t * ab[1])) ** 2)
    img = np.clip(1.0 - (np.sqrt(d2) - rad) / 0.9, 0.0, 1.0)
    return img.astype(np.float32)


def poisson_mnist_like(B: int, T: int = 250, n_inputs: int = 1, seed: int = 1, encoder=None, intensity: float = 128.0,
                       active: float = 0.19, shape=(1, 28, 28), bold=(), strokes: bool = False):
    """`n_inputs` spike trains u8 [T, B, *shape] of BASELINE.md's cfg1 / cfg2 generator, drawn consecutively from
    torch's global CPU generator seeded with `seed`.  Inputs whose index is in `bold` use intensity 255 on 45 % of
    the pixels for their odd samples (thick, saturated digits: > 32 events per sample and timestep), the
    worst-case input of tests/golden/full_cfg2_dc_n400_b32_bold.  `strokes`: digit-like images (stroke_digit) instead of scattered pixels
    -- tests/golden/full_cfg2_dc_n400_b32_strokes (the torch draws u, m are still made, so the generator position per sample is the same)."""
    import torch
    if encoder is None:
        from bindsnet_amd.encoding import poisson as encoder
    torch.manual_seed(seed)
    out = []
    for k in range(n_inputs):
        trains = []
        for b in range(B):
            u = torch.rand(*shape)
            m = torch.rand(*shape)
            if strokes:                         # digit-like images at eth_mnist.py's intensity (transforms.Lambda(x * 128), :117)
                img = intensity * torch.from_numpy(stroke_digit(7919 * seed + 131 * k + b, shape[-1])).view(*shape)
            elif k in bold and b % 2 == 1:
                img = 255.0 * (0.5 + 0.5 * u) * (m < 0.45).float()
            else:
                img = intensity * u * (m < active).float()
            trains.append(encoder(img, time=T, dt=1.0))
        out.append(np.ascontiguousarray(torch.stack(trains, dim=1).numpy().astype(np.uint8)))
    return out


# sha256 of the first three cfg2 batches poisson_mnist_like(32, 250, 3, seed=1) gives when `encoder` is the REFERENCE's
# bindsnet.encoding.poisson (tests/golden/full_cfg2_dc_n400_b32_poisson.npz r{0,1,2}_in_sha; pinned by
# tests/test_oracle_fullsize.py): bench.py reports whether the pool it timed is that input.
POISSON_CFG2_SHA = [
    "25e9f7ef5aae28fcad4cd3d76ee4cb0ea75474862e47937739e14949c8becf15",
    "cd5c019762b26b7b0117d54ba0b1f5ea99c870a2693d37672021317f6281dc42",
    "c29e90fcce2742c011db88455931ce21a499af4569c4de6f9d68416faf103a31",
]
