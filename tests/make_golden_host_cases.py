"""Case table shared by tests/golden/make_golden_host.py (which needs the reference) and tests/test_host_plumbing.py."""
import numpy as np

import synth

ENC_CASES = [  # (name, shape, scale, time, dt, kwargs)
    ("poisson", (1, 28, 28), 128.0, 250, 1.0, {}),
    ("poisson", (3, 5), 40.0, 60, 0.5, {}),
    ("poisson", (16,), 300.0, 30, 1.0, {"approx": True}),
    ("bernoulli", (1, 28, 28), 1.0, 40, 1.0, {}),
    ("bernoulli", (6, 7), 3.0, 25, 1.0, {"max_prob": 0.5}),
    ("bernoulli", (10,), 1.0, None, 1.0, {}),
    ("rank_order", (4, 9), 2.0, 50, 1.0, {}),
    ("single", (5, 5), 1.0, 20, 1.0, {"sparsity": 0.3}),
    ("repeat", (2, 3), 1.0, 7, 1.0, {}),
]


def datum_for(k, shape, scale):
    x = synth.uniform_f32(4000 + k, shape, 0.0, scale) * (synth.uniform_f32(4100 + k, shape, 0.0, 1.0) < 0.6)
    return x.astype(np.float32)
