"""Case table shared by tests/golden/make_golden_host.py (which needs the reference) and tests/test_host_plumbing.py."""
import numpy as np
import torch

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


def collate_batch():
    """Four samples with every kind of field time_aware_collate distinguishes (collate.py:27-85)."""
    import collections
    P = collections.namedtuple("P", "a b")
    rs = np.random.RandomState(1)

    def sample(k):
        return {"encoded_image": torch.from_numpy((rs.rand(20, 1, 6, 6) < 0.1).astype(np.uint8)), "image": torch.from_numpy(rs.rand(1, 6, 6).astype(np.float32)),
                "label": torch.tensor(k), "encoded_label": k, "vec": torch.arange(5.0) + k, "np": rs.rand(3, 2).astype(np.float32),
                "f": 0.5 * k, "npi": np.int64(k), "tup": P(torch.ones(2) * k, k), "lst": [k, torch.zeros(3)]}
    return [sample(k) for k in range(4)]


def flatten_collated(x, path="", out=None):
    """{path: tensor} of a collated structure, the container types recorded in the path."""
    out = {} if out is None else out
    if isinstance(x, torch.Tensor):
        out[path] = x
    elif isinstance(x, dict):
        for k in x:
            flatten_collated(x[k], f"{path}/{k}", out)
    else:
        for i, v in enumerate(x):
            flatten_collated(v, f"{path}<{type(x).__name__}>[{i}]", out)
    return out


MODEL_CASES = [("DiehlAndCook2015v2", dict(n_inpt=64, n_neurons=30, inh=20.0, inpt_shape=(1, 8, 8))),
               ("LocallyConnectedNetwork", dict(n_inpt=144, input_shape=(12, 12), kernel_size=4, stride=4, n_filters=5)),
               ("LocallyConnectedNetwork", dict(n_inpt=100, input_shape=(10, 10), kernel_size=(4, 2), stride=(3, 2), n_filters=3, inh=12.5, norm=0.4)),
               ("LocallyConnectedNetwork", dict(n_inpt=36, input_shape=(6, 6), kernel_size=6, stride=1, n_filters=4))]
