#!/usr/bin/env python3
"""Fixtures for the host-side plumbing that examples/mnist/eth_mnist.py needs around the hot path (build container only):
the reference's encoders (spike trains AND the state they leave the global CPU generator in), its evaluation read-outs
and its weight / assignment reshaping helpers, all on tests/synth.py inputs.

    python tests/golden/make_golden_host.py      -> tests/golden/op_encoding.npz, op_evaluation.npz, op_reshape_local.npz, op_reward.npz, op_collate.npz, op_models.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402
import make_golden as mg  # noqa: E402  (reference import recipe)
from make_golden import T_, save  # noqa: E402
from bindsnet.encoding import bernoulli, poisson, rank_order, repeat, single  # noqa: E402
from bindsnet.encoding import BernoulliEncoder, PoissonEncoder  # noqa: E402
from bindsnet.evaluation import all_activity, assign_labels, ngram, proportion_weighting, update_ngram_scores  # noqa: E402
from bindsnet.utils import get_square_assignments, get_square_weights, reshape_conv2d_weights  # noqa: E402

from make_golden_host_cases import ENC_CASES, collate_batch, datum_for, flatten_collated  # noqa: E402


def gen_encoding():
    fns = dict(poisson=poisson, bernoulli=bernoulli, rank_order=rank_order, single=single, repeat=repeat)
    out = {}
    for k, (name, shape, scale, time, dt, kw) in enumerate(ENC_CASES):
        torch.manual_seed(100 + k)
        x = T_(datum_for(k, shape, scale)).clone()
        y = fns[name](x, time=time, dt=dt, **kw)
        out[f"y{k}"] = np.packbits(y.numpy().astype(np.uint8)) if name != "repeat" else y.numpy()
        out[f"shape{k}"] = np.array(y.shape)
        out[f"dtype{k}"] = str(y.dtype)
        out[f"probe{k}"] = torch.rand(3).numpy()               # where the encoder left the generator
        out[f"x_after{k}"] = x.numpy().copy()                  # (bernoulli / rank_order normalise their argument in place)
    # encoder objects as dataset transforms
    torch.manual_seed(5)
    e = PoissonEncoder(time=30, dt=1.0)(T_(datum_for(0, (1, 28, 28), 128.0)))
    b = BernoulliEncoder(time=12, dt=1.0, max_prob=0.7)(T_(datum_for(3, (1, 28, 28), 1.0)))
    out.update(enc_poisson=np.packbits(e.numpy()), enc_bernoulli=np.packbits(b.numpy()), enc_probe=torch.rand(2).numpy())
    save("op_encoding", **out)


def gen_evaluation():
    rs = np.random.RandomState(9)
    n, T, N, L = 24, 30, 40, 10
    spikes = (rs.uniform(size=(n, T, N)) < 0.06).astype(np.float32)
    labels = rs.randint(0, L, size=n)
    a, p, r = assign_labels(T_(spikes), T_(labels), L)
    a2, p2, r2 = assign_labels(T_(spikes[:12]), T_(labels[:12]), L, rates=r.clone(), alpha=0.9)
    out = dict(assign=a.numpy(), prop=p.numpy(), rates=r.numpy(), assign2=a2.numpy(), prop2=p2.numpy(), rates2=r2.numpy(),
               all_act=all_activity(T_(spikes), a, L).numpy(), prop_w=proportion_weighting(T_(spikes), a, p, L).numpy())
    sparse = (rs.uniform(size=(6, 12, 8)) < 0.05).astype(np.float32)
    scores = update_ngram_scores(T_(sparse), T_(labels[:6]), L, 2, {})
    keys = sorted(scores)
    out.update(ngram_keys=np.array(keys), ngram_vals=np.array([scores[k].numpy() for k in keys]),
               ngram_pred=ngram(T_(sparse), scores, L, 2).numpy())
    W = synth.uniform_f32(77, (784, 90), 0.0, 1.0)
    out.update(sq_w=get_square_weights(T_(W), 10, 28).numpy(), sq_w_rect=get_square_weights(T_(W[:600]), 10, (20, 30)).numpy(),
               sq_a=get_square_assignments(T_(labels[:20].astype(np.float32)), 5).numpy(),
               conv=reshape_conv2d_weights(T_(synth.uniform_f32(78, (6, 3, 4, 5), 0.0, 1.0))).numpy())
    save("op_evaluation", **out)


def gen_reshape_local():
    """bindsnet.utils.reshape_locally_connected_weights (utils.py:112-180) on the receptive fields of real LocalConnections:
    3 x 3 positions of a 4 x 4 kernel with 5 filters, a non-square case, and the single-position case (kernel = input)."""
    from bindsnet.network.nodes import Input, LIFNodes
    from bindsnet.network.topology import LocalConnection
    from bindsnet.utils import reshape_locally_connected_weights
    out = {}
    for k, (shape, ks, st, nf) in enumerate((((12, 12), 4, 4, 5), ((10, 10), (4, 2), (3, 2), 3), ((6, 6), 6, 1, 7))):
        n_in = shape[0] * shape[1]
        src = Input(n=n_in, shape=(1, *shape))
        kp = (ks, ks) if np.isscalar(ks) else ks
        sp = (st, st) if np.isscalar(st) else st
        cs = (1, 1) if kp == shape else ((shape[0] - kp[0]) // sp[0] + 1, (shape[1] - kp[1]) // sp[1] + 1)
        tgt = LIFNodes(n=nf * cs[0] * cs[1])
        lc = LocalConnection(src, tgt, kernel_size=ks, stride=st, n_filters=nf, input_shape=shape)
        w = synth.uniform_f32(880 + k, tuple(lc.w.shape), 0.0, 1.0)
        out[f"loc{k}"] = lc.locations.numpy().copy()
        out[f"img{k}"] = reshape_locally_connected_weights(T_(w), nf, lc.kernel_size, lc.conv_size, lc.locations, shape).numpy()
        out[f"meta{k}"] = np.array([*shape, *lc.kernel_size, *lc.conv_size, nf, *w.shape])
        print(f"  reshape_locally_connected_weights case {k}: w {w.shape} conv {tuple(lc.conv_size)} -> image {out[f'img{k}'].shape}")
    save("op_reshape_local", **out)


def gen_collate():
    """bindsnet.datasets.collate.time_aware_collate (loaded from its file: importing bindsnet.datasets needs cv2)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_collate", "/root/reference/bindsnet/datasets/collate.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    flat = flatten_collated(ref.time_aware_collate(collate_batch()))
    out = {"paths": np.array(sorted(flat))}
    for i, k in enumerate(sorted(flat)):
        out[f"t{i}"] = flat[k].numpy()
    save("op_collate", **out)


IIN_CASES = [dict(n_input=64, n_neurons=25, inpt_shape=(1, 8, 8)),
             dict(n_input=30, n_neurons=40, start_inhib=2.5, max_inhib=33.0, theta_plus=0.1, exc_thresh=-50.0)]


def gen_models():
    """bindsnet.models.IncreasingInhibitionNetwork (models.py:349-454): construction only -- the input weights' draw, the
    distance-graded recurrent weights, and where the constructor leaves the global generator."""
    from bindsnet.models import IncreasingInhibitionNetwork
    out = {}
    for k, kw in enumerate(IIN_CASES):
        torch.manual_seed(4)
        net = IncreasingInhibitionNetwork(**kw)
        out[f"probe{k}"] = torch.rand(3).numpy()
        out[f"w_xy{k}"] = net.connections[("X", "Y")].w.detach().numpy().copy()
        out[f"w_yy{k}"] = net.connections[("Y", "Y")].w.detach().numpy().copy()
        Y = net.layers["Y"]
        out[f"consts{k}"] = np.array([float(Y.thresh), float(Y.rest), float(Y.reset), float(Y.refrac), float(Y.theta_plus),
                                      float(net.connections[("X", "Y")].norm), net.n_sqrt], np.float64)
    from bindsnet.models import DiehlAndCook2015v2, LocallyConnectedNetwork
    from make_golden_host_cases import MODEL_CASES
    for k, (name, kw) in enumerate(MODEL_CASES):
        torch.manual_seed(4)
        np.random.seed(7)                                   # (LocalConnection draws its weights from numpy, topology.py:1421)
        net = {"DiehlAndCook2015v2": DiehlAndCook2015v2, "LocallyConnectedNetwork": LocallyConnectedNetwork}[name](**kw)
        out[f"m{k}_probe"] = np.array([float(torch.rand(1)), np.random.rand()])
        out[f"m{k}_w_xy"] = net.connections[("X", "Y")].w.detach().numpy().copy()
        out[f"m{k}_w_yy"] = net.connections[("Y", "Y")].w.detach().numpy().copy()
        out[f"m{k}_n"] = np.array([net.layers["X"].n, net.layers["Y"].n])
    save("op_models", **out)


def reward_episodes():
    rs = np.random.RandomState(3)
    return [(float(rs.uniform(-3, 5)), int(rs.randint(5, 40)), [10.0, 4.0, 25.0][ep % 3]) for ep in range(12)]


def gen_reward():
    """bindsnet.learning.reward.MovingAvgRPE (reward.py:29-87) over 12 episodes: the prediction error handed to the rules
    before every episode and both moving averages after it."""
    from bindsnet.learning.reward import MovingAvgRPE
    r = MovingAvgRPE()
    rpe, per_step, per_episode = [], [], []
    for acc, steps, win in reward_episodes():
        rpe.append(r.compute(reward=torch.tensor(acc / steps)).numpy())
        r.update(accumulated_reward=acc, steps=steps, ema_window=win)
        per_step.append(r.reward_predict.numpy()); per_episode.append(r.reward_predict_episode.numpy())
    save("op_reward", rpe=np.array(rpe), per_step=np.array(per_step), per_episode=np.array(per_episode),
         history=np.array(r.rewards_predict_episode))


if __name__ == "__main__":
    torch.set_num_threads(1)
    jobs = sys.argv[1:] or ["encoding", "evaluation", "reshape_local", "reward", "collate", "models"]
    if "collate" in jobs:
        gen_collate()
    if "models" in jobs:
        gen_models()
    if "reward" in jobs:
        gen_reward()
    if "encoding" in jobs:
        gen_encoding()
    if "evaluation" in jobs:
        gen_evaluation()
    if "reshape_local" in jobs:
        gen_reshape_local()
