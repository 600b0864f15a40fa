"""Host-side plumbing around the hot path (what examples/mnist/eth_mnist.py imports besides the network classes) against
fixtures from the unmodified reference (tests/golden/make_golden_host.py): encoders -- spike trains bit for bit AND the
state they leave the global CPU generator in --, evaluation read-outs, reshaping helpers, the `bindsnet` import alias and
the dataset wrapper.  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

import cases
import synth
from cases import gold

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_encodings_match_reference_stream_for_stream():
    from make_golden_host_cases import ENC_CASES, datum_for
    from bindsnet_amd.encoding import bernoulli, poisson, rank_order, repeat, single
    fns = dict(poisson=poisson, bernoulli=bernoulli, rank_order=rank_order, single=single, repeat=repeat)
    g = gold("op_encoding")
    for k, (name, shape, scale, time, dt, kw) in enumerate(ENC_CASES):
        torch.manual_seed(100 + k)
        x = T_(datum_for(k, shape, scale)).clone()
        y = fns[name](x, time=time, dt=dt, **kw)
        assert list(y.shape) == list(g[f"shape{k}"]) and str(y.dtype) == str(g[f"dtype{k}"]), (name, k)
        got = np.packbits(y.numpy().astype(np.uint8)) if name != "repeat" else y.numpy()
        np.testing.assert_array_equal(got, g[f"y{k}"], err_msg=f"{name} case {k}")
        np.testing.assert_array_equal(torch.rand(3).numpy(), g[f"probe{k}"], err_msg=f"{name} case {k}: generator position")
        np.testing.assert_array_equal(x.numpy(), g[f"x_after{k}"], err_msg=f"{name} case {k}: in-place normalisation")


def test_encoders_cap_their_thread_count_and_put_it_back():
    """eth_mnist.py:77 asks for cpu_count() - 1 intra-op threads (255 on the GPU box), with which one poisson() call took
    1.6 s; the host encoders run with a handful of threads for their own duration.  Same fixture, same generator position,
    and the caller's setting is back afterwards."""
    from make_golden_host_cases import ENC_CASES, datum_for
    from bindsnet_amd.encoding import bernoulli, encodings, poisson
    g = gold("op_encoding")
    n0 = torch.get_num_threads()
    seen = []
    real = torch.poisson
    try:
        torch.set_num_threads(32)
        torch.poisson = lambda *a, **k: (seen.append(torch.get_num_threads()), real(*a, **k))[1]
        for k, (name, shape, scale, time, dt, kw) in enumerate(ENC_CASES):
            if name not in ("poisson", "bernoulli"):
                continue
            torch.manual_seed(100 + k)
            y = dict(poisson=poisson, bernoulli=bernoulli)[name](T_(datum_for(k, shape, scale)).clone(), time=time, dt=dt, **kw)
            np.testing.assert_array_equal(np.packbits(y.numpy().astype(np.uint8)), g[f"y{k}"], err_msg=f"{name} case {k}")
            np.testing.assert_array_equal(torch.rand(3).numpy(), g[f"probe{k}"], err_msg=f"{name} case {k}: generator position")
            assert torch.get_num_threads() == 32
    finally:
        torch.poisson = real
        torch.set_num_threads(n0)
    assert seen and all(n == encodings._ENCODER_THREADS for n in seen)


def test_encoder_objects():
    from make_golden_host_cases import datum_for
    from bindsnet_amd.encoding import BernoulliEncoder, NullEncoder, PoissonEncoder
    g = gold("op_encoding")
    torch.manual_seed(5)
    e = PoissonEncoder(time=30, dt=1.0)(T_(datum_for(0, (1, 28, 28), 128.0)))
    b = BernoulliEncoder(time=12, dt=1.0, max_prob=0.7)(T_(datum_for(3, (1, 28, 28), 1.0)))
    np.testing.assert_array_equal(np.packbits(e.numpy()), g["enc_poisson"])
    np.testing.assert_array_equal(np.packbits(b.numpy()), g["enc_bernoulli"])
    np.testing.assert_array_equal(torch.rand(2).numpy(), g["enc_probe"])
    assert NullEncoder()(7) == 7


def test_evaluation_and_reshaping_match_reference():
    from bindsnet_amd.evaluation import all_activity, assign_labels, ngram, proportion_weighting, update_ngram_scores
    from bindsnet_amd.utils import get_square_assignments, get_square_weights, reshape_conv2d_weights
    g = gold("op_evaluation")
    rs = np.random.RandomState(9)
    n, T, N, L = 24, 30, 40, 10
    spikes = (rs.uniform(size=(n, T, N)) < 0.06).astype(np.float32)
    labels = rs.randint(0, L, size=n)
    a, p, r = assign_labels(T_(spikes), T_(labels), L)
    for got, key in ((a, "assign"), (p, "prop"), (r, "rates")):
        np.testing.assert_array_equal(got.numpy(), g[key], err_msg=key)
    a2, p2, r2 = assign_labels(T_(spikes[:12]), T_(labels[:12]), L, rates=r.clone(), alpha=0.9)
    for got, key in ((a2, "assign2"), (p2, "prop2"), (r2, "rates2")):
        np.testing.assert_array_equal(got.numpy(), g[key], err_msg=key)
    np.testing.assert_array_equal(all_activity(T_(spikes), a, L).numpy(), g["all_act"])
    np.testing.assert_array_equal(proportion_weighting(T_(spikes), a, p, L).numpy(), g["prop_w"])
    sparse = (rs.uniform(size=(6, 12, 8)) < 0.05).astype(np.float32)
    scores = update_ngram_scores(T_(sparse), T_(labels[:6]), L, 2, {})
    keys = sorted(scores)
    np.testing.assert_array_equal(np.array(keys), g["ngram_keys"])
    np.testing.assert_array_equal(np.array([scores[k].numpy() for k in keys]), g["ngram_vals"])
    np.testing.assert_array_equal(ngram(T_(sparse), scores, L, 2).numpy(), g["ngram_pred"])
    W = synth.uniform_f32(77, (784, 90), 0.0, 1.0)
    np.testing.assert_array_equal(get_square_weights(T_(W), 10, 28).numpy(), g["sq_w"])
    np.testing.assert_array_equal(get_square_weights(T_(W[:600]), 10, (20, 30)).numpy(), g["sq_w_rect"])
    np.testing.assert_array_equal(get_square_assignments(T_(labels[:20].astype(np.float32)), 5).numpy(), g["sq_a"])
    np.testing.assert_array_equal(reshape_conv2d_weights(T_(synth.uniform_f32(78, (6, 3, 4, 5), 0.0, 1.0))).numpy(), g["conv"])


def test_bindsnet_alias_resolves_to_the_same_module_objects():
    import bindsnet
    import bindsnet.network.monitors as m_alias
    import bindsnet_amd.network.monitors as m_real
    from bindsnet.learning.MCC_learning import PostPre as A
    from bindsnet_amd.learning.MCC_learning import PostPre as B
    assert m_alias is m_real and A is B
    from bindsnet.models import DiehlAndCook2015
    from bindsnet.network import Network
    assert issubclass(DiehlAndCook2015, Network)
    assert bindsnet.encoding.PoissonEncoder is bindsnet.encoding.encoders.PoissonEncoder
    with pytest.raises(ImportError):
        import bindsnet.pipeline  # noqa: F401  (control plane outside the hot path: not provided)


def test_dataset_wrapper_and_plotting_with_the_torchvision_stand_in():
    os.environ["MPLBACKEND"] = "Agg"
    import matplotlib
    matplotlib.use("Agg", force=True)
    import tv_shim
    tv_shim.install()
    from torchvision import transforms
    from bindsnet.analysis.plotting import (plot_assignments, plot_input, plot_performance, plot_spikes, plot_voltages,
                                            plot_weights)
    from bindsnet.datasets import MNIST
    from bindsnet.encoding import PoissonEncoder
    ds = MNIST(PoissonEncoder(time=40, dt=1.0), None, root="x", download=True, train=True,
               transform=transforms.Compose([transforms.ToTensor(), transforms.Lambda(lambda x: x * 128)]))
    torch.manual_seed(0)
    item = ds[3]
    assert set(item) == {"image", "label", "encoded_image", "encoded_label"}
    assert tuple(item["encoded_image"].shape) == (40, 1, 28, 28) and item["encoded_image"].dtype == torch.uint8
    assert item["encoded_label"] == item["label"]
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=1, shuffle=True)))
    assert tuple(batch["encoded_image"].shape) == (1, 40, 1, 28, 28)
    # the six live views, first call and redraw
    s = {"Ae": torch.rand(40, 1, 25) < 0.05, "X": item["encoded_image"]}
    v = {"Ae": torch.randn(40, 1, 25), "Ai": torch.randn(40, 1, 25)}
    axes, ims = plot_input(item["image"].view(28, 28), item["encoded_image"].sum(0).view(28, 28), label=torch.tensor([3]))
    plot_input(item["image"].view(28, 28), item["encoded_image"].sum(0).view(28, 28), label=torch.tensor([4]), axes=axes, ims=ims)
    ims, axes = plot_spikes(s)
    plot_spikes(s, ims=ims, axes=axes)
    im = plot_weights(torch.rand(50, 50))
    plot_weights(torch.rand(50, 50), im=im)
    im = plot_assignments(-torch.ones(5, 5))
    plot_assignments(torch.zeros(5, 5), im=im)
    ax = plot_performance({"all": [], "proportion": []}, x_scale=4)
    plot_performance({"all": [10.0, 50.0], "proportion": [20.0, 60.0]}, x_scale=4, ax=ax)
    ims, axes = plot_voltages(v, plot_type="line")
    plot_voltages(v, ims=ims, axes=axes, plot_type="line")
    plot_voltages(v, plot_type="color")
    from bindsnet.analysis.plotting import plot_locally_connected_weights
    from bindsnet.network.nodes import Input, LIFNodes
    from bindsnet.network.topology import LocalConnection
    lc = LocalConnection(Input(n=144, shape=(1, 12, 12)), LIFNodes(n=45), kernel_size=4, stride=4, n_filters=5, input_shape=(12, 12))
    im = plot_locally_connected_weights(lc.w, 5, lc.kernel_size, lc.conv_size, lc.locations, 12, wmax=float(lc.wmax), title="X -> Y")
    assert im.get_array().shape == (36, 36) and len(im.axes.lines) == 4        # 3 x 3 regions: two separators each way
    plot_locally_connected_weights(lc.w * 0.5, 5, lc.kernel_size, lc.conv_size, lc.locations, 12, im=im)
    import matplotlib.pyplot as plt
    plt.close("all")


def test_monitors_record_by_hand_on_the_host():
    """Monitor.record / NetworkMonitor.record (monitors.py:94-111, 222-262) for code that steps a network by hand: plain
    tensor bookkeeping, no device involved -- layers' `s` / `v`, a connection's `w`, rolling windows."""
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor, NetworkMonitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    net = Network(dt=1.0)
    X, Y = Input(n=6, traces=True), LIFNodes(n=4, traces=True)
    conn = Connection(X, Y, w=torch.arange(24, dtype=torch.float32).view(6, 4) / 100)
    net.add_layer(X, "X"); net.add_layer(Y, "Y"); net.add_connection(conn, "X", "Y")
    mw, nm = Monitor(conn, ["w"], time=3), NetworkMonitor(net, time=3)
    assert sorted(f"{k}:{v}" for k in nm.get() for v in nm.get()[k]) == ["('X', 'Y'):w", "X:s", "Y:s", "Y:v"]
    assert nm.get()[("X", "Y")]["w"].shape == (3, 6, 4) and float(nm.get()[("X", "Y")]["w"].abs().sum()) == 0.0
    snaps = []
    for t in range(5):
        conn.w.data += 1.0
        Y.v = torch.full((1, 4), -60.0 - t)
        snaps.append(conn.w.detach().clone())
        mw.record(); nm.record()
    got = mw.get("w")
    assert got.shape == (3, 6, 4) and all(torch.equal(got[i], snaps[2 + i]) for i in range(3))      # the last `time` steps
    rec = nm.get()
    assert rec[("X", "Y")]["w"].shape == (3, 6, 4) and torch.equal(rec[("X", "Y")]["w"][-1], snaps[-1])
    assert torch.equal(rec["Y"]["v"][:, 0, 0], torch.tensor([-62.0, -63.0, -64.0])) and nm.i == 5
    nm.reset_state_variables()
    assert float(nm.get()[("X", "Y")]["w"].abs().sum()) == 0.0 and nm.i == 0
    grow = NetworkMonitor(net, connections=[("X", "Y")], layers=[], state_vars=["w"])              # no window: grows
    for t in range(4):
        grow.record()
    assert grow.get()[("X", "Y")]["w"].shape == (4, 6, 4)
    with pytest.raises(NotImplementedError):
        NetworkMonitor(net, state_vars=["w", "b"])


def test_reshape_locally_connected_weights_matches_reference():
    """utils.py:112-180 on the receptive fields of LocalConnections (3 x 3 positions, a non-square case, the
    single-position case); the `locations` table of the mirror's LocalConnection equals the reference's."""
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import LocalConnection
    from bindsnet_amd.utils import reshape_locally_connected_weights
    g = gold("op_reshape_local")
    for k, (shape, ks, st) in enumerate((((12, 12), 4, 4), ((10, 10), (4, 2), (3, 2)), ((6, 6), 6, 1))):
        m = [int(v) for v in g[f"meta{k}"]]
        kernel, conv, nf, wshape = tuple(m[2:4]), tuple(m[4:6]), m[6], tuple(m[7:9])
        lc = LocalConnection(Input(n=shape[0] * shape[1], shape=(1, *shape)), LIFNodes(n=nf * conv[0] * conv[1]), kernel_size=ks,
                             stride=st, n_filters=nf, input_shape=shape)
        assert tuple(lc.kernel_size) == kernel and tuple(lc.conv_size) == conv and tuple(lc.w.shape) == wshape
        np.testing.assert_array_equal(lc.locations.numpy(), g[f"loc{k}"])
        w = T_(synth.uniform_f32(880 + k, wshape, 0.0, 1.0))
        img = reshape_locally_connected_weights(w, nf, lc.kernel_size, lc.conv_size, lc.locations, shape)
        np.testing.assert_array_equal(img.numpy(), g[f"img{k}"], err_msg=f"case {k}")


def test_moving_average_reward_matches_reference():
    """bindsnet.learning.reward.MovingAvgRPE (reward.py:29-87), reachable under both package names; as `reward_fn` of a
    Network its compute() feeds the rules' `reward` keyword (network.py:318-320)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from bindsnet.learning.reward import AbstractReward, MovingAvgRPE
    from bindsnet_amd.learning import reward as native
    assert MovingAvgRPE is native.MovingAvgRPE and issubclass(MovingAvgRPE, AbstractReward)
    g = gold("op_reward")
    rs = np.random.RandomState(3)
    episodes = [(float(rs.uniform(-3, 5)), int(rs.randint(5, 40)), [10.0, 4.0, 25.0][ep % 3]) for ep in range(12)]
    r = MovingAvgRPE()
    for k, (acc, steps, win) in enumerate(episodes):
        np.testing.assert_array_equal(r.compute(reward=torch.tensor(acc / steps)).numpy(), g["rpe"][k])
        r.update(accumulated_reward=acc, steps=steps, ema_window=win)
        np.testing.assert_array_equal(r.reward_predict.numpy(), g["per_step"][k])
        np.testing.assert_array_equal(r.reward_predict_episode.numpy(), g["per_episode"][k])
    np.testing.assert_array_equal(np.array(r.rewards_predict_episode), g["history"])
    from bindsnet_amd.network import Network
    net = Network(dt=1.0, reward_fn=MovingAvgRPE)
    assert isinstance(net.reward_fn, MovingAvgRPE)


def test_time_major_dataloader_matches_reference():
    """bindsnet.datasets.DataLoader / time_aware_collate (datasets/dataloader.py, collate.py:27-85): every field kind
    against the reference's collate, and a loader over the MNIST wrapper yielding [time, batch, 1, 28, 28] spike trains --
    the layout examples/mnist/batch_eth_mnist.py hands to Network.run()."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import tv_shim
    tv_shim.install()
    from torchvision import transforms
    import bindsnet
    from bindsnet.datasets import MNIST, DataLoader, time_aware_collate
    from bindsnet.encoding import PoissonEncoder
    from make_golden_host_cases import collate_batch, flatten_collated
    g = gold("op_collate")
    flat = flatten_collated(time_aware_collate(collate_batch()))
    assert sorted(flat) == [str(p) for p in g["paths"]]
    for i, k in enumerate(sorted(flat)):
        want = g[f"t{i}"]
        assert flat[k].numpy().dtype == want.dtype and tuple(flat[k].shape) == want.shape, k
        np.testing.assert_array_equal(flat[k].numpy(), want, err_msg=k)
    ds = MNIST(PoissonEncoder(time=25, dt=1.0), None, root=os.path.join(bindsnet.ROOT_DIR, "data", "MNIST"), download=True, train=True,
               transform=transforms.Compose([transforms.ToTensor(), transforms.Lambda(lambda x: x * 128)]))
    torch.manual_seed(0)
    batch = next(iter(DataLoader(ds, batch_size=4, shuffle=True)))
    assert tuple(batch["encoded_image"].shape) == (25, 4, 1, 28, 28) and batch["encoded_image"].dtype == torch.uint8
    assert tuple(batch["image"].shape) == (1, 4, 28, 28) and batch["label"].numel() == 4
    assert os.path.isdir(os.path.join(bindsnet.ROOT_DIR, "bindsnet_amd"))


def test_increasing_inhibition_network_construction_matches_reference():
    """bindsnet.models.IncreasingInhibitionNetwork (models.py:349-454): seed for seed the same input weights, the same
    distance-graded recurrent weights (bit for bit) and the same generator position after construction."""
    from bindsnet.models import IncreasingInhibitionNetwork
    from bindsnet_amd.learning import PostPre
    g = gold("op_models")
    cases_ = [dict(n_input=64, n_neurons=25, inpt_shape=(1, 8, 8)),
              dict(n_input=30, n_neurons=40, start_inhib=2.5, max_inhib=33.0, theta_plus=0.1, exc_thresh=-50.0)]
    for k, kw in enumerate(cases_):
        torch.manual_seed(4)
        net = IncreasingInhibitionNetwork(**kw)
        np.testing.assert_array_equal(torch.rand(3).numpy(), g[f"probe{k}"])
        assert list(net.layers) == ["X", "Y"] and list(net.connections) == [("X", "Y"), ("Y", "Y")]
        for key, conn in (("w_xy", ("X", "Y")), ("w_yy", ("Y", "Y"))):
            np.testing.assert_array_equal(net.connections[conn].w.detach().numpy().view(np.uint32), g[f"{key}{k}"].view(np.uint32))
        Y = net.layers["Y"]
        got = [float(Y.thresh), float(Y.rest), float(Y.reset), float(Y.refrac), float(Y.theta_plus),
               float(net.connections[("X", "Y")].norm), net.n_sqrt]
        np.testing.assert_array_equal(np.array(got, np.float64), g[f"consts{k}"])
        assert isinstance(net.connections[("X", "Y")].update_rule, PostPre) and Y.one_spike and Y.traces


def test_v2_and_locally_connected_models_construction_matches_reference():
    """bindsnet.models.DiehlAndCook2015v2 (models.py:247-346) and LocallyConnectedNetwork (:457-600): both weight matrices
    bit for bit (incl. the sign of their zeros), layer sizes, and where the torch / numpy generators are left."""
    import bindsnet.models as models
    from make_golden_host_cases import MODEL_CASES
    g = gold("op_models")
    for k, (name, kw) in enumerate(MODEL_CASES):
        torch.manual_seed(4)
        np.random.seed(7)
        net = getattr(models, name)(**kw)
        np.testing.assert_array_equal(np.array([float(torch.rand(1)), np.random.rand()]), g[f"m{k}_probe"], err_msg=name)
        for key, conn in (("w_xy", ("X", "Y")), ("w_yy", ("Y", "Y"))):
            got = net.connections[conn].w.detach().numpy()
            assert got.shape == g[f"m{k}_{key}"].shape
            np.testing.assert_array_equal(got.view(np.uint32), g[f"m{k}_{key}"].view(np.uint32), err_msg=f"{name} {key}")
        assert [net.layers["X"].n, net.layers["Y"].n] == list(g[f"m{k}_n"])
