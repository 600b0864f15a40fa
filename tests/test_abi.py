"""CPU-side checks: the C-ABI library loads, exports every symbol include/snnhip.h declares,
rejects bad arguments without touching a GPU, and the host logic (generator-state codec,
monitors, API surface, delta merge) behaves.  No compute calls here."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from bindsnet_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "snnhip.h")).read()
    declared = set(re.findall(r"\b(snn_[a-z0-9_]+)\s*\(", hdr))
    L = _lib.lib()
    for sym in sorted(declared):
        assert hasattr(L, sym), f"libsnnhip.so does not export {sym}"
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    assert L.snn_abi_version() == _lib.ABI_VERSION == 8
    assert L.snn_error_string(-2).decode().startswith("size or mode")


def test_entry_points_validate_arguments_without_a_gpu():
    from bindsnet_amd import _lib
    L = _lib.lib()
    assert L.snn_prop_cascade_f32(None, None, None, 1, 1, 1, 0, None) == -1
    assert L.snn_prop_cascade_f32(1, 1, 1, 1, (1 << 19) + 1, 1, 0, None) == -2
    assert L.snn_stdp_postpre(1, 1, 1, 1, 1, 300, 4, 4, 0.1, 0.1, 1, 1.0, 1.0, 0, 0.0, 0, 0.0, 0, None) == -2
    assert L.snn_normalize(None, 1, 1, 1.0, 0, None, None) == -1
    assert L.snn_net_run(None, 0, None, 0, None, None) == -1


def test_struct_layouts_match_header():
    """ctypes mirrors of the descriptor structs: sizes computed independently with the C compiler."""
    import ctypes, subprocess, tempfile
    from bindsnet_amd import _lib
    src = '#include "snnhip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",' \
          'sizeof(snn_lif_params),sizeof(snn_dc_params),sizeof(snn_layer_desc),sizeof(snn_conn_desc),' \
          'sizeof(snn_run_desc),sizeof(snn_rng_state));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        sizes = [int(v) for v in subprocess.check_output([os.path.join(d, "t")]).split()]
    from bindsnet_amd import rng
    assert sizes == [ctypes.sizeof(_lib.LifParams), ctypes.sizeof(_lib.DcParams), ctypes.sizeof(_lib.LayerDesc),
                     ctypes.sizeof(_lib.ConnDesc), ctypes.sizeof(_lib.RunDesc), rng.RNG_STATE_BYTES]


def test_generator_state_codec_roundtrip_and_semantics():
    from bindsnet_amd import rng
    import oracle
    for seed, warm in ((1, 0), (2, 5), (3, 623), (4, 624), (5, 10000)):
        torch.manual_seed(seed)
        if warm:
            torch.rand(warm)
        st = torch.get_rng_state()
        img = rng.torch_state_to_words(st)
        assert torch.equal(rng.words_to_torch_state(img, st), st) or warm == 0   # seeded state: left=1,next=0 form
        # draw 1000 exponentials with the C mt19937 from the decoded state == torch's
        mt = img[:624].view(np.uint32).copy()
        out, pos = oracle.mt_exponential(mt, int(img[624]), 1000)
        ref = torch.empty(1000).exponential_(1).numpy()
        np.testing.assert_array_equal(out.view(np.uint32), ref.view(np.uint32))
        img2 = img.copy(); img2[:624] = mt.view(np.int32); img2[624] = pos
        torch.set_rng_state(rng.words_to_torch_state(img2, st))
        a = torch.rand(3)
        torch.set_rng_state(st); torch.empty(1000).exponential_(1)
        assert torch.equal(a, torch.rand(3))


def test_api_surface_matches_reference_names():
    from bindsnet_amd.models import DiehlAndCook2015, TwoLayerNetwork
    from bindsnet_amd.network import Network, load
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import DiehlAndCookNodes, Input, LIFNodes, Nodes
    from bindsnet_amd.network.topology import Connection, Conv2dConnection, MulticompartmentConnection
    from bindsnet_amd.network.topology_features import Weight
    from bindsnet_amd.learning import MSTDP, NoOp, PostPre
    from bindsnet_amd.learning.MCC_learning import PostPre as MCCPostPre
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=100, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05,
                           inpt_shape=(1, 28, 28))
    assert list(net.layers) == ["X", "Ae", "Ai"]
    assert list(net.connections) == [("X", "Ae"), ("Ae", "Ai"), ("Ai", "Ae")]
    Ae, Ai, X = net.layers["Ae"], net.layers["Ai"], net.layers["X"]
    assert isinstance(Ae, DiehlAndCookNodes) and isinstance(Ai, LIFNodes) and isinstance(X, Input)
    assert Ae.v.shape == (1, 100) and Ae.s.dtype == torch.bool and Ae.theta.shape == (100,)
    assert float(Ae.v[0, 0]) == -65.0 and float(Ai.v[0, 0]) == -60.0 and X.x.shape == (1, 1, 28, 28)
    feat = net.connections[("X", "Ae")].feature_index["weight"]
    assert isinstance(feat, Weight) and isinstance(feat.learning_rule, MCCPostPre) and feat.value.shape == (784, 100)
    assert feat.learning_rule.reduction is torch.sum and feat.norm == 78.4
    # same generator consumption as the reference constructor: first weights are 0.3 * rand
    torch.manual_seed(0)
    assert torch.equal(feat.value.data, 0.3 * torch.rand(784, 100))
    # decay constants are torch-computed like the reference
    assert float(Ae.decay) == float(torch.exp(-torch.tensor(1.0) / torch.tensor(100.0)))
    two = TwoLayerNetwork(784, 50, reduction=torch.sum)
    assert isinstance(two.connections[("X", "Y")].update_rule, PostPre)
    with pytest.raises(AssertionError):
        Connection(Input(n=4), LIFNodes(n=3), update_rule=PostPre, nu=1e-2)      # traces required
    with pytest.raises(AssertionError):
        net.run([1, 2], time=1)                                                   # inputs must be a dict
    # a network on the host runs the plain-PyTorch step loop (network/host_path.py; tests/test_host_path.py pins it) ...
    net.run({"X": torch.zeros(5, 1, 1, 28, 28, dtype=torch.uint8)}, time=5)
    assert net.last_plan == "host-torch", "the host path is not a plan of libsnnhip"
    # ... but nothing is moved between host and device behind the caller's back
    if torch.cuda.is_available():
        with pytest.raises(Exception) as e:
            net.run({"X": torch.zeros(5, 1, 1, 28, 28, dtype=torch.uint8, device="cuda")}, time=5)
        assert "network.to('cuda')" in str(e.value)
    with pytest.raises(NotImplementedError):
        Input(n=3, sum_input=True)


def test_monitor_window_semantics():
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import LIFNodes
    m = Monitor(LIFNodes(n=4), ["s"], time=5)
    assert m.get("s").numel() == 0
    m._append("s", torch.arange(3).view(3, 1, 1).expand(3, 1, 4))
    m._append("s", (10 + torch.arange(4)).view(4, 1, 1).expand(4, 1, 4))
    got = m.get("s")
    assert got.shape == (5, 1, 4) and got[:, 0, 0].tolist() == [2, 10, 11, 12, 13]
    m.reset_state_variables()
    assert m.get("s").numel() == 0


def test_merge_deltas_single_process():
    from bindsnet_amd import parallel
    b = [torch.ones(3, 2), torch.zeros(4)]
    a = [torch.full((3, 2), 1.5), torch.arange(4.0)]
    parallel.merge_deltas(b, a)
    assert torch.equal(a[0], torch.full((3, 2), 1.5)) and torch.equal(a[1], torch.arange(4.0))


def test_scalar_parameter_cache_and_cpu_reset():
    """Host logic that needs no GPU: the cached scalar reads follow in-place edits of the parameter tensors, and
    Network.reset_state_variables() on CPU tensors takes the per-layer path with the reference's semantics
    (s, x, refrac_count -> 0, v -> rest, theta untouched)."""
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.nodes import _f
    t = torch.tensor(-52.0)
    assert _f(t) == -52.0 and _f(t) == -52.0
    t.fill_(-50.0)
    assert _f(t) == -50.0
    t.add_(1.5)
    assert _f(t) == -48.5
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=16, n_neurons=4, exc=22.5, inh=120, dt=1.0, norm=1.6, theta_plus=0.05, inpt_shape=(1, 4, 4))
    Ae, Ai, X = net.layers["Ae"], net.layers["Ai"], net.layers["X"]
    for l in (Ae, Ai):
        l.set_batch_size(3)
    X.set_batch_size(3)
    Ae.v.fill_(-40.0); Ae.refrac_count.fill_(3.0); Ae.s.fill_(1); Ae.x.fill_(0.7); Ae.theta.fill_(0.3)
    Ai.v.fill_(-41.0); Ai.refrac_count.fill_(1.0)
    X.s = torch.ones(3, 1, 4, 4, dtype=torch.uint8); X.x.fill_(0.5)
    net.reset_state_variables()
    assert float(Ae.v.min()) == float(Ae.v.max()) == float(Ae.rest)
    assert float(Ai.v.min()) == float(Ai.v.max()) == float(Ai.rest)
    assert not Ae.s.any() and not X.s.any() and float(Ae.x.abs().sum()) == 0.0 and float(X.x.abs().sum()) == 0.0
    assert float(Ae.refrac_count.abs().sum()) == 0.0 and float(Ai.refrac_count.abs().sum()) == 0.0
    assert torch.all(Ae.theta == 0.3)
    net.run({"X": torch.zeros(5, 3, 1, 4, 4, dtype=torch.uint8)}, time=5)          # CPU tensors: the host path (test_host_path.py)
    assert net.last_plan == "host-torch"


def test_bench_roofline_accounting():
    """bench.py's byte figures: SURVEY.md 8(d)'s dense accounting (5.86 MB per timestep at cfg2, 1.03 MB at cfg1) and the
    sparse-effective variant it reports separately (weight rows of spiking sources only) -- pure host arithmetic."""
    import importlib.util
    import os
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.algorithmic_bytes_per_timestep() == 5860480
    assert bench.algorithmic_bytes_per_timestep(784, 100, 1) == 4 * (3 * 78400 + 2 * 10000) + (7840 + 2600 + 1800) + 800
    rs = np.random.RandomState(0)
    pool = [(rs.uniform(size=(250, 32, 1, 28, 28)) < 0.0117).astype(np.uint8)]
    sb, rows = bench.sparse_effective_bytes_per_timestep(pool)
    assert 150 < rows < 350 and sb < bench.algorithmic_bytes_per_timestep()
    assert abs(sb - (4 * (rows * 400 + 2 * 784 * 400 + 2 * 32 * 400) + 32 * (7840 + 10400 + 7200) + 3200)) < 1
