"""The hand-stepping API on the HOST -- layer.forward(x), connection.compute(s) / update() / normalize() with CPU tensors --
against the op-level fixtures of the unmodified reference (tests/golden/make_golden.py; the same ones the MI355X operators
and the oracle are pinned to).  A network object on the host takes network/host_path.py's plain-PyTorch statements; the
same calls on CUDA tensors take libsnnhip (tests/test_gpu_ops.py, test_gpu_rules.py)."""
import numpy as np
import pytest
import torch

import synth
from cases import check_packed, gold, unpack

u8, f32 = np.uint8, np.float32


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_layers_forward_on_the_host_match_reference():
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.nodes import DiehlAndCookNodes, Input, LIFNodes
    g = gold("op_nodes")
    B, N, T = int(g["B"]), int(g["N"]), int(g["T"])
    I = synth.uniform_f32(900, (T, B, N), -2.0, 6.0)
    net = Network(dt=1.0, batch_size=B)
    lif = LIFNodes(n=N, traces=True, rest=-60.0, reset=-45.0, thresh=-40.0, refrac=2, tc_decay=10.0, lbound=-62.0)
    net.add_layer(lif, "L")
    assert float(lif.decay) == float(g["lif_decay"])
    ras = np.zeros((T, B, N), u8)
    for t in range(T):
        lif.forward(T_(I[t].copy()))
        ras[t] = lif.s.numpy()
    np.testing.assert_array_equal(ras, unpack(g["lif_s"], (T, B, N)))
    for a, key in ((lif.v, "lif_v"), (lif.x, "lif_x"), (lif.refrac_count, "lif_r")):
        np.testing.assert_array_equal(bits(a.numpy()), bits(g[key]), err_msg=key)
    dc = DiehlAndCookNodes(n=N, traces=True, rest=-65.0, reset=-60.0, thresh=-52.0, refrac=5, tc_decay=100.0, theta_plus=0.05,
                           tc_theta_decay=1e7)
    net.add_layer(dc, "D")
    torch.manual_seed(77)                                         # the generator the one_spike draws come from
    for t in range(T):
        dc.forward(T_((I[t] * f32(2.0)).copy()))
        ras[t] = dc.s.numpy()
    np.testing.assert_array_equal(ras, unpack(g["dc_s"], (T, B, N)))
    for a, key in ((dc.v, "dc_v"), (dc.x, "dc_x"), (dc.refrac_count, "dc_r"), (dc.theta, "dc_theta")):
        np.testing.assert_array_equal(bits(a.numpy()), bits(g[key]), err_msg=key)
    torch.manual_seed(77)
    n = int(g["dc_consumed"])
    if n:
        torch.empty(n).exponential_(1)
    expect = torch.rand(3)
    torch.manual_seed(77)
    dc2 = DiehlAndCookNodes(n=N, traces=True, rest=-65.0, reset=-60.0, thresh=-52.0, refrac=5, tc_decay=100.0, theta_plus=0.05, tc_theta_decay=1e7)
    Network(dt=1.0, batch_size=B).add_layer(dc2, "D")
    for t in range(T):
        dc2.forward(T_((I[t] * f32(2.0)).copy()))
    assert torch.equal(torch.rand(3), expect), "host generator position after the one_spike draws"
    x = Input(n=N, traces=True)
    Network(dt=1.0, batch_size=B).add_layer(x, "X")
    sp = T_(synth.dense_spikes(5, (B, N), 0.3))
    x.forward(sp)
    assert x.s is sp and float(x.x[sp.bool()].min()) == 1.0


def test_connection_compute_update_normalize_on_the_host_match_reference():
    from bindsnet_amd.learning import PostPre
    from bindsnet_amd.learning.MCC_learning import PostPre as MCCPostPre
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection, MulticompartmentConnection
    from bindsnet_amd.network.topology_features import Weight
    g = gold("op_prop_mcc")
    for k, (B, Nin, N, p) in enumerate(g["cases"]):
        B, Nin, N = int(B), int(Nin), int(N)
        W = synth.uniform_f32(100 + k, (Nin, N), -1.0, 1.0)
        conn = MulticompartmentConnection(Input(n=Nin), LIFNodes(n=N), device="cpu", pipeline=[Weight("weight", T_(W).clone())])
        out = conn.compute(T_(synth.dense_spikes(200 + k, (B, Nin), float(p))))
        np.testing.assert_array_equal(bits(out.numpy().reshape(B, N)), bits(g[f"out{k}"]), err_msg=f"case {k}")
    g = gold("op_postpre")
    for k, (B, Nin, N) in enumerate(g["cases"]):
        B, Nin, N = int(B), int(Nin), int(N)
        W = synth.uniform_f32(300 + k, (Nin, N), 0.0, 1.0)
        s_src, s_tgt = synth.dense_spikes(400 + k, (B, Nin), 0.3), synth.dense_spikes(500 + k, (B, N), 0.2)
        x_src, x_tgt = synth.uniform_f32(600 + k, (B, Nin), 0.0, 1.0), synth.uniform_f32(700 + k, (B, N), 0.0, 1.0)
        for family in ("mcc", "dense"):
            src, tgt = Input(n=Nin, traces=True), LIFNodes(n=N, traces=True)
            src.batch_size = tgt.batch_size = B
            src.s, src.x, tgt.s, tgt.x = T_(s_src), T_(x_src), T_(s_tgt).bool(), T_(x_tgt)
            if family == "mcc":
                feat = Weight("weight", T_(W).clone(), range=[0.0, 1.0], nu=(1e-4, 1e-2), learning_rule=MCCPostPre, reduction=torch.sum)
                conn = MulticompartmentConnection(src, tgt, device="cpu", pipeline=[feat])
                conn.dt = 1.0
                conn.update(learning=True)
                got = feat.value.detach().numpy()
            else:
                conn = Connection(src, tgt, w=T_(W).clone(), update_rule=PostPre, nu=(1e-4, 1e-2), reduction=torch.sum, wmin=0.0, wmax=1.0)
                conn.update(learning=True)
                got = conn.w.detach().numpy()
            check_packed(g, f"{family}{k}", got)
    g = gold("op_normalize")
    for k, (Nin, N) in enumerate(g["cases"]):
        Nin, N = int(Nin), int(N)
        W = synth.uniform_f32(800 + k, (Nin, N), -0.2, 1.0)
        W[:, N // 2] = 0.0
        feat = Weight("weight", T_(W).clone(), norm=78.4)
        conn = MulticompartmentConnection(Input(n=Nin), LIFNodes(n=N), device="cpu", pipeline=[feat])
        conn.normalize()
        check_packed(g, f"mcc{k}", feat.value.detach().numpy())
        dense = Connection(Input(n=Nin), LIFNodes(n=N), w=T_(W).clone(), norm=78.4)
        dense.normalize()
        check_packed(g, f"dense{k}", dense.w.detach().numpy())


def test_conv2d_compute_on_the_host_matches_reference():
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Conv2dConnection
    g = gold("op_conv2d")
    for k, (B, Cin, H, Wd, Cout, K, stride, pad) in enumerate(g["cases"]):
        B, Cin, H, Wd, Cout, K, stride, pad = (int(v) for v in (B, Cin, H, Wd, Cout, K, stride, pad))
        OH = (H + 2 * pad - K) // stride + 1
        OW = (Wd + 2 * pad - K) // stride + 1
        W = synth.uniform_f32(1000 + k, (Cout, Cin, K, K), 0.0, 0.3)
        c = Conv2dConnection(Input(shape=(Cin, H, Wd)), LIFNodes(shape=(Cout, OH, OW)), kernel_size=K, stride=stride, padding=pad, w=T_(W).clone())
        out = c.compute(T_(synth.dense_spikes(1100 + k, (B, Cin, H, Wd), 0.2)))
        check_packed(g, f"out{k}", out.numpy())


def test_host_and_device_objects_do_not_mix():
    """Where a call runs is decided by where the object's tensors live, never silently: a host layer given a CUDA tensor (or
    the reverse) fails in torch / in the binding, it is not moved."""
    from bindsnet_amd import _lib
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.nodes import LIFNodes
    lif = LIFNodes(n=8)
    Network(dt=1.0, batch_size=2).add_layer(lif, "L")
    lif.forward(torch.zeros(2, 8))
    assert lif.s.shape == (2, 8) and not lif.s.any()
    if not torch.cuda.is_available():
        with pytest.raises((_lib.SnnError, RuntimeError, AssertionError)):
            from bindsnet_amd import ops
            ops.lif_step(lif.v, lif.refrac_count, lif.s, None, torch.zeros(2, 8), lif._lif_params())
