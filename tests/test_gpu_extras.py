"""What widens the boundary beyond the BASELINE graphs (SURVEY 8(f)-4 and the run() keyword arguments of row a1), each
against fixtures from the unmodified reference (tests/golden/make_golden_r2.py: run_extras.npz):
 * run(..., clamp=, unclamp=, injects_v=, masks=) on a dense two-layer network (network.py:395-449);
 * LocalConnection with PostPre (topology.py:1304-1485);
 * PostPre on a Conv2dConnection (learning.py:457-497): single updates (bit-exact vs the oracle's canonical order, 1e-5
   vs the reference's BLAS order) and a run."""
import numpy as np
import pytest
import torch

import oracle
import synth
from cases import f32, gold, u8, unpack

pytestmark = pytest.mark.gpu
DEV = "cuda"


def host(t):
    return t.detach().cpu().numpy()


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_run_keyword_arguments_match_reference():
    from bindsnet_amd.models import TwoLayerNetwork
    from bindsnet_amd.network.monitors import Monitor
    g = gold("run_extras")
    Nin, N, B, T = 196, 48, 3, 40
    torch.manual_seed(0)
    net = TwoLayerNetwork(n_inpt=Nin, n_neurons=N, reduction=torch.sum, norm=78.4 * Nin / 784)
    conn = net.connections[("X", "Y")]
    conn.w.data.copy_(T_(synth.weights_q12(11, Nin, N)))
    mon, mv = Monitor(net.layers["Y"], ["s"], time=T), Monitor(net.layers["Y"], ["v"], time=T)
    net.add_monitor(mon, "s"); net.add_monitor(mv, "v")
    net.to(DEV)
    spikes = synth.spike_train(30, T, B, Nin, active=0.3, max_rate=0.12)
    clamp = T_(synth.dense_spikes(51, (T, N), 0.03)).bool()
    unclamp = T_(synth.dense_spikes(52, (N,), 0.2)).bool()
    inject = T_(synth.uniform_f32(53, (N,), 0.0, 0.6))
    mask = T_(synth.dense_spikes(54, (Nin, N), 0.3)).bool()
    net.run({"X": T_(spikes).to(DEV)}, time=T, clamp={"Y": clamp}, unclamp={"Y": unclamp}, injects_v={"Y": inject},
            masks={("X", "Y"): mask})
    assert net.last_plan == "generic"
    np.testing.assert_array_equal(host(mon.get("s")).reshape(T, B, N).astype(u8), unpack(g["kw_sY"], (T, B, N)))
    np.testing.assert_allclose(host(mv.get("v")), g["kw_v"], rtol=0, atol=2e-4)       # (MKL order in the reference's `@`)
    W = host(conn.w)
    np.testing.assert_allclose(W, g["kw_W"], rtol=0, atol=1e-5)
    assert (W[mask.numpy()] == 0).all()


def test_local_connection_postpre_matches_reference():
    from bindsnet_amd.learning import PostPre
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import LocalConnection
    g = gold("run_extras")
    np.random.seed(7)
    T2 = 50
    net = Network(dt=1.0)
    X, Y = Input(n=144, traces=True), LIFNodes(n=4 * 16, traces=True)
    lc = LocalConnection(X, Y, kernel_size=6, stride=2, n_filters=4, update_rule=PostPre, nu=(1e-4, 1e-2), wmin=0.0, wmax=1.0, norm=0.2)
    net.add_layer(X, "X"); net.add_layer(Y, "Y"); net.add_connection(lc, "X", "Y")
    np.testing.assert_array_equal(host(lc.w).view(np.uint32), g["lc_W0"].view(np.uint32))          # numpy-generator initialisation
    np.testing.assert_array_equal(np.packbits(host(lc.mask)), g["lc_mask"])
    assert lc.norm == float(g["lc_norm"])
    mon = Monitor(Y, ["s"], time=T2)
    net.add_monitor(mon, "s")
    net.to(DEV)
    sp = synth.spike_train(31, T2, 1, 144, active=0.5, max_rate=0.25)
    net.run({"X": T_(sp).to(DEV)}, time=T2)
    np.testing.assert_array_equal(host(mon.get("s")).reshape(T2, 1, 64).astype(u8), unpack(g["lc_sY"], (T2, 1, 64)))
    np.testing.assert_allclose(host(lc.w), g["lc_W"], rtol=0, atol=1e-5)
    assert (host(lc.w)[host(lc.mask)] == 0).all()


def test_conv2d_postpre_updates_and_run():
    from bindsnet_amd.learning import PostPre
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Conv2dConnection
    g = gold("run_extras")
    for k, (B, Cin, H, Wd, Cout, K, stride, pad) in enumerate(g["cpp_cases"]):
        B, Cin, H, Wd, Cout, K, stride, pad = (int(v) for v in (B, Cin, H, Wd, Cout, K, stride, pad))
        OH = (H + 2 * pad - K) // stride + 1
        W0 = synth.uniform_f32(1200 + k, (Cout, Cin, K, K), 0.0, 0.5)
        s_src, x_src = synth.dense_spikes(1300 + k, (B, Cin, H, Wd), 0.15), synth.uniform_f32(1400 + k, (B, Cin, H, Wd), 0.0, 1.0)
        s_tgt, x_tgt = synth.dense_spikes(1500 + k, (B, Cout, OH, OH), 0.1), synth.uniform_f32(1600 + k, (B, Cout, OH, OH), 0.0, 1.0)
        src, tgt = Input(shape=(Cin, H, Wd), traces=True), LIFNodes(shape=(Cout, OH, OH), traces=True)
        c = Conv2dConnection(src, tgt, kernel_size=K, stride=stride, padding=pad, w=T_(W0).clone(), update_rule=PostPre, nu=(1e-3, 1e-2),
                             reduction=torch.sum, wmin=0.0, wmax=1.0).to(DEV)
        for l, s, x in ((src, s_src, x_src), (tgt, s_tgt, x_tgt)):
            l.batch_size = B
            l.s, l.x = T_(s).to(DEV), T_(x).to(DEV)
        c.update(learning=True)
        Wo = W0.copy()
        oracle.conv2d_postpre(Wo, s_src, x_src, s_tgt, x_tgt, stride=stride, pad=pad, nu0=np.float32(1e-3), nu1=np.float32(1e-2), wmin=0.0, wmax=1.0)
        np.testing.assert_array_equal(host(c.w).view(np.uint32), Wo.view(np.uint32), err_msg=f"case {k} vs oracle")
        np.testing.assert_allclose(host(c.w), g[f"cpp{k}"], rtol=0, atol=1e-5, err_msg=f"case {k} vs reference")
    B, T3 = 2, 30
    from bindsnet_amd import _lib
    for mode, plan in ((0, "convpp-fused"), (1, "generic")):       # the whole-run plan (round 6) and the per-operator plan, both against the reference's run
        _lib.lib().snn_set_plan_mode(mode)
        try:
            net = Network(dt=1.0)
            net.add_layer(Input(shape=(1, 12, 12), traces=True), "X")
            net.add_layer(LIFNodes(shape=(4, 10, 10), traces=True), "Y")
            cc = Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=3, stride=1, w=T_(synth.uniform_f32(1700, (4, 1, 3, 3), 0.0, 3.0)).clone(),
                                  update_rule=PostPre, nu=(1e-3, 1e-2), reduction=torch.sum, wmin=0.0, wmax=4.0)
            net.add_connection(cc, "X", "Y")
            mon = Monitor(net.layers["Y"], ["s"], time=T3)
            net.add_monitor(mon, "s")
            net.to(DEV)
            sp = synth.dense_spikes(1701, (T3, B, 1, 12, 12), 0.2)
            net.run({"X": T_(sp).to(DEV)}, time=T3)
            assert net.last_plan == plan
        finally:
            _lib.lib().snn_set_plan_mode(0)
        np.testing.assert_array_equal(host(mon.get("s")).reshape(T3, B, 400).astype(u8), unpack(g["crun_sY"], (T3, B, 400)), err_msg=plan)
        # every single update is bit-exact against the oracle (above); against the REFERENCE the batch/position sums of its two
        # torch.bmm calls run in BLAS order, and 30 updates accumulate that: the north star's 1e-5, relative to wmax = 4.0
        np.testing.assert_allclose(host(cc.w), g["crun_W"], rtol=0, atol=1e-5 * 4.0, err_msg=plan)


def test_mstdp_on_conv2d_connection_matches_oracle_and_reference():
    """MSTDP on a Conv2dConnection (learning.py:1942-2015; batch 1).  Update sequences through the rule object: bit-exact
    against the oracle, within the BLAS tolerance against the reference (its eligibility comes out of two torch.bmm
    calls).  A Network.run(): against a hand-stepped oracle (conv2d propagation, LIF step, rule) bit for bit, against
    the reference fixture rasters exactly and weights / eligibility within tolerance."""
    from test_oracle_golden import conv_mstdp_run_oracle, conv_mstdp_sequence, unfold_np
    from bindsnet_amd.learning import MSTDP
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Conv2dConnection
    g = gold("op_conv_mstdp")
    dp, dm = np.float32(g["decay_plus"]), np.float32(g["decay_minus"])
    for k in range(len(g["cases"])):
        Cin, H, Wd, Cout, K, stride, pad = (int(v) for v in g["cases"][k])
        OH = (H + 2 * pad - K) // stride + 1
        wdec = 1e-3 if k == 1 else 0.0
        src, tgt = Input(shape=(Cin, H, Wd), traces=True), LIFNodes(shape=(Cout, OH, OH), traces=True)
        c = Conv2dConnection(src, tgt, kernel_size=K, stride=stride, padding=pad, w=T_(synth.uniform_f32(2300 + k, (Cout, Cin, K, K), 0.0, 0.5)).clone(),
                             update_rule=MSTDP, nu=(2e-2, 1e-2), wmin=0.0, wmax=0.6, weight_decay=wdec).to(DEV)
        c.dt = 1.0
        src.batch_size = tgt.batch_size = 1

        def gpu_step(W, E, P, Q, s_src, s_tgt, reward):
            src.s, tgt.s = T_(s_src[None]).to(DEV), T_(s_tgt[None]).to(DEV)
            c.update(learning=True, reward=reward, a_plus=1.0, a_minus=-0.8)

        def orc_step(W, E, P, Q, s_src, s_tgt, reward):
            oracle.conv2d_mstdp(W, E, P, Q, s_src, s_tgt, stride=stride, pad=pad, reward=reward, nu0=np.float32(2e-2), a_plus=1.0,
                                a_minus=-0.8, decay_plus=dp, decay_minus=dm, wdecay=np.float32(1.0 - wdec) if wdec else 1.0, wmin=0.0, wmax=0.6)

        conv_mstdp_sequence(g, k, gpu_step)
        W, E, P, Q, _ = conv_mstdp_sequence(g, k, orc_step)
        ur = c.update_rule
        for got, want, name in ((c.w, W, "w"), (ur.eligibility, E, "eligibility"), (ur.p_plus, P, "p_plus"), (ur.p_minus, Q.reshape(Cout, -1), "p_minus")):
            np.testing.assert_array_equal(host(got).view(np.uint32), want.view(np.uint32), err_msg=f"case {k} {name} vs oracle")
        np.testing.assert_allclose(host(c.w), g[f"w{k}"], rtol=0, atol=1e-5, err_msg=f"case {k} w vs reference")
        np.testing.assert_allclose(host(ur.eligibility), g[f"elig{k}"], rtol=0, atol=1e-5, err_msg=f"case {k} eligibility vs reference")
        np.testing.assert_array_equal(unfold_np(host(ur.p_plus), K, stride, pad).view(np.uint32), g[f"p_plus{k}"][0].view(np.uint32))
        with pytest.raises(NotImplementedError):
            src.batch_size = 2
            c.update(learning=True, reward=0.5)
    # ---- a run
    T3 = 40
    W0 = synth.uniform_f32(2290, (4, 1, 3, 3), 0.0, 3.0)
    net = Network(dt=1.0)
    net.add_layer(Input(shape=(1, 12, 12), traces=True), "X")
    net.add_layer(LIFNodes(shape=(4, 10, 10), traces=True), "Y")
    cc = Conv2dConnection(net.layers["X"], net.layers["Y"], kernel_size=3, stride=1, w=T_(W0).clone(), update_rule=MSTDP, nu=(2e-3, 1e-3),
                          wmin=0.0, wmax=4.0)
    net.add_connection(cc, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T3)
    net.add_monitor(mon, "s")
    net.to(DEV)
    sp = synth.dense_spikes(2291, (T3, 1, 1, 12, 12), 0.2)
    net.run({"X": T_(sp).to(DEV)}, time=T3, reward=0.6)
    assert net.last_plan == "generic"
    ras = host(mon.get("s")).reshape(T3, 400).astype(u8)
    ras_o, W, E = conv_mstdp_run_oracle(g, T3)     # hand-stepped oracle (conv2d propagation, LIF step, rule), pinned on the CPU
    np.testing.assert_array_equal(ras, ras_o)
    np.testing.assert_array_equal(host(cc.w).view(np.uint32), W.view(np.uint32))
    np.testing.assert_array_equal(host(cc.update_rule.eligibility).view(np.uint32), E.view(np.uint32))
    np.testing.assert_array_equal(ras, unpack(g["run_sY"], (T3, 400)))
    # (bit-exact against the hand-stepped oracle above; the reference's eligibility comes out of two torch.bmm calls in
    #  BLAS order: 1e-5 relative to wmax = 4.0 for the weights; the eligibility is a sum over up to 100 output positions
    #  of products of O(1) traces, compared relative to its own magnitude)
    np.testing.assert_allclose(host(cc.w), g["run_W"], rtol=0, atol=1e-5 * 4.0)
    np.testing.assert_allclose(host(cc.update_rule.eligibility), g["run_elig"], rtol=0, atol=1e-5 * max(1.0, float(np.abs(g["run_elig"]).max())))
    assert ras.sum() > 500 and np.abs(host(cc.w) - W0).max() > 1e-2


def test_one_step_mode_matches_reference():
    """run(..., one_step=True): every layer's input comes from the CURRENT spikes of its sources (network.py:388-393);
    Input -> A -> B with a feedback B -> A.  Both modes against the reference, bit for bit (MCC path: ATen order)."""
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import MulticompartmentConnection
    from bindsnet_amd.network.topology_features import Weight
    g = gold("run_one_step")
    nX, nA, nB, B, T = 64, 40, 24, 2, 30
    net = Network(dt=1.0, learning=False)
    net.add_layer(Input(n=nX), "X"); net.add_layer(LIFNodes(n=nA, thresh=-60.0), "A"); net.add_layer(LIFNodes(n=nB, thresh=-61.0), "B")
    for k, (src, dst, ns, nd, sc) in enumerate((("X", "A", nX, nA, 2.0), ("A", "B", nA, nB, 3.0), ("B", "A", nB, nA, -1.0))):
        w = (synth.uniform_f32(2200 + k, (ns, nd), 0.0, abs(sc)) * np.sign(sc)).astype(np.float32)
        net.add_connection(MulticompartmentConnection(net.layers[src], net.layers[dst], device="cpu", pipeline=[Weight("weight", T_(w).clone())]), src, dst)
    mons = {l: Monitor(net.layers[l], ["s"], time=T) for l in ("A", "B")}
    for l, m in mons.items():
        net.add_monitor(m, l)
    net.to(DEV)
    sp = synth.dense_spikes(2210, (T, B, nX), 0.15)
    for tag, flag in (("one", True), ("sync", False)):
        net.reset_state_variables()
        net.run({"X": T_(sp).to(DEV)}, time=T, one_step=flag)
        for l, n in (("A", nA), ("B", nB)):
            np.testing.assert_array_equal(host(mons[l].get("s")).reshape(T, B, n).astype(u8), unpack(g[f"{tag}_{l}"], (T, B, n)), err_msg=f"{tag} {l}")
    assert not np.array_equal(g["one_A"], g["sync_A"])


def _r3_chain(T):
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import MulticompartmentConnection
    from bindsnet_amd.network.topology_features import Weight
    nX, nA, nB = 64, 40, 24
    net = Network(dt=1.0, learning=False)
    net.add_layer(Input(n=nX), "X"); net.add_layer(LIFNodes(n=nA, thresh=-60.0), "A"); net.add_layer(LIFNodes(n=nB, thresh=-61.0), "B")
    for k, (src, dst, ns, nd, sc) in enumerate((("X", "A", nX, nA, 2.0), ("A", "B", nA, nB, 0.35), ("B", "A", nB, nA, -1.0))):
        w = (synth.uniform_f32(3200 + k, (ns, nd), 0.0, abs(sc)) * np.sign(sc)).astype(np.float32)
        net.add_connection(MulticompartmentConnection(net.layers[src], net.layers[dst], device="cpu", pipeline=[Weight("weight", T_(w).clone())]), src, dst)
    mons = {l: Monitor(net.layers[l], ["s", "v"], time=T) for l in ("A", "B")}
    for l, m in mons.items():
        net.add_monitor(m, l)
    return net.to(DEV), mons


def test_external_currents_into_non_input_layers_match_reference():
    """run(inputs={"X": spikes, "A": currents, "B": currents}): an entry for a non-Input layer is an external current,
    added to that layer's summed input behind its connections' contributions (network.py:386-392).  Rasters and membrane
    potentials of both layers, every timestep, bit for bit against the reference (tests/golden/make_golden_r3.py)."""
    g = gold("run_ext_current")
    nX, nA, nB, B, T = 64, 40, 24, 3, 30
    net, mons = _r3_chain(T)
    sp = synth.dense_spikes(3210, (T, B, nX), 0.10)
    cA = synth.uniform_f32(3211, (T, B, nA), -1.0, 4.0)
    cB = synth.uniform_f32(3212, (T, B, nB), 0.0, 2.5)
    net.run({"X": T_(sp).to(DEV), "A": T_(cA).to(DEV), "B": T_(cB)}, time=T)      # (one of them left on the host: moved by run())
    assert net.last_plan == "generic"
    for l, n in (("A", nA), ("B", nB)):
        np.testing.assert_array_equal(host(mons[l].get("s")).reshape(T, B, n).astype(u8), unpack(g[f"s_{l}"], (T, B, n)), err_msg=f"raster {l}")
        np.testing.assert_array_equal(host(mons[l].get("v")).reshape(T, B, n).view(np.uint32), g[f"v_{l}"].reshape(T, B, n).view(np.uint32), err_msg=f"v {l}")
    # ... and a later call WITHOUT the currents must not see the old ones (the kept descriptors are rebound per call)
    net.reset_state_variables()
    net.run({"X": T_(sp).to(DEV)}, time=T)
    a = host(mons["A"].get("s")).copy()
    net.reset_state_variables()
    net.run({"X": T_(sp).to(DEV), "A": torch.zeros(T, B, nA), "B": torch.zeros(T, B, nB)}, time=T)
    np.testing.assert_array_equal(a, host(mons["A"].get("s")))


def test_one_step_with_clamp_matches_reference():
    """one_step=True together with clamp / unclamp: the reference clamps a layer right behind its own step
    (network.py:394-429), so the layers behind it take their currents from the CLAMPED spikes in the same timestep."""
    g = gold("run_one_step_clamp")
    nX, nA, nB, B, T = 64, 40, 24, 3, 30
    net, mons = _r3_chain(T)
    sp = synth.dense_spikes(3220, (T, B, nX), 0.15)
    clampA = torch.zeros(nA, dtype=torch.bool); clampA[::7] = True
    unclampA = torch.zeros(nA, dtype=torch.bool); unclampA[3::5] = True
    for tag, flag in (("one", True), ("sync", False)):
        net.reset_state_variables()
        net.run({"X": T_(sp).to(DEV)}, time=T, one_step=flag, clamp={"A": clampA}, unclamp={"A": unclampA})
        for l, n in (("A", nA), ("B", nB)):
            np.testing.assert_array_equal(host(mons[l].get("s")).reshape(T, B, n).astype(u8), unpack(g[f"{tag}_{l}"], (T, B, n)), err_msg=f"{tag} {l}")
    assert not np.array_equal(g["one_B"], g["sync_B"])


def test_one_step_drops_the_external_current_of_fed_layers_like_the_reference():
    """one_step=True + external currents into non-Input layers on the generic plan (csrc/snn_run.hip): the reference replaces the
    current of a layer that a connection feeds (network.py:386-393); the host twin is tests/test_host_path.py."""
    from test_host_path import one_step_ext_current_runs
    one_step_ext_current_runs(DEV)
