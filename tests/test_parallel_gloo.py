"""Multi-process (gloo, world size 2, CPU) coverage of the batch-shard merge used at N > 1 GPUs:
bindsnet_amd.parallel.merge_deltas / sharded_run.  test_sharded_merge_world2_gloo stubs the run itself and checks the
distributed logic (normalisation postponed until after the merge, weight and theta deltas summed over ranks with one
all-reduce, clamp, then normalise -- identical on every rank); test_sharded_run_world2_gloo_real_shards_vs_oracle runs
the real thing: host-path shards, the real sharded_run, the oracle per shard as the checker.
test_exact_batch_sharded_mode_*: bindsnet_amd.parallel.exact_run (SURVEY 8(e) "exact": batch shards + one all-gather of the
spikes per timestep) at world sizes 2 / 3 / 4 on the host operators == the REFERENCE's single-process global batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bindsnet_amd import parallel
        from bindsnet_amd.models import DiehlAndCook2015
        torch.manual_seed(0)                              # identical replicas
        net = DiehlAndCook2015(n_inpt=64, n_neurons=16, exc=22.5, inh=120, dt=1.0, norm=7.0, theta_plus=0.05,
                               inpt_shape=(1, 8, 8))
        feat = net.connections[("X", "Ae")].pipeline[0]
        W0 = feat.value.data.clone()
        calls = []

        def fake_run(inputs, time, **kw):                 # stands in for the device run of this rank's shard
            assert net.__dict__.get("_defer_norm") is True, "normalisation must be postponed until after the merge"
            g = torch.Generator().manual_seed(100 + rank)
            feat.value.data += 0.01 * torch.rand(W0.shape, generator=g) * (rank + 1)
            net.layers["Ae"].theta += 0.05 * (rank + 1)
            calls.append("run")

        net.run = fake_run                                # (the normalisation after the merge is the host path's own)
        # 1) raw merge
        b = [torch.ones(3), torch.zeros(2, 2)]
        a = [torch.ones(3) + (rank + 1), torch.full((2, 2), float(rank))]
        parallel.merge_deltas(b, a)
        assert torch.equal(a[0], torch.ones(3) + 3.0) and torch.equal(a[1], torch.full((2, 2), 1.0))
        # 2) whole sharded step
        parallel.sharded_run(net, {"X": torch.zeros(2, 1, 1, 8, 8, dtype=torch.uint8)}, 2)
        assert calls == ["run"] and feat.norm == 7.0 and net.__dict__.get("_defer_norm") is False
        exp = W0.clone()
        for r in range(world):
            g = torch.Generator().manual_seed(100 + r)
            exp += 0.01 * torch.rand(W0.shape, generator=g) * (r + 1)
        exp.clamp_(0.0, 1.0)
        exp *= 7.0 / exp.sum(0, keepdim=True)
        torch.testing.assert_close(feat.value.data, exp, rtol=0, atol=1e-6)
        assert torch.allclose(net.layers["Ae"].theta, torch.full((16,), 0.05 * 3))
        out.put((rank, feat.value.data.numpy().tobytes()))
    finally:
        dist.destroy_process_group()


def test_sharded_merge_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == res[1], "replicas diverged after the merge"


# ------------------------------------------------------------------------------------------------ exact mode
def _column_worker(rank, world, port, q):
    """One rank of the exact multi-GPU mode, with the CPU oracle standing in for the device: it runs ITS column slice of
    cfg3's graph (all samples), nothing is exchanged during the run, and the slices are gathered at the end."""
    import os, sys
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import numpy as np
    import torch
    import torch.distributed as dist
    import oracle
    import synth
    from bindsnet_amd.parallel import column_shard_bounds, gather_columns
    from test_oracle_fullsize import two_state_cols
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    P, spikes, W0 = _column_case()
    lo, hi = column_shard_bounds(P.N, world, rank)
    N_full = P.N
    P.N = hi - lo
    st = two_state_cols(P, W0[:, lo:hi])
    ras = oracle.run_two_layer(P, st, spikes)
    W = gather_columns(torch.from_numpy(st["W"]), N_full)
    R = gather_columns(torch.from_numpy(ras), N_full)
    V = gather_columns(torch.from_numpy(st["vY"]), N_full)
    if rank == 0:
        q.put((W.numpy(), R.numpy(), V.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _column_case():
    import numpy as np
    import oracle
    import synth
    P = oracle.TwoParams()
    P.B, P.Nin, P.N, P.T, P.rule, P.dt = 48, 784, 160, 40, 1, 1.0           # B > 32: three 16-sample cascade blocks
    P.x_trace_decay = P.y_trace_decay = float(np.exp(np.float32(-1.0 / 20.0))); P.x_trace_scale = P.y_trace_scale = 1.0
    P.x_traces = P.y_traces = 1
    P.decay = float(np.exp(np.float32(-1.0 / 100.0))); P.rest, P.reset, P.thresh, P.refrac = -65.0, -65.0, -52.0, 5.0
    P.nu0, P.nu1, P.has_min, P.has_max, P.wmin, P.wmax, P.has_norm, P.norm, P.learning = 1e-4, 1e-2, 1, 1, 0.0, 1.0, 1, 78.4, 1
    spikes = synth.dense_spikes(2, (P.T, P.B, P.Nin), 0.02)
    return P, spikes, synth.weights_q12(11, P.Nin, P.N)


@pytest.mark.parametrize("world", [2, 3])
def test_column_sharded_exact_mode_equals_the_unsharded_run(world):
    """SURVEY 8(e) "exact": column slices (32-aligned) of a coupling-free graph computed independently by `world`
    processes and gathered == the single-process result for the global batch, bit for bit -- weights (after the
    per-column normalisation), rasters, membrane state."""
    import torch.multiprocessing as mp
    import oracle
    from test_oracle_fullsize import two_state_cols
    P, spikes, W0 = _column_case()
    st = two_state_cols(P, W0)
    ras = oracle.run_two_layer(P, st, spikes)
    assert ras.sum() > 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_column_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        W, R, V = q.get(timeout=120)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    np.testing.assert_array_equal(W.view(np.uint32), st["W"].view(np.uint32))
    np.testing.assert_array_equal(R, ras)
    np.testing.assert_array_equal(V.view(np.uint32), st["vY"].view(np.uint32))


def test_column_shard_builds_the_slice_network():
    import torch
    from bindsnet_amd.learning import MSTDP
    from bindsnet_amd.models import TwoLayerNetwork
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    from bindsnet_amd.parallel import column_shard, column_shard_bounds
    assert [column_shard_bounds(1600, 8, r) for r in range(8)][-1] == (1408, 1600)
    assert column_shard_bounds(100, 4, 2) == (64, 100) and column_shard_bounds(100, 4, 3) == (100, 100)
    torch.manual_seed(0)
    net = TwoLayerNetwork(n_inpt=784, n_neurons=256, reduction=torch.sum)
    covered = []
    for r in range(3):
        s, lo, hi = column_shard(net, r, 3)
        c = s.connections[("X", "Y")]
        assert torch.equal(c.w, net.connections[("X", "Y")].w[:, lo:hi]) and s.layers["Y"].n == hi - lo
        assert type(c.update_rule).__name__ == "PostPre" and c.norm == 78.4 and float(c.wmax) == 1.0
        covered += list(range(lo, hi))
    assert covered == list(range(256))
    net2 = Network(dt=1.0)
    net2.add_layer(Input(n=6400, shape=(1, 80, 80), traces=True), "X")
    net2.add_layer(LIFNodes(n=500, traces=True), "Y")
    net2.add_connection(Connection(net2.layers["X"], net2.layers["Y"], wmin=0, wmax=1, update_rule=MSTDP, nu=1e-1, norm=3200.0,
                                   reduction=torch.sum), "X", "Y")
    s, lo, hi = column_shard(net2, 1, 2)
    assert (lo, hi) == (256, 500) and type(s.connections[("X", "Y")].update_rule).__name__ == "MSTDP"
    from bindsnet_amd.models import DiehlAndCook2015
    with pytest.raises(NotImplementedError):
        column_shard(DiehlAndCook2015(784, n_neurons=64), 0, 2)


# ------------------------------------------------------------------------------------------------ the real thing on CPU
def _real_worker(rank, world, port, q):
    """One rank of the north-star schedule with NOTHING stubbed: its own batch shard of a D&C network on the host
    (network/host_path.py), bindsnet_amd.parallel.sharded_run for two consecutive inputs, gloo all-reduce of the deltas."""
    import os, sys
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import numpy as np
    import torch
    import torch.distributed as dist
    import synth
    from bindsnet_amd import parallel
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        N, per, T = 100, 3, 60
        torch.manual_seed(0)
        net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
        feat = net.connections[("X", "Ae")].pipeline[0]
        feat.value.data.copy_(torch.from_numpy(synth.weights_q12(10, 784, N)))
        mon = Monitor(net.layers["Ae"], ["s"], time=T)
        net.add_monitor(mon, "Ae_s")
        torch.manual_seed(2 + rank)
        out = {}
        for k in range(2):
            shard = synth.spike_train(20 + k, T, per * world, 784, max_rate=0.25)[:, rank * per:(rank + 1) * per]
            parallel.sharded_run(net, {"X": torch.from_numpy(np.ascontiguousarray(shard)).view(T, per, 1, 28, 28)}, T)
            assert net.last_plan == "host-torch"
            out[f"i{k}_sE"] = mon.get("s").numpy().reshape(T, per, N).astype(np.uint8).copy()
            out[f"i{k}_W"] = feat.value.detach().numpy().copy()
            out[f"i{k}_theta"] = net.layers["Ae"].theta.numpy().copy()
            net.reset_state_variables()
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_run_world2_gloo_real_shards_vs_oracle():
    """World size 2 over gloo with real shards (host path) and the real sharded_run: every rank's rasters equal the oracle
    on its shard from the merged weights / theta of the previous input, the merged tensors equal the numpy merge -- the
    CPU twin of tests/test_gpu_parallel.py::test_two_ranks_on_one_gpu_gloo."""
    import sys
    sys.path[:0] = [os.path.join(ROOT, "tests")]
    import cases
    import oracle
    import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    N, per, T = 100, 3, 60
    W = synth.weights_q12(10, 784, N)
    theta = np.zeros(N, np.float32)
    P = oracle.eth_mnist_dc_params(N, per, T)
    P.norm = 0.0
    Q = [oracle.exp_noise(2 + r, per * N * T * 2 + 16) for r in range(2)]
    cur = [np.zeros(1, np.int64) for _ in range(2)]
    for k in range(2):
        Wg, thg = [], []
        full = synth.spike_train(20 + k, T, per * 2, 784, max_rate=0.25)
        for r in range(2):
            st = oracle.eth_mnist_dc_state(N, per, W.copy())
            st["theta"][:] = theta
            rasE, _ = oracle.run_dc2015(P, st, np.ascontiguousarray(full[:, r * per:(r + 1) * per]), Q[r], cur[r])
            np.testing.assert_array_equal(res[r][f"i{k}_sE"], rasE, err_msg=f"input {k} rank {r}")
            Wg.append(st["W_xe"]); thg.append(st["theta"])
        W = (W + ((Wg[0] - W) + (Wg[1] - W))).astype(np.float32)
        np.clip(W, 0.0, 1.0, out=W)
        oracle.normalize(W, 78.4, False)
        theta = (theta + ((thg[0] - theta) + (thg[1] - theta))).astype(np.float32)
        for r in range(2):
            np.testing.assert_array_equal(res[r][f"i{k}_W"].view(np.uint32), W.view(np.uint32), err_msg=f"input {k}: merged weights, rank {r}")
            np.testing.assert_array_equal(res[r][f"i{k}_theta"].view(np.uint32), theta.view(np.uint32), err_msg=f"input {k}: merged theta, rank {r}")


# ------------------------------------------------------------------------------------------------ exact mode, D&C graph
@pytest.mark.parametrize("world,fixture", [(2, "run_dc_n400_b4"), (4, "run_dc_n400_b4"), (3, "run_dc_n100_b3_busy")])
def test_exact_batch_sharded_mode_equals_the_reference_global_batch(world, fixture, tmp_path):
    """SURVEY 8(e) "exact" for the graph whose batch IS coupled (theta, one_spike row order, the PostPre batch sum):
    `world` processes over gloo, each with ITS rows of the batch, parallel.exact_run for two consecutive inputs (reset
    between) -- rows side by side == what the unmodified reference computed for the global batch in ONE process (fixtures of
    tests/golden/make_golden.py): rasters, weights, theta, membrane state, traces bit for bit, and every rank's host
    generator stands where the reference's does.  (`busy`: many crossings per step; world 3 = one sample per rank.)"""
    import exact_harness as H
    res = H.launch(world, fixture, "cpu", tmp_path)
    H.check_against_reference(res, fixture)


@pytest.mark.parametrize("world,fixture", [(2, "run_dc_n400_b4"), (3, "run_dc_n100_b3_busy")])
def test_exact_gathered_mode_equals_the_reference_global_batch(world, fixture, tmp_path):
    """exact_run(mode="gathered"): ONE all-gather of the inputs and the layer state per run, then the global batch through Network.run on
    every rank (here the host path; on the MI355X the resident kernel) -- rows side by side == the reference's single-process global batch,
    bit for bit, like the per-step schedule above, incl. the state carried from the first input into the second without a reset."""
    import exact_harness as H
    res = H.launch(world, fixture, "cpu", tmp_path, mode="gathered")
    H.check_against_reference(res, fixture)
    assert all(str(r["r0_plan"]).startswith("exact-gathered:") for r in res)


def test_exact_batch_sharded_mode_full_cfg2_world4(tmp_path):
    """The same at BASELINE cfg2's full size on BASELINE.md's stated input: DiehlAndCook2015 784 -> 400, B = 32 = 4 ranks x 8,
    T = 250, three consecutive inputs, against the reference's single-process run (full_cfg2_dc_n400_b32_poisson)."""
    import exact_harness as H
    name = "full_cfg2_dc_n400_b32_poisson"
    res = H.launch(4, name, "cpu", tmp_path, timeout=1500)
    H.check_against_reference(res, name)


def test_exact_mode_refuses_what_it_does_not_implement():
    from bindsnet_amd import parallel
    from bindsnet_amd.models import TwoLayerNetwork
    net = TwoLayerNetwork(n_inpt=64, n_neurons=32, reduction=torch.sum)
    with pytest.raises(NotImplementedError):
        parallel.exact_run(net, {"X": torch.zeros(3, 2, 64, dtype=torch.uint8)}, 3)


@pytest.mark.parametrize("learning", [True, False])
def test_exact_run_single_rank_equals_run_on_the_host(learning):
    """exact_run at world size 1 (no process group) against Network.run on the host, same seeds: voltage and spike monitors,
    learning on and off (test mode: no theta adaptation, no PostPre), two consecutive inputs with a reset between."""
    import synth
    from bindsnet_amd import parallel
    from bindsnet_amd.models import DiehlAndCook2015
    from bindsnet_amd.network.monitors import Monitor
    T, B, N = 40, 3, 64

    def make():
        torch.manual_seed(0)
        net = DiehlAndCook2015(n_inpt=784, n_neurons=N, exc=22.5, inh=17.5, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
        net.connections[("X", "Ae")].pipeline[0].value.data.copy_(torch.from_numpy(synth.weights_q12(10, 784, N)))
        mons = {("Ae", "s"): Monitor(net.layers["Ae"], ["s", "v"], time=T), ("Ai", "s"): Monitor(net.layers["Ai"], ["s", "v"], time=T),
                ("X", "s"): Monitor(net.layers["X"], ["s"], time=T)}
        for (l, _), m in mons.items():
            net.add_monitor(m, l)
        net.train(learning)
        return net, mons

    a, ma = make()
    b, mb = make()
    for r in range(2):
        sp = torch.from_numpy(synth.spike_train(40 + r, T, B, 784, max_rate=0.3)).view(T, B, 1, 28, 28)
        torch.manual_seed(5 + r)
        a.run({"X": sp.clone()}, time=T)
        torch.manual_seed(5 + r)
        parallel.exact_run(b, {"X": sp.clone()}, T)
        for key in ma:
            for var in ma[key].state_vars:
                assert torch.equal(ma[key].get(var), mb[key].get(var)), (r, key, var)
        assert int(ma[("Ae", "s")].get("s").sum()) > 5
        for name in ("Ae", "Ai", "X"):
            la, lb = a.layers[name], b.layers[name]
            for attr in ("v", "refrac_count", "x", "theta", "s"):
                if hasattr(la, attr) and isinstance(getattr(la, attr), torch.Tensor) and getattr(la, attr).numel():
                    assert torch.equal(getattr(la, attr).reshape(-1).float(), getattr(lb, attr).reshape(-1).float()), (r, name, attr)
        assert torch.equal(a.connections[("X", "Ae")].pipeline[0].value, b.connections[("X", "Ae")].pipeline[0].value)
        a.reset_state_variables()
        b.reset_state_variables()
