"""The N > 1 path of bench.py on ONE GPU: a real RCCL process group of world size 1.  Checks that
bindsnet_amd.parallel.sharded_run drives the device run (resident plan, normalisation postponed), the
all-reduce / barrier calls work on device tensors, and that with a single rank the result equals a plain
run() up to the rounding of `before + (after - before)`."""
import os
import socket

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def build():
    from bindsnet_amd.models import DiehlAndCook2015
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=784, n_neurons=100, exc=22.5, inh=120, dt=1.0, norm=78.4, theta_plus=0.05, inpt_shape=(1, 28, 28))
    net.connections[("X", "Ae")].pipeline[0].value.data.copy_(torch.from_numpy(synth.weights_q12(10, 784, 100)))
    net.to(DEV)
    return net


def test_sharded_run_single_rank_rccl():
    import torch.distributed as dist
    from bindsnet_amd import parallel
    T, B = 60, 8
    spikes = [torch.from_numpy(synth.spike_train(20 + r, T, B, 784)).view(T, B, 1, 28, 28).to(DEV) for r in range(2)]
    ref = build()
    for r in range(2):
        torch.manual_seed(3 + r)
        ref.run({"X": spikes[r]}, time=T)
    saved = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE")}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    except Exception as e:                                  # no usable RCCL on this box: an infrastructure matter
        pytest.skip(f"RCCL process group could not be created: {e}")
    try:
        net = build()
        kept = []
        for r in range(2):
            torch.manual_seed(3 + r)
            parallel.sharded_run(net, {"X": spikes[r]}, T)
            assert net.last_plan.startswith("dc2015-resident")
            assert net.__dict__["_run_cache"]["defer_norm"] is True
            kept.append(net.__dict__["_run_cache"]["L"])
        assert kept[0] is kept[1], "consecutive sharded runs must re-use the kept descriptor arrays"
        dist.barrier()
        t = torch.ones(4, device=DEV)
        dist.all_reduce(t)
        assert float(t.sum()) == 4.0
    finally:
        dist.destroy_process_group()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    Wr = ref.connections[("X", "Ae")].pipeline[0].value.cpu().numpy()
    Ws = net.connections[("X", "Ae")].pipeline[0].value.cpu().numpy()
    np.testing.assert_allclose(Ws, Wr, rtol=0, atol=2e-6)
    np.testing.assert_allclose(Ws.sum(0), 78.4, rtol=1e-5)
    np.testing.assert_allclose(net.layers["Ae"].theta.cpu().numpy(), ref.layers["Ae"].theta.cpu().numpy(), rtol=0, atol=1e-6)
    assert net.connections[("X", "Ae")].pipeline[0].norm == 78.4


@pytest.mark.parametrize("kind", ["postpre", "mstdp"])
def test_column_shards_on_the_device_equal_the_full_network(kind):
    """The exact multi-GPU mode, executed for real on the one GPU there is: the column slices `column_shard` builds are
    run one after the other (each on the whole batch, two consecutive inputs) and put side by side -- weights after
    normalisation, rasters and membrane state must equal the unsharded network's, bit for bit."""
    from bindsnet_amd.learning import MSTDP
    from bindsnet_amd.models import TwoLayerNetwork
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    from bindsnet_amd.parallel import column_shard
    T, B, Nin, N = 40, 24, 784, 352                 # 11 blocks of 32 columns over 3 ranks: slices of 128, 128, 96

    def make():
        torch.manual_seed(0)
        if kind == "postpre":
            net = TwoLayerNetwork(n_inpt=Nin, n_neurons=N, reduction=torch.sum)
        else:
            net = Network(dt=1.0)
            net.add_layer(Input(n=Nin, traces=True), "X")
            net.add_layer(LIFNodes(n=N, traces=True), "Y")
            net.add_connection(Connection(net.layers["X"], net.layers["Y"], w=0.3 * torch.rand(Nin, N), wmin=0, wmax=1, update_rule=MSTDP,
                                          nu=1e-1, norm=0.1 * Nin, reduction=torch.sum), "X", "Y")
        return net

    kw = {"reward": 1.0} if kind == "mstdp" else {}
    inputs = [torch.from_numpy(synth.dense_spikes(40 + r, (T, B, Nin), 0.03)).to(DEV) for r in range(2)]

    def run(net):
        mon = Monitor(net.layers["Y"], ["s"], time=T)
        net.add_monitor(mon, "Y_s")
        net.to(DEV)
        out = []
        for x in inputs:
            net.run({"X": x}, time=T, **kw)
            out.append((mon.get("s").reshape(T, B, -1).clone(), net.connections[("X", "Y")].w.detach().clone(), net.layers["Y"].v.clone()))
        return out

    full = run(make())
    world = 3
    parts = []
    for r in range(world):
        shard, lo, hi = column_shard(make(), r, world)
        parts.append(run(shard))
    for k in range(2):
        for idx, name in ((0, "raster"), (1, "weights"), (2, "membrane")):
            got = torch.cat([p[k][idx] for p in parts], dim=-1)
            assert torch.equal(got, full[k][idx]), f"input {k}: {name}"
    assert int(full[1][0].sum()) > 100


def test_column_shards_of_a_network_in_mid_training_carry_the_rule_state():
    """An MSTDP rule keeps p_plus / p_minus / the previous spikes across run() and reset_state_variables(); sharding a network
    that has already run must carry them (advisor finding, round 2): input 0 on the full network, THEN shard, input 1 on the
    shards -- weights, rasters and the rule's state equal the full network's second run bit for bit."""
    from bindsnet_amd.learning import MSTDP
    from bindsnet_amd.network import Network
    from bindsnet_amd.network.monitors import Monitor
    from bindsnet_amd.network.nodes import Input, LIFNodes
    from bindsnet_amd.network.topology import Connection
    from bindsnet_amd.parallel import column_shard
    T, B, Nin, N = 40, 24, 784, 352

    def make():
        torch.manual_seed(0)
        net = Network(dt=1.0)
        net.add_layer(Input(n=Nin, traces=True), "X")
        net.add_layer(LIFNodes(n=N, traces=True), "Y")
        net.add_connection(Connection(net.layers["X"], net.layers["Y"], w=0.3 * torch.rand(Nin, N), wmin=0, wmax=1, update_rule=MSTDP,
                                      nu=1e-1, norm=0.1 * Nin, reduction=torch.sum), "X", "Y")
        return net.to(DEV)

    inputs = [torch.from_numpy(synth.dense_spikes(50 + r, (T, B, Nin), 0.03)).to(DEV) for r in range(2)]
    full = make()
    mon = Monitor(full.layers["Y"], ["s"], time=T)
    full.add_monitor(mon, "Y_s")
    full.run({"X": inputs[0].clone()}, time=T, reward=1.0)      # (clones: reset_state_variables() zeroes the last slice of the tensor Input.s aliases)
    full.reset_state_variables()
    base = make()
    base.run({"X": inputs[0].clone()}, time=T, reward=1.0)
    base.reset_state_variables()
    assert float(base.connections[("X", "Y")].update_rule.p_plus.abs().sum()) > 0      # the rule holds state now
    full.run({"X": inputs[1].clone()}, time=T, reward=1.0)
    parts = []
    for r in range(3):
        shard, lo, hi = column_shard(base, r, 3)
        m = Monitor(shard.layers["Y"], ["s"], time=T)
        shard.add_monitor(m, "Y_s")
        shard.run({"X": inputs[1].clone()}, time=T, reward=1.0)
        rule = shard.connections[("X", "Y")].update_rule
        parts.append((m.get("s").reshape(T, B, -1).clone(), shard.connections[("X", "Y")].w.detach().clone(), rule.p_minus.clone(), rule.p_plus.clone()))
    fr = full.connections[("X", "Y")].update_rule
    assert torch.equal(torch.cat([p[0] for p in parts], dim=-1), mon.get("s").reshape(T, B, -1)), "rasters"
    assert torch.equal(torch.cat([p[1] for p in parts], dim=-1), full.connections[("X", "Y")].w.detach()), "weights"
    assert torch.equal(torch.cat([p[2] for p in parts], dim=-1), fr.p_minus), "p_minus"
    for p_ in parts:
        assert torch.equal(p_[3], fr.p_plus), "p_plus"


def test_native_rccl_communicator_single_rank():
    """include/snnhip.h snn_dist_*: the C ABI's own RCCL collectives (no torch.distributed), world size 1."""
    from bindsnet_amd.parallel import NativeComm
    from bindsnet_amd._lib import SnnError
    try:
        comm = NativeComm(0, 1)
    except SnnError as e:
        pytest.skip(f"RCCL communicator could not be created on this box: {e}")
    try:
        t = torch.arange(1000, dtype=torch.float32, device=DEV)
        comm.allreduce_(t)
        g = comm.allgather(torch.arange(64, dtype=torch.uint8, device=DEV))
        torch.cuda.synchronize()
        assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32)) and tuple(g.shape) == (1, 64)
        assert torch.equal(g[0].cpu(), torch.arange(64, dtype=torch.uint8))
    finally:
        comm.close()


# ---------------------------------------------------------------------------------------------------------------
# N = 2 for real: two processes, each with its own batch shard ON THE DEVICE, a torch.distributed group between them,
# bindsnet_amd.parallel.sharded_run per input -- against the oracle run per shard + the merge in numpy.
def _two_rank_run(backend, tmp_path):
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "parallel_worker.py"), "--rank", str(r), "--world", "2", "--port", str(port),
                               "--backend", backend, "--out", outs[r]], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=420)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            p.kill()
            logs.append("timeout")
    return [p.returncode for p in procs], logs, outs


def _check_two_ranks(outs):
    import cases
    import oracle
    name = "full_cfg2_dc_n400_b32_poisson"
    g = cases.gold(name)
    N, B, T = int(g["N"]), int(g["B"]), int(g["T"])
    per = B // 2
    res = [np.load(o) for o in outs]
    torch.manual_seed(0)
    W = (0.3 * torch.rand(784, N)).numpy()                        # the replicas' start (models.py:184 after manual_seed(0))
    theta = np.zeros(N, np.float32)
    P = oracle.eth_mnist_dc_params(N, per, T)
    P.norm = 0.0                                                  # the shards do not normalise: the merged weights are (below)
    Q = [oracle.exp_noise(2 + r, per * N * T * 2) for r in range(2)]
    cur = [np.zeros(1, np.int64) for _ in range(2)]
    for k in range(2):
        Wg, thg = [], []
        for r in range(2):
            st = oracle.eth_mnist_dc_state(N, per, W.copy())
            st["theta"][:] = theta
            shard = np.ascontiguousarray(cases.fixture_input(g, k, T, B)[:, r * per:(r + 1) * per])
            rasE, rasI = oracle.run_dc2015(P, st, shard, Q[r], cur[r])
            np.testing.assert_array_equal(np.unpackbits(res[r][f"i{k}_sE"])[:T * per * N].reshape(T, per, N), rasE, err_msg=f"input {k} rank {r} Ae raster")
            np.testing.assert_array_equal(np.unpackbits(res[r][f"i{k}_sI"])[:T * per * N].reshape(T, per, N), rasI, err_msg=f"input {k} rank {r} Ai raster")
            Wg.append(st["W_xe"]); thg.append(st["theta"])
            assert str(res[r][f"i{k}_plan"]).startswith("dc2015-resident")
        # the merge of parallel.sharded_run in numpy f32: before + sum_over_ranks(after - before), clamp, normalise
        W = (W + ((Wg[0] - W) + (Wg[1] - W))).astype(np.float32)
        np.clip(W, 0.0, 1.0, out=W)
        oracle.normalize(W, 78.4, False)
        theta = (theta + ((thg[0] - theta) + (thg[1] - theta))).astype(np.float32)
        for r in range(2):
            np.testing.assert_array_equal(res[r][f"i{k}_W"].view(np.uint32), W.view(np.uint32), err_msg=f"input {k}: merged weights on rank {r}")
            np.testing.assert_array_equal(res[r][f"i{k}_theta"].view(np.uint32), theta.view(np.uint32), err_msg=f"input {k}: merged theta on rank {r}")


def test_two_ranks_on_one_gpu_gloo(tmp_path):
    """World size 2 on the one GPU of the box (gloo between the processes, the shards and the merged tensors on the
    device): B = 16 + 16 of the cfg2 Poisson fixture, two consecutive inputs.  Every rank's rasters = the oracle on its
    shard from the merged weights / theta of the previous input, bit for bit; merged W and theta = the numpy merge."""
    rcs, logs, outs = _two_rank_run("gloo", tmp_path)
    assert rcs == [0, 0], "\n".join(logs)[-3000:]
    _check_two_ranks(outs)


def test_two_ranks_on_one_gpu_rccl(tmp_path):
    """The same through RCCL (backend nccl) with both ranks on device 0 -- if RCCL accepts two ranks on one device;
    where it refuses (duplicate GPU), that is reported as a skip with RCCL's own message."""
    rcs, logs, outs = _two_rank_run("nccl", tmp_path)
    if rcs != [0, 0]:
        tail = "\n".join(logs)[-1500:]
        pytest.skip("RCCL did not form a 2-rank group on one device: " + tail.replace("\n", " | ")[-600:])
    _check_two_ranks(outs)



# ---------------------------------------------------------------------------------------------------- bench.py --gpus N, end to end
def _bench(args, env_extra=None, timeout=400):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    return res, (json.loads(lines[-1]) if lines else None)


def test_bench_two_ranks_on_one_gpu_gloo_end_to_end():
    """`python bench.py --gpus 2 --backend gloo`: the launcher re-exec, the rendezvous on 127.0.0.1, sharded_run inside a pipelined
    section on both ranks, the merge collective, barrier + MAX of the elapsed times and rank 0's ONE JSON line -- everything the
    driver's N-GPU command goes through except RCCL itself, with both ranks on the one GPU of the box."""
    res, line = _bench(["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert res.returncode == 0 and line is not None, (res.stdout[-1500:], res.stderr[-3000:])
    assert line.get("error") is None, line
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["config"]["global_batch"] == 64 and line["config"]["backend"] == "gloo"
    assert abs(line["value"] - 2 * 3 * 250 / (line["ms_per_step"] * 3e-3)) / line["value"] < 1e-3          # whole-job aggregate over both ranks
    assert len(line["devices"]) == 2 and line["config"]["per_input_collective"]["bytes"] > 0
    assert line["config"]["plan"].startswith("dc2015-resident")


def test_bench_prints_an_error_line_when_a_stage_hangs():
    """A rank that never shows up: rank 0's rendezvous cannot complete; the watchdog (here 8 s) prints a JSON line with `error` and
    `stage` instead of hanging until the driver's clock runs out."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    res, line = _bench(["--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--watchdog", "8"],
                       env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}, timeout=200)
    assert res.returncode != 0 and line is not None, (res.stdout[-1500:], res.stderr[-2000:])
    assert line["value"] is None and "error" in line and line["stage"] == "init_process_group" and line["n_gpus"] == 2


@pytest.mark.parametrize("world", [1, 2])
def test_bench_cfg3_leg_both_multi_gpu_modes(world):
    """`python bench.py --config cfg3 --gpus N [--backend gloo]` (BASELINE.json configs[2]: TwoLayerNetwork 784 -> 1600, global batch 128): the
    north-star schedule (batch shards + one all-reduce of the deltas per input) as `value`, the exact column-sharded mode beside it, one JSON
    line of the headline's schema -- with N = 2 both ranks on the one GPU of the box (gloo), what an 8-GPU node runs with `--gpus 8`."""
    args = ["--config", "cfg3", "--gpus", str(world), "--steps", "3", "--warmup", "1"] + (["--backend", "gloo"] if world > 1 else [])
    res, line = _bench(args)
    assert res.returncode == 0 and line is not None, (res.stdout[-1500:], res.stderr[-3000:])
    assert line.get("error") is None, line
    assert line["n_gpus"] == world and line["steps"] == 3 and line["scaling"] == "strong" and line["unit"] == "timesteps/s"
    assert line["config"]["global_batch"] == 128 and line["config"]["batch_per_gpu"] == 128 // world and line["config"]["plan"] == "twolayer-fused"
    assert abs(line["value"] - 3 * 100 / (line["ms_per_step"] * 3e-3)) / line["value"] < 1e-3
    ex = line["exact_column_shard"]
    assert ex["value"] > 0 and ex["collectives_during_the_run"] == 0 and ex["plan"] == "twolayer-fused" and ex["batch_per_gpu"] == 128
    assert ex["columns_of_rank0"] == ([0, 1600] if world == 1 else [0, 800])
    if world > 1:
        assert line["config"]["per_input_collective"]["bytes"] == 784 * 1600 * 4
