#!/usr/bin/env python3
"""Generate golden fixtures by running the UNMODIFIED reference (BindsNET @ /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

Writes tests/golden/*.npz.  Inputs come from tests/synth.py (numpy RandomState, frozen
streams) and are NOT stored -- the tests regenerate them; only reference OUTPUTS are stored
(spike rasters bit-packed, large weight matrices as sha256 + a strided sample).

Import recipe: SURVEY.md Appendix C (skip bindsnet/__init__.py, which needs torchvision etc.).
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402

REF = "/root/reference/bindsnet"
pkg = types.ModuleType("bindsnet")
pkg.__path__ = [REF]
sys.modules["bindsnet"] = pkg
import bindsnet.network  # noqa: E402  (must be first: import cycle)
from bindsnet.learning import MSTDP, PostPre  # noqa: E402
from bindsnet.learning.MCC_learning import PostPre as MCCPostPre  # noqa: E402
from bindsnet.models import DiehlAndCook2015, TwoLayerNetwork  # noqa: E402
from bindsnet.network import Network  # noqa: E402
from bindsnet.network.monitors import Monitor  # noqa: E402
from bindsnet.network.nodes import DiehlAndCookNodes, Input, LIFNodes  # noqa: E402
from bindsnet.network.topology import Connection, Conv2dConnection, MulticompartmentConnection  # noqa: E402
from bindsnet.network.topology_features import Weight  # noqa: E402

torch.set_num_threads(8)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def pack(out: dict, key: str, arr: np.ndarray, limit: int = 48 * 1024) -> None:
    """Store small arrays whole; big ones as sha256 + a strided sample (tests check both)."""
    arr = np.ascontiguousarray(arr)
    if arr.nbytes <= limit:
        out[key] = arr
    else:
        out[key + "_sha"] = sha(arr)
        step = max(1, arr.size // 4096)
        out[key + "_sample"] = arr.reshape(-1)[::step].copy()


# ----------------------------------------------------------------------------- op-level
PROP_CASES = [  # (B, Nin, N, p)
    (3, 784, 100, 0.3), (2, 784, 400, 0.5), (2, 1000, 37, 0.4), (1, 5000, 70, 0.6),
    (4, 400, 400, 0.05), (2, 33, 15, 0.5), (2, 64, 8, 0.5), (1, 20, 33, 0.9),
]


def gen_prop():
    out = {}
    for k, (B, Nin, N, p) in enumerate(PROP_CASES):
        W = synth.uniform_f32(100 + k, (Nin, N), -1.0, 1.0)
        s = synth.dense_spikes(200 + k, (B, Nin), p)
        src, tgt = Input(n=Nin), LIFNodes(n=N)
        conn = MulticompartmentConnection(src, tgt, device="cpu", pipeline=[Weight("weight", value=T_(W).clone())])
        out[f"out{k}"] = conn.compute(T_(s)).numpy()
        # bool spikes (what non-input layers carry) must give the same numbers
        assert torch.equal(conn.compute(T_(s).bool()), T_(out[f"out{k}"]))
    save("op_prop_mcc", cases=np.array(PROP_CASES), **out)


PP_CASES = [  # (B, Nin, N)
    (1, 40, 24), (3, 40, 24), (16, 40, 24), (32, 40, 24), (48, 40, 24), (37, 7, 9), (32, 784, 100), (5, 33, 31),
]


def _set_layer(layer, B, s, x):
    layer.batch_size = B
    layer.s = T_(s)
    if x is not None:
        layer.x = T_(x).clone()


def gen_postpre():
    out = {}
    for k, (B, Nin, N) in enumerate(PP_CASES):
        W0 = synth.uniform_f32(300 + k, (Nin, N), 0.0, 1.0)
        s_src = synth.dense_spikes(400 + k, (B, Nin), 0.3)
        s_tgt = synth.dense_spikes(500 + k, (B, N), 0.2)
        x_src = synth.uniform_f32(600 + k, (B, Nin), 0.0, 1.0)
        x_tgt = synth.uniform_f32(700 + k, (B, N), 0.0, 1.0)
        # --- MCC PostPre (MCC_learning.py:224-302)
        src, tgt = Input(n=Nin, traces=True), LIFNodes(n=N, traces=True)
        conn = MulticompartmentConnection(
            src, tgt, device="cpu",
            pipeline=[Weight("weight", value=T_(W0).clone(), range=[0.0, 1.0], nu=(1e-4, 1e-2),
                             learning_rule=MCCPostPre)])
        conn.dt = 1.0
        _set_layer(src, B, s_src, x_src)
        _set_layer(tgt, B, s_tgt.astype(bool), x_tgt)
        conn.update(learning=True)
        pack(out, f"mcc{k}", conn.pipeline[0].value.detach().numpy())
        # --- dense PostPre (learning.py:390-420), wmin/wmax finite
        src, tgt = Input(n=Nin, traces=True), LIFNodes(n=N, traces=True)
        conn = Connection(src, tgt, w=T_(W0).clone(), update_rule=PostPre, nu=(1e-4, 1e-2),
                          reduction=torch.sum, wmin=0.0, wmax=1.0)
        _set_layer(src, B, s_src, x_src)
        _set_layer(tgt, B, s_tgt.astype(bool), x_tgt)
        conn.update(learning=True)
        pack(out, f"dense{k}", conn.w.detach().numpy())
    save("op_postpre", cases=np.array(PP_CASES), **out)


NORM_CASES = [(784, 100), (784, 400), (1000, 37), (20, 33), (6400, 50)]


def gen_normalize():
    out = {}
    for k, (Nin, N) in enumerate(NORM_CASES):
        W0 = synth.uniform_f32(800 + k, (Nin, N), -0.2, 1.0)
        W0[:, N // 2] = 0.0  # a zero column (colsum == 0 -> 1 path)
        src, tgt = Input(n=Nin), LIFNodes(n=N)
        f = Weight("weight", value=T_(W0).clone(), norm=78.4)
        MulticompartmentConnection(src, tgt, device="cpu", pipeline=[f])
        f.normalize()
        pack(out, f"mcc{k}", f.value.detach().numpy())
        conn = Connection(src, tgt, w=T_(W0).clone(), norm=78.4)
        conn.normalize()
        pack(out, f"dense{k}", conn.w.detach().numpy())
    save("op_normalize", cases=np.array(NORM_CASES), **out)


def gen_nodes():
    """LIFNodes / DiehlAndCookNodes / Input forward over a few steps with random currents."""
    B, N, T = 5, 70, 40
    out = {}
    I = synth.uniform_f32(900, (T, B, N), -2.0, 6.0)
    net = Network(dt=1.0, batch_size=B)
    lif = LIFNodes(n=N, traces=True, rest=-60.0, reset=-45.0, thresh=-40.0, tc_decay=10.0, refrac=2,
                   tc_trace=20.0, lbound=-62.0)
    net.add_layer(lif, "Y")
    ras, vs = [], []
    for t in range(T):
        lif.forward(T_(I[t]).clone())
        ras.append(lif.s.numpy().copy()); vs.append(lif.v.numpy().copy())
    out.update(lif_s=np.packbits(np.array(ras)), lif_v=np.array(vs)[-1], lif_x=lif.x.numpy().copy(),
               lif_r=lif.refrac_count.numpy().copy(), lif_decay=lif.decay.numpy(),
               lif_trace_decay=lif.trace_decay.numpy())
    # additive traces variant
    net = Network(dt=1.0, batch_size=B)
    lif = LIFNodes(n=N, traces=True, traces_additive=True, trace_scale=0.5, tc_trace=15.0)
    net.add_layer(lif, "Y")
    for t in range(T):
        lif.forward(T_(I[t] * 3).clone())
    out.update(lifadd_v=lif.v.numpy().copy(), lifadd_x=lif.x.numpy().copy(), lifadd_decay=lif.decay.numpy(),
               lifadd_trace_decay=lif.trace_decay.numpy())
    # D&C nodes with one_spike; count multinomial draws
    net = Network(dt=1.0, batch_size=B)
    dc = DiehlAndCookNodes(n=N, traces=True, rest=-65.0, reset=-60.0, thresh=-52.0, refrac=5, tc_decay=100.0,
                           tc_trace=20.0, theta_plus=0.05, tc_theta_decay=1e7)
    net.add_layer(dc, "E")
    torch.manual_seed(77)
    ras = []
    for t in range(T):
        dc.forward(T_(I[t] * 2.0).clone())
        ras.append(dc.s.numpy().copy())
    st = torch.get_rng_state()
    torch.manual_seed(77)
    # count draws: advance a fresh stream until the state matches
    consumed = 0
    while not torch.equal(torch.get_rng_state(), st):
        torch.empty(N).exponential_(1); consumed += N
        assert consumed < B * N * T + N
    out.update(dc_s=np.packbits(np.array(ras)), dc_v=dc.v.numpy().copy(), dc_x=dc.x.numpy().copy(),
               dc_r=dc.refrac_count.numpy().copy(), dc_theta=dc.theta.numpy().copy(),
               dc_decay=dc.decay.numpy(), dc_theta_decay=dc.theta_decay.numpy(),
               dc_trace_decay=dc.trace_decay.numpy(), dc_consumed=np.int64(consumed))
    save("op_nodes", B=B, N=N, T=T, **out)


def gen_conv():
    out = {}
    cases = [(4, 1, 28, 28, 32, 5, 1, 0), (2, 1, 12, 12, 4, 3, 2, 1), (2, 3, 10, 10, 5, 3, 1, 0),
             (2, 8, 8, 8, 16, 3, 1, 1), (2, 16, 8, 8, 32, 5, 1, 2), (3, 4, 9, 11, 7, 3, 2, 1)]
    for k, (B, Cin, H, Wd, Cout, K, stride, pad) in enumerate(cases):
        W = synth.uniform_f32(1000 + k, (Cout, Cin, K, K), 0.0, 0.3)
        s = synth.dense_spikes(1100 + k, (B, Cin, H, Wd), 0.2)
        OH = (H + 2 * pad - K) // stride + 1
        OW = (Wd + 2 * pad - K) // stride + 1
        src, tgt = Input(shape=(Cin, H, Wd)), LIFNodes(shape=(Cout, OH, OW))
        conn = Conv2dConnection(src, tgt, kernel_size=K, stride=stride, padding=pad, w=T_(W).clone())
        pack(out, f"out{k}", conn.compute(T_(s)).numpy())
    save("op_conv2d", cases=np.array(cases), **out)


def gen_rng():
    torch.manual_seed(1234)
    st = torch.get_rng_state().numpy().copy()
    x = torch.empty(3000).exponential_(1).numpy()
    st2 = torch.get_rng_state().numpy().copy()
    save("op_rng", state0=st, draws=x, state1=st2)


# ----------------------------------------------------------------------------- full runs
def dc_case(name, N, B, T, runs, full_w, inh=120.0, max_rate=0.0625, dt=1.0):
    """T TIMESTEPS of length dt per run (network.py:376: timesteps = int(time / dt))."""
    Nin = 784
    torch.manual_seed(0)
    net = DiehlAndCook2015(n_inpt=Nin, n_neurons=N, exc=22.5, inh=inh, dt=dt, norm=78.4, theta_plus=0.05,
                           inpt_shape=(1, 28, 28))
    feat = net.connections[("X", "Ae")].pipeline[0]
    W0 = synth.weights_q12(10, Nin, N)
    feat.value.data.copy_(T_(W0))
    mons = {}
    for l in ("Ae", "Ai"):
        mons[l] = Monitor(net.layers[l], ["s"], time=T)
        net.add_monitor(mons[l], l + "_s")
    counter = {"n": 0}
    orig = torch.multinomial

    def counting(p, n, *a, **k):
        counter["n"] += p.numel()
        return orig(p, n, *a, **k)

    torch.multinomial = counting
    out = {}
    Ae, Ai, X = net.layers["Ae"], net.layers["Ai"], net.layers["X"]
    for r in range(runs):
        spikes = synth.spike_train(20 + r, T, B, Nin, max_rate=max_rate)
        torch.manual_seed(2 + r)
        counter["n"] = 0
        net.run({"X": T_(spikes).view(T, B, 1, 28, 28)}, time=T * dt)
        out[f"r{r}_sE"] = np.packbits(mons["Ae"].get("s").numpy().astype(np.uint8))
        out[f"r{r}_sI"] = np.packbits(mons["Ai"].get("s").numpy().astype(np.uint8))
        out[f"r{r}_consumed"] = np.int64(counter["n"])
        W = feat.value.detach().numpy()
        out[f"r{r}_W_sha"] = sha(W)
        out[f"r{r}_W_sample"] = W.reshape(-1)[::97].copy()
        out[f"r{r}_theta"] = Ae.theta.numpy().copy()
        out[f"r{r}_vE"] = Ae.v.numpy().copy(); out[f"r{r}_rE"] = Ae.refrac_count.numpy().copy()
        out[f"r{r}_xE"] = Ae.x.numpy().copy(); out[f"r{r}_xX"] = X.x.numpy().reshape(B, Nin).copy()
        out[f"r{r}_vI"] = Ai.v.numpy().copy(); out[f"r{r}_rI"] = Ai.refrac_count.numpy().copy()
        if full_w and r == runs - 1:
            out["W_final"] = W.copy()
        print(f"  {name} run {r}: exc spikes {int(mons['Ae'].get('s').sum())}, inh {int(mons['Ai'].get('s').sum())},"
              f" draws {counter['n']}")
        if r % 2 == 0:
            net.reset_state_variables()   # eth_mnist.py:276 does this after every sample
    torch.multinomial = orig
    consts = dict(
        x_trace_decay=X.trace_decay.numpy(), e_decay=Ae.decay.numpy(), e_theta_decay=Ae.theta_decay.numpy(),
        e_trace_decay=Ae.trace_decay.numpy(), i_decay=Ai.decay.numpy())
    extra = {} if dt == 1.0 else {"dt": np.float32(dt)}      # (the dt = 1 fixtures keep their original key set)
    save(name, N=N, B=B, T=T, runs=runs, inh=np.float32(inh), max_rate=np.float64(max_rate), **extra, **consts, **out)


def two_layer_case(name, rule, Nin, N, B, T):
    """Input -> Connection -> LIFNodes with learning.PostPre / MSTDP (dense family)."""
    torch.manual_seed(0)
    W0 = synth.weights_q12(11, Nin, N)
    if rule == "postpre":
        net = TwoLayerNetwork(n_inpt=Nin, n_neurons=N, reduction=torch.sum, norm=78.4 * Nin / 784)
        conn = net.connections[("X", "Y")]
        conn.w.data.copy_(T_(W0))
    else:
        net = Network(dt=1.0)
        net.add_layer(Input(n=Nin, traces=True), "X")
        net.add_layer(LIFNodes(n=N, traces=True), "Y")
        conn = Connection(net.layers["X"], net.layers["Y"], w=T_(W0).clone(), wmin=0, wmax=1, update_rule=MSTDP,
                          nu=1e-1, norm=0.1 * Nin, reduction=torch.sum)
        net.add_connection(conn, "X", "Y")
    mon = Monitor(net.layers["Y"], ["s"], time=T)
    net.add_monitor(mon, "Y_s")
    forced = []
    orig_compute = conn.compute

    def capture(s):
        o = orig_compute(s)
        forced.append(o.detach().numpy().copy())
        return o

    conn.compute = capture
    spikes = synth.spike_train(30, T, B, Nin, active=0.3, max_rate=0.12)
    kw = {"reward": 1.0} if rule == "mstdp" else {}
    net.run({"X": T_(spikes)}, time=T, **kw)
    Y, X = net.layers["Y"], net.layers["X"]
    out = dict(sY=np.packbits(mon.get("s").numpy().astype(np.uint8)), W=conn.w.detach().numpy().copy(),
               I_forced=np.array(forced).astype(np.float32), vY=Y.v.numpy().copy(), xY=Y.x.numpy().copy(),
               xX=X.x.numpy().copy(), decay=Y.decay.numpy(), y_trace_decay=Y.trace_decay.numpy(),
               x_trace_decay=X.trace_decay.numpy())
    if rule == "mstdp":
        ur = conn.update_rule
        out.update(p_plus=ur.p_plus.numpy().copy(), p_minus=ur.p_minus.numpy().copy(),
                   elig_sha=sha(ur.eligibility.numpy()),
                   decay_plus=torch.exp(-torch.tensor(1.0) / ur.tc_plus).numpy(),
                   decay_minus=torch.exp(-torch.tensor(1.0) / ur.tc_minus).numpy())
    print(f"  {name}: Y spikes {int(mon.get('s').sum())}")
    save(name, Nin=Nin, N=N, B=B, T=T, **out)


if __name__ == "__main__":
    # Op-level fixtures pin ATen's SERIAL summation order (torch.set_num_threads(1)): with >1 thread
    # and >= 32768 summands ATen's parallel_dim_reduction splits the output columns at 128-byte
    # boundaries, which moves 1..7 leftover tail columns (N % 32 in 1..7) from row_sum to the
    # 4-column multi_row_sum path depending on the thread count (probe: N=37 differs between 1 and
    # 8 threads; N=100 differs at 9 or 12 threads; N % 32 == 0 or >= 8 -- 400, 1600 -- never differs).
    # The full runs below use 8 threads like the rest of the survey; for their shapes (N=100, 400)
    # the 8-thread order equals the serial order.
    torch.set_num_threads(1)
    gen_prop()
    gen_postpre()
    gen_normalize()
    gen_nodes()
    gen_conv()
    gen_rng()
    torch.set_num_threads(8)
    dc_case("run_dc_n100_b1", 100, 1, 100, 3, True)
    dc_case("run_dc_n100_b3", 100, 3, 60, 2, True)
    dc_case("run_dc_n100_b3_busy", 100, 3, 60, 2, False, inh=17.5, max_rate=0.25)
    dc_case("run_dc_n400_b4", 400, 4, 50, 2, False)
    dc_case("run_dc_n400_b32", 400, 32, 40, 2, False)
    two_layer_case("run_two_postpre_b4", "postpre", 196, 64, 4, 60)
    two_layer_case("run_two_postpre_b32", "postpre", 196, 64, 32, 40)
    two_layer_case("run_two_mstdp_b4", "mstdp", 196, 48, 4, 40)
