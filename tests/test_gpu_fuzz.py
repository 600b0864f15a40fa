"""Randomised differential test: every fused / resident plan against the generic per-operator plan, bit for bit, on
seeded random shapes (column counts that do / do not leave ATen row_sum tail columns, batch 1..32, odd time lengths,
sparse and busy inputs, multi-valued spike bytes).  The generic plan itself is pinned to the oracle and the reference
fixtures by the other GPU tests."""
import os

import numpy as np
import pytest
import torch

import synth
import test_gpu_convlif as conv
import test_gpu_fused_stress as dc
import test_gpu_twolayer as two

pytestmark = pytest.mark.gpu
u8 = np.uint8
MORE = int(os.environ.get("SNN_FUZZ_MORE", "0"))      # extra seeds per family for a longer soak (default suite: 30 cases)


def _same(a, b, what):
    for r, (x, y) in enumerate(zip(a, b)):
        for k in x:
            np.testing.assert_array_equal(x[k].view(u8), y[k].view(u8), err_msg=f"{what} input {r}: {k}")


@pytest.mark.parametrize("seed", range(10 + MORE))
def test_fuzz_dc2015(seed):
    rs = np.random.RandomState(1000 + seed)
    N = int(rs.choice([8, 24, 32, 40, 64, 100, 200, 333, 400, 512, 600]))
    B = int(rs.choice([1, 2, 3, 8, 15, 16, 17, 31, 32]))
    T = int(rs.randint(3, 30))
    dens = float(rs.choice([0.005, 0.02, 0.05, 0.15, 0.4]))
    wsc = float(rs.choice([0.05, 0.3, 1.0]))
    vmax = int(rs.choice([1, 1, 1, 3]))
    learning = bool(rs.rand() < 0.8)
    inh = float(rs.choice([120.0, 17.5, 60.0]))
    spikes = []
    for r in range(2):
        s = synth.dense_spikes(500 + 7 * seed + r, (T, B, 784), dens)
        if vmax > 1:
            s = (s * rs.randint(1, vmax + 1, size=s.shape)).astype(u8)
        spikes.append(s)
    res, plan = dc.run(0, N, B, T, spikes, w_scale=wsc, learning=learning, inh=inh)
    assert plan.startswith("dc2015-resident")           # the lean form where it applies (repeated on the general one if it gives up)
    gres, plan_r = dc.run(3, N, B, T, spikes, w_scale=wsc, learning=learning, inh=inh)
    assert plan_r == "dc2015-resident"                   # the general resident kernel
    step, plan_s = dc.run(2, N, B, T, spikes, w_scale=wsc, learning=learning, inh=inh)
    assert plan_s == "dc2015-fused"
    gen, plan_g = dc.run(1, N, B, T, spikes, w_scale=wsc, learning=learning, inh=inh)
    assert plan_g == "generic"
    _same(res, gen, f"resident (auto) N={N} B={B} T={T} dens={dens}")
    _same(gres, gen, f"resident (general) N={N} B={B} T={T} dens={dens}")
    _same(step, gen, f"per-step N={N} B={B} T={T} dens={dens}")


@pytest.mark.parametrize("seed", range(8 + MORE))
def test_fuzz_twolayer(seed):
    rs = np.random.RandomState(2000 + seed)
    kind = str(rs.choice(["dense", "mcc"]))
    Nin = int(rs.choice([16, 64, 256, 784, 1024, 2048]))
    N = int(rs.choice([5, 32, 37, 64, 100, 257]))
    B = int(rs.choice([1, 4, 7, 16, 23, 32]))
    T = int(rs.randint(4, 30))
    rule = bool(rs.rand() < 0.75)
    bias = bool(kind == "dense" and rs.rand() < 0.5)
    dens = float(rs.choice([0.01, 0.05, 0.2]))
    f, plan = two.run(False, kind, Nin, N, B, T, rule, bias, dens=dens)
    assert plan == "twolayer-fused"
    g, plan_g = two.run(True, kind, Nin, N, B, T, rule, bias, dens=dens)
    assert plan_g == "generic"
    _same(f, g, f"twolayer {kind} Nin={Nin} N={N} B={B} T={T} rule={rule} bias={bias}")


@pytest.mark.parametrize("seed", range(6 + MORE))
def test_fuzz_twolayer_mstdp(seed):
    rs = np.random.RandomState(3000 + seed)
    Nin = int(rs.choice([64, 256, 784, 1600]))
    N = int(rs.choice([8, 37, 64, 130]))
    B = int(rs.choice([1, 3, 16, 32]))
    T = int(rs.randint(4, 25))
    reward = rs.choice([1.0, -1.0, 0.25, "vec"])
    reward = "vec" if reward == "vec" else float(reward)
    dens = float(rs.choice([0.02, 0.08, 0.3]))
    vmax = int(rs.choice([1, 1, 2]))
    f, plan = two.run_mstdp(False, Nin, N, B, T, reward, dens=dens, vmax=vmax)
    assert plan == "twolayer-fused"
    g, plan_g = two.run_mstdp(True, Nin, N, B, T, reward, dens=dens, vmax=vmax)
    assert plan_g == "generic"
    _same(f, g, f"mstdp Nin={Nin} N={N} B={B} T={T} reward={reward} dens={dens}")


@pytest.mark.parametrize("seed", range(6 + MORE))
def test_fuzz_convlif(seed):
    rs = np.random.RandomState(4000 + seed)
    k = int(rs.choice([1, 3, 5, 4]))
    H, W = int(rs.randint(k + 1, 30)), int(rs.randint(k + 1, 30))
    case = (int(rs.choice([1, 2, 5])), int(rs.randint(3, 25)), int(rs.choice([1, 1, 3])), H, W, int(rs.choice([1, 7, 8, 20])), k,
            int(rs.choice([1, 2])), int(rs.choice([0, 1, 2])), float(rs.choice([0.05, 0.3])), int(rs.choice([1, 1, 2])),
            bool(rs.rand() < 0.5), bool(rs.rand() < 0.5))
    f, plan = conv.run(0, case)
    assert plan == "convlif-fused"
    g, plan_g = conv.run(1, case)
    assert plan_g == "generic"
    _same(f, g, f"conv {case}")


@pytest.mark.parametrize("N,B", [(30, 5), (98, 32), (402, 17), (6, 32)])
def test_lean_resident_partial_last_tile(N, B):
    """Column counts with N % 4 == 2 (and Nin * N % 32 == 0, so the lean form applies): the last workgroup's 4-column tile
    is half empty -- the row-per-thread PostPre writes back single weights there, the exchange carries dead lanes."""
    T = 24
    spikes = [synth.dense_spikes(900 + r + N, (T, B, 784), 0.04) for r in range(2)]
    res, plan = dc.run(0, N, B, T, spikes, w_scale=0.5)
    assert plan.startswith("dc2015-resident")
    gen, plan_g = dc.run(1, N, B, T, spikes, w_scale=0.5)
    assert plan_g == "generic"
    assert sum(int(r["sE"].sum()) for r in gen) > 0
    _same(res, gen, f"resident (auto) N={N} B={B}")
    gres, _ = dc.run(3, N, B, T, spikes, w_scale=0.5)
    _same(gres, gen, f"resident (general) N={N} B={B}")


SOAK = int(os.environ.get("SNN_SOAK_CASES", "144"))


@pytest.mark.parametrize("chunk", range(8))
def test_soak_resident_plan_against_the_cpu_oracle(chunk):
    """A slice of the round-4 soak (tools/soak.py: the resident plan against the generic plan, 1 524 cases) promoted to a test with the CPU
    ORACLE as the checker: SNN_SOAK_CASES (default 144; the GPU tier has to fit the driver's clock) seeded random D&C networks at sizes the scalar oracle finishes in a fraction of a
    second each, three consecutive inputs with learning on (weights, thresholds and the generator position carried over), input densities at
    which the lean form stays in charge; rasters, weights, theta, membrane potentials, traces and the host generator's position bit for bit."""
    n_lean = n_cases = 0
    for seed in range(chunk, SOAK, 8):
        rs = np.random.RandomState(12000 + seed)
        N = int(rs.choice([16, 36, 64, 100, 128]))              # (sizes at which the scalar oracle takes ~0.3 s per case: the slice must fit
        B = int(rs.choice([1, 4, 8, 16, 32] if N <= 64 else [1, 4, 8, 16]))   #  the GPU tier; tools/r04_soak.py covers N up to 1024, T up to 250)
        T = int(rs.choice([20, 30, 40]))
        Nin = int(rs.choice([784, 784, 400, 256]))      # (multiples of 16: what the fused plan takes)
        dens = float(rs.choice([0.006, 0.012, 0.02, 0.03]))
        kw = dict(w_scale=float(rs.choice([0.3, 0.6, 1.0])), n_inputs=3, learning=bool(rs.rand() < 0.9), Nin=Nin,
                  inh=float(rs.choice([120.0, 17.5, 60.0])), exc=float(rs.choice([22.5, 22.5, 30.0])),
                  nu=[(1e-4, 1e-2), (1e-4, 1e-2), (0.0, 1e-2), (1e-3, 0.0), (5e-4, 5e-2)][int(rs.randint(5))])
        shape = {784: (1, 28, 28), 400: (1, 20, 20), 256: (1, 16, 16)}[Nin]
        spikes = [synth.dense_spikes(800 + 13 * seed + r, (T, B, Nin), dens) for r in range(3)]
        res, plan = dc.run(0, N, B, T, spikes, shape=shape, **kw)
        assert plan.startswith("dc2015-resident"), (seed, plan)
        n_lean += plan == "dc2015-resident-lean"
        n_cases += 1
        try:
            dc.same_as_oracle(res, dc.oracle_run(N, B, T, spikes, **kw))
        except AssertionError as e:
            raise AssertionError(f"soak seed {seed}: N={N} B={B} T={T} Nin={Nin} dens={dens} {kw} plan {plan}: {str(e)[:400]}") from None
    assert n_cases == 0 or n_lean >= n_cases // 2, "the lean form should be what runs in most of these cases"
