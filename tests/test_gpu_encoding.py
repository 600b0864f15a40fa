"""Device encoders.
 * bernoulli on the MI355X draws from the HOST generator's stream: spike trains bit-identical to the host path (which
   tests/test_host_plumbing.py pins to the reference), and the host generator ends up where torch.bernoulli leaves it.
 * poisson on the MI355X uses its own seeded Philox stream -- NOT stream-compatible with the reference (documented) --
   so it is checked distributionally against the host path: firing rates per intensity, inter-spike-interval statistics,
   determinism per seed."""
import numpy as np
import pytest
import torch

import synth
from make_golden_host_cases import ENC_CASES, datum_for

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("k", [k for k, c in enumerate(ENC_CASES) if c[0] == "bernoulli"])
def test_bernoulli_device_is_the_host_stream(k):
    from bindsnet_amd.encoding import bernoulli
    name, shape, scale, time, dt, kw = ENC_CASES[k]
    torch.manual_seed(100 + k)
    host = bernoulli(T_(datum_for(k, shape, scale)).clone(), time=time, dt=dt, **kw)
    probe_host = torch.rand(3)
    torch.manual_seed(100 + k)
    dev = bernoulli(T_(datum_for(k, shape, scale)).clone(), time=time, dt=dt, device=DEV, **kw)
    probe_dev = torch.rand(3)
    assert dev.is_cuda and dev.dtype == torch.uint8 and tuple(dev.shape) == tuple(host.shape)
    assert torch.equal(dev.cpu(), host), f"case {k}"
    assert torch.equal(probe_host, probe_dev), "host generator position after the device encoder"


def test_bernoulli_device_long_stream_and_generator_offsets():
    """196 000 draws (one eth_mnist-sized sample), starting mid-block and exactly at a block boundary."""
    from bindsnet_amd.encoding import bernoulli
    x = T_(synth.uniform_f32(5, (1, 28, 28), 0.0, 1.0))
    for warm in (0, 7, 624, 1000):
        torch.manual_seed(9)
        torch.rand(warm) if warm else None
        host = bernoulli(x.clone(), time=250)
        ph = torch.rand(2)
        torch.manual_seed(9)
        torch.rand(warm) if warm else None
        dev = bernoulli(x.clone(), time=250, device=DEV)
        assert torch.equal(dev.cpu(), host) and torch.equal(ph, torch.rand(2)), f"warm-up {warm}"


def test_poisson_device_matches_host_distribution():
    from bindsnet_amd.encoding import poisson, poisson_device
    T, reps = 250, 48
    levels = np.array([0.0, 2.0, 8.0, 32.0, 64.0, 128.0, 255.0], np.float32)
    x = T_(np.repeat(levels, 64))                                 # 64 elements per intensity
    torch.manual_seed(1)
    host = torch.stack([poisson(x.clone(), time=T).float() for _ in range(reps)])         # [reps, T, n]
    dev = torch.stack([poisson_device(x.clone(), time=T, device=DEV, seed=1000 + r).float().cpu() for r in range(reps)])
    assert dev.shape == host.shape
    for li, lv in enumerate(levels):
        h, d = host[:, :, li * 64:(li + 1) * 64], dev[:, :, li * 64:(li + 1) * 64]
        if lv == 0:
            assert d.sum() == 0 and h.sum() == 0
            continue
        rh, rd = h.mean().item(), d.mean().item()
        se = np.sqrt(rh / (reps * T * 64)) * 4 + 0.01 * rh        # 4 standard errors + 1 % (intervals are discretised alike)
        assert abs(rh - rd) <= se, f"intensity {lv}: host rate {rh:.5f} device rate {rd:.5f}"
        # inter-spike intervals: mean and variance of the first interval lengths
        def isi(a):
            out = []
            for r in range(min(reps, 8)):
                for e in range(16):
                    t = torch.nonzero(a[r, :, e]).view(-1)
                    if len(t) > 2:
                        out.append((t[1:] - t[:-1]).float())
            return torch.cat(out) if out else torch.zeros(1)
        ih, idv = isi(h), isi(d)
        if len(ih) > 200 and len(idv) > 200:
            assert abs(ih.mean() - idv.mean()) <= 0.06 * ih.mean() + 0.1, f"intensity {lv}: ISI mean"
            assert abs(ih.std() - idv.std()) <= 0.12 * ih.std() + 0.15, f"intensity {lv}: ISI spread"
    # determinism per seed, and the default seed comes from the global CPU generator
    a = poisson_device(x.clone(), time=T, device=DEV, seed=5)
    assert torch.equal(a, poisson_device(x.clone(), time=T, device=DEV, seed=5))
    assert not torch.equal(a, poisson_device(x.clone(), time=T, device=DEV, seed=6))
    torch.manual_seed(3)
    b = poisson(x.clone(), time=T, device=DEV)
    torch.manual_seed(3)
    assert torch.equal(b, poisson(x.clone(), time=T, device=DEV))


def test_poisson_device_equals_the_oracle_restatement_bit_for_bit():
    """snn_encode_poisson's stream is specified operation by operation (csrc/snn_encode.hip); oracle/snn_oracle.c restates it in plain C
    (orc_encode_poisson).  Same seeds, same data -> the same spike trains, bit for bit: all three sampler branches (lambda < 30 by the
    multiplication method, lambda >= 30 by PTRS incl. its log / log k! rejection test, x = 0), several dt, several seeds, a size that is not a
    multiple of the launch's block."""
    import oracle
    from bindsnet_amd.encoding import poisson_device
    rng = np.random.default_rng(11)
    levels = np.concatenate([np.array([0.0, 0.3, 1.0, 2.0, 5.0, 8.0, 20.0, 33.0, 33.4, 40.0, 64.0, 128.0, 255.0, 1000.0, 5000.0], np.float32),
                             (255.0 * rng.random(986)).astype(np.float32)])          # 1001 elements
    for dt, T, seed in ((1.0, 250, 7), (0.5, 100, 2 ** 40 + 3), (2.0, 300, 0)):
        steps = int(T / dt)
        dev = poisson_device(T_(levels.copy()), time=T, dt=dt, device=DEV, seed=seed).cpu().numpy()
        ref = oracle.encode_poisson(levels, steps, dt, seed)
        assert dev.shape == ref.shape == (steps, levels.size)
        assert int(ref.sum()) > 1000
        np.testing.assert_array_equal(dev, ref, err_msg=f"dt {dt}, seed {seed}")


def test_poisson_encoder_object_and_environment_route_to_the_device(monkeypatch):
    """PoissonEncoder(time, dt, device="cuda") -- and, for a script that builds it without a device (examples/mnist/eth_mnist.py), the
    environment switch SNN_ENCODE_DEVICE=cuda -- encode on the MI355X: a tensor of the reference's shape (on the device asked for: the
    switch hands a host tensor back, which the script's DataLoader may pin), repeatable under torch.manual_seed (the stream's seed is one
    draw from the host generator)."""
    from bindsnet_amd.encoding import PoissonEncoder
    img = T_(synth.uniform_f32(9, (1, 28, 28), 0.0, 128.0))
    torch.manual_seed(4)
    a = PoissonEncoder(time=250, dt=1.0, device=DEV)(img.clone())
    assert a.is_cuda and a.shape == (250, 1, 28, 28) and a.dtype == torch.uint8 and int(a.sum()) > 0
    monkeypatch.setenv("SNN_ENCODE_DEVICE", DEV)
    torch.manual_seed(4)
    b = PoissonEncoder(time=250, dt=1.0)(img.clone())
    assert not b.is_cuda and torch.equal(a.cpu(), b)          # (encoded on the device, handed back where the caller asked for it)
    monkeypatch.delenv("SNN_ENCODE_DEVICE")
    assert not PoissonEncoder(time=250, dt=1.0)(img.clone()).is_cuda
