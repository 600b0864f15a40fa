"""Shared builders: golden-file access, D&C / two-layer state + parameter construction.

Used by the oracle tests (CPU) and the HIP parity tests (GPU) so both sides are driven with
byte-identical inputs.
"""
import hashlib
import os

import numpy as np

import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
f32, u8 = np.float32, np.uint8


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check_packed(g, key, arr):
    """Compare `arr` with a fixture stored either whole or as sha + strided sample. Bit-exact."""
    arr = np.ascontiguousarray(arr)
    if key in g.files:
        np.testing.assert_array_equal(arr.view(np.uint32), g[key].view(np.uint32), err_msg=key)
    else:
        step = max(1, arr.size // 4096)
        np.testing.assert_array_equal(arr.reshape(-1)[::step].view(np.uint32),
                                      g[key + "_sample"].view(np.uint32), err_msg=key + " sample")
        assert sha(arr) == str(g[key + "_sha"]), key + " sha256"


def unpack(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(shape).astype(u8)


def fixture_input(g, r, T, B, n=784):
    """Input `r` of a D&C full-size fixture: the stored reference-encoded Poisson train (bit-packed,
    sha-checked) when the fixture has one, else round 1/2's synth.spike_train(1000 + r)."""
    if f"r{r}_in" in g.files:
        sp = unpack(g[f"r{r}_in"], (T, B, n))
        assert sha(sp) == str(g[f"r{r}_in_sha"])
        return sp
    return synth.spike_train(1000 + r, T, B, n)


DC_FULL = ["full_cfg1_dc_n100_b1", "full_cfg2_dc_n400_b32", "full_cfg1_dc_n100_b1_poisson", "full_cfg2_dc_n400_b32_poisson",
           "full_cfg2_dc_n400_b32_bold", "full_cfg2_dc_n400_b32_strokes"]      # (_strokes, round 5: digit-like images -- synth.stroke_digit -- at eth_mnist.py's intensity)


def exp_noise(seed, n):
    """The Exp(1) stream torch.multinomial consumes after torch.manual_seed(seed)."""
    import torch
    torch.manual_seed(int(seed))
    return torch.empty(int(n)).exponential_(1).numpy()


# DiehlAndCook2015 constants that are plain python floats in the reference constructor
# (bindsnet/models/models.py:156-236 as called by examples/mnist/eth_mnist.py:91-100)
DC_CONST = dict(e_rest=-65.0, e_reset=-60.0, e_thresh=-52.0, e_refrac=5.0, e_theta_plus=0.05,
                i_rest=-60.0, i_reset=-45.0, i_thresh=-40.0, i_refrac=2.0,
                nu0=1e-4, nu1=1e-2, wmin=0.0, wmax=1.0, norm=78.4, exc=22.5)


def dc_weights(N, inh, exc=22.5):
    W_ei = (exc * np.eye(N)).astype(f32)
    W_ie = (-inh * (np.ones((N, N)) - np.eye(N))).astype(f32)
    return W_ei, W_ie


def dc_state(N, B, Nin=784, inh=120.0):
    W_ei, W_ie = dc_weights(N, inh)
    return dict(
        W_xe=synth.weights_q12(10, Nin, N), W_ei=W_ei, W_ie=W_ie,
        sX=np.zeros((B, Nin), u8), xX=np.zeros((B, Nin), f32),
        vE=np.full((B, N), -65.0, f32), rE=np.zeros((B, N), f32), sE=np.zeros((B, N), u8),
        xE=np.zeros((B, N), f32), theta=np.zeros(N, f32),
        vI=np.full((B, N), -60.0, f32), rI=np.zeros((B, N), f32), sI=np.zeros((B, N), u8))


def dc_reset(st):
    """Network.reset_state_variables() for the D&C graph (theta and weights persist)."""
    st["sX"][:] = 0; st["xX"][:] = 0
    st["vE"][:] = -65.0; st["rE"][:] = 0; st["sE"][:] = 0; st["xE"][:] = 0
    st["vI"][:] = -60.0; st["rI"][:] = 0; st["sI"][:] = 0
