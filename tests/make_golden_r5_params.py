"""Sizes, thresholds and weights of tests/golden/make_golden_r5.py's network (run_lif_vector_thresh), for the tests that rebuild it from this package."""
import numpy as np

import synth

n_in, n_out, B, T = 96, 70, 5, 60


def thresholds():
    return (-60.0 + 12.0 * synth.uniform_f32(9, (n_out,), 0.0, 1.0)).astype(np.float32)


def weights():
    return synth.uniform_f32(5, (n_in, n_out), 0.0, 2.5), synth.uniform_f32(6, (n_out, n_out), -0.5, 0.5)
