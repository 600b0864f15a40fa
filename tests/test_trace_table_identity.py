"""The premise of the SNN_LDS_XTRACE developer build (csrc/snn_dc2015_async.hip): a non-additive trace (nodes.py:96-103: x <- trace_scale on a
spike, x * trace_decay otherwise, one f32 multiplication per step) that enters a run as zero equals, bit for bit, table[steps since the input's last
spike] with table[k] = trace_scale multiplied k times by the decay -- so the won branch of the D&C kernel can take its X-trace values from a
last-spike step per (sample, input) and a 256-entry table in LDS instead of the [T+1][B][Nin] array in global memory.  Checked here against the
oracle's trace recurrence in plain numpy f32 (no GPU)."""
import numpy as np
import pytest


@pytest.mark.parametrize("tc_trace,scale,dens", [(20.0, 1.0, 0.012), (5.0, 0.5, 0.2), (100.0, 1.0, 0.001)])
def test_trace_equals_table_of_steps_since_last_spike(tc_trace, scale, dens):
    rs = np.random.RandomState(7)
    T, B, Nin = 250, 3, 131
    decay = np.float32(np.exp(np.float32(-1.0) / np.float32(tc_trace)))       # nodes.py:122-131 (dt = 1)
    scale = np.float32(scale)
    s = rs.rand(T, B, Nin) < dens
    x = np.zeros((B, Nin), np.float32)
    table = np.zeros(256, np.float32)
    v = scale
    for k in range(256):
        table[k] = v
        v = np.float32(v * decay)
    last = np.full((B, Nin), 255, np.int64)
    for t in range(T):
        x = np.where(s[t], scale, (x * decay).astype(np.float32)).astype(np.float32)      # the recurrence, as the pre-pass kernel runs it
        last = np.where(s[t], t, last)
        tab = np.where(last == 255, np.float32(0), table[np.clip(t - last, 0, 255)])
        np.testing.assert_array_equal(tab.view(np.uint32), x.view(np.uint32), err_msg=f"step {t}")
