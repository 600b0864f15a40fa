"""ctypes binding of libsnnhip.so (include/snnhip.h).

The library is the ONLY compute path of this package: if it is missing or a call fails, an
exception is raised -- there is no PyTorch/CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsnnhip.so")
if os.environ.get("SNN_DEVELOPER") == "1" and os.environ.get("SNN_LIB_OVERRIDE"):
    # developer A/B of two builds (tools/): honoured only with SNN_DEVELOPER=1, and never silently
    import warnings
    LIB_PATH = os.environ["SNN_LIB_OVERRIDE"]
    warnings.warn(f"bindsnet_amd: SNN_LIB_OVERRIDE is active, loading {LIB_PATH} instead of the in-tree library", RuntimeWarning)

SNN_OK, SNN_ERR_NOISE, SNN_ERR_TIMEOUT, SNN_ERR_RETRY = 0, -4, -6, -7
ABI_VERSION = 8


# ---- descriptor-cache invalidation (network/network.py): every attribute assignment on a network object (layer,
# connection, feature, learning rule, monitor, the network itself) and every Module._apply (.to(), .cuda(), .float())
# advances this counter; Network.run() re-uses the descriptor arrays of its previous call only while it stands still.
_EPOCH = [0]


def epoch() -> int:
    return _EPOCH[0]


def touch() -> None:
    _EPOCH[0] += 1


class Touching:
    """Mixin: assignments to attributes of the object invalidate cached run descriptors."""

    def __setattr__(self, key, value):
        _EPOCH[0] += 1
        super().__setattr__(key, value)

    def __delattr__(self, key):
        _EPOCH[0] += 1
        super().__delattr__(key)


class TouchingModule(Touching):
    """Touching for torch.nn.Module subclasses: moving / casting the module replaces its tensors without __setattr__."""

    def _apply(self, fn, *args, **kwargs):
        _EPOCH[0] += 1
        return super()._apply(fn, *args, **kwargs)


class SnnError(RuntimeError):
    pass


class LifParams(C.Structure):
    _fields_ = [("decay", C.c_float), ("rest", C.c_float), ("reset", C.c_float), ("thresh", C.c_float),
                ("refrac", C.c_float), ("dt", C.c_float), ("has_lbound", C.c_int), ("lbound", C.c_float),
                ("traces", C.c_int), ("trace_decay", C.c_float), ("trace_scale", C.c_float),
                ("traces_additive", C.c_int)]


class DcParams(C.Structure):
    _fields_ = [("lif", LifParams), ("theta_decay", C.c_float), ("theta_plus", C.c_float),
                ("learning", C.c_int), ("one_spike", C.c_int)]


class LayerDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("n", C.c_int), ("p", DcParams),
                ("v", C.c_void_p), ("refrac", C.c_void_p), ("x", C.c_void_p), ("theta", C.c_void_p),
                ("s", C.c_void_p), ("ext_spikes", C.c_void_p), ("raster_s", C.c_void_p),
                ("raster_v", C.c_void_p), ("current", C.c_void_p),
                ("clamp", C.c_void_p), ("unclamp", C.c_void_p), ("clamp_per_step", C.c_int), ("unclamp_per_step", C.c_int),
                ("inject_v", C.c_void_p), ("inject_per_step", C.c_int), ("inject_len", C.c_int),
                ("ext_current", C.c_void_p), ("thresh_vec", C.c_void_p)]


class ConnDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("src", C.c_int), ("dst", C.c_int), ("w", C.c_void_p), ("bias", C.c_void_p),
                ("cin", C.c_int), ("h", C.c_int), ("wd", C.c_int), ("cout", C.c_int), ("kh", C.c_int),
                ("kw", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
                ("rule", C.c_int), ("nu0", C.c_float), ("nu1", C.c_float), ("use_dt", C.c_int),
                ("wdecay", C.c_float), ("has_min", C.c_int), ("wmin", C.c_float), ("has_max", C.c_int),
                ("wmax", C.c_float),
                ("p_plus", C.c_void_p), ("p_minus", C.c_void_p), ("s_src_prev", C.c_void_p),
                ("s_tgt_prev", C.c_void_p), ("reward", C.c_float), ("reward_vec", C.c_void_p),
                ("a_plus", C.c_float), ("a_minus", C.c_float), ("decay_plus", C.c_float),
                ("decay_minus", C.c_float),
                ("has_norm", C.c_int), ("norm", C.c_float), ("norm_abs", C.c_int), ("norm_ws", C.c_void_p),
                ("e_trace", C.c_void_p), ("decay_e", C.c_float), ("tc_e", C.c_float), ("rule_ws", C.c_void_p),
                ("mask", C.c_void_p), ("raster_w", C.c_void_p)]


class RunDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("dt", C.c_float), ("learning", C.c_int),
                ("noise_q", C.c_void_p), ("q_len", C.c_longlong), ("rng", C.c_void_p), ("qbuf", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_ulonglong),
                ("cursor", C.c_void_p), ("status", C.c_void_p), ("one_step", C.c_int), ("plan", C.c_int),
                ("status2", C.c_void_p), ("host_state", C.c_void_p)]


class FillSegment(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("bytes", C.c_ulonglong), ("pattern", C.c_uint32)]


MAX_FILL_SEGMENTS = 32
LAYER_INPUT, LAYER_LIF, LAYER_DC = 0, 1, 2
CONN_MCC, CONN_DENSE, CONN_CONV2D = 0, 1, 2
RULE_NONE, RULE_POSTPRE, RULE_MSTDP, RULE_HEBBIAN, RULE_WDPOSTPRE, RULE_MSTDPET = 0, 1, 2, 3, 4, 5

_lib = None

_vp, _i, _f, _l, _ll = C.c_void_p, C.c_int, C.c_float, C.c_long, C.c_longlong
_SIGS = {
    "snn_abi_version": ([], _i),
    "snn_error_string": ([_i], C.c_char_p),
    "snn_last_hip_error": ([], C.c_char_p),
    "snn_device_count": ([], _i),
    "snn_prop_cascade_f32": ([_vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "snn_prop_dense_f32": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "snn_prop_dense_mfma_f32": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "snn_prop_conv2d_f32": ([_vp, _vp, _vp, _vp] + [_i] * 10 + [_vp], _i),
    "snn_input_step": ([_vp, _vp, _l, _f, _f, _i, _vp, _vp], _i),
    "snn_lif_step": ([_vp, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(LifParams), _vp, _vp, _vp], _i),
    "snn_lif_step_vth": ([_vp, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(LifParams), _vp, _vp, _vp, _vp], _i),
    "snn_dc_step": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(DcParams), _vp, _ll, _vp, _vp, _vp, _vp, _vp], _i),
    "snn_dc_arbitrate": ([_vp, _vp, _i, _i, C.POINTER(DcParams), _vp, _ll, _vp, _vp, _vp, _vp], _i),
    "snn_stdp_postpre": ([_vp] * 5 + [_i, _i, _i, _f, _f, _i, _f, _f, _i, _f, _i, _f, _i, _vp], _i),
    "snn_conv2d_postpre": ([_vp] * 5 + [_i] * 9 + [_f, _f, _f, _i, _f, _i, _f, _vp, _vp], _i),
    "snn_stdp_hebbian": ([_vp] * 5 + [_i, _i, _i, _f, _f, _i, _f, _i, _f, _i, _f, _vp], _i),
    "snn_conv2d_mstdp_step": ([_vp] * 6 + [_i] * 8 + [_f] * 7 + [_i, _f, _i, _f, _vp], _i),
    "snn_mstdpet_step": ([_vp] * 8 + [_i, _i] + [_f] * 10 + [_i, _f, _i, _f, _vp], _i),
    "snn_mstdp_step": ([_vp] * 7 + [_i, _i, _i, _f, _vp, _f, _f, _f, _f, _f, _f, _i, _f, _i, _f, _vp], _i),
    "snn_normalize": ([_vp, _i, _i, _f, _i, _vp, _vp], _i),
    "snn_normalize_conv2d": ([_vp, _i, _i, _f, _vp], _i),
    "snn_rng_fill_exponential": ([_vp, _vp, _i, _i, _vp, _vp, _vp], _i),
    "snn_encode_bernoulli": ([_vp, _vp, _i, _i, _f, _vp, _vp], _i),
    "snn_encode_poisson": ([_vp, _i, _i, _f, C.c_ulonglong, _vp, _vp], _i),
    "snn_fill_segments": ([C.POINTER(FillSegment), _i, _vp], _i),
    "snn_dist_unique_id": ([_vp], _i),
    "snn_dist_init": ([_i, _i, _vp, C.POINTER(_vp)], _i),
    "snn_dist_world": ([_vp, C.POINTER(_i), C.POINTER(_i)], _i),
    "snn_dist_allreduce_dw": ([_vp, _vp, _ll, _vp], _i),
    "snn_dist_allgather_step": ([_vp, _vp, _vp, _ll, _vp], _i),
    "snn_dist_destroy": ([_vp], _i),
    "snn_net_run": ([C.POINTER(LayerDesc), _i, C.POINTER(ConnDesc), _i, C.POINTER(RunDesc), _vp], _i),
    "snn_net_workspace_bytes": ([C.POINTER(LayerDesc), _i, C.POINTER(ConnDesc), _i, C.POINTER(RunDesc)], C.c_ulonglong),
    "snn_plan_name": ([], C.c_char_p),
    "snn_dc2015_last_form": ([], C.c_int),
    "snn_set_plan_mode": ([_i], None),
    "snn_graph_stats": ([C.POINTER(C.c_int)] * 3, None),
    "snn_profile_enable": ([_i], None),
    "snn_profile_collect": ([C.POINTER(C.c_double), C.POINTER(C.c_int)], _i),
}


def exported_symbols():
    """Every symbol include/snnhip.h declares (kept in sync by tests/test_abi.py)."""
    return sorted(_SIGS)


def lib():
    """Load libsnnhip.so (built by __graft_entry__.build() / csrc/Makefile). Fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SnnError(f"{LIB_PATH} is missing: build it with `make -C bindsnet_amd/csrc` "
                           "(python __graft_entry__.py build). There is no fallback path.")
        L = C.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(L, name)          # AttributeError if the .so is stale -> loud
            fn.argtypes, fn.restype = args, res
        if L.snn_abi_version() != ABI_VERSION:
            raise SnnError(f"libsnnhip ABI {L.snn_abi_version()} != binding {ABI_VERSION}")
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != SNN_OK:
        L = lib()
        msg = L.snn_error_string(rc).decode()
        if rc == -3:
            msg += ": " + L.snn_last_hip_error().decode()
        raise SnnError(f"libsnnhip {what}: {msg} (code {rc})")


def resident_kernel_name(form: int) -> str:
    return {3: "k_dc2015_async [lean form, third generation: compute workgroups + arbiter + raster writers + producer workgroups (the digest / X-trace "
               "pre-passes and the input monitor's copy run inside the launch)]", 2: "k_dc2015_spec [lean form, second generation]",
            1: "k_dc2015_run [lean form]"}.get(form, "k_dc2015_run")


def profile_run(net, inputs, time, stride=4, repeats=5, pipelined=False):
    """Extra run()s with HIP events around the plan's dominant launches (bench.py roofline): every `stride`-th
    timestep's launch for the per-step plans, the one launch of each run for the resident plan (`repeats` runs).
    `pipelined`: the runs are those of a Network.pipelined() section (the launch mode the caller's timed region used).
    One untimed run first: the first launch of a mode pays one-time set-up (the runtime creates the queue cooperative
    launches go through at the first of them: several ms).
    Returns {"kernel", "avg_ms", "n", "timesteps_per_launch"} or None."""
    import contextlib
    import torch
    L = lib()
    with (net.pipelined() if pipelined else contextlib.nullcontext()):
        net.run(dict(inputs), time=time)
        net.reset_state_variables()
        net.sync()
        torch.cuda.synchronize()
        L.snn_profile_enable(stride)
        try:
            for _ in range(max(1, repeats)):
                net.run(dict(inputs), time=time)
                net.reset_state_variables()
                if not net.last_plan.startswith("dc2015-resident"):
                    break
            net.sync()
            torch.cuda.synchronize()
            s, n = C.c_double(0), C.c_int(0)
            check(L.snn_profile_collect(C.byref(s), C.byref(n)), "profile_collect")
        finally:
            L.snn_profile_enable(0)
    if n.value == 0:
        return None
    plan = net.last_plan
    if plan.startswith("dc2015-resident"):
        form = L.snn_dc2015_last_form()
        return {"kernel": resident_kernel_name(form) + " (one launch per network.run())", "resident_form": form, "avg_ms": s.value / n.value, "n": n.value,
                "timesteps_per_launch": int(round(time / net.dt))}
    kernel = "k_dc2015_step (one launch per timestep)" if plan != "generic" else "generic plan: all launches of one timestep"
    return {"kernel": kernel, "avg_ms": s.value / n.value, "n": n.value, "timesteps_per_launch": 1}


def graph_stats():
    """(plain, captured, replayed) run counts of the fused plan."""
    a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
    lib().snn_graph_stats(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value
