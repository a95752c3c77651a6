"""ctypes binding of libsavfi_hip.so -- the only door between the Python host code and the kernels.

The prototypes below are the ones declared in include/savfi_hip.h.  There is NO fallback: if the
shared object is missing or a call returns non-zero, a SavfiHipError is raised.  On a machine
without a GPU the library still loads (hipcc cross-compiled it) so that the symbol table can be
checked, but nothing may be launched.
"""
import ctypes
import os
import re
from ctypes import c_double, c_float, c_int, c_int64, c_void_p, POINTER

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAVFI_HIP_LIB") or os.path.join(_PKG_DIR, "lib", "libsavfi_hip.so")     # override: kernel experiments
HEADER_PATH = os.path.join(os.path.dirname(_PKG_DIR), "include", "savfi_hip.h")

RULE_SGD, RULE_ADAM, RULE_ADAMAX_LSLR, RULE_ADAMAX_MSGD = 0, 1, 2, 3
LR_SCALAR, LR_ELEMENT = 0, 1
ABI_VERSION = 20

_ERRORS = {-1: "SAVFI_E_NULL (a required pointer is NULL)",
           -2: "SAVFI_E_SHAPE (bad or inconsistent dimension)",
           -3: "SAVFI_E_UNSUPPORTED",
           -4: "SAVFI_E_TOOBIG (index arithmetic would overflow)"}


class SavfiHipError(RuntimeError):
    pass


_P = c_void_p          # device pointer
_PP = POINTER(c_void_p)  # host array of device pointers
_I64P = POINTER(c_int64)
_FP = POINTER(c_float)

_PROTOTYPES = {
    "savfi_version": [],
    "savfi_sepconv_fwd_f32": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "savfi_sepconv_bwd_f32": [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "savfi_sepconv_taps_strided_supported": [c_int] * 6,
    "savfi_sepconv_fwd_taps_strided_f32": [_P, _P, _P, _P] + [c_int] * 6 + [_P],
    "savfi_sepconv_bwd_taps_strided_f32": [_P, _P, _P, _P, _P, _P] + [c_int] * 6 + [_P],
    "savfi_frames8_classify_f32": [_P, c_int64, _P, _P],
    "savfi_sepconv_fwd_frames8_f32": [_P, _P, _P, _P, _P] + [c_int] * 7 + [_P],
    "savfi_sepconv_bwd_frames8_f32": [_P, _P, _P, _P, _P, _P, _P] + [c_int] * 7 + [_P],
    "savfi_sepconv_bwd_pair_frames8_f32": [_P] * 7 + [c_int] * 6 + [_P],
    "savfi_sepconv_fwd_pair_frames8_f32": [_P] * 6 + [c_int] * 6 + [_P],
    "savfi_sepconv_ws_errors": [],
    "savfi_sepconv_ws_watch": [],
    "savfi_sepconv_ws_errors_peek": [],
    "savfi_sepconv_ws_errors_reset": [],
    "savfi_sepconv_ws_debug_spin_limit": [c_int, POINTER(c_int)],
    "savfi_conv3x3_dgrad_masked_f32": [_P, _P, _P, c_float, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "savfi_convk_dgrad_masked_f32": [_P, _P, _P, c_float, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "savfi_voxelwarp_fwd_f32": [_P, _P, _P, c_int, c_int, c_int, _P],
    "savfi_voxelwarp_bwd_f32": [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    "savfi_avgpool2x2_fwd_f32": [_P, _P, c_int64, c_int, c_int, _P],
    "savfi_avgpool2x2_bwd_f32": [_P, _P, c_int64, c_int, c_int, _P],
    "savfi_avgpool2x2_bwd_fused_f32": [_P, _P, _P, c_float, _P, c_int64, c_int, c_int, _P],
    "savfi_flowwarp_fwd_f32": [_P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "savfi_flowwarp_bwd_f32": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "savfi_pixel_unshuffle_f32": [_P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "savfi_pixel_shuffle_f32": [_P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "savfi_mt_update_f32": [c_int, c_int, c_int, _PP, _PP, _PP, _PP, _PP, _PP, _PP, _I64P, _FP, _FP,
                            c_double, c_double, c_double, _P],
    "savfi_mt_update_bwd_f32": [c_int, c_int, _PP, _PP, _PP, _I64P, c_float, _P],
    "savfi_mt_mean_f32": [c_int, _PP, _I64P, _P, _P],
    "savfi_mt_scale_f32": [c_int, _PP, _P, _PP, _I64P, _P],
    "savfi_mt_scale_bwd_f32": [c_int, _PP, _PP, _P, _PP, _P, _I64P, _P],
    "savfi_l1_mse_scratch_floats": [c_int, c_int64],
    "savfi_l1_mse_f32": [c_int, _P, _P, _P, _P, c_int, c_int64, _P],
    "savfi_l1_mse_bwd_f32": [c_int, _P, _P, _P, _P, c_int, c_int64, _P],
    "savfi_upsample2x_fwd_f32": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "savfi_upsample2x_bwd_f32": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "savfi_conv3x3_workspace_floats": [c_int] * 7,
    "savfi_conv3x3_f32": [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P],
    "savfi_conv3x3_tasks_workspace_floats": [c_int] * 8,
    "savfi_conv3x3_tasks_f32": [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P],
    "savfi_conv3x3_filter_floats": [c_int] * 4,
    "savfi_conv3x3_f4_workgroups": [c_int] * 7,
    "savfi_conv3x3_filters_form_f32": [_P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "savfi_conv3x3_filters_multi_form_f32": [_P, _P, _P, _P, _P, _P, _P, c_int, _P],
    "savfi_conv3x3_dgrad_masked_form_f32": [_P, _P, _P, c_float, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "savfi_conv3x3_filters_f32": [_P, _P, _P, c_int, c_int, c_int, _P],
    "savfi_conv3x3_tasks_pre_workspace_floats": [c_int] * 8,
    "savfi_conv3x3_tasks_pre_f32": [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P],
    "savfi_conv3x3_unit16_supported": [c_int] * 7,
    "savfi_conv3x3_in_unit16_supported": [c_int] * 7,
    "savfi_conv3x3_dgrad_in_unit16_f32": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "savfi_conv3x3_tasks_pre_unit16_f32": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P],
    "savfi_conv3x3_wgrad_tasks_workspace_floats": [c_int] * 7,
    "savfi_conv3x3_wgrad_tasks_f32": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "savfi_conv3x3_wgrad_wino_tasks_workspace_floats": [c_int] * 7,
    "savfi_conv3x3_wgrad_wino_tasks_f32": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "savfi_conv3x3_wgrad_wino_tasks_bias_workspace_floats": [c_int] * 7,
    "savfi_conv3x3_wgrad_wino_tasks_bias_f32": [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "savfi_conv3x3_wgrad_workspace_floats": [c_int] * 6,
    "savfi_conv3x3_wgrad_f32": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "savfi_convk_filter_floats": [c_int] * 5,
    "savfi_convk_filters_f32": [_P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "savfi_convk_filters_multi_f32": [_P, _P, _P, _P, _P, _P, _P, c_int, _P],
    "savfi_conv3x3_filters_multi_f32": [_P, _P, _P, _P, _P, _P, c_int, _P],
    "savfi_convk_tasks_pre_f32": [_P, _P, _P, _P] + [c_int] * 9 + [c_float, c_int, _P],
    "savfi_convk_tasks_pre_reflect_f32": [_P, _P, _P, _P] + [c_int] * 9 + [c_float, c_int, c_int, _P],
    "savfi_convk_wgrad_tasks_reflect_f32": [_P, _P, _P, _P] + [c_int] * 10 + [_P],
    "savfi_reflect_pad_bwd_f32": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "savfi_reflect_pad_fwd_f32": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "savfi_reflect_pad_bwd_add_f32": [_P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "savfi_convk_wgrad_workspace_floats": [c_int] * 8,
    "savfi_convk_wgrad_sums_bias": [c_int] * 8,
    "savfi_convk_wgrad_tasks_bias_f32": [_P, _P, _P, _P, _P] + [c_int] * 9 + [_P],
    "savfi_convk_wgrad_tasks_f32": [_P, _P, _P, _P] + [c_int] * 9 + [_P],
    "savfi_ca_pool_f32": [_P, _P, _P, c_int64, c_int, c_float, _P],
    "savfi_ca_mlp_fwd_f32": [_P] * 7 + [c_int] * 4 + [_P],
    "savfi_ca_mlp_bwd_f32": [_P] * 11 + [c_int] * 4 + [c_float, _P],
    "savfi_ca_apply_f32": [_P] * 5 + [c_int64, c_int, _P],
    "savfi_ca_apply_mlp_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "savfi_ca_apply_bwd_mlp_f32": [_P] * 12 + [c_int, c_int, c_int, c_int, c_int, _P],
    "savfi_sub_mean_workspace_floats": [c_int64, c_int],
    "savfi_sub_mean_f32": [_P, _P, _P, _P, c_int64, c_int, _P],
    "savfi_frames_u8_to_f32": [_P, _P, c_int64, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, _P],
    "savfi_upsample2x_window_fwd_f32": [_P, _P] + [c_int] * 12 + [_P],
    "savfi_upsample2x_window_bwd_f32": [_P, _P] + [c_int] * 12 + [_P],
    "savfi_upsample2x_window_bwd_masked_f32": [_P, _P, c_float, _P] + [c_int] * 12 + [_P],
    "savfi_bias_act_fwd_f32": [_P, _P, c_int, c_int, c_int, c_float, _P],
    "savfi_bias_act_scratch_floats": [c_int, c_int, c_int],
    "savfi_bias_act_bwd_f32": [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P],
}

_lib = None


def declared_symbols():
    """Function names declared in include/savfi_hip.h (used by the symbol-export test)."""
    with open(HEADER_PATH) as fh:
        text = fh.read()
    return sorted(set(re.findall(r"^\s*(?:int|int64_t)\s+(savfi_\w+)\s*\(", text, flags=re.M)))


def lib():
    """Load (once) and return the ctypes handle.  Raises SavfiHipError when the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SavfiHipError(
            "%s is missing: build it with `python __graft_entry__.py build` "
            "(there is no CPU or PyTorch fallback for the savfi kernels)" % LIB_PATH)
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the host's ROCm install
        raise SavfiHipError("cannot load %s: %s" % (LIB_PATH, exc))
    for name, argtypes in _PROTOTYPES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise SavfiHipError("%s does not export %s" % (LIB_PATH, name))
        fn.argtypes = argtypes
        fn.restype = c_int64 if name.endswith(('_floats', '_workgroups')) else c_int
    got = handle.savfi_version()
    if got != ABI_VERSION:
        raise SavfiHipError("libsavfi_hip ABI %d != expected %d; rebuild" % (got, ABI_VERSION))
    _lib = handle
    return _lib


def check(code, what):
    if code == 0:
        return
    if code < 0:
        raise SavfiHipError("%s: %s" % (what, _ERRORS.get(code, "error %d" % code)))
    raise SavfiHipError("%s: HIP launch failed with hipError_t %d" % (what, code))


_ws_watched = set()


def ws_watch():
    """Arm the no-sync error word of the wave-specialised SepConv kernels on the current device (once; not inside a stream capture)."""
    import torch
    dev = torch.cuda.current_device()
    if dev not in _ws_watched:
        check(lib().savfi_sepconv_ws_watch(), "savfi_sepconv_ws_watch")
        _ws_watched.add(dev)


def ws_check(where=""):
    """Raise when a bounded in-kernel wait of csrc/sepconv_ws.hip has given up since the library was loaded: the launch it happened in
    returned wrong numbers (the kernels never hang).  Reads one mapped host word -- no HIP call, no synchronisation: call it where the
    device has been synchronised anyway.  A no-op until ws_watch() armed the word on this device."""
    if _lib is None or not _ws_watched:
        return
    n = _lib.savfi_sepconv_ws_errors_peek()
    if n > 0:
        _lib.savfi_sepconv_ws_errors_reset()      # reported once: a caller that handles the exception continues from a clean word
        raise SavfiHipError("%d bounded wait(s) of the wave-specialised SepConv kernels gave up (csrc/sepconv_ws.hip)%s: the results of "
                            "the launches since the last check are wrong.  A kernel bug, or the GPU was shared with another process for "
                            "longer than the spin limit." % (n, (" -- " + where) if where else ""))


def ws_armed():
    """Has ws_watch() armed the error word on some device of this process (i.e. can ws_check() see anything)?"""
    return bool(_ws_watched)


def ptr_array(tensors):
    """Host array of device pointers (None -> NULL)."""
    arr = (c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def i64_array(values):
    return (c_int64 * len(values))(*values)


def f32_array(values):
    return (c_float * len(values))(*values)


def current_stream():
    """The raw hipStream_t of torch's current stream on the current device (an int; ctypes passes it as void*).
    torch.cuda.current_stream() builds a Python Stream object per call (10 us): too slow for ~1400 launches an iteration."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def require_cuda(*tensors):
    """The kernels only take fp32 contiguous device tensors; anything else is a caller bug."""
    import torch
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise NotImplementedError(
                "savfi HIP ops need device tensors (the reference raises NotImplementedError for "
                "CPU tensors too, sepconv/sepconv_op/sepconv.py:293-294)")
        if t.dtype != torch.float32:
            raise TypeError("savfi HIP ops are fp32-only, got %s" % t.dtype)
        assert t.is_contiguous(), "savfi HIP ops need contiguous tensors"


class KernelTimer:
    """Optional HIP-event timing of individual launches (used by bench.py for the roofline line).

    Events are recorded on torch's current stream, which is the stream every savfi launch uses.
    Nothing is synchronised until `summary()` is called.
    """

    def __init__(self, only=None):
        self.records = {}
        # time only launches whose name starts with this prefix / one of these prefixes (None = all)
        self.only = (only,) if isinstance(only, str) else (None if only is None else tuple(only))

    def wants(self, name):
        return self.only is None or name.startswith(self.only)

    def launch(self, name, fn, nbytes=0, flops=0):
        import torch
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self.records.setdefault(name, []).append((a, b, nbytes, flops))

    def summary(self):
        import torch
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _, _ in evs]
            nbytes = sum(n for _, _, n, _ in evs)
            flops = sum(f for _, _, _, f in evs)
            out[name] = {"launches": len(ms), "total_ms": sum(ms), "avg_us": 1e3 * sum(ms) / len(ms),
                         "min_us": 1e3 * min(ms), "algorithmic_bytes": nbytes,
                         "achieved_GBps": (nbytes / (sum(ms) * 1e-3) / 1e9) if sum(ms) > 0 else None}
            if flops:
                out[name]["direct_flops"] = flops
                out[name]["direct_TFLOPs"] = flops / (sum(ms) * 1e-3) / 1e12 if sum(ms) > 0 else None
        return out


TIMER = None  # set to a KernelTimer instance to time launches


def launch(name, fn, nbytes=0, flops=0):
    """Run one C-ABI launch; `nbytes` = its algorithmic HBM bytes (each operand once), `flops` = its direct-convolution
    floating-point operations, for the timer."""
    if TIMER is None or not TIMER.wants(name):
        fn()
    else:
        TIMER.launch(name, fn, nbytes, flops)
