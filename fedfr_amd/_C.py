"""ctypes binding of libfedfr_hip.so (C ABI: include/fedfr_hip.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError carrying
``fedfr_last_error_string()`` is raised.  Device memory is owned by torch tensors; raw
``data_ptr()`` values and the current HIP stream handle are passed through.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# FEDFR_HIP_LIB_NAME selects another build of the same library in this directory (same-box A/B of compiler options)
LIB_PATH = os.path.join(_HERE, os.environ.get("FEDFR_HIP_LIB_NAME", "libfedfr_hip.so"))

_lib: Optional[C.CDLL] = None

vp, i32, i64, u64, f32, f64, sz = C.c_void_p, C.c_int, C.c_longlong, C.c_ulonglong, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes).  Must list every symbol include/fedfr_hip.h declares (tests/test_abi.py checks).
SIGNATURES = {
    "fedfr_version": (i32, []),
    "fedfr_storage_dtype": (i32, []),
    "fedfr_last_error_string": (C.c_char_p, []),
    "fedfr_set_option": (i32, [C.c_char_p, i32]),
    "fedfr_get_option": (i32, [C.c_char_p, C.POINTER(i32)]),
    "fedfr_option_count": (i32, []),
    "fedfr_option_info": (i32, [i32, C.POINTER(C.c_char_p), C.POINTER(i32), C.POINTER(i32)]),
    "fedfr_profile_enable": (i32, [i32]),
    "fedfr_profile_read": (i32, [i32, C.POINTER(f64), C.POINTER(i64), C.POINTER(f64)]),
    "fedfr_profile_read_bytes": (i32, [i32, C.POINTER(f64)]),
    "fedfr_net_create": (vp, [C.POINTER(i32), i32, i32, i32]),
    "fedfr_block_create": (vp, [i32, i32, i32, i32, i32]),
    "fedfr_net_create_sphere": (vp, [i32, i32]),
    "fedfr_net_debug_capture": (i32, [vp, sz]),
    "fedfr_net_set_dropout": (i32, [vp, f32, u64, C.POINTER(i64)]),
    "fedfr_net_set_dropout_step": (i32, [vp, u64]),
    "fedfr_net_destroy": (None, [vp]),
    "fedfr_net_query": (i32, [vp, i32, C.POINTER(i64)]),
    "fedfr_net_tensor_info": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64),
                                    C.POINTER(i32), C.POINTER(i32)]),
    "fedfr_net_act_info": (i32, [vp, i32, i32, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)]),
    "fedfr_net_prepare_weights": (i32, [vp, vp, vp, i32, vp]),
    "fedfr_net_forward": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "fedfr_net_backward": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "fedfr_net_backward2": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "fedfr_net_f32_arena_floats": (sz, [vp]),
    "fedfr_net_f32_ws_floats": (sz, [vp]),
    "fedfr_net_f32_forward": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "fedfr_net_f32_backward": (i32, [vp, vp, vp, vp, vp, vp, vp]),
    "fedfr_net_backward2_sgd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, i32, C.POINTER(i64), vp, vp]),
    "fedfr_net_backward2_sgd_scaled": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, i32, f32, vp, C.POINTER(i64), vp, vp]),
    "fedfr_conv2d_stat_rows": (i32, [i32, i32, i32]),
    "fedfr_conv2d_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "fedfr_conv2d_dgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "fedfr_conv2d_dgrad_bnbwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i32), vp]),
    "fedfr_stream_create_low_priority": (i32, [vp]),
    "fedfr_stream_destroy": (i32, [vp]),
    "fedfr_conv2d_wgrad_ws_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "fedfr_conv2d_wgrad": (i32, [vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, vp]),
    "fedfr_conv2d_wgrad_pair": (i32, [vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, vp]),
    "fedfr_weight_shadows": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "fedfr_gemm_nt": (i32, [vp, vp, vp, vp, sz, i32, i32, i32, vp]),
    "fedfr_gemm_tn": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "fedfr_stem_stat_rows": (i32, [i32, i32]),
    "fedfr_stem_fwd": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "fedfr_stem_wgrad_ws_bytes": (sz, [i32, i32]),
    "fedfr_stem_wgrad": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "fedfr_bn_finalize": (i32, [vp, i32, i32, f64, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp]),
    "fedfr_bn_apply_stat_rows": (i32, [i32, i32]),
    "fedfr_bn_apply": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "fedfr_bn_bwd_rows": (i32, [i32, i32]),
    "fedfr_bn_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]),
    "fedfr_bn_sliced_rows": (i32, [i32, i32, i32]),
    "fedfr_bn_sliced_ok": (i32, [i32, i32, i32, i32]),
    "fedfr_conv2d_fwd_moments": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, C.POINTER(i32), vp]),
    "fedfr_bn_apply2_sliced_ok": (i32, [i32, i32, i32]),
    "fedfr_bn_apply2_sliced": (i32, [vp, i32, f64, f32, f32] + [vp] * 22 + [i32, i32, vp]),
    "fedfr_bn_apply_sliced": (i32, [vp, i32, f64, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    "fedfr_bn_bwd_sliced": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "fedfr_normalize_rows": (i32, [vp, vp, vp, i32, i32, f32, vp]),
    "fedfr_normalize_rows_bwd": (i32, [vp, vp, vp, vp, i32, i32, f32, vp]),
    "fedfr_sgemm": (i32, [vp, vp, vp, i32, i32, i32, i64, i64, i64, i64, i32, f32, f32, vp, vp]),
    "fedfr_sgemm_splitk": (i32, [vp, vp, vp, i32, i32, i32, i64, i64, i64, i64, i32, f32, i32, i64, vp]),
    "fedfr_softmax_ce_fused": (i32, [vp, vp, i32, i32, i32, f32, f32, i32, f32, vp, i32, i64, vp]),
    "fedfr_normalize_rows_bwd_slabs": (i32, [vp, vp, vp, i32, i64, vp, i32, i32, f32, vp]),
    "fedfr_margin_rowmax": (i32, [vp, vp, i32, i32, i32, f32, f32, i32, vp, vp, vp]),
    "fedfr_exp_rowsum": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "fedfr_softmax_grad": (i32, [vp, vp, i32, i32, i32, vp, vp, f32, f32, vp, vp]),
    "fedfr_margin_bwd": (i32, [vp, vp, vp, f32, i32, i32, vp, vp]),
    "fedfr_nll_mean": (i32, [vp, i32, f32, vp, vp]),
    "fedfr_exp_rowsum_target": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
    "fedfr_nll_mean_ratio": (i32, [vp, vp, i32, f32, vp, vp]),
    "fedfr_bce_logits": (i32, [vp, vp, vp, i32, i32, f32, f32, f32, vp, vp, vp, vp]),
    "fedfr_bce_loss": (i32, [vp, vp, vp, i32, i32, f32, f32, f32, vp, vp, vp, vp]),
    "fedfr_colsum_f32": (i32, [vp, i32, i32, vp, vp]),
    "fedfr_sgemm_colflag": (i32, [vp, vp, i32, i32, i32, i64, i64, i64, i64, f32, f32, vp, vp]),
    "fedfr_class_accumulate": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
    "fedfr_bias_prelu_bwd": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "fedfr_pad_input_nhwc": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "fedfr_preprocess_u8": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "fedfr_roc_histogram": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "fedfr_contrastive": (i32, [vp, vp, vp, i32, i32, f32, vp, vp, vp]),
    "fedfr_sum_scale": (i32, [vp, i32, f32, vp, vp]),
    "fedfr_sgd_step": (i32, [vp, vp, vp, vp, sz, f32, f32, f32, i32, vp]),
    "fedfr_sgd_step_scaled": (i32, [vp, vp, vp, vp, sz, f32, f32, f32, i32, f32, vp, vp]),
    "fedfr_fedavg_axpy": (i32, [vp, vp, f32, sz, i32, vp]),
    "fedfr_fedavg_multi": (i32, [vp, vp, vp, i32, sz, i32, vp]),
    "fedfr_fedavg_i64": (i32, [vp, vp, f32, i32, i32, vp, vp]),
    "fedfr_pfc_rand": (i32, [vp, i32, u64, u64, vp]),
    "fedfr_pfc_localize": (i32, [vp, i32, i64, i32, vp, vp]),
    "fedfr_pfc_topk": (i32, [vp, i32, i32, vp, vp, vp]),
    "fedfr_pfc_positive": (i32, [vp, i32, vp, vp, vp]),
    "fedfr_pfc_remap": (i32, [vp, i32, vp, i32, vp]),
    "fedfr_rows_gather": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "fedfr_rows_scatter": (i32, [vp, vp, vp, i32, i32, i32, vp]),
}

# query keys (include/fedfr_hip.h)
Q_PARAM_COUNT, Q_TRAINABLE_COUNT, Q_BUFFER_COUNT, Q_NBT_COUNT, Q_SHADOW_COUNT, Q_ACT_BYTES, Q_WS_BYTES, \
    Q_NUM_TENSORS, Q_FC_IN = range(9)


def lib() -> C.CDLL:
    """Load (once) and return the library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "fedfr_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C fedfr_amd/csrc`.  There is no CPU/PyTorch fallback for the hot path." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
        # tuning / validation switches: FEDFR_OPTIONS="wgrad9p_bg=0,tn_glds=0" (see fedfr_set_option)
        for kv in filter(None, os.environ.get("FEDFR_OPTIONS", "").split(",")):
            k, v = kv.split("=")
            if l.fedfr_set_option(k.strip().encode(), int(v)) != 0:
                raise RuntimeError("FEDFR_OPTIONS: " + l.fedfr_last_error_string().decode())
    return _lib


def storage_dtype() -> torch.dtype:
    """torch dtype of the library's 16-bit storage (activations, shadows): float16 for the product library (libfedfr_hip.so), bfloat16 for libfedfr_hip_bf16.so."""
    return torch.float16 if lib().fedfr_storage_dtype() == 1 else torch.bfloat16


def loss_scale() -> float:
    """INITIAL loss scale applied to the gradient entering a backbone's backward pass (and removed from the parameter gradients): FEDFR_LOSS_SCALE
    (default 256) for the fp16-storage library, whose activation gradients would otherwise fall into fp16's subnormals (the reference's fp16
    autocast uses a GradScaler for the same reason, client.py:301,394-396); 1 for the bf16-storage build (fp32 exponent range).  The value in
    force on a device — lowered after an overflow, grown back afterwards — is ``loss_scale_state(device).scale``."""
    if lib().fedfr_storage_dtype() != 1:
        return 1.0
    return float(os.environ.get("FEDFR_LOSS_SCALE", "256"))


class LossScaleState:
    """The loss scale of ONE device, alive for the whole process (trainers come and go every FL round: the reference keeps ONE GradScaler per
    client process, client.py:301).  ``word`` is the device word the update kernels set when they skipped a non-finite gradient element
    (fedfr_sgd_step_scaled); ``poll()`` reads it (host sync: call it where the host has just drained a loss anyway), halves the scale after
    an overflow (GradScaler's backoff_factor 0.5) and doubles it again after ``growth_interval`` clean steps, never beyond the initial scale
    (the static scale is sized for the path's parity; there is nothing to gain above it)."""

    growth_interval = 2000          # torch.cuda.amp.GradScaler's default

    def __init__(self, device):
        self.initial = self.scale = loss_scale()
        self.enabled = self.initial != 1.0          # fp16-storage library: every update goes through the guarded kernels
        self.word = torch.zeros(1, dtype=torch.int32, device=device)
        self.overflows = 0                          # overflowing intervals seen so far
        self.clean_steps = 0                        # steps since the last overflow that a poll has confirmed clean
        self.unpolled_steps = 0

    def count_step(self, n: int = 1):
        self.unpolled_steps += n

    def poll(self) -> bool:
        """True if an update kernel skipped non-finite elements since the last poll (then: scale halved, warning issued)."""
        if not self.enabled:
            return False
        steps, self.unpolled_steps = self.unpolled_steps, 0
        if int(self.word.item()) == 0:
            self.clean_steps += steps
            if self.scale < self.initial and self.clean_steps >= self.growth_interval:
                self.scale = min(self.initial, self.scale * 2.0)
                self.clean_steps = 0
            return False
        import warnings
        self.word.zero_()
        self.overflows += 1
        self.clean_steps = 0
        self.scale = max(2.0 ** -14, self.scale * 0.5)
        warnings.warn("fedfr_amd: non-finite gradients under the fp16 loss scale — the affected parameter elements were not updated; "
                      "loss scale lowered to %g" % self.scale)
        return True


_LOSS_SCALE_STATES = {}


def loss_scale_state(device) -> LossScaleState:
    """The process-wide LossScaleState of ``device`` (created on first use)."""
    key = str(torch.device(device))
    st = _LOSS_SCALE_STATES.get(key)
    if st is None:
        st = _LOSS_SCALE_STATES[key] = LossScaleState(device)
    return st


def last_error() -> str:
    return lib().fedfr_last_error_string().decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise RuntimeError("fedfr_hip %s failed (rc=%d): %s" % (what, rc, last_error()))


def call(name: str, *args) -> None:
    """Call an int-returning entry point and raise on failure."""
    check(getattr(lib(), name)(*args), name)


def get_option(name: str) -> int:
    v = C.c_int(0)
    call("fedfr_get_option", name.encode(), C.byref(v))
    return v.value


def options() -> dict:
    """{name: (current value, library default)} of every switch the loaded library has."""
    out = {}
    for i in range(lib().fedfr_option_count()):
        n, v, d = C.c_char_p(), C.c_int(0), C.c_int(0)
        call("fedfr_option_info", i, C.byref(n), C.byref(v), C.byref(d))
        out[n.value.decode()] = (v.value, d.value)
    return out


def options_non_default() -> dict:
    """{name: value} of the switches that are not at the value the library starts with (FEDFR_OPTIONS, option_scope, set_option)."""
    return {k: v for k, (v, d) in options().items() if v != d}


class option_scope:
    """``with option_scope("wgrad9p", 1): ...`` — set a library switch for the duration of a block and put the PREVIOUS value back (a
    user's FEDFR_OPTIONS setting survives; the switches are process-global)."""

    def __init__(self, name: str, value: int):
        self.name, self.value, self.prev = name, int(value), None

    def __enter__(self):
        self.prev = get_option(self.name)
        call("fedfr_set_option", self.name.encode(), self.value)
        return self

    def __exit__(self, *exc):
        call("fedfr_set_option", self.name.encode(), self.prev)
        return False


def stream(ref=None) -> int:
    """HIP stream handle every call is enqueued on: the current stream of ``ref``'s device (a tensor or a device), of the current
    device if omitted.  Kernels touch raw pointers, so the stream MUST belong to the tensors' device: ``require_gpu_tensor`` rejects
    tensors of a non-current device and the package's entry points run under ``on_device`` (below)."""
    if ref is None:
        return torch.cuda.current_stream().cuda_stream
    dev = ref.device if isinstance(ref, torch.Tensor) else torch.device(ref)
    return torch.cuda.current_stream(dev).cuda_stream


def on_device(get_device):
    """Decorator for methods that enqueue kernels: makes ``get_device(self)`` the current device for the duration of the call, so that
    ``stream()`` — and every allocation — lands on the device the object's tensors live on (a ``Client(device=cuda:1)`` used while
    cuda:0 is current would otherwise enqueue on the wrong device's stream)."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(self, *a, **k):
            dev = get_device(self)
            if dev is None or torch.device(dev).type != "cuda":
                return fn(self, *a, **k)
            with torch.cuda.device(dev):
                return fn(self, *a, **k)
        return wrapped
    return deco


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def require_gpu_tensor(t: torch.Tensor, dtype=None, name: str = "tensor") -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("fedfr_amd: %s must live on an MI355X device (got %s); no CPU fallback exists" % (name, t.device))
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        raise RuntimeError("fedfr_amd: %s lives on %s but the current device is cuda:%d — kernels are enqueued on the current device's "
                           "stream; wrap the call in `with torch.cuda.device(%d):`" % (name, t.device, torch.cuda.current_device(), t.device.index))
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("fedfr_amd: %s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("fedfr_amd: %s must be contiguous" % name)
    return t
