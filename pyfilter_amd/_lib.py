"""
ctypes binding of ``libpfamd.so`` (C ABI in ``include/pf_amd.h``).

PyTorch-ROCm tensors cross this boundary as raw device pointers only (``tensor.data_ptr()``), launches go to
``torch.cuda.current_stream()``.  There is **no CPU fallback**: if the shared library is missing, or a tensor is not on
a GPU, the call raises.
"""
import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpfamd.so")

PF_F32, PF_F64 = 0, 1
HID_LINEAR, HID_SINE_EM, HID_VERHULST_EM, HID_LORENZ63_EM, HID_OU, HID_USER_AFFINE = 0, 1, 2, 3, 4, 5
OBS_LINEAR, OBS_SV = 0, 1
PROP_BOOTSTRAP, PROP_LGO = 0, 1
FILTER_SISR, FILTER_APF = 0, 1
RESAMPLE_SYSTEMATIC, RESAMPLE_MULTINOMIAL = 0, 1
MAX_D = 3
MAX_O = 3

EXPORTS = (
    "pf_version", "pf_abi_version", "pf_error_string", "pf_workspace_bytes", "pf_normalize", "pf_systematic", "pf_systematic_cdf_free", "pf_systematic_logw",
    "pf_multinomial", "pf_gather", "pf_loglik", "pf_moments", "pf_pre_weight", "pf_sample_and_weight",
    "pf_initial_sample", "pf_filter_run", "pf_filter_run_timed", "pf_filter_graph_create", "pf_filter_graph_launch",
    "pf_filter_graph_destroy", "pf_columns_gather", "pf_columns_exchange", "pf_debug_draw_normals", "pf_debug_launch_trace",
    "pf_smooth_fixed_lag", "pf_smooth_ffbs", "pf_observed_flags", "pf_theta_ess", "pf_theta_fit", "pf_theta_propose",
    "pf_theta_accept", "pf_theta_path", "pf_theta_resample", "pf_initial_sample_cols", "pf_theta_step", "pf_host_alloc",
    "pf_host_free", "pf_filter_observe",
)


THETA_MAXP = 8


class PfThetaPriors(C.Structure):
    _fields_ = [("P", C.c_int32), ("kind", C.c_int32 * THETA_MAXP), ("a", C.c_double * THETA_MAXP), ("b", C.c_double * THETA_MAXP)]


class PfModel(C.Structure):
    _fields_ = [
        ("hid_kind", C.c_int32), ("obs_kind", C.c_int32), ("dim", C.c_int32), ("obs_dim", C.c_int32),
        ("dt", C.c_double), ("inc_scale", C.c_double), ("params", C.c_void_p),
    ]


ABI_VERSION = 4  # include/pf_amd.h: PF_ABI_VERSION


class PfRunHints(C.Structure):
    _fields_ = [("route", C.c_int32), ("column_max_n", C.c_int32), ("tile_target", C.c_int32), ("ancestor_search", C.c_int32),
                ("resume", C.c_int32), ("prepare_next", C.c_int32), ("cluster_generation", C.c_int32), ("cluster_patience", C.c_int32)]


class PfFilterArgs(C.Structure):
    """``pf_filter_args``; a fresh instance carries its own size (the library refuses a block of another ABI version)."""

    _fields_ = [
        ("struct_size", C.c_uint64),
        ("model", PfModel),
        ("filter", C.c_int32), ("proposal", C.c_int32), ("resampler", C.c_int32), ("dtype", C.c_int32),
        ("N", C.c_int64), ("B", C.c_int64),
        ("ess_threshold", C.c_double),
        ("seed", C.c_uint64),
        ("x", C.c_void_p * 2), ("logw", C.c_void_p * 2),
        ("anc", C.c_void_p), ("cdf", C.c_void_p), ("pos", C.c_void_p),
        ("y", C.c_void_p), ("y_rows", C.c_int64), ("observed", C.c_void_p),
        ("z_tape", C.c_void_p), ("u_tape", C.c_void_p),
        ("means", C.c_void_p), ("vars", C.c_void_p), ("ll_steps", C.c_void_p), ("ll_total", C.c_void_p),
        ("step_counter", C.c_void_p),
        ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
        ("observed_dev", C.c_void_p),
        ("ring", C.c_int64),
        ("user_loc", C.c_void_p), ("user_scale", C.c_void_p), ("user_scale_per_column", C.c_int64),
        ("user_dt", C.c_double),
        ("status", C.c_void_p),
        ("hints", PfRunHints),
    ]

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.struct_size = C.sizeof(PfFilterArgs)


_lib = None


class PfAmdError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Loads the shared library (once).  Raises if it has not been built - there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH  # (development tools load an instrumented build of the same sources by setting ``_lib.LIB_PATH`` first)
    if not os.path.exists(path):
        raise PfAmdError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  pyfilter_amd has no CPU / eager fallback."
        )
    lib = C.CDLL(path)
    lib.pf_version.restype = C.c_char_p
    lib.pf_error_string.restype = C.c_char_p
    lib.pf_error_string.argtypes = [C.c_int]
    vp, i64, i32, u64, u32, sz = C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_uint32, C.c_size_t
    lib.pf_workspace_bytes.argtypes = [i64, i64, i64, C.POINTER(sz)]
    lib.pf_normalize.argtypes = [vp, vp, vp, vp, i64, i64, i32, vp, sz, vp]
    lib.pf_systematic.argtypes = [vp, vp, i32, vp, vp, vp, i64, i64, i32, vp, sz, vp]
    lib.pf_systematic_logw.argtypes = [vp, vp, i32, vp, vp, vp, i64, i64, i32, vp, sz, vp]
    lib.pf_systematic_cdf_free.argtypes = [i64, i64, i32, i32, C.POINTER(C.c_int)]
    lib.pf_multinomial.argtypes = [vp, vp, u64, u32, vp, vp, vp, i64, i64, i32, vp, sz, vp]
    lib.pf_gather.argtypes = [vp, vp, vp, vp, i64, i64, i64, i32, vp]
    lib.pf_loglik.argtypes = [vp, vp, vp, i64, i64, i32, vp, sz, vp]
    lib.pf_columns_gather.argtypes = [vp, vp, vp, i64, i64, i64, i32, vp]
    lib.pf_columns_exchange.argtypes = [vp, vp, vp, i64, i64, i64, i32, vp]
    lib.pf_moments.argtypes = [vp, vp, vp, vp, i64, i64, i64, i32, vp, sz, vp]
    lib.pf_pre_weight.argtypes = [C.POINTER(PfModel), i32, vp, vp, i64, vp, i64, i64, i32, vp]
    lib.pf_sample_and_weight.argtypes = [C.POINTER(PfModel), i32, i32, vp, vp, i64, vp, u64, u32, vp, vp, i64, i64, i32, vp]
    lib.pf_initial_sample.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), vp, u64, vp, i64, i64, i64, i32, vp]
    lib.pf_filter_run.argtypes = [C.POINTER(PfFilterArgs), i64, i64, i32, vp]
    lib.pf_filter_run_timed.argtypes = [C.POINTER(PfFilterArgs), i64, i64, i32, vp, C.POINTER(C.c_float)]
    lib.pf_filter_graph_create.argtypes = [C.POINTER(PfFilterArgs), i64, i64, i32, vp, C.POINTER(vp)]
    lib.pf_filter_graph_launch.argtypes = [vp, vp]
    lib.pf_filter_graph_destroy.argtypes = [vp]
    lib.pf_smooth_fixed_lag.argtypes = [vp, vp, vp, i64, i64, i64, i64, i32, vp]
    lib.pf_smooth_ffbs.argtypes = [C.POINTER(PfModel), vp, vp, vp, vp, u64, vp, i64, i64, i64, i32, vp]
    lib.pf_observed_flags.argtypes = [vp, i64, i64, i32, vp, vp]
    lib.pf_theta_ess.argtypes = [vp, i64, i64, i32, vp, vp]
    lib.pf_theta_path.argtypes = [vp, vp, i64, i64, i32, vp, vp, vp, u64, vp, vp]
    lib.pf_theta_resample.argtypes = [vp, i64, C.c_double, i32, vp, vp, vp]
    lib.pf_theta_step.argtypes = [vp, vp, i64, i32, vp, vp, u64, vp, vp, vp]
    lib.pf_filter_observe.argtypes = [C.POINTER(PfFilterArgs), i64, i64, i32, vp, vp, vp, vp, u64, vp, vp]
    lib.pf_host_alloc.argtypes = [sz, C.POINTER(vp)]
    lib.pf_host_free.argtypes = [vp]
    lib.pf_initial_sample_cols.argtypes = [vp, i64, i64, vp, i64, i64, vp, u64, vp, i64, i64, i64, i32, vp]
    lib.pf_theta_fit.argtypes = [vp, vp, i64, i32, C.c_double, i32, vp, vp, vp]
    lib.pf_theta_propose.argtypes = [C.POINTER(PfThetaPriors), vp, vp, vp, i64, i32, vp, C.POINTER(vp), vp, vp]
    lib.pf_theta_accept.argtypes = [vp] * 11 + [i64, i32, i32, vp, vp, vp, vp]
    lib.pf_debug_draw_normals.argtypes = [u64, u32, i64, vp, i64, i64, i64, i32, vp]
    lib.pf_debug_launch_trace.argtypes = [C.POINTER(C.c_int32), i32]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("pf_version", "pf_error_string"):
            fn.restype = C.c_int
    if lib.pf_abi_version() != ABI_VERSION:
        raise PfAmdError(f"{path} implements ABI {lib.pf_abi_version()}, this package speaks ABI {ABI_VERSION}: rebuild it "
                         "(python -c 'import __graft_entry__ as g; g.build()')")
    _lib = lib
    return lib


def version() -> str:
    """``pf_version()``: library version, ABI and the sha256 of the sources the binary was built from."""
    return load().pf_version().decode()


def check(rc: int, what: str):
    if rc != 0:
        raise PfAmdError(f"{what} failed: {load().pf_error_string(rc).decode()} (code {rc})")


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return PF_F32
    if dtype == torch.float64:
        return PF_F64
    raise PfAmdError(f"unsupported dtype {dtype}: the HIP kernels compute in float32 or float64")


def require_gpu(*tensors: Optional[torch.Tensor]):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise PfAmdError(
                "pyfilter_amd runs on MI355X only: got a CPU tensor (there is no CPU fallback; move the model / data "
                "to 'cuda')."
            )


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr() -> int:
    """The caller's current HIP stream (raw handle) - the fast accessor: ``torch.cuda.current_stream()`` builds a Python
    object per call (~15 us), which the online filter move pays once per observation."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


_ws_cache = {}


def workspace(n: int, b: int, device: torch.device) -> torch.Tensor:
    """Scratch for the stand-alone primitives, cached per (N, B, device, stream)."""
    key = (n, b, device.index, stream_ptr())
    ws = _ws_cache.get(key)
    nbytes = C.c_size_t(0)
    if ws is None:  # (the size is a bound over every tile geometry: include/pf_amd.h)
        check(load().pf_workspace_bytes(n, b, MAX_D, C.byref(nbytes)), "pf_workspace_bytes")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
        if len(_ws_cache) > 64:
            _ws_cache.clear()
        _ws_cache[key] = ws
    return ws


def new_workspace(n: int, b: int, device: torch.device) -> torch.Tensor:
    nbytes = C.c_size_t(0)
    check(load().pf_workspace_bytes(n, b, MAX_D, C.byref(nbytes)), "pf_workspace_bytes")
    return torch.zeros(nbytes.value, dtype=torch.uint8, device=device)
