"""
Thin python wrappers over the C ABI working on the library's *internal* layouts:

* ``cols``: ``(B, N)`` contiguous - one row per filter of the batch dim ("column" in the kernels' vocabulary);
* ``soa`` : ``(D, B, N)`` contiguous - the particle ensemble, one plane per state component.

The reference-layout API (``(N, [B], [D])``, particles first: pyfilter/filters/particle/base.py:51-62) lives in
``pyfilter_amd.utils`` / ``resampling`` / ``filters`` and hands *views* of these buffers to the user.
"""
import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib as L


# ----------------------------------------------------------------------------------------------------------------
# layout helpers (views where possible, copies only for foreign layouts)
# ----------------------------------------------------------------------------------------------------------------
def to_cols(w: torch.Tensor) -> torch.Tensor:
    """``(N,)`` or ``(N, B)`` -> ``(B, N)`` contiguous (no copy if ``w`` already is a view of such a buffer)."""
    if w.dim() == 1:
        return w.contiguous().unsqueeze(0)
    if w.dim() != 2:
        raise L.PfAmdError("weights must be (N,) or (N, B): nested batches are not supported (filters/base.py:116-119)")
    return w.t().contiguous()


def from_cols(c: torch.Tensor, batched: bool) -> torch.Tensor:
    """``(B, N)`` -> the reference's ``(N, B)`` view, or ``(N,)`` when unbatched."""
    return c.t() if batched else c[0]


def to_soa(x: torch.Tensor, batched: bool, has_event: bool) -> torch.Tensor:
    """``(N, [B], [D])`` -> ``(D, B, N)`` contiguous."""
    if not has_event:
        x = x.unsqueeze(-1)
    if not batched:
        x = x.unsqueeze(1)
    return x.permute(2, 1, 0).contiguous()


def from_soa(s: torch.Tensor, batched: bool, has_event: bool) -> torch.Tensor:
    """``(D, B, N)`` -> the reference's ``(N, [B], [D])`` view."""
    v = s.permute(2, 1, 0)
    if not batched:
        v = v[:, 0]
    if not has_event:
        v = v[..., 0]
    return v


def _mask_ptr(colmask: Optional[torch.Tensor]):
    if colmask is None:
        return None
    assert colmask.dtype in (torch.uint8, torch.bool) and colmask.is_contiguous()
    return colmask.data_ptr()


# ----------------------------------------------------------------------------------------------------------------
# whole-filter moves along the batch dim
# ----------------------------------------------------------------------------------------------------------------
def _batch_view(t: torch.Tensor):
    """``(N, B, ...)`` reference-layout tensor -> (``(planes, B, N)`` contiguous buffer, rebuild) where ``rebuild`` maps
    such a buffer back to the reference layout.  No copy when ``t`` already is a view of a library buffer."""
    if t.dim() < 2:
        raise L.PfAmdError("a batch dimension is required: tensor must be (N, B, ...)")
    n, b, tail = t.shape[0], t.shape[1], tuple(t.shape[2:])
    flat = t.unsqueeze(-1) if t.dim() == 2 else (t if t.dim() == 3 else t.reshape(n, b, -1))
    buf = flat.permute(2, 1, 0).contiguous()  # (planes, B, N); a no-op for views of (D, B, N) / (B, N) buffers

    def rebuild(o: torch.Tensor) -> torch.Tensor:
        v = o.permute(2, 1, 0)
        if len(tail) == 0:
            return v[..., 0]
        return v if len(tail) == 1 else v.reshape(n, b, *tail)

    return buf, rebuild


_checked_index = None  # (tensor, its in-place version, bound): the last ancestor vector that passed the bounds check


def _check_filter_index(idx: torch.Tensor, b: int):
    """Bounds check of a vector of filter ancestors without a host round trip where torch offers the asynchronous assert
    (as torch's own CUDA indexing does).  A resampling moves every buffer of a filter set with the SAME vector (three to
    six gathers per rejuvenation): it is checked once - five small launches - not once per buffer."""
    global _checked_index
    c = _checked_index
    if c is not None and c[0] is idx and c[1] == idx._version and c[2] == b:
        return
    ok = ((idx >= -b) & (idx < b)).all()
    if hasattr(torch, "_assert_async") and not SYNC_CHECKS:
        torch._assert_async(ok)
    elif not bool(ok):
        raise IndexError(f"index out of bounds for dimension 1 with size {b}")
    _checked_index = (idx, idx._version, b)


def gather_columns(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """``src[:, idx]`` for a library buffer ``(planes, B, N)`` and ``B`` int64 ancestors: a fresh buffer (pf_columns_gather)."""
    L.require_gpu(src, idx)
    planes, b, n = src.shape
    assert src.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() == b
    _check_filter_index(idx, b)
    dst = torch.empty_like(src)
    L.check(L.load().pf_columns_gather(src.data_ptr(), idx.data_ptr(), dst.data_ptr(), n, b, planes, src.element_size(),
                                       L.stream_ptr()), "pf_columns_gather")
    return dst


def exchange_columns(dst: torch.Tensor, src: torch.Tensor, mask: torch.Tensor):
    """``dst[:, mask] = src[:, mask]`` in place for library buffers ``(planes, B, N)`` (pf_columns_exchange)."""
    L.require_gpu(dst, src, mask)
    planes, b, n = dst.shape
    assert dst.is_contiguous() and src.is_contiguous() and src.shape == dst.shape and src.dtype == dst.dtype
    assert mask.dtype == torch.bool and mask.is_contiguous() and mask.numel() == b
    L.check(L.load().pf_columns_exchange(dst.data_ptr(), src.data_ptr(), mask.data_ptr(), n, b, planes, dst.element_size(),
                                         L.stream_ptr()), "pf_columns_exchange")


def gather_filters(t: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
    """``t[:, indices]`` for an ``(N, B, ...)`` tensor, as a view of a freshly gathered ``(planes, B, N)`` buffer
    (pf_columns_gather; ParticleFilterCorrection.resample, particle/state.py:150-158)."""
    L.require_gpu(t, indices)
    if indices.dtype == torch.bool:
        indices = indices.nonzero().reshape(-1)
    idx = indices.to(torch.int64).contiguous()
    b = t.shape[1]
    if idx.numel() != b:
        return t[:, indices]  # a different number of filters out than in: not the in-place resample of the reference
    _check_filter_index(idx, b)
    src, rebuild = _batch_view(t)
    dst = torch.empty_like(src)
    planes, _, n = src.shape
    L.check(L.load().pf_columns_gather(src.data_ptr(), idx.data_ptr(), dst.data_ptr(), n, b, planes, src.element_size(),
                                       L.stream_ptr()), "pf_columns_gather")
    return rebuild(dst)


def exchange_filters(dst_t: torch.Tensor, src_t: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """``dst[:, mask] = src[:, mask]`` for ``(N, B, ...)`` tensors (pf_columns_exchange; ParticleFilterCorrection.exchange,
    particle/state.py:160-168).  In place when ``dst_t`` is a view of a library buffer; returns the updated tensor."""
    L.require_gpu(dst_t, src_t, mask)
    if mask.dtype != torch.bool or mask.dim() != 1 or mask.numel() != dst_t.shape[1] or dst_t.shape != src_t.shape \
            or dst_t.dtype != src_t.dtype:
        dst_t[:, mask] = src_t[:, mask]
        return dst_t
    dst, rebuild = _batch_view(dst_t)
    src, _ = _batch_view(src_t)
    planes, b, n = dst.shape
    m = mask.contiguous()
    L.check(L.load().pf_columns_exchange(dst.data_ptr(), src.data_ptr(), m.data_ptr(), n, b, planes, dst.element_size(),
                                         L.stream_ptr()), "pf_columns_exchange")
    out = rebuild(dst)
    if out.data_ptr() != dst_t.data_ptr():  # dst_t was in a foreign layout: write the result back into it
        dst_t.copy_(out)
        return dst_t
    return out


# ----------------------------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------------------------
def normalize_cols(logw: torch.Tensor, want_w=True, want_lse=False, want_ess=False):
    """pf_normalize on a ``(B, N)`` buffer (sanitised in place).  Returns (W|None, lse|None, ess|None)."""
    L.require_gpu(logw)
    b, n = logw.shape
    W = torch.empty_like(logw) if want_w else None
    lse = torch.empty(b, dtype=logw.dtype, device=logw.device) if want_lse else None
    ess = torch.empty(b, dtype=logw.dtype, device=logw.device) if want_ess else None
    ws = L.workspace(n, b, logw.device)
    L.check(
        L.load().pf_normalize(L.ptr(logw), L.ptr(W), L.ptr(lse), L.ptr(ess), n, b, L.dtype_code(logw.dtype),
                              ws.data_ptr(), ws.numel(), L.stream_ptr()),
        "pf_normalize",
    )
    return W, lse, ess


def systematic_cols(w: torch.Tensor, u: torch.Tensor, normalized: bool, colmask: Optional[torch.Tensor] = None,
                    idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Systematic resampling of ``(B, N)`` weights (``normalized``) or log-weights; returns int32 ``(B, N)``."""
    L.require_gpu(w, u)
    b, n = w.shape
    per_elem = 1 if (u.dim() == 2 and u.shape == (b, n) and n > 1) else 0
    u = (u if per_elem else u.reshape(-1)).to(w.dtype).contiguous()
    assert per_elem or u.numel() == b, f"u must hold one uniform per column: {u.numel()} != {b}"
    if idx is None:
        idx = torch.empty((b, n), dtype=torch.int32, device=w.device)
        if colmask is not None:
            idx.copy_(torch.arange(n, device=w.device, dtype=torch.int32).unsqueeze(0).expand(b, n))
    # several tiles per column: no cdf is materialised (two launches; include/pf_amd.h: pf_systematic, cdf == NULL)
    cdf = None if (SYSTEMATIC_CDF_FREE and _cdf_free(n, b, w.dtype, per_elem)) else torch.empty_like(w)
    ws = L.workspace(n, b, w.device)
    fn = L.load().pf_systematic if normalized else L.load().pf_systematic_logw
    L.check(
        fn(L.ptr(w), L.ptr(u), per_elem, _mask_ptr(colmask), L.ptr(cdf), L.ptr(idx), n, b, L.dtype_code(w.dtype),
           ws.data_ptr(), ws.numel(), L.stream_ptr()),
        "pf_systematic",
    )
    return idx


SYSTEMATIC_CDF_FREE = True  # (tests switch it off to run the three-launch form, the one a C caller that wants the cdf gets)
_CDF_FREE = {}


def _cdf_free(n: int, b: int, dtype, per_elem: int) -> bool:
    key = (n, b, dtype, per_elem)
    hit = _CDF_FREE.get(key)
    if hit is None:
        yes = C.c_int(0)
        L.check(L.load().pf_systematic_cdf_free(n, b, L.dtype_code(dtype), per_elem, C.byref(yes)), "pf_systematic_cdf_free")
        if len(_CDF_FREE) > 256:
            _CDF_FREE.clear()
        hit = _CDF_FREE[key] = bool(yes.value)
    return hit


def multinomial_cols(W: torch.Tensor, seed: int, step: int = 0, v: Optional[torch.Tensor] = None,
                     colmask: Optional[torch.Tensor] = None, idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    L.require_gpu(W, v)
    b, n = W.shape
    if idx is None:
        idx = torch.empty((b, n), dtype=torch.int32, device=W.device)
        if colmask is not None:
            idx.copy_(torch.arange(n, device=W.device, dtype=torch.int32).unsqueeze(0).expand(b, n))
    cdf = torch.empty_like(W)
    ws = L.workspace(n, b, W.device)
    L.check(
        L.load().pf_multinomial(L.ptr(W), L.ptr(v), seed, step, _mask_ptr(colmask), L.ptr(cdf), L.ptr(idx), n, b,
                                L.dtype_code(W.dtype), ws.data_ptr(), ws.numel(), L.stream_ptr()),
        "pf_multinomial",
    )
    return idx


def gather_soa(x: torch.Tensor, idx: torch.Tensor, colmask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out[d, b, i] = x[d, b, idx[b, i]]`` (columns with ``colmask == 0`` copied through)."""
    L.require_gpu(x, idx)
    d, b, n = x.shape
    assert idx.dtype == torch.int32 and idx.shape == (b, n) and idx.is_contiguous()
    out = torch.empty_like(x)
    L.check(
        L.load().pf_gather(L.ptr(x), L.ptr(idx), _mask_ptr(colmask), L.ptr(out), n, b, d, L.dtype_code(x.dtype),
                           L.stream_ptr()),
        "pf_gather",
    )
    return out


def loglik_cols(v: torch.Tensor, W: Optional[torch.Tensor]) -> torch.Tensor:
    L.require_gpu(v, W)
    b, n = v.shape
    out = torch.empty(b, dtype=v.dtype, device=v.device)
    ws = L.workspace(n, b, v.device)
    L.check(
        L.load().pf_loglik(L.ptr(v), L.ptr(W), L.ptr(out), n, b, L.dtype_code(v.dtype), ws.data_ptr(), ws.numel(),
                           L.stream_ptr()),
        "pf_loglik",
    )
    return out


def moments_soa(x: torch.Tensor, W: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Weighted mean / variance per column: returns ``(B, D)`` tensors."""
    L.require_gpu(x, W)
    d, b, n = x.shape
    mean = torch.empty((b, d), dtype=x.dtype, device=x.device)
    var = torch.empty((b, d), dtype=x.dtype, device=x.device)
    ws = L.workspace(n, b, x.device)
    L.check(
        L.load().pf_moments(L.ptr(x), L.ptr(W), L.ptr(mean), L.ptr(var), n, b, d, L.dtype_code(x.dtype),
                            ws.data_ptr(), ws.numel(), L.stream_ptr()),
        "pf_moments",
    )
    return mean, var


# ----------------------------------------------------------------------------------------------------------------
# built-in model kernels
# ----------------------------------------------------------------------------------------------------------------
def make_model_struct(kind, params: torch.Tensor) -> L.PfModel:
    m = L.PfModel()
    m.hid_kind, m.obs_kind, m.dim, m.obs_dim = kind.hid_kind, kind.obs_kind, kind.dim, kind.obs_dim
    m.dt, m.inc_scale = float(kind.dt), float(kind.inc_scale)
    m.params = params.data_ptr()
    return m


def _y_rows(y: torch.Tensor, b: int, obs_dim: int) -> Tuple[torch.Tensor, int]:
    """Canonical ``(rows, O)`` device layout of one observation: rows = 1 (shared) or B (one series per filter)."""
    yy = y.reshape(-1, obs_dim) if y.numel() != obs_dim else y.reshape(1, obs_dim)
    if yy.shape[0] not in (1, b):
        raise L.PfAmdError(f"observation of shape {tuple(y.shape)} does not broadcast against batch {b}")
    return yy.contiguous(), yy.shape[0]


def pre_weight_soa(kind, params, proposal: int, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    L.require_gpu(x, y, params)
    d, b, n = x.shape
    yy, rows = _y_rows(y.to(x.dtype), b, kind.obs_dim)
    out = torch.empty((b, n), dtype=x.dtype, device=x.device)
    m = make_model_struct(kind, params)
    L.check(
        L.load().pf_pre_weight(C.byref(m), proposal, L.ptr(x), L.ptr(yy), rows, L.ptr(out), n, b,
                               L.dtype_code(x.dtype), L.stream_ptr()),
        "pf_pre_weight",
    )
    return out


def sample_and_weight_soa(kind, params, proposal: int, x: torch.Tensor, y: Optional[torch.Tensor],
                          z: Optional[torch.Tensor], seed: int, step: int, weigh: bool = True):
    L.require_gpu(x, y, z, params)
    d, b, n = x.shape
    x_out = torch.empty_like(x)
    w_out = torch.empty((b, n), dtype=x.dtype, device=x.device) if weigh else None
    yy, rows = (None, 1)
    if weigh:
        yy, rows = _y_rows(y.to(x.dtype), b, kind.obs_dim)
    if z is not None:
        assert z.shape == x.shape and z.is_contiguous() and z.dtype == x.dtype
    m = make_model_struct(kind, params)
    L.check(
        L.load().pf_sample_and_weight(C.byref(m), proposal, 1 if weigh else 0, L.ptr(x), L.ptr(yy), rows, L.ptr(z),
                                      seed, step, L.ptr(x_out), L.ptr(w_out), n, b, L.dtype_code(x.dtype),
                                      L.stream_ptr()),
        "pf_sample_and_weight",
    )
    return x_out, w_out


def initial_sample_soa(m0, s0, n: int, b: int, d: int, dtype, device, seed: int, z: Optional[torch.Tensor] = None):
    x = torch.empty((d, b, n), dtype=dtype, device=device)
    L.require_gpu(x, z)
    am = (C.c_double * d)(*[float(v) for v in m0])
    asd = (C.c_double * d)(*[float(v) for v in s0])
    L.check(
        L.load().pf_initial_sample(am, asd, L.ptr(z), seed, L.ptr(x), n, b, d, L.dtype_code(dtype), L.stream_ptr()),
        "pf_initial_sample",
    )
    return x


def initial_sample_cols(m0: torch.Tensor, s0: torch.Tensor, n: int, b: int, d: int, seed: int, z: Optional[torch.Tensor] = None):
    """``(D, B, N)`` initial particles ``m0 + s0 * z`` for per-filter initial parameters: ``m0`` / ``s0`` are ``(B, D)``
    tensors or broadcast views of that shape (their strides travel, nothing is materialised) - pf_initial_sample_cols."""
    assert tuple(m0.shape) == (b, d) and tuple(s0.shape) == (b, d) and m0.dtype == s0.dtype and m0.device == s0.device
    x = torch.empty((d, b, n), dtype=m0.dtype, device=m0.device)
    L.require_gpu(x, z, m0, s0)
    (mb, md), (sb, sd) = m0.stride(), s0.stride()
    L.check(L.load().pf_initial_sample_cols(m0.data_ptr(), mb, md, s0.data_ptr(), sb, sd, L.ptr(z), seed, x.data_ptr(), n, b, d,
                                            L.dtype_code(x.dtype), L.stream_ptr()), "pf_initial_sample_cols")
    return x


# ----------------------------------------------------------------------------------------------------------------
# test support
# ----------------------------------------------------------------------------------------------------------------
SYNC_CHECKS = False  # True: index bounds of whole-filter moves are checked with a host round trip (an IndexError) instead
# of torch's asynchronous device-side assert

TRACE_FIELDS = ("step", "tbytes", "D", "VEC", "MODE", "PROP", "FAST", "SPEC", "MK", "MULTI")


def debug_draw_normals(seed: int, steps: int, n: int, b: int, d: int, dtype, device, step0: int = 0) -> torch.Tensor:
    """``(steps, D, B, N)``: the standard normals a fused run with effective Philox seed ``seed`` (= the filter's seed
    + the plan's epoch) draws at local steps ``step0 ..`` (pf_debug_draw_normals)."""
    out = torch.empty((steps, d, b, n), dtype=dtype, device=device)
    L.require_gpu(out)
    L.check(L.load().pf_debug_draw_normals(seed & 0xFFFFFFFFFFFFFFFF, step0, steps, out.data_ptr(), n, b, d,
                                           L.dtype_code(dtype), L.stream_ptr()), "pf_debug_draw_normals")
    return out


TRACE_LEN = 2048  # launch records the library keeps per thread (PF_TRACE_LEN)


def debug_launch_trace(last: int = 64):
    """The step-kernel instantiations of this thread's latest fused launches, oldest first, as dicts (pf_debug_launch_trace)."""
    buf = (C.c_int32 * (len(TRACE_FIELDS) * last))()
    n = L.load().pf_debug_launch_trace(buf, last)
    if n < 0:
        L.check(n, "pf_debug_launch_trace")
    k = len(TRACE_FIELDS)
    return [dict(zip(TRACE_FIELDS, buf[i * k:(i + 1) * k])) for i in range(n)]


def observed_flags(y: torch.Tensor) -> torch.Tensor:
    """``y (T, ...)`` -> ``(T,)`` uint8 on the device: 1 unless the whole observation is NaN (pf_observed_flags;
    ``filters/base.py:212``)."""
    L.require_gpu(y)
    y = y.contiguous()
    steps = y.shape[0]
    out = torch.empty(steps, dtype=torch.uint8, device=y.device)
    if steps:
        L.check(L.load().pf_observed_flags(y.data_ptr(), steps, y.numel() // steps, L.dtype_code(y.dtype), out.data_ptr(),
                                           L.stream_ptr()), "pf_observed_flags")
    return out


def theta_ess(log_w: torch.Tensor) -> torch.Tensor:
    """``(B,)`` theta log-weights -> ``(2,)``: their effective sample size and a "all finite" flag; ``(rows, B)`` ->
    ``(rows, 2)``.  One launch (pf_theta_ess; ``sequential/state.py:35-44``, ``smc2.py:59-62``)."""
    L.require_gpu(log_w)
    log_w = log_w.contiguous()
    b = log_w.shape[-1]
    rows = log_w.numel() // b
    out = torch.empty(log_w.shape[:-1] + (2,), dtype=log_w.dtype, device=log_w.device)
    L.check(L.load().pf_theta_ess(log_w.data_ptr(), rows, b, L.dtype_code(log_w.dtype), out.data_ptr(), L.stream_ptr()),
            "pf_theta_ess")
    return out


def theta_path(w0: torch.Tensor, ll: torch.Tensor, rows: Optional["HostRows"] = None, status: Optional[torch.Tensor] = None):
    """``w0 (B,)``, ``ll (n, B)`` -> ``(w0 + ll.cumsum(0) (n, B), (n, 2) ESS / all-finite rows)`` in one launch
    (pf_theta_path; ``sequential/state.py:35-44`` for the n observations of a block).  ``rows`` (a ``HostRows`` of at least
    ``n`` rows): every row also lands in host memory (``rows.wait(q)``), with ``status`` - the int32 status word of the run that
    produced ``ll`` - next to it."""
    L.require_gpu(w0, ll)
    n, b = ll.shape
    assert w0.shape == (b,) and w0.dtype == ll.dtype and w0.is_contiguous() and ll.is_contiguous()
    assert rows is None or rows.n >= n
    w_path = torch.empty_like(ll)
    stats = torch.empty((n, 2), dtype=ll.dtype, device=ll.device)
    rp, seq = (None, 0) if rows is None else (rows.ptr, rows.seq + 1)
    L.check(L.load().pf_theta_path(w0.data_ptr(), ll.data_ptr(), n, b, L.dtype_code(ll.dtype), w_path.data_ptr(), stats.data_ptr(),
                                   rp, seq, L.ptr(status) if rows is not None else None, L.stream_ptr()), "pf_theta_path")
    if rows is not None:
        rows.seq = seq
    return w_path, stats


def _free_host(ptr: int):
    try:
        L.load().pf_host_free(ptr)
    except Exception:  # (interpreter shutdown)
        pass


class HostSlot:
    """32 bytes of host memory the DEVICE writes and the host polls (``pf_host_alloc``: coherent, mapped): the (ESS, all finite)
    pair of ``theta_step`` followed by a sequence number and the status word of the move that was folded in.  ``wait()`` spins
    on the sequence number - the reference's host test of the ESS after every observation (``smc2.py:59-62``) without a
    device -> host copy command and its synchronisation."""

    SPINS = 1 << 22  # (~0.5 s of polling: then the stream is synchronised and the slot read once more)

    def __init__(self):
        import weakref

        p = C.c_void_p()
        L.check(L.load().pf_host_alloc(64, C.byref(p)), "pf_host_alloc")
        self.ptr = p.value
        self._vals = (C.c_double * 2).from_address(self.ptr)
        self._seq = C.c_uint64.from_address(self.ptr + 16)
        self._status = C.c_uint64.from_address(self.ptr + 24)
        self.seq = 0
        fin = weakref.finalize(self, _free_host, self.ptr)
        fin.atexit = False

    def __deepcopy__(self, memo):
        return HostSlot()

    def wait(self):
        """(ESS, all-finite flag) of the latest ``theta_step`` issued with this slot, as Python floats.  ``self.status`` holds
        the status word that step was given (non-zero: the move behind it gave up - nothing was updated)."""
        want, cell = self.seq, self._seq
        for _ in range(self.SPINS):
            if cell.value == want:
                break
        else:
            torch.cuda.current_stream().synchronize()
            if cell.value != want:
                raise L.PfAmdError("pf_theta_step: the device's write to the host slot never became visible")
        self.status = int(self._status.value)
        return self._vals[0], self._vals[1]


class HostRows:
    """``n`` 32-byte slots of host memory the device writes (``pf_theta_path``'s ``host_rows``): row q of a block's statistics -
    (ESS, all finite), the launch's sequence number, the status word of the run behind the block - written by row q's own workgroup.
    ``wait(q)`` spins on row q's sequence number: the host has a block's rows in order, each as soon as it is done, without a
    device -> host copy command or an event behind the block."""

    SPINS = HostSlot.SPINS

    def __init__(self, n: int):
        import weakref

        p = C.c_void_p()
        L.check(L.load().pf_host_alloc(32 * n, C.byref(p)), "pf_host_alloc")
        self.ptr, self.n, self.seq = p.value, n, 0
        self._d = (C.c_double * (4 * n)).from_address(self.ptr)
        self._u = (C.c_uint64 * (4 * n)).from_address(self.ptr)
        fin = weakref.finalize(self, _free_host, self.ptr)
        fin.atexit = False

    def __deepcopy__(self, memo):
        return HostRows(self.n)

    def wait(self, q: int):
        """(ESS, all-finite flag, status word) of row ``q`` of the latest ``theta_path`` issued with these rows."""
        want, u, at = self.seq, self._u, 4 * q + 2
        for _ in range(self.SPINS):
            if u[at] == want:
                break
        else:
            torch.cuda.current_stream().synchronize()
            if u[at] != want:
                raise L.PfAmdError("pf_theta_path: the device's write to the host rows never became visible")
        return self._d[4 * q], self._d[4 * q + 1], int(u[4 * q + 3])


def theta_step(w: torch.Tensor, ll: torch.Tensor, slot: Optional[HostSlot] = None, acc: Optional[torch.Tensor] = None,
               status: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``w (B,) += ll (B,)`` in place and its ``(2,)`` statistics (ESS, all finite) in one launch (pf_theta_step: one
    observation of ``sequential/state.py:35-44``); with a ``slot`` the pair also lands in host memory (``slot.wait()``).
    ``acc (B,)``: the filters' running log-likelihood, ``acc += ll`` in the same launch.  ``status``: the int32 status word of
    the move that produced ``ll`` (``pf_filter_args.status``) - non-zero on the device: nothing is updated, the slot reports it."""
    stats = torch.empty(2, dtype=w.dtype, device=w.device)
    if acc is not None:
        assert acc.dtype == w.dtype and acc.shape == w.shape and acc.is_contiguous() and acc.device == w.device
    sp, seq = (None, 0)
    if slot is not None:
        sp, seq = slot.ptr, slot.seq + 1
    L.check(L.load().pf_theta_step(w.data_ptr(), ll.data_ptr(), w.shape[0], L.dtype_code(w.dtype), stats.data_ptr(), sp, seq,
                                   L.ptr(acc), L.ptr(status), L.stream_ptr()), "pf_theta_step")
    if slot is not None:
        slot.seq = seq  # (committed only once the launch was accepted: a refused call leaves host and device counters in step)
    return stats


def theta_resample(log_w: torch.Tensor, u: float) -> torch.Tensor:
    """Systematic resampling of the theta-particles from their log-weights: ``(B,)`` int64 ancestors, one launch
    (pf_theta_resample; ``kernels/mh.py:52-56``)."""
    L.require_gpu(log_w)
    log_w = log_w.contiguous()
    b = log_w.shape[0]
    idx = torch.empty(b, dtype=torch.int64, device=log_w.device)
    scratch = torch.empty_like(log_w)
    L.check(L.load().pf_theta_resample(log_w.data_ptr(), b, float(u), L.dtype_code(log_w.dtype), idx.data_ptr(), scratch.data_ptr(),
                                       L.stream_ptr()), "pf_theta_resample")
    return idx


def theta_fit(values: torch.Tensor, log_w, scale: float = 1.0):
    """``values (B, P)``, ``log_w (B,)`` or ``None`` (equal weights) -> ``(mean (P,), scale * chol (P, P))`` of the weighted
    Gaussian fit (pf_theta_fit; ``inference/utils.py:42-76``).  One launch."""
    L.require_gpu(values)
    values = values.contiguous()
    b, p = values.shape
    out = torch.empty(p + p * p, dtype=values.dtype, device=values.device)
    if log_w is not None:
        log_w = log_w.contiguous()
        assert log_w.shape == (b,) and log_w.dtype == values.dtype and log_w.device == values.device
    L.check(L.load().pf_theta_fit(values.data_ptr(), None if log_w is None else log_w.data_ptr(), b, p, float(scale),
                                  L.dtype_code(values.dtype), out.data_ptr(), out[p:].data_ptr(), L.stream_ptr()), "pf_theta_fit")
    return out[:p], out[p:].view(p, p)


def theta_propose(priors, mean: torch.Tensor, chol: torch.Tensor, eps: torch.Tensor, x_out, prior_out: Optional[torch.Tensor] = None):
    """theta* = mean + chol eps -> ``(u* (B, P), log prior of u* (B,))``; the constrained values are written into the ``P``
    tensors ``x_out`` (pf_theta_propose; ``mcmc/utils.py:48-50``, ``prior.py:98-123``).  ``priors``: a ``PfThetaPriors``."""
    L.require_gpu(eps, mean, chol)
    eps = eps.contiguous()
    b, p = eps.shape
    assert p == priors.P == len(x_out) and mean.is_contiguous() and chol.is_contiguous() and mean.dtype == chol.dtype == eps.dtype
    for x in x_out:
        assert x.is_contiguous() and x.numel() == b and x.dtype == eps.dtype and x.device == eps.device
    ptrs = (C.c_void_p * p)(*[x.data_ptr() for x in x_out])
    u = torch.empty_like(eps)
    lp = prior_out if prior_out is not None else torch.empty(b, dtype=eps.dtype, device=eps.device)
    assert lp.shape == (b,) and lp.dtype == eps.dtype and lp.is_contiguous()
    L.check(L.load().pf_theta_propose(C.byref(priors), mean.data_ptr(), chol.data_ptr(), eps.data_ptr(), b, L.dtype_code(eps.dtype),
                                      u.data_ptr(), ptrs, lp.data_ptr(), L.stream_ptr()), "pf_theta_propose")
    return u, lp


def theta_accept(u_cur, u_star, fwd, rev, prior_cur, prior_star, ll_cur, ll_star, unif):
    """The acceptance step of one PMMH move (pf_theta_accept; ``mcmc/utils.py:57-70``): ``fwd`` / ``rev`` = ``(mean, chol)`` of
    the forward / reverse Gaussian kernels.  Returns ``(log_acc (B,), accepted (B,) bool, rate ())``."""
    L.require_gpu(u_cur, u_star, unif)
    b, p = u_cur.shape
    dt = u_cur.dtype
    args = [u_cur, u_star, fwd[0], fwd[1], rev[0], rev[1], prior_cur, prior_star, ll_cur, ll_star, unif]
    args = [a.contiguous() for a in args]
    assert all(a.dtype == dt and a.device == u_cur.device for a in args), "theta_accept: one dtype / device"
    assert all(a.numel() == b for a in args[6:]) and u_star.shape == (b, p)
    log_acc = torch.empty(b + 1, dtype=dt, device=u_cur.device)
    accepted = torch.empty(b, dtype=torch.bool, device=u_cur.device)
    L.check(L.load().pf_theta_accept(*[a.data_ptr() for a in args], b, p, L.dtype_code(dt), log_acc.data_ptr(), accepted.data_ptr(),
                                     log_acc[b:].data_ptr(), L.stream_ptr()), "pf_theta_accept")
    return log_acc[:b], accepted, log_acc[b]


# ----------------------------------------------------------------------------------------------------------------
# smoothing over a recorded state history
# ----------------------------------------------------------------------------------------------------------------
def smooth_fixed_lag(x_hist: torch.Tensor, anc_hist: torch.Tensor) -> torch.Tensor:
    """``x_hist (S, D, B, N)``, ``anc_hist (S, B, N)`` int32 -> the ancestral lines of the last state's particles
    ``(S, D, B, N)`` (pf_smooth_fixed_lag; ``_do_sample_fl``, particle/base.py:136-152)."""
    L.require_gpu(x_hist, anc_hist)
    s, d, b, n = x_hist.shape
    assert anc_hist.shape == (s, b, n) and anc_hist.dtype == torch.int32 and anc_hist.is_contiguous() and x_hist.is_contiguous()
    out = torch.empty_like(x_hist)
    L.check(L.load().pf_smooth_fixed_lag(x_hist.data_ptr(), anc_hist.data_ptr(), out.data_ptr(), s, n, b, d,
                                         L.dtype_code(x_hist.dtype), L.stream_ptr()), "pf_smooth_fixed_lag")
    return out


def smooth_ffbs(kind, params: torch.Tensor, x_hist: torch.Tensor, logw_hist: torch.Tensor, x_last: torch.Tensor,
                u: Optional[torch.Tensor], seed: int) -> torch.Tensor:
    """Backward simulation over ``x_hist (S, D, B, N)`` / ``logw_hist (S, B, N)`` starting from ``x_last (D, B, N)``
    (pf_smooth_ffbs; ``_do_sample_ffbs``, particle/base.py:105-134).  ``u (S - 1, B, N)`` or None (Philox)."""
    L.require_gpu(x_hist, logw_hist, x_last, u, params)
    s, d, b, n = x_hist.shape
    assert logw_hist.shape == (s, b, n) and x_last.shape == (d, b, n)
    assert x_hist.is_contiguous() and logw_hist.is_contiguous() and x_last.is_contiguous()
    if u is not None:
        u = u.to(x_hist.dtype).contiguous()
        assert u.shape == (s - 1, b, n)
    out = torch.empty_like(x_hist)
    m = make_model_struct(kind, params)
    L.check(L.load().pf_smooth_ffbs(C.byref(m), x_hist.data_ptr(), logw_hist.data_ptr(), x_last.data_ptr(), L.ptr(u),
                                    seed & 0xFFFFFFFFFFFFFFFF, out.data_ptr(), s, n, b, L.dtype_code(x_hist.dtype),
                                    L.stream_ptr()), "pf_smooth_ffbs")
    return out
