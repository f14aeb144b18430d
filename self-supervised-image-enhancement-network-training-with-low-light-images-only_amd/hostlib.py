"""ctypes binding of libssie_hip.so (the C-ABI declared in include/ssie_hip.h).

PyTorch is used only for device memory and streams; every tensor crosses the boundary as a raw
device pointer.  Fails loudly when the HIP library is missing or cannot be built: there is no
fallback path.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import build as _build

_LIB = None


class SrcT(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int), ("cstride", C.c_int), ("coff", C.c_int),
                ("Hs", C.c_int), ("Ws", C.c_int)]


ERRORS = {1: "SSIE_E_ARG", 2: "SSIE_E_SHAPE", 3: "SSIE_E_WORKSPACE", 4: "SSIE_E_LAUNCH"}


class SsieError(RuntimeError):
    pass


# development switches (include/ssie_debug.h), honoured ONLY when SSIE_DEBUG=1 (tools/ and A/B scripts set it): a product process
# with stray SSIE_* variables in its environment behaves exactly like one without (tests/test_host_cpu.py)
_DEBUG_ENV = (("SSIE_OVERLAP", "ssie_debug_set_overlap"), ("SSIE_GRAPH", "ssie_debug_set_graph"), ("SSIE_MIN_TILES16", "ssie_debug_set_fprop_min_tiles16"),
              ("SSIE_WGRAD_SLIDING", "ssie_debug_set_wgrad_sliding"), ("SSIE_V2_STRIDE2", "ssie_debug_set_fprop_v2_stride2"),
              ("SSIE_WIDE", "ssie_debug_set_fprop_wide"), ("SSIE_BF16_WS", "ssie_debug_set_bf16_ws"), ("SSIE_BF16_RESW", "ssie_debug_set_bf16_resw"),
              ("SSIE_BF16_WS_GEO", "ssie_debug_set_bf16_ws_geo"), ("SSIE_BF16_CONV9", "ssie_debug_set_bf16_conv9"),
              ("SSIE_ATTN_PREPASS", "ssie_debug_set_attn_bf16_prepass"), ("SSIE_FFT_CHUNK_MB", "ssie_debug_set_fft_chunk_mb"),
              ("SSIE_REDUCE_WIDE_MIN", "ssie_debug_set_wgrad_reduce_wide_min"), ("SSIE_FFT_GROUPED", "ssie_debug_set_fft_grouped"),
              ("SSIE_LOSS_CHUNKED", "ssie_debug_set_loss_chunked"), ("SSIE_LOSS_CHUNK_LPP", "ssie_debug_set_loss_chunk_lpp"),
              ("SSIE_LOSS_GENERIC", "ssie_debug_set_loss_generic"), ("SSIE_V2_SPLIT", "ssie_debug_set_fprop_v2_split"),
              ("SSIE_FOLD_MASKS", "ssie_debug_set_fold_masks"), ("SSIE_BATCHED_REDUCE", "ssie_debug_set_batched_reduce"), ("SSIE_BF16_TWO_WGS", "ssie_debug_set_bf16_two_wgs"), ("SSIE_TCONV_SPLIT_BELOW", "ssie_debug_set_tconv_split_below"), ("SSIE_WINO_HALF_BELOW", "ssie_debug_set_wino_half_below"), ("SSIE_V2_ONETAP", "ssie_debug_set_fprop_v2_onetap"), ("SSIE_TCONV_HALF_BELOW", "ssie_debug_set_tconv_half_tiles_below"), ("SSIE_QKV_FUSED", "ssie_debug_set_qkv_fused"),
              ("SSIE_WINO4_MIN_TILES", "ssie_debug_set_wino4_min_tiles"), ("SSIE_WINO_MIN_TILES", "ssie_debug_set_wino_min_tiles"),
              ("SSIE_WGRAD_WINO_MIN_TILES", "ssie_debug_set_wgrad_wino_min_tiles"), ("SSIE_TCONV_MIN_TILES", "ssie_debug_set_tconv_min_tiles"))


def debug_enabled() -> bool:
    return os.environ.get("SSIE_DEBUG") == "1"


def declared_symbols():
    """every entry point include/ssie_hip.h declares (the drop-in boundary)"""
    import re
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ssie_hip.h")
    return sorted(set(re.findall(r"\b(ssie_[a-z0-9_]+)\s*\(", open(hdr).read())))


def lib():
    """Load libssie_hip.so.  Load-only: building happens in `__graft_entry__.build()` / `python -m ssie_amd.build` BEFORE any
    process touches the GPU (a GPU-initialised process must not spawn compiler children, and N ranks must not race on one
    output file).  A missing or stale library is a loud error, never a fallback.  With SSIE_DEBUG=1 (development only) a second
    build can be selected with SSIE_HIP_LIB for A/B runs inside one GPU session and the SSIE_* switches above are applied; either
    way the library must export the whole boundary and identify itself as a gfx950 build."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB
    if debug_enabled() and os.environ.get("SSIE_HIP_LIB"):
        path = os.environ["SSIE_HIP_LIB"]
        if not os.path.exists(path):
            raise SsieError(f"SSIE_HIP_LIB={path} does not exist")
    elif not os.path.exists(path):
        raise SsieError(f"{path} is missing: build it first with `python -c 'import __graft_entry__ as g; g.build()'`")
    elif not _build.up_to_date():
        raise SsieError(f"{path} is older than its sources: rebuild with `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(path)
    missing = [n for n in declared_symbols() if not hasattr(L, n)]
    if missing:
        raise SsieError(f"{path} does not export {missing[:4]}{'...' if len(missing) > 4 else ''}: not a build of this source tree")
    L.ssie_version.restype = C.c_char_p
    if b"gfx950" not in (L.ssie_version() or b""):
        raise SsieError(f"{path}: ssie_version() = {L.ssie_version()!r}, expected a gfx950 build")
    L.ssie_op_workspace_bytes.restype = C.c_size_t
    if debug_enabled():
        for env, fn in _DEBUG_ENV:
            if os.environ.get(env) is not None:
                getattr(L, fn)(int(os.environ[env]))
    _LIB = L
    return L


def check(rc, what):
    if rc != 0:
        raise SsieError(f"{what} failed: {ERRORS.get(rc, rc)}")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def nhwc(t: torch.Tensor) -> torch.Tensor:
    """(N,C,H,W) logical tensor -> dense NHWC float32 device buffer with C padded to a multiple of 4 (zeros)."""
    n, c, h, w = t.shape
    cp = (c + 3) // 4 * 4
    out = torch.zeros(n, h, w, cp, device=t.device, dtype=torch.float32)
    out[..., :c] = t.permute(0, 2, 3, 1)
    return out


def src_of(buf: torch.Tensor, c: int | None = None, coff: int = 0) -> SrcT:
    """buf: (N,H,W,Cs) dense NHWC buffer."""
    n, h, w, cs = buf.shape
    return SrcT(buf.data_ptr(), cs if c is None else c, cs, coff, h, w)


def workspace(cin, cout, k, device):
    nbytes = lib().ssie_op_workspace_bytes(cin, cout, k)
    return torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=device)


# ---- granular operators (NHWC buffers in / out) -------------------------------------------------
def conv2d_fwd(srcs, hv, wv, weight, bias, k, stride=1, act=0, addsrc=None, want_out2=False):
    n = srcs[0][0].shape[0]
    arr = (SrcT * len(srcs))(*[src_of(b, c, o) for (b, c, o) in srcs])
    cin = sum(s.C for s in arr)
    cout = weight.shape[0]
    pad = (k - 1) // 2
    ho, wo = (hv + 2 * pad - k) // stride + 1, (wv + 2 * pad - k) // stride + 1
    cp = (cout + 3) // 4 * 4
    dev = weight.device
    out = torch.zeros(n, ho, wo, cp, device=dev)
    out2 = torch.zeros_like(out) if want_out2 else None
    ws = workspace(cin, cout, k, dev)
    rc = lib().ssie_conv2d_fwd(arr, len(srcs), n, hv, wv, ptr(weight), weight.shape[1], ptr(bias), cout, k, stride, act,
                               ptr(addsrc), ptr(out2), ptr(out), cp, 0, ptr(ws), C.c_size_t(ws.numel() * 4), stream_ptr())
    check(rc, "ssie_conv2d_fwd")
    return (out, out2) if want_out2 else out


def conv_transpose2d_fwd(x, weight, bias, act=0):
    n, h, w, cs = x.shape
    cout = weight.shape[1]
    cp = (cout + 3) // 4 * 4
    out = torch.zeros(n, 2 * h, 2 * w, cp, device=x.device)
    ws = workspace(weight.shape[0], cout, 3, x.device)
    s = src_of(x, weight.shape[0])
    rc = lib().ssie_conv_transpose2d_fwd(C.byref(s), n, ptr(weight), ptr(bias), cout, act, ptr(out), cp, 0,
                                         ptr(ws), C.c_size_t(ws.numel() * 4), stream_ptr())
    check(rc, "ssie_conv_transpose2d_fwd")
    return out


def conv2d_dgrad(g, cout, weight, ci_off, cs, k, stride, hin, win, mask_y=None, mask_mode=0, gx=None):
    n, ho, wo, gcs = g.shape
    cin_total = weight.shape[1]
    cp = (cs + 3) // 4 * 4
    acc = gx is not None
    if gx is None:
        gx = torch.zeros(n, hin, win, cp, device=g.device)
    ws = workspace(cin_total, cout, k, g.device)
    rc = lib().ssie_conv2d_dgrad(ptr(g), gcs, 0, n, ho, wo, cout, ptr(weight), cin_total, ci_off, cs, k, stride,
                                 ptr(gx), hin, win, gx.shape[3], 0, ptr(mask_y), mask_mode, int(acc),
                                 ptr(ws), C.c_size_t(ws.numel() * 4), stream_ptr())
    check(rc, "ssie_conv2d_dgrad")
    return gx


def conv_transpose2d_dgrad(g, weight, mask_y=None, mask_mode=0):
    n, h2, w2, gcs = g.shape
    cin, cout = weight.shape[0], weight.shape[1]
    gx = torch.zeros(n, h2 // 2, w2 // 2, cin, device=g.device)
    ws = workspace(cin, cout, 3, g.device)
    rc = lib().ssie_conv_transpose2d_dgrad(ptr(g), gcs, 0, n, h2 // 2, w2 // 2, cout, ptr(weight), cin,
                                           ptr(gx), cin, 0, ptr(mask_y), mask_mode, 0,
                                           ptr(ws), C.c_size_t(ws.numel() * 4), stream_ptr())
    check(rc, "ssie_conv_transpose2d_dgrad")
    return gx


def conv2d_wgrad(src, hv, wv, g, cout, k, stride, cin_total, ci_off, dw=None, db=None):
    buf, c, o = src
    n = buf.shape[0]
    acc = dw is not None
    if dw is None:
        dw = torch.zeros(cout, cin_total, k, k, device=g.device)
        db = torch.zeros(cout, device=g.device)
    ws = workspace(max(cin_total, 64), max(cout, 64), k, g.device)
    s = src_of(buf, c, o)
    rc = lib().ssie_conv2d_wgrad(C.byref(s), n, hv, wv, ptr(g), g.shape[3], 0, cout, k, stride, cin_total, ci_off,
                                 ptr(dw), ptr(db), int(acc), ptr(ws), C.c_size_t(ws.numel() * 4), stream_ptr())
    check(rc, "ssie_conv2d_wgrad")
    return dw, db


def conv_transpose2d_wgrad(x, g, cin, cout):
    n = x.shape[0]
    dw = torch.zeros(cin, cout, 3, 3, device=g.device)
    db = torch.zeros(cout, device=g.device)
    ws = workspace(max(cin, 64), max(cout, 64), 3, g.device)
    s = src_of(x, cin)
    rc = lib().ssie_conv_transpose2d_wgrad(C.byref(s), n, ptr(g), g.shape[3], 0, cout, ptr(dw), ptr(db), 0,
                                           ptr(ws), C.c_size_t(ws.numel() * 4), stream_ptr())
    check(rc, "ssie_conv_transpose2d_wgrad")
    return dw, db


# ---- plan executor (whole hot path) ---------------------------------------------------------------
COEF_ORDER = ("c_rec", "c_rf", "c_il", "c_id", "c_f", "c_sp", "alpha_low", "alpha_delta")
LOSS_KEYS = ("total_loss", "L_reconstruction", "L_R_fidelity", "L_I_smooth_low",
             "L_I_smooth_delta", "L_fourier", "L_spectral_cons")


def _proto():
    L = lib()
    L.ssie_plan_create.restype = C.c_void_p
    L.ssie_plan_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.ssie_plan_destroy.argtypes = [C.c_void_p]
    L.ssie_plan_workspace_bytes.restype = C.c_size_t
    L.ssie_plan_workspace_bytes.argtypes = [C.c_void_p]
    L.ssie_plan_param_floats.restype = C.c_size_t
    L.ssie_plan_param_floats.argtypes = [C.c_void_p]
    L.ssie_plan_num_params.argtypes = [C.c_void_p]
    L.ssie_plan_param_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ssie_plan_buffer.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    L.ssie_plan_set_coefs.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.ssie_plan_set_graph.argtypes = [C.c_void_p, C.c_int]
    L.ssie_plan_bind.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ssie_plan_enhance_fwd.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_long), C.c_void_p]
    L.ssie_plan_enhance_fwd_bf16.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_long), C.c_void_p]
    L.ssie_plan_loss_fwd_bwd.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_long), C.c_int, C.c_void_p]
    L.ssie_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float,
                                 C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.ssie_fourier_mask.argtypes = [C.c_int, C.c_int, C.c_float, C.c_void_p]
    L.ssie_plan_backward_from_cotangents.argtypes = [C.c_void_p, C.c_void_p]
    L.ssie_selfsup_loss_workspace_bytes.restype = C.c_size_t
    L.ssie_selfsup_loss_workspace_bytes.argtypes = [C.c_int] * 4
    L.ssie_selfsup_loss_fwd_bwd.argtypes = ([C.c_void_p, C.c_int] * 5 + [C.c_int] * 4 + [C.POINTER(C.c_float), C.c_void_p]
                                            + [C.c_void_p] * 5 + [C.c_void_p, C.c_size_t, C.c_void_p])
    L.ssie_plan_profile_step.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_long), C.c_void_p,
                                         C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    return L


def param_table(bands: int):
    """[(state-dict key, float offset in the flat buffer, shape)], total floats — from the C side (single source of truth)."""
    L = _proto()
    coefs = (C.c_float * 8)(*([0.0] * 8))
    h = L.ssie_plan_create(1, bands, 16, 16, coefs)
    if not h:
        raise SsieError("ssie_plan_create failed")
    try:
        out = []
        name = C.create_string_buffer(128)
        for i in range(L.ssie_plan_num_params(h)):
            off = C.c_size_t(); nd = C.c_int(); shp = (C.c_int * 4)()
            check(L.ssie_plan_param_info(h, i, name, 128, C.byref(off), C.byref(nd), shp), "ssie_plan_param_info")
            out.append((name.value.decode(), off.value, tuple(shp[:nd.value])))
        return out, L.ssie_plan_param_floats(h)
    finally:
        L.ssie_plan_destroy(h)


def fourier_mask(h: int, w: int, cutoff: float = 0.1):
    import numpy as np
    m = np.zeros((h, w), dtype=np.uint8)
    check(_proto().ssie_fourier_mask(h, w, cutoff, m.ctypes.data_as(C.c_void_p)), "ssie_fourier_mask")
    return m


class Plan:
    """One (N, bands, H, W) instance of the hot path bound to flat parameter / gradient buffers."""

    def __init__(self, n, bands, h, w, coefs: dict, flat_params: torch.Tensor, flat_grads: torch.Tensor | None):
        self.L = _proto()
        self.shape = (n, bands, h, w)
        cf = (C.c_float * 8)(*[float(coefs[k]) for k in COEF_ORDER])
        self.h = self.L.ssie_plan_create(n, bands, h, w, cf)
        if not self.h:
            raise SsieError(f"unsupported plan shape N={n} B={bands} H={h} W={w} (H, W must be even and >= 8)")
        dev = flat_params.device
        nbytes = self.L.ssie_plan_workspace_bytes(self.h)
        self.ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
        self.flat_params, self.flat_grads = flat_params, flat_grads
        check(self.L.ssie_plan_bind(self.h, self.ws.data_ptr(), nbytes, flat_params.data_ptr(),
                                    0 if flat_grads is None else flat_grads.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream), "ssie_plan_bind")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.ssie_plan_destroy(self.h); self.h = None
        except Exception:
            pass

    def set_graph(self, on: bool):
        """replay the train step behind the input conversion as one hipGraph (ssie_plan_set_graph)"""
        check(self.L.ssie_plan_set_graph(self.h, int(bool(on))), "ssie_plan_set_graph")

    def set_coefs(self, coefs: dict):
        cf = (C.c_float * 8)(*[float(coefs[k]) for k in COEF_ORDER])
        check(self.L.ssie_plan_set_coefs(self.h, cf), "ssie_plan_set_coefs")

    def buffer(self, name: str) -> torch.Tensor:
        """NHWC view (N,H,W,C) of a named workspace buffer (padding channels dropped)."""
        off = C.c_size_t(); d = (C.c_int * 5)()
        check(self.L.ssie_plan_buffer(self.h, name.encode(), C.byref(off), d), f"ssie_plan_buffer({name})")
        n, h, w, c, cs = d[0], d[1], d[2], d[3], d[4]
        return torch.as_strided(self.ws, (n, h, w, c), (h * w * cs, w * cs, cs, 1), off.value)

    def nchw(self, name: str, c0: int = 0, c1: int | None = None) -> torch.Tensor:
        """logical (N,C,H,W) view of channels [c0, c1) — the reference's tensor convention."""
        b = self.buffer(name)
        return b[..., c0:c1].permute(0, 3, 1, 2)

    def _strides(self, x):
        n, b, h, w = self.shape
        if tuple(x.shape) != (n, b, h, w) or x.dtype != torch.float32 or not x.is_cuda:
            raise SsieError(f"expected float32 cuda tensor of shape {(n, b, h, w)}, got {tuple(x.shape)} {x.dtype} {x.device}")
        return (C.c_long * 4)(*x.stride())

    def enhance_fwd(self, x, bf16=False):
        """model.py:229-234 into the plan buffers; bf16=True: bf16 storage + bf16 MFMA (fp32 accumulate), fp32 outputs"""
        if bf16:
            check(self.L.ssie_plan_enhance_fwd_bf16(self.h, x.data_ptr(), self._strides(x), torch.cuda.current_stream().cuda_stream),
                  "ssie_plan_enhance_fwd_bf16")
            return
        check(self.L.ssie_plan_enhance_fwd(self.h, x.data_ptr(), self._strides(x), torch.cuda.current_stream().cuda_stream),
              "ssie_plan_enhance_fwd")

    def has_bf16(self) -> bool:
        """the bf16 enhance-only list exists for every band count since round 4 (the bf16 input cube and the bf16 twin of the
        R|I output carry their own pixel strides, padded to 8 channels); kept so callers need not know that"""
        return True

    def loss_fwd_bwd(self, x, backward=True):
        check(self.L.ssie_plan_loss_fwd_bwd(self.h, x.data_ptr(), self._strides(x), int(backward),
                                            torch.cuda.current_stream().cuda_stream), "ssie_plan_loss_fwd_bwd")

    def loss_scalars(self) -> torch.Tensor:
        return self.buffer("scalars").reshape(7)

    def backward_from_cotangents(self):
        """TEST ENTRY (include/ssie_debug.h): backward schedule only, on cotangents written into gRL / gD / gS / G8_2"""
        check(self.L.ssie_plan_backward_from_cotangents(self.h, torch.cuda.current_stream().cuda_stream),
              "ssie_plan_backward_from_cotangents")

    KINDS = ("conv_fprop(+dgrad) 64-ch tile", "conv_fprop(+dgrad) 32-ch tile", "conv_wgrad_kernel", "wgrad_reduce_kernel", "colsum",
             "pack_weights", "loss_direct", "fft_loss_kernel", "attention", "elementwise", "spectral 9x9 conv (fwd + dgrad + wgrad)",
             "winograd 3x3 conv (fwd + dgrad)", "winograd 3x3 weight gradient", "winograd F(4x4,3x3) conv (fwd + dgrad)")

    def profile_step(self, x):
        """{kernel class: (device ms, algorithmic FLOPs, launches)} of one loss+backward step (HIP events)."""
        nk = len(self.KINDS)
        ms = (C.c_double * nk)(); fl = (C.c_double * nk)(); cnt = (C.c_int * nk)()
        check(self.L.ssie_plan_profile_step(self.h, x.data_ptr(), self._strides(x), torch.cuda.current_stream().cuda_stream,
                                            ms, fl, cnt), "ssie_plan_profile_step")
        return {k: (ms[i], fl[i], cnt[i]) for i, k in enumerate(self.KINDS)}


def selfsup_loss_fwd_bwd(x, R, I, D, S, E, coefs: dict):
    """Standalone loss operator (ssie_selfsup_loss_fwd_bwd) on logical (N,C,H,W) cuda tensors.
    -> (scalars7 device tensor, dict(gR, gI, gD, gS, gE) as logical (N,C,H,W) tensors)"""
    L = _proto()
    n, b, h, w = x.shape
    dev = x.device
    xb, Sb, Db = nhwc(x), nhwc(S), nhwc(D)
    RLb = nhwc(torch.cat([R, I], 1))
    Eb = nhwc(torch.cat([E, torch.zeros_like(I)], 1))
    gRL, gD, gS, gE = torch.zeros_like(RLb), torch.zeros_like(Db), torch.zeros_like(Sb), torch.zeros_like(Eb)
    scal = torch.zeros(8, device=dev)
    mask = torch.from_numpy(fourier_mask(h, w)).to(dev)
    nbytes = L.ssie_selfsup_loss_workspace_bytes(n, b, h, w)
    ws = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=dev)
    cf = (C.c_float * 8)(*[float(coefs[k]) for k in COEF_ORDER])
    check(L.ssie_selfsup_loss_fwd_bwd(xb.data_ptr(), xb.shape[3], RLb.data_ptr(), RLb.shape[3], Db.data_ptr(), Db.shape[3],
                                      Sb.data_ptr(), Sb.shape[3], Eb.data_ptr(), Eb.shape[3], n, b, h, w, cf, mask.data_ptr(),
                                      gRL.data_ptr(), gD.data_ptr(), gS.data_ptr(), gE.data_ptr(), scal.data_ptr(),
                                      ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream), "ssie_selfsup_loss_fwd_bwd")
    nc = lambda t, c0, c1: t[..., c0:c1].permute(0, 3, 1, 2)
    return scal[:7], dict(gR=nc(gRL, 0, b), gI=nc(gRL, b, b + 1), gD=nc(gD, 0, 1), gS=nc(gS, 0, b), gE=nc(gE, 0, b))


def adam_step(params, grads, m, v, step, lr, grad_scale=1.0, b1=0.9, b2=0.999, eps=1e-8):
    check(_proto().ssie_adam_step(params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(), params.numel(),
                                  grad_scale, lr, step, b1, b2, eps, torch.cuda.current_stream().cuda_stream), "ssie_adam_step")


def attention_fwd(qkv: torch.Tensor):
    """qkv: (N, T, 192) -> out (N, T, 64), lse (N, 4, T)"""
    n, t, _ = qkv.shape
    out = torch.empty(n, t, 64, device=qkv.device); lse = torch.empty(n, 4, t, device=qkv.device)
    check(lib().ssie_attention_fwd(ptr(qkv), ptr(out), ptr(lse), n, t, stream_ptr()), "ssie_attention_fwd")
    return out, lse


def attention_fwd_bf16(qkv: torch.Tensor, prepass: bool = True):
    """TEST ENTRY (include/ssie_debug.h): the enhance-only path's bf16 attention.  qkv fp32 (N, T, 192) -> out bf16 (N, T, 64)"""
    n, t, _ = qkv.shape
    L = lib()
    L.ssie_debug_attention_bf16_scratch_bytes.restype = C.c_size_t
    out = torch.empty(n, t, 64, device=qkv.device, dtype=torch.bfloat16)
    nb = L.ssie_debug_attention_bf16_scratch_bytes(n, t) if prepass else 0
    scratch = torch.empty(max(nb, 16), dtype=torch.uint8, device=qkv.device)
    check(L.ssie_debug_attention_fwd_bf16(ptr(qkv), ptr(out), n, t, ptr(scratch) if prepass else None, C.c_size_t(nb), stream_ptr()),
          "ssie_debug_attention_fwd_bf16")
    return out


def attention_bwd(qkv, out, gout, lse):
    n, t, _ = qkv.shape
    gqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
    check(lib().ssie_attention_bwd(ptr(qkv), ptr(out), ptr(gout), ptr(lse), ptr(delta), ptr(gqkv), n, t, stream_ptr()),
          "ssie_attention_bwd")
    return gqkv


class CropT(C.Structure):
    _fields_ = [("cube", C.c_void_p), ("H", C.c_int), ("W", C.c_int), ("x0", C.c_int), ("y0", C.c_int), ("mode", C.c_int)]


def assemble_batch(cubes, crops, patch: int, bands: int, staging: torch.Tensor | None = None) -> torch.Tensor:
    """cubes: list of (H,W,C) fp32 cuda tensors; crops: [(cube index, x0, y0, mode)] -> logical (n, C, P, P) channels_last batch.
    staging: a PINNED uint8 host tensor of >= n * sizeof(CropT) bytes the caller keeps alive (and does not overwrite) until the
    copy has run: the crop records then go up with a non-blocking H2D on the current stream instead of a pageable copy, which
    would synchronise the host with everything queued on the stream (harness.train_model passes a two-slot ring)."""
    n = len(crops)
    dev = cubes[0].device
    recs = (CropT * n)()
    for i, (ci, x0, y0, mode) in enumerate(crops):
        cb = cubes[ci]
        assert cb.is_contiguous() and cb.dtype == torch.float32 and cb.shape[2] == bands
        assert 0 <= x0 <= cb.shape[0] - patch and 0 <= y0 <= cb.shape[1] - patch and 0 <= mode < 8
        recs[i] = CropT(cb.data_ptr(), cb.shape[0], cb.shape[1], x0, y0, mode)
    nb = C.sizeof(recs)
    if staging is not None:
        if not (staging.is_pinned() and staging.dtype == torch.uint8 and staging.numel() >= nb):
            raise SsieError("assemble_batch: staging must be a pinned uint8 tensor large enough for the crop records")
        C.memmove(staging.data_ptr(), C.addressof(recs), nb)
        raw = staging[:nb].to(dev, non_blocking=True)
    else:
        raw = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8).to(dev)
    cs = (bands + 3) // 4 * 4
    out = torch.empty(n, patch, patch, cs, device=dev)
    check(lib().ssie_assemble_batch(ptr(raw), n, ptr(out), patch, bands, cs, stream_ptr()), "ssie_assemble_batch")
    return out[..., :bands].permute(0, 3, 1, 2)
