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


def lib():
    """Load (building in-tree if needed) libssie_hip.so; raises if that is impossible."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB
    if not _build.up_to_date():
        if _build.hipcc() is None and not os.path.exists(path):
            raise SsieError("libssie_hip.so is missing and hipcc is unavailable; run __graft_entry__.build()")
        if _build.hipcc() is not None:
            _build.build(verbose=False)
    L = C.CDLL(path)
    L.ssie_version.restype = C.c_char_p
    L.ssie_op_workspace_bytes.restype = C.c_size_t
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
