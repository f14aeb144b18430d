"""Hand-derived cotangents of the six self-supervised losses (TEST INFRASTRUCTURE ONLY).

Closed-form restatement (no autograd) of d(total_loss)/d{R_low, I_low, I_delta, S, R_enh}
for the *direct* loss dependencies of /root/reference/model.py:544-564, i.e. with
(R, I, D, S, E) treated as independent leaves.  This is the blueprint the fused HIP loss
kernels implement; `tests/test_oracle_golden.py` proves it equal to autograd on the
restated losses in `ssie_oracle.loss_terms`.  SURVEY.md §2.2 items 1-6.

Conventions: sg = sign with sg(0)=0 (torch.abs subgradient); dx/dy are forward differences
along W/H (model.py:483-489); adjoint of a forward difference scatters +g to j+1 and -g to j.
"""
from __future__ import annotations

import torch

from .ssie_oracle import fourier_mask


def _dx(t):
    return t[..., :, 1:] - t[..., :, :-1]


def _dy(t):
    return t[..., 1:, :] - t[..., :-1, :]


def _dxT(g):
    out = torch.zeros(g.shape[:-1] + (g.shape[-1] + 1,), dtype=g.dtype)
    out[..., 1:] += g
    out[..., :-1] -= g
    return out


def _dyT(g):
    out = torch.zeros(g.shape[:-2] + (g.shape[-2] + 1, g.shape[-1]), dtype=g.dtype)
    out[..., 1:, :] += g
    out[..., :-1, :] -= g
    return out


def direct_cotangents(x, R, I, D, S, E, coefs):
    """-> dict(gR, gI, gD, gS, gE) of the weighted total loss w.r.t. the five leaves."""
    N, C, H, W = R.shape
    sg = torch.sign
    a1, a2 = coefs["alpha_low"], coefs["alpha_delta"]
    n0 = N * C * H * W
    gR = torch.zeros_like(R); gI = torch.zeros_like(I); gD = torch.zeros_like(D)
    gS = torch.zeros_like(S)

    # 1. reconstruction  (model.py:551)
    s = sg(R * I - x) / n0
    gR += coefs["c_rec"] * s * I
    gI += coefs["c_rec"] * (s * R).sum(1, keepdim=True)

    # 2. I_low edge-aware smoothness, channel-mean weights (model.py:500-515)
    for d, dT, nI in ((_dx, _dxT, N * H * (W - 1)), (_dy, _dyT, N * (H - 1) * W)):
        dR = d(R)
        w = torch.exp(-a1 * dR.abs().mean(1, keepdim=True))
        u = d(I)
        gI += coefs["c_il"] * dT(w * sg(u) / nI)
        gR += coefs["c_il"] * dT(-a1 * w * u.abs() * sg(dR) / (C * nI))

    # 3. R fidelity (model.py:521-534), beta = 0.5
    delta = R - E
    g_delta = sg(delta) / n0 \
        + 0.5 * _dxT(sg(_dx(delta)) / (N * C * H * (W - 1))) \
        + 0.5 * _dyT(sg(_dy(delta)) / (N * C * (H - 1) * W))
    gR += coefs["c_rf"] * g_delta
    gE = -coefs["c_rf"] * g_delta

    # 4. I_delta smoothness, per-channel weights, 1-ch D broadcast over C (model.py:450-454)
    for d, dT, nR in ((_dx, _dxT, N * C * H * (W - 1)), (_dy, _dyT, N * C * (H - 1) * W)):
        dR = d(R)
        e = torch.exp(-a2 * dR.abs())
        u = d(D)
        gD += coefs["c_id"] * dT(sg(u) * e.sum(1, keepdim=True) / nR)
        gR += coefs["c_id"] * dT(-a2 * u.abs() * e * sg(dR) / nR)

    # 5. Fourier magnitude (model.py:456-473); unnormalised fft2 => adjoint = H*W*ifft2
    m = fourier_mask(H, W, dtype=x.dtype)[None, None]
    Zx = torch.fft.fft2(x) * m
    Z = torch.fft.fft2(S) * m
    A = Z.abs()
    gA = -sg(Zx.abs() - A) / n0
    gZ = torch.where(A > 0, gA * Z / torch.where(A > 0, A, torch.ones_like(A)), torch.zeros_like(Z))
    gS += coefs["c_f"] * (H * W * torch.fft.ifft2(m * gZ)).real

    # 6. spectral TV (model.py:475-481)
    if C > 1:
        t = S[:, 1:] - S[:, :-1]
        g = sg(t) / (N * (C - 1) * H * W)
        gS[:, 1:] += coefs["c_sp"] * g
        gS[:, :-1] -= coefs["c_sp"] * g
    return dict(gR=gR, gI=gI, gD=gD, gS=gS, gE=gE)


def close_product_node(gS, R, I, D):
    """S = R*(D+I) (model.py:233): -> (gR_add, gD_add, gI_add)."""
    q = (gS * R).sum(1, keepdim=True)
    return gS * (D + I), q, q
