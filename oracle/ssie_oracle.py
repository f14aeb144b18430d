"""CPU oracle for the SS-HSLIE hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain-PyTorch, CPU restatement of the reference's L2 layer
(model + six self-supervised losses + Adam step).  It exists only so that
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg can
check / time the HIP path against the reference's arithmetic on the GPU box,
where `/root/reference` does not exist.  Nothing under the product package may
import it.

Parity pin: `tests/golden/make_golden.py` imports the real reference
(`/root/reference/model.py`) in the build container, drives it with the
closed-form parameters/inputs defined here, and commits the reference outputs
as `tests/golden/*.npz`; `tests/test_oracle_golden.py` asserts this oracle
reproduces them (and, when `/root/reference` is present, compares live).

Reference lines restated (all in /root/reference/model.py):
  conv helper ............................ :17-23
  DecompositionNet.forward ............... :49-70
  TransformerBlock.forward ............... :99-119
  IllumAdjustmentNet.forward ............. :143-175
  LowLightEnhance.forward ................ :229-234
  compute_gradients / smooth_loss ........ :445-454
  fourier_spectrum_loss .................. :456-473
  spectral_smoothness_loss ............... :475-481
  structure_aware_loss ................... :491-542
  compute_loss ........................... :544-575
  Adam (torch.optim.Adam defaults) ....... :213, :316
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

CHANNEL = 64          # model.py:26,122 default `channel`
HEADS, HEAD_DIM = 4, 16   # model.py:88

LOSS_KEYS = ("total_loss", "L_reconstruction", "L_R_fidelity", "L_I_smooth_low",
             "L_I_smooth_delta", "L_fourier", "L_spectral_cons")

DEFAULT_COEFS = dict(c_rec=10.0, c_rf=1.0, c_il=1.0, c_id=20.0, c_f=0.2, c_sp=1.0,
                     alpha_low=1.0, alpha_delta=10.0)          # main.py:41-48
JYU_COEFS = dict(c_rec=10.0, c_rf=1.0, c_il=1.0, c_id=2000.0, c_f=20.0, c_sp=1.0,
                 alpha_low=1.0, alpha_delta=10.0)              # config_outdoor_jyu.yml:24-31


# --------------------------------------------------------------------------
# parameter table: state-dict key -> shape   (model.py:33-47,93-97,125-141)
# --------------------------------------------------------------------------
def param_shapes(bands: int, channel: int = CHANNEL) -> "OrderedDict[str, tuple]":
    c = channel
    d = "decomposition_net."
    i = "illum_adjust_net."
    t = HEADS * HEAD_DIM
    spec = OrderedDict()

    def cw(name, co, ci, k):
        spec[name + ".weight"] = (co, ci, k, k)
        spec[name + ".bias"] = (co,)

    cw(d + "conv0.0", c // 2, bands, 3)
    cw(d + "shallow_conv.0", c, bands, 9)
    cw(d + "conv1.0", c, c, 3)
    cw(d + "conv2.0", 2 * c, c, 3)
    cw(d + "conv3.0", 2 * c, 2 * c, 3)
    spec[d + "deconv.0.weight"] = (2 * c, c, 3, 3)      # ConvTranspose2d: (in, out, k, k)
    spec[d + "deconv.0.bias"] = (c,)
    cw(d + "conv5.0", c, 2 * c, 3)
    cw(d + "conv7.0", c, c + c // 2, 3)
    cw(d + "recon", bands + 1, c, 3)
    cw(i + "conv0.0", c, bands + 1, 3)
    cw(i + "conv1.0", c, c, 3)
    cw(i + "conv2.0", c, c, 3)
    cw(i + "conv3.0", c, c, 3)
    for nm, (o, n) in (("q_linear", (t, c)), ("k_linear", (t, c)), ("v_linear", (t, c)),
                       ("ff_linear1", (64, t)), ("ff_linear2", (c, 64))):
        spec[i + "attn." + nm + ".weight"] = (o, n)
        spec[i + "attn." + nm + ".bias"] = (o,)
    cw(i + "deconv1.0", c, c, 3)
    cw(i + "deconv2.0", c, c, 3)
    cw(i + "deconv3.0", c, c, 3)
    cw(i + "feature_fusion.0", c, 3 * c, 1)
    cw(i + "final_conv", 1, c, 3)
    return spec


def _hash_uniform(n: int, stream: int) -> np.ndarray:
    """n pseudo-random float64 in [0,1): splitmix64-style integer hash of (index, stream) — pure integer
    arithmetic, so it is bit-identical on every machine (no libm involved)."""
    with np.errstate(over="ignore"):
        h = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
             + np.uint64(stream + 1) * np.uint64(0xBF58476D1CE4E5B9))
        h ^= h >> np.uint64(30); h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(27); h *= np.uint64(0x94D049BB133111EB)
        h ^= h >> np.uint64(31)
    return (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def closed_form_params(bands: int, channel: int = CHANNEL, dtype=torch.float32, gain: float = 1.0):
    """Deterministic, RNG-library-free parameter fill (full-rank, unlike a sinusoid):
    p.flat[i] = a_t * (2 u(i, t) - 1), u = integer hash, a_t = gain / sqrt(fan_in)  (PyTorch's default bound).
    Regenerated bit-identically wherever numpy runs, so fixtures carry reference OUTPUTS only."""
    out = OrderedDict()
    for t, (name, shape) in enumerate(param_shapes(bands, channel).items()):
        n = int(np.prod(shape))
        wname = name[:-5] + ".weight" if name.endswith(".bias") else name
        wshape = param_shapes(bands, channel)[wname]
        fan_in = int(np.prod(wshape[1:]))
        a = gain / math.sqrt(fan_in)
        vals = a * (2.0 * _hash_uniform(n, t) - 1.0)
        out[name] = torch.from_numpy(vals.reshape(shape)).to(dtype)
    return out


def synthetic_patches(n: int, bands: int, h: int, w: int, seed: int = 41, dtype=torch.float32):
    """Closed-form low-light cubes in [0, 0.3] (smooth illumination x band-correlated
    reflectance + hash noise), logical NCHW / channels_last memory like model.py:301,312."""
    nn_, cc, hh, ww = np.meshgrid(np.arange(n), np.arange(bands), np.arange(h), np.arange(w), indexing="ij")
    nn_ = nn_.astype(np.float64); cc = cc.astype(np.float64); hh = hh.astype(np.float64); ww = ww.astype(np.float64)
    illum = 0.55 + 0.35 * np.sin(0.11 * hh + 0.3 * nn_ + 0.01 * seed) * np.cos(0.07 * ww - 0.2 * nn_)
    refl = 0.5 + 0.3 * np.sin(0.45 * cc + 0.05 * hh - 0.04 * ww + 0.7 * nn_) \
               + 0.15 * np.cos(0.9 * cc - 0.13 * ww + 0.21 * hh)
    noise = _hash_uniform(n * bands * h * w, 1000 + seed).reshape(n, bands, h, w) - 0.5
    x = np.clip(illum * refl + 0.01 * noise, 0.0, 1.0) * 0.3
    t = torch.from_numpy(x).to(dtype)
    return t.contiguous(memory_format=torch.channels_last)


# --------------------------------------------------------------------------
# model forward
# --------------------------------------------------------------------------
def _relu(y, masks=None, name=None):
    """nn.ReLU (model.py:21).  `masks` (tests only): {buffer name: bool tensor}; where given, the 0/1 decision comes from the
    mask instead of sign(y).  tests/test_backward_gpu.py passes the HIP path's own decisions so that an fp64 evaluation of the
    backward chain takes exactly the same branches as the fp32 one (they can differ where |y| ~ 1e-7)."""
    if masks is not None and name in masks:
        return y * masks[name].to(y.dtype)
    return F.relu(y)


def _conv(x, P, key, stride=1, relu=False, masks=None, name=None):
    w = P[key + ".weight"]
    y = F.conv2d(x, w, P[key + ".bias"], stride=stride, padding=(w.shape[-1] - 1) // 2)   # model.py:18-20
    return _relu(y, masks, name) if relu else y


def _rec(tr, tag, **kw):
    """record named intermediates (buffer names of the HIP plan) for stage-by-stage parity tests"""
    if tr is None:
        return
    for k, v in kw.items():
        if v.requires_grad:
            v.retain_grad()
        tr[k + tag] = v


def decomposition(P, x, pre="decomposition_net.", tr=None, tag="", masks=None):
    """model.py:49-70 -> (R, L)"""
    bands = x.shape[1]
    c0 = _conv(x, P, pre + "conv0.0", relu=True, masks=masks, name="c0" + tag)
    sh = _conv(x, P, pre + "shallow_conv.0")
    c1 = _conv(sh, P, pre + "conv1.0", relu=True, masks=masks, name="c1" + tag)
    c2 = _conv(c1, P, pre + "conv2.0", stride=2, relu=True, masks=masks, name="c2" + tag)
    c3 = _conv(c2, P, pre + "conv3.0", relu=True, masks=masks, name="c3" + tag)
    dc = _relu(F.conv_transpose2d(c3, P[pre + "deconv.0.weight"], P[pre + "deconv.0.bias"],
                                  stride=2, padding=1, output_padding=1), masks, "dc" + tag)   # model.py:39-43
    c5 = _conv(torch.cat([dc, c1], 1), P, pre + "conv5.0", relu=True, masks=masks, name="c5" + tag)
    c7 = _conv(torch.cat([c5, c0], 1), P, pre + "conv7.0")
    c8 = _conv(c7, P, pre + "recon")
    _rec(tr, tag, c0=c0, sh=sh, c1=c1, c2=c2, c3=c3, dc=dc, c5=c5, c7=c7, c8=c8)
    return torch.sigmoid(c8[:, :bands]), torch.sigmoid(c8[:, bands:])


def attention_block(P, x, pre="illum_adjust_net.attn.", tr=None, masks=None):
    """model.py:99-119; tokens = H*W, 4 heads x 16, no LayerNorm, residual on tokens."""
    n, c, h, w = x.shape
    s = h * w
    tok = x.reshape(n, c, s).permute(0, 2, 1)
    lin = lambda t, k: F.linear(t, P[pre + k + ".weight"], P[pre + k + ".bias"])
    q, k, v = (lin(tok, nm).reshape(n, s, HEADS, HEAD_DIM).permute(0, 2, 1, 3)
               for nm in ("q_linear", "k_linear", "v_linear"))
    att = torch.softmax(q @ k.transpose(-2, -1) / (HEAD_DIM ** 0.5), dim=-1)
    o = (att @ v).permute(0, 2, 1, 3).reshape(n, s, HEADS * HEAD_DIM)
    f1 = lin(o, "ff_linear1")
    f1 = _relu(f1, None if masks is None or "f1" not in masks else {"f1": masks["f1"].reshape(n, c, s).permute(0, 2, 1)}, "f1")
    ff = lin(f1, "ff_linear2")
    _rec(tr, "", ao=o.permute(0, 2, 1).reshape(n, c, h, w), f1=f1.permute(0, 2, 1).reshape(n, c, h, w))
    return (tok + ff).permute(0, 2, 1).reshape(n, c, h, w)


def _up(x, like):
    return F.interpolate(x, size=like.shape[2:], mode="nearest")      # model.py:156,160,164,168,169


def illum_adjust(P, I, R, pre="illum_adjust_net.", tr=None, masks=None):
    """model.py:143-175 -> I_delta (N,1,H,W); note cat order [R, I]."""
    c0 = _conv(torch.cat([R, I], 1), P, pre + "conv0.0")
    c1 = _conv(c0, P, pre + "conv1.0", stride=2, relu=True, masks=masks, name="a1")
    c2 = _conv(c1, P, pre + "conv2.0", stride=2, relu=True, masks=masks, name="a2")
    c3 = _conv(c2, P, pre + "conv3.0", stride=2, relu=True, masks=masks, name="a3")
    t3 = attention_block(P, c3, pre + "attn.", tr, masks)
    u1 = _conv(_up(t3, c2), P, pre + "deconv1.0", relu=True, masks=masks, name="u1")
    d1 = u1 + c2
    u2 = _conv(_up(d1, c1), P, pre + "deconv2.0", relu=True, masks=masks, name="u2")
    d2 = u2 + c1
    u3 = _conv(_up(d2, c0), P, pre + "deconv3.0", relu=True, masks=masks, name="u3")
    d3 = u3 + c0
    gather = torch.cat([_up(d1, d3), _up(d2, d3), d3], 1)
    f = _conv(gather, P, pre + "feature_fusion.0")
    _rec(tr, "", a0=c0, a1=c1, a2=c2, a3=c3, t3=t3, u1=u1, d1=d1, u2=u2, d2=d2, u3=u3, d3=d3, f=f)
    return _conv(f, P, pre + "final_conv")


def enhance_forward(P, x, tr=None, masks=None):
    """model.py:229-234 -> (R_low, I_low, I_delta, S)"""
    R, I = decomposition(P, x, tr=tr, tag="_1", masks=masks)
    D = illum_adjust(P, I, R, tr=tr, masks=masks)
    S = R * D + R * I
    _rec(tr, "", R=R, I=I, D=D, S=S)
    return R, I, D, S


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------
def fourier_mask(h, w, cutoff=0.1, dtype=torch.float32):
    """model.py:460-464: radial mask on linspace(-1,1) grids, NOT fftshifted."""
    y = torch.linspace(-1, 1, h, dtype=dtype)
    x = torch.linspace(-1, 1, w, dtype=dtype)
    Y, X = torch.meshgrid(y, x, indexing="ij")
    return (torch.sqrt(X ** 2 + Y ** 2) >= cutoff).to(dtype)


def _dx(t):
    return t[..., :, 1:] - t[..., :, :-1]       # model.py:483-485


def _dy(t):
    return t[..., 1:, :] - t[..., :-1, :]       # model.py:487-489


def loss_terms(x, R, I, D, S, E, coefs):
    """The six terms of model.py:551-555 as 0-d tensors (order = LOSS_KEYS[1:])."""
    a1, a2 = coefs["alpha_low"], coefs["alpha_delta"]
    l_rec = torch.mean(torch.abs(R * I - x))                                             # :551
    # structure_aware_loss(R, I, E, alpha=a1, beta=0.5)                                  # :491-542
    wx = torch.exp(-a1 * _dx(R).abs().mean(dim=1, keepdim=True))
    wy = torch.exp(-a1 * _dy(R).abs().mean(dim=1, keepdim=True))
    l_il = torch.mean(wx * _dx(I).abs()) + torch.mean(wy * _dy(I).abs())
    l_rf = torch.mean(torch.abs(R - E)) + 0.5 * (torch.mean(torch.abs(_dx(R) - _dx(E)))
                                                 + torch.mean(torch.abs(_dy(R) - _dy(E))))
    # smooth_loss(I_delta, R, alpha=a2)                                                  # :450-454
    l_id = torch.mean(_dx(D).abs() * torch.exp(-a2 * _dx(R).abs())) \
        + torch.mean(_dy(D).abs() * torch.exp(-a2 * _dy(R).abs()))
    # fourier_spectrum_loss(x, S, cutoff=0.1, 'l1')                                      # :456-473
    m = fourier_mask(x.shape[2], x.shape[3], dtype=x.dtype)[None, None]
    l_f = torch.mean(torch.abs(torch.abs(torch.fft.fft2(x) * m) - torch.abs(torch.fft.fft2(S) * m)))
    l_sp = torch.mean(torch.abs(S[:, 1:] - S[:, :-1]))                                   # :475-481
    return l_rec, l_rf, l_il, l_id, l_f, l_sp


def total_from_terms(terms, coefs):
    l_rec, l_rf, l_il, l_id, l_f, l_sp = terms
    return (coefs["c_rec"] * l_rec + coefs["c_rf"] * l_rf + coefs["c_il"] * l_il
            + coefs["c_id"] * l_id + coefs["c_f"] * l_f + coefs["c_sp"] * l_sp)      # :557-564


def compute_loss(P, x, coefs, tr=None):
    """model.py:544-575 -> (total 0-d tensor, dict of 7 floats, (R, I, D, S, E))."""
    R, I, D, S = enhance_forward(P, x, tr)
    E, _ = decomposition(P, S, tr=tr, tag="_2")                                          # :546
    terms = loss_terms(x, R, I, D, S, E, coefs)
    total = total_from_terms(terms, coefs)
    vals = dict(zip(LOSS_KEYS, [float(total.detach())] + [float(t.detach()) for t in terms]))
    return total, vals, (R, I, D, S, E)


def loss_and_grads(P, x, coefs, tr=None):
    """compute_loss + autograd backward (model.py:314-315) -> (loss dict, grads dict, outputs)."""
    Pg = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in P.items())
    total, vals, outs = compute_loss(Pg, x, coefs, tr)
    total.backward()
    grads = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v))) for k, v in Pg.items())
    return vals, grads, tuple(o.detach() for o in outs)


def grads_from_cotangents(P, x, cot, tr=None, masks=None):
    """Backward chain alone (model.py:315 with the loss section replaced by FIXED direct cotangents): parameter gradients of the
    linear surrogate <gR,R> + <gI,I> + <gD,D> + <gS,S> + <gE,E>, where cot = dict(gR, gI, gD, gS, gE) are constants and
    (R, I, D, S) = forward(x), E = decomposition(S)[0] (model.py:545-546).  No sg() of a loss term is involved, so two
    evaluations of this chain differ only by ordinary rounding (and ReLU decisions at |y| ~ 0: pass the other side's decisions as
    `masks`, see _relu)."""
    Pg = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in P.items())
    R, I, D, S = enhance_forward(Pg, x, tr, masks)
    E, _ = decomposition(Pg, S, tr=tr, tag="_2", masks=masks)
    sur = sum((cot[k].to(t.dtype) * t).sum() for k, t in (("gR", R), ("gI", I), ("gD", D), ("gS", S), ("gE", E)))
    sur.backward()
    grads = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v))) for k, v in Pg.items())
    return grads, tuple(o.detach() for o in (R, I, D, S, E))


# --------------------------------------------------------------------------
# Adam, torch.optim.Adam defaults (model.py:213): betas (0.9, 0.999), eps 1e-8,
# no weight decay, no amsgrad;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
# --------------------------------------------------------------------------
class AdamState:
    def __init__(self, P):
        self.step = 0
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in P.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in P.items())


def adam_step(P, grads, state: AdamState, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    state.step += 1
    t = state.step
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    for k in P:
        g = grads[k]
        state.m[k].mul_(b1).add_(g, alpha=1 - b1)
        state.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (state.v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        P[k] = P[k] - (lr / bc1) * (state.m[k] / denom)
    return P


def train_step(P, x, coefs, state: AdamState, lr=1e-3):
    """zero_grad -> compute_loss -> backward -> Adam.step  (model.py:313-316)."""
    vals, grads, outs = loss_and_grads(P, x, coefs)
    P = adam_step(P, grads, state, lr=lr)
    return P, vals, grads, outs


def psnr(a, b, data_range=1.0):
    """10 log10(data_range^2 / MSE) over all elements (torchmetrics semantics used at metrics.py:122)."""
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * math.log10(data_range ** 2 / mse)
