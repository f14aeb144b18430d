"""Standalone loss operator `ssie_selfsup_loss_fwd_bwd` (C-ABI) vs the hand-derived fp64 cotangents of
oracle/loss_cotangents.py (proven equal to autograd in tests/test_oracle_golden.py) - ELEMENTWISE.

Every loss of /root/reference/model.py:551-555 is an L1, so each cotangent element is a sum of sg(.) terms.  The test
names, per output element, the sg() arguments that feed it; an element may differ from the fp64 value ONLY where one of
those arguments is smaller than 1e-6 in magnitude AND involves a rounded intermediate (fp32 evaluation may land on the
other side of 0 there).  Every other element must agree to 1e-5 of the tensor's largest magnitude.  To keep a large term from hiding a small one the check runs
with each coefficient alone (one-hot) as well as with the reference's two coefficient sets.

The Fourier term's cotangent is a dense inverse transform of per-bin signs, so it is checked in the frequency domain:
fft2(gS) is the Hermitian part of M*g_Z, which must agree bin by bin except where |M F(x)| - |M F(S)| is ambiguous.
"""
import numpy as np
import pytest
import torch

from oracle import loss_cotangents as LC
from oracle import ssie_oracle as O

pytestmark = pytest.mark.gpu
EPS = 1e-6

ZERO = dict(c_rec=0.0, c_rf=0.0, c_il=0.0, c_id=0.0, c_f=0.0, c_sp=0.0, alpha_low=1.0, alpha_delta=10.0)
COEF_SETS = {"jyu_nofourier": dict(O.JYU_COEFS, c_f=0.0), "default_nofourier": dict(O.DEFAULT_COEFS, c_f=0.0)}
for _k in ("c_rec", "c_rf", "c_il", "c_id", "c_sp"):
    COEF_SETS["only_" + _k] = dict(ZERO, **{_k: O.JYU_COEFS[_k]})

GEOMS = {"b5_16": (1, 5, 16, 16), "b31_32": (2, 31, 32, 32), "b31_64": (2, 31, 64, 64), "b8_24x40": (2, 8, 24, 40),
         "b31_128": (1, 31, 128, 128), "b40_48": (1, 40, 48, 48),       # 40 bands: 16 lanes per pixel in the tiled kernel
         "b64_24x40": (1, 64, 24, 40), "b130_16x24": (1, 130, 16, 24),   # 32 and 64 lanes per pixel (shipped configs use 64 bands)
         "b12_20x36": (2, 12, 20, 36),                                   # B % 4 == 0: I_low sits in a float4 of its own
         # more than 252 bands: the band-chunked tiled kernel (loss_chunk_kernel) is the default path.  256 = BASELINE configs[2]
         # (I_low in a float4 of its own, four full 64-band chunks), 258: I_low inside the last band float4, ragged last chunk
         "b256_16x24": (1, 256, 16, 24), "b258_12x20": (1, 258, 12, 20)}


@pytest.fixture(scope="module")
def H():
    import ssie
    ssie.load()
    from ssie_amd import hostlib
    assert hostlib.lib().ssie_device_ok() == 1
    return hostlib


@pytest.fixture(params=["tiled", "half_wave", "chunked"])
def loss_kernel(H, request):
    """all three loss kernels through every case: the LDS-tiled one the plan runs up to 252 bands, the band-chunked tiled one it
    runs above that (forced here for every band count), and the half-wave-per-pixel one for layouts neither takes
    (include/ssie_debug.h)"""
    H.lib().ssie_debug_set_loss_generic(1 if request.param == "half_wave" else 0)
    H.lib().ssie_debug_set_loss_chunked(1 if request.param == "chunked" else 0)
    yield request.param
    H.lib().ssie_debug_set_loss_generic(0)
    H.lib().ssie_debug_set_loss_chunked(0)


_cache = {}


def leaves(geom):
    """(x, R, I, D, S, E) fp32 from the CPU oracle forward with the closed-form parameters"""
    if geom not in _cache:
        n, b, h, w = GEOMS[geom]
        P = O.closed_form_params(b)
        x = O.synthetic_patches(n, b, h, w)
        with torch.no_grad():
            R, I, D, S = O.enhance_forward(P, x)
            E, _ = O.decomposition(P, S)
        _cache[geom] = tuple(t.contiguous() for t in (x, R, I, D, S, E))
    return _cache[geom]


def _edges_x(a):     # a over (..., H, W-1) edge set -> (..., H, W): pixel touches edge w and edge w-1
    out = torch.zeros(a.shape[:-1] + (a.shape[-1] + 1,), dtype=torch.bool)
    out[..., 1:] |= a; out[..., :-1] |= a
    return out


def _edges_y(a):
    out = torch.zeros(a.shape[:-2] + (a.shape[-2] + 1, a.shape[-1]), dtype=torch.bool)
    out[..., 1:, :] |= a; out[..., :-1, :] |= a
    return out


def ambiguity(x, R, I, D, S, E):
    """boolean masks: True where an sg() argument feeding that cotangent element may round across 0 in fp32.
    Arguments that are ONE subtraction of two given fp32 values (dR, dI, dD, band differences of S, R - E) have an exact sign
    in fp32 and in fp64 alike, so they are never ambiguous; only R*I - x (a rounded product) and the differences of
    differences d(R - E) can be, and only within EPS of 0."""
    near = lambda t: t.abs() < EPS
    a_rec = near(R * I - x)
    d = R - E
    a_dd = _edges_x(near(LC._dx(d))) | _edges_y(near(LC._dy(d)))
    false = lambda t: torch.zeros_like(t, dtype=torch.bool)
    return {"gR": a_rec | a_dd, "gE": a_dd, "gI": a_rec.any(1, keepdim=True), "gD": false(D), "gS": false(S)}


@pytest.mark.parametrize("coefset", list(COEF_SETS))
@pytest.mark.parametrize("geom", list(GEOMS))
def test_spatial_terms_elementwise(H, geom, coefset, loss_kernel):
    coefs = COEF_SETS[coefset]
    t32 = leaves(geom)
    scal, got = H.selfsup_loss_fwd_bwd(*[t.cuda() for t in t32], coefs)
    torch.cuda.synchronize()
    t64 = [t.double() for t in t32]
    ref = LC.direct_cotangents(*t64, coefs)
    amb = ambiguity(*t64)
    terms = O.loss_terms(*t64, coefs)
    ref_scal = [float(O.total_from_terms(terms, coefs))] + [float(t) for t in terms]
    for k, (g, r) in enumerate(zip(scal.cpu().double().tolist(), ref_scal)):
        if k == 5:
            continue                                   # L_fourier is reported whatever c_f is; checked in the Fourier test
        if k == 0:
            r -= coefs["c_f"] * ref_scal[5]
            g -= coefs["c_f"] * float(scal[5])
        assert abs(g - r) <= 2e-5 * abs(r) + 1e-12, (O.LOSS_KEYS[k], g, r)
    for key in ("gR", "gI", "gD", "gS", "gE"):
        g = got[key].cpu().double(); r = ref[key]
        scale = r.abs().max().item()
        if scale == 0.0:
            assert g.abs().max().item() == 0.0, key
            continue
        bad = (g - r).abs() > 1e-5 * scale
        frac_amb = amb[key].double().mean().item()
        assert frac_amb < 0.05, (key, frac_amb)                     # the exemption must stay a small minority
        unexplained = bad & ~amb[key]
        assert not unexplained.any(), (key, int(unexplained.sum()), int(bad.sum()), scale,
                                       ((g - r).abs() * (~amb[key])).max().item() / scale)


@pytest.fixture(params=["grouped", "whole_plane"])
def fft_kernel(H, request):
    """planes of 64 x 64 and more with a power-of-two width take the three-pass path with band-grouped rows
    (fft_rows_*_grouped_kernel) by default; the second value keeps the whole-plane-in-LDS kernel covered on them (smaller planes
    and the planes that exceed the LDS run the same kernels either way)"""
    H.lib().ssie_debug_set_fft_grouped(1 if request.param == "grouped" else 0)
    yield request.param
    H.lib().ssie_debug_set_fft_grouped(1)


@pytest.mark.parametrize("geom", ["b5_16", "b31_32", "b31_64", "b8_24x40", "b31_128", "b5_256", "b4_160x288", "b31_128_n10", "b6_64x128", "b3_256x64"])
def test_fourier_term_per_bin(H, geom, fft_kernel):
    """only c_f non-zero: gS = c_f Re(HW ifft2(M g_Z));  fft2(gS)/ (HW) = Hermitian part of c_f M g_Z, compared bin by bin.
    b5_256 / b4_160x288 take the three-pass path for planes larger than the LDS (model.py:456-473 accepts any patch size)."""
    if fft_kernel != "grouped" and geom not in ("b31_64", "b31_128"):
        pytest.skip("same kernel as the first parameter value")
    if geom == "b5_256":
        n, b, h, w = 1, 5, 256, 256
    elif geom == "b31_128_n10":                        # 10 patches of 31 bands (groups of 16 + 15)
        n, b, h, w = 10, 31, 128, 128
    elif geom == "b4_160x288":
        n, b, h, w = 2, 4, 160, 288
    elif geom == "b6_64x128":                          # band groups of 8; H != W
        n, b, h, w = 3, 6, 64, 128
    elif geom == "b3_256x64":                          # band groups of 4 (one float4 holds all bands + padding); 256 rows
        n, b, h, w = 2, 3, 256, 64
    else:
        n, b, h, w = GEOMS[geom]
    if geom in GEOMS:
        x, R, I, D, S, E = leaves(geom)
    else:                                               # big planes: only x and S matter for this term
        x = O.synthetic_patches(n, b, h, w)
        S = (x * (2.2 + 0.5 * torch.sin(0.05 * torch.arange(w, dtype=torch.float32))) + 0.02).contiguous()
        R = torch.full_like(x, 0.5); E = R.clone(); I = torch.full((n, 1, h, w), 0.4); D = torch.full((n, 1, h, w), 0.2)
    coefs = dict(ZERO, c_f=O.JYU_COEFS["c_f"])
    scal, got = H.selfsup_loss_fwd_bwd(*[t.cuda() for t in (x, R, I, D, S, E)], coefs)
    torch.cuda.synchronize()
    x64, S64 = x.double(), S.double()
    m = O.fourier_mask(h, w, dtype=torch.float32).double()[None, None]
    Zx = torch.fft.fft2(x64) * m; Z = torch.fft.fft2(S64) * m
    A = Z.abs(); diff = Zx.abs() - A
    l_f = diff.abs().mean().item()
    assert abs(float(scal[5]) - l_f) <= 2e-5 * l_f, (float(scal[5]), l_f)
    assert abs(float(scal[0]) - coefs["c_f"] * l_f) <= 2e-5 * coefs["c_f"] * l_f
    n0 = x.numel()
    gZ = torch.where(A > 0, -torch.sign(diff) / n0 * Z / torch.where(A > 0, A, torch.ones_like(A)), torch.zeros_like(Z))
    Y = coefs["c_f"] * m * gZ
    flip = lambda t: torch.roll(torch.flip(t, (-2, -1)), (1, 1), (-2, -1))       # bin -k
    Yh = 0.5 * (Y + flip(Y).conj())                                              # spectrum of the REAL part of the adjoint
    # an fp32 FFT carries an absolute rounding error of ~1e-7 * ||plane||_2 in EVERY bin (torch's CPU fp32 fft2 too): `thr`
    # bounds it.  A bin is ambiguous when | |Zx| - |Z| | < thr; elsewhere the direction Z/|Z| is known to thr/|Z|.
    thr = 1e-6 * torch.sqrt((x64 ** 2 + S64 ** 2).sum((-2, -1), keepdim=True))
    near = (diff.abs() < thr) & (m > 0)
    amb = near | flip(near)
    inv = torch.where(A > 0, 1.0 / torch.where(A > 0, A, torch.ones_like(A)), torch.zeros_like(A))
    G = torch.fft.fft2(got["gS"].cpu().double()) / (h * w)
    err = (G - Yh).abs()
    scale = Yh.abs().max().item()                      # = c_f / n0 (a bin's value is that times a unit phase)
    tol = scale * (1e-4 + 8.0 * thr * (inv + flip(inv)))
    bad = err > tol
    assert amb.double().mean().item() < 0.02
    unexplained = bad & ~amb
    assert not unexplained.any(), (int(unexplained.sum()), int(bad.sum()), ((err / tol) * (~amb)).max().item())
    # the other cotangents get nothing from this term
    for key in ("gR", "gI", "gD", "gE"):
        assert got[key].abs().max().item() == 0.0, key
