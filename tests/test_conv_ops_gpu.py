"""GPU parity of the MFMA implicit-GEMM conv operators (through the C-ABI) against a plain PyTorch
CPU float64 reference of the same op.  Tolerance: |err| <= 2e-5 * max|ref| (fp32 fma chains of up to
K = 2592 terms; SURVEY §8(c) measured fp32-vs-fp64 self-agreement of the reference at ~1e-7..1e-6).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(scope="module")
def H():
    import ssie
    ssie.load()
    from ssie_amd import hostlib
    assert hostlib.lib().ssie_device_ok() == 1
    return hostlib


@pytest.fixture(autouse=True, params=["heuristic", "tile16x16", "tile16x32", "winograd", "winograd4", "tconv1"])
def kernel_mode(H, request):
    """The launch heuristics pick 8x16 tiles (register-staged kernel) for launches as small as these tests; the two forced
    modes run the same cases through the DMA kernels the bench-size layers use: 16x16 tiles (conv_fprop_v2_kernel) and
    16x32 tiles (conv_fprop_v2w_kernel, where eligible: stride 1, <= 9 taps, 64-multiple Cout, Wo >= 32).  "winograd" runs
    every stride-1 3x3 forward / data-gradient launch on conv_wino_kernel (F(2x2, 3x3), conv_wino.hip), whatever its size;
    the other three modes keep it off so the direct kernels stay covered.  "tconv1" runs every eligible stride-2 transposed 3x3
    convolution (ConvTranspose2d forward, stride-2 data gradient) on the one-launch kernel (conv_tconv.hip) whatever its size."""
    L = H.lib()
    # "winograd4": every eligible stride-1 3x3 forward / data-gradient launch (sources at their own resolution, width >= 48) on
    # conv_wino4_kernel (F(4x4, 3x3), conv_wino4.hip), the rest on F(2x2, 3x3); "winograd" keeps F(4x4) off so F(2x2) stays covered
    L.ssie_debug_set_wino4_min_tiles(1 if request.param == "winograd4" else 1 << 30)
    L.ssie_debug_set_wino_min_tiles(1 if request.param in ("winograd", "winograd4") else 1 << 30)
    L.ssie_debug_set_tconv_min_tiles(1 if request.param == "tconv1" else 1 << 30)
    L.ssie_debug_set_wgrad_wino_min_tiles(1 if request.param == "winograd" else 1 << 30)   # weight gradients: F(3x3,2x2)
    if request.param not in ("heuristic", "winograd", "winograd4"):
        L.ssie_debug_set_fprop_min_tiles16(0)
        L.ssie_debug_set_fprop_wide_min_tiles(1 if request.param == "tile16x32" else 1 << 30)
        L.ssie_debug_set_fprop_v2_split_min_tiles(1 if request.param == "tile16x32" else 1 << 30)   # 32-channel layers: two 4-wave workgroups per CU
    L.ssie_debug_set_fprop_v2_onetap(2 if request.param == "tile16x16" else 1)      # 1 x 1 layers: the 1-tap instantiation whatever the size
    # F(2x2,3x3) workgroup width: "winograd" = 16 output channels wherever the launch is whole tiles (the under-filled form), "winograd4" = 32
    L.ssie_debug_set_wino_half_below(0 if request.param == "winograd4" else 1 << 30)
    yield request.param
    L.ssie_debug_set_wino_half_below(256)
    L.ssie_debug_set_fprop_v2_onetap(1)
    L.ssie_debug_set_tconv_min_tiles(-1)
    L.ssie_debug_set_wino4_min_tiles(-1)              # the library's default
    L.ssie_debug_set_wino_min_tiles(-1)               # the library's default
    L.ssie_debug_set_wgrad_wino_min_tiles(-1)
    L.ssie_debug_set_fprop_min_tiles16(256)
    L.ssie_debug_set_fprop_wide_min_tiles(512)
    L.ssie_debug_set_fprop_v2_split_min_tiles(1024)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


def to_dev(H, t):
    """NCHW float64 cpu -> NHWC fp32 cuda padded"""
    return H.nhwc(t.float().cuda())


def from_dev(buf, c):
    return buf[..., :c].permute(0, 3, 1, 2).double().cpu()


def close(got, ref, tol=TOL):
    scale = max(ref.abs().max().item(), 1e-30)
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e} (rel {err/scale:.3e})"


@pytest.mark.parametrize("cin,cout,k,stride,h,w,act", [
    (32, 64, 3, 1, 16, 32, 1),
    (32, 64, 3, 1, 20, 24, 0),
    (31, 64, 9, 1, 16, 16, 0),
    (31, 32, 3, 1, 18, 22, 1),
    (64, 128, 3, 2, 32, 32, 1),
    (64, 64, 3, 2, 26, 50, 1),
    (128, 128, 3, 1, 16, 16, 1),
    (64, 32, 3, 1, 16, 16, 2),
    (64, 1, 3, 1, 16, 32, 0),
    (5, 32, 3, 1, 16, 16, 1),
    (5, 64, 9, 1, 24, 16, 0),
    (64, 192, 1, 1, 8, 32, 0),
    (64, 64, 1, 1, 32, 16, 1),          # whole 16 x 16 tiles: the 1-tap two-workgroups-per-CU form in the tile16x16 mode
    (64, 64, 3, 1, 32, 64, 1),
    (96, 64, 3, 1, 20, 40, 0),
    (128, 128, 3, 1, 16, 48, 1),
    (48, 96, 3, 1, 33, 47, 1),          # three K chunks, 96 -> padded 128 output channels, odd ragged sizes
    (128, 31, 3, 1, 17, 40, 0),
    (256, 32, 3, 1, 16, 32, 1),         # the 256-band layers of BASELINE configs[2]: conv0 (16 K chunks),
    (64, 257, 3, 1, 16, 32, 2),         # recon (257 = 9 output-channel blocks, ragged last one), sigmoid
    (257, 64, 3, 1, 16, 32, 1),         # the illumination net's conv0 on cat[R, I]
    (256, 64, 9, 1, 16, 16, 0),         # shallow_conv (direct 9 x 9 kernels; the spectral path is a plan-level test)
    (65, 64, 3, 1, 16, 32, 0),          # 64 bands, the reference's own count (model.py:178): illumination conv0 on cat[R, I] = 65 channels,
    (64, 65, 3, 1, 16, 32, 2),          # recon with 65 outputs (a ragged third 32-channel block), sigmoid,
    (64, 64, 9, 1, 16, 16, 0),          # shallow_conv at 64 input channels
    (64, 32, 3, 1, 18, 22, 1),          # conv0 at 64 bands
    (64, 64, 3, 1, 20, 64, 1),          # widths the F(4x4, 3x3) kernel takes (>= 48 columns, <= 1/4 of the 64-wide tile wasted):
    (31, 32, 3, 1, 33, 130, 1),         # ragged rows and columns (3 tile columns, the last one 2 wide), partial channel step (31 = 3 x 8 + 7)
    (96, 64, 3, 1, 16, 48, 0),          # 12 K steps
    (128, 128, 3, 1, 32, 64, 1),        # four output-channel blocks
    (68, 64, 3, 1, 17, 64, 0),          # 65 -> 68 padded channels: the last step holds one channel quad
    (64, 31, 3, 1, 16, 128, 2),         # ragged output-channel block, sigmoid
])
def test_conv2d_fwd(H, cin, cout, k, stride, h, w, act):
    n = 2
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, k, k, seed=2, scale=0.2)
    b = rnd(cout, seed=3)
    ref = F.conv2d(x, wt, b, stride=stride, padding=(k - 1) // 2)
    ref = F.relu(ref) if act == 1 else (torch.sigmoid(ref) if act == 2 else ref)
    xb = to_dev(H, x)
    out = H.conv2d_fwd([(xb, (cin + 3) // 4 * 4, 0)], h, w, wt.float().cuda(), b.float().cuda(), k, stride, act)
    torch.cuda.synchronize()
    close(from_dev(out, cout), ref)


def test_conv2d_fwd_concat_upsample_skip(H):
    """conv7-like concat (64+32), deconvN-like nearest-upsample-on-read + skip add + out2, fusion-like 3 sources."""
    n, h, w = 2, 16, 32
    a = rnd(n, 64, h, w, seed=1); c = rnd(n, 32, h, w, seed=2)
    wt = rnd(64, 96, 3, 3, seed=3, scale=0.1); b = rnd(64, seed=4)
    ref = F.conv2d(torch.cat([a, c], 1), wt, b, padding=1)
    out = H.conv2d_fwd([(to_dev(H, a), 64, 0), (to_dev(H, c), 32, 0)], h, w, wt.float().cuda(), b.float().cuda(), 3)
    close(from_dev(out, 64), ref)

    for (hs, ws_, hv, wv) in ((8, 16, 16, 32), (13, 7, 25, 13)):
        lo = rnd(n, 64, hs, ws_, seed=5); skip = rnd(n, 64, hv, wv, seed=6)
        up = F.interpolate(lo.float(), size=(hv, wv), mode="nearest").double()
        pre = F.relu(F.conv2d(up, wt[:, :64], b, padding=1))
        out, out2 = H.conv2d_fwd([(to_dev(H, lo), 64, 0)], hv, wv, wt[:, :64].contiguous().float().cuda(),
                                 b.float().cuda(), 3, act=1, addsrc=to_dev(H, skip), want_out2=True)
        close(from_dev(out2, 64), pre)
        close(from_dev(out, 64), pre + skip)

    d1 = rnd(n, 64, 4, 8, seed=7); d2 = rnd(n, 64, 8, 16, seed=8); d3 = rnd(n, 64, 16, 32, seed=9)
    w1 = rnd(64, 192, 1, 1, seed=10, scale=0.1)
    cat = torch.cat([F.interpolate(d1.float(), size=(16, 32), mode="nearest").double(),
                     F.interpolate(d2.float(), size=(16, 32), mode="nearest").double(), d3], 1)
    ref = F.conv2d(cat, w1, b)
    out = H.conv2d_fwd([(to_dev(H, d1), 64, 0), (to_dev(H, d2), 64, 0), (to_dev(H, d3), 64, 0)], 16, 32,
                       w1.float().cuda(), b.float().cuda(), 1)
    close(from_dev(out, 64), ref)


@pytest.mark.parametrize("cin,cout,h,w", [(128, 64, 16, 16), (64, 64, 9, 20), (16, 32, 8, 8)])
def test_conv_transpose2d_fwd(H, cin, cout, h, w):
    n = 2
    x = rnd(n, cin, h, w, seed=1); wt = rnd(cin, cout, 3, 3, seed=2, scale=0.1); b = rnd(cout, seed=3)
    ref = F.relu(F.conv_transpose2d(x, wt, b, stride=2, padding=1, output_padding=1))
    out = H.conv_transpose2d_fwd(to_dev(H, x), wt.float().cuda(), b.float().cuda(), act=1)
    close(from_dev(out, cout), ref)


@pytest.mark.parametrize("cin,cout,h,w", [(128, 64, 16, 16), (64, 64, 9, 20), (64, 64, 24, 40)])
def test_conv_transpose2d_16_row_tiles(H, kernel_mode, cin, cout, h, w):
    """conv_tconv_kernel has two tile heights: 8 input rows where 16-row tiles would leave most CUs without one (every case of this
    file, and the reference's shipped batch of 2), 16 rows otherwise (the bench sizes).  The cases above run the 8-row form in the
    "tconv1" mode; this one forces 16 rows on the same shapes (incl. ragged 9 x 20 and a multi-tile 24 x 40)."""
    if kernel_mode != "tconv1":
        pytest.skip("the one-launch transposed convolution is forced in the tconv1 mode only")
    L = H.lib()
    L.ssie_debug_set_tconv_half_tiles_below(0)
    try:
        n = 2
        x = rnd(n, cin, h, w, seed=1); wt = rnd(cin, cout, 3, 3, seed=2, scale=0.1); b = rnd(cout, seed=3)
        ref = F.relu(F.conv_transpose2d(x, wt, b, stride=2, padding=1, output_padding=1))
        out = H.conv_transpose2d_fwd(to_dev(H, x), wt.float().cuda(), b.float().cuda(), act=1)
        close(from_dev(out, cout), ref)
    finally:
        L.ssie_debug_set_tconv_half_tiles_below(257)


def _conv_grads(x, wt, stride, g, transposed=False):
    x = x.clone().requires_grad_(True); wt = wt.clone().requires_grad_(True)
    b = torch.zeros(wt.shape[1] if transposed else wt.shape[0], dtype=torch.float64, requires_grad=True)
    if transposed:
        y = F.conv_transpose2d(x, wt, b, stride=2, padding=1, output_padding=1)
    else:
        y = F.conv2d(x, wt, b, stride=stride, padding=(wt.shape[-1] - 1) // 2)
    (y * g).sum().backward()
    return x.grad, wt.grad, b.grad


@pytest.mark.parametrize("cin,cout,k,stride,h,w", [
    (64, 64, 3, 1, 16, 32),
    (31, 64, 9, 1, 16, 16),
    (96, 64, 3, 1, 12, 20),
    (64, 128, 3, 2, 32, 32),
    (64, 64, 3, 2, 25, 26),
    (64, 32, 3, 1, 16, 16),
    (64, 1, 3, 1, 16, 16),
    (192, 64, 1, 1, 8, 16),
    (64, 64, 1, 1, 16, 32),
    (5, 32, 3, 1, 16, 16),
    (128, 128, 3, 1, 16, 16),
    (31, 32, 3, 1, 18, 22),
    (32, 64, 3, 1, 25, 13),
    (48, 96, 3, 1, 33, 47),
    (128, 31, 3, 1, 17, 40),
    (256, 32, 3, 1, 16, 32),
    (64, 257, 3, 1, 16, 32),
    (257, 64, 3, 1, 16, 32),
    (256, 64, 9, 1, 16, 16),
    (65, 64, 3, 1, 16, 32),
    (64, 65, 3, 1, 16, 32),
    (64, 64, 9, 1, 16, 16),
    (64, 64, 3, 1, 20, 64),             # F(4x4, 3x3)-eligible widths (data gradient incl. mask + accumulate, split input channels)
    (32, 64, 3, 1, 33, 130),
    (128, 128, 3, 1, 16, 64),
    (96, 64, 3, 1, 12, 48),
])
def test_conv2d_dgrad_wgrad(H, cin, cout, k, stride, h, w):
    n = 2
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    x = rnd(n, cin, h, w, seed=1); wt = rnd(cout, cin, k, k, seed=2, scale=0.1); g = rnd(n, cout, ho, wo, seed=3)
    gx_ref, dw_ref, db_ref = _conv_grads(x, wt, stride, g)
    gb = to_dev(H, g); wd = wt.float().cuda()
    # dgrad in one piece, and (when cin splits into 2 sources) piecewise with accumulate + relu mask
    gx = H.conv2d_dgrad(gb, cout, wd, 0, cin, k, stride, h, w)
    close(from_dev(gx, cin), gx_ref)
    y = rnd(n, cin, h, w, seed=4)
    gx2 = torch.ones(n, h, w, (cin + 3) // 4 * 4, device="cuda")
    gx2 = H.conv2d_dgrad(gb, cout, wd, 0, cin, k, stride, h, w, mask_y=to_dev(H, y), mask_mode=1, gx=gx2)
    close(from_dev(gx2, cin), 1.0 + gx_ref * (y > 0))
    if cin % 32 == 0 and cin >= 64:
        half = cin // 2 if cin != 96 else 64
        lo = H.conv2d_dgrad(gb, cout, wd, 0, half, k, stride, h, w)
        hi = H.conv2d_dgrad(gb, cout, wd, half, cin - half, k, stride, h, w)
        close(from_dev(lo, half), gx_ref[:, :half]); close(from_dev(hi, cin - half), gx_ref[:, half:])
    # wgrad
    dw, db = H.conv2d_wgrad((to_dev(H, x), (cin + 3) // 4 * 4, 0), h, w, gb, cout, k, stride, cin, 0)
    torch.cuda.synchronize()
    close(dw.double().cpu(), dw_ref); close(db.double().cpu(), db_ref)


def test_conv2d_wgrad_concat_offset_and_upsample(H):
    n, h, w = 2, 16, 16
    a = rnd(n, 64, h, w, seed=1); c = rnd(n, 32, h, w, seed=2)
    wt = rnd(64, 96, 3, 3, seed=3, scale=0.1); g = rnd(n, 64, h, w, seed=4)
    _, dw_ref, db_ref = _conv_grads(torch.cat([a, c], 1), wt, 1, g)
    gb = to_dev(H, g)
    dw = torch.zeros(64, 96, 3, 3, device="cuda"); db = torch.zeros(64, device="cuda")
    H.conv2d_wgrad((to_dev(H, a), 64, 0), h, w, gb, 64, 3, 1, 96, 0, dw=dw, db=db)
    H.conv2d_wgrad((to_dev(H, c), 32, 0), h, w, gb, 64, 3, 1, 96, 64, dw=dw, db=None)
    close(dw.double().cpu(), dw_ref); close(db.double().cpu(), db_ref)
    lo = rnd(n, 64, 8, 8, seed=5)
    up = F.interpolate(lo.float(), size=(16, 16), mode="nearest").double()
    _, dw_ref, _ = _conv_grads(up, wt[:, :64].contiguous(), 1, g)
    dw, _ = H.conv2d_wgrad((to_dev(H, lo), 64, 0), 16, 16, gb, 64, 3, 1, 64, 0)
    close(dw.double().cpu(), dw_ref)


@pytest.mark.parametrize("cin,cout,h,w", [(128, 64, 16, 16), (64, 64, 9, 20)])
def test_conv_transpose2d_grads(H, cin, cout, h, w):
    n = 2
    x = rnd(n, cin, h, w, seed=1); wt = rnd(cin, cout, 3, 3, seed=2, scale=0.1); g = rnd(n, cout, 2 * h, 2 * w, seed=3)
    gx_ref, dw_ref, db_ref = _conv_grads(x, wt, 2, g, transposed=True)
    gb = to_dev(H, g)
    gx = H.conv_transpose2d_dgrad(gb, wt.float().cuda())
    close(from_dev(gx, cin), gx_ref)
    dw, db = H.conv_transpose2d_wgrad(to_dev(H, x), gb, cin, cout)
    close(dw.double().cpu(), dw_ref); close(db.double().cpu(), db_ref)
