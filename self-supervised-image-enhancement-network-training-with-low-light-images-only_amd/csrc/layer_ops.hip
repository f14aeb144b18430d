// Host-side geometry builders: turn "a layer of the reference network" into launches of the
// implicit-GEMM kernels, and the granular C-ABI operators built on them.
#include "layer_ops.h"
#include "../../include/ssie_hip.h"
#include <string.h>

int ssie_fprop_tile16 = 0;   // tuning knob (tools/): 0 forces 8-row tiles everywhere
int ssie_fprop_min_tiles16 = 256;   // fewer 16 x 16 tiles than this: 8 x 16 tiles (and 32-channel splits) instead
extern "C" void ssie_debug_set_fprop_min_tiles16(int v) { ssie_fprop_min_tiles16 = v; }
int ssie_fprop_wide = 1;     // 1: 16 x 32 tiles (conv_fprop_v2w_kernel) for the big 64-channel stride-1 3x3 layers
int ssie_fprop_wide_min_tiles = 512;   // ... when the launch has at least this many of them (tests set 1 to force the kernel)
extern "C" void ssie_debug_set_fprop_wide(int v) { ssie_fprop_wide = v; }
extern "C" void ssie_debug_set_fprop_wide_min_tiles(int v) { ssie_fprop_wide_min_tiles = v; }
extern "C" void ssie_debug_set_fprop_tile16(int v) { ssie_fprop_tile16 = v; }

// ---------------------------------------------------------------------------------------------
// tap lists
// ---------------------------------------------------------------------------------------------
TapList ssie_taps_conv(int k)
{
    TapList t; t.n = 0; const int pad = (k - 1) / 2;
    for (int kh = 0; kh < k; ++kh) for (int kw = 0; kw < k; ++kw) {
        t.dy[t.n] = (int8_t)(kh - pad); t.dx[t.n] = (int8_t)(kw - pad); t.sel[t.n] = (int8_t)(kh * k + kw); ++t.n;
    }
    return t;
}
// data gradient of a stride-1 conv: gX[y] = sum_kh G[y + pad - kh] W[kh]
TapList ssie_taps_dgrad_s1(int k)
{
    TapList t; t.n = 0; const int pad = (k - 1) / 2;
    for (int kh = 0; kh < k; ++kh) for (int kw = 0; kw < k; ++kw) {
        t.dy[t.n] = (int8_t)(pad - kh); t.dx[t.n] = (int8_t)(pad - kw); t.sel[t.n] = (int8_t)(kh * k + kw); ++t.n;
    }
    return t;
}
// transposed conv (stride 2): output y = 2a - pad + kh.  For output parity py only taps with
// kh = py + pad (mod 2) contribute, reading input row a' + (py + pad - kh)/2.
TapList ssie_taps_transposed(int k, int pad, int py, int px)
{
    TapList t; t.n = 0;
    for (int kh = 0; kh < k; ++kh) {
        int ny = py + pad - kh; if (ny & 1) continue;
        for (int kw = 0; kw < k; ++kw) {
            int nx = px + pad - kw; if (nx & 1) continue;
            t.dy[t.n] = (int8_t)(ny / 2); t.dx[t.n] = (int8_t)(nx / 2); t.sel[t.n] = (int8_t)(kh * k + kw); ++t.n;
        }
    }
    return t;
}

TapList ssie_taps_transposed_all(void)
{
    TapList t; t.n = 0;
    for (int py = 0; py < 2; ++py) for (int px = 0; px < 2; ++px) {
        const TapList c = ssie_taps_transposed(3, 1, py, px);
        for (int i = 0; i < c.n; ++i) { t.dy[t.n] = c.dy[i]; t.dx[t.n] = c.dx[i]; t.sel[t.n] = c.sel[i]; ++t.n; }
    }
    return t;
}

// ---- one-launch transposed convolution (conv_tconv.hip) ----
int ssie_fprop_tconv = 1;              // A/B switch: 1 = eligible stride-2 transposed 3x3 convolutions run conv_tconv_kernel
#define SSIE_TCONV_MIN_TILES 8          // (32 until the kernel got its 8-row tiles for under-filled launches: at batch 2 the 32 x 32 level is faster here too, -1 %)
int ssie_fprop_tconv_min_tiles = SSIE_TCONV_MIN_TILES;   // ... when the input has at least this many 16 x 16 tiles (tests set 1): below 256 tiles the four
                                       // parity-class launches are launch-latency-bound (4 x ~20 us at the 16 x 16 / 32 x 32 pyramid levels)
extern "C" void ssie_debug_set_tconv(int v) { ssie_fprop_tconv = v; }
extern "C" void ssie_debug_set_tconv_min_tiles(int v) { ssie_fprop_tconv_min_tiles = v < 0 ? SSIE_TCONV_MIN_TILES : v; }   // v < 0: the default

bool ssie_tconv_eligible(const SrcDesc& in, int N, int Hin, int Win, int Nc)
{
    if (!ssie_fprop_tconv || Nc <= 32 || Nc > 64 || in.sy != 1.f || in.sx != 1.f || in.Hs != Hin || in.Ws != Win || Hin < 16 || Win < 16) return false;
    return (long)N * ssie_ceil_div(Hin, 16) * ssie_ceil_div(Win, 16) >= ssie_fprop_tconv_min_tiles;
}

// re-target a geometry built over ssie_taps_transposed_all (si = 1, so = 2, Ho x Wo = the INPUT grid) at conv_tconv_kernel
// (8-row tiles when the 16-row ones would leave more than half of the CUs without a tile)
int ssie_tconv_half_tiles_below = 257;   // (at most one 16-row tile per CU)
extern "C" void ssie_debug_set_tconv_half_tiles_below(int v) { ssie_tconv_half_tiles_below = v; }
void ssie_conv_to_tconv(ConvParams& p)
{
    const long tiles16 = (long)p.N * ssie_ceil_div(p.Ho, 16) * ssie_ceil_div(p.Wo, 16);
    p.tconv = 1; p.th = tiles16 < ssie_tconv_half_tiles_below ? 8 : 16; p.tw = 16; p.hp_h = p.th + 1; p.hp_w = 17;
    p.tiles_y = ssie_ceil_div(p.Ho, p.th); p.tiles_x = ssie_ceil_div(p.Wo, 16); p.co_blocks = 1;
}

// The kernels index activations with 32-bit element offsets (conv_wino.hip tb_, conv_wgrad_wino.hip tbx_/tbg_, conv_tconv.hip
// off_, the DMA slot tables): a tensor of 2^31 or more floats would wrap into wrong addresses, so geometry builders reject it.
bool ssie_fits_i32(long n, long h, long w, long cstride)
{
    if (n < 0 || h < 0 || w < 0 || cstride < 0) return false;
    const unsigned __int128 e = (unsigned __int128)(unsigned long)n * (unsigned long)h * (unsigned long)w * (unsigned long)cstride;
    return e <= (unsigned __int128)0x7fffffff;
}

static void tap_extent(const TapList& t, int& mn_y, int& mx_y, int& mn_x, int& mx_x)
{
    mn_y = mn_x = 127; mx_y = mx_x = -127;
    for (int i = 0; i < t.n; ++i) {
        if (t.dy[i] < mn_y) mn_y = t.dy[i]; if (t.dy[i] > mx_y) mx_y = t.dy[i];
        if (t.dx[i] < mn_x) mn_x = t.dx[i]; if (t.dx[i] > mx_x) mx_x = t.dx[i];
    }
}

SrcDesc ssie_make_src(const float* ptr, int C, int cstride, int coff, int Hs, int Ws, int Hv, int Wv)
{
    SrcDesc s; s.ptr = ptr; s.C = C; s.cstride = cstride; s.coff = coff; s.Hs = Hs; s.Ws = Ws;
    // F.interpolate(mode='nearest', size=...) uses scale = (float)in / out (model.py:156-169)
    s.sy = (Hs == Hv) ? 1.0f : (float)Hs / (float)Hv;
    s.sx = (Ws == Wv) ? 1.0f : (float)Ws / (float)Wv;
    return s;
}

size_t ssie_packed_floats(int K, int N, int T)
{
    const int npad = N > 32 ? ssie_round_up(N, 64) : 32;
    // + one tap group of padding: the fprop kernel prefetches whole groups and may over-read a short last group
    return (size_t)ssie_ceil_div(K, SSIE_CK) * T * 16 * npad + (size_t)(SSIE_TG + 1) * 16 * npad;
}

PackDesc ssie_make_pack(const float* w, float* dst, int K, int N, const TapList& t, int s_k, int s_n, int s_t)
{
    PackDesc d; memset(&d, 0, sizeof(d));
    d.w = w; d.dst = dst; d.K = K; d.N = N; d.Npad = N > 32 ? ssie_round_up(N, 64) : 32; d.T = t.n;
    d.nchunks = ssie_ceil_div(K, SSIE_CK); d.s_k = s_k; d.s_n = s_n; d.s_t = s_t;
    for (int i = 0; i < t.n; ++i) d.tapsel[i] = t.sel[i];
    return d;
}

int ssie_make_conv(ConvParams& p, const SrcDesc* srcs, int nsrc, int N, int Hv, int Wv, const TapList& t, int si,
                   int Ho, int Wo, const float* wpacked, int Cout,
                   float* out, int Hout, int Wout, int out_cstride, int out_coff, int so, int py, int px,
                   const Epilogue& e)
{
    memset(&p, 0, sizeof(p));
    if (nsrc < 1 || nsrc > SSIE_MAX_SRC || t.n < 1 || t.n > SSIE_MAX_TAPS) return SSIE_E_ARG;
    int cin = 0;
    for (int s = 0; s < nsrc; ++s) {
        p.src[s] = srcs[s];
        if (srcs[s].C % 4 || srcs[s].cstride % 4 || srcs[s].coff % 4) return SSIE_E_SHAPE;
        if (nsrc > 1 && srcs[s].C % SSIE_CK) return SSIE_E_SHAPE;
        if (((uintptr_t)srcs[s].ptr) % 16) return SSIE_E_SHAPE;
        if (!ssie_fits_i32(N, srcs[s].Hs, srcs[s].Ws, srcs[s].cstride)) return SSIE_E_SHAPE;
        cin += srcs[s].C;
    }
    if (!ssie_fits_i32(N, Hout, Wout, out_cstride)) return SSIE_E_SHAPE;
    // the epilogue moves 16 bytes along the channel axis per access (conv_device.h, ssie_epilogue_t)
    if (out_cstride % 4 || out_coff % 4 || ((uintptr_t)out | (uintptr_t)e.bias | (uintptr_t)e.addsrc | (uintptr_t)e.out2 | (uintptr_t)e.mask_y) % 16)
        return SSIE_E_SHAPE;
    p.nsrc = nsrc; p.N = N; p.Hv = Hv; p.Wv = Wv; p.Cin = cin; p.nchunks = ssie_ceil_div(cin, SSIE_CK);
    p.Ho = Ho; p.Wo = Wo; p.si = si; p.ntaps = t.n;
    int mny, mxy, mnx, mxx; tap_extent(t, mny, mxy, mnx, mxx);
    p.min_dy = mny; p.min_dx = mnx;
    // 16-row tiles for the stride-1 layers with a small halo (3x3, 1x1, parity classes): half the weight staging,
    // barriers and tile boundaries per MFMA
    const int span = (mxy - mny) > (mxx - mnx) ? (mxy - mny) : (mxx - mnx);
    // ... but only when 16 x 16 tiles still give every CU a workgroup: at batch 1-2 (the reference configs' batch) a
    // 128 x 128 layer is 64 such tiles, and the 8 x 16 kernel (2-4x the workgroups) finishes it sooner
    const long tiles16 = (long)N * ssie_ceil_div(Ho, 16) * ssie_ceil_div(Wo, 16) * (Cout > 32 ? ssie_round_up(Cout, 64) / 64 : 1);
    p.th = (si == 1 && Ho >= 16 && Wo >= 16 && tiles16 >= ssie_fprop_min_tiles16 &&
            ((span <= 2 && ssie_fprop_tile16) || (span <= 8 && ssie_fprop_use_v2))) ? 16 : 8;
    p.wpacked = wpacked; p.Cout = Cout; p.Cout_pad = Cout > 32 ? ssie_round_up(Cout, 64) : 32;
    // 16 x 32 tiles (wide v2 kernel): 64-channel-multiple outputs, small halo, and at least two tiles per CU
    p.tw = SSIE_TW;
    if (ssie_fprop_wide && p.th == 16 && ssie_fprop_use_v2 && span <= 2 && p.Cout_pad % 64 == 0 && Wo >= 32 &&
        (long)N * ssie_ceil_div(Ho, 16) * ssie_ceil_div(Wo, 32) * (p.Cout_pad / 64) >= ssie_fprop_wide_min_tiles)
        p.tw = 32;
    p.hp_h = (p.th - 1) * si + (mxy - mny) + 1;
    p.hp_w = (p.tw - 1) * si + (mxx - mnx) + 1;
    for (int i = 0; i < t.n; ++i) { p.tap_dy[i] = t.dy[i]; p.tap_dx[i] = t.dx[i]; }
    p.out = out; p.out_cstride = out_cstride; p.out_coff = out_coff; p.Hout = Hout; p.Wout = Wout;
    p.so = so; p.py = py; p.px = px;
    p.bias = e.bias; p.act = e.act; p.addsrc = e.addsrc; p.out2 = e.out2; p.mask_y = e.mask_y;
    p.mask_mode = e.mask_y ? e.mask_mode : MASK_NONE; p.accumulate = e.accumulate;
    p.out2_mode = e.out2 ? e.out2_mode : 0;
    if (p.out2_mode && (!e.mask_y || e.addsrc || e.act != ACT_NONE)) return SSIE_E_ARG;     // modes 1 / 2 are backward-pass forms
    p.tiles_y = ssie_ceil_div(Ho, p.th); p.tiles_x = ssie_ceil_div(Wo, p.tw);
    p.co_blocks = p.Cout_pad > 32 ? p.Cout_pad / 64 : 1;
    return 0;
}

// geometry for conv_fprop_bf16_kernel: bf16 sources (channel counts / strides / offsets multiples of 8, concat pieces
// multiples of 32), 32-channel chunks, 16 x 16 tiles for stride 1 and 8 x 16 for stride 2
int ssie_make_conv_bf16(ConvParams& p, const SrcDesc* srcs, int nsrc, int N, int Hv, int Wv, const TapList& t, int si,
                        int Ho, int Wo, const float* wpacked, int Cout,
                        float* out, int out_bf16, int Hout, int Wout, int out_cstride, int out_coff, int so, int py, int px,
                        const Epilogue& e)
{
    int rc = ssie_make_conv(p, srcs, nsrc, N, Hv, Wv, t, si, Ho, Wo, wpacked, Cout, out, Hout, Wout, out_cstride, out_coff, so, py, px, e);
    if (rc) return rc;
    if (e.mask_y || e.accumulate || (si != 1 && si != 2)) return SSIE_E_ARG;
    for (int s = 0; s < nsrc; ++s) {
        if (srcs[s].C % 8 || srcs[s].cstride % 8 || srcs[s].coff % 8) return SSIE_E_SHAPE;
        if (nsrc > 1 && srcs[s].C % 32) return SSIE_E_SHAPE;
    }
    int mny, mxy, mnx, mxx; tap_extent(t, mny, mxy, mnx, mxx);
    p.nchunks = ssie_ceil_div(p.Cin, 32);
    p.th = si == 1 ? 16 : 8;
    // 16 x 32 tiles (conv_fprop_bf16w_kernel) under the same conditions as the fp32 wide kernel
    const int span = (mxy - mny) > (mxx - mnx) ? (mxy - mny) : (mxx - mnx);
    p.tw = (ssie_fprop_wide && si == 1 && span <= 2 && p.Cout_pad % 64 == 0 && Wo >= 32 &&
            (long)N * ssie_ceil_div(Ho, 16) * ssie_ceil_div(Wo, 32) * (p.Cout_pad / 64) >= ssie_fprop_wide_min_tiles) ? 32 : SSIE_TW;
    p.hp_h = (p.th - 1) * si + (mxy - mny) + 1;
    p.hp_w = (p.tw - 1) * si + (mxx - mnx) + 1;
    p.tiles_y = ssie_ceil_div(Ho, p.th);
    p.tiles_x = ssie_ceil_div(Wo, p.tw);
    p.out_bf16 = out_bf16;
    return 0;
}

// ---- Winograd F(2x2, 3x3) (conv_wino.hip) ----
int ssie_fprop_wino = 1;              // A/B switch: 1 = eligible stride-1 3x3 launches run conv_wino_kernel
#define SSIE_WINO_MIN_TILES 32        // 32-channel tiles.  128 (half a chip) until the kernel got its 16-channel-workgroup form for under-filled
                                      // launches (conv_wino.hip NH = 1): with it, at batch 2 of 128 x 128 (the reference's shipped configuration) the
                                      // 64- and 32-tile layers are faster on Winograd too (train64 at batch 2: 3.05 -> 2.95 ms; nothing changes at batch 32)
int ssie_fprop_wino_min_tiles = SSIE_WINO_MIN_TILES;  // ... when the launch has at least this many 16 x 32 x 32-channel tiles (tests set 1)
extern "C" void ssie_debug_set_wino(int v) { ssie_fprop_wino = v; }
extern "C" void ssie_debug_set_wino_min_tiles(int v) { ssie_fprop_wino_min_tiles = v < 0 ? SSIE_WINO_MIN_TILES : v; }   // v < 0: the default

int ssie_fprop_wino4 = 1;             // A/B switch: 1 = eligible launches run the F(4x4, 3x3) kernel (conv_wino4.hip) instead of F(2x2, 3x3)
#ifndef SSIE_WINO4_MIN_TILES
#define SSIE_WINO4_MIN_TILES (1 << 30)      /* off until it beats F(2x2,3x3) on the bench layers: tests and tools force it */
#endif
int ssie_fprop_wino4_min_tiles = SSIE_WINO4_MIN_TILES;   // ... when the launch has at least this many 16 x 64 x 32-channel tiles (tests set 1)
extern "C" void ssie_debug_set_wino4(int v) { ssie_fprop_wino4 = v; }
extern "C" void ssie_debug_set_wino4_min_tiles(int v) { ssie_fprop_wino4_min_tiles = v < 0 ? SSIE_WINO4_MIN_TILES : v; }   // v < 0: the default

// room for either Winograd form: F(2x2,3x3) = 16 transform positions in 16-channel chunks, F(4x4,3x3) = 36 in 8-channel steps
size_t ssie_wino_packed_floats(int K, int N)
{
    const int npad = N > 32 ? ssie_round_up(N, 64) : 32;
    const size_t f2 = (size_t)ssie_ceil_div(K, SSIE_CK) * 16 * 16 * npad, f4 = (size_t)ssie_ceil_div(K, 8) * 36 * 8 * npad;
    return f2 > f4 ? f2 : f4;
}

// a full 3 x 3 tap list with offsets in [-1, 1]^2 (forward or flipped data-gradient order)
static bool taps_are_3x3(const TapList& t)
{
    if (t.n != 9) return false;
    int seen = 0;
    for (int i = 0; i < 9; ++i) {
        if (t.dy[i] < -1 || t.dy[i] > 1 || t.dx[i] < -1 || t.dx[i] > 1) return false;
        seen |= 1 << ((t.dy[i] + 1) * 3 + t.dx[i] + 1);
    }
    return seen == 0x1ff;
}

// would a Winograd kernel take this launch?  (geometry from ssie_make_conv, its tap list)  0 = no, 1 = conv_wino_kernel F(2x2,3x3),
// 2 = conv_wino4_kernel F(4x4,3x3): sources at the launch's own resolution (no up-sampling on read), at most a quarter of the 64-wide
// tile columns wasted, slot offsets inside the DMA table's 24 bits, and enough 16 x 64 tiles to fill the chip
int ssie_wino_eligible(const ConvParams& p, const TapList& t)
{
    if (!taps_are_3x3(t) || p.si != 1 || p.so != 1 || p.py || p.px) return 0;
    if (p.out2_mode) return 0;          // the Winograd epilogues have no out2_mode forms (adding them cost the F(2x2) kernel 4 %: A/B'd)
    if (p.Ho != p.Hv || p.Wo != p.Wv || p.Hout != p.Ho || p.Wout != p.Wo) return 0;
    if (ssie_fprop_wino4) {
        bool ok = ssie_round_up(p.Wo, 64) * 3 <= p.Wo * 4;
        for (int s = 0; s < p.nsrc && ok; ++s)
            ok = p.src[s].sy == 1.f && p.src[s].sx == 1.f && p.src[s].Hs == p.Hv && p.src[s].Ws == p.Wv &&
                 (size_t)(18 * p.Wv + 66) * p.src[s].cstride * 4 < (1u << 24) && (size_t)p.Hv * p.Wv * p.src[s].cstride * 4 < (1u << 31);
        const long tiles4 = (long)p.N * ssie_ceil_div(p.Ho, 16) * ssie_ceil_div(p.Wo, 64) * (p.Cout_pad / 32);
        if (ok && tiles4 >= ssie_fprop_wino4_min_tiles) return 2;
    }
    if (!ssie_fprop_wino) return 0;
    const long tiles = (long)p.N * ssie_ceil_div(p.Ho, 16) * ssie_ceil_div(p.Wo, 32) * (p.Cout_pad / 32);
    return tiles >= ssie_fprop_wino_min_tiles ? 1 : 0;
}

PackDesc ssie_make_pack_wino(const float* w, float* dst, int K, int N, const TapList& t, int s_k, int s_n, int s_t, int kind)
{
    PackDesc d = ssie_make_pack(w, dst, K, N, t, s_k, s_n, s_t);
    d.wino = kind;
    if (kind == 2) d.nchunks = ssie_ceil_div(K, 8);
    for (int i = 0; i < 9; ++i) d.tapsel[(t.dy[i] + 1) * 3 + t.dx[i] + 1] = t.sel[i];
    return d;
}

// re-target a stride-1 3 x 3 geometry at conv_wino_kernel (16 x 32 tiles) / conv_wino4_kernel (16 x 64 tiles): 32-channel blocks, weights = U
void ssie_conv_to_wino(ConvParams& p, const float* u, int kind)
{
    p.wino = kind; p.wpacked = u;
    p.th = 16; p.tw = kind == 2 ? 64 : 32; p.hp_h = 18; p.hp_w = p.tw + 2;
    p.tiles_y = ssie_ceil_div(p.Ho, 16); p.tiles_x = ssie_ceil_div(p.Wo, p.tw);
    p.co_blocks = p.Cout_pad / 32;
}

PackDesc ssie_make_pack_bf16(const float* w, float* dst, int K, int N, const TapList& t, int s_k, int s_n, int s_t)
{
    PackDesc d = ssie_make_pack(w, dst, K, N, t, s_k, s_n, s_t);
    d.nchunks = ssie_ceil_div(K, 32); d.bf16 = 1;
    return d;
}

extern int ssie_wgrad_rows2, ssie_wgrad_sliding;
int ssie_wgrad_wino = 1;              // A/B switch: 1 = stride-1 3x3 weight gradients on Winograd F(3x3,2x2) (conv_wgrad_wino.hip)
#define SSIE_WGRAD_WINO_MIN_TILES 64   // (256 until round 4: at the reference's shipped batch of 2 the 64 x 64 layers' 128 tiles are faster here too - train64 at batch 2 -0.8 %)
int ssie_wgrad_wino_min_tiles = SSIE_WGRAD_WINO_MIN_TILES;  // ... when the launch has at least this many 8 x 16 position tiles (tests set 1)
extern "C" void ssie_debug_set_wgrad_wino(int v) { ssie_wgrad_wino = v; }
extern "C" void ssie_debug_set_wgrad_wino_min_tiles(int v) { ssie_wgrad_wino_min_tiles = v < 0 ? SSIE_WGRAD_WINO_MIN_TILES : v; }   // v < 0: the default
int ssie_make_wgrad(WgradParams& p, const SrcDesc& src, int N, int Hv, int Wv, int ci0_weight,
                    const float* g, int g_cstride, int g_coff, int Cout, int Ho, int Wo, int si,
                    const TapList& t, float* slabs, int target_wgs)
{
    memset(&p, 0, sizeof(p));
    if (src.C % 4 || src.cstride % 4 || src.coff % 4 || g_cstride % 4 || g_coff % 4) return SSIE_E_SHAPE;
    if (!ssie_fits_i32(N, src.Hs, src.Ws, src.cstride) || !ssie_fits_i32(N, Ho, Wo, g_cstride)) return SSIE_E_SHAPE;
    p.src = src; p.N = N; p.Hv = Hv; p.Wv = Wv; p.ci0_total = ci0_weight; p.Cin = src.C;
    p.g = g; p.g_cstride = g_cstride; p.g_coff = g_coff; p.Cout = Cout; p.Ho = Ho; p.Wo = Wo; p.si = si;
    p.ntaps = t.n;
    int mny, mxy, mnx, mxx; tap_extent(t, mny, mxy, mnx, mxx);
    p.min_dy = mny; p.min_dx = mnx;
    p.th = si == 1 ? 8 : 4;
    p.hp_h = (p.th - 1) * si + (mxy - mny) + 1;
    p.hp_w = (SSIE_TW - 1) * si + (mxx - mnx) + 1;
    for (int i = 0; i < t.n; ++i) { p.tap_dy[i] = t.dy[i]; p.tap_dx[i] = t.dx[i]; }
    const int cib = src.C > 32 ? 64 : 32, cob = Cout > 32 ? 64 : 32;
    p.ci_blocks = ssie_ceil_div(src.C, cib); p.co_blocks = ssie_ceil_div(Cout, cob);
    p.ci_pad = p.ci_blocks * cib; p.co_pad = p.co_blocks * cob;
    p.wsplit = 4 / ((cib / 32) * (cob / 32));
    p.tap_groups = ssie_ceil_div(t.n, SSIE_TG);
    // 9 x 9 with two wave pairs per workgroup (32 x 64 blocks): one wave pair per kernel ROW, two rows per workgroup
    p.rows2 = (ssie_wgrad_rows2 && ssie_wgrad_sliding && t.n == 81 && si == 1 && cib == 32 && cob == 64 && mxx - mnx == 8 && mxy - mny == 8) ? 1 : 0;
    if (p.rows2) p.tap_groups = ssie_ceil_div(p.tap_groups, 2);
    p.tiles_y = ssie_ceil_div(Ho, p.th); p.tiles_x = ssie_ceil_div(Wo, SSIE_TW);
    p.tiles_total = N * p.tiles_y * p.tiles_x;
    // Winograd F(3x3,2x2): full 3 x 3 in forward tap order, stride 1, same-size output (an up-sampled source: 64 x 64 blocks only); one workgroup per CU
    // (256 accumulator registers per wave), so half the slices of the direct kernel
    const bool up = src.sy != 1.f || src.sx != 1.f || src.Hs != Hv || src.Ws != Wv;
    p.wino = ssie_wgrad_wino && taps_are_3x3(t) && si == 1 && Ho == Hv && Wo == Wv && (!up || (cib == 64 && cob == 64)) &&
             p.tiles_total >= ssie_wgrad_wino_min_tiles;
    for (int i = 0; p.wino && i < 9; ++i) if (t.dy[i] != i / 3 - 1 || t.dx[i] != i % 3 - 1) p.wino = 0;
    if (p.wino) { p.tap_groups = 1; p.rows2 = 0; target_wgs = (target_wgs + 1) / 2; }
    int per = p.ci_blocks * p.co_blocks * p.tap_groups;
    int ns = target_wgs / per; if (ns < 1) ns = 1; if (ns > p.tiles_total) ns = p.tiles_total;
    p.nslices = ns;
    p.slabs = slabs;
    return 0;
}

size_t ssie_wgrad_slab_floats(const WgradParams& p) { return (size_t)p.nslices * p.ntaps * p.ci_pad * p.co_pad; }

// ---------------------------------------------------------------------------------------------
// granular C-ABI
// ---------------------------------------------------------------------------------------------
static const int kTargetWgs = 512;

extern "C" const char* ssie_version(void) { return "ssie-hip 0.1 (gfx950, fp32 MFMA 32x32x2)"; }

extern "C" int ssie_device_ok(void)
{
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

extern "C" size_t ssie_op_workspace_bytes(int cin, int cout, int k)
{
    const int T = k * k;
    size_t packed = 4 * ssie_packed_floats(cin > cout ? cin : cout, cin > cout ? cin : cout, T);
    // slabs: the direct kernels write kTargetWgs slices of one tap group, the Winograd weight gradient (3 x 3) kTargetWgs / 2
    // slices of nine taps; behind them the per-layer remainder
    const int tg = T < SSIE_TG ? T : SSIE_TG;
    size_t slabs = (size_t)kTargetWgs * tg * 64 * 64 + (size_t)T * ssie_round_up(cin, 64) * ssie_round_up(cout, 64);
    size_t partial = (size_t)256 * ssie_round_up(cout > cin ? cout : cin, 4);
    return (packed + slabs * 2 + partial) * sizeof(float) + 4096;
}

static inline float* ws_take(char*& cur, char* end, size_t floats)
{
    size_t bytes = (floats * 4 + 255) & ~(size_t)255;
    if (cur + bytes > end) return nullptr;
    float* r = (float*)cur; cur += bytes; return r;
}

// n zeroed tile-queue counters for the persistent fprop kernel
static inline int* take_counters(char*& cur, char* end, int n, hipStream_t st)
{
    int* c = (int*)ws_take(cur, end, 64);
    if (!c || n > 64) return nullptr;
    if (hipMemsetAsync(c, 0, 256, st) != hipSuccess) return nullptr;
    return c;
}

extern "C" int ssie_conv2d_fwd(const ssie_src_t* srcs, int nsrc, int N, int Hv, int Wv,
                               const float* weight, int cin_w, const float* bias, int cout, int k, int stride, int act,
                               const float* addsrc, float* out2, float* out, int out_cstride, int out_coff,
                               void* ws, size_t ws_bytes, void* stream)
{
    if (!srcs || !weight || !out || !ws) return SSIE_E_ARG;
    if (!(stride == 1 || (stride == 2 && k == 3)) || !(k & 1) || k > 9) return SSIE_E_SHAPE;
    SrcDesc sd[SSIE_MAX_SRC]; int cin = 0;
    if (nsrc < 1 || nsrc > SSIE_MAX_SRC) return SSIE_E_ARG;
    for (int s = 0; s < nsrc; ++s) { sd[s] = ssie_make_src(srcs[s].ptr, srcs[s].C, srcs[s].cstride, srcs[s].coff, srcs[s].Hs, srcs[s].Ws, Hv, Wv); cin += srcs[s].C; }
    const int pad = (k - 1) / 2, T = k * k;
    const int Ho = (Hv + 2 * pad - k) / stride + 1, Wo = (Wv + 2 * pad - k) / stride + 1;
    char* cur = (char*)ws; char* end = cur + ws_bytes;
    if (cin_w > cin || ssie_ceil_div(cin_w, SSIE_CK) != ssie_ceil_div(cin, SSIE_CK)) return SSIE_E_SHAPE;
    const size_t wf = ssie_packed_floats(cin_w, cout, T), uf = k == 3 ? ssie_wino_packed_floats(cin_w, cout) : 0;
    float* wp = ws_take(cur, end, wf > uf ? wf : uf);
    if (!wp) return SSIE_E_WORKSPACE;
    TapList t = ssie_taps_conv(k);
    hipStream_t st = (hipStream_t)stream;
    Epilogue e; memset(&e, 0, sizeof(e)); e.bias = bias; e.act = act; e.addsrc = addsrc; e.out2 = out2;
    ConvParams p;
    int rc = ssie_make_conv(p, sd, nsrc, N, Hv, Wv, t, stride, Ho, Wo, wp, cout, out, Ho, Wo, out_cstride, out_coff, 1, 0, 0, e);
    if (rc) return rc;
    const int wino = ssie_wino_eligible(p, t);
    PackDesc pd = wino ? ssie_make_pack_wino(weight, wp, cin_w, cout, t, /*s_k*/ T, /*s_n*/ cin_w * T, 1, wino)
                       : ssie_make_pack(weight, wp, cin_w, cout, t, /*s_k*/ T, /*s_n*/ cin_w * T, 1);
    if (ssie_launch_pack(pd, st)) return SSIE_E_LAUNCH;
    if (wino) ssie_conv_to_wino(p, wp, wino);
    if (!(p.tile_counter = take_counters(cur, end, 1, st))) return SSIE_E_WORKSPACE;
    return ssie_launch_fprop(p, st) ? SSIE_E_LAUNCH : 0;
}

// shared by ConvTranspose2d fprop and the data gradient of a stride-2 conv:
// out[2a+py][2b+px][n] = sum_taps in[a+dy][b+dx][k] * W(k, n, tap)
static int transposed_like(const SrcDesc& in, int N, int Hin, int Win, int Kc, int Nc, const float* weight,
                           int s_k, int s_n, float* out, int Hout, int Wout, int out_cstride, int out_coff,
                           const Epilogue& e, char*& cur, char* end, hipStream_t st)
{
    int* counters = take_counters(cur, end, 4, st);
    if (!counters) return SSIE_E_WORKSPACE;
    if (ssie_tconv_eligible(in, N, Hin, Win, Nc)) {
        // all four output-parity classes in one launch
        TapList t = ssie_taps_transposed_all();
        float* wp = ws_take(cur, end, ssie_packed_floats(Kc, Nc, t.n));
        if (!wp) return SSIE_E_WORKSPACE;
        PackDesc pd = ssie_make_pack(weight, wp, Kc, Nc, t, s_k, s_n, 1);
        if (ssie_launch_pack(pd, st)) return SSIE_E_LAUNCH;
        ConvParams p;
        int rc = ssie_make_conv(p, &in, 1, N, Hin, Win, t, 1, Hin, Win, wp, Nc, out, Hout, Wout, out_cstride, out_coff, 2, 0, 0, e);
        if (rc) return rc;
        ssie_conv_to_tconv(p);
        p.tile_counter = counters;
        return ssie_launch_fprop(p, st) ? SSIE_E_LAUNCH : 0;
    }
    for (int py = 0; py < 2; ++py) for (int px = 0; px < 2; ++px) {
        TapList t = ssie_taps_transposed(3, 1, py, px);
        float* wp = ws_take(cur, end, ssie_packed_floats(Kc, Nc, t.n));
        if (!wp) return SSIE_E_WORKSPACE;
        PackDesc pd = ssie_make_pack(weight, wp, Kc, Nc, t, s_k, s_n, 1);
        if (ssie_launch_pack(pd, st)) return SSIE_E_LAUNCH;
        ConvParams p;
        const int Ho = ssie_ceil_div(Hout - py, 2), Wo = ssie_ceil_div(Wout - px, 2);
        int rc = ssie_make_conv(p, &in, 1, N, Hin, Win, t, 1, Ho, Wo, wp, Nc, out, Hout, Wout, out_cstride, out_coff, 2, py, px, e);
        if (rc) return rc;
        p.tile_counter = counters + py * 2 + px;
        if (ssie_launch_fprop(p, st)) return SSIE_E_LAUNCH;
    }
    return 0;
}

extern "C" int ssie_conv_transpose2d_fwd(const ssie_src_t* src, int N, const float* weight, const float* bias, int cout,
                                         int act, float* out, int out_cstride, int out_coff,
                                         void* ws, size_t ws_bytes, void* stream)
{
    if (!src || !weight || !out || !ws) return SSIE_E_ARG;
    SrcDesc in = ssie_make_src(src->ptr, src->C, src->cstride, src->coff, src->Hs, src->Ws, src->Hs, src->Ws);
    char* cur = (char*)ws; char* end = cur + ws_bytes;
    Epilogue e; memset(&e, 0, sizeof(e)); e.bias = bias; e.act = act;
    // weight (in, out, 3, 3): k = ci -> stride cout*9, n = co -> stride 9
    return transposed_like(in, N, src->Hs, src->Ws, src->C, cout, weight, cout * 9, 9, out, 2 * src->Hs, 2 * src->Ws,
                           out_cstride, out_coff, e, cur, end, (hipStream_t)stream);
}

extern "C" int ssie_conv2d_dgrad(const float* g, int g_cstride, int g_coff, int N, int Ho, int Wo, int cout,
                                 const float* weight, int cin_total, int ci_off, int cs, int k, int stride,
                                 float* gx, int Hin, int Win, int gx_cstride, int gx_coff,
                                 const float* mask_y, int mask_mode, int accumulate,
                                 void* ws, size_t ws_bytes, void* stream)
{
    if (!g || !weight || !gx || !ws) return SSIE_E_ARG;
    if (!(stride == 1 || (stride == 2 && k == 3))) return SSIE_E_SHAPE;
    const int T = k * k;
    hipStream_t st = (hipStream_t)stream;
    char* cur = (char*)ws; char* end = cur + ws_bytes;
    SrcDesc in = ssie_make_src(g, ssie_round_up(cout, 4), g_cstride, g_coff, Ho, Wo, Ho, Wo);
    Epilogue e; memset(&e, 0, sizeof(e)); e.mask_y = mask_y; e.mask_mode = mask_mode; e.accumulate = accumulate;
    const float* wbase = weight + (size_t)ci_off * T;
    if (stride == 1) {
        TapList t = ssie_taps_dgrad_s1(k);
        const size_t wf = ssie_packed_floats(cout, cs, T), uf = k == 3 ? ssie_wino_packed_floats(cout, cs) : 0;
        float* wp = ws_take(cur, end, wf > uf ? wf : uf);
        if (!wp) return SSIE_E_WORKSPACE;
        ConvParams p;
        int rc = ssie_make_conv(p, &in, 1, N, Ho, Wo, t, 1, Hin, Win, wp, cs, gx, Hin, Win, gx_cstride, gx_coff, 1, 0, 0, e);
        if (rc) return rc;
        // OIHW: k = co -> stride cin_total*T, n = ci -> stride T
        const int wino = ssie_wino_eligible(p, t);
        PackDesc pd = wino ? ssie_make_pack_wino(wbase, wp, cout, cs, t, cin_total * T, T, 1, wino) : ssie_make_pack(wbase, wp, cout, cs, t, cin_total * T, T, 1);
        if (ssie_launch_pack(pd, st)) return SSIE_E_LAUNCH;
        if (wino) ssie_conv_to_wino(p, wp, wino);
        if (!(p.tile_counter = take_counters(cur, end, 1, st))) return SSIE_E_WORKSPACE;
        return ssie_launch_fprop(p, st) ? SSIE_E_LAUNCH : 0;
    }
    return transposed_like(in, N, Ho, Wo, cout, cs, wbase, cin_total * T, T, gx, Hin, Win, gx_cstride, gx_coff, e, cur, end, st);
}

extern "C" int ssie_conv_transpose2d_dgrad(const float* g, int g_cstride, int g_coff, int N, int Hin, int Win, int cout,
                                           const float* weight, int cin,
                                           float* gx, int gx_cstride, int gx_coff,
                                           const float* mask_y, int mask_mode, int accumulate,
                                           void* ws, size_t ws_bytes, void* stream)
{
    if (!g || !weight || !gx || !ws) return SSIE_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    char* cur = (char*)ws; char* end = cur + ws_bytes;
    // gX[a][ci] = sum_{kh,co} G[2a-1+kh][co] W[ci][co][kh]  == stride-2 conv of G with W read as OIHW (O=ci, I=co)
    SrcDesc in = ssie_make_src(g, ssie_round_up(cout, 4), g_cstride, g_coff, 2 * Hin, 2 * Win, 2 * Hin, 2 * Win);
    TapList t = ssie_taps_conv(3);
    float* wp = ws_take(cur, end, ssie_packed_floats(cout, cin, 9));
    if (!wp) return SSIE_E_WORKSPACE;
    PackDesc pd = ssie_make_pack(weight, wp, cout, cin, t, /*k=co*/ 9, /*n=ci*/ cout * 9, 1);
    if (ssie_launch_pack(pd, st)) return SSIE_E_LAUNCH;
    Epilogue e; memset(&e, 0, sizeof(e)); e.mask_y = mask_y; e.mask_mode = mask_mode; e.accumulate = accumulate;
    ConvParams p;
    int rc = ssie_make_conv(p, &in, 1, N, 2 * Hin, 2 * Win, t, 2, Hin, Win, wp, cin, gx, Hin, Win, gx_cstride, gx_coff, 1, 0, 0, e);
    if (rc) return rc;
    if (!(p.tile_counter = take_counters(cur, end, 1, st))) return SSIE_E_WORKSPACE;
    return ssie_launch_fprop(p, st) ? SSIE_E_LAUNCH : 0;
}

int ssie_run_wgrad(const SrcDesc& x, int x_creal, int N, int Hv, int Wv, const float* g, int g_cstride, int g_coff, int gC,
                   int Ho, int Wo, int si, const TapList& t, float* dw, long s_co, long s_ci, long s_t, float* db,
                   int accumulate, float* slabs, size_t slab_cap_floats, hipStream_t st)
{
    WgradParams p;
    int rc = ssie_make_wgrad(p, x, N, Hv, Wv, 0, g, g_cstride, g_coff, gC, Ho, Wo, si, t, slabs, kTargetWgs);
    if (rc) return rc;
    const size_t need = ssie_wgrad_slab_floats(p);
    const size_t bneed = db ? (size_t)p.nslices * p.co_pad : 0;
    if (need + bneed > slab_cap_floats) return SSIE_E_WORKSPACE;   // checked BEFORE anything is enqueued
    p.bias_slabs = db ? slabs + need : nullptr;
    if (ssie_launch_wgrad(p, st)) return SSIE_E_LAUNCH;
    if (ssie_launch_wgrad_reduce(slabs, p.nslices, p.ntaps, p.ci_pad, p.co_pad, x_creal, gC, dw, s_co, s_ci, s_t,
                                 p.bias_slabs, db, accumulate, st)) return SSIE_E_LAUNCH;
    return 0;
}

extern "C" int ssie_conv2d_wgrad(const ssie_src_t* src, int N, int Hv, int Wv,
                                 const float* g, int g_cstride, int g_coff, int cout, int k, int stride,
                                 int cin_total, int ci_off, float* dw, float* db, int accumulate,
                                 void* ws, size_t ws_bytes, void* stream)
{
    if (!src || !g || !dw || !ws) return SSIE_E_ARG;
    if (!(stride == 1 || (stride == 2 && k == 3))) return SSIE_E_SHAPE;
    const int pad = (k - 1) / 2, T = k * k;
    const int Ho = (Hv + 2 * pad - k) / stride + 1, Wo = (Wv + 2 * pad - k) / stride + 1;
    hipStream_t st = (hipStream_t)stream;
    char* cur = (char*)ws; char* end = cur + ws_bytes;
    float* partial = ws_take(cur, end, (size_t)256 * cout);
    if (!partial) return SSIE_E_WORKSPACE;
    size_t cap = (size_t)(end - cur) / 4;
    SrcDesc x = ssie_make_src(src->ptr, src->C, src->cstride, src->coff, src->Hs, src->Ws, Hv, Wv);
    TapList t = ssie_taps_conv(k);
    int creal = src->C < cin_total - ci_off ? src->C : cin_total - ci_off;
    int rc = ssie_run_wgrad(x, creal, N, Hv, Wv, g, g_cstride, g_coff, cout, Ho, Wo, stride, t,
                            dw + (size_t)ci_off * T, (long)cin_total * T, T, 1, db, accumulate, (float*)cur, cap, st);
    return rc;
}

extern "C" int ssie_conv_transpose2d_wgrad(const ssie_src_t* x, int N, const float* g, int g_cstride, int g_coff, int cout,
                                           float* dw, float* db, int accumulate,
                                           void* ws, size_t ws_bytes, void* stream)
{
    if (!x || !g || !dw || !ws) return SSIE_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    char* cur = (char*)ws; char* end = cur + ws_bytes;
    float* partial = ws_take(cur, end, (size_t)256 * cout);
    if (!partial) return SSIE_E_WORKSPACE;
    size_t cap = (size_t)(end - cur) / 4;
    const int Hin = x->Hs, Win = x->Ws, cin = x->C;
    // dW[ci][co][kh][kw] = sum_{a,b} x[a][b][ci] * G[2a-1+kh][2b-1+kw][co]: a stride-2 wgrad with the roles
    // swapped: "input" = G (hi-res, cout channels), "output gradient" = x (lo-res, cin channels)
    SrcDesc gs = ssie_make_src(g, ssie_round_up(cout, 4), g_cstride, g_coff, 2 * Hin, 2 * Win, 2 * Hin, 2 * Win);
    TapList t = ssie_taps_conv(3);
    // slab [t][ci' = co][co' = ci]  ->  dw[(ci*cout + co)*9 + t]
    int rc = ssie_run_wgrad(gs, cout, N, 2 * Hin, 2 * Win, x->ptr, x->cstride, x->coff, cin, Hin, Win, 2, t,
                            dw, /*s_co (co'=ci)*/ (long)cout * 9, /*s_ci (ci'=co)*/ 9, 1, nullptr, accumulate, (float*)cur, cap, st);
    if (rc) return rc;
    if (db && ssie_launch_colsum(g, (long)N * 4 * Hin * Win, g_cstride, g_coff, cout, partial, 256, db, accumulate, st)) return SSIE_E_LAUNCH;
    return 0;
}
