// Device helpers shared by the convolution kernels (virtual-input addressing, LDS swizzle).
#pragma once
#include "ssie_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int ssie_swz(int hp) { return (hp >> 2) & 3; }

struct SrcSel {
    const float* ptr; int C, cstride, coff, Hs, Ws; float sy, sx; int cbeg;
};

// Picking the source of a chunk must not index the by-value kernarg struct at run time (hipcc then spills
// the whole 470-byte struct to scratch) and must not select between loads either (instcombine folds that
// back into a load of a selected ADDRESS).  So the three descriptors are blended arithmetically:
// v = v0 + m1*(v1-v0) + m2*(v2-v1), m1 = [which >= 1], m2 = [which == 2] - a few scalar integer ops per chunk.
template <typename PT>
__device__ __forceinline__ SrcSel ssie_pick_src(const PT& p, int c_first)
{
    const int c1 = p.src[0].C, c2 = p.src[0].C + p.src[1].C;
    const int m1 = (p.nsrc > 1 && c_first >= c1) ? 1 : 0;
    const int m2 = (p.nsrc > 2 && c_first >= c2) ? 1 : 0;
    auto bl = [&](int v0, int v1, int v2) { return v0 + m1 * (v1 - v0) + m2 * (v2 - v1); };
    SrcSel r;
    const long long q0 = (long long)p.src[0].ptr, q1 = (long long)p.src[1].ptr, q2 = (long long)p.src[2].ptr;
    r.ptr = (const float*)(q0 + m1 * (q1 - q0) + m2 * (q2 - q1));
    r.C = bl(p.src[0].C, p.src[1].C, p.src[2].C);
    r.cstride = bl(p.src[0].cstride, p.src[1].cstride, p.src[2].cstride);
    r.coff = bl(p.src[0].coff, p.src[1].coff, p.src[2].coff);
    r.Hs = bl(p.src[0].Hs, p.src[1].Hs, p.src[2].Hs);
    r.Ws = bl(p.src[0].Ws, p.src[1].Ws, p.src[2].Ws);
    r.sy = __int_as_float(bl(__float_as_int(p.src[0].sy), __float_as_int(p.src[1].sy), __float_as_int(p.src[2].sy)));
    r.sx = __int_as_float(bl(__float_as_int(p.src[0].sx), __float_as_int(p.src[1].sx), __float_as_int(p.src[2].sx)));
    r.cbeg = m1 * c1 + m2 * (c2 - c1);
    return r;
}

// single-source layers: the descriptor of source 0
template <typename PT>
__device__ __forceinline__ SrcSel ssie_only_src(const PT& p)
{
    SrcSel r;
    r.ptr = p.src[0].ptr; r.C = p.src[0].C; r.cstride = p.src[0].cstride; r.coff = p.src[0].coff;
    r.Hs = p.src[0].Hs; r.Ws = p.src[0].Ws; r.sy = p.src[0].sy; r.sx = p.src[0].sx; r.cbeg = 0;
    return r;
}

// load 4 consecutive channels of virtual pixel (n, vy, vx); zero outside the image / channel range.
// Branch-free: the address is clamped into the tensor and the value is zeroed afterwards, so a batch of these
// compiles to back-to-back global_load_dwordx4 without exec-mask regions (hipcc serialises predicated loads
// with s_waitcnt vmcnt(0) between them).
__device__ __forceinline__ f32x4 ssie_load_virtual(const SrcSel& s, int n, int vy, int vx, int Hv, int Wv, int c)
{
    const bool ok = vy >= 0 && vy < Hv && vx >= 0 && vx < Wv && c < s.C;
    const int cy = min(max(vy, 0), Hv - 1), cx = min(max(vx, 0), Wv - 1), cc = min(c, s.C - 4);
    const int y = min((int)floorf((float)cy * s.sy), s.Hs - 1);
    const int x = min((int)floorf((float)cx * s.sx), s.Ws - 1);
    f32x4 v = *(const f32x4*)(s.ptr + ((size_t)(n * s.Hs + y) * s.Ws + x) * s.cstride + s.coff + cc);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return ok ? v : z;
}


// Epilogue of one 32 x 32 accumulator tile whose 2 x 16 output positions all lie inside the output (the caller checked).
// C/D layout: register r of a lane holds tile row r>>3, tile column (r&3) + 8*((r>>2)&1) (+4h, already in o0); every
// element offset is a compile-time multiple of two run-time strides.  The fused extras (activation-derivative mask,
// second output, residual add, accumulate) are applied in the same order as the general path, but each one loads its
// 16 operands back-to-back BEFORE using them: a load -> use -> store chain per element exposes one HBM latency per
// element (16-32 per tile), which is what made the masked data-gradient launches ~40 % slower than the plain ones.
#define SSIE_EOFF(r) ((long)((r) >> 3) * rowstride + (long)(((r) & 3) + 8 * (((r) >> 2) & 1)) * pixstride)
template <int EPI = 0, typename PT>
__device__ __forceinline__ void ssie_epilogue_full(const PT& p, const f32x16& acc, size_t o0, long rowstride, long pixstride, float bv)
{
    float v[16];
    if (EPI == 1) {            // plain forward layer (ssie_epi_shape): bias + ReLU / nothing
        if (p.act == ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaxf(acc[r] + bv, 0.f);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[r] + bv;
        }
        float* ob = p.out + o0;
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[SSIE_EOFF(r)] = v[r];
        return;
    }
    if (EPI == 2) {            // plain data gradient: optional ReLU mask, optional accumulate
        float* ob = p.out + o0;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[r];
        if (p.mask_mode != MASK_NONE) {
            const float* mp = p.mask_y + o0;
            float y[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = mp[SSIE_EOFF(r)];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = y[r] > 0.f ? v[r] : 0.f;
        }
        if (p.accumulate) {
            float a[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = ob[SSIE_EOFF(r)];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += a[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[SSIE_EOFF(r)] = v[r];
        return;
    }
    if (p.act == ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(acc[r] + bv, 0.f);
    } else if (p.act == ACT_SIGMOID) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = 1.f / (1.f + expf(-(acc[r] + bv)));
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[r] + bv;
    }
    if (p.out2_mode) {
        // backward forms (ConvParams.out2_mode; no activation, no skip add): one launch writes a gradient AND its masked copy
        const float* mp = p.mask_y + o0;
        float y[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = mp[SSIE_EOFF(r)];
        float* ob = p.out + o0; float* o2 = p.out2 + o0;
        if (p.accumulate) {
            float a[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = ob[SSIE_EOFF(r)];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += a[r];
        }
        float m[16];
        if (p.mask_mode == MASK_RELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) m[r] = y[r] > 0.f ? v[r] : 0.f;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) m[r] = v[r] * (y[r] * (1.f - y[r]));
        }
        // mode 1: out = mask * v, out2 = v;  mode 2: out = total, out2 = mask * total
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[SSIE_EOFF(r)] = p.out2_mode == 1 ? m[r] : v[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) o2[SSIE_EOFF(r)] = p.out2_mode == 1 ? v[r] : m[r];
        return;
    }
    if (p.mask_mode != MASK_NONE) {
        const float* mp = p.mask_y + o0;
        float y[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = mp[SSIE_EOFF(r)];
        if (p.mask_mode == MASK_RELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = y[r] > 0.f ? v[r] : 0.f;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] *= y[r] * (1.f - y[r]);
        }
    }
    if (p.out2) {
        float* o2 = p.out2 + o0;
#pragma unroll
        for (int r = 0; r < 16; ++r) o2[SSIE_EOFF(r)] = v[r];
    }
    if (p.addsrc) {
        const float* ap = p.addsrc + o0;
        float a[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = ap[SSIE_EOFF(r)];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += a[r];
    }
    float* ob = p.out + o0;
    if (p.accumulate) {
        float a[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = ob[SSIE_EOFF(r)];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += a[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) ob[SSIE_EOFF(r)] = v[r];
}

// one output element through the whole fused epilogue (edge tiles, element by element): v = accumulator + bias
template <typename PT>
__device__ __forceinline__ void ssie_epilogue_elem(const PT& p, size_t o, float v)
{
    if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
    else if (p.act == ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
    if (p.out2_mode) {                       // backward forms, see ssie_epilogue_full
        const float y = p.mask_y[o];
        if (p.accumulate) v += p.out[o];
        const float m = p.mask_mode == MASK_RELU ? (y > 0.f ? v : 0.f) : v * (y * (1.f - y));
        p.out[o] = p.out2_mode == 1 ? m : v;
        p.out2[o] = p.out2_mode == 1 ? v : m;
        return;
    }
    if (p.mask_mode == MASK_RELU) v = p.mask_y[o] > 0.f ? v : 0.f;
    else if (p.mask_mode == MASK_SIGMOID) { float y = p.mask_y[o]; v *= y * (1.f - y); }
    if (p.out2) p.out2[o] = v;
    if (p.addsrc) v += p.addsrc[o];
    if (p.accumulate) v += p.out[o];
    p.out[o] = v;
}

// General (edge-tile) variant of the above: per-element bounds checks; arow / bcol = output-grid position of tile
// element (row 0, column 4h).
template <typename PT>
__device__ __forceinline__ void ssie_epilogue_ragged(const PT& p, const f32x16& acc, size_t o0, long rowstride, long pixstride,
                                                     float bv, int arow, int bcol)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int tr = r >> 3, tc = (r & 3) + 8 * ((r >> 2) & 1);
        const int a = arow + tr, b = bcol + tc;
        if (a >= p.Ho || b >= p.Wo) continue;
        if (a * p.so + p.py >= p.Hout || b * p.so + p.px >= p.Wout) continue;
        const size_t o = o0 + tr * rowstride + tc * pixstride;
        ssie_epilogue_elem(p, o, acc[r] + bv);
    }
}

// Transposed accumulator tiles (bf16 kernels: MFMAs issued as D^T = W x X, lane = output position li of the 2 x 16 M-tile):
// output-pixel element offset (channel out_coff) and validity of this lane's position; arow / bcol = first tile row / column
template <typename PT>
__device__ __forceinline__ size_t ssie_epilogue_pos(const PT& p, int n, int arow, int bcol, int li, bool& ok)
{
    const int a = arow + (li >> 4), b = bcol + (li & 15);
    const int oy = a * p.so + p.py, ox = b * p.so + p.px;
    ok = a < p.Ho && b < p.Wo && oy < p.Hout && ox < p.Wout;
    return ((size_t)(n * p.Hout + oy) * p.Wout + ox) * p.out_cstride + p.out_coff;
}
