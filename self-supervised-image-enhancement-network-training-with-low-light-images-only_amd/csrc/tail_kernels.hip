// Fused tail of the enhance-only forward (inference entry, /root/reference/model.py:168-175 and :233):
//   gather = cat[up(d1), up(d2), d3] -> feature_fusion (1x1, 192 -> 64, no activation) -> final_conv (3x3, 64 -> 1) -> I_delta
//   S = R_low * I_delta + R_low * I_low
// Both convolutions are linear with nothing in between, so I_delta is ONE 3x3 convolution of the 192-channel gather with the
// composite weights Wc[tap][k] = sum_j W_final[j][tap] * W_fusion[j][k].  The 64-channel tensor `f` (written and read once per
// pixel: 2 x 128 MB per 1024 x 1024 bf16 image) never exists, and a skinny 192 -> 1 contraction is VALU work, not a GEMM tile.
// The one subtlety is the border: final_conv zero-pads f, and f = W_fusion*gather + b_fusion is NOT zero where gather is, so
// the fusion bias enters per tap and only for taps that fall inside the image:
//   I_delta[p] = b_final + sum_{tap: p+tap inside} ( cb[tap] + Wc[tap] . gather[p+tap] ),   cb[tap] = W_final[:, tap] . b_fusion
// Nearest up-sampling is resolved on read exactly as in the convolution kernels (src = min(floor(dst * in/out), in - 1)):
// the dot products T_s[q][tap] = Wc[tap][source s] . d_s[q] are taken once per SOURCE pixel q of each pyramid level (phase 1,
// eight lanes per pixel, 16 bytes per lane), and an output pixel sums 27 of them (phase 2).  Phase 3 forms S with the
// band axis on the lanes.  Only the training forward keeps the two layers apart (their weight gradients need f).
#include "loss_kernels.h"

// Composite-weight buffer (floats): Wc [3][9][64] | cb [9] | b_final | pad to 1744 | the bf16 B-operand images of the MFMA phase:
// uint4 index ((source * 2 + k-half) * 2 + hi/lo) * 64 + lane, lane (tap = lane & 15, k-group = lane >> 4) holding channels
// 32 k-half + 8 k-group + 0..7 of its tap (taps 9 - 15: zero).  hi = bf16(w), lo = bf16(w - hi): two MFMAs per operand keep the
// weights at ~16 mantissa bits (the fp32 tail multiplies fp32 weights), the activations are bf16 either way.
#define TAILW_PACK_OFF 1744
#define TAILW_FLOATS (TAILW_PACK_OFF + 3 * 2 * 2 * 64 * 4)
size_t ssie_tail_weight_floats() { return TAILW_FLOATS; }

namespace {
const int TT = 16;                       // output tile edge
const int R3 = TT + 2;                   // full-resolution pixels per tile edge incl. the 3x3 halo
const int R2MAX = 12, R1MAX = 8;         // source pixels per edge of the 1/2 and 1/4 level (exact x2 / x4: 10 and 6)
}

struct TailParams {
    const void* d1; const void* d2; const void* d3;     // NHWC, 64 channels per pixel (cstride 64), bf16 or fp32
    int H, W, H2, W2, H4, W4, N;
    float sy2, sx2, sy4, sx4;                            // nearest scales in / out (1.0f = same size)
    const float* wc;                                     // [3 sources][9 taps][64] composite weights, then cb[9], then b_final
    const float* RL; int rl_cs;                          // R_low | I_low (fp32)
    float* D; int d_cs;                                  // I_delta out
    float* S; int s_cs;                                  // enhanced cube out
    int B;
};

// Wc / cb / b_final from the two layers' parameters (re-made every forward: the weights may have been stepped)
__global__ void tail_weights_kernel(const float* __restrict__ wf, const float* __restrict__ bf, const float* __restrict__ wl,
                                    const float* __restrict__ bl, float* __restrict__ out)
{
    // wf: feature_fusion weight (64, 192, 1, 1), bf: its bias (64); wl: final_conv weight (1, 64, 3, 3), bl: its bias (1)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * 9 * 64) {
        const int s = i / (9 * 64), tap = (i / 64) % 9, k = i % 64;
        float a = 0.f;
        for (int j = 0; j < 64; ++j) a += wl[j * 9 + tap] * wf[j * 192 + s * 64 + k];
        out[i] = a;
    } else if (i < 3 * 9 * 64 + 9) {
        const int tap = i - 3 * 9 * 64;
        float a = 0.f;
        for (int j = 0; j < 64; ++j) a += wl[j * 9 + tap] * bf[j];
        out[i] = a;
    } else if (i == 3 * 9 * 64 + 9) out[i] = bl[0];
    else if (i >= TAILW_PACK_OFF && i < TAILW_PACK_OFF + 3 * 2 * 64 * 8) {
        // one element of the B-operand images: (source s, k-half kk, lane, e)
        const int j0 = i - TAILW_PACK_OFF, e = j0 & 7, lane = (j0 >> 3) & 63, kk = (j0 >> 9) & 1, s = j0 >> 10;
        const int tap = lane & 15, k = 32 * kk + 8 * (lane >> 4) + e;
        float a = 0.f;
        if (tap < 9)
            for (int j = 0; j < 64; ++j) a += wl[j * 9 + tap] * wf[j * 192 + s * 64 + k];
        const unsigned hi = ssie_f2bf(a);
        const unsigned lo = ssie_f2bf(a - __uint_as_float(hi << 16));
        unsigned short* img = (unsigned short*)(out + TAILW_PACK_OFF);
        img[(((s * 2 + kk) * 2 + 0) * 64 + lane) * 8 + e] = (unsigned short)hi;
        img[(((s * 2 + kk) * 2 + 1) * 64 + lane) * 8 + e] = (unsigned short)lo;
    }
}

__device__ __forceinline__ int nearest_src(int v, float s, int n) { return min((int)floorf((float)v * s), n - 1); }

// 8 consecutive channels of a pixel as they sit in memory (16 B bf16 / 32 B fp32); the conversion is a separate step so that
// several pixels' loads can be in flight before the first one is used
template <bool BF16> struct Raw8 { uint4 a, b; };
template <bool BF16>
__device__ __forceinline__ Raw8<BF16> load8_raw(const void* base, size_t pix, int c8)
{
    Raw8<BF16> r;
    if (BF16) { r.a = *(const uint4*)((const unsigned short*)base + pix * 64 + c8); r.b = r.a; }
    else { r.a = *(const uint4*)((const float*)base + pix * 64 + c8); r.b = *(const uint4*)((const float*)base + pix * 64 + c8 + 4); }
    return r;
}
template <bool BF16>
__device__ __forceinline__ void unpack8(const Raw8<BF16>& r, float* v)
{
    if (BF16) {
        const unsigned w[4] = {r.a.x, r.a.y, r.a.z, r.a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    } else {
        v[0] = __uint_as_float(r.a.x); v[1] = __uint_as_float(r.a.y); v[2] = __uint_as_float(r.a.z); v[3] = __uint_as_float(r.a.w);
        v[4] = __uint_as_float(r.b.x); v[5] = __uint_as_float(r.b.y); v[6] = __uint_as_float(r.b.z); v[7] = __uint_as_float(r.b.w);
    }
}

typedef __bf16 tail_bf16x8 __attribute__((ext_vector_type(8)));
#define TAIL_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tail_bf16x8, (a)), __builtin_bit_cast(tail_bf16x8, (b)), (c), 0, 0, 0)

template <bool BF16>
__global__ __launch_bounds__(256) void tail_kernel(const TailParams p)
{
    __shared__ __attribute__((aligned(16))) float wcs[3 * 9 * 64 + 16];
    __shared__ float T3[R3 * R3 * 9], T2[R2MAX * R2MAX * 9], T1[R1MAX * R1MAX * 9];
    __shared__ float Dt[TT * TT];
    const int tid = threadIdx.x;
    const int tiles_x = (p.W + TT - 1) / TT, tiles_y = (p.H + TT - 1) / TT;
    const int n = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x % (tiles_x * tiles_y);
    const int y0 = (tr / tiles_x) * TT, x0 = (tr % tiles_x) * TT;
    for (int i = tid + (BF16 ? 3 * 9 * 64 : 0); i < 3 * 9 * 64 + 10; i += 256) wcs[i] = p.wc[i];     // (bf16: only cb / b_final - the weights are MFMA operands)
    // source-pixel windows of the two coarse levels that the (clamped) full-resolution window maps to
    const int vy_lo = max(y0 - 1, 0), vy_hi = min(y0 + TT, p.H - 1), vx_lo = max(x0 - 1, 0), vx_hi = min(x0 + TT, p.W - 1);
    const int y2lo = nearest_src(vy_lo, p.sy2, p.H2), x2lo = nearest_src(vx_lo, p.sx2, p.W2);
    const int n2y = nearest_src(vy_hi, p.sy2, p.H2) - y2lo + 1, n2x = nearest_src(vx_hi, p.sx2, p.W2) - x2lo + 1;
    const int y1lo = nearest_src(vy_lo, p.sy4, p.H4), x1lo = nearest_src(vx_lo, p.sx4, p.W4);
    const int n1y = nearest_src(vy_hi, p.sy4, p.H4) - y1lo + 1, n1x = nearest_src(vx_hi, p.sx4, p.W4) - x1lo + 1;
    __syncthreads();

    // ---- phase 1: nine dot products per source pixel, eight lanes per pixel (8 channels = 16 B bf16 / 32 B fp32 each).
    // One source level at a time, so that this lane's 9 x 8 composite weights of the level live in registers (72 VGPRs)
    // instead of being re-read from LDS for every pixel (18 ds_read_b128 per pixel and lane). ----
    const int sub = tid & 7, slot = tid >> 3;
    const int n3 = R3 * R3, n2 = n2y * n2x, n1 = n1y * n1x;
    if (BF16) {
        // bf16 sources: the nine dot products of 16 source pixels are two v_mfma_f32_16x16x32_bf16 (x 2 for the hi / lo weight split):
        // A = the pixels' 64 channels straight from global memory (lane (pixel = lane & 15, k-group = lane >> 4): 16 bytes of its
        // pixel), B = the weight images, D = lane (tap = lane & 15) holds pixels 4 (lane >> 4) + 0..3.  The VALU form below cost
        // ~130 vector instructions per 8 pixels and wave (72 FMAs, 27 cross-lane adds): the kernel was VALU-bound at 2.3 TB/s.
        const int wave = tid >> 6, lane = tid & 63, li = lane & 15, kg = lane >> 4;
        const uint4* wimg = (const uint4*)(p.wc + TAILW_PACK_OFF);
        constexpr int G = 6;                                        // 16-pixel groups per wave in flight (21 groups / 4 waves at full resolution)
#pragma unroll 1
        for (int lvl = 0; lvl < 3; ++lvl) {
            const int items = lvl == 0 ? n1 : lvl == 1 ? n2 : n3;
            const unsigned short* base = (const unsigned short*)(lvl == 0 ? p.d1 : lvl == 1 ? p.d2 : p.d3);
            float* Tl = lvl == 0 ? T1 : lvl == 1 ? T2 : T3;
            const uint4 bh0 = wimg[((lvl * 2 + 0) * 2 + 0) * 64 + lane], bl0 = wimg[((lvl * 2 + 0) * 2 + 1) * 64 + lane];
            const uint4 bh1 = wimg[((lvl * 2 + 1) * 2 + 0) * 64 + lane], bl1 = wimg[((lvl * 2 + 1) * 2 + 1) * 64 + lane];
            const int ngroups = (items + 15) >> 4;
            for (int g0 = wave; g0 < ngroups; g0 += 4 * G) {
                uint4 a0[G], a1[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int it = (g0 + 4 * u) * 16 + li;
                    size_t pix = 0; bool inside = it < items;
                    if (lvl == 2) {
                        const int yy = y0 - 1 + it / R3, xx = x0 - 1 + it % R3;
                        inside = inside && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                        pix = ((size_t)n * p.H + yy) * p.W + xx;
                    } else if (lvl == 1) {
                        pix = ((size_t)n * p.H2 + y2lo + it / n2x) * p.W2 + x2lo + it % n2x;
                    } else {
                        pix = ((size_t)n * p.H4 + y1lo + it / n1x) * p.W4 + x1lo + it % n1x;
                    }
                    a0[u] = make_uint4(0u, 0u, 0u, 0u); a1[u] = a0[u];
                    if (inside) {
                        a0[u] = *(const uint4*)(base + pix * 64 + kg * 8);
                        a1[u] = *(const uint4*)(base + pix * 64 + 32 + kg * 8);
                    }
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int g = g0 + 4 * u;
                    if (g >= ngroups) break;                        // wave-uniform
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    acc = TAIL_MFMA(a0[u], bh0, acc); acc = TAIL_MFMA(a0[u], bl0, acc);
                    acc = TAIL_MFMA(a1[u], bh1, acc); acc = TAIL_MFMA(a1[u], bl1, acc);
                    if (li < 9) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int it = g * 16 + 4 * kg + r;
                            if (it < items) Tl[it * 9 + li] = acc[r];
                        }
                    }
                }
            }
        }
    } else
#pragma unroll 1
    for (int lvl = 0; lvl < 3; ++lvl) {
        const int items = lvl == 0 ? n1 : lvl == 1 ? n2 : n3;
        const void* base = lvl == 0 ? p.d1 : lvl == 1 ? p.d2 : p.d3;
        float* Tl = lvl == 0 ? T1 : lvl == 1 ? T2 : T3;
        float wr[9][8];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const f32x4 wa = *(const f32x4*)(wcs + (lvl * 9 + t) * 64 + sub * 8), wb = *(const f32x4*)(wcs + (lvl * 9 + t) * 64 + sub * 8 + 4);
            wr[t][0] = wa[0]; wr[t][1] = wa[1]; wr[t][2] = wa[2]; wr[t][3] = wa[3]; wr[t][4] = wb[0]; wr[t][5] = wb[1]; wr[t][6] = wb[2]; wr[t][7] = wb[3];
        }
        // FOUR pixels per lane and pass: their loads are issued together (one load in flight per lane made the 11 + 4 + 2 passes of
        // a tile a chain of memory latencies - the kernel ran at 2.3 TB/s of 16-byte loads with the VALU idle)
        constexpr int U = 4;
        for (int it0 = slot; it0 < ((items + 31) & ~31); it0 += 32 * U) {
            Raw8<BF16> raw[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int it = it0 + 32 * u;
                size_t pix = 0; bool inside = it < items;
                if (lvl == 2) {
                    const int yy = y0 - 1 + it / R3, xx = x0 - 1 + it % R3;
                    inside = inside && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                    pix = ((size_t)n * p.H + yy) * p.W + xx;
                } else if (lvl == 1) {
                    pix = ((size_t)n * p.H2 + y2lo + it / n2x) * p.W2 + x2lo + it % n2x;
                } else {
                    pix = ((size_t)n * p.H4 + y1lo + it / n1x) * p.W4 + x1lo + it % n1x;
                }
                ok[u] = inside;
                raw[u].a = make_uint4(0u, 0u, 0u, 0u); raw[u].b = raw[u].a;
                if (inside) raw[u] = load8_raw<BF16>(base, pix, sub * 8);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int it = it0 + 32 * u;
                if (it >= ((items + 31) & ~31)) break;                 // whole groups of 32 slots leave together (the shuffles below)
                float acc[9], v[8];
                unpack8<BF16>(raw[u], v);                              // zeros outside the image / past the level's window
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    acc[t] = v[0] * wr[t][0] + v[1] * wr[t][1] + v[2] * wr[t][2] + v[3] * wr[t][3] + v[4] * wr[t][4] + v[5] * wr[t][5] + v[6] * wr[t][6] + v[7] * wr[t][7];
#pragma unroll
                for (int t = 0; t < 9; ++t) { float a = acc[t]; a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4); acc[t] = a; }
                if (it < items) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) if (sub == (t & 7)) Tl[it * 9 + t] = acc[t];
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 2: one output pixel per thread ----
    {
        const int ty = tid >> 4, tx = tid & 15, y = y0 + ty, x = x0 + tx;
        float d = 0.f;
        if (y < p.H && x < p.W) {
            d = wcs[3 * 9 * 64 + 9];
            // the three rows / columns of the 3 x 3 window in each level's table (nearest up-sampling resolved once per row / column)
            int r3[3], c3[3], r2[3], c2[3], r1[3], c1[3]; bool rin[3], cin[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int qy = y + k - 1, qx = x + k - 1;
                rin[k] = qy >= 0 && qy < p.H; cin[k] = qx >= 0 && qx < p.W;
                const int cy = min(max(qy, 0), p.H - 1), cx = min(max(qx, 0), p.W - 1);
                r3[k] = (qy - y0 + 1) * R3; c3[k] = qx - x0 + 1;
                r2[k] = (nearest_src(cy, p.sy2, p.H2) - y2lo) * n2x; c2[k] = nearest_src(cx, p.sx2, p.W2) - x2lo;
                r1[k] = (nearest_src(cy, p.sy4, p.H4) - y1lo) * n1x; c1[k] = nearest_src(cx, p.sx4, p.W4) - x1lo;
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (!rin[t / 3] || !cin[t % 3]) continue;                                   // zero padding of f (and of the gather)
                d += wcs[3 * 9 * 64 + t] + T3[(r3[t / 3] + c3[t % 3]) * 9 + t] + T2[(r2[t / 3] + c2[t % 3]) * 9 + t] + T1[(r1[t / 3] + c1[t % 3]) * 9 + t];
            }
            p.D[(((size_t)n * p.H + y) * p.W + x) * p.d_cs] = d;
        }
        Dt[tid] = d;
    }
    __syncthreads();

    // ---- phase 3: S = R * I_delta + R * I_low (model.py:233), four bands per lane ----
    // (four pixel-quads per lane and pass, loads first: the load -> store -> load chain of a one-at-a-time loop is a latency chain too)
    const int nq = p.s_cs >> 2, tot = TT * TT * nq;
    constexpr int U3 = 4;
    for (int i0 = tid; i0 < tot; i0 += 256 * U3) {
        f32x4 r[U3]; float il[U3], dl[U3]; size_t pixs[U3]; int qs[U3]; bool ok[U3];
#pragma unroll
        for (int u = 0; u < U3; ++u) {
            const int i = i0 + 256 * u;
            const int px = i / nq, q = i - px * nq, y = y0 + (px >> 4), x = x0 + (px & 15);
            ok[u] = i < tot && y < p.H && x < p.W;
            pixs[u] = ((size_t)n * p.H + y) * p.W + x; qs[u] = q;
            r[u] = f32x4{0.f, 0.f, 0.f, 0.f}; il[u] = 0.f; dl[u] = 0.f;
            if (ok[u]) {
                const float* rl = p.RL + pixs[u] * p.rl_cs;
                il[u] = rl[p.B]; dl[u] = Dt[px];
                if (4 * q + 4 <= p.rl_cs) r[u] = *(const f32x4*)(rl + 4 * q);
            }
        }
#pragma unroll
        for (int u = 0; u < U3; ++u) {
            if (!ok[u]) continue;
            f32x4 sv;
#pragma unroll
            for (int j = 0; j < 4; ++j) sv[j] = (4 * qs[u] + j < p.B) ? r[u][j] * dl[u] + r[u][j] * il[u] : 0.f;
            *(f32x4*)(p.S + pixs[u] * p.s_cs + 4 * qs[u]) = sv;
        }
    }
}

int ssie_launch_tail_weights(const float* wf, const float* bf, const float* wl, const float* bl, float* out, hipStream_t st)
{
    hipLaunchKernelGGL(tail_weights_kernel, dim3((TAILW_PACK_OFF + 3 * 2 * 64 * 8 + 255) / 256), dim3(256), 0, st, wf, bf, wl, bl, out);
    return hipGetLastError() == hipSuccess ? 0 : 81;
}

int ssie_tail_supported(int H, int W, int H2, int W2, int H4, int W4)
{
    // the coarse-level windows must fit their LDS tables: (18 * in/out) + 2 source pixels per edge
    auto span = [](int hv, int hs) { return (int)((double)(TT + 2) * hs / hv) + 2; };
    return span(H, H2) <= R2MAX && span(W, W2) <= R2MAX && span(H, H4) <= R1MAX && span(W, W4) <= R1MAX;
}

int ssie_launch_tail(const void* d1, const void* d2, const void* d3, int bf16_in, int N, int H, int W, int H2, int W2, int H4, int W4,
                     const float* wc, const float* RL, int rl_cs, float* D, int d_cs, float* S, int s_cs, int B, hipStream_t st)
{
    if (!ssie_tail_supported(H, W, H2, W2, H4, W4) || s_cs % 4 || rl_cs % 4) return 82;
    TailParams p;
    p.d1 = d1; p.d2 = d2; p.d3 = d3; p.H = H; p.W = W; p.H2 = H2; p.W2 = W2; p.H4 = H4; p.W4 = W4; p.N = N;
    p.sy2 = (H2 == H) ? 1.f : (float)H2 / (float)H; p.sx2 = (W2 == W) ? 1.f : (float)W2 / (float)W;
    p.sy4 = (H4 == H) ? 1.f : (float)H4 / (float)H; p.sx4 = (W4 == W) ? 1.f : (float)W4 / (float)W;
    p.wc = wc; p.RL = RL; p.rl_cs = rl_cs; p.D = D; p.d_cs = d_cs; p.S = S; p.s_cs = s_cs; p.B = B;
    const int tiles = N * ((H + TT - 1) / TT) * ((W + TT - 1) / TT);
    if (bf16_in) hipLaunchKernelGGL(tail_kernel<true>, dim3(tiles), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(tail_kernel<false>, dim3(tiles), dim3(256), 0, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 83;
}

// ---------------------------------------------------------------------------------------------
// final_conv (3x3, 64 -> 1, model.py:141,174) in the TRAINING step: forward, data gradient and weight gradient as HBM-bound
// VALU kernels.  On the 32/64-wide MFMA tiles a 1-channel output (or 1-channel contraction) wastes 31/32 of every MFMA: the
// three launches took 0.47 ms of a 28 ms step at 3-6 TFLOP/s; their floor is one pass over the 64-channel tensor each.
// ---------------------------------------------------------------------------------------------
// forward: D[p] = b + sum_tap w[tap] . f[p + tap].  Phase 1: nine dots per INPUT pixel of the 18 x 18 halo tile (16 lanes per
// pixel, one float4 each); phase 2: an output pixel sums nine of them.
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float* __restrict__ f, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ D, int d_cs, int N, int H, int W)
{
    __shared__ __attribute__((aligned(16))) float ws[9 * 64];
    __shared__ float T[R3 * R3 * 9];
    const int tid = threadIdx.x;
    const int tiles_x = (W + TT - 1) / TT, tiles_y = (H + TT - 1) / TT;
    const int n = blockIdx.x / (tiles_x * tiles_y), tr = blockIdx.x % (tiles_x * tiles_y);
    const int y0 = (tr / tiles_x) * TT, x0 = (tr % tiles_x) * TT;
    for (int i = tid; i < 9 * 64; i += 256) ws[i] = w[(i & 63) * 9 + (i >> 6)];       // OIHW (1, 64, 3, 3) -> [tap][c]
    __syncthreads();
    const int sub = tid & 15, slot = tid >> 4;
    f32x4 wr[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t] = *(const f32x4*)(ws + t * 64 + sub * 4);
    // 21 halo pixels per 16-lane slot, in three batches of seven whose loads all go out before the first dot product (one load ->
    // nine dots -> shuffles per iteration exposed 21 HBM latencies in series per workgroup: 93 us for a 27-us pass over f)
    constexpr int NIT = ((R3 * R3 + 15) & ~15) / 16, NB = 7;
    static_assert(NIT % NB == 0, "whole batches");
    for (int i0 = 0; i0 < NIT; i0 += NB) {
        f32x4 v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int it = slot + 16 * (i0 + u);
            const int yy = y0 - 1 + it / R3, xx = x0 - 1 + it % R3;
            const bool ok = it < R3 * R3 && yy >= 0 && yy < H && xx >= 0 && xx < W;
            v[u] = *(const f32x4*)(f + (ok ? (((size_t)n * H + yy) * W + xx) * 64 : 0) + sub * 4);
            if (!ok) v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int it = slot + 16 * (i0 + u);
            float acc[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = v[u][0] * wr[t][0] + v[u][1] * wr[t][1] + v[u][2] * wr[t][2] + v[u][3] * wr[t][3];
#pragma unroll
            for (int t = 0; t < 9; ++t) { float a = acc[t]; a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4); a += __shfl_xor(a, 8); acc[t] = a; }
            if (it < R3 * R3) {
#pragma unroll
                for (int t = 0; t < 9; ++t) if (sub == t) T[it * 9 + t] = acc[t];
            }
        }
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15, y = y0 + ty, x = x0 + tx;
    if (y < H && x < W) {
        float d = bias ? bias[0] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int qy = ty + t / 3, qx = tx + t % 3;                      // halo-tile coordinates of p + tap
            d += T[(qy * R3 + qx) * 9 + t];                                    // outside the image: T = 0 (zero padding)
        }
        D[(((size_t)n * H + y) * W + x) * d_cs] = d;
    }
}

// data gradient: Gf[q][c] = sum_tap w[tap][c] * gD[q - tap]   (16 lanes per pixel, one float4 of channels each)
__global__ __launch_bounds__(256) void skinny_dgrad_kernel(const float* __restrict__ gD, int d_cs, const float* __restrict__ w,
                                                           float* __restrict__ Gf, int N, int H, int W)
{
    const int sub = threadIdx.x & 15;
    f32x4 wr[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[t][j] = w[(sub * 4 + j) * 9 + t];
    const long npix = (long)N * H * W;
    for (long pix = (long)blockIdx.x * 16 + (threadIdx.x >> 4); pix < npix; pix += (long)gridDim.x * 16) {
        const int x = (int)(pix % W), y = (int)((pix / W) % H);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        float g9[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {                  // nine unconditional loads (clamped address, value zeroed): no load -> wait chain
            const int py = y - (t / 3 - 1), px = x - (t % 3 - 1);              // output position p with p + tap = q
            const bool ok = py >= 0 && py < H && px >= 0 && px < W;
            g9[t] = gD[(ok ? pix + (long)(py - y) * W + (px - x) : pix) * d_cs];
            if (!ok) g9[t] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += g9[t] * wr[t][j];
        *(f32x4*)(Gf + pix * 64 + sub * 4) = a;
    }
}

// weight + bias gradient: dw[c][tap] = sum_p gD[p] * f[p + tap][c] = sum_q f[q][c] * gD[q - tap], db = sum_p gD[p].
// Every block reduces its grid-stride share of the pixels to one [9][64] (+1) partial (fixed order: bit-reproducible);
// skinny_wgrad_final_kernel adds the partials in block order into the parameter-gradient buffer.
#define SKINNY_WGRAD_BLOCKS 1024      // 4 resident blocks per CU: the loop is a chain of dependent loads, so memory-level parallelism comes from occupancy
__global__ __launch_bounds__(256) void skinny_wgrad_kernel(const float* __restrict__ f, const float* __restrict__ gD, int d_cs,
                                                           float* __restrict__ part, int N, int H, int W)
{
    __shared__ float red[16][9 * 64 + 1];
    const int sub = threadIdx.x & 15, slot = threadIdx.x >> 4;
    f32x4 acc[9];
    float accb = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long npix = (long)N * H * W;
    for (long pix = (long)blockIdx.x * 16 + slot; pix < npix; pix += (long)gridDim.x * 16) {
        const int x = (int)(pix % W), y = (int)((pix / W) % H);
        const f32x4 v = *(const f32x4*)(f + pix * 64 + sub * 4);
        float g9[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {                  // unconditional loads (see skinny_dgrad_kernel)
            const int py = y - (t / 3 - 1), px = x - (t % 3 - 1);
            const bool ok = py >= 0 && py < H && px >= 0 && px < W;
            g9[t] = gD[(ok ? pix + (long)(py - y) * W + (px - x) : pix) * d_cs];
            if (!ok) g9[t] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] += v * g9[t];
        if (sub == 0) accb += g9[4];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[slot][t * 64 + sub * 4 + j] = acc[t][j];
    if (sub == 0) red[slot][9 * 64] = accb;
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 64 + 1; i += 256) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][i];
        part[(size_t)i * SKINNY_WGRAD_BLOCKS + blockIdx.x] = s;              // [output][block]: the final pass reads rows
    }
}

// one wave per output element: lanes stride over the blocks' partials (coalesced), fixed-order butterfly => deterministic
__global__ __launch_bounds__(256) void skinny_wgrad_final_kernel(const float* __restrict__ part, int nblk, float* __restrict__ dw, float* __restrict__ db)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i > 9 * 64) return;
    float s = 0.f;
    for (int b = lane; b < nblk; b += 64) s += part[(size_t)i * nblk + b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane != 0) return;
    if (i == 9 * 64) { if (db) db[0] += s; }
    else dw[(i & 63) * 9 + (i >> 6)] += s;                                   // [tap][c] -> OIHW (1, 64, 3, 3)
}

int ssie_launch_skinny_fwd(const float* f, const float* w, const float* bias, float* D, int d_cs, int N, int H, int W, hipStream_t st)
{
    const int tiles = N * ((H + TT - 1) / TT) * ((W + TT - 1) / TT);
    hipLaunchKernelGGL(skinny_fwd_kernel, dim3(tiles), dim3(256), 0, st, f, w, bias, D, d_cs, N, H, W);
    return hipGetLastError() == hipSuccess ? 0 : 84;
}
int ssie_launch_skinny_dgrad(const float* gD, int d_cs, const float* w, float* Gf, int N, int H, int W, hipStream_t st)
{
    long blocks = ((long)N * H * W + 15) / 16; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(skinny_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gD, d_cs, w, Gf, N, H, W);
    return hipGetLastError() == hipSuccess ? 0 : 85;
}
size_t ssie_skinny_wgrad_ws_floats() { return (size_t)SKINNY_WGRAD_BLOCKS * (9 * 64 + 1); }
int ssie_launch_skinny_wgrad(const float* f, const float* gD, int d_cs, float* part, float* dw, float* db, int N, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(skinny_wgrad_kernel, dim3(SKINNY_WGRAD_BLOCKS), dim3(256), 0, st, f, gD, d_cs, part, N, H, W);
    hipLaunchKernelGGL(skinny_wgrad_final_kernel, dim3((9 * 64 + 1 + 3) / 4), dim3(256), 0, st, (const float*)part, SKINNY_WGRAD_BLOCKS, dw, db);
    return hipGetLastError() == hipSuccess ? 0 : 86;
}
