// conv_wino_kernel: the stride-1 3 x 3 convolutions (forward and data gradient) as Winograd F(2x2, 3x3) on the fp32 MFMA.
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      d = 4 x 4 input patch, g = 3 x 3 kernel, Y = 2 x 2 outputs
// 16 multiplications per 2 x 2 outputs and channel pair instead of 36: the 16 "transform positions" xi are 16 independent
// GEMMs  M[xi][tile][co] = sum_ci V[xi][tile][ci] U[xi][ci][co]  which run on v_mfma_f32_16x16x4_f32.
//   * U = G g G^T is produced by the weight-packing launch (pack_weights: PackDesc.wino) in the layout of a 16-"tap" packed
//     weight, so a K-chunk of it is DMA'd into LDS exactly like a tap group of the direct kernels
//   * the raw 18 x 34 halo tile of a 16-channel chunk is DMA'd into LDS (global_load_lds_dwordx4) with the direct kernels'
//     virtual-input addressing: channel concatenation, nearest up-sampling and zero padding resolved per 16-byte slot
//   * V = B^T d B is never materialised: each lane reads the 4 x 4 patch of ITS Winograd tile (16 ds_read_b128 per chunk) and
//     transforms it in registers right before the MFMAs that consume it; the output transform A^T M A is lane-local too
//   * one workgroup = 8 waves = 16 x 32 output positions (8 x 16 Winograd tiles) x 32 output channels; wave w owns tile row w
//     (16 tiles = M) x both 16-channel halves (N) x all 16 xi = 32 accumulator tiles of 4 registers = 128 registers, so two
//     waves share a SIMD and each one's LDS latency, barrier wait and epilogue run underneath the other's MFMAs
//   * persistent workgroups + dynamic tile queue + cross-tile prefetch as in the v2 kernels
// Measured on MI355X while building it: VALU instructions and MFMAs SHARE a SIMD's issue slot - every v_add
// between two MFMAs costs its 4 cycles whether the SIMD has one wave or two (tools/coexec_bench.hip) - so the transform runs on
// v_pk_add_f32, the DMA address arithmetic comes from a per-lane table, and nothing is recomputed per step that can be avoided.
//   A operand (16 tiles x 4 k): lane l = tile column l%16, k-group g = l/16 = channel quad g of the chunk; the 4 components of the
//   lane's float4 are the 4 MFMAs of a (xi, N-half).   B operand: lane = channel l%16 of the half, same quad g.
//   D (16 x 16): lane l = channel l%16, registers r = tile columns 4g + r.
#include "conv_device.h"
#include <type_traits>

__device__ f32x4 wino_zero_page[4];   // zero-initialised: source of padding slots

// Diagnostic build only (-DSSIE_STAMP, tools/stamp_wino.py): wave 0's s_memtime per phase, summed per workgroup:
// [0] start [1] wait for the step's DMA (vmcnt) [2] wait at the barrier [3] end [5] epilogue + tile bookkeeping [6] tiles
// [7] the step bodies.  The shipped library never executes a stamp.
#ifdef SSIE_STAMP
__device__ unsigned long long* ssie_stamp_buf_wino = nullptr;
extern "C" int ssie_debug_set_stamp_buffer_wino(void* buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(ssie_stamp_buf_wino), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#define ST_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t_ = __builtin_amdgcn_s_memtime(); st_[0] = st_t_;
#define ST_ACC(k) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[k] += t_ - st_t_; st_t_ = t_; } while (0)
#define ST_FLUSH do { st_[3] = __builtin_amdgcn_s_memtime(); if (ssie_stamp_buf_wino && threadIdx.x == 0) \
    for (int k_ = 0; k_ < 8; ++k_) ssie_stamp_buf_wino[(size_t)blockIdx.x * 8 + k_] = st_[k_]; } while (0)
#else
#define ST_DECL
#define ST_ACC(k)
#define ST_FLUSH
#endif

#define GLDS16(gptr, lptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

namespace {

template <bool UP>
__device__ __forceinline__ const f32x4* wino_virtual_addr(const SrcSel& s, int n, int vy, int vx, int Hv, int Wv, int c)
{
    const bool ok = (unsigned)vy < (unsigned)Hv && (unsigned)vx < (unsigned)Wv && c < s.C;
    int y = vy, x = vx;
    if (UP) {
        const int cy = min(max(vy, 0), Hv - 1), cx = min(max(vx, 0), Wv - 1);
        y = min((int)floorf((float)cy * s.sy), s.Hs - 1);
        x = min((int)floorf((float)cx * s.sx), s.Ws - 1);
    }
    const unsigned off = (unsigned)((n * s.Hs + y) * s.Ws + x) * (unsigned)s.cstride + (unsigned)(s.coff + c);
    // select by mask arithmetic: written as `ok ? a : z` hipcc turns the address computation into a branch per slot
    const unsigned long long a = (unsigned long long)(s.ptr + off), z = (unsigned long long)wino_zero_page;
    const unsigned long long m = ok ? ~0ull : 0ull;
    return (const f32x4*)((a & m) | (z & ~m));
}

constexpr int W_TH = 16, W_TW = 32, W_HPH = 18, W_HPW = 34, W_HP4 = W_HPH * W_HPW * 4;   // 2448 16-byte slots per halo tile
constexpr int W_HPB = (W_HP4 + 511) / 512 * 512;                                        // 16-byte slots per halo BUFFER (tile + padding): 5 DMA rounds of 512 lanes
constexpr int W_PLANE = W_HPB / 4;          // slots per channel-quad plane (612 used)
constexpr int W_HALF = W_HPH * (W_HPW / 2);   // slots per column-parity half plane (18 rows x 17 columns)
static_assert(W_PLANE >= 2 * W_HALF, "halo plane too small");
constexpr int W_BSZ_MAX = 16 * 4 * 32;                                                  // float4 per U chunk (16 xi x 16 ci x 32 co), the 32-channel form

}  // namespace

#ifndef SSIE_WINO_DMA_ROW_HI
#define SSIE_WINO_DMA_ROW_HI 1      // transform row after which waves 4-7 issue the next step's DMA (waves 0-3: row 0); A/B builds: tools/build_variant.sh <name> WORK -DSSIE_WINO_DMA_ROW_HI=k
#endif
#ifndef SSIE_WINO_ZEROC
#define SSIE_WINO_ZEROC 1
#endif
#ifndef SSIE_WINO_DMA_ROW_LO
#define SSIE_WINO_DMA_ROW_LO 0
#endif
#ifndef SSIE_WINO_DMA_SPLIT
#define SSIE_WINO_DMA_SPLIT 0       // 1: a wave's U pieces and halo pieces go out one transform row apart
#endif
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed fp32 add / subtract as inline asm: hipcc splits float2 additions whose lanes are consumed one by one (by the MFMAs) into two v_add_f32, and every
// VALU instruction costs MFMA time here.  The hazard recognizer does not see an asm as a VALU write, so the values pass through
// PK_FENCE (one s_nop covering the VALU-write -> MFMA-read wait states) before the first MFMA reads them.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define PK_FENCE8(a, b, c, d, e, f, g, h) asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h))

namespace {
// 16 outputs of one lane = (tile column 4g + r, r = k&3) x (pixel e = k>>2 of the 2 x 2 tile), one channel; o0 includes 8g pixels
#define WN_EOFF(k) ((long)((k) >> 3) * rowstride + (long)(2 * ((k) & 3) + (((k) >> 2) & 1)) * pixstride)
// EPI (conv_wino_kernel): 0 = every fused form; 1 = forward layers with bias + ReLU / nothing only; 2 = data gradients with an optional ReLU
// mask and accumulate only.  The kernel is sensitive to what its epilogue carries (round 4: two more run-time branches here cost every
// launch 4 %), so the two common shapes get instantiations without the forms they never use
template <int EPI, typename PT>
__device__ __forceinline__ void wino_epilogue16(const PT& p, float v[16], size_t o0, long rowstride, long pixstride, float bv)
{
    if (EPI == 1) {
        if (p.act == ACT_RELU) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = fmaxf(v[k] + bv, 0.f);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] += bv;
        }
        float* ob = p.out + o0;
#pragma unroll
        for (int k = 0; k < 16; ++k) ob[WN_EOFF(k)] = v[k];
        return;
    }
    if (EPI == 2) {
        float* ob = p.out + o0;
        if (p.mask_mode != MASK_NONE) {
            const float* mp = p.mask_y + o0;
            float y[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) y[k] = mp[WN_EOFF(k)];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = y[k] > 0.f ? v[k] : 0.f;
        }
        if (p.accumulate) {
            float a[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = ob[WN_EOFF(k)];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] += a[k];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) ob[WN_EOFF(k)] = v[k];
        return;
    }
    if (p.act == ACT_RELU) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = fmaxf(v[k] + bv, 0.f);
    } else if (p.act == ACT_SIGMOID) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 1.f / (1.f + expf(-(v[k] + bv)));
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] += bv;
    }
    if (p.mask_mode != MASK_NONE) {
        const float* mp = p.mask_y + o0;
        float y[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) y[k] = mp[WN_EOFF(k)];
        if (p.mask_mode == MASK_RELU) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = y[k] > 0.f ? v[k] : 0.f;
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] *= y[k] * (1.f - y[k]);
        }
    }
    if (p.out2) {
        float* o2 = p.out2 + o0;
#pragma unroll
        for (int k = 0; k < 16; ++k) o2[WN_EOFF(k)] = v[k];
    }
    if (p.addsrc) {
        const float* ap = p.addsrc + o0;
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = ap[WN_EOFF(k)];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] += a[k];
    }
    float* ob = p.out + o0;
    if (p.accumulate) {
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = ob[WN_EOFF(k)];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] += a[k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) ob[WN_EOFF(k)] = v[k];
}

template <typename PT>
__device__ __forceinline__ void wino_epilogue16_ragged(const PT& p, const float v[16], size_t o0, long rowstride, long pixstride, float bv,
                                                        int oy, int ox)
{
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int y = oy + (k >> 3), x = ox + 2 * (k & 3) + ((k >> 2) & 1);
        if (y >= p.Hout || x >= p.Wout) continue;
        const size_t o = o0 + WN_EOFF(k);
        float t = v[k] + bv;
        if (p.act == ACT_RELU) t = fmaxf(t, 0.f);
        else if (p.act == ACT_SIGMOID) t = 1.f / (1.f + expf(-t));
        if (p.mask_mode == MASK_RELU) t = p.mask_y[o] > 0.f ? t : 0.f;
        else if (p.mask_mode == MASK_SIGMOID) { const float yy = p.mask_y[o]; t *= yy * (1.f - yy); }
        if (p.out2) p.out2[o] = t;
        if (p.addsrc) t += p.addsrc[o];
        if (p.accumulate) t += p.out[o];
        p.out[o] = t;
    }
}
}  // namespace

// NH = 16-channel halves of output channels per workgroup: 2 (a 32-channel tile), or 1 for launches whose 32-channel tiles would leave
// half the CUs without one (the reference's shipped batch of 2: 128 tiles per 64-channel layer) - twice the workgroups, each with half
// the MFMAs and half of U; the input transform is repeated per workgroup, which an under-filled chip does not notice
template <bool SINGLE, bool UP, int EPI, bool RAG, int NH = 2>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int NTHR = 512;
    constexpr int NC = 16 * NH;                     // output channels per workgroup
    constexpr int W_BSZ = 16 * 4 * NC;              // float4 per U chunk (16 xi x 16 ci x NC co)
    f32x4* As0 = (f32x4*)smem_f;                    // [2][W_HPB]
    f32x4* Bs0 = As0 + 2 * W_HPB;                   // [2][W_BSZ]
    int* s_next = (int*)(Bs0 + 2 * W_BSZ);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, tx = lane & 15;
    // Halo buffer layout: [channel quad q][column parity][halo row][column >> 1] 16-byte slots.  The 16 lanes of a k-group read
    // patch element (a, b) of 16 Winograd tiles = every second column: split by parity these are CONSECUTIVE slots (conflict-
    // free ds_read_b128, SQ_LDS_BANK_CONFLICT = 0), and every element is a compile-time offset from one per-lane base address.
    // This lane's tile = (row `wave`, column tx), its channel quad = g.
    const int abase = ((2 * wave) * (W_HPW / 2) + tx + g * W_PLANE) * 16;
    const int nsteps = p.nchunks;
    const int total_tiles = p.N * p.tiles_y * p.tiles_x * p.co_blocks;

#define WN_DECODE(T, N_, A0_, B0_, CO0_)                                                  \
    {                                                                                     \
        int q_ = (T);                                                                     \
        CO0_ = (q_ % p.co_blocks) * NC; q_ /= p.co_blocks;                                \
        B0_ = (q_ % p.tiles_x) * W_TW; q_ /= p.tiles_x;                                   \
        A0_ = (q_ % p.tiles_y) * W_TH; N_ = q_ / p.tiles_y;                               \
    }
    // DMA of one step's operands: 4 pieces of U and 5 halo slots per lane.  VALU instructions and MFMAs SHARE a SIMD's issue slot on
    // this chip (tools/coexec_bench.hip: every v_add between two MFMAs costs its 4 cycles, with one or two waves per SIMD), so
    // the per-slot address arithmetic is kept minimal.  The operands are fetched through BUFFER descriptors (buffer_load_dwordx4
    // ... lds): one scalar resource per (source, image) whose num_records is the image's byte size, so a halo row above or below
    // the image is an out-of-range offset and the hardware writes zeros - no row compare, no 64-bit address, no zero page.  A
    // per-lane table in LDS holds, for each of the lane's 5 slots,
    //   byte offset of the slot relative to the tile origin (24 bits; sources of one channel stride: the offset of halo pixel
    //   (hy, hx), channel quad j)  |  hx << 24 (6 bits)  |  j << 30          [sources of different strides: pixel offset, not bytes]
    // and a step adds the scalar tile origin; only tiles that touch the left / right image border or a partial channel chunk
    // compare hx / j (a column outside the image is a valid address of the neighbouring row).  Padding slots (never read back)
    // fetch offset 0.  (UP variants: sources of different sizes: every slot is decoded arithmetically - x / 17 = x * 241 >> 12.)
    int* dma_tab = s_next + 4;                      // [W_HPB / NTHR][NTHR]
    // all sources of this launch share one channel stride (every layer but conv7's c5 | c0 concat): the table holds bytes
    const int cs_common = SINGLE ? p.src[0].cstride
                                 : ((p.nsrc < 2 || p.src[1].cstride == p.src[0].cstride) && (p.nsrc < 3 || p.src[2].cstride == p.src[0].cstride)
                                    ? p.src[0].cstride : 0);
    if (!UP) {
#pragma unroll
        for (int i = 0; i < W_HPB / NTHR; ++i) {
            const int id = i * NTHR + tid;
            const int j = (id >= W_PLANE) + (id >= 2 * W_PLANE) + (id >= 3 * W_PLANE), r = id - j * W_PLANE;
            const int par = r >= W_HALF, rr = r - par * W_HALF;
            const bool real = r < 2 * W_HALF;
            const int hy = real ? (rr * 241) >> 12 : 0;
            const int hx = real ? 2 * (rr - hy * (W_HPW / 2)) + par : 0;
            const int pix = hy * p.Wv + hx;
            const int lo = cs_common ? (pix * cs_common + 4 * j) * 4 : pix;
            dma_tab[i * NTHR + tid] = real ? (lo | (hx << 24) | (j << 30)) : 0;
        }
    }
    // U pieces: per-lane byte offset inside a piece pair (two (xi, q) rows of 32 float4), constant for the whole kernel
    // (NH = 1: a piece = four (xi, q) rows of 16 float4)
    const unsigned u_lane = NH == 2 ? (unsigned)(((lane >> 5) * p.Cout_pad + (lane & 31)) * 16) : (unsigned)(((lane >> 4) * p.Cout_pad + (lane & 15)) * 16);
    constexpr int U_ROWS = NH == 2 ? 2 : 4, U_PIECES = 64 / U_ROWS / 8;      // rows per piece; pieces per wave and step
#define WN_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)
#define WN_BLDS(rs, lptr, vo, so) __builtin_amdgcn_raw_ptr_buffer_load_lds((rs), (__attribute__((address_space(3))) void*)(lptr), 16, (int)(vo), (int)(so), 0, 0)
#define WN_PREFETCH(CHUNK, N_, A0_, B0_, CO0_, BUF, PART)                                                         \
    {                                                                                                         \
        f32x4* bbuf_ = Bs0 + (BUF) * W_BSZ;                                                                   \
        if ((PART) == 2) {                                                                                    \
        } else if (UP) {                                                                                      \
            const f32x4* wsrc_ = (const f32x4*)p.wpacked + (size_t)(CHUNK) * 64 * p.Cout_pad + (CO0_);        \
            _Pragma("unroll") for (int q_ = 0; q_ < U_PIECES; ++q_) {                                         \
                const int pc_ = q_ * 8 + wave;                    /* piece = two (xi, q) rows of 32 float4 */   \
                GLDS16((const char*)(wsrc_ + (unsigned)(pc_ * U_ROWS * p.Cout_pad)) + u_lane, bbuf_ + pc_ * 64);  \
            }                                                                                                 \
        } else {                                                                                              \
            const __amdgpu_buffer_rsrc_t ur_ = WN_RSRC((const f32x4*)p.wpacked + (size_t)(CHUNK) * 64 * p.Cout_pad + (CO0_), 0x7fffffff); \
            _Pragma("unroll") for (int q_ = 0; q_ < U_PIECES; ++q_) {                                         \
                const int pc_ = q_ * 8 + wave;                                                                \
                WN_BLDS(ur_, bbuf_ + pc_ * 64, u_lane, (unsigned)(pc_ * U_ROWS * p.Cout_pad * 16));           \
            }                                                                                                 \
        }                                                                                                     \
        const SrcSel s_ = SINGLE ? ssie_only_src(p) : ssie_pick_src(p, (CHUNK) * SSIE_CK);                    \
        const int vy0_ = (A0_) - 1, vx0_ = (B0_) - 1;                                                         \
        f32x4* abuf_ = As0 + (BUF) * W_HPB;                                                                   \
        if ((PART) == 1) {                                                                                    \
        } else if (UP) {                                                                                      \
            _Pragma("unroll") for (int i_ = 0; i_ < W_HPB / NTHR; ++i_) {                                     \
                const int id_ = i_ * NTHR + tid;                                                              \
                const int j_ = (id_ >= W_PLANE) + (id_ >= 2 * W_PLANE) + (id_ >= 3 * W_PLANE), r_ = id_ - j_ * W_PLANE; \
                const int par_ = r_ >= W_HALF, rr_ = r_ - par_ * W_HALF;                                      \
                const int hy_ = r_ < 2 * W_HALF ? (rr_ * 241) >> 12 : 255;                                    \
                const int hx_ = 2 * (rr_ - hy_ * (W_HPW / 2)) + par_;                                         \
                const f32x4* g_ = wino_virtual_addr<UP>(s_, (N_), vy0_ + hy_, vx0_ + hx_,                     \
                                                        p.Hv, p.Wv, (CHUNK) * SSIE_CK + 4 * j_ - s_.cbeg);    \
                GLDS16(g_, abuf_ + i_ * NTHR + wave * 64);                                                    \
            }                                                                                                 \
        } else {                                                                                              \
            const int c0_ = (CHUNK) * SSIE_CK - s_.cbeg;                                                      \
            /* one resource per (source, image): rows outside the image are out-of-range offsets -> zeros */  \
            const __amdgpu_buffer_rsrc_t ar_ = WN_RSRC(s_.ptr + (size_t)(N_) * p.Hv * p.Wv * s_.cstride,       \
                                                       (unsigned)(p.Hv * p.Wv * s_.cstride) * 4u);            \
            const int tb_ = ((vy0_ * p.Wv + vx0_) * s_.cstride + s_.coff + c0_) * 4;      /* bytes, may be negative */ \
            const unsigned xlo_ = max(0, -vx0_), xn_ = min(W_HPW, p.Wv - vx0_) - xlo_;                        \
            const int jn_ = (s_.C - c0_ + 3) >> 2;                                                            \
            const unsigned cs4_ = (unsigned)s_.cstride * 4u;                                                  \
            /* all table entries first: a DMA writes LDS, so the compiler will not move a table read above the previous DMA */ \
            unsigned te_[W_HPB / NTHR];                                                                       \
            _Pragma("unroll") for (int i_ = 0; i_ < W_HPB / NTHR; ++i_) te_[i_] = (unsigned)dma_tab[i_ * NTHR + tid]; \
            if (xn_ == (unsigned)W_HPW && jn_ >= 4) {          /* no column outside the image, full channel chunk */ \
                _Pragma("unroll") for (int i_ = 0; i_ < W_HPB / NTHR; ++i_) {                                 \
                    const unsigned lo_ = te_[i_] & 0xffffffu;                                                 \
                    const unsigned off_ = (unsigned)tb_ + (cs_common ? lo_ : __umul24(lo_, cs4_) + ((te_[i_] >> 30) << 4)); \
                    WN_BLDS(ar_, abuf_ + i_ * NTHR + wave * 64, off_, 0);                                     \
                }                                                                                             \
            } else {                                                                                          \
                _Pragma("unroll") for (int i_ = 0; i_ < W_HPB / NTHR; ++i_) {                                 \
                    const unsigned e_ = te_[i_], lo_ = e_ & 0xffffffu, hx_ = (e_ >> 24) & 63, j_ = e_ >> 30;  \
                    const bool ok_ = hx_ - xlo_ < xn_ && (int)j_ < jn_;                                       \
                    const unsigned off_ = (unsigned)tb_ + (cs_common ? lo_ : __umul24(lo_, cs4_) + (j_ << 4));         \
                    WN_BLDS(ar_, abuf_ + i_ * NTHR + wave * 64, ok_ ? off_ : 0x80000000u, 0);                 \
                }                                                                                             \
            }                                                                                                 \
        }                                                                                                     \
    }

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int n, a0, b0, co0;
    WN_DECODE(tile, n, a0, b0, co0)
    int gstep = 0;
    WN_PREFETCH(0, n, a0, b0, co0, 0, 0)
    int fetched = 0x7fffffff;
    ST_DECL

    while (tile < total_tiles) {
        f32x4 acc[16][NH];
#if !SSIE_WINO_ZEROC
#pragma unroll
        for (int x = 0; x < 16; ++x)
#pragma unroll
            for (int c = 0; c < NH; ++c) acc[x][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
        float bv[NH];
#pragma unroll
        for (int c = 0; c < NH; ++c) bv[c] = (EPI != 2 && p.bias && co0 + 16 * c + tx < p.Cout) ? p.bias[co0 + 16 * c + tx] : 0.f;
        int ntile = 0x7fffffff;
        int nn = n, na0 = a0, nb0 = b0, nco0 = co0;

        // SSIE_WINO_ZEROC: the first K step of a tile is its own instantiation of the step body whose first MFMA per accumulator takes a
        // zero C operand - the 128 accumulator clears per tile and wave go away (vector instructions on the MFMA's issue slot)
        auto step_body = [&](auto first_c, const int step) {
            constexpr bool FIRST = SSIE_WINO_ZEROC && decltype(first_c)::value;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            const int buf = gstep & 1;
            if (tid == 0) {
                if (nsteps == 1 || !p.tile_counter) {
                    if (step == 0) *s_next = p.tile_counter ? (int)gridDim.x + atomicAdd(p.tile_counter, 1) : tile + (int)gridDim.x;
                } else if (step == 1) *s_next = fetched;
            }
            ST_ACC(5);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ST_ACC(1);
            __syncthreads();
            ST_ACC(2);
            if (step == (nsteps > 1 ? 1 : 0)) {
                ntile = *s_next;
                if (ntile < total_tiles) WN_DECODE(ntile, nn, na0, nb0, nco0)
            }
            const char* Ab = (const char*)(As0 + buf * W_HPB) + abase;
            const f32x4* Bl = Bs0 + buf * W_BSZ + g * NC + tx;
            // the 4 x 4 patch of this lane's tile, channel quad g -> B^T d B, one transform row at a time; the additions are written
            // on float2 halves so that they compile to v_pk_add_f32 (half the VALU instructions)
            f32x2 d[4][4][2];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const f32x4 t = *(const f32x4*)(Ab + ((a * (W_HPW / 2) + (b >> 1)) + (b & 1) * W_HALF) * 16);
                    d[a][b][0] = f32x2{t.x, t.y}; d[a][b][1] = f32x2{t.z, t.w};
                }
            // U fragments are fetched one transform position ahead of the MFMAs that use them
            f32x4 bfn0 = Bl[0], bfn1 = Bl[NH == 2 ? 16 : 0];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 r[4][2], v[4][2];
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
                        r[b][hh] = i == 0 ? pk_sub(d[0][b][hh], d[2][b][hh]) : i == 1 ? pk_add(d[1][b][hh], d[2][b][hh])
                                 : i == 2 ? pk_sub(d[2][b][hh], d[1][b][hh]) : pk_sub(d[1][b][hh], d[3][b][hh]);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    v[0][hh] = pk_sub(r[0][hh], r[2][hh]); v[1][hh] = pk_add(r[1][hh], r[2][hh]);
                    v[2][hh] = pk_sub(r[2][hh], r[1][hh]); v[3][hh] = pk_sub(r[1][hh], r[3][hh]);
                }
                PK_FENCE8(v[0][0], v[0][1], v[1][0], v[1][1], v[2][0], v[2][1], v[3][0], v[3][1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int xi = i * 4 + j;
                    const f32x4 bf0 = bfn0, bf1 = bfn1;
                    if (xi < 15) { bfn0 = Bl[((xi + 1) * 4) * NC]; if (NH == 2) bfn1 = Bl[((xi + 1) * 4) * NC + 16]; }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[xi][0] = MFMA16(v[j][c >> 1][c & 1], bf0[c], (FIRST && c == 0) ? zero4 : acc[xi][0]);
                        if (NH == 2) acc[xi][NH - 1] = MFMA16(v[j][c >> 1][c & 1], bf1[c], (FIRST && c == 0) ? zero4 : acc[xi][NH - 1]);
                    }
                }
                // the next step's DMA is issued AFTER the first transform row's MFMAs: right behind the barrier all 8 waves would
                // do address arithmetic (VALU = no MFMA) while the matrix pipe has nothing queued yet
                // Waves w and w + 4 share a SIMD.  An LDS-DMA instruction holds its wave at issue while the wave's previous pieces are
                // in flight (measured: 340-550 cycles per piece in the bf16 kernels), and a wave stalled there issues no MFMAs - so the
                // partners must not issue their nine pieces at the same time: waves 4-7 do it SSIE_WINO_DMA_ROW_HI transform rows
                // later (-0.3 ms per train step against both at row 0), and with SSIE_WINO_DMA_SPLIT the U pieces and the halo pieces
                // of a wave go out one row apart.
                {
                    const bool more = step + 1 < nsteps;
                    const int r0 = wave >= 4 ? SSIE_WINO_DMA_ROW_HI : SSIE_WINO_DMA_ROW_LO;
                    if (!SSIE_WINO_DMA_SPLIT || UP) {
                        // ONE prefetch site (the next K chunk of this tile, or chunk 0 of the next tile, chosen by scalar selects): two
                        // sites doubled the DMA code of the hot path, and this kernel is faster the less code it carries (round 4)
                        if (i == r0 && (more || ntile < total_tiles)) {
                            const int ps = more ? step + 1 : 0, pn = more ? n : nn, pa0 = more ? a0 : na0, pb0 = more ? b0 : nb0, pco0 = more ? co0 : nco0;
                            WN_PREFETCH(ps, pn, pa0, pb0, pco0, buf ^ 1, 0)
                        }
                    } else {
                        if (i == r0) {
                            if (more) WN_PREFETCH(step + 1, n, a0, b0, co0, buf ^ 1, 1)
                            else if (ntile < total_tiles) WN_PREFETCH(0, nn, na0, nb0, nco0, buf ^ 1, 1)
                        }
                        if (i == r0 + 1) {
                            if (more) WN_PREFETCH(step + 1, n, a0, b0, co0, buf ^ 1, 2)
                            else if (ntile < total_tiles) WN_PREFETCH(0, nn, na0, nb0, nco0, buf ^ 1, 2)
                        }
                    }
                }
            }
            if (tid == 0 && step == 0 && nsteps > 1 && p.tile_counter)
                fetched = (int)gridDim.x + atomicAdd(p.tile_counter, 1);
            ST_ACC(7);
        };
#if SSIE_WINO_ZEROC
        step_body(std::true_type{}, 0); ++gstep;
        for (int step = 1; step < nsteps; ++step, ++gstep) step_body(std::false_type{}, step);
#else
        for (int step = 0; step < nsteps; ++step, ++gstep) step_body(std::false_type{}, step);
#endif

        // output transform (lane-local) + epilogue: one pass of 16 outputs per 16-channel half
        {
            const long rowstride = (long)p.Wout * p.out_cstride, pixstride = p.out_cstride;
            const int oy0 = a0 + 2 * wave, ox0 = b0 + 8 * g;
            const bool full = !RAG || (a0 + W_TH <= p.Hout && b0 + W_TW <= p.Wout);      // RAG = false: the output is whole 16 x 32 tiles
#pragma unroll
            for (int c = 0; c < NH; ++c) {
                const int co = co0 + 16 * c + tx;
                if (co >= p.Cout) continue;
                float y[16];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s0[4], s1[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        s0[i] = acc[i * 4 + 0][c][r] + acc[i * 4 + 1][c][r] + acc[i * 4 + 2][c][r];
                        s1[i] = acc[i * 4 + 1][c][r] - acc[i * 4 + 2][c][r] - acc[i * 4 + 3][c][r];
                    }
                    y[0 + r] = s0[0] + s0[1] + s0[2]; y[4 + r] = s1[0] + s1[1] + s1[2];
                    y[8 + r] = s0[1] - s0[2] - s0[3]; y[12 + r] = s1[1] - s1[2] - s1[3];
                }
                const size_t o0 = ((size_t)(n * p.Hout + oy0) * p.Wout + ox0) * p.out_cstride + p.out_coff + co;
                if (full) wino_epilogue16<EPI>(p, y, o0, rowstride, pixstride, bv[c]);
                else wino_epilogue16_ragged(p, y, o0, rowstride, pixstride, bv[c], oy0, ox0);
            }
        }
        n = nn; a0 = na0; b0 = nb0; co0 = nco0; tile = ntile;
#ifdef SSIE_STAMP
        st_[6] += 1;
#endif
    }
    ST_ACC(5);
    ST_FLUSH;
#undef WN_PREFETCH
#undef WN_DECODE
#undef WN_RSRC
#undef WN_BLDS
}

#define WINO_INST(S, U) template __global__ void conv_wino_kernel<S, U, 0, true>(const ConvParams); \
                        template __global__ void conv_wino_kernel<S, U, 1, true>(const ConvParams); \
                        template __global__ void conv_wino_kernel<S, U, 2, true>(const ConvParams); \
                        template __global__ void conv_wino_kernel<S, U, 0, false>(const ConvParams); \
                        template __global__ void conv_wino_kernel<S, U, 1, false>(const ConvParams); \
                        template __global__ void conv_wino_kernel<S, U, 2, false>(const ConvParams);
WINO_INST(false, false) WINO_INST(true, false) WINO_INST(false, true) WINO_INST(true, true)
#undef WINO_INST
#define WINO_INST1(S, U) template __global__ void conv_wino_kernel<S, U, 0, false, 1>(const ConvParams); \
                         template __global__ void conv_wino_kernel<S, U, 1, false, 1>(const ConvParams); \
                         template __global__ void conv_wino_kernel<S, U, 2, false, 1>(const ConvParams);
WINO_INST1(false, false) WINO_INST1(true, false) WINO_INST1(false, true) WINO_INST1(true, true)
#undef WINO_INST1


size_t ssie_wino_lds_bytes() { return (size_t)(2 * W_HPB + 2 * W_BSZ_MAX) * 16 + 64 + (size_t)W_HPB * 4; }

int ssie_wino_half_below = 256;        // 16-channel workgroups when the launch has fewer 32-channel tiles than this (one per CU)
extern "C" void ssie_debug_set_wino_half_below(int v) { ssie_wino_half_below = v; }

int ssie_launch_fprop_wino(const ConvParams& p, hipStream_t st)
{
    if (p.ntaps != 9 || p.si != 1 || p.so != 1 || p.py || p.px || p.min_dy != -1 || p.min_dx != -1 || p.Cout_pad % 32) return 31;
    if (p.th != W_TH || p.tw != W_TW || p.hp_h != W_HPH || p.hp_w != W_HPW || p.co_blocks != p.Cout_pad / 32) return 32;
    const size_t tiles32 = (size_t)p.N * p.tiles_y * p.tiles_x * p.co_blocks;
    const bool rag = p.Hout % W_TH != 0 || p.Wout % W_TW != 0;       // some tile sticks out of the output: keep the element-wise epilogue
    const bool half = !rag && (long)tiles32 < ssie_wino_half_below;   // under-filled: 16-channel workgroups (whole-tile launches only)
    ConvParams ph = p;
    if (half) ph.co_blocks = p.Cout_pad / 16;
    const size_t tiles = half ? 2 * tiles32 : tiles32;
    const dim3 grid((unsigned)(tiles < 256 ? tiles : 256));
    const size_t lds = ssie_wino_lds_bytes();
    // the DMA table holds a slot's byte offset from the tile origin in 24 bits (18 halo rows); larger images take the arithmetic decode
    bool up = false;
    for (int s = 0; s < p.nsrc; ++s)
        up = up || p.src[s].sy != 1.f || p.src[s].sx != 1.f || (size_t)(W_HPH * p.Wv + W_HPW) * p.src[s].cstride * 4 >= (1u << 24) ||
             (size_t)p.Hv * p.Wv * p.src[s].cstride * 4 >= (1u << 31);       // (and an image must fit a 32-bit num_records)
    // epilogue shape (see wino_epilogue16): 1 = plain forward layer, 2 = plain data gradient, 0 = anything else
    const bool plain = !p.out2 && !p.addsrc;
    const int epi = (plain && !p.mask_y && !p.accumulate && p.act != ACT_SIGMOID) ? 1
                  : (plain && !p.bias && p.act == ACT_NONE && p.mask_mode != MASK_SIGMOID) ? 2 : 0;
    static unsigned seen[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    static unsigned seen1[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define WINO_GO(S, U, E, SLOT) { if (half) { ssie_allow_full_lds((const void*)conv_wino_kernel<S, U, E, false, 1>, seen1[SLOT]); \
                                             hipLaunchKernelGGL((conv_wino_kernel<S, U, E, false, 1>), grid, dim3(512), lds, st, ph); } \
                                 else if (rag) { ssie_allow_full_lds((const void*)conv_wino_kernel<S, U, E, true>, seen[SLOT]); \
                                            hipLaunchKernelGGL((conv_wino_kernel<S, U, E, true>), grid, dim3(512), lds, st, p); } \
                                 else { ssie_allow_full_lds((const void*)conv_wino_kernel<S, U, E, false>, seen[12 + SLOT]); \
                                        hipLaunchKernelGGL((conv_wino_kernel<S, U, E, false>), grid, dim3(512), lds, st, p); } }
#define WINO_EPI(S, U, SLOT) { if (epi == 1) WINO_GO(S, U, 1, SLOT + 1) else if (epi == 2) WINO_GO(S, U, 2, SLOT + 2) else WINO_GO(S, U, 0, SLOT) }
    if (p.nsrc == 1 && !up) WINO_EPI(true, false, 0)
    else if (p.nsrc == 1) WINO_EPI(true, true, 3)
    else if (!up) WINO_EPI(false, false, 6)
    else WINO_EPI(false, true, 9)
#undef WINO_EPI
#undef WINO_GO
    return hipGetLastError() == hipSuccess ? 0 : 33;
}
