// conv_wino4_kernel: the big stride-1 3 x 3 convolutions (forward and data gradient) as Winograd F(4x4, 3x3) on the fp32 MFMA.
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      d = 6 x 6 input patch, g = 3 x 3 kernel, Y = 4 x 4 outputs
// 36 multiplications per 16 outputs and channel pair = 2.25 per output, against 4 for F(2x2, 3x3) (conv_wino.hip) and 9 for the direct
// kernels: the matrix pipe executes a QUARTER of the direct convolution's FLOPs.  The 36 transform positions xi = 6 i + j are 36
// independent GEMMs  M[xi][tile][co] = sum_ci V[xi][tile][ci] U[xi][ci][co]  on v_mfma_f32_16x16x4_f32.
// Matrices (Lavin & Gray, points 0, +-1, +-2, inf):
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// fp32 accuracy, measured on the network's real operands before the kernel was written (tools/wino43_error.py):
// max error 4e-7 ... 9e-6 of the tensor maximum (direct fp32: 2e-7 ... 1e-6), inside the 2e-5 operator bar and the 1e-5 output bar.
//
// Structure (what differs from conv_wino.hip, whose DMA / tile-queue / epilogue scheme this kernel keeps):
//   * workgroup = 8 waves = 16 x 64 output positions (4 x 16 tiles of 4 x 4) x 32 output channels.  Wave w owns tile row w & 3
//     (16 tiles = M) x the 16-channel half w >> 2 (N) x all 36 xi = 36 accumulator tiles of 4 registers = 144 registers: two waves
//     per SIMD (waves w and w + 4: same tiles, the two channel halves)
//   * K steps of 8 channels (36 xi make U four times the size of F(2x2)'s per channel: 16-channel steps do not fit the LDS twice):
//     per step the raw 18 x 66 halo tile (38 KB) and U (36 x 8 x 32 floats = 36 KB) are DMA'd into LDS, double-buffered
//   * A operand (16 tiles x 4 k): lane l = tile column l % 16, k-group g = l / 16 = channel PAIR g of the step; the lane reads patch
//     element (a, b) of its tile as ONE 8-byte word = both channels, so every transform instruction is a packed one (v_pk_fma_f32 /
//     v_pk_add_f32) and the two halves of a result feed the step's two MFMAs of a xi
//   * halo layout [shift group = (col >> 2) & 1][channel half][phase = col & 3][halo row][col >> 3] 16-byte slots, group 1 shifted by
//     8 bytes: the even / odd tiles of a k-group read the two groups, i.e. disjoint bank pairs: conflict-free 8-byte reads (see V_SHIFT)
//   * B^T d B in two halves of three output columns each (the 1-D transform's outputs {0,1,2} and {3,4,5} share no subexpression, so
//     splitting costs no arithmetic): the 6 x 6 patch is read twice per step instead of holding 72 registers of it
//   * U comes out of the weight-packing launch (PackDesc.wino = 2) in exactly the LDS image of a step: [xi][half][g][channel][2]
#include "conv_device.h"
#include <type_traits>

#ifndef SSIE_WINO4_UREG
#define SSIE_WINO4_UREG 0           // 1: U of the next step through registers instead of LDS-DMA (A/B: slower, see the step body)
#endif
#ifndef SSIE_WINO4_DMA_COL
#define SSIE_WINO4_DMA_COL(w) ((w) >> 1)      // transform column (0 .. 5) behind which wave w issues the next step's DMA
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// ablation builds (tools/build_variants.py conv_wino4.hip name=-DSSIE_X4_...): one phase compiled out, results wrong, timing only
#ifdef SSIE_X4_NOMFMA
#define MFMA16(a, b, c) ({ f32x4 c_ = (c); asm volatile("" : "+v"(c_) : "v"(a), "v"(b)); c_; })
#else
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#endif

// Diagnostic build only (-DSSIE_STAMP, tools/stamp_wino4.py): s_memtime of wave SSIE_STAMP_WAVE (default 0) per phase, summed per workgroup:
// [0] start [1] wait for the step's DMA (vmcnt) [2] wait at the barrier [3] end [4] DMA issue [5] epilogue + tile bookkeeping [6] tiles
// [7] the rest of the step bodies.  The shipped library never executes a stamp.
#ifdef SSIE_STAMP
#ifndef SSIE_STAMP_WAVE
#define SSIE_STAMP_WAVE 0
#endif
__device__ unsigned long long* ssie_stamp_buf_wino4 = nullptr;
extern "C" int ssie_debug_set_stamp_buffer_wino4(void* buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(ssie_stamp_buf_wino4), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#define ST_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t_ = __builtin_amdgcn_s_memtime(); st_[0] = st_t_;
#define ST_ACC(k) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[k] += t_ - st_t_; st_t_ = t_; } while (0)
#define ST_FLUSH do { st_[3] = __builtin_amdgcn_s_memtime(); if (ssie_stamp_buf_wino4 && threadIdx.x == 64 * SSIE_STAMP_WAVE) \
    for (int k_ = 0; k_ < 8; ++k_) ssie_stamp_buf_wino4[(size_t)blockIdx.x * 8 + k_] = st_[k_]; } while (0)
#else
#define ST_DECL
#define ST_ACC(k)
#define ST_FLUSH
#endif

namespace {

constexpr int V_TH = 16, V_TW = 64, V_HPH = 18, V_HPW = 66, V_CK = 8;
// Halo buffer: 16-byte slots (one pixel, one channel half = two channel pairs) in the order
//   [shift group sg = (col >> 2) & 1][channel half h][phase ph = col & 3][halo row][column group idx = col >> 3]
// and group 1 sits 8 BYTES further than its slot index says.  A k-group's 16 lanes read 8 bytes (their channel pair) of 16 slots: the
// even tiles' columns 8 u + b and the odd tiles' 8 u + 4 + b lie in phases p and p ^ 4, i.e. in the two shift groups, so the even
// lanes cover banks {4 u, 4 u + 1} and the odd lanes {4 u + 2, 4 u + 3}: conflict-free (unshifted, lanes u and u + 8 of the 16-byte-
// strided slots share a bank pair: measured SQ_LDS_BANK_CONFLICT = 38 % of the LDS cycles).
constexpr int V_IDX = 9;                            // column groups per phase plane (66 columns = 8.25 groups of 8)
constexpr int V_PLANE = V_HPH * V_IDX;              // slots per (group, half, phase) plane: 162
constexpr int V_HSL = 4 * V_PLANE;                  // per (group, half): 648
constexpr int V_SG = (2 * V_HSL + 63) / 64 * 64;    // slots per shift group, whole DMA pieces: 1344 (1296 used)
constexpr int V_SLOTS = 2 * V_SG;                   // 2688
constexpr int V_PIECES = V_SLOTS / 64;              // 42 DMA pieces of 64 slots
constexpr int V_ROUNDS = (V_PIECES + 7) / 8;        // 6 (the last one: pieces 40, 41)
constexpr int V_HP = V_SLOTS + 1;                   // float4 per halo buffer incl. the 8-byte shift (rounded to 16)
constexpr int V_SHIFT = V_SG * 16 + 8;              // byte distance group 0 -> group 1 of the same (h, ph, row, idx)
constexpr int V_BSZ = 36 * 2 * 4 * 16 * 2 / 4;      // float4 per U step: 2304 (36 KB)
constexpr int V_UPIECES = V_BSZ / 64;               // 36 DMA pieces of 1 KB

// packed fp32 arithmetic as inline asm (hipcc splits float2 operations whose halves are consumed one by one by MFMAs into scalar
// instructions, and every VALU instruction costs matrix-pipe time here).  The hazard recognizer does not see an asm as a VALU write:
// transformed values pass through V4_FENCE6 (s_nop) before the first MFMA reads them.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// a * k + c and c - a * k, k a uniform constant pair
// (k lives in an SGPR pair: one constant-bus operand per instruction is allowed, and it costs no vector register)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 k, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c)); return r; }
__device__ __forceinline__ f32x2 pk_fnma(f32x2 a, f32x2 k, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "s"(k), "v"(c)); return r; }
#define V4_FENCE6(a, b, c, d, e, f) asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f))

// one output row of a tile pair = 2 x 4 consecutive pixels of one channel: v[0..3] tile A, v[4..7] tile B (4 * pixstride further... the
// tiles are neighbours: 8 consecutive pixels).  Fused operands are loaded for all 8 elements before use.
template <int EPI, typename PT>     // EPI: epilogue shape (ssie_epi_shape, ssie_common.h)
__device__ __forceinline__ void wino4_epilogue8(const PT& p, float v[8], size_t o0, long pixstride, float bv)
{
    if (EPI == 1) {
        if (p.act == ACT_RELU) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k] + bv, 0.f);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += bv;
        }
        float* ob = p.out + o0;
#pragma unroll
        for (int k = 0; k < 8; ++k) ob[k * pixstride] = v[k];
        return;
    }
    if (EPI == 2) {
        float* ob = p.out + o0;
        if (p.mask_mode != MASK_NONE) {
            const float* mp = p.mask_y + o0;
            float y[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] = mp[k * pixstride];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = y[k] > 0.f ? v[k] : 0.f;
        }
        if (p.accumulate) {
            float a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = ob[k * pixstride];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += a[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) ob[k * pixstride] = v[k];
        return;
    }
    if (p.act == ACT_RELU) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k] + bv, 0.f);
    } else if (p.act == ACT_SIGMOID) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 1.f / (1.f + expf(-(v[k] + bv)));
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += bv;
    }
    if (p.mask_mode != MASK_NONE) {
        const float* mp = p.mask_y + o0;
        float y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = mp[k * pixstride];
        if (p.mask_mode == MASK_RELU) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = y[k] > 0.f ? v[k] : 0.f;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= y[k] * (1.f - y[k]);
        }
    }
    if (p.out2) {
        float* o2 = p.out2 + o0;
#pragma unroll
        for (int k = 0; k < 8; ++k) o2[k * pixstride] = v[k];
    }
    if (p.addsrc) {
        const float* ap = p.addsrc + o0;
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = ap[k * pixstride];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += a[k];
    }
    float* ob = p.out + o0;
    if (p.accumulate) {
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = ob[k * pixstride];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += a[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) ob[k * pixstride] = v[k];
}

template <typename PT>
__device__ __forceinline__ void wino4_epilogue8_ragged(const PT& p, const float v[8], size_t o0, long pixstride, float bv, int oy, int ox)
{
    if (oy >= p.Hout) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (ox + k >= p.Wout) continue;
        const size_t o = o0 + k * pixstride;
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

template <bool SINGLE, int EPI, bool RAG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino4_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    f32x4* As0 = (f32x4*)smem_f;                    // [2][V_HP]
    f32x4* Bs0 = As0 + 2 * V_HP;                    // [2][V_BSZ]
    int* s_next = (int*)(Bs0 + 2 * V_BSZ);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, nh = wave >> 2;        // tile row of the workgroup tile, 16-channel half of the 32-channel block
    const int g = lane >> 4, tx = lane & 15;
    // this lane's patch: halo rows 4 wm + a, halo columns 4 tx + b = 8 u + 4 e + b (tx = 2 u + e).  Element (a, b) of an even tile sits at
    // V4_AOFF(a, b) from abase0 / abase1 (b < 4 / b >= 4); an odd tile's column is 4 further: the OTHER shift group, and for b >= 4 the
    // next column group - folded into the two per-lane bases, so every element stays a compile-time offset
    const int u8 = tx >> 1, e8 = tx & 1;
    const int acommon = ((((g >> 1) * 4) * V_HPH + 4 * wm) * V_IDX + u8) * 16 + (g & 1) * 8;
    const int abase0 = acommon + e8 * V_SHIFT, abase1 = acommon + e8 * (16 - V_SHIFT);
#define V4_AOFF(a, b) (((b) >> 2) * V_SHIFT + ((((b) & 3) * V_HPH + (a)) * V_IDX) * 16)
    const int bbase = ((nh * 4 + g) * 16 + tx) * 8;           // U[xi][nh][g][tx][2 channels]; + xi * 1024 bytes
    const int nsteps = (p.Cin + V_CK - 1) / V_CK;
    const int total_tiles = p.N * p.tiles_y * p.tiles_x * p.co_blocks;

#define V4_DECODE(T, N_, A0_, B0_, CB_)                                                   \
    {                                                                                     \
        int q_ = (T);                                                                     \
        CB_ = q_ % p.co_blocks; q_ /= p.co_blocks;                                        \
        B0_ = (q_ % p.tiles_x) * V_TW; q_ /= p.tiles_x;                                   \
        A0_ = (q_ % p.tiles_y) * V_TH; N_ = q_ / p.tiles_y;                               \
    }
    // Per-lane DMA table, kept in REGISTERS (the LDS has no room for it): for each of the lane's 6 halo slots (slot i * 512 + tid) the byte
    // offset of its source relative to the tile origin (24 bits; sources of one channel stride) | halo column << 24 (7 bits) | channel
    // half << 31; 0xffffffff = a padding slot of the layout (fetches nothing).  [sources of different strides: pixel offset]
    const int cs_common = SINGLE ? p.src[0].cstride
                                 : ((p.nsrc < 2 || p.src[1].cstride == p.src[0].cstride) && (p.nsrc < 3 || p.src[2].cstride == p.src[0].cstride)
                                    ? p.src[0].cstride : 0);
    unsigned te[V_ROUNDS];
#pragma unroll
    for (int i = 0; i < V_ROUNDS; ++i) {
        const int id = i * 512 + tid;
        const int sg = id >= V_SG, r = id - sg * V_SG;
        const int h = r >= V_HSL, r1 = r - h * V_HSL;
        const int ph = (r1 >= V_PLANE) + (r1 >= 2 * V_PLANE) + (r1 >= 3 * V_PLANE), r2 = r1 - ph * V_PLANE;
        const int hy = (r2 * 7282) >> 16;                       // r2 / 9 for r2 < 256
        const int hx = 8 * (r2 - hy * V_IDX) + 4 * sg + ph;
        const bool real = id < V_SLOTS && r < 2 * V_HSL && hx < V_HPW;
        const int pix = hy * p.Wv + hx;
        const int lo = cs_common ? (pix * cs_common + 4 * h) * 4 : pix;
        te[i] = real ? ((unsigned)lo | ((unsigned)hx << 24) | ((unsigned)h << 31)) : 0xffffffffu;
    }
#define V4_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)
#define V4_BLDS(rs, lptr, vo, so) __builtin_amdgcn_raw_ptr_buffer_load_lds((rs), (__attribute__((address_space(3))) void*)(lptr), 16, (int)(vo), (int)(so), 0, 0)
    // Operands of one K step.  Halo: LDS-DMA, 42 pieces of 64 slots (wave w: pieces w, w + 8, ...; pieces 21 .. 41 = shift group 1 land 8
    // bytes further); padding slots fetch an out-of-range offset (zeros, never read).  U (36 pieces of 1 KB, wave w: pieces w, w + 8, ...)
    // goes THROUGH REGISTERS - an ordinary buffer load per piece, written to LDS one transform column later: the CU's LDS-DMA path
    // moves about 20 bytes per clock (stamped: ~400 cycles of issue stall per 1 KB piece with 78 pieces per step in flight), and
    // 74 KB per 8-channel step was more than it carries in a step's time; ordinary loads have their own, wider path.
#define V4_URSRC(STEP, CB_) V4_RSRC((const f32x4*)p.wpacked + ((size_t)(STEP) * p.co_blocks + (CB_)) * V_BSZ, V_BSZ * 16)
#define V4_PREFETCH_U(STEP, CB_, BUF)          /* first step of the kernel only: U by LDS-DMA like the halo */    \
    {                                                                                                         \
        f32x4* bbuf_ = Bs0 + (BUF) * V_BSZ;                                                                   \
        const __amdgpu_buffer_rsrc_t ur_ = V4_URSRC(STEP, CB_);                                               \
        _Pragma("unroll") for (int q_ = 0; q_ < (V_UPIECES + 7) / 8; ++q_) {                                  \
            const int pc_ = q_ * 8 + wave;                                                                    \
            if (pc_ < V_UPIECES) V4_BLDS(ur_, bbuf_ + pc_ * 64, lane * 16, pc_ * 1024);                       \
        }                                                                                                     \
    }
#define V4_PREFETCH(STEP, N_, A0_, B0_, CB_, BUF)                                                             \
    {                                                                                                         \
        const SrcSel s_ = SINGLE ? ssie_only_src(p) : ssie_pick_src(p, (STEP) * V_CK);                        \
        const int vy0_ = (A0_) - 1, vx0_ = (B0_) - 1;                                                         \
        char* abuf_ = (char*)(As0 + (BUF) * V_HP);                                                            \
        const int c0_ = (STEP) * V_CK - s_.cbeg;                                                              \
        /* one resource per (source, image): rows outside the image are out-of-range offsets -> zeros */      \
        const __amdgpu_buffer_rsrc_t ar_ = V4_RSRC(s_.ptr + (size_t)(N_) * p.Hv * p.Wv * s_.cstride,          \
                                                   (unsigned)(p.Hv * p.Wv * s_.cstride) * 4u);                \
        const int tb_ = ((vy0_ * p.Wv + vx0_) * s_.cstride + s_.coff + c0_) * 4;      /* bytes, may be negative */ \
        const unsigned xlo_ = max(0, -vx0_), xn_ = min(V_HPW, p.Wv - vx0_) - xlo_;                            \
        const int jn_ = (s_.C - c0_ + 3) >> 2;                                                                \
        const unsigned cs4_ = (unsigned)s_.cstride * 4u;                                                      \
        const bool inner_ = xn_ == (unsigned)V_HPW && jn_ >= 2;   /* no column outside the image, full channel step */ \
        _Pragma("unroll") for (int i_ = 0; i_ < V_ROUNDS; ++i_) {                                             \
            const int pc_ = i_ * 8 + wave;                                                                    \
            if (pc_ < V_PIECES) {                                                                             \
                const unsigned e_ = te[i_], lo_ = e_ & 0xffffffu, hx_ = (e_ >> 24) & 127, j_ = e_ >> 31;      \
                const bool ok_ = e_ != 0xffffffffu && (inner_ || (hx_ - xlo_ < xn_ && (int)j_ < jn_));        \
                const unsigned off_ = (unsigned)tb_ + (cs_common ? lo_ : __umul24(lo_, cs4_) + (j_ << 4));    \
                V4_BLDS(ar_, abuf_ + pc_ * 1024 + (pc_ >= V_PIECES / 2 ? 8 : 0), ok_ ? off_ : 0x80000000u, 0); \
            }                                                                                                 \
        }                                                                                                     \
    }

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int n, a0, b0, cb;
    V4_DECODE(tile, n, a0, b0, cb)
    int gstep = 0;
    V4_PREFETCH_U(0, cb, 0)
    V4_PREFETCH(0, n, a0, b0, cb, 0)
    int fetched = 0x7fffffff;
    ST_DECL
    const f32x2 k2 = {2.f, 2.f}, k4 = {4.f, 4.f}, k5 = {5.f, 5.f}, k8 = {8.f, 8.f};

    while (tile < total_tiles) {
        f32x4 acc[36];
        const int co = cb * 32 + nh * 16 + tx;
        const float bv = (EPI != 2 && p.bias && co < p.Cout) ? p.bias[co] : 0.f;
        int ntile = 0x7fffffff;
        int nn = n, na0 = a0, nb0 = b0, ncb = cb;

        auto step_body = [&](auto first_c, const int step) {
            constexpr bool FIRST = decltype(first_c)::value;
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
                if (ntile < total_tiles) V4_DECODE(ntile, nn, na0, nb0, ncb)
            }
            // what the next step is: the next 8 channels of this tile, or step 0 of the next tile
            const bool more = step + 1 < nsteps, have_next = more || ntile < total_tiles;
            const int ps = more ? step + 1 : 0, pn = more ? n : nn, pa0 = more ? a0 : na0, pb0 = more ? b0 : nb0, pcb = more ? cb : ncb;
#if SSIE_WINO4_UREG
            const __amdgpu_buffer_rsrc_t unext = V4_URSRC(ps, pcb);
            f32x4* const ubuf_next = Bs0 + (buf ^ 1) * V_BSZ;
            u32x4 ureg = {0u, 0u, 0u, 0u};
#endif
            const char* Ab0 = (const char*)(As0 + buf * V_HP) + abase0;
            const char* Ab1 = (const char*)(As0 + buf * V_HP) + abase1;
            const char* Bl = (const char*)(Bs0 + buf * V_BSZ) + bbase;
#pragma unroll
            for (int H = 0; H < 2; ++H) {
                // horizontal transform of the six patch rows: outputs 3 H .. 3 H + 2
                f32x2 rt[6][3];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    f32x2 d[6];
#pragma unroll
                    for (int b = 0; b < 6; ++b) d[b] = *(const f32x2*)((b < 4 ? Ab0 : Ab1) + V4_AOFF(a, b));
#ifdef SSIE_X4_NOXFORM
                    rt[a][0] = d[0 + 3 * H]; rt[a][1] = d[1 + 3 * H]; rt[a][2] = d[2 + 3 * H];
#else
                    if (H == 0) {
                        rt[a][0] = pk_fma(d[0], k4, pk_fnma(d[2], k5, d[4]));
                        const f32x2 s = pk_fnma(d[2], k4, d[4]), t = pk_fnma(d[1], k4, d[3]);
                        rt[a][1] = pk_add(s, t); rt[a][2] = pk_sub(s, t);
                    } else {
                        const f32x2 s = pk_sub(d[4], d[2]), t = pk_sub(d[3], d[1]);
                        rt[a][0] = pk_fma(t, k2, s); rt[a][1] = pk_fnma(t, k2, s);
                        rt[a][2] = pk_fma(d[1], k4, pk_fnma(d[3], k5, d[5]));
                    }
#endif
                }
#pragma unroll
                for (int bq = 0; bq < 3; ++bq) {
                    // vertical transform of column 3 H + bq -> the six xi = 6 i + 3 H + bq
                    f32x2 v[6];
#ifdef SSIE_X4_NOXFORM
#pragma unroll
                    for (int i = 0; i < 6; ++i) v[i] = rt[i][bq];
#else
                    {
                        const f32x2 d0 = rt[0][bq], d1 = rt[1][bq], d2 = rt[2][bq], d3 = rt[3][bq], d4 = rt[4][bq], d5 = rt[5][bq];
                        v[0] = pk_fma(d0, k4, pk_fnma(d2, k5, d4));
                        const f32x2 s = pk_fnma(d2, k4, d4), t = pk_fnma(d1, k4, d3);
                        v[1] = pk_add(s, t); v[2] = pk_sub(s, t);
                        const f32x2 s2 = pk_sub(d4, d2), t2 = pk_sub(d3, d1);
                        v[3] = pk_fma(t2, k2, s2); v[4] = pk_fnma(t2, k2, s2);
                        v[5] = pk_fma(d1, k4, pk_fnma(d3, k5, d5));
                    }
#endif
                    f32x2 bf[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) bf[i] = *(const f32x2*)(Bl + (i * 6 + 3 * H + bq) * 1024);
                    V4_FENCE6(v[0], v[1], v[2], v[3], v[4], v[5]);
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const int xi = i * 6 + 3 * H + bq;
                        acc[xi] = MFMA16(v[i].x, bf[i].x, FIRST ? zero4 : acc[xi]);
                    }
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const int xi = i * 6 + 3 * H + bq;
                        acc[xi] = MFMA16(v[i].y, bf[i].y, acc[xi]);
                    }
#ifndef SSIE_X4_NODMA
#if SSIE_WINO4_UREG
                    // (measured slower, 219 -> 236 us on 64 -> 64 at 128 x 128: vmcnt retires in order, so an ordinary load queued behind
                    // the halo's LDS-DMA pieces is waited for as long as they are)  the next step's U through registers, one piece per column
                    if (have_next) {
                        const int col = 3 * H + bq;
                        if (col >= 1 && (col - 1) * 8 + wave < V_UPIECES) *(u32x4*)(ubuf_next + ((col - 1) * 8 + wave) * 64 + lane) = ureg;
                        if (col <= 4 && col * 8 + wave < V_UPIECES) ureg = __builtin_amdgcn_raw_buffer_load_b128(unext, lane * 16, (col * 8 + wave) * 1024, 0);
                    }
#endif
                    // The next step's DMA.  The CU's LDS-DMA path takes a 1 KB piece about every 50 cycles and a wave sits at issue until its
                    // pieces are accepted: waves that issue TOGETHER each wait for all of their pieces (stamped: 2 600 - 4 300 cycles per step
                    // with four waves at a time), so every wave pair gets its own transform column - waves 2 c, 2 c + 1 behind column c -
                    // which also keeps SIMD partners (w, w + 4) two columns apart
                    if (3 * H + bq == SSIE_WINO4_DMA_COL(wave)) {
                        ST_ACC(7);
                        if (have_next) {
#if !SSIE_WINO4_UREG
                            V4_PREFETCH_U(ps, pcb, buf ^ 1)
#endif
                            V4_PREFETCH(ps, pn, pa0, pb0, pcb, buf ^ 1)
                        }
                        ST_ACC(4);
                    }
#endif
                }
            }
            if (tid == 0 && step == 0 && nsteps > 1 && p.tile_counter)
                fetched = (int)gridDim.x + atomicAdd(p.tile_counter, 1);
            ST_ACC(7);
        };
        step_body(std::true_type{}, 0); ++gstep;
        for (int step = 1; step < nsteps; ++step, ++gstep) step_body(std::false_type{}, step);

        // output transform (lane-local: register r of all 36 accumulators = tile column 4 g + r, channel co), packed over register pairs
        // (r, r + 1) = two tiles, + epilogue
#ifdef SSIE_X4_NOEPI
        if (co < p.Cout && acc[0][0] == 123.456f) {
#else
        if (co < p.Cout) {
#endif
            const long pixstride = p.out_cstride;
            const bool full = !RAG || (a0 + V_TH <= p.Hout && b0 + V_TW <= p.Wout);
            const int oy0 = a0 + 4 * wm;
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                f32x2 t[4][6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    f32x2 m[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) m[i] = f32x2{acc[i * 6 + j][2 * rp], acc[i * 6 + j][2 * rp + 1]};
                    const f32x2 s1 = pk_add(m[1], m[2]), d1 = pk_sub(m[1], m[2]), s2 = pk_add(m[3], m[4]), d2 = pk_sub(m[3], m[4]);
                    t[0][j] = pk_add(pk_add(m[0], s1), s2); t[1][j] = pk_fma(d2, k2, d1);
                    t[2][j] = pk_fma(s2, k4, s1); t[3][j] = pk_add(pk_fma(d2, k8, d1), m[5]);
                }
                // one output row at a time: 8 consecutive pixels (tile 4 g + 2 rp and its right neighbour), this lane's channel
                const int ox0 = b0 + 4 * (4 * g + 2 * rp);
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) {
                    const f32x2 s1 = pk_add(t[yy][1], t[yy][2]), d1 = pk_sub(t[yy][1], t[yy][2]), s2 = pk_add(t[yy][3], t[yy][4]), d2 = pk_sub(t[yy][3], t[yy][4]);
                    const f32x2 y0 = pk_add(pk_add(t[yy][0], s1), s2), y1 = pk_fma(d2, k2, d1), y2 = pk_fma(s2, k4, s1), y3 = pk_add(pk_fma(d2, k8, d1), t[yy][5]);
                    float yv[8] = {y0.x, y1.x, y2.x, y3.x, y0.y, y1.y, y2.y, y3.y};
                    const size_t o0 = ((size_t)(n * p.Hout + oy0 + yy) * p.Wout + ox0) * p.out_cstride + p.out_coff + co;
                    if (full) wino4_epilogue8<EPI>(p, yv, o0, pixstride, bv);
                    else wino4_epilogue8_ragged(p, yv, o0, pixstride, bv, oy0 + yy, ox0);
                }
            }
        }
        n = nn; a0 = na0; b0 = nb0; cb = ncb; tile = ntile;
#ifdef SSIE_STAMP
        st_[6] += 1;
#endif
    }
    ST_ACC(5);
    ST_FLUSH;
#undef V4_PREFETCH
#undef V4_PREFETCH_U
#undef V4_URSRC
#undef V4_DECODE
#undef V4_RSRC
#undef V4_BLDS
#undef V4_AOFF
}


size_t ssie_wino4_lds_bytes() { return (size_t)(2 * V_HP + 2 * V_BSZ) * 16 + 64; }

// p from ssie_conv_to_wino4 (layer_ops.hip: ssie_wino4_eligible has checked sources, sizes and the 24-bit slot offsets)
int ssie_launch_fprop_wino4(const ConvParams& p, hipStream_t st)
{
    if (p.ntaps != 9 || p.si != 1 || p.so != 1 || p.py || p.px || p.min_dy != -1 || p.min_dx != -1 || p.Cout_pad % 32) return 34;
    if (p.th != V_TH || p.tw != V_TW || p.hp_h != V_HPH || p.hp_w != V_HPW || p.co_blocks != p.Cout_pad / 32) return 35;
    for (int s = 0; s < p.nsrc; ++s)
        if (p.src[s].sy != 1.f || p.src[s].sx != 1.f || p.src[s].Hs != p.Hv || p.src[s].Ws != p.Wv ||
            (size_t)(V_HPH * p.Wv + V_HPW) * p.src[s].cstride * 4 >= (1u << 24) || (size_t)p.Hv * p.Wv * p.src[s].cstride * 4 >= (1u << 31)) return 36;
    const size_t tiles = (size_t)p.N * p.tiles_y * p.tiles_x * p.co_blocks;
    const dim3 grid((unsigned)(tiles < 256 ? tiles : 256));
    const size_t lds = ssie_wino4_lds_bytes();
    if (lds > 160 * 1024) return 37;
    const int epi = ssie_epi_shape(p);
    const bool rag = p.Hout % V_TH != 0 || p.Wout % V_TW != 0;
    static unsigned seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define W4_GO(S, E, R, SLOT) { ssie_allow_full_lds((const void*)conv_wino4_kernel<S, E, R>, seen[SLOT]); \
                               hipLaunchKernelGGL((conv_wino4_kernel<S, E, R>), grid, dim3(512), lds, st, p); }
#define W4_PICK(S, B) { if (rag) W4_GO(S, 0, true, B) else if (epi == 1) W4_GO(S, 1, false, B + 1) else if (epi == 2) W4_GO(S, 2, false, B + 2) else W4_GO(S, 0, false, B + 3) }
    if (p.nsrc == 1) W4_PICK(true, 0) else W4_PICK(false, 4)
#undef W4_PICK
#undef W4_GO
    return hipGetLastError() == hipSuccess ? 0 : 38;
}
