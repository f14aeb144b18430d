// conv_wgrad_wino_kernel: the weight gradient of a stride-1 3 x 3 convolution as Winograd F(3x3, 2x2).
//   dW[kh][kw] = sum over 2 x 2 tiles t of the output gradient  A^T [ (G g_t G^T) (.) (B^T d_t B) ] A
// g_t = the 2 x 2 gradient tile, d_t = the 4 x 4 input patch around it (pad 1): 16 multiplications per tile and (ci, co) instead of
// the 36 of the direct sum.  A^T (3 x 4) = [[1,1,1,0],[0,1,-1,0],[0,1,1,1]], G (4 x 2) = [[1,0],[1/2,1/2],[1/2,-1/2],[0,1]],
// B^T (4 x 4) = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,-1,0,1]] (checked against the direct correlation in fp64).  The sum over tiles is
// linear, so the kernel accumulates dU[xi][ci][co] = sum_t V_t[xi][ci] H_t[xi][co] for the 16 transform positions xi - 16 GEMMs
// with M = ci, N = co and K = TILES - and the 3 x 3 taps are formed once per workgroup, from its accumulators, before the partial slab is written (the fixed-order
// slab reduction), together with the 1/2 factors of G, which are left out of H.
//   * NO operand re-layout: on v_mfma_f32_32x32x2_f32 the A operand wants (row = ci, k = tile) and the B operand (k = tile,
//     column = co), i.e. lane l holds channel l % 32 of tile l / 32 of the pair.  That lane reads the 4 x 4 patch of ITS channel and
//     tile from the NHWC halo tile in LDS (consecutive lanes = consecutive channels: conflict-free) and transforms it in
//     registers: the 16 results ARE its A operands of the 16 xi; likewise the 2 x 2 gradient tile of its output channel -> B.
//   * VALU instructions cost matrix-pipe time on this chip, so a lane transforms TWO tile pairs at once on
//     v_pk_add_f32 (x = pair s, y = pair s + 1): 44 packed instructions per 32 MFMAs.
//   * workgroup = 4 waves = a (64 ci x 64 co) block of dU as 2 x 2 wave blocks of 32 x 32 x 16 xi = 256 accumulator registers per
//     wave (one wave per SIMD); position tiles of 8 x 16 outputs (4 x 8 Winograd tiles) are staged by global->LDS DMA into TWO
//     buffers, so the next tile lands underneath the current one's MFMAs; slices of position tiles per workgroup and a
//     fixed-order slab reduction as in conv_wgrad_kernel (bit-reproducible, no float atomics); fused bias gradient likewise.
#include "conv_device.h"

__device__ f32x4 wgw_zero_page[4];     // zero-initialised: source of the padding slots
#define GLDS16G(gptr, lptr)                                                                            \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

#ifndef SSIE_WGW_REGSTAGE
#define SSIE_WGW_REGSTAGE 0         // 1: the next position tile through registers (ordinary loads + ds_write) instead of LDS-DMA: A/B'd in round 4, 14.49 against 14.41 ms per step (equal at 256 bands) - not adopted
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed add / subtract as inline asm (hipcc splits float2 arithmetic whose lanes feed MFMAs one by one) + the fence that stands
// in for the VALU-write -> MFMA-read wait states the hazard recognizer cannot see behind an asm (conv_wino.hip)
__device__ __forceinline__ f32x2 wg_add(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 wg_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define WG_FENCE8(a) asm volatile("s_nop 1" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))

namespace {
constexpr int GW_TH = 8, GW_TW = 16, GW_HPH = 10, GW_HPW = 18;     // position tile and its halo (3 x 3, pad 1)
constexpr int GW_NT = (GW_TH / 2) * (GW_TW / 2);                    // 32 Winograd tiles per position tile
}

// UP: the source is nearest-up-sampled on read (F.interpolate(mode='nearest'), model.py:156-169), resolved per staged slot
template <int CIB, int COB, bool UP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_wgrad_wino_kernel(const WgradParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int MI = CIB / 32, NI = COB / 32, NPAIR = MI * NI, WSPLIT = 4 / NPAIR;
    constexpr int XSZ = GW_HPH * GW_HPW * CIB, GSZ = GW_TH * GW_TW * COB;          // floats per buffer
    constexpr int CI4 = CIB / 4, CO4 = COB / 4;
    constexpr int NX = (GW_HPH * GW_HPW * CI4 + 255) / 256, NG = GW_TH * GW_TW * CO4 / 256;
    float* Xs0 = smem_f;                           // [2][HP][CIB]
    float* Gs0 = smem_f + 2 * XSZ;                 // [2][PT][COB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, li = lane & 31;
    const int pair = wave % NPAIR, wsub = wave / NPAIR;
    const int mi = pair / NI, ni = pair % NI;
    const int slice = blockIdx.x;
    const int cib = blockIdx.y / p.co_blocks, cob = blockIdx.y % p.co_blocks;
    const int ci0 = cib * CIB, co0 = cob * COB;

    f32x16 acc[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

    const int tps = (p.tiles_total + p.nslices - 1) / p.nslices;
    const int tile_beg = slice * tps, tile_end = min(tile_beg + tps, p.tiles_total);
    const bool do_bias = p.bias_slabs != nullptr && cib == 0;
    constexpr int BROWS = 256 / COB;
    const int bcol = tid % COB, brow = tid / COB;
    float bsum = 0.f;

    // Plain (not up-sampled) sources are fetched through buffer descriptors (conv_wino.hip): one scalar resource per image whose
    // num_records is the image's byte size - rows of the halo outside the image are out-of-range offsets and arrive as zeros - and
    // ONE precomputed 32-bit byte offset per slot and thread (this kernel runs one wave per SIMD: registers are plentiful), so a
    // slot costs one vector add instead of ~14 address instructions; tiles on the left / right border and partial channel blocks
    // still compare the column / channel quad of a slot.
    unsigned xoff[NX], goff[NG];
    if (!UP) {
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int id = it * 256 + tid;
            const unsigned pix = (unsigned)id / CI4, j = (unsigned)id % CI4;
            const unsigned hy = (pix * 3641u) >> 16, hx = pix - hy * GW_HPW;
            xoff[it] = ((hy * (unsigned)p.Wv + hx) * (unsigned)p.src.cstride + 4u * j) * 4u;
        }
#pragma unroll
        for (int it = 0; it < NG; ++it) {
            const int id = it * 256 + tid;
            const unsigned pix = (unsigned)id / CO4, j = (unsigned)id % CO4;
            const unsigned gy = pix / GW_TW, gx = pix % GW_TW;
            goff[it] = ((gy * (unsigned)p.Wo + gx) * (unsigned)p.g_cstride + 4u * j) * 4u;
        }
    }
#define GW_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)
#define GW_BLDS(rs, lptr, vo) __builtin_amdgcn_raw_ptr_buffer_load_lds((rs), (__attribute__((address_space(3))) void*)(lptr), 16, (int)(vo), 0, 0, 0)
    // DMA of position tile TILE into buffer BUF: slot id = it * 256 + tid (the DMA writes LDS linearly) = (pixel, channel quad)
#define GW_STAGE(TILE, BUF)                                                                                   \
    {                                                                                                         \
        int tt_ = (TILE);                                                                                     \
        const int tx_ = tt_ % p.tiles_x; tt_ /= p.tiles_x;                                                    \
        const int ty_ = tt_ % p.tiles_y, n_ = tt_ / p.tiles_y;                                                \
        const int a0_ = ty_ * GW_TH, b0_ = tx_ * GW_TW;                                                       \
        if (!UP) {                                                                                            \
            const __amdgpu_buffer_rsrc_t xr_ = GW_RSRC(p.src.ptr + (size_t)n_ * p.Hv * p.Wv * p.src.cstride,   \
                                                       (unsigned)(p.Hv * p.Wv * p.src.cstride) * 4u);         \
            const unsigned xb_ = (unsigned)((((a0_ - 1) * p.Wv + b0_ - 1) * p.src.cstride + p.src.coff + ci0) * 4); \
            const unsigned xlo_ = b0_ == 0, xn_ = min(GW_HPW, p.Wv - b0_ + 1) - xlo_;                         \
            const unsigned jn_ = (unsigned)((p.src.C - ci0 + 3) >> 2);                                        \
            if (xn_ == (unsigned)GW_HPW && jn_ >= (unsigned)CI4) {                                            \
                _Pragma("unroll") for (int it_ = 0; it_ < NX; ++it_)                                          \
                    if (NX * 256 == GW_HPH * GW_HPW * CI4 || it_ * 256 + tid < GW_HPH * GW_HPW * CI4)         \
                        GW_BLDS(xr_, (f32x4*)(Xs0 + (BUF) * XSZ) + it_ * 256 + wave * 64, xb_ + xoff[it_]);   \
            } else {                                                                                          \
                _Pragma("unroll") for (int it_ = 0; it_ < NX; ++it_) {                                        \
                    const unsigned id_ = it_ * 256 + tid, pix_ = id_ / CI4, j_ = id_ % CI4;                   \
                    const unsigned hx_ = pix_ - ((pix_ * 3641u) >> 16) * GW_HPW;     /* column of the slot (border tiles only) */ \
                    const bool ok_ = hx_ - xlo_ < xn_ && j_ < jn_;                                            \
                    if (NX * 256 == GW_HPH * GW_HPW * CI4 || it_ * 256 + tid < GW_HPH * GW_HPW * CI4)         \
                        GW_BLDS(xr_, (f32x4*)(Xs0 + (BUF) * XSZ) + it_ * 256 + wave * 64, ok_ ? xb_ + xoff[it_] : 0x80000000u); \
                }                                                                                             \
            }                                                                                                 \
            const __amdgpu_buffer_rsrc_t gr_ = GW_RSRC(p.g + (size_t)n_ * p.Ho * p.Wo * p.g_cstride,          \
                                                       (unsigned)(p.Ho * p.Wo * p.g_cstride) * 4u);           \
            const unsigned gb_ = (unsigned)(((a0_ * p.Wo + b0_) * p.g_cstride + p.g_coff + co0) * 4);          \
            const unsigned gxn_ = min(GW_TW, p.Wo - b0_);                                                     \
            const unsigned gjn_ = (unsigned)((((p.Cout + 3) & ~3) - co0 + 3) >> 2);                           \
            if (gxn_ == (unsigned)GW_TW && gjn_ >= (unsigned)CO4) {                                           \
                _Pragma("unroll") for (int it_ = 0; it_ < NG; ++it_)                                          \
                    GW_BLDS(gr_, (f32x4*)(Gs0 + (BUF) * GSZ) + it_ * 256 + wave * 64, gb_ + goff[it_]);       \
            } else {                                                                                          \
                _Pragma("unroll") for (int it_ = 0; it_ < NG; ++it_) {                                        \
                    const unsigned id_ = it_ * 256 + tid, gx_ = (id_ / CO4) % GW_TW, j_ = id_ % CO4;          \
                    const bool ok_ = gx_ < gxn_ && j_ < gjn_;                                                 \
                    GW_BLDS(gr_, (f32x4*)(Gs0 + (BUF) * GSZ) + it_ * 256 + wave * 64, ok_ ? gb_ + goff[it_] : 0x80000000u); \
                }                                                                                             \
            }                                                                                                 \
        } else {                                                                                              \
        const int tbx_ = ((n_ * p.Hv + a0_ - 1) * p.Wv + b0_ - 1) * p.src.cstride + p.src.coff + ci0;         \
        const unsigned ylo_ = a0_ == 0, yn_ = min(GW_HPH, p.Hv - a0_ + 1) - ylo_;                             \
        const unsigned xlo_ = b0_ == 0, xn_ = min(GW_HPW, p.Wv - b0_ + 1) - xlo_;                             \
        const int jn_ = (p.src.C - ci0 + 3) >> 2;                                                             \
        const unsigned long long zp_ = (unsigned long long)wgw_zero_page;                                     \
        _Pragma("unroll") for (int it_ = 0; it_ < NX; ++it_) {                                                \
            const int id_ = it_ * 256 + tid;                                                                  \
            const unsigned pix_ = (unsigned)id_ / CI4, j_ = (unsigned)id_ % CI4;                              \
            const unsigned hy_ = (pix_ * 3641u) >> 16, hx_ = pix_ - hy_ * GW_HPW;   /* / 18 for pix < 4000 */   \
            const bool ok_ = hy_ - ylo_ < yn_ && hx_ - xlo_ < xn_ && (int)j_ < jn_ && id_ < GW_HPH * GW_HPW * CI4; \
            int off_ = tbx_ + (int)(hy_ * p.Wv + hx_) * p.src.cstride + 4 * (int)j_;                          \
            if (UP) {                                                                                         \
                const int cy_ = min(max(a0_ - 1 + (int)hy_, 0), p.Hv - 1), cx_ = min(max(b0_ - 1 + (int)hx_, 0), p.Wv - 1); \
                const int sy_ = min((int)floorf((float)cy_ * p.src.sy), p.src.Hs - 1), sx_ = min((int)floorf((float)cx_ * p.src.sx), p.src.Ws - 1); \
                off_ = ((n_ * p.src.Hs + sy_) * p.src.Ws + sx_) * p.src.cstride + p.src.coff + ci0 + 4 * (int)j_; \
            }                                                                                                 \
            const unsigned long long a_ = (unsigned long long)(p.src.ptr + off_), m_ = ok_ ? ~0ull : 0ull;    \
            if (NX * 256 == GW_HPH * GW_HPW * CI4 || id_ < GW_HPH * GW_HPW * CI4)                             \
                GLDS16G((const f32x4*)((a_ & m_) | (zp_ & ~m_)), (f32x4*)(Xs0 + (BUF) * XSZ) + it_ * 256 + wave * 64); \
        }                                                                                                     \
        const int tbg_ = ((n_ * p.Ho + a0_) * p.Wo + b0_) * p.g_cstride + p.g_coff + co0;                     \
        const unsigned gyn_ = min(GW_TH, p.Ho - a0_), gxn_ = min(GW_TW, p.Wo - b0_);                          \
        const int gjn_ = (((p.Cout + 3) & ~3) - co0 + 3) >> 2;                                                \
        _Pragma("unroll") for (int it_ = 0; it_ < NG; ++it_) {                                                \
            const int id_ = it_ * 256 + tid;                                                                  \
            const unsigned pix_ = (unsigned)id_ / CO4, j_ = (unsigned)id_ % CO4;                              \
            const unsigned gy_ = pix_ / GW_TW, gx_ = pix_ % GW_TW;                                            \
            const bool ok_ = gy_ < gyn_ && gx_ < gxn_ && (int)j_ < gjn_;                                      \
            const int off_ = tbg_ + (int)(gy_ * p.Wo + gx_) * p.g_cstride + 4 * (int)j_;                      \
            const unsigned long long a_ = (unsigned long long)(p.g + off_), m_ = ok_ ? ~0ull : 0ull;          \
            GLDS16G((const f32x4*)((a_ & m_) | (zp_ & ~m_)), (f32x4*)(Gs0 + (BUF) * GSZ) + it_ * 256 + wave * 64); \
        }                                                                                                     \
        }                                                                                                     \
    }

    // Round-4 experiment (SSIE_WGW_REGSTAGE, off): the NEXT tile through registers instead of LDS-DMA (plain sources).  An LDS-DMA piece holds
    // its wave at issue for 230-470 cycles (stamped in conv_wino4.hip) and this kernel runs ONE wave per SIMD with 20 pieces
    // per wave and position tile, so ordinary buffer loads (which do not block at issue) behind the barrier + 20 ds_write_b128 behind the
    // tile's MFMAs looked like a win; measured 0.5 % SLOWER on the step - the stall is not what bounds this kernel.
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t xreg[UP ? 1 : NX], greg[UP ? 1 : NG];
#define GW_LOADREG(TILE)                                                                                      \
    {                                                                                                         \
        int tt_ = (TILE);                                                                                     \
        const int tx_ = tt_ % p.tiles_x; tt_ /= p.tiles_x;                                                    \
        const int ty_ = tt_ % p.tiles_y, n_ = tt_ / p.tiles_y;                                                \
        const int a0_ = ty_ * GW_TH, b0_ = tx_ * GW_TW;                                                       \
        const __amdgpu_buffer_rsrc_t xr_ = GW_RSRC(p.src.ptr + (size_t)n_ * p.Hv * p.Wv * p.src.cstride,       \
                                                   (unsigned)(p.Hv * p.Wv * p.src.cstride) * 4u);             \
        const unsigned xb_ = (unsigned)((((a0_ - 1) * p.Wv + b0_ - 1) * p.src.cstride + p.src.coff + ci0) * 4); \
        const unsigned xlo_ = b0_ == 0, xn_ = min(GW_HPW, p.Wv - b0_ + 1) - xlo_;                             \
        const unsigned jn_ = (unsigned)((p.src.C - ci0 + 3) >> 2);                                            \
        const bool xin_ = xn_ == (unsigned)GW_HPW && jn_ >= (unsigned)CI4;                                    \
        _Pragma("unroll") for (int it_ = 0; it_ < NX; ++it_) {                                                \
            const unsigned id_ = it_ * 256 + tid, pix_ = id_ / CI4, j_ = id_ % CI4;                           \
            const unsigned hx_ = pix_ - ((pix_ * 3641u) >> 16) * GW_HPW;                                      \
            const bool ok_ = xin_ || (hx_ - xlo_ < xn_ && j_ < jn_);                                          \
            xreg[it_] = __builtin_amdgcn_raw_buffer_load_b128(xr_, ok_ ? xb_ + xoff[it_] : 0x80000000u, 0, 0); \
        }                                                                                                     \
        const __amdgpu_buffer_rsrc_t gr_ = GW_RSRC(p.g + (size_t)n_ * p.Ho * p.Wo * p.g_cstride,              \
                                                   (unsigned)(p.Ho * p.Wo * p.g_cstride) * 4u);               \
        const unsigned gb_ = (unsigned)(((a0_ * p.Wo + b0_) * p.g_cstride + p.g_coff + co0) * 4);              \
        const unsigned gxn_ = min(GW_TW, p.Wo - b0_);                                                         \
        const unsigned gjn_ = (unsigned)((((p.Cout + 3) & ~3) - co0 + 3) >> 2);                               \
        const bool gin_ = gxn_ == (unsigned)GW_TW && gjn_ >= (unsigned)CO4;                                   \
        _Pragma("unroll") for (int it_ = 0; it_ < NG; ++it_) {                                                \
            const unsigned id_ = it_ * 256 + tid, gx_ = (id_ / CO4) % GW_TW, j_ = id_ % CO4;                  \
            const bool ok_ = gin_ || (gx_ < gxn_ && j_ < gjn_);                                               \
            greg[it_] = __builtin_amdgcn_raw_buffer_load_b128(gr_, ok_ ? gb_ + goff[it_] : 0x80000000u, 0, 0); \
        }                                                                                                     \
    }
#define GW_STOREREG(BUF)                                                                                      \
    {                                                                                                         \
        _Pragma("unroll") for (int it_ = 0; it_ < NX; ++it_)                                                  \
            if (NX * 256 == GW_HPH * GW_HPW * CI4 || it_ * 256 + tid < GW_HPH * GW_HPW * CI4)                 \
                *((u32x4_t*)(Xs0 + (BUF) * XSZ) + it_ * 256 + tid) = xreg[it_];                               \
        _Pragma("unroll") for (int it_ = 0; it_ < NG; ++it_) *((u32x4_t*)(Gs0 + (BUF) * GSZ) + it_ * 256 + tid) = greg[it_]; \
    }
    if (tile_beg < tile_end) GW_STAGE(tile_beg, 0)
    int buf = 0;
    for (int tile = tile_beg; tile < tile_end; ++tile, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // this tile has landed; every wave is done with the other buffer
        if (tile + 1 < tile_end) {
            if (UP || !SSIE_WGW_REGSTAGE) GW_STAGE(tile + 1, buf ^ 1)
            else GW_LOADREG(tile + 1)
        }
        const float* Xs = Xs0 + buf * XSZ;
        const float* Gs = Gs0 + buf * GSZ;
        if (do_bias)
            for (int px = brow; px < GW_TH * GW_TW; px += BROWS) bsum += Gs[px * COB + bcol];
        // K loop: Winograd tile t = 0..31 of the position tile = (row t >> 3, column t & 7); one MFMA contracts the tile pair
        // (2s, 2s + 1), lane half h taking tile 2s + h; a lane transforms the pairs s and s + 1 together (x / y of a float2)
        const float* xl = Xs + mi * 32 + li;
        const float* gl = Gs + ni * 32 + li;
        for (int s = 2 * wsub; s < GW_NT / 2; s += 2 * WSPLIT) {
            const int tA = 2 * s + h, tB = tA + 2;
            const float* xa = xl + ((2 * (tA >> 3)) * GW_HPW + 2 * (tA & 7)) * CIB;
            const float* xb = xl + ((2 * (tB >> 3)) * GW_HPW + 2 * (tB & 7)) * CIB;
            const float* ga = gl + ((2 * (tA >> 3)) * GW_TW + 2 * (tA & 7)) * COB;
            const float* gb = gl + ((2 * (tB >> 3)) * GW_TW + 2 * (tB & 7)) * COB;
            // H' = G' g G'^T with G' = [[1,0],[1,1],[1,-1],[0,1]] (the 1/2 factors are applied once, after the reduction)
            f32x2 H[16];
            {
                f32x2 g[2][2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) g[a][b] = f32x2{ga[(a * GW_TW + b) * COB], gb[(a * GW_TW + b) * COB]};
                f32x2 m[4][2];
#pragma unroll
                for (int b = 0; b < 2; ++b) { m[0][b] = g[0][b]; m[1][b] = wg_add(g[0][b], g[1][b]); m[2][b] = wg_sub(g[0][b], g[1][b]); m[3][b] = g[1][b]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) { H[i * 4 + 0] = m[i][0]; H[i * 4 + 1] = wg_add(m[i][0], m[i][1]); H[i * 4 + 2] = wg_sub(m[i][0], m[i][1]); H[i * 4 + 3] = m[i][1]; }
            }
            // V = B^T d B
            f32x2 V[16];
            {
                f32x2 d[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) d[a][b] = f32x2{xa[(a * GW_HPW + b) * CIB], xb[(a * GW_HPW + b) * CIB]};
                f32x2 t[4][4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    t[0][b] = wg_sub(d[0][b], d[2][b]); t[1][b] = wg_add(d[1][b], d[2][b]);
                    t[2][b] = wg_sub(d[2][b], d[1][b]); t[3][b] = wg_sub(d[3][b], d[1][b]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    V[i * 4 + 0] = wg_sub(t[i][0], t[i][2]); V[i * 4 + 1] = wg_add(t[i][1], t[i][2]);
                    V[i * 4 + 2] = wg_sub(t[i][2], t[i][1]); V[i * 4 + 3] = wg_sub(t[i][3], t[i][1]);
                }
            }
            WG_FENCE8((&H[0])); WG_FENCE8((&H[8])); WG_FENCE8((&V[0])); WG_FENCE8((&V[8]));
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u] = MFMA32(V[u].x, H[u].x, acc[u]);
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u] = MFMA32(V[u].y, H[u].y, acc[u]);
        }
        if (!UP && SSIE_WGW_REGSTAGE && tile + 1 < tile_end) GW_STOREREG(buf ^ 1)
    }
#undef GW_LOADREG
#undef GW_STOREREG
#undef GW_STAGE
#undef GW_RSRC
#undef GW_BLDS

    if (do_bias) {
        __syncthreads();
        float* red = smem_f;
        red[brow * COB + bcol] = bsum;
        __syncthreads();
        if (tid < COB) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < BROWS; ++r) t += red[r * COB + tid];
            p.bias_slabs[(size_t)slice * p.co_pad + co0 + tid] = t;
        }
    }
    // waves that shared a block pair add their partial accumulators through LDS in fixed order, so the workgroup writes ONE
    // partial slab [slice][xi][ci_pad][co_pad]; row (M) = ci, column (N) = co
    if (WSPLIT > 1) {
        float* red = smem_f;                      // NPAIR x 16 x 16 x 64 floats <= 128 KB
        for (int w = 1; w < WSPLIT; ++w) {
            __syncthreads();
            if (wsub == w) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((pair * 16 + u) * 16 + r) * 64 + lane] = acc[u][r];
            }
            __syncthreads();
            if (wsub == 0) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[u][r] += red[((pair * 16 + u) * 16 + r) * 64 + lane];
            }
        }
        if (wsub != 0) return;
    }
    // dU -> the nine taps, per accumulator element (lane-local; once per workgroup): scale by the 1/2 factors left out of H, then
    // A^T (.) A with A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,1]].  The partial slabs are therefore [slice][tap 0..8][ci_pad][co_pad] - the
    // layout of the direct kernel's - 9/16 of the bytes of a dU slab, and the fixed-order slice reduction writes dW directly.
    float* dst0 = p.slabs + (((size_t)slice * 9) * p.ci_pad + ci0 + mi * 32) * p.co_pad + co0 + ni * 32 + li;
    const size_t tstride = (size_t)p.ci_pad * p.co_pad;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        float rr[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sj = (j == 1 || j == 2) ? 0.5f : 1.f;
            const float u0 = acc[0 * 4 + j][r] * sj, u1 = acc[1 * 4 + j][r] * (0.5f * sj), u2 = acc[2 * 4 + j][r] * (0.5f * sj), u3 = acc[3 * 4 + j][r] * sj;
            rr[0][j] = u0 + u1 + u2; rr[1][j] = u1 - u2; rr[2][j] = u1 + u2 + u3;
        }
        float* d = dst0 + (size_t)i * p.co_pad;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d[(size_t)(k * 3 + 0) * tstride] = rr[k][0] + rr[k][1] + rr[k][2];
            d[(size_t)(k * 3 + 1) * tstride] = rr[k][1] - rr[k][2];
            d[(size_t)(k * 3 + 2) * tstride] = rr[k][1] + rr[k][2] + rr[k][3];
        }
    }
}

template __global__ void conv_wgrad_wino_kernel<64, 64, false>(const WgradParams);
template __global__ void conv_wgrad_wino_kernel<32, 64, false>(const WgradParams);
template __global__ void conv_wgrad_wino_kernel<64, 32, false>(const WgradParams);
template __global__ void conv_wgrad_wino_kernel<32, 32, false>(const WgradParams);
template __global__ void conv_wgrad_wino_kernel<64, 64, true>(const WgradParams);

size_t ssie_wgrad_wino_lds_bytes(int cib, int cob)
{
    const size_t stage = (size_t)2 * (GW_HPH * GW_HPW * cib + GW_TH * GW_TW * cob) * 4;
    const int npair = (cib / 32) * (cob / 32);
    const size_t red = npair < 4 ? (size_t)npair * 16 * 16 * 64 * 4 : 0;      // in-workgroup K-split reduction scratch
    return stage > red ? stage : red;
}

int ssie_launch_wgrad_wino(const WgradParams& p, hipStream_t st)
{
    const int cib = p.ci_pad / p.ci_blocks, cob = p.co_pad / p.co_blocks;
    if (p.ntaps != 9 || p.si != 1 || p.th != GW_TH || p.hp_h != GW_HPH || p.hp_w != GW_HPW || p.min_dy != -1 || p.min_dx != -1) return 41;
    for (int t = 0; t < 9; ++t) if (p.tap_dy[t] != t / 3 - 1 || p.tap_dx[t] != t % 3 - 1) return 42;
    const bool up = p.src.sy != 1.f || p.src.sx != 1.f || p.src.Hs != p.Hv || p.src.Ws != p.Wv;
    if (up && !(cib == 64 && cob == 64)) return 43;              // the up-sampled sources of this network are 64 -> 64 layers
    const size_t lds = ssie_wgrad_wino_lds_bytes(cib, cob);
    dim3 grid(p.nslices, p.ci_blocks * p.co_blocks, 1);
    static unsigned seen[5] = {0, 0, 0, 0, 0};
#define GW_LAUNCH(CI, CO, U, K)                                                                               \
    { ssie_allow_full_lds((const void*)conv_wgrad_wino_kernel<CI, CO, U>, seen[K]);                           \
      hipLaunchKernelGGL((conv_wgrad_wino_kernel<CI, CO, U>), grid, dim3(256), lds, st, p); }
    if (up) GW_LAUNCH(64, 64, true, 4)
    else if (cib == 64 && cob == 64) GW_LAUNCH(64, 64, false, 0)
    else if (cib == 32 && cob == 64) GW_LAUNCH(32, 64, false, 1)
    else if (cib == 64 && cob == 32) GW_LAUNCH(64, 32, false, 2)
    else if (cib == 32 && cob == 32) GW_LAUNCH(32, 32, false, 3)
    else return 44;
#undef GW_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : 45;
}

