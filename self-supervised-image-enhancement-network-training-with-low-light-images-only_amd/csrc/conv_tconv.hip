// conv_tconv_kernel: a stride-2 transposed 3 x 3 convolution (ConvTranspose2d forward, model.py:39-43, and the data gradient of a
// stride-2 convolution) in ONE launch.
//   out[2a + py][2b + px][n] = sum over the taps of output-parity class (py, px) of in[a + dy][b + dx][k] * W(k, n, tap)
// The four parity classes have 1 / 2 / 2 / 4 taps with offsets dy, dx in {0, 1}: without zero insertion they are four small
// convolutions of the SAME input.  Run as four launches (conv_fprop_v2_kernel) each stages the input halo again and has a 1-4 tap
// K loop per staged chunk (45-90 TFLOP/s); here one workgroup stages the 17 x 17 halo of a 16 x 16 input tile once and runs all
// nine taps over it into four accumulator sets (one per parity class), then writes the four interleaved 16 x 16 output grids.
// Structure of conv_fprop_v2_kernel otherwise (conv_fprop_v2.hip): 512 threads, 8 waves = 4 (M) x 2 (N), 64 output channels,
// LDS double-buffered by global->LDS DMA, one barrier per 16-channel chunk, persistent workgroups with a dynamic tile queue,
// same packed-weight layout (the nine taps in class order), same fused epilogue.
#include "conv_device.h"

__device__ f32x4 tconv_zero_page[4];   // zero-initialised: source of padding slots

#define GLDS16T(gptr, lptr)                                                                            \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

namespace {
constexpr int TC_TW = 16, TC_HP = 17;                                              // tile width / halo row pitch (pixels)
constexpr int TC_BSZ_MAX = 9 * 4 * 64;                                             // float4 per weight chunk (9 taps x 16 ci x 64 co), the 64-channel form
__device__ __forceinline__ constexpr int tc_class(int t) { return t == 0 ? 0 : t < 3 ? 1 : t < 5 ? 2 : 3; }
}

// TC_TH = 16 input rows per tile, or 8 (ssie_conv_to_tconv: launches whose 16-row tiles would leave more than half the CUs idle - the
// reference's shipped batch of 2 patches: 32 tiles of 38 MFLOP each were 83 us per launch; 64 half tiles ~45)
// NWN = waves along the output channels: 2 (a 64-channel tile, 8 waves), or 1 (32 channels, 4 waves) for launches that the 8-row tiles
// still leave under-filled (batch 2: 64 tiles -> 128 workgroups, each staging the halo for half the channels)
template <int EPI, bool RAG, int TC_TH = 16, int NWN = 2>       // epilogue shape (ssie_epi_shape) / some tile sticks out of the output: see conv_fprop_v2_kernel
__global__ __launch_bounds__(256 * NWN, 2) void conv_tconv_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int NTHR = 256 * NWN, NW = 4 * NWN, BN = 32 * NWN, MT = TC_TH / 8;
    constexpr int TC_HP4 = (TC_TH + 1) * TC_HP * 4;                                // 1156 (612) 16-byte slots per halo tile
    constexpr int TC_NA = (TC_HP4 + NTHR - 1) / NTHR;                              // DMA slots per lane
    constexpr int TC_BSZ = 9 * 4 * BN;                                             // float4 per weight chunk (9 taps x 16 ci x BN co)
    f32x4* As0 = (f32x4*)smem_f;                    // [2][TC_HP4]
    f32x4* Bs0 = As0 + 2 * TC_HP4;                  // [2][TC_BSZ]
    int* s_next = (int*)(Bs0 + 2 * TC_BSZ);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, li = lane & 31;
    const int wn = NWN == 2 ? (wave & 1) : 0, wm = NWN == 2 ? (wave >> 1) : wave;

    // byte offset of this lane's A fragment (k-quad 0; k-quad 1 = ^32) for (M-tile m, tap t): tile-invariant
    int aaddr[9][MT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int hp = (2 * (wm * MT + m) + (li >> 4) + (int)p.tap_dy[t]) * TC_HP + (li & 15) + (int)p.tap_dx[t];
            aaddr[t][m] = (hp * 4 + (h ^ ssie_swz(hp))) * 16;
        }
    const int nsteps = p.nchunks;
    const int total_tiles = p.N * p.tiles_y * p.tiles_x * (NWN == 2 ? 1 : 2);
    // (NWN = 1: the two 32-channel halves of a position tile are neighbouring tile ids)
#define TC_DECODE(T, N_, A0_, B0_, CO0_)                                                  \
    {                                                                                     \
        int q_ = (T);                                                                     \
        CO0_ = 0;                                                                         \
        if (NWN == 1) { CO0_ = (q_ & 1) * 32; q_ >>= 1; }                                 \
        B0_ = (q_ % p.tiles_x) * TC_TW; q_ /= p.tiles_x;                                  \
        A0_ = (q_ % p.tiles_y) * TC_TH; N_ = q_ / p.tiles_y;                              \
    }
    // DMA of chunk CHUNK of tile (N_, A0_, B0_): halo tile (slot id = i*NTHR + tid holds channel quad (id&3) ^ swz(pixel)) + weights
#define TC_PREFETCH(CHUNK, N_, A0_, B0_, CO0_, BUF)                                                               \
    {                                                                                                         \
        f32x4* abuf_ = As0 + (BUF) * TC_HP4;                                                                  \
        const unsigned long long zp_ = (unsigned long long)tconv_zero_page;                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < TC_NA; ++i_) {                                                \
            const int id_ = i_ * NTHR + tid;                                                                  \
            const int pix_ = id_ >> 2, hy_ = (pix_ * 3856) >> 16, hx_ = pix_ - hy_ * TC_HP;   /* / 17 for pix < 4000 */ \
            const int j_ = (id_ & 3) ^ ssie_swz(pix_);                                                        \
            const int vy_ = (A0_) + hy_, vx_ = (B0_) + hx_, c_ = (CHUNK) * SSIE_CK + 4 * j_;                  \
            const bool ok_ = vy_ < p.Hv && vx_ < p.Wv && c_ < p.src[0].C;                                     \
            const unsigned off_ = (unsigned)(((N_) * p.Hv + vy_) * p.Wv + vx_) * (unsigned)p.src[0].cstride + (unsigned)(p.src[0].coff + c_); \
            const unsigned long long a_ = (unsigned long long)(p.src[0].ptr + off_), m_ = ok_ ? ~0ull : 0ull;  \
            if (i_ + 1 < TC_NA || id_ < TC_HP4) GLDS16T((const f32x4*)((a_ & m_) | (zp_ & ~m_)), abuf_ + i_ * NTHR + wave * 64); \
        }                                                                                                     \
        const f32x4* wsrc_ = (const f32x4*)p.wpacked + (size_t)(CHUNK) * 9 * 4 * p.Cout_pad + (CO0_);         \
        f32x4* bbuf_ = Bs0 + (BUF) * TC_BSZ;                                                                  \
        /* a piece = 64 float4 = one (tap, k-quad) row of 64 channels, or two rows of 32 */                   \
        for (int q_ = wave; q_ < 9 * 4 * BN / 64; q_ += NW)                                                   \
            GLDS16T(wsrc_ + (size_t)(NWN == 2 ? q_ : 2 * q_ + (lane >> 5)) * p.Cout_pad + (NWN == 2 ? lane : (lane & 31)), bbuf_ + q_ * 64); \
    }

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int n, a0, b0, co0;
    TC_DECODE(tile, n, a0, b0, co0)
    int gstep = 0;
    TC_PREFETCH(0, n, a0, b0, co0, 0)
    int fetched = 0x7fffffff;

    while (tile < total_tiles) {
        f32x16 acc[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][c][r] = 0.f;
        const float bv = (p.bias && co0 + wn * 32 + li < p.Cout) ? p.bias[co0 + wn * 32 + li] : 0.f;
        int ntile = 0x7fffffff;
        int nn = n, na0 = a0, nb0 = b0, nco0 = co0;

        for (int step = 0; step < nsteps; ++step, ++gstep) {
            const int buf = gstep & 1;
            if (tid == 0) {
                if (nsteps == 1 || !p.tile_counter) {
                    if (step == 0) *s_next = p.tile_counter ? (int)gridDim.x + atomicAdd(p.tile_counter, 1) : tile + (int)gridDim.x;
                } else if (step == 1) *s_next = fetched;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (step == (nsteps > 1 ? 1 : 0)) {
                ntile = *s_next;
                if (ntile < total_tiles) TC_DECODE(ntile, nn, na0, nb0, nco0)
            }
            if (step + 1 < nsteps) TC_PREFETCH(step + 1, n, a0, b0, co0, buf ^ 1)
            else if (ntile < total_tiles) TC_PREFETCH(0, nn, na0, nb0, nco0, buf ^ 1)

            const char* Ab = (const char*)(As0 + buf * TC_HP4);
            const f32x4* Bl = Bs0 + buf * TC_BSZ + h * BN + wn * 32 + li;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int c = tc_class(t);
#pragma unroll
                for (int kq = 0; kq < 2; ++kq) {
                    const f32x4 bf = Bl[(t * 4 + kq * 2) * BN];
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const f32x4 af = *(const f32x4*)(Ab + (kq ? (aaddr[t][m] ^ 32) : aaddr[t][m]));
                        acc[m][c] = MFMA32(af.x, bf.x, acc[m][c]); acc[m][c] = MFMA32(af.y, bf.y, acc[m][c]);
                        acc[m][c] = MFMA32(af.z, bf.z, acc[m][c]); acc[m][c] = MFMA32(af.w, bf.w, acc[m][c]);
                    }
                }
            }
            if (tid == 0 && step == 0 && nsteps > 1 && p.tile_counter)
                fetched = (int)gridDim.x + atomicAdd(p.tile_counter, 1);
        }

        // epilogue: class c = (py, px) writes out[2a + py][2b + px]; wave (wm, wn) holds tile rows 4wm .. 4wm+3 (M-tile m = rows
        // 2(wm*MT+m), +1), channels 32wn .. 32wn+31
        {
            const int co = co0 + wn * 32 + li;
            if (co < p.Cout) {
                const long rowstride = 2L * p.Wout * p.out_cstride, pixstride = 2L * p.out_cstride;
                // (written out per (class, M-tile): left as loops hipcc keeps them rolled - the fused epilogue is large - and indexes
                // the accumulators dynamically, i.e. through scratch)
#define TC_EPI(C, M)                                                                                          \
                {                                                                                             \
                    constexpr int py_ = (C) >> 1, px_ = (C) & 1;                                              \
                    const bool full_ = !RAG || (2 * (a0 + TC_TH - 1) + py_ < p.Hout && 2 * (b0 + TC_TW - 1) + px_ < p.Wout); \
                    const int arow_ = a0 + 2 * (wm * MT + (M)), bcol_ = b0 + 4 * h;                           \
                    const size_t o0_ = ((size_t)(n * p.Hout + 2 * arow_ + py_) * p.Wout + 2 * bcol_ + px_) * p.out_cstride + p.out_coff + co; \
                    if (full_) ssie_epilogue_full<EPI>(p, acc[M][C], o0_, rowstride, pixstride, bv);          \
                    else {                                                                                    \
                        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                   \
                            const int tr_ = r_ >> 3, tcn_ = (r_ & 3) + 8 * ((r_ >> 2) & 1);                   \
                            if (2 * (arow_ + tr_) + py_ >= p.Hout || 2 * (bcol_ + tcn_) + px_ >= p.Wout) continue; \
                            const size_t o_ = o0_ + tr_ * rowstride + tcn_ * pixstride;                       \
                            float v_ = acc[M][C][r_] + bv;                                                    \
                            if (p.act == ACT_RELU) v_ = fmaxf(v_, 0.f);                                       \
                            else if (p.act == ACT_SIGMOID) v_ = 1.f / (1.f + expf(-v_));                      \
                            if (p.mask_mode == MASK_RELU) v_ = p.mask_y[o_] > 0.f ? v_ : 0.f;                 \
                            else if (p.mask_mode == MASK_SIGMOID) { const float y_ = p.mask_y[o_]; v_ *= y_ * (1.f - y_); } \
                            if (p.out2) p.out2[o_] = v_;                                                      \
                            if (p.addsrc) v_ += p.addsrc[o_];                                                 \
                            if (p.accumulate) v_ += p.out[o_];                                                \
                            p.out[o_] = v_;                                                                   \
                        }                                                                                     \
                    }                                                                                         \
                }
                TC_EPI(0, 0) TC_EPI(1, 0) TC_EPI(2, 0) TC_EPI(3, 0)
                if constexpr (MT == 2) { TC_EPI(0, 1) TC_EPI(1, 1) TC_EPI(2, 1) TC_EPI(3, 1) }
#undef TC_EPI
            }
        }
        n = nn; a0 = na0; b0 = nb0; co0 = nco0; tile = ntile;
    }
#undef TC_PREFETCH
#undef TC_DECODE
}

size_t ssie_tconv_lds_bytes(int th) { return (size_t)(2 * (th + 1) * TC_HP * 4 + 2 * TC_BSZ_MAX) * 16 + 64; }

int ssie_tconv_split_below = 256;      // 8-row-tile launches with fewer tiles than this: 32-channel workgroups (NWN = 1)
extern "C" void ssie_debug_set_tconv_split_below(int v) { ssie_tconv_split_below = v; }

// p from ssie_make_conv over the nine taps in class order (ssie_taps_transposed_all), si = 1, so = 2, single 1:1 source, 64 outputs
int ssie_launch_tconv(const ConvParams& p, hipStream_t st)
{
    if (p.ntaps != 9 || p.si != 1 || p.so != 2 || p.nsrc != 1 || p.Cout_pad != 64 || (p.th != 16 && p.th != 8) || p.tw != TC_TW ||
        p.hp_h != p.th + 1 || p.hp_w != TC_HP || p.min_dy != 0 || p.min_dx != 0) return 51;
    if (p.src[0].sy != 1.f || p.src[0].sx != 1.f || p.src[0].Hs != p.Hv || p.src[0].Ws != p.Wv) return 52;
    if (p.tiles_y != ssie_ceil_div(p.Ho, p.th) || p.tiles_x != ssie_ceil_div(p.Wo, TC_TW)) return 51;
    const size_t tiles1 = (size_t)p.N * p.tiles_y * p.tiles_x;
    const bool rag = p.Ho % p.th != 0 || p.Wo % TC_TW != 0 || p.Hout != 2 * p.Ho || p.Wout != 2 * p.Wo;
    const bool split = p.th == 8 && !rag && (long)tiles1 < ssie_tconv_split_below;       // whole-tile launches only (fewer instantiations)
    const size_t tiles = split ? 2 * tiles1 : tiles1;
    const dim3 grid((unsigned)(tiles < 256 ? tiles : 256));
    const int epi = ssie_epi_shape(p);
    static unsigned seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TC_GO(E, R, TH_, SLOT) { ssie_allow_full_lds((const void*)conv_tconv_kernel<E, R, TH_>, seen[SLOT]); \
                                 hipLaunchKernelGGL((conv_tconv_kernel<E, R, TH_>), grid, dim3(512), ssie_tconv_lds_bytes(TH_), st, p); }
#define TC_PICK(TH_, B) { if (rag) TC_GO(0, true, TH_, B) else if (epi == 1) TC_GO(1, false, TH_, B + 1) else if (epi == 2) TC_GO(2, false, TH_, B + 2) else TC_GO(0, false, TH_, B + 3) }
    if (split) {
        static unsigned seen1[3] = {0, 0, 0};
#define TC_GO1(E, SLOT) { ssie_allow_full_lds((const void*)conv_tconv_kernel<E, false, 8, 1>, seen1[SLOT]); \
                          hipLaunchKernelGGL((conv_tconv_kernel<E, false, 8, 1>), grid, dim3(256), ssie_tconv_lds_bytes(8), st, p); }
        if (epi == 1) TC_GO1(1, 0) else if (epi == 2) TC_GO1(2, 1) else TC_GO1(0, 2)
#undef TC_GO1
    } else if (p.th == 16) TC_PICK(16, 0) else TC_PICK(8, 4)
#undef TC_PICK
#undef TC_GO
    return hipGetLastError() == hipSuccess ? 0 : 53;
}
