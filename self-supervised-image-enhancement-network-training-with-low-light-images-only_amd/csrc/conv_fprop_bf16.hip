// conv_fprop_bf16_kernel: the enhance-only (inference) convolution with bf16 storage and bf16 MFMA, fp32 accumulate
// (BASELINE.json configs[4]: "bf16 mixed precision + full-resolution inference").
//
// Same implicit-GEMM structure as conv_fprop_v2_kernel (one 512-thread workgroup per CU, LDS double-buffered by
// global->LDS DMA, one barrier per step, persistent workgroups on a dynamic tile queue), with these differences:
//   * activations are bf16 NHWC in memory (channel stride a multiple of 8); a K-chunk is 32 channels, so a pixel of the
//     halo tile is still 64 B = four 16-byte slots (slot j = channels 8j .. 8j+7) and the DMA / swizzle code is unchanged
//   * one v_mfma_f32_32x32x16_bf16 consumes a 16-channel half chunk: lane (i, h) feeds slot 2*sc + h of pixel / column i
//   * packed weights: [chunk32][tap][slot][Cout_pad][8 bf16] - byte-for-byte the fp32 pack's shape
//   * the input stride may be 2 (tile of 8 x 16 positions) - the fp32 v2 kernel is stride-1 only
//   * epilogue: bias + ReLU / sigmoid, optional bf16 skip-add, output as bf16 or fp32, optional second (bf16) copy
// Accumulation order per output: channels ascending within a tap, taps in list order within a chunk - like the fp32 path.
#include "conv_device.h"

__device__ f32x4 ssie_zero_page_h[4];   // zero-initialised: source of padding slots

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// Ablation builds only (tools/build_variants.py: -DSSIE_X_NOMFMA / _NODMA / _NOSTORE): one phase of the kernel removed to see
// what the others cost.  The shipped library defines none of them.
#ifdef SSIE_X_NOMFMA
#define MFMA_BF16(a, b, c) ({ asm volatile("" :: "v"(a), "v"(b)); (c); })
#else
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, (a)), __builtin_bit_cast(bf16x8_t, (b)), (c), 0, 0, 0)
#endif

#ifdef SSIE_X_NODMA
#define GLDS16(gptr, lptr) do { asm volatile("" :: "v"(gptr), "v"(lptr)); } while (0)
#else
#define GLDS16(gptr, lptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
#endif
#ifdef SSIE_X_NOSTORE
#define SSIE_X_KEEP(okr, val) ((okr) && (val) == 1234.56789f)
#else
#define SSIE_X_KEEP(okr, val) (okr)
#endif

__device__ __forceinline__ float ssie_bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// source address of 8 consecutive bf16 channels of virtual pixel (n, vy, vx), or the zero page
__device__ __forceinline__ const f32x4* ssie_virtual_addr_h(const SrcSel& s, bool up, int n, int vy, int vx, int Hv, int Wv, int c)
{
    const bool ok = (unsigned)vy < (unsigned)Hv && (unsigned)vx < (unsigned)Wv && c < s.C;
    int y = vy, x = vx;
    if (up) {
        const int cy = min(max(vy, 0), Hv - 1), cx = min(max(vx, 0), Wv - 1);
        y = min((int)floorf((float)cy * s.sy), s.Hs - 1);
        x = min((int)floorf((float)cx * s.sx), s.Ws - 1);
    }
    const unsigned off = (unsigned)((n * s.Hs + y) * s.Ws + x) * (unsigned)s.cstride + (unsigned)(s.coff + c);
    return ok ? (const f32x4*)((const unsigned short*)s.ptr + off) : (const f32x4*)ssie_zero_page_h;
}

__device__ __forceinline__ uint2 ssie_pack4bf(const f32x4& v) { return make_uint2(ssie_pack2bf(v[0], v[1]), ssie_pack2bf(v[2], v[3])); }
__device__ __forceinline__ f32x4 ssie_unpack4bf(uint2 u)
{
    f32x4 r = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
    return r;
}

// one 32 x 32 accumulator tile -> memory, TRANSPOSED layout (the MFMAs are issued as D^T = W x X; see ssie_epilogue_t in
// conv_device.h): lane (li, h) = output position li of the 2 x 16 M-tile, register r = channel 8*(r>>2) + 4h + (r&3).  Four
// groups of four consecutive channels per lane: bf16 tensors move 8 bytes per group, fp32 outputs 16.
//   opix = element offset of (this lane's pixel, channel out_coff); c0 = first channel of the N-tile + 4h
__device__ __forceinline__ void ssie_epilogue_ht(const ConvParams& p, const f32x16& acc, size_t opix, int c0, bool pos_ok)
{
    if (!pos_ok) return;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[4];
    bool full[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = c0 + 8 * g;
        full[g] = c + 4 <= p.Cout;
        const f32x4 b = (p.bias && full[g]) ? *(const f32x4*)(p.bias + c) : z4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = acc[4 * g + j] + b[j];
            if (p.act == ACT_RELU) t = fmaxf(t, 0.f);
            else if (p.act == ACT_SIGMOID) t = 1.f / (1.f + expf(-t));
            v[g][j] = t;
        }
    }
    if (p.out2) {
        unsigned short* o2 = (unsigned short*)p.out2 + opix + c0;
#pragma unroll
        for (int g = 0; g < 4; ++g) if (SSIE_X_KEEP(full[g], v[g][0])) *(uint2*)(o2 + 8 * g) = ssie_pack4bf(v[g]);
    }
    if (p.addsrc) {
        const unsigned short* ap = (const unsigned short*)p.addsrc + opix + c0;
        uint2 a[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) a[g] = full[g] ? *(const uint2*)(ap + 8 * g) : make_uint2(0u, 0u);
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] += ssie_unpack4bf(a[g]);
    }
    if (p.out_bf16) {
        unsigned short* ob = (unsigned short*)p.out + opix + c0;
#pragma unroll
        for (int g = 0; g < 4; ++g) if (SSIE_X_KEEP(full[g], v[g][0])) *(uint2*)(ob + 8 * g) = ssie_pack4bf(v[g]);
    } else {
        float* ob = p.out + opix + c0;
#pragma unroll
        for (int g = 0; g < 4; ++g) if (SSIE_X_KEEP(full[g], v[g][0])) *(f32x4*)(ob + 8 * g) = v[g];
    }
    // the group that straddles Cout (Cout % 4 != 0, e.g. the 1-channel final_conv), element by element
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = c0 + 8 * g;
        if (full[g] || c >= p.Cout) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (c + j >= p.Cout) break;
            const size_t o = opix + c + j;
            float t = acc[4 * g + j] + (p.bias ? p.bias[c + j] : 0.f);
            if (p.act == ACT_RELU) t = fmaxf(t, 0.f);
            else if (p.act == ACT_SIGMOID) t = 1.f / (1.f + expf(-t));
            if (p.out2) ((unsigned short*)p.out2)[o] = ssie_f2bf(t);
            if (p.addsrc) t += ssie_bf2f(((const unsigned short*)p.addsrc)[o]);
            if (p.out_bf16) ((unsigned short*)p.out)[o] = ssie_f2bf(t); else p.out[o] = t;
        }
    }
}

template <int NT, int NA2, int TH>
__global__ __launch_bounds__(512, 2) void conv_fprop_bf16_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int NW = 8, NTHR = 64 * NW, CKH = 32;
    constexpr int BN = 32 * NT;
    constexpr int WM = (NT == 2) ? NW / 2 : NW;     // waves along M; a TH x 16 tile has TH/2 M-tiles of 2 x 16 positions
    constexpr int MT = (TH / 2) / WM;
    static_assert(MT >= 1, "tile too small for the wave grid");
    constexpr int BSZ = SSIE_TG * 4 * BN;           // 16-byte slots per B buffer
    const int HP = p.hp_h * p.hp_w, HP4 = HP * 4;
    f32x4* As0 = (f32x4*)smem_f;                    // [2][HP4]
    f32x4* Bs0 = As0 + 2 * HP4;                     // [2][BSZ]
    int* tapoff = (int*)(Bs0 + 2 * BSZ);
    int* s_next = tapoff + SSIE_MAX_TAPS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, li = lane & 31;
    const int wn = (NT == 1) ? 0 : (wave & 1);
    const int wm = (NT == 1) ? wave : (wave >> 1);

    for (int t = tid; t < p.ntaps; t += NTHR)
        tapoff[t] = ((int)p.tap_dy[t] - p.min_dy) * p.hp_w + ((int)p.tap_dx[t] - p.min_dx);

    int pixbase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mt = wm * MT + m;
        pixbase[m] = (2 * mt + (li >> 4)) * p.si * p.hp_w + (li & 15) * p.si;
    }
    const int ngroups = (p.ntaps + SSIE_TG - 1) / SSIE_TG;
    const int nsteps = p.nchunks * ngroups;
    const int total_tiles = p.N * p.tiles_y * p.tiles_x * p.co_blocks;

    // this lane's halo slots: LDS slot id = i*NTHR + tid (linear), it holds channel octet j = (id&3) ^ swz(pixel)
    int ahy[NA2], ahx[NA2], aj[NA2];
#pragma unroll
    for (int i = 0; i < NA2; ++i) {
        const int id = min(tid + i * NTHR, HP4 - 1);
        const int pix = id >> 2;
        ahy[i] = pix / p.hp_w; ahx[i] = pix - ahy[i] * p.hp_w; aj[i] = (id & 3) ^ ssie_swz(pix);
    }

#define H_DECODE(T, N_, A0_, B0_, CO0_)                                                   \
    {                                                                                     \
        int q_ = (T);                                                                     \
        CO0_ = (q_ % p.co_blocks) * BN; q_ /= p.co_blocks;                                \
        B0_ = (q_ % p.tiles_x) * SSIE_TW; q_ /= p.tiles_x;                                \
        A0_ = (q_ % p.tiles_y) * TH; N_ = q_ / p.tiles_y;                                 \
    }
#define H_PREFETCH(CHUNK, G, N_, A0_, B0_, CO0_, BUF, ABUF)                                                   \
    {                                                                                                         \
        if ((G) == 0) {                                                                                       \
            const SrcSel s_ = ssie_pick_src(p, (CHUNK) * CKH);                                                \
            const bool up_ = s_.sy != 1.f || s_.sx != 1.f;                                                    \
            const int vy0_ = (A0_) * p.si + p.min_dy, vx0_ = (B0_) * p.si + p.min_dx;                         \
            f32x4* abuf_ = As0 + (ABUF) * HP4;                                                                \
            _Pragma("unroll") for (int i_ = 0; i_ < NA2; ++i_) {                                              \
                if (tid + i_ * NTHR < HP4) {                                                                  \
                    const f32x4* g_ = ssie_virtual_addr_h(s_, up_, (N_), vy0_ + ahy[i_], vx0_ + ahx[i_], p.Hv, p.Wv, \
                                                          (CHUNK) * CKH + 8 * aj[i_] - s_.cbeg);              \
                    GLDS16(g_, abuf_ + i_ * NTHR + wave * 64);                                                \
                }                                                                                             \
            }                                                                                                 \
        }                                                                                                     \
        const int t0_ = (G) * SSIE_TG;                                                                        \
        const int pieces_ = min(SSIE_TG, p.ntaps - t0_) * 4 * BN / 64;                                        \
        const f32x4* wsrc_ = (const f32x4*)p.wpacked + ((size_t)((CHUNK) * p.ntaps + t0_) * 4) * p.Cout_pad + (CO0_); \
        f32x4* bbuf_ = Bs0 + (BUF) * BSZ;                                                                     \
        for (int q_ = wave; q_ < pieces_; q_ += NW) {                                                         \
            const int slot_ = q_ * 64 + lane;                                                                 \
            GLDS16(wsrc_ + (size_t)(slot_ / BN) * p.Cout_pad + (slot_ % BN), bbuf_ + q_ * 64);                \
        }                                                                                                     \
    }

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int n, a0, b0, co0;
    H_DECODE(tile, n, a0, b0, co0)
    int gstep = 0;
    int a_cur = 0;
    H_PREFETCH(0, 0, n, a0, b0, co0, 0, 0)
    int fetched = 0x7fffffff;

    while (tile < total_tiles) {
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        int ntile = 0x7fffffff;
        int nn = n, na0 = a0, nb0 = b0, nco0 = co0;

        int chunk = 0, g = 0;
        for (int step = 0; step < nsteps; ++step, ++gstep) {
            const int buf = gstep & 1;
            const int t0 = g * SSIE_TG;
            const int tg = min(SSIE_TG, p.ntaps - t0);
            if (tid == 0) {
                if (nsteps == 1 || !p.tile_counter) {
                    if (step == 0) *s_next = p.tile_counter ? (int)gridDim.x + atomicAdd(p.tile_counter, 1) : tile + (int)gridDim.x;
                } else if (step == 1) *s_next = fetched;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (step == (nsteps > 1 ? 1 : 0)) {
                ntile = *s_next;
                if (ntile < total_tiles) H_DECODE(ntile, nn, na0, nb0, nco0)
            }
            int nchunk = chunk, ng = g + 1;
            if (ng == ngroups) { ng = 0; ++nchunk; }
            const bool more = step + 1 < nsteps;
            const int a_nxt = ((more ? ng : 0) == 0) ? (a_cur ^ 1) : a_cur;
            if (more) H_PREFETCH(nchunk, ng, n, a0, b0, co0, buf ^ 1, a_nxt)
            else if (ntile < total_tiles) H_PREFETCH(0, 0, nn, na0, nb0, nco0, buf ^ 1, a_nxt)

            const f32x4* As = As0 + a_cur * HP4;
            const f32x4* Bs = Bs0 + buf * BSZ;
            for (int tl = 0; tl < tg; ++tl) {
                const int off = tapoff[t0 + tl];
#pragma unroll
                for (int sc = 0; sc < 2; ++sc) {
                    const f32x4 bf = Bs[(tl * 4 + sc * 2 + h) * BN + wn * 32 + li];
                    f32x4 af[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int hp = pixbase[m] + off;
                        af[m] = As[hp * 4 + ((sc * 2 + h) ^ ssie_swz(hp))];
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = MFMA_BF16(bf, af[m], acc[m]);
                }
            }
            if (tid == 0 && step == 0 && nsteps > 1 && p.tile_counter)
                fetched = (int)gridDim.x + atomicAdd(p.tile_counter, 1);
            chunk = nchunk; g = ng; a_cur = a_nxt;
        }

        // epilogue: D^T = W x X above, so lane li = position of the 2 x 16 M-tile, registers = channels (ssie_epilogue_ht)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int mt = wm * MT + m;
            bool ok;
            const size_t opix = ssie_epilogue_pos(p, n, a0 + 2 * mt, b0, li, ok);
            ssie_epilogue_ht(p, acc[m], opix, co0 + wn * 32 + 4 * h, ok);
        }
        n = nn; a0 = na0; b0 = nb0; co0 = nco0; tile = ntile;
    }
#undef H_PREFETCH
#undef H_DECODE
}

// Wide variant for the 64-channel stride-1 layers with <= 9 taps: tile = 16 x 32 positions x 64 channels, every wave owns a
// 64-position x 64-channel register tile (4 accumulators) so one (tap, half chunk) costs 4 ds_read_b128 per 4 MFMAs instead of
// 3 per 2 - the bf16 MFMA drains operands 16x faster than the fp32 one and the kernel is LDS-bandwidth-bound - and a step
// moves 76 KB of DMA for twice the FLOPs of the 16 x 16 tile's 57.6 KB.  A-fragment addresses are tile-invariant and precomputed.
template <int NA2, bool SINGLE>
__global__ __launch_bounds__(512, 2) void conv_fprop_bf16w_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int NT = 2, NW = 8, BN = 64, TH = 16, TWW = 32, CKH = 32;
    constexpr int NTHR = 64 * NW;
    constexpr int MT = 2;                           // wave w owns tile rows 2w, 2w+1: M-tile m = its columns 16m .. 16m+15
    constexpr int BSZ = SSIE_TG * 4 * BN;           // float4 per B buffer
    const int HP = p.hp_h * p.hp_w, HP4 = HP * 4;
    f32x4* As0 = (f32x4*)smem_f;                    // [2][HP4]
    f32x4* Bs0 = As0 + 2 * HP4;                     // [2][BSZ]
    int* tapoff = (int*)(Bs0 + 2 * BSZ);
    int* s_next = tapoff + SSIE_MAX_TAPS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, li = lane & 31;

    for (int t = tid; t < p.ntaps; t += NTHR)
        tapoff[t] = ((int)p.tap_dy[t] - p.min_dy) * p.hp_w + ((int)p.tap_dx[t] - p.min_dx);

    int pixbase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        pixbase[m] = (2 * wave + (li >> 4)) * p.hp_w + 16 * m + (li & 15);
    }
    // the wide kernel only runs layers with <= 9 taps (one tap group), so every A-fragment address is tile-invariant:
    // byte offset inside a halo buffer of (M-tile m, tap t, k-quad 0); k-quad 1 is the same address ^ 32 (slot index ^ 2).
    // This takes the ~7 address VALU ops per 128-bit fragment read out of the MFMA loop (energy per FLOP).
    int aaddr[SSIE_TG][MT];
#pragma unroll
    for (int t = 0; t < SSIE_TG; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int hp = pixbase[m] + ((int)p.tap_dy[t] - p.min_dy) * p.hp_w + ((int)p.tap_dx[t] - p.min_dx);
            aaddr[t][m] = t < p.ntaps ? (hp * 4 + (h ^ ssie_swz(hp))) * 16 : 0;
        }
    const int ngroups = (p.ntaps + SSIE_TG - 1) / SSIE_TG;
    const int nsteps = p.nchunks * ngroups;
    const int total_tiles = p.N * p.tiles_y * p.tiles_x * p.co_blocks;

    // this lane's halo slots: LDS slot id = i*NTHR + tid (linear), it holds channel quad j = (id&3) ^ swz(pixel)
    int ahy[NA2], ahx[NA2], aj[NA2];
#pragma unroll
    for (int i = 0; i < NA2; ++i) {
        const int id = min(tid + i * NTHR, HP4 - 1);
        const int pix = id >> 2;
        ahy[i] = pix / p.hp_w; ahx[i] = pix - ahy[i] * p.hp_w; aj[i] = (id & 3) ^ ssie_swz(pix);
    }

#define HW_DECODE(T, N_, A0_, B0_, CO0_)                                                  \
    {                                                                                     \
        int q_ = (T);                                                                     \
        CO0_ = (q_ % p.co_blocks) * BN; q_ /= p.co_blocks;                                \
        B0_ = (q_ % p.tiles_x) * TWW; q_ /= p.tiles_x;                                  \
        A0_ = (q_ % p.tiles_y) * TH; N_ = q_ / p.tiles_y;                                 \
    }
    // DMA the operands of step (CHUNK, G) of tile (N_, A0_, B0_, CO0_): weights into B buffer BUF, and (first tap group
    // of a chunk only) the halo tile into A buffer ABUF
#define HW_PREFETCH(CHUNK, G, N_, A0_, B0_, CO0_, BUF, ABUF)                                                        \
    {                                                                                                         \
        if ((G) == 0) {                                                                                       \
            const SrcSel s_ = SINGLE ? ssie_only_src(p) : ssie_pick_src(p, (CHUNK) * CKH);                \
            const bool up_ = s_.sy != 1.f || s_.sx != 1.f;                                                    \
            const int vy0_ = (A0_) + p.min_dy, vx0_ = (B0_) + p.min_dx;                                       \
            f32x4* abuf_ = As0 + (ABUF) * HP4;                                                                \
            _Pragma("unroll") for (int i_ = 0; i_ < NA2; ++i_) {                                              \
                if (tid + i_ * NTHR < HP4) {                                                                  \
                    const f32x4* g_ = ssie_virtual_addr_h(s_, up_, (N_), vy0_ + ahy[i_], vx0_ + ahx[i_], p.Hv, p.Wv, \
                                                          (CHUNK) * CKH + 8 * aj[i_] - s_.cbeg);              \
                    GLDS16(g_, abuf_ + i_ * NTHR + wave * 64);                                                \
                }                                                                                             \
            }                                                                                                 \
        }                                                                                                     \
        const int t0_ = (G) * SSIE_TG;                                                                        \
        const int pieces_ = min(SSIE_TG, p.ntaps - t0_) * 4 * BN / 64;                                        \
        const f32x4* wsrc_ = (const f32x4*)p.wpacked + ((size_t)((CHUNK) * p.ntaps + t0_) * 4) * p.Cout_pad + (CO0_); \
        f32x4* bbuf_ = Bs0 + (BUF) * BSZ;                                                                     \
        for (int q_ = wave; q_ < pieces_; q_ += NW) {                                                         \
            const int slot_ = q_ * 64 + lane;                                                                 \
            GLDS16(wsrc_ + (size_t)(slot_ / BN) * p.Cout_pad + (slot_ % BN), bbuf_ + q_ * 64);                \
        }                                                                                                     \
    }

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int n, a0, b0, co0;
    HW_DECODE(tile, n, a0, b0, co0)
    int gstep = 0;
    int a_cur = 0;          // A buffer holding the halo tile of the step about to be computed
    HW_PREFETCH(0, 0, n, a0, b0, co0, 0, 0)
    int fetched = 0x7fffffff;
    // The two waves of a SIMD (w and w+4) run the same program; issuing the next step's DMA (address VALU work) at
    // the same moment would leave the SIMD's MFMA pipe idle.  Waves 4-7 therefore issue it in the middle of their
    // tap loop while waves 0-3 issue it up front.
    const bool late_prefetch = false;   // measured: issuing the DMA inside the tap loop (waves 4-7) was 5-25 % SLOWER

    while (tile < total_tiles) {
        f32x16 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][c][r] = 0.f;
        int ntile = 0x7fffffff;
        int nn = n, na0 = a0, nb0 = b0, nco0 = co0;

        int chunk = 0, g = 0;
        for (int step = 0; step < nsteps; ++step, ++gstep) {
            const int buf = gstep & 1;
            const int t0 = g * SSIE_TG;
            const int tg = min(SSIE_TG, p.ntaps - t0);
            // dynamic tile queue: the counter is drawn one step ahead and handed over through LDS across this barrier
            if (tid == 0) {
                if (nsteps == 1 || !p.tile_counter) {
                    if (step == 0) *s_next = p.tile_counter ? (int)gridDim.x + atomicAdd(p.tile_counter, 1) : tile + (int)gridDim.x;
                } else if (step == 1) *s_next = fetched;
            }
            // ONE barrier per step: my DMA for this step has landed (vmcnt) and every wave has finished reading the
            // other buffer (previous step), which the prefetch below overwrites
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (step == (nsteps > 1 ? 1 : 0)) {
                ntile = *s_next;
                if (ntile < total_tiles) HW_DECODE(ntile, nn, na0, nb0, nco0)
            }
            int nchunk = chunk, ng = g + 1;
            if (ng == ngroups) { ng = 0; ++nchunk; }
            const bool more = step + 1 < nsteps;
            const int a_nxt = ((more ? ng : 0) == 0) ? (a_cur ^ 1) : a_cur;     // a new halo tile goes to the other A buffer
#define HW_ISSUE_NEXT                                                                                     \
            {                                                                                             \
                if (more) HW_PREFETCH(nchunk, ng, n, a0, b0, co0, buf ^ 1, a_nxt)                         \
                else if (ntile < total_tiles) HW_PREFETCH(0, 0, nn, na0, nb0, nco0, buf ^ 1, a_nxt)       \
            }
            HW_ISSUE_NEXT

            const char* Ab = (const char*)(As0 + a_cur * HP4);
            const f32x4* Bl = Bs0 + buf * BSZ + h * BN + li;          // this lane's column of the weight group
#define W_LD(BF, AF, TL, SC)                                                                              \
            {                                                                                             \
                _Pragma("unroll") for (int c_ = 0; c_ < NT; ++c_) BF[c_] = Bl[((TL) * 4 + (SC) * 2) * BN + c_ * 32]; \
                _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_)                                         \
                    AF[m_] = *(const f32x4*)(Ab + ((SC) ? (aaddr[TL][m_] ^ 32) : aaddr[TL][m_]));         \
            }
#define W_MFMA(BF, AF)                                                                                    \
            _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_)                                             \
            _Pragma("unroll") for (int c_ = 0; c_ < NT; ++c_) acc[m_][c_] = MFMA_BF16(BF[c_], AF[m_], acc[m_][c_]);
            {
                f32x4 bX[NT], bY[NT], aX[MT], aY[MT];
                W_LD(bX, aX, 0, 0)
#pragma unroll
                for (int tl = 0; tl < SSIE_TG; ++tl) {
                    if (tl >= tg) break;
                    W_LD(bY, aY, tl, 1)
                    W_MFMA(bX, aX)
                    if (tl + 1 < SSIE_TG && tl + 1 < tg) W_LD(bX, aX, (tl + 1 < SSIE_TG ? tl + 1 : 0), 0)
                    W_MFMA(bY, aY)
                }
            }
#undef W_LD
#undef W_MFMA
#undef HW_ISSUE_NEXT
            // draw the tile after next from the queue; its value is only needed at the next step's hand-off
            if (tid == 0 && step == 0 && nsteps > 1 && p.tile_counter)
                fetched = (int)gridDim.x + atomicAdd(p.tile_counter, 1);
            chunk = nchunk; g = ng; a_cur = a_nxt;
        }

        // epilogue: wave w holds tile rows 2w, 2w+1; M-tile m = columns 16m .. 16m+15, N-tile c = channels 32c .. 32c+31;
        // D^T = W x X above, so lane li = position, registers = channels (ssie_epilogue_ht)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            bool ok;
            const size_t opix = ssie_epilogue_pos(p, n, a0 + 2 * wave, b0 + 16 * m, li, ok);
#pragma unroll
            for (int c = 0; c < NT; ++c) ssie_epilogue_ht(p, acc[m][c], opix, co0 + c * 32 + 4 * h, ok);
        }
        n = nn; a0 = na0; b0 = nb0; co0 = nco0; tile = ntile;
    }
#undef HW_PREFETCH
#undef HW_DECODE
}



template __global__ void conv_fprop_bf16w_kernel<5, false>(const ConvParams);
template __global__ void conv_fprop_bf16w_kernel<5, true>(const ConvParams);

#define INST_H(NT, NA2, TH) template __global__ void conv_fprop_bf16_kernel<NT, NA2, TH>(const ConvParams);
INST_H(2, 3, 16) INST_H(2, 5, 16) INST_H(1, 3, 16) INST_H(1, 5, 16) INST_H(2, 5, 8)

static size_t lds_bytes_h(const ConvParams& p, int nt)
{
    return 2 * ((size_t)p.hp_h * p.hp_w * 64 + (size_t)SSIE_TG * 4 * 32 * nt * 16) + (size_t)SSIE_MAX_TAPS * 4 + 16;
}

template <int NT, int NA2, int TH>
static int launch_h_t(const ConvParams& p, size_t lds, hipStream_t st)
{
    static bool set = false;
    if (!set) { hipFuncSetAttribute((const void*)conv_fprop_bf16_kernel<NT, NA2, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    const size_t tiles = (size_t)p.N * p.tiles_y * p.tiles_x * p.co_blocks;
    const size_t wgs = tiles < 256 ? tiles : 256;
    hipLaunchKernelGGL((conv_fprop_bf16_kernel<NT, NA2, TH>), dim3((unsigned)wgs), dim3(512), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 61;
}

// p must come from ssie_make_conv_bf16 (th = 16 for stride 1, 8 for stride 2; 32-channel chunks)
int ssie_launch_fprop_bf16(const ConvParams& p, hipStream_t st)
{
    const int nt = (p.Cout_pad % 64 == 0) ? 2 : 1;
    const int na2 = (p.hp_h * p.hp_w * 4 + 511) / 512;
    const size_t lds = lds_bytes_h(p, nt);
    if (na2 > 5 || lds > 160 * 1024) return 62;
    if (p.tw == 32) {                      // geometry built for the wide kernel (ssie_make_conv_bf16)
        static bool set = false;
        if (!set) {
            hipFuncSetAttribute((const void*)conv_fprop_bf16w_kernel<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute((const void*)conv_fprop_bf16w_kernel<5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            set = true;
        }
        const size_t tiles = (size_t)p.N * p.tiles_y * p.tiles_x * p.co_blocks;
        const dim3 grid((unsigned)(tiles < 256 ? tiles : 256));
        if (p.nsrc == 1) hipLaunchKernelGGL((conv_fprop_bf16w_kernel<5, true>), grid, dim3(512), lds, st, p);
        else hipLaunchKernelGGL((conv_fprop_bf16w_kernel<5, false>), grid, dim3(512), lds, st, p);
        return hipGetLastError() == hipSuccess ? 0 : 65;
    }
    if (p.th == 8) return nt == 2 ? launch_h_t<2, 5, 8>(p, lds, st) : 63;
    if (p.th != 16) return 64;
    if (nt == 2) return na2 <= 3 ? launch_h_t<2, 3, 16>(p, lds, st) : launch_h_t<2, 5, 16>(p, lds, st);
    return na2 <= 3 ? launch_h_t<1, 3, 16>(p, lds, st) : launch_h_t<1, 5, 16>(p, lds, st);
}
