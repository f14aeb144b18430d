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

// Diagnostic build only (-DSSIE_STAMP, tools/stamp_bf16.py): wave 0's s_memtime per phase of the wide kernel, summed per workgroup:
// [0] start [1] barrier wait, first step of a tile [2] barrier wait, other steps [3] end [4] DMA issue [5] epilogue + bookkeeping
// [6] tiles [7] MFMA tap loops.  The shipped library never executes a stamp.
#ifdef SSIE_STAMP
__device__ unsigned long long* ssie_stamp_buf_h = nullptr;
static unsigned long long* g_stamp_host_buf = nullptr;
static int g_stamp_target = -1, g_stamp_launch = 0;
extern "C" int ssie_debug_set_stamp_buffer_h(void* buf, int launch_index) { g_stamp_host_buf = (unsigned long long*)buf; g_stamp_target = launch_index; g_stamp_launch = 0; return 0; }
#define HT_DECL unsigned long long st_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t_ = __builtin_amdgcn_s_memtime(); st_[0] = st_t_;
#define HT_ACC(k) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[k] += t_ - st_t_; st_t_ = t_; } while (0)
#define HT_FLUSH do { st_[3] = __builtin_amdgcn_s_memtime(); if (ssie_stamp_buf_h && threadIdx.x == 0) \
    for (int k_ = 0; k_ < 12; ++k_) ssie_stamp_buf_h[(size_t)blockIdx.x * 12 + k_] = st_[k_]; } while (0)
#define HT_TILE st_[6] += 1
#else
#define HT_DECL
#define HT_ACC(k)
#define HT_FLUSH
#define HT_TILE
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

// 16-byte stores from the transposed accumulator layout: lane (li, h) holds channels 8g + 4h + {0..3} of its position for the four
// groups g of a 32-channel accumulator tile, its partner lane (li, h ^ 1) the other four of every group.  v_permlane32_swap
// exchanges the upper 32 lanes of one register with the lower 32 of another: for a group pair (g0, g0 + 1), X = packed g0 and
// Y = packed g0 + 1, after the swap the lower half-wave holds (own X, partner's X) = channels 8 g0 .. 8 g0 + 7 and the upper one
// (partner's Y, own Y) = channels 8 (g0 + 1) .. + 7 - both as (X, Y).  One dwordx4 store per pair instead of two dwordx2: the
// epilogue is store-ISSUE bound (MI355X_MICROARCH.md, T21).
__device__ __forceinline__ uint4 ssie_pair_swap(uint2 x, uint2 y)
{
    const auto a = __builtin_amdgcn_permlane32_swap(x.x, y.x, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(x.y, y.y, false, false);
    return make_uint4(a[0], b[0], a[1], b[1]);
}

// one 32 x 32 accumulator tile -> memory, TRANSPOSED layout (the MFMAs are issued as D^T = W x X; see ssie_epilogue_t in
// conv_device.h): lane (li, h) = output position li of the 2 x 16 M-tile, register r = channel 8*(r>>2) + 4h + (r&3).  Four
// groups of four consecutive channels per lane: bf16 tensors move 8 bytes per group, fp32 outputs 16.
//   opix = element offset of (this lane's pixel, channel out_coff); c0 = first channel of the N-tile + 4h
// The bias comes from an LDS copy (bias_s, zero beyond Cout), NOT from global memory: vmcnt retires in issue order, so a global
// load here would sit behind the next tile's whole halo + weight prefetch (issued during the last MFMA step) - in-kernel
// stamps showed the epilogue of a 64 -> 64 3x3 bf16 tile waiting 19k cycles, half the tile's time, for exactly that.
__device__ __forceinline__ void ssie_epilogue_ht(const ConvParams& p, const float* bias_s, const f32x16& acc, size_t opix, int c0, bool pos_ok)
{
    if (!pos_ok) return;
    f32x4 v[4];
    bool full[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = c0 + 8 * g;
        full[g] = c + 4 <= p.Cout;
        const f32x4 b = *(const f32x4*)(bias_s + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[g][j] = acc[4 * g + j] + b[j];
    }
    // ONE branch on the activation for the whole tile (inside the element loop hipcc emits the scalar compare-and-branch chain
    // once per element: 64 x per wave and tile, ~12k cycles - stamped - against ~10k for the tile's MFMAs)
    if (p.act == ACT_RELU) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[g][j] = fmaxf(v[g][j], 0.f);
    } else if (p.act == ACT_SIGMOID) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[g][j] = __frcp_rn(1.f + __expf(-v[g][j]));
    }
    // the second output may have its own pixel stride (one launch per image: the R|I tensor and its bf16 twin)
    const size_t opix2 = (p.out2 && p.out2_cstride) ? (size_t)((unsigned)(opix - p.out_coff) / (unsigned)p.out_cstride) * p.out2_cstride + p.out_coff : opix;
    if (p.out2) {
        unsigned short* o2 = (unsigned short*)p.out2 + opix2 + c0;
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
    if (p.out_bf16 && (p.Cout & 7) == 0) {
        // whole 8-channel groups: 16-byte stores (ssie_pair_swap; both lanes of a pair share pos_ok, so both are here)
        const int hh = (c0 >> 2) & 1, cb = c0 - 4 * hh;
        unsigned short* ob = (unsigned short*)p.out + opix + cb + 8 * hh;
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
            const uint4 u = ssie_pair_swap(ssie_pack4bf(v[g]), ssie_pack4bf(v[g + 1]));
            if (SSIE_X_KEEP(cb + 8 * (g + hh) + 8 <= p.Cout, v[g][0])) *(uint4*)(ob + 8 * g) = u;
        }
        return;
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
            float t = acc[4 * g + j] + bias_s[c + j];
            if (p.act == ACT_RELU) t = fmaxf(t, 0.f);
            else if (p.act == ACT_SIGMOID) t = 1.f / (1.f + expf(-t));
            if (p.out2) ((unsigned short*)p.out2)[opix2 + c + j] = ssie_f2bf(t);
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
    float* bias_s = (float*)(s_next + 4);          // [Cout_pad]: the bias vector, staged once (see ssie_epilogue_ht)
    float* zero_bias_s = bias_s + p.Cout_pad;      // [Cout_pad] zeros: edge tiles of the lean path (bias already in the accumulators)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, li = lane & 31;
    const int wn = (NT == 1) ? 0 : (wave & 1);
    const int wm = (NT == 1) ? wave : (wave >> 1);

    for (int t = tid; t < p.ntaps; t += NTHR)
        tapoff[t] = ((int)p.tap_dy[t] - p.min_dy) * p.hp_w + ((int)p.tap_dx[t] - p.min_dx);
    for (int t = tid; t < p.Cout_pad; t += NTHR) { bias_s[t] = (p.bias && t < p.Cout) ? p.bias[t] : 0.f; zero_bias_s[t] = 0.f; }

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
            // Tap loop with a COMPILE-TIME trip count (groups are 9, 4, 2 or 1 taps): the A-fragment byte addresses of the
            // group are computed once per step (k-half 1 = the same address ^ 32) and the fragments of tap t+1 / k-half 1 are
            // in flight under the MFMAs before them.  The rolled loop recomputed the swizzled address per fragment (~7 VALU
            // per 128-bit read) and issued each read right before its use: the 81-tap layer ran its MFMA pipe at 29 %.
            const char* Ab = (const char*)As;
            const f32x4* Bl = Bs + h * BN + wn * 32 + li;
#define G_LD(BF, AF, TL, SC)                                                                              \
            {                                                                                             \
                BF = Bl[((TL) * 4 + (SC) * 2) * BN];                                                      \
                _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_) AF[m_] = *(const f32x4*)(Ab + ((SC) ? (ad[TL][m_] ^ 32) : ad[TL][m_])); \
            }
#define G_MFMA(BF, AF) _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_) acc[m_] = MFMA_BF16(BF, AF[m_], acc[m_]);
#define G_TAPS(TG_)                                                                                       \
            {                                                                                             \
                int ad[TG_][MT];                                                                          \
                _Pragma("unroll") for (int tl = 0; tl < TG_; ++tl) {                                      \
                    const int off_ = tapoff[t0 + tl];                                                     \
                    _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_) {                                   \
                        const int hp_ = pixbase[m_] + off_;                                               \
                        ad[tl][m_] = (hp_ * 4 + (h ^ ssie_swz(hp_))) * 16;                                \
                    }                                                                                     \
                }                                                                                         \
                f32x4 bX, bY, aX[MT], aY[MT];                                                             \
                G_LD(bX, aX, 0, 0)                                                                        \
                _Pragma("unroll") for (int tl = 0; tl < TG_; ++tl) {                                      \
                    G_LD(bY, aY, tl, 1)                                                                   \
                    G_MFMA(bX, aX)                                                                        \
                    if (tl + 1 < TG_) G_LD(bX, aX, (tl + 1 < TG_ ? tl + 1 : 0), 0)                         \
                    G_MFMA(bY, aY)                                                                        \
                }                                                                                         \
            }
            switch (tg) {
            case 9: G_TAPS(9) break;
            case 4: G_TAPS(4) break;
            case 2: G_TAPS(2) break;
            case 1: G_TAPS(1) break;
            default:
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
                break;
            }
#undef G_TAPS
#undef G_MFMA
#undef G_LD
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
            ssie_epilogue_ht(p, bias_s, acc[m], opix, co0 + wn * 32 + 4 * h, ok);
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
template <int NA2, bool SINGLE, bool ILV>
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
    float* bias_s = (float*)(s_next + 4);          // [Cout_pad]: the bias vector, staged once (see ssie_epilogue_ht)
    float* zero_bias_s = bias_s + p.Cout_pad;      // [Cout_pad] zeros: edge tiles of the lean path (bias already in the accumulators)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform: DMA piece indices and LDS destinations stay scalar
    const int h = lane >> 5, li = lane & 31;

    for (int t = tid; t < p.ntaps; t += NTHR)
        tapoff[t] = ((int)p.tap_dy[t] - p.min_dy) * p.hp_w + ((int)p.tap_dx[t] - p.min_dx);
    for (int t = tid; t < p.Cout_pad; t += NTHR) { bias_s[t] = (p.bias && t < p.Cout) ? p.bias[t] : 0.f; zero_bias_s[t] = 0.f; }

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
    // Interior tiles of plain (not up-sampled) sources whose channel count fills whole chunks take a lean address path: the
    // halo slot's address is ONE uniform base per step plus a per-lane offset of two multiplies - no bounds checks, no clamping,
    // no 64-bit vector arithmetic (the DMA-issue phase was ~20 % of a tile's time on the general path).
#define HW_PREFETCH(CHUNK, G, N_, A0_, B0_, CO0_, BUF, ABUF)                                                        \
    {                                                                                                         \
        if ((G) == 0) {                                                                                       \
            const SrcSel s_ = SINGLE ? ssie_only_src(p) : ssie_pick_src(p, (CHUNK) * CKH);                \
            const bool up_ = s_.sy != 1.f || s_.sx != 1.f;                                                    \
            const int vy0_ = (A0_) + p.min_dy, vx0_ = (B0_) + p.min_dx;                                       \
            f32x4* abuf_ = As0 + (ABUF) * HP4;                                                                \
            const bool lean_ = !up_ && (s_.C & 31) == 0 && vy0_ >= 0 && vx0_ >= 0 && vy0_ + p.hp_h <= p.Hv && vx0_ + p.hp_w <= p.Wv; \
            if (lean_) {                                                                                      \
                const unsigned short* sb_ = (const unsigned short*)s_.ptr +                                   \
                    ((size_t)((N_) * s_.Hs + vy0_) * s_.Ws + vx0_) * s_.cstride + s_.coff + (CHUNK) * CKH - s_.cbeg; \
                _Pragma("unroll") for (int i_ = 0; i_ < NA2; ++i_) {                                          \
                    if (tid + i_ * NTHR < HP4) {                                                              \
                        const int lo_ = (ahy[i_] * s_.Ws + ahx[i_]) * s_.cstride + 8 * aj[i_];               \
                        GLDS16(sb_ + lo_, abuf_ + i_ * NTHR + wave * 64);                                     \
                    }                                                                                         \
                }                                                                                             \
            } else {                                                                                          \
                _Pragma("unroll") for (int i_ = 0; i_ < NA2; ++i_) {                                          \
                    if (tid + i_ * NTHR < HP4) {                                                              \
                        const f32x4* g_ = ssie_virtual_addr_h(s_, up_, (N_), vy0_ + ahy[i_], vx0_ + ahx[i_], p.Hv, p.Wv, \
                                                              (CHUNK) * CKH + 8 * aj[i_] - s_.cbeg);          \
                        GLDS16(g_, abuf_ + i_ * NTHR + wave * 64);                                            \
                    }                                                                                         \
                }                                                                                             \
            }                                                                                                 \
        }                                                                                                     \
        const int t0_ = (G) * SSIE_TG;                                                                        \
        const int pieces_ = min(SSIE_TG, p.ntaps - t0_) * 4;              /* rows of BN = 64 slots: one piece each */ \
        const f32x4* wsrc_ = (const f32x4*)p.wpacked + ((size_t)((CHUNK) * p.ntaps + t0_) * 4) * p.Cout_pad + (CO0_); \
        f32x4* bbuf_ = Bs0 + (BUF) * BSZ;                                                                     \
        for (int q_ = wave; q_ < pieces_; q_ += NW)                                                           \
            GLDS16(wsrc_ + (size_t)q_ * p.Cout_pad + lane, bbuf_ + q_ * 64);                                  \
    }

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int n, a0, b0, co0;
    HW_DECODE(tile, n, a0, b0, co0)
    int gstep = 0;
    int a_cur = 0;          // A buffer holding the halo tile of the step about to be computed
    HW_PREFETCH(0, 0, n, a0, b0, co0, 0, 0)
    int fetched = 0x7fffffff;
    HT_DECL
    // The two waves of a SIMD (w and w+4) run the same program; issuing the next step's DMA (address VALU work) at
    // the same moment would leave the SIMD's MFMA pipe idle.  Waves 4-7 therefore issue it in the middle of their
    // tap loop while waves 0-3 issue it up front.
    const bool late_prefetch = false;   // measured: issuing the DMA inside the tap loop (waves 4-7) was 5-25 % SLOWER

    // LEAN EPILOGUE (most layers: bf16 output, ReLU or none, no residual / second output, Cout a multiple of 32, tile inside
    // the image).  The kernel is bound by the non-MFMA instructions its two waves per SIMD issue (stamps: the general
    // epilogue took 8-12k cycles per tile against ~10k for the tile's MFMAs), so everything tile-invariant is hoisted:
    //   * the bias enters as the accumulators' initial value (read from LDS right after the step-0 barrier, where vmcnt has
    //     been drained anyway - a ds_read later forces hipcc to drain the in-flight LDS-DMA prefetch first)
    //   * the lane's element offsets inside the tile are computed once per kernel; a tile contributes ONE uniform base
    //   * per accumulator what is left is 8 packed max, 8 packed converts and 4 eight-byte stores
    const bool lean = p.out_bf16 && !p.addsrc && !p.out2 && (p.Cout % 32) == 0 && p.act != ACT_SIGMOID;
    const float relu_lo = p.act == ACT_RELU ? 0.f : -3.0e38f;
    int lane_off[MT];                                   // element offset of (this lane's pixel of M-tile m, channel 4h) inside a tile
#pragma unroll
    for (int m = 0; m < MT; ++m)
        lane_off[m] = ((2 * wave + (li >> 4)) * p.so * p.Wout + (16 * m + (li & 15)) * p.so) * p.out_cstride + 8 * h;   // 16-byte stores: ssie_pair_swap

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
            HT_ACC(5);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            HT_ACC(step == 0 ? 1 : 2);
            if (step == 0 && lean) {
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 b = *(const f32x4*)(bias_s + co0 + 32 * c + 8 * g4 + 4 * h);
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[m][c][4 * g4 + j] = b[j];
                    }
            }
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
            // ILV (9-tap layers): the next step's DMA is issued PIECE BY PIECE between the taps of this step's MFMA loop - one halo
            // slot after each of taps 0-4, the (up to five) weight pieces after taps 5-7 - instead of as one block right behind
            // the barrier.  An LDS-DMA instruction holds its wave at issue for several hundred cycles when the wave's previous ones
            // are still in flight (stamps: 340 cycles per piece with eight waves issuing ten each back to back, 550 with four
            // waves issuing nineteen), and an in-order wave has its MFMAs queued up behind; one piece per ~8 MFMAs finds the
            // queue empty.
            // Only steps whose halo tile lies inside the image take this path (one uniform base + a per-lane offset per slot); border
            // tiles and up-sampled sources issue the block up front as before.
            const bool pf_any = more || ntile < total_tiles;
            const int pchunk = more ? nchunk : 0;
            const int pn = more ? n : nn, pa0 = more ? a0 : na0, pb0 = more ? b0 : nb0, pco0 = more ? co0 : nco0;
            const SrcSel ps = SINGLE ? ssie_only_src(p) : ssie_pick_src(p, pchunk * CKH);
            const int pvy0 = pa0 + p.min_dy, pvx0 = pb0 + p.min_dx;
            const bool ilv_now = ILV && tg == 9 && NA2 == 5 && pf_any && ps.sy == 1.f && ps.sx == 1.f && (ps.C & 31) == 0 &&
                                 pvy0 >= 0 && pvx0 >= 0 && pvy0 + p.hp_h <= p.Hv && pvx0 + p.hp_w <= p.Wv;
            f32x4* const pabuf = As0 + a_nxt * HP4;
            const unsigned short* const psb = (const unsigned short*)ps.ptr +
                ((long)(pn * ps.Hs + pvy0) * ps.Ws + pvx0) * ps.cstride + ps.coff + pchunk * CKH - ps.cbeg;
            const f32x4* const pwsrc = (const f32x4*)p.wpacked + ((size_t)(pchunk * p.ntaps) * 4) * p.Cout_pad + pco0;
            f32x4* const pbbuf = Bs0 + (buf ^ 1) * BSZ;
#define HW_HALO_PIECE(I_)                                                                                 \
            if (tid + (I_) * NTHR < HP4) {                                                                \
                const int lo_ = (ahy[I_] * ps.Ws + ahx[I_]) * ps.cstride + 8 * aj[I_];                    \
                GLDS16(psb + lo_, pabuf + (I_) * NTHR + wave * 64);                                       \
            }
#define HW_W_PIECE(J_)                                                                                    \
            { const int q_ = wave + NW * (J_); if (q_ < 36) GLDS16(pwsrc + (size_t)q_ * p.Cout_pad + lane, pbbuf + q_ * 64); }
#define HW_ILV_HOOK(TL_)                                                                                  \
            if (ilv_now) {                                                                                \
                if ((TL_) == 0) { HW_HALO_PIECE(0) } else if ((TL_) == 1) { HW_HALO_PIECE(1) }             \
                else if ((TL_) == 2) { HW_HALO_PIECE(2) } else if ((TL_) == 3) { HW_HALO_PIECE(3) }        \
                else if ((TL_) == 4) { HW_HALO_PIECE(4) }                                                 \
                else if ((TL_) == 5) { HW_W_PIECE(0) HW_W_PIECE(1) }                                      \
                else if ((TL_) == 6) { HW_W_PIECE(2) HW_W_PIECE(3) }                                      \
                else if ((TL_) == 7) { HW_W_PIECE(4) }                                                    \
            }
            if (!ilv_now) HW_ISSUE_NEXT
            HT_ACC(4);

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
            // the tap loop with a COMPILE-TIME trip count: aaddr[][] then stays in fixed registers and the fragment reads of
            // tap t+1 are in flight under the MFMAs of tap t.  (With a run-time bound hipcc keeps the loop rolled, indexes the
            // address array through s_set_gpr_idx and drains lgkmcnt(0) every tap - the kernel was instruction-bound.)
#define W_TAPS(TG_)                                                                                       \
            {                                                                                             \
                f32x4 bX[NT], bY[NT], aX[MT], aY[MT];                                                     \
                W_LD(bX, aX, 0, 0)                                                                        \
                _Pragma("unroll") for (int tl = 0; tl < TG_; ++tl) {                                      \
                    W_LD(bY, aY, tl, 1)                                                                   \
                    W_MFMA(bX, aX)                                                                        \
                    if (tl + 1 < TG_) W_LD(bX, aX, (tl + 1 < TG_ ? tl + 1 : 0), 0)                         \
                    if (TG_ == 9) HW_ILV_HOOK(tl)                                                         \
                    W_MFMA(bY, aY)                                                                        \
                }                                                                                         \
            }
            switch (tg) {
            case 9: W_TAPS(9) break;
            case 4: W_TAPS(4) break;
            case 2: W_TAPS(2) break;
            case 1: W_TAPS(1) break;
            default: {
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
            } break;
            }
#undef W_TAPS
#undef W_LD
#undef W_MFMA
#undef HW_ISSUE_NEXT
#undef HW_ILV_HOOK
#undef HW_W_PIECE
#undef HW_HALO_PIECE
            // draw the tile after next from the queue; its value is only needed at the next step's hand-off
            if (tid == 0 && step == 0 && nsteps > 1 && p.tile_counter)
                fetched = (int)gridDim.x + atomicAdd(p.tile_counter, 1);
            chunk = nchunk; g = ng; a_cur = a_nxt;
            HT_ACC(7);
        }

        // epilogue: wave w holds tile rows 2w, 2w+1; M-tile m = columns 16m .. 16m+15, N-tile c = channels 32c .. 32c+31;
        // D^T = W x X above, so lane li = position, registers = channels (ssie_epilogue_ht)
        HT_ACC(8);
        const bool inside = a0 + TH <= p.Ho && b0 + TWW <= p.Wo && (a0 + TH - 1) * p.so + p.py < p.Hout && (b0 + TWW - 1) * p.so + p.px < p.Wout;
        if (lean && inside) {
            unsigned short* tbase = (unsigned short*)p.out + ((size_t)(n * p.Hout + a0 * p.so + p.py) * p.Wout + b0 * p.so + p.px) * p.out_cstride + p.out_coff + co0;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int g2 = 0; g2 < 4; g2 += 2) {
                        const f32x16& a = acc[m][c];
                        uint2 u[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            u[e] = make_uint2(ssie_pack2bf(fmaxf(a[4 * (g2 + e)], relu_lo), fmaxf(a[4 * (g2 + e) + 1], relu_lo)),
                                              ssie_pack2bf(fmaxf(a[4 * (g2 + e) + 2], relu_lo), fmaxf(a[4 * (g2 + e) + 3], relu_lo)));
                        const uint4 v = ssie_pair_swap(u[0], u[1]);
                        if (SSIE_X_KEEP(true, a[0])) *(uint4*)(tbase + lane_off[m] + 32 * c + 8 * g2) = v;
                    }
        } else {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                bool ok;
                const size_t opix = ssie_epilogue_pos(p, n, a0 + 2 * wave, b0 + 16 * m, li, ok);
#pragma unroll
                for (int c = 0; c < NT; ++c) ssie_epilogue_ht(p, lean ? zero_bias_s : bias_s, acc[m][c], opix, co0 + c * 32 + 4 * h, ok);
            }
        }
        HT_ACC(9);
        n = nn; a0 = na0; b0 = nb0; co0 = nco0; tile = ntile;
        HT_TILE;
    }
    HT_ACC(5);
    HT_FLUSH;
#undef HW_PREFETCH
#undef HW_DECODE
}


// Wave-specialised form of the wide kernel (the shipped one for the 16 x 32 geometry): 12 waves per workgroup, three per SIMD.
//   waves 0-7  CONSUMERS: fragment reads, MFMAs and the epilogue - no global loads at all
//   waves 8-11 PRODUCERS: the LDS-DMA of the next step's halo tile and weight group (address arithmetic + 19-20
//              global_load_lds per wave and step), then s_waitcnt vmcnt(0)
// and ONE s_barrier per step that all twelve waves pass: it tells the consumers that the step's operands have landed and the
// producers that the other buffer has been read.  In the eight-wave kernel every wave issued its share of the DMA between the
// barrier and its tap loop; stamps (profiles/r02_bf16_stamps.md) put that phase at 6 800 of a tile's 29 500 cycles - the waves
// sit in the vector-memory issue queue (76 KB per step against ~22 B/clk) with their MFMAs queued up behind - and another 6 500
// in the barrier that follows a step that is shorter than the DMA's flight.  A stalled producer holds up nobody's MFMAs.
// Static tile assignment (tile += gridDim.x), one tap group per chunk (<= 9 taps), so a step = one 32-channel chunk.
// GEO: 0 = 16 x 32 output tile, input stride 1 (the wide geometry); 1 = 8 x 16 tile, input stride 2 (ssie_make_conv_bf16's stride-2
// geometry, 17 x 33 halo); 2 = 16 x 16 tile, stride 1 (the transposed-convolution classes and other 64-channel layers)
template <int TGT, bool SINGLE, int NWC, bool RESW, bool ADD = false, int GEO = 0>
__global__ __launch_bounds__(64 * (NWC + 4)) void conv_fprop_bf16ws_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int NT = 2, NWP = 4, BN = 64, CKH = 32;
    constexpr int TH = GEO == 1 ? 8 : 16, TWW = GEO == 0 ? 32 : 16, SI = GEO == 1 ? 2 : 1, MPR = TWW / 16;
    constexpr int RW = TH / NWC, MT = RW / 2 * MPR;    // a consumer wave owns RW tile rows = MT M-tiles of 2 rows x 16 columns
    static_assert(RW >= 2 && RW % 2 == 0, "a consumer wave needs whole M-tiles");
    constexpr int PTHR = 64 * NWP, NTHR = 64 * (NWC + NWP);
    constexpr int NAP = 10;                         // DMA rounds of the 256 producer lanes over the 18 x 34 x 4 halo slots
    constexpr int BSZ = SSIE_TG * 4 * BN;           // float4 per B buffer
    const int HP = p.hp_h * p.hp_w, HP4 = HP * 4;
    f32x4* As0 = (f32x4*)smem_f;                    // [2][HP4]
    f32x4* Bs0 = As0 + 2 * HP4;                     // [2][BSZ]
    float* bias_s = (float*)(Bs0 + 2 * BSZ) + SSIE_MAX_TAPS + 4;   // [Cout_pad] (same LDS map as the eight-wave kernel)
    float* zero_bias_s = bias_s + p.Cout_pad;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, li = lane & 31;
    for (int t = tid; t < p.Cout_pad; t += NTHR) { bias_s[t] = (p.bias && t < p.Cout) ? p.bias[t] : 0.f; zero_bias_s[t] = 0.f; }

    __syncthreads();                                // bias_s is read at the top of a tile, before that tile's first step barrier

    const int nsteps = p.nchunks;
    const int total_tiles = p.N * p.tiles_y * p.tiles_x * p.co_blocks;
    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
#define WS_DECODE(T, N_, A0_, B0_, CO0_)                                                  \
    {                                                                                     \
        int q_ = (T);                                                                     \
        CO0_ = (q_ % p.co_blocks) * BN; q_ /= p.co_blocks;                                \
        B0_ = (q_ % p.tiles_x) * TWW; q_ /= p.tiles_x;                                    \
        A0_ = (q_ % p.tiles_y) * TH; N_ = q_ / p.tiles_y;                                 \
    }
    int n, a0, b0, co0;
    WS_DECODE(tile, n, a0, b0, co0)

    if (wave >= NWC) {
        // ------------------------------------------------ producers ------------------------------------------------
        const int ptid = tid - 64 * NWC, pwave = wave - NWC;
        // this lane's halo slots: LDS slot id = i*PTHR + ptid (linear), it holds channel octet j = (id&3) ^ swz(pixel)
        int ahy[NAP], ahx[NAP], aj[NAP];
#pragma unroll
        for (int i = 0; i < NAP; ++i) {
            const int id = min(ptid + i * PTHR, HP4 - 1);
            const int pix = id >> 2;
            ahy[i] = pix / p.hp_w; ahx[i] = pix - ahy[i] * p.hp_w; aj[i] = (id & 3) ^ ssie_swz(pix);
        }
#define WS_WEIGHTS(CHUNK, CO0_, BUF)                                                                          \
        {                                                                                                     \
            const f32x4* wsrc_ = (const f32x4*)p.wpacked + ((size_t)((CHUNK) * p.ntaps) * 4) * p.Cout_pad + (CO0_); \
            f32x4* bbuf_ = Bs0 + (BUF) * BSZ;                                                                 \
            for (int q_ = pwave; q_ < p.ntaps * 4; q_ += NWP)           /* rows of BN = 64 slots: one piece each */ \
                GLDS16(wsrc_ + (size_t)q_ * p.Cout_pad + lane, bbuf_ + q_ * 64);                              \
        }
#define WS_PREFETCH(CHUNK, N_, A0_, B0_, CO0_, BUF)                                                           \
        {                                                                                                     \
            const SrcSel s_ = SINGLE ? ssie_only_src(p) : ssie_pick_src(p, (CHUNK) * CKH);                    \
            const bool up_ = s_.sy != 1.f || s_.sx != 1.f;                                                    \
            const int vy0_ = (A0_) * SI + p.min_dy, vx0_ = (B0_) * SI + p.min_dx;                             \
            f32x4* abuf_ = As0 + (BUF) * HP4;                                                                 \
            const bool lean_ = !up_ && (s_.C & 31) == 0 && vy0_ >= 0 && vx0_ >= 0 && vy0_ + p.hp_h <= p.Hv && vx0_ + p.hp_w <= p.Wv; \
            if (lean_) {                                                                                      \
                const unsigned short* sb_ = (const unsigned short*)s_.ptr +                                   \
                    ((size_t)((N_) * s_.Hs + vy0_) * s_.Ws + vx0_) * s_.cstride + s_.coff + (CHUNK) * CKH - s_.cbeg; \
                _Pragma("unroll") for (int i_ = 0; i_ < NAP; ++i_) {                                          \
                    if (ptid + i_ * PTHR < HP4) {                                                             \
                        const int lo_ = (ahy[i_] * s_.Ws + ahx[i_]) * s_.cstride + 8 * aj[i_];               \
                        GLDS16(sb_ + lo_, abuf_ + i_ * PTHR + pwave * 64);                                    \
                    }                                                                                         \
                }                                                                                             \
            } else {                                                                                          \
                _Pragma("unroll") for (int i_ = 0; i_ < NAP; ++i_) {                                          \
                    if (ptid + i_ * PTHR < HP4) {                                                             \
                        const f32x4* g_ = ssie_virtual_addr_h(s_, up_, (N_), vy0_ + ahy[i_], vx0_ + ahx[i_], p.Hv, p.Wv, \
                                                              (CHUNK) * CKH + 8 * aj[i_] - s_.cbeg);          \
                        GLDS16(g_, abuf_ + i_ * PTHR + pwave * 64);                                           \
                    }                                                                                         \
                }                                                                                             \
            }                                                                                                 \
            if (!RESW) WS_WEIGHTS(CHUNK, CO0_, BUF)                                                           \
        }
        // RESW (at most two chunks, one output-channel block): the layer's whole packed weight stays in LDS - chunk c in B
        // buffer c - and a step moves only its halo tile: half the LDS-DMA pieces per tile
        if (RESW) for (int c_ = 0; c_ < nsteps; ++c_) WS_WEIGHTS(c_, 0, c_)
        WS_PREFETCH(0, n, a0, b0, co0, 0)
        int gstep = 0;
        HT_DECL
        while (tile < total_tiles) {
            const int ntile = tile + (int)gridDim.x;
            int nn = n, na0 = a0, nb0 = b0, nco0 = co0;
            if (ntile < total_tiles) WS_DECODE(ntile, nn, na0, nb0, nco0)
            for (int step = 0; step < nsteps; ++step, ++gstep) {
                const int buf = gstep & 1;
                // my share of this step's operands has landed; behind the barrier everybody's has, and the consumers are done
                // with the other buffer
                HT_ACC(4);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                HT_ACC(10);
                asm volatile("s_barrier" ::: "memory");
                HT_ACC(11);
                if (step + 1 < nsteps) WS_PREFETCH(step + 1, n, a0, b0, co0, buf ^ 1)
                else if (ntile < total_tiles) WS_PREFETCH(0, nn, na0, nb0, nco0, buf ^ 1)
            }
            n = nn; a0 = na0; b0 = nb0; co0 = nco0; tile = ntile;
        }
#ifdef SSIE_STAMP
        if (ssie_stamp_buf_h && tid == 64 * NWC) { ssie_stamp_buf_h[(size_t)blockIdx.x * 12 + 4] = st_[4]; ssie_stamp_buf_h[(size_t)blockIdx.x * 12 + 10] = st_[10]; ssie_stamp_buf_h[(size_t)blockIdx.x * 12 + 11] = st_[11]; }
#endif
#undef WS_PREFETCH
#undef WS_WEIGHTS
        return;
    }

    // ---------------------------------------------------- consumers ----------------------------------------------------
    int pixbase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) pixbase[m] = (RW * wave + 2 * (m / MPR) + (li >> 4)) * SI * p.hp_w + (16 * (m % MPR) + (li & 15)) * SI;
    // byte offset inside a halo buffer of (M-tile m, tap t, k-quad 0); k-quad 1 is the same address ^ 32 (slot index ^ 2)
    int aaddr[SSIE_TG][MT];
#pragma unroll
    for (int t = 0; t < SSIE_TG; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int hp = pixbase[m] + ((int)p.tap_dy[t] - p.min_dy) * p.hp_w + ((int)p.tap_dx[t] - p.min_dx);
            aaddr[t][m] = t < p.ntaps ? (hp * 4 + (h ^ ssie_swz(hp))) * 16 : 0;
        }
    // lean epilogue only (see conv_fprop_bf16w_kernel; the launcher sends every other layer to that kernel)
    const float relu_lo = p.act == ACT_RELU ? 0.f : -3.0e38f;
    int lane_off[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
        lane_off[m] = ((RW * wave + 2 * (m / MPR) + (li >> 4)) * p.so * p.Wout + (16 * (m % MPR) + (li & 15)) * p.so) * p.out_cstride + 8 * h;

    int gstep = 0;
    HT_DECL
    while (tile < total_tiles) {
        // the bias is the accumulators' initial value
        f32x16 acc[MT][NT];
        {
            const float* bsrc = bias_s + co0 + 4 * h;
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 b = *(const f32x4*)(bsrc + 32 * c + 8 * g4);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[m][c][4 * g4 + j] = b[j];
                }
        }

        for (int step = 0; step < nsteps; ++step, ++gstep) {
            const int buf = gstep & 1;
            // no vmcnt here: a consumer's only vector-memory traffic is its epilogue stores, which nobody waits for
            HT_ACC(5);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            HT_ACC(step == 0 ? 1 : 2);
            const char* Ab = (const char*)(As0 + buf * HP4);
            const f32x4* Bl = Bs0 + (RESW ? step : buf) * BSZ + h * BN + li;          // this lane's column of the weight group
#define W_LD(BF, AF, TL, SC)                                                                              \
            {                                                                                             \
                _Pragma("unroll") for (int c_ = 0; c_ < NT; ++c_) BF[c_] = Bl[((TL) * 4 + (SC) * 2) * BN + c_ * 32]; \
                _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_)                                         \
                    AF[m_] = *(const f32x4*)(Ab + ((SC) ? (aaddr[TL][m_] ^ 32) : aaddr[TL][m_]));         \
            }
#define W_MFMA(BF, AF)                                                                                    \
            _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_)                                             \
            _Pragma("unroll") for (int c_ = 0; c_ < NT; ++c_) acc[m_][c_] = MFMA_BF16(BF[c_], AF[m_], acc[m_][c_]);
#define W_TAPS(TG_)                                                                                       \
            {                                                                                             \
                f32x4 bX[NT], bY[NT], aX[MT], aY[MT];                                                     \
                W_LD(bX, aX, 0, 0)                                                                        \
                _Pragma("unroll") for (int tl = 0; tl < TG_; ++tl) {                                      \
                    W_LD(bY, aY, tl, 1)                                                                   \
                    W_MFMA(bX, aX)                                                                        \
                    if (tl + 1 < TG_) W_LD(bX, aX, (tl + 1 < TG_ ? tl + 1 : 0), 0)                         \
                    W_MFMA(bY, aY)                                                                        \
                }                                                                                         \
            }
            W_TAPS(TGT)
#undef W_TAPS
#undef W_LD
#undef W_MFMA
            HT_ACC(7);
        }

        const bool inside = a0 + TH <= p.Ho && b0 + TWW <= p.Wo && (a0 + TH - 1) * p.so + p.py < p.Hout && (b0 + TWW - 1) * p.so + p.px < p.Wout;
        if (inside) {
            unsigned short* tbase = (unsigned short*)p.out + ((size_t)(n * p.Hout + a0 * p.so + p.py) * p.Wout + b0 * p.so + p.px) * p.out_cstride + p.out_coff + co0;
            const unsigned short* abase = (const unsigned short*)p.addsrc + (tbase - (unsigned short*)p.out);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int g2 = 0; g2 < 4; g2 += 2) {
                        const f32x16& a = acc[m][c];
                        uint2 u[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            f32x4 v = {fmaxf(a[4 * (g2 + e)], relu_lo), fmaxf(a[4 * (g2 + e) + 1], relu_lo), fmaxf(a[4 * (g2 + e) + 2], relu_lo), fmaxf(a[4 * (g2 + e) + 3], relu_lo)};
                            // ADD: the residual (bf16, same geometry as the output) is added to the activated value, in fp32, before
                            // rounding: this lane's own four channels of the group, i.e. the un-swapped 8-byte position
                            if (ADD) v += ssie_unpack4bf(*(const uint2*)(abase + lane_off[m] - 4 * h + 32 * c + 8 * (g2 + e)));
                            u[e] = ssie_pack4bf(v);
                        }
                        *(uint4*)(tbase + lane_off[m] + 32 * c + 8 * g2) = ssie_pair_swap(u[0], u[1]);
                    }
        } else {                                    // edge tile: the same stores, per-position validity
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                bool ok;
                const size_t opix = ssie_epilogue_pos(p, n, a0 + RW * wave + 2 * (m / MPR), b0 + 16 * (m % MPR), li, ok);
                unsigned short* ob = (unsigned short*)p.out + opix + co0 + 8 * h;
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int g2 = 0; g2 < 4; g2 += 2) {
                        const f32x16& a = acc[m][c];
                        uint2 u[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            f32x4 v = {fmaxf(a[4 * (g2 + e)], relu_lo), fmaxf(a[4 * (g2 + e) + 1], relu_lo), fmaxf(a[4 * (g2 + e) + 2], relu_lo), fmaxf(a[4 * (g2 + e) + 3], relu_lo)};
                            if (ADD && ok) v += ssie_unpack4bf(*(const uint2*)((const unsigned short*)p.addsrc + opix + co0 + 4 * h + 32 * c + 8 * (g2 + e)));
                            u[e] = ssie_pack4bf(v);
                        }
                        const uint4 v = ssie_pair_swap(u[0], u[1]);         // (every lane takes part in the swap: no branch around it)
                        if (ok) *(uint4*)(ob + 32 * c + 8 * g2) = v;
                    }
            }
        }
        HT_ACC(9);
        HT_TILE;
        tile += (int)gridDim.x;
        if (tile < total_tiles) WS_DECODE(tile, n, a0, b0, co0)
    }
#ifdef SSIE_STAMP
    st_[3] = __builtin_amdgcn_s_memtime();
    if (ssie_stamp_buf_h && tid == 0)
        for (int k_ = 0; k_ < 10; ++k_) if (k_ != 4) ssie_stamp_buf_h[(size_t)blockIdx.x * 12 + k_] = st_[k_];
#endif
#undef WS_DECODE
}

template __global__ void conv_fprop_bf16ws_kernel<9, false, 8, false>(const ConvParams);
template __global__ void conv_fprop_bf16ws_kernel<9, true, 8, false>(const ConvParams);
template __global__ void conv_fprop_bf16ws_kernel<1, false, 8, false>(const ConvParams);
template __global__ void conv_fprop_bf16ws_kernel<1, true, 8, false>(const ConvParams);
template __global__ void conv_fprop_bf16ws_kernel<9, false, 4, false>(const ConvParams);
template __global__ void conv_fprop_bf16ws_kernel<9, true, 4, false>(const ConvParams);
template __global__ void conv_fprop_bf16ws_kernel<9, true, 8, true>(const ConvParams);
template __global__ void conv_fprop_bf16ws_kernel<9, true, 8, true, true>(const ConvParams);
template __global__ void conv_fprop_bf16ws_kernel<9, true, 4, false, false, 1>(const ConvParams);      // stride-2 3 x 3
template __global__ void conv_fprop_bf16ws_kernel<4, true, 4, false, false, 2>(const ConvParams);      // transposed-convolution classes
template __global__ void conv_fprop_bf16ws_kernel<2, true, 4, false, false, 2>(const ConvParams);
template __global__ void conv_fprop_bf16ws_kernel<1, true, 4, false, false, 2>(const ConvParams);
template __global__ void conv_fprop_bf16ws_kernel<9, true, 4, true>(const ConvParams);


// The 9 x 9 layer (shallow_conv, <= 32 input channels -> 64) on the wave-specialised structure: tile = 16 x 32 positions x 64
// channels, 8 consumer waves (64 positions x 64 channels each) + 4 producer waves.  The layer's packed weights are 332 KB, so
// they stream through LDS once per TILE, one kernel row (9 taps, 36.9 KB) per step, double-buffered; the generic kernel's
// 16 x 16 tile made that 369 KB of LDS-DMA per 20 k cycles of MFMA work - DMA-bound (0.43 ms for a 0.14 ms MFMA floor) - and its
// 64 x 32 wave tile read 1.5 fragments per MFMA.  Here a tile moves 393 KB for 41 k MFMA cycles and a wave reads one per MFMA.
//   * halo tile 24 x 40 pixels x 32 channels (61 KB), ONE buffer: all nine steps of a tile read it.  Row pitch 40 is not a
//     multiple of 16 pixels, so the slot swizzle of kernel row g differs from row 0's by (10 g) & 3 = 2 (g & 1): odd rows swap
//     the two k-halves (address ^ 32) - the per-tap addresses are computed once per kernel, a step adds g * 2560 bytes
//   * barriers: one per step (weights of the step landed / the other weight buffer free) and one at the end of a tile, after
//     which the producers overwrite the halo with the next tile's while the consumers run their epilogue
// Static tile assignment; lean epilogue (bf16 output, ReLU or none) with 16-byte stores.
__global__ __launch_bounds__(768) void conv9x9_bf16ws_kernel(const ConvParams p, int tiles_x, int tiles_y)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int NT = 2, NWC = 8, NWP = 4, BN = 64, TH = 16, TWW = 32, MT = 2;
    constexpr int HPH = 24, HPW = 40, HP4 = HPH * HPW * 4;          // 3 840 16-byte slots = 15 DMA rounds of the 256 producer lanes
    constexpr int PTHR = 64 * NWP, NTHR = 64 * (NWC + NWP), NAP = HP4 / PTHR;
    static_assert(HP4 % PTHR == 0, "halo slots must fill whole DMA rounds");
    constexpr int BSZ = SSIE_TG * 4 * BN;
    constexpr int ROWB = HPW * 64;                                   // bytes per halo row
    f32x4* As0 = (f32x4*)smem_f;                    // [HP4]
    f32x4* Bs0 = As0 + HP4;                         // [2][BSZ]
    float* bias_s = (float*)(Bs0 + 2 * BSZ);        // [64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, li = lane & 31;
    for (int t = tid; t < 64; t += NTHR) bias_s[t] = (p.bias && t < p.Cout) ? p.bias[t] : 0.f;
    __syncthreads();

    const int total_tiles = p.N * tiles_y * tiles_x;
    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
#define N9_DECODE(T, N_, A0_, B0_) { int q_ = (T); B0_ = (q_ % tiles_x) * TWW; q_ /= tiles_x; A0_ = (q_ % tiles_y) * TH; N_ = q_ / tiles_y; }
    int n, a0, b0;
    N9_DECODE(tile, n, a0, b0)

    if (wave >= NWC) {
        // ------------------------------------------------ producers ------------------------------------------------
        const int ptid = tid - 64 * NWC, pwave = wave - NWC;
        int ahy[NAP], ahx[NAP], aj[NAP];
#pragma unroll
        for (int i = 0; i < NAP; ++i) {
            const int id = ptid + i * PTHR, pix = id >> 2;
            ahy[i] = pix / HPW; ahx[i] = pix - ahy[i] * HPW; aj[i] = (id & 3) ^ ssie_swz(pix);
        }
        const SrcSel s = ssie_only_src(p);
#define N9_HALO(N_, A0_, B0_)                                                                                 \
        {                                                                                                     \
            const int vy0_ = (A0_) - 4, vx0_ = (B0_) - 4;                                                     \
            if ((s.C & 31) == 0 && vy0_ >= 0 && vx0_ >= 0 && vy0_ + HPH <= p.Hv && vx0_ + HPW <= p.Wv) {     \
                const unsigned short* sb_ = (const unsigned short*)s.ptr + ((size_t)((N_) * s.Hs + vy0_) * s.Ws + vx0_) * s.cstride + s.coff; \
                _Pragma("unroll") for (int i_ = 0; i_ < NAP; ++i_)                                            \
                    GLDS16(sb_ + (ahy[i_] * s.Ws + ahx[i_]) * s.cstride + 8 * aj[i_], As0 + i_ * PTHR + pwave * 64); \
            } else {                                                                                          \
                _Pragma("unroll") for (int i_ = 0; i_ < NAP; ++i_)                                            \
                    GLDS16(ssie_virtual_addr_h(s, false, (N_), vy0_ + ahy[i_], vx0_ + ahx[i_], p.Hv, p.Wv, 8 * aj[i_]), As0 + i_ * PTHR + pwave * 64); \
            }                                                                                                 \
        }
#define N9_WEIGHTS(G, BUF)                                                                                    \
        {                                                                                                     \
            const f32x4* wsrc_ = (const f32x4*)p.wpacked + ((size_t)((G) * SSIE_TG) * 4) * 64;                \
            f32x4* bbuf_ = Bs0 + (BUF) * BSZ;                                                                 \
            for (int q_ = pwave; q_ < SSIE_TG * 4; q_ += NWP) GLDS16(wsrc_ + (size_t)q_ * 64 + lane, bbuf_ + q_ * 64); \
        }
        N9_HALO(n, a0, b0)
        N9_WEIGHTS(0, 0)
        int gstep = 0;
        while (tile < total_tiles) {
            const int ntile = tile + (int)gridDim.x;
            for (int g = 0; g < 9; ++g, ++gstep) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (g < 8) N9_WEIGHTS(g + 1, (gstep + 1) & 1)
                else if (ntile < total_tiles) N9_WEIGHTS(0, (gstep + 1) & 1)
            }
            asm volatile("s_barrier" ::: "memory");             // the consumers have read the halo tile for the last time
            tile = ntile;
            if (tile < total_tiles) { N9_DECODE(tile, n, a0, b0) N9_HALO(n, a0, b0) }
        }
#undef N9_WEIGHTS
#undef N9_HALO
        return;
    }

    // ---------------------------------------------------- consumers ----------------------------------------------------
    // byte offset inside the halo tile of (M-tile m, tap column dx, kernel row 0, k-quad 0)
    int aaddr[SSIE_TG][MT];
#pragma unroll
    for (int t = 0; t < SSIE_TG; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int hp = (2 * wave + (li >> 4)) * HPW + 16 * m + (li & 15) + t;
            aaddr[t][m] = (hp * 4 + (h ^ ssie_swz(hp))) * 16;
        }
    const float relu_lo = p.act == ACT_RELU ? 0.f : -3.0e38f;
    int lane_off[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) lane_off[m] = ((2 * wave + (li >> 4)) * p.Wout + 16 * m + (li & 15)) * p.out_cstride + 8 * h;

    int gstep = 0;
    while (tile < total_tiles) {
        f32x16 acc[MT][NT];
        {
            const float* bsrc = bias_s + 4 * h;
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 b = *(const f32x4*)(bsrc + 32 * c + 8 * g4);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[m][c][4 * g4 + j] = b[j];
                }
        }
        for (int g = 0; g < 9; ++g, ++gstep) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const char* Ab = (const char*)As0 + g * ROWB;
            const int xg = (g & 1) * 32;                        // odd kernel rows: the k-halves trade places (header)
            const f32x4* Bl = Bs0 + (gstep & 1) * BSZ + h * BN + li;
#define N9_LD(BF, AF, TL, SC)                                                                             \
            {                                                                                             \
                _Pragma("unroll") for (int c_ = 0; c_ < NT; ++c_) BF[c_] = Bl[((TL) * 4 + (SC) * 2) * BN + c_ * 32]; \
                _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_)                                         \
                    AF[m_] = *(const f32x4*)(Ab + (aaddr[TL][m_] ^ ((SC) ? (xg ^ 32) : xg)));             \
            }
#define N9_MFMA(BF, AF)                                                                                   \
            _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_)                                             \
            _Pragma("unroll") for (int c_ = 0; c_ < NT; ++c_) acc[m_][c_] = MFMA_BF16(BF[c_], AF[m_], acc[m_][c_]);
            {
                f32x4 bX[NT], bY[NT], aX[MT], aY[MT];
                N9_LD(bX, aX, 0, 0)
#pragma unroll
                for (int tl = 0; tl < SSIE_TG; ++tl) {
                    N9_LD(bY, aY, tl, 1)
                    N9_MFMA(bX, aX)
                    if (tl + 1 < SSIE_TG) N9_LD(bX, aX, (tl + 1 < SSIE_TG ? tl + 1 : 0), 0)
                    N9_MFMA(bY, aY)
                }
            }
#undef N9_LD
#undef N9_MFMA
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // end of tile: the halo buffer may be overwritten

        const bool inside = a0 + TH <= p.Hout && b0 + TWW <= p.Wout;
        if (inside) {
            unsigned short* tbase = (unsigned short*)p.out + ((size_t)(n * p.Hout + a0) * p.Wout + b0) * p.out_cstride + p.out_coff;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int g2 = 0; g2 < 4; g2 += 2) {
                        const f32x16& a = acc[m][c];
                        uint2 u[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            u[e] = make_uint2(ssie_pack2bf(fmaxf(a[4 * (g2 + e)], relu_lo), fmaxf(a[4 * (g2 + e) + 1], relu_lo)),
                                              ssie_pack2bf(fmaxf(a[4 * (g2 + e) + 2], relu_lo), fmaxf(a[4 * (g2 + e) + 3], relu_lo)));
                        *(uint4*)(tbase + lane_off[m] + 32 * c + 8 * g2) = ssie_pair_swap(u[0], u[1]);
                    }
        } else {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int oy = a0 + 2 * wave + (li >> 4), ox = b0 + 16 * m + (li & 15);
                const bool ok = oy < p.Hout && ox < p.Wout;
                unsigned short* ob = (unsigned short*)p.out + ((size_t)(n * p.Hout + oy) * p.Wout + ox) * p.out_cstride + p.out_coff + 8 * h;
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int g2 = 0; g2 < 4; g2 += 2) {
                        const f32x16& a = acc[m][c];
                        uint2 u[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            u[e] = make_uint2(ssie_pack2bf(fmaxf(a[4 * (g2 + e)], relu_lo), fmaxf(a[4 * (g2 + e) + 1], relu_lo)),
                                              ssie_pack2bf(fmaxf(a[4 * (g2 + e) + 2], relu_lo), fmaxf(a[4 * (g2 + e) + 3], relu_lo)));
                        const uint4 v = ssie_pair_swap(u[0], u[1]);
                        if (ok) *(uint4*)(ob + 32 * c + 8 * g2) = v;
                    }
            }
        }
        tile += (int)gridDim.x;
        if (tile < total_tiles) N9_DECODE(tile, n, a0, b0)
    }
#undef N9_DECODE
}



template __global__ void conv_fprop_bf16w_kernel<5, false, false>(const ConvParams);
template __global__ void conv_fprop_bf16w_kernel<5, true, false>(const ConvParams);
template __global__ void conv_fprop_bf16w_kernel<5, false, true>(const ConvParams);
template __global__ void conv_fprop_bf16w_kernel<5, true, true>(const ConvParams);

#define INST_H(NT, NA2, TH) template __global__ void conv_fprop_bf16_kernel<NT, NA2, TH>(const ConvParams);
INST_H(2, 3, 16) INST_H(2, 5, 16) INST_H(1, 3, 16) INST_H(1, 5, 16) INST_H(2, 5, 8)

int ssie_bf16_two_wgs = 1;            // ssie_debug_set_bf16_two_wgs: 0 = one workgroup per CU for the 32-channel-tile kernel too
extern "C" void ssie_debug_set_bf16_two_wgs(int v) { ssie_bf16_two_wgs = v; }
static size_t lds_bytes_h(const ConvParams& p, int nt)
{
    return 2 * ((size_t)p.hp_h * p.hp_w * 64 + (size_t)SSIE_TG * 4 * 32 * nt * 16) + (size_t)SSIE_MAX_TAPS * 4 + 16 + (size_t)p.Cout_pad * 8;
}

template <int NT, int NA2, int TH>
static int launch_h_t(const ConvParams& p, size_t lds, hipStream_t st)
{
    static unsigned seen = 0;
    ssie_allow_full_lds((const void*)conv_fprop_bf16_kernel<NT, NA2, TH>, seen);
    const size_t tiles = (size_t)p.N * p.tiles_y * p.tiles_x * p.co_blocks;
    // 32-channel tiles (conv0, the R|I layer): 18 - 36 MFMAs per wave and tile against a DMA round trip of several microseconds, with
    // ONE tile of prefetch - a workgroup per CU waited ~15k cycles per tile for its operands.  Two workgroups fit a CU (2 x 77 KB of
    // LDS, 116 registers) and cover each other's flight.
    const size_t per_cu = (NT == 1 && ssie_bf16_two_wgs && 2 * (lds + 256) <= 160 * 1024 && tiles >= 512) ? 2 : 1;
    const size_t wgs = tiles < 256 * per_cu ? tiles : 256 * per_cu;
    hipLaunchKernelGGL((conv_fprop_bf16_kernel<NT, NA2, TH>), dim3((unsigned)wgs), dim3(512), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 61;
}

// p must come from ssie_make_conv_bf16 (th = 16 for stride 1, 8 for stride 2; 32-channel chunks)
int ssie_bf16_dynamic_queue = 0;      // see below
int ssie_bf16_ws = 3;                 // 16 x 32 geometry: 0 = eight-wave kernel, 1 = wave-specialised 8 consumers + 4 producers, 2 = eight-wave kernel with the DMA interleaved between the taps (measured slower), 3 = wave-specialised 4 consumers (64 positions x 64 channels x 2 each) + 4 producers for the 9-tap layers
extern "C" void ssie_debug_set_bf16_ws(int v) { ssie_bf16_ws = v; }
int ssie_bf16_conv9 = 1, ssie_bf16_conv9_min_tiles = 256;   // the 9 x 9 layer on conv9x9_bf16ws_kernel (16 x 32 tiles) from this many tiles on
extern "C" void ssie_debug_set_bf16_conv9(int v) { ssie_bf16_conv9 = v; }
extern "C" void ssie_debug_set_bf16_conv9_min_tiles(int v) { ssie_bf16_conv9_min_tiles = v; }
int ssie_bf16_ws_geo = 1, ssie_bf16_ws_geo_min_tiles = 256;   // stride-2 / transposed 64-channel layers on the wave-specialised kernel
extern "C" void ssie_debug_set_bf16_ws_geo(int v) { ssie_bf16_ws_geo = v; }
extern "C" void ssie_debug_set_bf16_ws_geo_min_tiles(int v) { ssie_bf16_ws_geo_min_tiles = v; }
int ssie_bf16_resw = 1;               // single-source 9-tap layers of <= 64 input channels: weights resident in LDS (1 = 8 consumer waves, 2 = 4; 0 = off)
extern "C" void ssie_debug_set_bf16_resw(int v) { ssie_bf16_resw = v; }
int ssie_launch_fprop_bf16(const ConvParams& p_in, hipStream_t st)
{
    // Static tile assignment (tile += gridDim.x).  The dynamic queue of the fp32 kernels draws every tile with a returning
    // atomic on ONE counter; a bf16 tile is ~16x shorter than an fp32 one, so at 2048 tiles per launch the same-address
    // atomics (serialised at the memory side) and their in-order return behind the prefetch DMA set the pace.
    ConvParams p = p_in;
    if (!ssie_bf16_dynamic_queue) p.tile_counter = nullptr;
    if (ssie_bf16_conv9 && p.ntaps == 81 && p.si == 1 && p.so == 1 && p.py == 0 && p.px == 0 && p.nchunks == 1 && p.Cout_pad == 64 &&
        p.nsrc == 1 && p.src[0].sy == 1.f && p.src[0].sx == 1.f && p.min_dy == -4 && p.min_dx == -4 && p.Ho == p.Hout && p.Wo == p.Wout &&
        p.out_bf16 && !p.addsrc && !p.out2 && (p.Cout % 32) == 0 && p.act != ACT_SIGMOID) {
        bool rowmajor = true;
        for (int t = 0; t < 81; ++t) rowmajor = rowmajor && p.tap_dy[t] == t / 9 - 4 && p.tap_dx[t] == t % 9 - 4;
        const int tx = ssie_ceil_div(p.Wo, 32), ty = ssie_ceil_div(p.Ho, 16);
        const size_t tiles = (size_t)p.N * tx * ty;
        if (rowmajor && tiles >= (size_t)ssie_bf16_conv9_min_tiles) {
            static unsigned seen9 = 0;
            ssie_allow_full_lds((const void*)conv9x9_bf16ws_kernel, seen9);
            const size_t lds9 = (size_t)(24 * 40 * 4 + 2 * SSIE_TG * 4 * 64) * 16 + 64 * 4;
            hipLaunchKernelGGL(conv9x9_bf16ws_kernel, dim3((unsigned)(tiles < 256 ? tiles : 256)), dim3(768), lds9, st, p, tx, ty);
            return hipGetLastError() == hipSuccess ? 0 : 66;
        }
    }
    const int nt = (p.Cout_pad % 64 == 0) ? 2 : 1;
    const int na2 = (p.hp_h * p.hp_w * 4 + 511) / 512;
    const size_t lds = lds_bytes_h(p, nt);
    if (na2 > 5 || lds > 160 * 1024) return 62;
    if (p.tw == 32) {                      // geometry built for the wide kernel (ssie_make_conv_bf16)
        static unsigned seen_a = 0, seen_b = 0, seen_e = 0, seen_f = 0;
        ssie_allow_full_lds((const void*)conv_fprop_bf16w_kernel<5, false, false>, seen_a);
        ssie_allow_full_lds((const void*)conv_fprop_bf16w_kernel<5, true, false>, seen_b);
        ssie_allow_full_lds((const void*)conv_fprop_bf16w_kernel<5, false, true>, seen_e);
        ssie_allow_full_lds((const void*)conv_fprop_bf16w_kernel<5, true, true>, seen_f);
        const size_t tiles = (size_t)p.N * p.tiles_y * p.tiles_x * p.co_blocks;
        const dim3 grid((unsigned)(tiles < 256 ? tiles : 256));
#ifdef SSIE_STAMP
        const bool stamp_this = g_stamp_host_buf && g_stamp_launch++ == g_stamp_target;
        if (stamp_this) { hipStreamSynchronize(st); hipMemcpyToSymbol(HIP_SYMBOL(ssie_stamp_buf_h), &g_stamp_host_buf, sizeof(void*)); }
#endif
        // layers that end in the lean epilogue (bf16 output, ReLU or none, no residual / second output, whole 32-channel groups)
        // with 9 taps or 1 run the wave-specialised kernel
        const bool lean = p.out_bf16 && !p.addsrc && !p.out2 && (p.Cout % 32) == 0 && p.act != ACT_SIGMOID;
        // the same with a residual added to the activated value (deconv1-3 of the illumination net)
        if (ssie_bf16_ws >= 1 && ssie_bf16_resw && p.out_bf16 && p.addsrc && !p.out2 && (p.Cout % 32) == 0 && p.act != ACT_SIGMOID &&
            p.ntaps == 9 && p.nsrc == 1 && p.nchunks <= 2 && p.co_blocks == 1 && p.hp_h * p.hp_w * 4 <= 2560 && !p.tile_counter) {
            static unsigned seen_ra = 0;
            ssie_allow_full_lds((const void*)conv_fprop_bf16ws_kernel<9, true, 8, true, true>, seen_ra);
            hipLaunchKernelGGL((conv_fprop_bf16ws_kernel<9, true, 8, true, true>), grid, dim3(768), lds, st, p);
        }
        else if (ssie_bf16_ws >= 1 && ssie_bf16_resw && lean && p.ntaps == 9 && p.nsrc == 1 && p.nchunks <= 2 && p.co_blocks == 1 &&
            p.hp_h * p.hp_w * 4 <= 2560 && !p.tile_counter) {
            static unsigned seen_rw[2] = {0, 0};
            ssie_allow_full_lds((const void*)conv_fprop_bf16ws_kernel<9, true, 8, true>, seen_rw[0]);
            ssie_allow_full_lds((const void*)conv_fprop_bf16ws_kernel<9, true, 4, true>, seen_rw[1]);
            if (ssie_bf16_resw == 1) hipLaunchKernelGGL((conv_fprop_bf16ws_kernel<9, true, 8, true>), grid, dim3(768), lds, st, p);
            else hipLaunchKernelGGL((conv_fprop_bf16ws_kernel<9, true, 4, true>), grid, dim3(512), lds, st, p);
        }
        else if (ssie_bf16_ws == 3 && lean && p.ntaps == 9 && p.hp_h * p.hp_w * 4 <= 2560 && !p.tile_counter) {
            static unsigned seen_w4[2] = {0, 0};
            ssie_allow_full_lds((const void*)conv_fprop_bf16ws_kernel<9, false, 4, false>, seen_w4[0]);
            ssie_allow_full_lds((const void*)conv_fprop_bf16ws_kernel<9, true, 4, false>, seen_w4[1]);
            if (p.nsrc == 1) hipLaunchKernelGGL((conv_fprop_bf16ws_kernel<9, true, 4, false>), grid, dim3(512), lds, st, p);
            else hipLaunchKernelGGL((conv_fprop_bf16ws_kernel<9, false, 4, false>), grid, dim3(512), lds, st, p);
        }
        else if ((ssie_bf16_ws == 1 || ssie_bf16_ws == 3) && lean && (p.ntaps == 9 || p.ntaps == 1) && p.hp_h * p.hp_w * 4 <= 2560 && !p.tile_counter) {
            static unsigned seen_ws[4] = {0, 0, 0, 0};
            ssie_allow_full_lds((const void*)conv_fprop_bf16ws_kernel<9, false, 8, false>, seen_ws[0]);
            ssie_allow_full_lds((const void*)conv_fprop_bf16ws_kernel<9, true, 8, false>, seen_ws[1]);
            ssie_allow_full_lds((const void*)conv_fprop_bf16ws_kernel<1, false, 8, false>, seen_ws[2]);
            ssie_allow_full_lds((const void*)conv_fprop_bf16ws_kernel<1, true, 8, false>, seen_ws[3]);
            if (p.ntaps == 9 && p.nsrc == 1) hipLaunchKernelGGL((conv_fprop_bf16ws_kernel<9, true, 8, false>), grid, dim3(768), lds, st, p);
            else if (p.ntaps == 9) hipLaunchKernelGGL((conv_fprop_bf16ws_kernel<9, false, 8, false>), grid, dim3(768), lds, st, p);
            else if (p.nsrc == 1) hipLaunchKernelGGL((conv_fprop_bf16ws_kernel<1, true, 8, false>), grid, dim3(768), lds, st, p);
            else hipLaunchKernelGGL((conv_fprop_bf16ws_kernel<1, false, 8, false>), grid, dim3(768), lds, st, p);
        }
        else if (ssie_bf16_ws == 2 && p.ntaps == 9 && p.nsrc == 1) hipLaunchKernelGGL((conv_fprop_bf16w_kernel<5, true, true>), grid, dim3(512), lds, st, p);
        else if (ssie_bf16_ws == 2 && p.ntaps == 9) hipLaunchKernelGGL((conv_fprop_bf16w_kernel<5, false, true>), grid, dim3(512), lds, st, p);
        else if (p.nsrc == 1) hipLaunchKernelGGL((conv_fprop_bf16w_kernel<5, true, false>), grid, dim3(512), lds, st, p);
        else hipLaunchKernelGGL((conv_fprop_bf16w_kernel<5, false, false>), grid, dim3(512), lds, st, p);
#ifdef SSIE_STAMP
        if (stamp_this) { hipStreamSynchronize(st); void* z = nullptr; hipMemcpyToSymbol(HIP_SYMBOL(ssie_stamp_buf_h), &z, sizeof(void*)); }
#endif
        return hipGetLastError() == hipSuccess ? 0 : 65;
    }
    {
        // 64-channel-block layers of the other geometries on the wave-specialised kernel (4 consumer + 4 producer waves): the stride-2
        // 3 x 3 layers (8 x 16 tiles) and the four parity classes of the transposed convolutions (16 x 16 tiles, 4 / 2 / 2 / 1 taps)
        const bool lean = p.out_bf16 && !p.addsrc && !p.out2 && (p.Cout % 32) == 0 && p.act != ACT_SIGMOID;
        const size_t tiles = (size_t)p.N * p.tiles_y * p.tiles_x * p.co_blocks;
        const dim3 grid((unsigned)(tiles < 256 ? tiles : 256));
        if (ssie_bf16_ws_geo && ssie_bf16_ws >= 1 && lean && nt == 2 && p.nsrc == 1 && p.hp_h * p.hp_w * 4 <= 2560 && !p.tile_counter && p.tw == 16 &&
            tiles >= (size_t)ssie_bf16_ws_geo_min_tiles) {
#define WSG_LAUNCH(TG_, GEO_, K_)                                                                             \
            { static unsigned seen_g = 0;                                                                     \
              ssie_allow_full_lds((const void*)conv_fprop_bf16ws_kernel<TG_, true, 4, false, false, GEO_>, seen_g); \
              hipLaunchKernelGGL((conv_fprop_bf16ws_kernel<TG_, true, 4, false, false, GEO_>), grid, dim3(512), lds, st, p); \
              return hipGetLastError() == hipSuccess ? 0 : 67; }
            if (p.th == 8 && p.si == 2 && p.ntaps == 9) WSG_LAUNCH(9, 1, 0)
            if (p.th == 16 && p.si == 1 && p.ntaps == 4) WSG_LAUNCH(4, 2, 1)
            if (p.th == 16 && p.si == 1 && p.ntaps == 2) WSG_LAUNCH(2, 2, 2)
            if (p.th == 16 && p.si == 1 && p.ntaps == 1) WSG_LAUNCH(1, 2, 3)
#undef WSG_LAUNCH
        }
    }
    if (p.th == 8) return nt == 2 ? launch_h_t<2, 5, 8>(p, lds, st) : 63;
    if (p.th != 16) return 64;
    if (nt == 2) return na2 <= 3 ? launch_h_t<2, 3, 16>(p, lds, st) : launch_h_t<2, 5, 16>(p, lds, st);
    return na2 <= 3 ? launch_h_t<1, 3, 16>(p, lds, st) : launch_h_t<1, 5, 16>(p, lds, st);
}
