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
//   * halo layout [channel half][column phase = col & 3][halo row][col >> 2] 16-byte slots: the 16 lanes of a k-group read every
//     fourth column = consecutive slots of one phase plane, the four k-groups the two 8-byte halves of two slot planes: conflict-free
//   * B^T d B in two halves of three output columns each (the 1-D transform's outputs {0,1,2} and {3,4,5} share no subexpression, so
//     splitting costs no arithmetic): the 6 x 6 patch is read twice per step instead of holding 72 registers of it
//   * U comes out of the weight-packing launch (PackDesc.wino = 2) in exactly the LDS image of a step: [xi][half][g][channel][2]
#include "conv_device.h"
#include <type_traits>

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int V_TH = 16, V_TW = 64, V_HPH = 18, V_HPW = 66, V_CK = 8;
constexpr int V_IDX = 17;                           // column groups per phase plane (66 columns)
constexpr int V_PHASE = V_HPH * V_IDX;              // slots per (half, phase) plane: 306
constexpr int V_HALF = 4 * V_PHASE;                 // 16-byte slots per channel half: 1224
constexpr int V_HP = 2 * V_HALF;                    // slots per halo buffer: 2448 (the fifth DMA round is masked to them)
constexpr int V_ROUNDS = (V_HP + 511) / 512;        // 5
constexpr int V_BSZ = 36 * 2 * 4 * 16 * 2 / 4;      // float4 per U step: 2304 (36 KB)
constexpr int V_UPIECES = V_BSZ / 64;               // 36 DMA pieces of 1 KB

// packed fp32 arithmetic as inline asm (hipcc splits float2 operations whose halves are consumed one by one by MFMAs into scalar
// instructions, and every VALU instruction costs matrix-pipe time here).  The hazard recognizer does not see an asm as a VALU write:
// transformed values pass through V4_FENCE6 (s_nop) before the first MFMA reads them.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// a * k + c and c - a * k, k a uniform constant pair
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 k, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(k), "v"(c)); return r; }
__device__ __forceinline__ f32x2 pk_fnma(f32x2 a, f32x2 k, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(k), "v"(c)); return r; }
#define V4_FENCE6(a, b, c, d, e, f) asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f))

// 16 outputs of one lane = the 4 x 4 pixels of one tile, one channel: k = 4 y + x
#define V4_EOFF(k) ((long)((k) >> 2) * rowstride + (long)((k) & 3) * pixstride)
template <typename PT>
__device__ __forceinline__ void wino4_epilogue16(const PT& p, float v[16], size_t o0, long rowstride, long pixstride, float bv)
{
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
        for (int k = 0; k < 16; ++k) y[k] = mp[V4_EOFF(k)];
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
        for (int k = 0; k < 16; ++k) o2[V4_EOFF(k)] = v[k];
    }
    if (p.addsrc) {
        const float* ap = p.addsrc + o0;
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = ap[V4_EOFF(k)];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] += a[k];
    }
    float* ob = p.out + o0;
    if (p.accumulate) {
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = ob[V4_EOFF(k)];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] += a[k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) ob[V4_EOFF(k)] = v[k];
}

template <typename PT>
__device__ __forceinline__ void wino4_epilogue16_ragged(const PT& p, const float v[16], size_t o0, long rowstride, long pixstride, float bv,
                                                         int oy, int ox)
{
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int y = oy + (k >> 2), x = ox + (k & 3);
        if (y >= p.Hout || x >= p.Wout) continue;
        const size_t o = o0 + V4_EOFF(k);
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

template <bool SINGLE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino4_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int NTHR = 512;
    f32x4* As0 = (f32x4*)smem_f;                    // [2][V_HP]
    f32x4* Bs0 = As0 + 2 * V_HP;                    // [2][V_BSZ]
    int* s_next = (int*)(Bs0 + 2 * V_BSZ);
    int* dma_tab = s_next + 16;                     // [V_ROUNDS][NTHR]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, nh = wave >> 2;        // tile row of the workgroup tile, 16-channel half of the 32-channel block
    const int g = lane >> 4, tx = lane & 15;
    // this lane's patch: halo rows 4 wm + a, halo columns 4 tx + b -> phase b & 3, column group tx + (b >> 2); channel pair g lives in
    // the 8-byte half g & 1 of slot plane g >> 1
    const int abase = ((((g >> 1) * 4) * V_HPH + 4 * wm) * V_IDX + tx) * 16 + (g & 1) * 8;
#define V4_AOFF(a, b) (((((b) & 3) * V_HPH + (a)) * V_IDX + ((b) >> 2)) * 16)
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
    // Per-lane DMA table (conv_wino.hip): for each of the lane's 5 halo slots the byte offset of its source relative to the tile origin
    // (24 bits; sources of one channel stride) | halo column << 24 (7 bits) | channel half << 31.  [different strides: pixel offset]
    const int cs_common = SINGLE ? p.src[0].cstride
                                 : ((p.nsrc < 2 || p.src[1].cstride == p.src[0].cstride) && (p.nsrc < 3 || p.src[2].cstride == p.src[0].cstride)
                                    ? p.src[0].cstride : 0);
#pragma unroll
    for (int i = 0; i < V_ROUNDS; ++i) {
        const int id = i * NTHR + tid;
        const int h = id >= V_HALF, r = id - h * V_HALF;
        const int ph = (r >= V_PHASE) + (r >= 2 * V_PHASE) + (r >= 3 * V_PHASE), r2 = r - ph * V_PHASE;
        const int hy = (r2 * 3856) >> 16;                       // r2 / 17 for r2 < 384
        const int hx = 4 * (r2 - hy * V_IDX) + ph;
        const bool real = id < V_HP && hx < V_HPW;
        const int pix = hy * p.Wv + hx;
        const int lo = cs_common ? (pix * cs_common + 4 * h) * 4 : pix;
        dma_tab[i * NTHR + tid] = real ? (int)((unsigned)lo | ((unsigned)hx << 24) | ((unsigned)h << 31)) : 0;
    }
#define V4_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)
#define V4_BLDS(rs, lptr, vo, so) __builtin_amdgcn_raw_ptr_buffer_load_lds((rs), (__attribute__((address_space(3))) void*)(lptr), 16, (int)(vo), (int)(so), 0, 0)
    // DMA of one K step: U = 36 pieces of 1 KB (wave w: pieces w, w + 8, ...), halo = 5 rounds of 512 slots
#define V4_PREFETCH(STEP, N_, A0_, B0_, CB_, BUF)                                                             \
    {                                                                                                         \
        f32x4* bbuf_ = Bs0 + (BUF) * V_BSZ;                                                                   \
        const __amdgpu_buffer_rsrc_t ur_ = V4_RSRC((const f32x4*)p.wpacked + ((size_t)(STEP) * p.co_blocks + (CB_)) * V_BSZ, V_BSZ * 16); \
        _Pragma("unroll") for (int q_ = 0; q_ < (V_UPIECES + 7) / 8; ++q_) {                                  \
            const int pc_ = q_ * 8 + wave;                                                                    \
            if (pc_ < V_UPIECES) V4_BLDS(ur_, bbuf_ + pc_ * 64, lane * 16, pc_ * 1024);                       \
        }                                                                                                     \
        const SrcSel s_ = SINGLE ? ssie_only_src(p) : ssie_pick_src(p, (STEP) * V_CK);                        \
        const int vy0_ = (A0_) - 1, vx0_ = (B0_) - 1;                                                         \
        f32x4* abuf_ = As0 + (BUF) * V_HP;                                                                    \
        const int c0_ = (STEP) * V_CK - s_.cbeg;                                                              \
        /* one resource per (source, image): rows outside the image are out-of-range offsets -> zeros */      \
        const __amdgpu_buffer_rsrc_t ar_ = V4_RSRC(s_.ptr + (size_t)(N_) * p.Hv * p.Wv * s_.cstride,          \
                                                   (unsigned)(p.Hv * p.Wv * s_.cstride) * 4u);                \
        const int tb_ = ((vy0_ * p.Wv + vx0_) * s_.cstride + s_.coff + c0_) * 4;      /* bytes, may be negative */ \
        const unsigned xlo_ = max(0, -vx0_), xn_ = min(V_HPW, p.Wv - vx0_) - xlo_;                            \
        const int jn_ = (s_.C - c0_ + 3) >> 2;                                                                \
        const unsigned cs4_ = (unsigned)s_.cstride * 4u;                                                      \
        unsigned te_[V_ROUNDS];                                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < V_ROUNDS; ++i_) te_[i_] = (unsigned)dma_tab[i_ * NTHR + tid]; \
        if (xn_ == (unsigned)V_HPW && jn_ >= 2) {          /* no column outside the image, full channel step */ \
            _Pragma("unroll") for (int i_ = 0; i_ < V_ROUNDS; ++i_) {                                         \
                const unsigned lo_ = te_[i_] & 0xffffffu;                                                     \
                const unsigned off_ = (unsigned)tb_ + (cs_common ? lo_ : __umul24(lo_, cs4_) + ((te_[i_] >> 31) << 4)); \
                if (i_ < V_ROUNDS - 1 || tid < V_HP - (V_ROUNDS - 1) * NTHR) V4_BLDS(ar_, abuf_ + i_ * NTHR + wave * 64, off_, 0); \
            }                                                                                                 \
        } else {                                                                                              \
            _Pragma("unroll") for (int i_ = 0; i_ < V_ROUNDS; ++i_) {                                         \
                const unsigned e_ = te_[i_], lo_ = e_ & 0xffffffu, hx_ = (e_ >> 24) & 127, j_ = e_ >> 31;     \
                const bool ok_ = hx_ - xlo_ < xn_ && (int)j_ < jn_;                                           \
                const unsigned off_ = (unsigned)tb_ + (cs_common ? lo_ : __umul24(lo_, cs4_) + (j_ << 4));    \
                if (i_ < V_ROUNDS - 1 || tid < V_HP - (V_ROUNDS - 1) * NTHR) V4_BLDS(ar_, abuf_ + i_ * NTHR + wave * 64, ok_ ? off_ : 0x80000000u, 0); \
            }                                                                                                 \
        }                                                                                                     \
    }

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int n, a0, b0, cb;
    V4_DECODE(tile, n, a0, b0, cb)
    int gstep = 0;
    __syncthreads();                                // the DMA table is complete before anyone reads it
    V4_PREFETCH(0, n, a0, b0, cb, 0)
    int fetched = 0x7fffffff;
    const f32x2 k2 = {2.f, 2.f}, k4 = {4.f, 4.f}, k5 = {5.f, 5.f};

    while (tile < total_tiles) {
        f32x4 acc[36];
        const int co = cb * 32 + nh * 16 + tx;
        const float bv = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (step == (nsteps > 1 ? 1 : 0)) {
                ntile = *s_next;
                if (ntile < total_tiles) V4_DECODE(ntile, nn, na0, nb0, ncb)
            }
            const char* Ab = (const char*)(As0 + buf * V_HP) + abase;
            const char* Bl = (const char*)(Bs0 + buf * V_BSZ) + bbase;
#pragma unroll
            for (int H = 0; H < 2; ++H) {
                // horizontal transform of the six patch rows: outputs 3 H .. 3 H + 2
                f32x2 rt[6][3];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    f32x2 d[6];
#pragma unroll
                    for (int b = 0; b < 6; ++b) d[b] = *(const f32x2*)(Ab + V4_AOFF(a, b));
                    if (H == 0) {
                        rt[a][0] = pk_fma(d[0], k4, pk_fnma(d[2], k5, d[4]));
                        const f32x2 s = pk_fnma(d[2], k4, d[4]), t = pk_fnma(d[1], k4, d[3]);
                        rt[a][1] = pk_add(s, t); rt[a][2] = pk_sub(s, t);
                    } else {
                        const f32x2 s = pk_sub(d[4], d[2]), t = pk_sub(d[3], d[1]);
                        rt[a][0] = pk_fma(t, k2, s); rt[a][1] = pk_fnma(t, k2, s);
                        rt[a][2] = pk_fma(d[1], k4, pk_fnma(d[3], k5, d[5]));
                    }
                }
#pragma unroll
                for (int bq = 0; bq < 3; ++bq) {
                    // vertical transform of column 3 H + bq -> the six xi = 6 i + 3 H + bq
                    f32x2 v[6];
                    {
                        const f32x2 d0 = rt[0][bq], d1 = rt[1][bq], d2 = rt[2][bq], d3 = rt[3][bq], d4 = rt[4][bq], d5 = rt[5][bq];
                        v[0] = pk_fma(d0, k4, pk_fnma(d2, k5, d4));
                        const f32x2 s = pk_fnma(d2, k4, d4), t = pk_fnma(d1, k4, d3);
                        v[1] = pk_add(s, t); v[2] = pk_sub(s, t);
                        const f32x2 s2 = pk_sub(d4, d2), t2 = pk_sub(d3, d1);
                        v[3] = pk_fma(t2, k2, s2); v[4] = pk_fnma(t2, k2, s2);
                        v[5] = pk_fma(d1, k4, pk_fnma(d3, k5, d5));
                    }
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
                    // the next step's DMA goes out behind the first column's MFMAs (waves 0-3) / the second column's (waves 4-7, their SIMD
                    // partners: a wave sits at DMA issue while its previous pieces are in flight, so partners must not do it together)
                    if (H == 0 && bq == (wave >= 4 ? 1 : 0)) {
                        if (step + 1 < nsteps) V4_PREFETCH(step + 1, n, a0, b0, cb, buf ^ 1)
                        else if (ntile < total_tiles) V4_PREFETCH(0, nn, na0, nb0, ncb, buf ^ 1)
                    }
                }
            }
            if (tid == 0 && step == 0 && nsteps > 1 && p.tile_counter)
                fetched = (int)gridDim.x + atomicAdd(p.tile_counter, 1);
        };
        step_body(std::true_type{}, 0); ++gstep;
        for (int step = 1; step < nsteps; ++step, ++gstep) step_body(std::false_type{}, step);

        // output transform (lane-local: register r of all 36 accumulators = tile column 4 g + r, channel co) + epilogue
        if (co < p.Cout) {
            const long rowstride = (long)p.Wout * p.out_cstride, pixstride = p.out_cstride;
            const bool full = a0 + V_TH <= p.Hout && b0 + V_TW <= p.Wout;
            const int oy0 = a0 + 4 * wm;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t[4][6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const float m0 = acc[0 * 6 + j][r], m1 = acc[1 * 6 + j][r], m2 = acc[2 * 6 + j][r], m3 = acc[3 * 6 + j][r],
                                m4 = acc[4 * 6 + j][r], m5 = acc[5 * 6 + j][r];
                    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                    t[0][j] = m0 + s1 + s2; t[1][j] = d1 + 2.f * d2; t[2][j] = s1 + 4.f * s2; t[3][j] = d1 + 8.f * d2 + m5;
                }
                float y[16];
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) {
                    const float s1 = t[yy][1] + t[yy][2], d1 = t[yy][1] - t[yy][2], s2 = t[yy][3] + t[yy][4], d2 = t[yy][3] - t[yy][4];
                    y[4 * yy + 0] = t[yy][0] + s1 + s2; y[4 * yy + 1] = d1 + 2.f * d2;
                    y[4 * yy + 2] = s1 + 4.f * s2; y[4 * yy + 3] = d1 + 8.f * d2 + t[yy][5];
                }
                const int ox0 = b0 + 4 * (4 * g + r);
                const size_t o0 = ((size_t)(n * p.Hout + oy0) * p.Wout + ox0) * p.out_cstride + p.out_coff + co;
                if (full) wino4_epilogue16(p, y, o0, rowstride, pixstride, bv);
                else wino4_epilogue16_ragged(p, y, o0, rowstride, pixstride, bv, oy0, ox0);
            }
        }
        n = nn; a0 = na0; b0 = nb0; cb = ncb; tile = ntile;
    }
#undef V4_PREFETCH
#undef V4_DECODE
#undef V4_RSRC
#undef V4_BLDS
#undef V4_AOFF
}

template __global__ void conv_wino4_kernel<false>(const ConvParams);
template __global__ void conv_wino4_kernel<true>(const ConvParams);

size_t ssie_wino4_lds_bytes() { return (size_t)(2 * V_HP + 2 * V_BSZ) * 16 + 64 + (size_t)V_ROUNDS * 512 * 4; }

// p from ssie_conv_to_wino4 (layer_ops.hip: ssie_wino4_eligible has checked sources, sizes and the 24-bit slot offsets)
int ssie_launch_fprop_wino4(const ConvParams& p, hipStream_t st)
{
    if (p.ntaps != 9 || p.si != 1 || p.so != 1 || p.py || p.px || p.min_dy != -1 || p.min_dx != -1 || p.Cout_pad % 32) return 34;
    if (p.th != V_TH || p.tw != V_TW || p.hp_h != V_HPH || p.hp_w != V_HPW || p.co_blocks != p.Cout_pad / 32) return 35;
    for (int s = 0; s < p.nsrc; ++s)
        if (p.src[s].sy != 1.f || p.src[s].sx != 1.f || p.src[s].Hs != p.Hv || p.src[s].Ws != p.Wv ||
            (size_t)(V_HPH * p.Wv + V_HPW) * p.src[s].cstride * 4 >= (1u << 24) || (size_t)p.Hv * p.Wv * p.src[s].cstride * 4 >= (1u << 31)) return 36;
    static unsigned seen[2] = {0, 0};
    ssie_allow_full_lds((const void*)conv_wino4_kernel<false>, seen[0]);
    ssie_allow_full_lds((const void*)conv_wino4_kernel<true>, seen[1]);
    const size_t tiles = (size_t)p.N * p.tiles_y * p.tiles_x * p.co_blocks;
    const dim3 grid((unsigned)(tiles < 256 ? tiles : 256));
    const size_t lds = ssie_wino4_lds_bytes();
    if (lds > 160 * 1024) return 37;
    if (p.nsrc == 1) hipLaunchKernelGGL((conv_wino4_kernel<true>), grid, dim3(512), lds, st, p);
    else hipLaunchKernelGGL((conv_wino4_kernel<false>), grid, dim3(512), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 38;
}
