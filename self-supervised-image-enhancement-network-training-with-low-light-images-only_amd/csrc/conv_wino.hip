// conv_wino_kernel: the stride-1 3 x 3 convolutions (forward and data gradient) as Winograd F(2x2, 3x3) on the fp32 MFMA.
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      d = 4 x 4 input patch, g = 3 x 3 kernel, Y = 2 x 2 outputs
// 16 multiplications per 2 x 2 outputs and channel pair instead of 36: the 16 "transform positions" xi are 16 independent
// GEMMs  M[xi][tile][co] = sum_ci V[xi][tile][ci] U[xi][ci][co]  which run on v_mfma_f32_32x32x2_f32.
//   * U = G g G^T is produced by the weight-packing launch (pack_weights: PackDesc.wino) in the layout of a 16-"tap" packed
//     weight, so a K-chunk of it is DMA'd into LDS exactly like a tap group of the direct kernels
//   * the raw 18 x 34 halo tile of a 16-channel chunk is DMA'd into LDS like in conv_fprop_v2w_kernel (same swizzle, same
//     virtual-input addressing: channel concatenation, nearest up-sampling and zero padding resolved per slot)
//   * V = B^T d B is never materialised: each lane reads the 4 x 4 patch of ITS Winograd tile (16 ds_read_b128 per 8
//     channels) and transforms it in registers (32 float4 additions) right before the MFMAs that consume it
//   * one workgroup = 4 waves = 16 x 32 output positions (8 x 16 Winograd tiles) x 32 output channels; wave w owns tile rows
//     2w, 2w+1 (32 tiles = the M dimension of the MFMA) and ALL 16 xi: 16 accumulator tiles = 256 registers, which is why
//     the kernel runs one wave per SIMD (512 registers per lane); the output transform A^T M A is then lane-local
//   * persistent workgroups + dynamic tile queue + cross-tile prefetch as in the v2 kernels
#include "conv_device.h"

__device__ f32x4 wino_zero_page[4];   // zero-initialised: source of padding slots

// Diagnostic build only (-DSSIE_STAMP, tools/stamp_wino.py): wave 0's s_memtime per phase, summed per workgroup:
// [0] start [1] barrier waits [2] output transform (accumulators -> 2 x 2 outputs) [3] end [4] step prologue (first LDS reads +
// transform) [5] epilogue stores + tile bookkeeping [6] tiles [7] the 8 MFMA groups [8] epilogue stores.  The shipped library never executes a stamp.
#ifdef SSIE_STAMP
__device__ unsigned long long* ssie_stamp_buf_wino = nullptr;
extern "C" int ssie_debug_set_stamp_buffer_wino(void* buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(ssie_stamp_buf_wino), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#define ST_DECL unsigned long long st_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t_ = __builtin_amdgcn_s_memtime(); st_[0] = st_t_;
#define ST_ACC(k) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[k] += t_ - st_t_; st_t_ = t_; } while (0)
#define ST_FLUSH do { st_[3] = __builtin_amdgcn_s_memtime(); if (ssie_stamp_buf_wino && threadIdx.x == 0) \
    for (int k_ = 0; k_ < 16; ++k_) ssie_stamp_buf_wino[(size_t)blockIdx.x * 16 + k_] = st_[k_]; } while (0)
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
    // select by mask arithmetic: written as `ok ? a : z` hipcc turns the address computation into a branch per slot, which
    // splits the MFMA groups into basic blocks the scheduler cannot interleave across
    const unsigned long long a = (unsigned long long)(s.ptr + off), z = (unsigned long long)wino_zero_page;
    const unsigned long long m = ok ? ~0ull : 0ull;
    return (const f32x4*)((a & m) | (z & ~m));
}

constexpr int W_TH = 16, W_TW = 32, W_HPH = 18, W_HPW = 34, W_HP4 = W_HPH * W_HPW * 4;   // 2448 16-byte slots per halo tile
constexpr int W_NA = (W_HP4 + 255) / 256;                                               // 10 DMA slots per lane
constexpr int W_HPB = W_NA * 256;                                                       // 16-byte slots per halo BUFFER (tile + padding)
constexpr int W_PLANE = W_HPB / 4;          // slots per channel-quad plane (612 used)
constexpr int W_HALF = W_HPH * (W_HPW / 2);   // slots per column-parity half plane (18 rows x 17 columns)
static_assert(W_PLANE >= 2 * W_HALF, "halo plane too small");
constexpr int W_BSZ = 16 * 4 * 32;                                                      // float4 per U chunk (16 xi x 16 ci x 32 co)

// Fused epilogue of 16 outputs of one lane = the 2 x 2 output pixels of 4 Winograd tiles (accumulator registers r = 4*RQ .. 4*RQ+3).
// Register r = Winograd tile (row r>>3, column (r&3) + 8*((r>>2)&1) [+ 4h, in o0]) of the wave's 2 x 16 tiles; element k of the
// pass = (tile rr = k&3, pixel e = k>>2): every offset is a compile-time multiple of two run-time strides.  Same order of the
// fused extras as ssie_epilogue_full (conv_device.h), each one loading its 16 operands back-to-back before the first use.
#define WN_TOFF(r) ((long)(2 * ((r) >> 3)) * rowstride + (long)(2 * (((r) & 3) + 8 * (((r) >> 2) & 1))) * pixstride)
#define WN_EOFF(k) (WN_TOFF(4 * RQ + ((k) & 3)) + (long)((k) >> 3) * rowstride + (long)(((k) >> 2) & 1) * pixstride)
template <int RQ, typename PT>
__device__ __forceinline__ void wino_epilogue16(const PT& p, float v[16], size_t o0, long rowstride, long pixstride, float bv)
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

// edge tiles: per-element bounds checks; (oy, ox) = output pixel (e = 0) of tile r = 0
template <int RQ, typename PT>
__device__ __forceinline__ void wino_epilogue16_ragged(const PT& p, const float v[16], size_t o0, long rowstride, long pixstride, float bv,
                                                       int oy, int ox)
{
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = 4 * RQ + (k & 3);
        const int y = oy + 2 * (r >> 3) + (k >> 3), x = ox + 2 * ((r & 3) + 8 * ((r >> 2) & 1)) + ((k >> 2) & 1);
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

// UP: some source is nearest-up-sampled on read (a compile-time choice keeps the DMA address code free of branches, so the
// scheduler can spread it between the MFMAs)
template <bool SINGLE, bool UP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_wino_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int NTHR = 256, NW = 4;
    f32x4* As0 = (f32x4*)smem_f;                    // [2][W_HPB]  (halo tile + padding up to a whole number of DMA rounds)
    f32x4* Bs0 = As0 + 2 * W_HPB;                   // [2][W_BSZ]
    int* s_next = (int*)(Bs0 + 2 * W_BSZ);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // in an SGPR: everything derived from it stays scalar
    const int h = lane >> 5, li = lane & 31;

    // Halo buffer layout: [channel quad q][column parity][halo row][column >> 1] 16-byte slots.  The 32 lanes of a half-wave
    // read patch element (a, b) of 16 x 2 Winograd tiles = every second column: split by parity these are CONSECUTIVE slots
    // (conflict-free ds_read_b128), and every element is a compile-time offset from one per-lane base address.
    const int abase = ((2 * (2 * wave + (li >> 4))) * (W_HPW / 2) + (li & 15) + h * W_PLANE) * 16;
    const int nsteps = p.nchunks;
    const int total_tiles = p.N * p.tiles_y * p.tiles_x * p.co_blocks;

    // DMA slot id = i*NTHR + tid (the DMA writes LDS linearly) -> (q = id / W_PLANE, parity, row, column >> 1), decoded
    // arithmetically where needed (x / 17 = x * 241 >> 12 for x < 306): a table in registers costs 10 of the 256 VGPRs left beside
    // the accumulators, one in LDS costs an lgkmcnt(0) stall per MFMA group.  Slots past the 612 used ones of a plane fetch the
    // zero page.

#define WN_DECODE(T, N_, A0_, B0_, CO0_)                                                  \
    {                                                                                     \
        int q_ = (T);                                                                     \
        CO0_ = (q_ % p.co_blocks) * 32; q_ /= p.co_blocks;                                \
        B0_ = (q_ % p.tiles_x) * W_TW; q_ /= p.tiles_x;                                   \
        A0_ = (q_ % p.tiles_y) * W_TH; N_ = q_ / p.tiles_y;                               \
    }
    // DMA of the next step, in parts that are spread over the MFMA groups of the current one: part 0 = the chunk of U,
    // parts 1..5 = two halo slots each
#define WN_DMA_U(CHUNK, CO0_, BUF)                                                                            \
    {                                                                                                         \
        const f32x4* wsrc_ = (const f32x4*)p.wpacked + (size_t)(CHUNK) * 64 * p.Cout_pad + (CO0_);            \
        f32x4* bbuf_ = Bs0 + (BUF) * W_BSZ;                                                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                                                    \
            const int pc_ = q_ * NW + wave;                       /* piece = two (xi, q) rows of 32 float4 */   \
            GLDS16(wsrc_ + (unsigned)((pc_ * 2 + h + opaque0) * p.Cout_pad + li), bbuf_ + pc_ * 64);          \
        }                                                                                                     \
    }
#define WN_DMA_HALO(PART, CHUNK, N_, A0_, B0_, BUF)                                                           \
    {                                                                                                         \
        const SrcSel s_ = SINGLE ? ssie_only_src(p) : ssie_pick_src(p, (CHUNK) * SSIE_CK);                    \
        const int vy0_ = (A0_) - 1, vx0_ = (B0_) - 1;                                                         \
        f32x4* abuf_ = As0 + (BUF) * W_HPB;                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                    \
            const int id_ = (2 * (PART) + i_) * NTHR + tid + opaque0;                                         \
            const int j_ = (id_ >= W_PLANE) + (id_ >= 2 * W_PLANE) + (id_ >= 3 * W_PLANE), r_ = id_ - j_ * W_PLANE; \
            const int par_ = r_ >= W_HALF, rr_ = r_ - par_ * W_HALF;                                          \
            const int hy_ = r_ < 2 * W_HALF ? (rr_ * 241) >> 12 : 255;                                        \
            const int hx_ = 2 * (rr_ - hy_ * (W_HPW / 2)) + par_;                                             \
            const f32x4* g_ = wino_virtual_addr<UP>(s_, (N_), vy0_ + hy_, vx0_ + hx_,                         \
                                                    p.Hv, p.Wv, (CHUNK) * SSIE_CK + 4 * j_ - s_.cbeg);        \
            GLDS16(g_, abuf_ + (2 * (PART) + i_) * NTHR + wave * 64);                                         \
        }                                                                                                     \
    }
#define WN_PREFETCH_ALL(CHUNK, N_, A0_, B0_, CO0_, BUF)                                   \
    {                                                                                     \
        WN_DMA_U(CHUNK, CO0_, BUF)                                                        \
        _Pragma("unroll") for (int pt_ = 0; pt_ < 5; ++pt_) WN_DMA_HALO(pt_, CHUNK, N_, A0_, B0_, BUF) \
    }

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int n, a0, b0, co0;
    WN_DECODE(tile, n, a0, b0, co0)
    int gstep = 0;
    {
        const int opaque0 = 0;
        WN_PREFETCH_ALL(0, n, a0, b0, co0, 0)
    }
    int fetched = 0x7fffffff;
    ST_DECL

    while (tile < total_tiles) {
        f32x16 acc[16];
#pragma unroll
        for (int x = 0; x < 16; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
        const float bv = (p.bias && co0 + li < p.Cout) ? p.bias[co0 + li] : 0.f;
        int ntile = 0x7fffffff;
        int nn = n, na0 = a0, nb0 = b0, nco0 = co0;

        for (int step = 0; step < nsteps; ++step, ++gstep) {
            const int buf = gstep & 1;
            if (tid == 0) {
                if (nsteps == 1 || !p.tile_counter) {
                    if (step == 0) *s_next = p.tile_counter ? (int)gridDim.x + atomicAdd(p.tile_counter, 1) : tile + (int)gridDim.x;
                } else if (step == 1) *s_next = fetched;
            }
            ST_ACC(5);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            ST_ACC(1);
            if (step == (nsteps > 1 ? 1 : 0)) {
                ntile = *s_next;
                if (ntile < total_tiles) WN_DECODE(ntile, nn, na0, nb0, nco0)
            }
            // operands of the next step (next chunk of this tile, or chunk 0 of the next tile)
            // (after the last step of the last tile this re-fetches chunk 0 of the same tile: unconditional, so the DMA address
            // code stays in the MFMA groups' basic blocks; the kernel drains it before it ends)
            const bool more = step + 1 < nsteps;
            const int pchunk = more ? step + 1 : 0;
            const int pn = more ? n : nn, pa0 = more ? a0 : na0, pb0 = more ? b0 : nb0, pco0 = more ? co0 : nco0;

            const char* Ab = (const char*)(As0 + buf * W_HPB);
            const f32x4* Bl = Bs0 + buf * W_BSZ + h * 32 + li;
            // an opaque zero per step keeps the (loop-invariant) slot decode inside the loop: hoisted, it is 30 more live registers
            int opaque0;
            asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
            // ---- software pipeline over the 8 groups g = (kq, i): group g multiplies the 4 transform positions xi = 4i .. 4i+3 of
            // 8 input channels.  While the 16 MFMAs of group g run, the lane transforms the patch rows of group g+1 (VALU) and
            // issues the LDS reads of group g+2 / the DMA of the next step.
            f32x4 rows[2][4][4];         // [kq][patch row a][patch column b]
            f32x4 bfr[2][4];             // U fragments, double-buffered by group parity
            f32x4 vv[2][4];              // transformed patch rows, double-buffered by group parity
#define WN_LD_ROW(KQ, A)                                                                                      \
            _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_)                                                  \
                rows[KQ][A][b_] = *(const f32x4*)(Ab + abase + (((A) * (W_HPW / 2) + (b_ >> 1)) + (b_ & 1) * W_HALF + (KQ) * 2 * W_PLANE) * 16);
#define WN_LD_B(G)                                                                                            \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                  \
                bfr[(G) & 1][j_] = Bl[((((G) & 3) * 4 + j_) * 4 + ((G) >> 2) * 2) * 32];
            // B^T d B for transform row i of k-quad KQ -> vv[G & 1]
#ifdef SSIE_X_NOXFORM
#define WN_XFORM(G) { _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) vv[(G) & 1][j_] = rows[(G) >> 2][(G) & 3][j_]; }
#else
#define WN_XFORM(G)                                                                                           \
            {                                                                                                 \
                constexpr int kq_ = (G) >> 2, i_ = (G) & 3;                                                   \
                f32x4 r_[4];                                                                                  \
                _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_)                                              \
                    r_[b_] = i_ == 0 ? rows[kq_][0][b_] - rows[kq_][2][b_] : i_ == 1 ? rows[kq_][1][b_] + rows[kq_][2][b_] \
                           : i_ == 2 ? rows[kq_][2][b_] - rows[kq_][1][b_] : rows[kq_][1][b_] - rows[kq_][3][b_]; \
                vv[(G) & 1][0] = r_[0] - r_[2]; vv[(G) & 1][1] = r_[1] + r_[2];                                \
                vv[(G) & 1][2] = r_[2] - r_[1]; vv[(G) & 1][3] = r_[1] - r_[3];                                \
            }
#endif
#ifdef SSIE_X_NOMFMA
#define WN_MFMA(G) _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) acc[((G) & 3) * 4 + j_][0] += vv[(G) & 1][j_].x + bfr[(G) & 1][j_].x;
#else
#define WN_MFMA(G)                                                                                            \
            _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_)                                                  \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                \
                constexpr int xi0_ = ((G) & 3) * 4;                                                           \
                acc[xi0_ + j_] = MFMA32(vv[(G) & 1][j_][c_], bfr[(G) & 1][j_][c_], acc[xi0_ + j_]);           \
            }
#endif
            // prologue: everything group 0 needs, and the rows of group 1
            WN_LD_ROW(0, 0) WN_LD_ROW(0, 2) WN_LD_B(0) WN_LD_ROW(0, 1)
            WN_XFORM(0)
            __builtin_amdgcn_sched_barrier(0);
            ST_ACC(4);
            // group 0
            WN_LD_B(1) WN_LD_ROW(0, 3)
#ifndef SSIE_X_NODMA
            WN_DMA_U(pchunk, pco0, buf ^ 1)
#endif
            WN_XFORM(1)
            WN_MFMA(0)
            __builtin_amdgcn_sched_barrier(0);
            // group 1
            WN_LD_B(2)
#ifndef SSIE_X_NODMA
            WN_DMA_HALO(0, pchunk, pn, pa0, pb0, buf ^ 1)
#endif
            WN_XFORM(2)
            WN_MFMA(1)
            __builtin_amdgcn_sched_barrier(0);
            // group 2
            WN_LD_B(3) WN_LD_ROW(1, 0) WN_LD_ROW(1, 2)
#ifndef SSIE_X_NODMA
            WN_DMA_HALO(1, pchunk, pn, pa0, pb0, buf ^ 1)
#endif
            WN_XFORM(3)
            WN_MFMA(2)
            __builtin_amdgcn_sched_barrier(0);
            // group 3
            WN_LD_B(4) WN_LD_ROW(1, 1)
#ifndef SSIE_X_NODMA
            WN_DMA_HALO(2, pchunk, pn, pa0, pb0, buf ^ 1)
#endif
            WN_XFORM(4)
            WN_MFMA(3)
            __builtin_amdgcn_sched_barrier(0);
            // group 4
            WN_LD_B(5) WN_LD_ROW(1, 3)
#ifndef SSIE_X_NODMA
            WN_DMA_HALO(3, pchunk, pn, pa0, pb0, buf ^ 1)
#endif
            WN_XFORM(5)
            WN_MFMA(4)
            __builtin_amdgcn_sched_barrier(0);
            // group 5
            WN_LD_B(6)
#ifndef SSIE_X_NODMA
            WN_DMA_HALO(4, pchunk, pn, pa0, pb0, buf ^ 1)
#endif
            WN_XFORM(6)
            WN_MFMA(5)
            __builtin_amdgcn_sched_barrier(0);
            // group 6
            WN_LD_B(7)
            WN_XFORM(7)
            WN_MFMA(6)
            __builtin_amdgcn_sched_barrier(0);
            // group 7
            WN_MFMA(7)
            ST_ACC(7);
#undef WN_LD_ROW
#undef WN_LD_B
#undef WN_XFORM
#undef WN_MFMA
            if (tid == 0 && step == 0 && nsteps > 1 && p.tile_counter)
                fetched = (int)gridDim.x + atomicAdd(p.tile_counter, 1);
        }

        // output transform A^T M A (lane-local: register r of all 16 accumulators belongs to the same tile and channel) + epilogue,
        // four accumulator registers (= four Winograd tiles = 16 outputs) per pass: reading all 256 accumulators at once would
        // push the loop-carried registers to scratch, and a scratch reload waits for every store still in flight
        {
            const int co = co0 + li;
            if (co < p.Cout) {
                const long rowstride = (long)p.Wout * p.out_cstride, pixstride = p.out_cstride;
                const int oy0 = a0 + 4 * wave, ox0 = b0 + 8 * h;
                const size_t o0 = ((size_t)(n * p.Hout + oy0) * p.Wout + ox0) * p.out_cstride + p.out_coff + co;
                const bool full = a0 + W_TH <= p.Hout && b0 + W_TW <= p.Wout;
#define WN_PASS(RQ)                                                                                           \
                {                                                                                             \
                    float y_[16];                                                                             \
                    _Pragma("unroll") for (int rr_ = 0; rr_ < 4; ++rr_) {                                     \
                        constexpr int r0_ = 4 * (RQ);                                                         \
                        float s0_[4], s1_[4];                                                                 \
                        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                    \
                            s0_[i_] = acc[i_ * 4 + 0][r0_ + rr_] + acc[i_ * 4 + 1][r0_ + rr_] + acc[i_ * 4 + 2][r0_ + rr_]; \
                            s1_[i_] = acc[i_ * 4 + 1][r0_ + rr_] - acc[i_ * 4 + 2][r0_ + rr_] - acc[i_ * 4 + 3][r0_ + rr_]; \
                        }                                                                                     \
                        y_[0 + rr_] = s0_[0] + s0_[1] + s0_[2]; y_[4 + rr_] = s1_[0] + s1_[1] + s1_[2];       \
                        y_[8 + rr_] = s0_[1] - s0_[2] - s0_[3]; y_[12 + rr_] = s1_[1] - s1_[2] - s1_[3];      \
                    }                                                                                         \
                    __builtin_amdgcn_sched_barrier(0);                                                        \
                    if (full) wino_epilogue16<RQ>(p, y_, o0, rowstride, pixstride, bv);                       \
                    else wino_epilogue16_ragged<RQ>(p, y_, o0, rowstride, pixstride, bv, oy0, ox0);           \
                    __builtin_amdgcn_sched_barrier(0);                                                        \
                }
#ifdef SSIE_X_NOEPI
                { float t_ = 0.f; _Pragma("unroll") for (int x_ = 0; x_ < 16; ++x_) t_ += acc[x_][0]; if (t_ == 123.456f) p.out[o0] = t_; }
#else
                WN_PASS(0) WN_PASS(1) WN_PASS(2) WN_PASS(3)
#endif
#undef WN_PASS
            }
        }
        ST_ACC(8);
        n = nn; a0 = na0; b0 = nb0; co0 = nco0; tile = ntile;
#ifdef SSIE_STAMP
        st_[6] += 1;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the last (redundant) prefetch must land before the LDS is released
    ST_ACC(5);
    ST_FLUSH;
#undef WN_PREFETCH_ALL
#undef WN_DMA_HALO
#undef WN_DMA_U
#undef WN_DECODE
}

// ---------------------------------------------------------------------------------------------------------------------------
// conv_wino8_kernel: the same algorithm with TWO waves per SIMD.  One workgroup = 8 waves = the same 16 x 32 positions x 32
// channels, but the MFMA is v_mfma_f32_16x16x4_f32: wave w owns Winograd tile row w (16 tiles = M) x both 16-channel halves (N)
// x all 16 xi = 32 accumulator tiles of 4 registers = 128 registers, so two waves fit a SIMD and each one's LDS reads, input
// transform, DMA address arithmetic and epilogue run underneath the other's MFMAs (measured on the one-wave kernel: MFMA time
// and everything-else time ADD when a SIMD has a single wave, tools/build_variants.py ablations).
//   A operand (16 tiles x 4 k): lane l = tile column l%16, k-group g = l/16 = channel quad g of the chunk; the 4 components of the
//   lane's float4 are the 4 MFMAs of a (xi, N-half).   B operand: lane = channel l%16 of the half, same quad g.
//   D (16 x 16): lane l = channel l%16, registers r = tile columns 4g + r.
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed fp32 add / subtract
// as inline asm: hipcc splits float2 additions whose lanes are consumed one by one (by the MFMAs) into two v_add_f32, and every
// VALU instruction costs MFMA time here.  The hazard recognizer does not see an asm as a VALU write, so the values pass through
// PK_FENCE (one s_nop covering the VALU-write -> MFMA-read wait states) before the first MFMA reads them.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define PK_FENCE8(a, b, c, d, e, f, g, h) asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h))

namespace {
// 16 outputs of one lane = (tile column 4g + r, r = k&3) x (pixel e = k>>2 of the 2 x 2 tile), one channel; o0 includes 8g pixels
#define W8_EOFF(k) ((long)((k) >> 3) * rowstride + (long)(2 * ((k) & 3) + (((k) >> 2) & 1)) * pixstride)
template <typename PT>
__device__ __forceinline__ void wino8_epilogue16(const PT& p, float v[16], size_t o0, long rowstride, long pixstride, float bv)
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
        for (int k = 0; k < 16; ++k) y[k] = mp[W8_EOFF(k)];
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
        for (int k = 0; k < 16; ++k) o2[W8_EOFF(k)] = v[k];
    }
    if (p.addsrc) {
        const float* ap = p.addsrc + o0;
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = ap[W8_EOFF(k)];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] += a[k];
    }
    float* ob = p.out + o0;
    if (p.accumulate) {
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = ob[W8_EOFF(k)];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] += a[k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) ob[W8_EOFF(k)] = v[k];
}

template <typename PT>
__device__ __forceinline__ void wino8_epilogue16_ragged(const PT& p, const float v[16], size_t o0, long rowstride, long pixstride, float bv,
                                                        int oy, int ox)
{
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int y = oy + (k >> 3), x = ox + 2 * (k & 3) + ((k >> 2) & 1);
        if (y >= p.Hout || x >= p.Wout) continue;
        const size_t o = o0 + W8_EOFF(k);
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

template <bool SINGLE, bool UP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wino8_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int NTHR = 512;
    f32x4* As0 = (f32x4*)smem_f;                    // [2][W_HPB]
    f32x4* Bs0 = As0 + 2 * W_HPB;                   // [2][W_BSZ]
    int* s_next = (int*)(Bs0 + 2 * W_BSZ);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, tx = lane & 15;
    // halo buffer layout as in conv_wino_kernel; this lane's tile = (row `wave`, column tx), its channel quad = g
    const int abase = ((2 * wave) * (W_HPW / 2) + tx + g * W_PLANE) * 16;
    const int nsteps = p.nchunks;
    const int total_tiles = p.N * p.tiles_y * p.tiles_x * p.co_blocks;

#define W8_DECODE(T, N_, A0_, B0_, CO0_)                                                  \
    {                                                                                     \
        int q_ = (T);                                                                     \
        CO0_ = (q_ % p.co_blocks) * 32; q_ /= p.co_blocks;                                \
        B0_ = (q_ % p.tiles_x) * W_TW; q_ /= p.tiles_x;                                   \
        A0_ = (q_ % p.tiles_y) * W_TH; N_ = q_ / p.tiles_y;                               \
    }
    // DMA of one step's operands: 4 pieces of U and 5 halo slots per lane.  VALU instructions and MFMAs SHARE a SIMD's issue slot on
    // this chip (tools/coexec_bench.hip: every v_add between two MFMAs costs its 4 cycles, with one or two waves per SIMD), so
    // the per-slot address arithmetic is kept minimal: a per-lane table in LDS holds, for each of the lane's 5 slots,
    //   halo pixel offset hy*Wv + hx (15 bits) | channel quad j << 15 | hy << 17 | hx << 22      (hy = 31: padding slot)
    // and a step only adds the scalar tile base and compares hy / hx / j against scalar ranges.  (UP variants: sources of
    // different sizes, decoded arithmetically as in conv_wino_kernel.)
    int* dma_tab = s_next + 4;                      // [W_HPB / NTHR][NTHR]
    if (!UP) {
#pragma unroll
        for (int i = 0; i < W_HPB / NTHR; ++i) {
            const int id = i * NTHR + tid;
            const int j = (id >= W_PLANE) + (id >= 2 * W_PLANE) + (id >= 3 * W_PLANE), r = id - j * W_PLANE;
            const int par = r >= W_HALF, rr = r - par * W_HALF;
            const int hy = r < 2 * W_HALF ? (rr * 241) >> 12 : 31;
            const int hx = r < 2 * W_HALF ? 2 * (rr - hy * (W_HPW / 2)) + par : 0;
            dma_tab[i * NTHR + tid] = (r < 2 * W_HALF ? hy * p.Wv + hx : 0) | (j << 15) | (hy << 17) | (hx << 22);
        }
    }
#ifdef SSIE_X_NOTAB
#define W8_NOTAB 1
#else
#define W8_NOTAB 0
#endif
#define W8_PREFETCH(CHUNK, N_, A0_, B0_, CO0_, BUF)                                                           \
    {                                                                                                         \
        const f32x4* wsrc_ = (const f32x4*)p.wpacked + (size_t)(CHUNK) * 64 * p.Cout_pad + (CO0_);            \
        f32x4* bbuf_ = Bs0 + (BUF) * W_BSZ;                                                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                    \
            const int pc_ = q_ * 8 + wave;                        /* piece = two (xi, q) rows of 32 float4 */   \
            GLDS16(wsrc_ + (unsigned)((pc_ * 2 + (lane >> 5)) * p.Cout_pad + (lane & 31)), bbuf_ + pc_ * 64); \
        }                                                                                                     \
        const SrcSel s_ = SINGLE ? ssie_only_src(p) : ssie_pick_src(p, (CHUNK) * SSIE_CK);                    \
        const int vy0_ = (A0_) - 1, vx0_ = (B0_) - 1;                                                         \
        f32x4* abuf_ = As0 + (BUF) * W_HPB;                                                                   \
        if (UP || W8_NOTAB) {                                                                                 \
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
            const int tb_ = (((N_) * p.Hv + vy0_) * p.Wv + vx0_) * s_.cstride + s_.coff + c0_;                \
            const unsigned ylo_ = max(0, -vy0_), yn_ = min(W_HPH, p.Hv - vy0_) - ylo_;                        \
            const unsigned xlo_ = max(0, -vx0_), xn_ = min(W_HPW, p.Wv - vx0_) - xlo_;                        \
            const int jn_ = (s_.C - c0_ + 3) >> 2;                                                            \
            const unsigned long long zp_ = (unsigned long long)wino_zero_page;                                \
            _Pragma("unroll") for (int i_ = 0; i_ < W_HPB / NTHR; ++i_) {                                     \
                const unsigned e_ = (unsigned)dma_tab[i_ * NTHR + tid];                                       \
                const unsigned hy_ = (e_ >> 17) & 31, hx_ = e_ >> 22, j_ = (e_ >> 15) & 3;                    \
                const bool ok_ = hy_ - ylo_ < yn_ && hx_ - xlo_ < xn_ && (int)j_ < jn_;                       \
                const int off_ = tb_ + (int)(e_ & 0x7fff) * s_.cstride + 4 * (int)j_;                         \
                const unsigned long long a_ = (unsigned long long)(s_.ptr + off_), m_ = ok_ ? ~0ull : 0ull;    \
                GLDS16((const f32x4*)((a_ & m_) | (zp_ & ~m_)), abuf_ + i_ * NTHR + wave * 64);               \
            }                                                                                                 \
        }                                                                                                     \
    }

    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int n, a0, b0, co0;
    W8_DECODE(tile, n, a0, b0, co0)
    int gstep = 0;
    W8_PREFETCH(0, n, a0, b0, co0, 0)
    int fetched = 0x7fffffff;

    while (tile < total_tiles) {
        f32x4 acc[16][2];
#pragma unroll
        for (int x = 0; x < 16; ++x)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[x][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        float bv[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) bv[c] = (p.bias && co0 + 16 * c + tx < p.Cout) ? p.bias[co0 + 16 * c + tx] : 0.f;
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
                if (ntile < total_tiles) W8_DECODE(ntile, nn, na0, nb0, nco0)
            }
            const char* Ab = (const char*)(As0 + buf * W_HPB) + abase;
            const f32x4* Bl = Bs0 + buf * W_BSZ + g * 32 + tx;
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
                    const f32x4 bf0 = Bl[(xi * 4) * 32], bf1 = Bl[(xi * 4) * 32 + 16];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[xi][0] = MFMA16(v[j][c >> 1][c & 1], bf0[c], acc[xi][0]);
                        acc[xi][1] = MFMA16(v[j][c >> 1][c & 1], bf1[c], acc[xi][1]);
                    }
                }
                // the next step's DMA is issued AFTER the first transform row's MFMAs: right behind the barrier all 8 waves would
                // do address arithmetic (VALU = no MFMA) while the matrix pipe has nothing queued yet
                if (i == 0) {
                    const bool more = step + 1 < nsteps;
                    if (more) W8_PREFETCH(step + 1, n, a0, b0, co0, buf ^ 1)
                    else if (ntile < total_tiles) W8_PREFETCH(0, nn, na0, nb0, nco0, buf ^ 1)
                }
            }
            if (tid == 0 && step == 0 && nsteps > 1 && p.tile_counter)
                fetched = (int)gridDim.x + atomicAdd(p.tile_counter, 1);
        }

        // output transform (lane-local) + epilogue: one pass of 16 outputs per 16-channel half
        {
            const long rowstride = (long)p.Wout * p.out_cstride, pixstride = p.out_cstride;
            const int oy0 = a0 + 2 * wave, ox0 = b0 + 8 * g;
            const bool full = a0 + W_TH <= p.Hout && b0 + W_TW <= p.Wout;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
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
                if (full) wino8_epilogue16(p, y, o0, rowstride, pixstride, bv[c]);
                else wino8_epilogue16_ragged(p, y, o0, rowstride, pixstride, bv[c], oy0, ox0);
            }
        }
        n = nn; a0 = na0; b0 = nb0; co0 = nco0; tile = ntile;
    }
#undef W8_PREFETCH
#undef W8_DECODE
}

template __global__ void conv_wino8_kernel<false, false>(const ConvParams);
template __global__ void conv_wino8_kernel<true, false>(const ConvParams);
template __global__ void conv_wino8_kernel<false, true>(const ConvParams);
template __global__ void conv_wino8_kernel<true, true>(const ConvParams);

template __global__ void conv_wino_kernel<false, false>(const ConvParams);
template __global__ void conv_wino_kernel<true, false>(const ConvParams);
template __global__ void conv_wino_kernel<false, true>(const ConvParams);
template __global__ void conv_wino_kernel<true, true>(const ConvParams);

int ssie_wino_waves8 = 1;      // A/B switch: 1 = conv_wino8_kernel (two waves per SIMD), 0 = conv_wino_kernel (one)
extern "C" void ssie_debug_set_wino_waves8(int v) { ssie_wino_waves8 = v; }

size_t ssie_wino_lds_bytes() { return (size_t)(2 * W_HPB + 2 * W_BSZ) * 16 + 64 + (size_t)W_HPB * 4; }

int ssie_launch_fprop_wino(const ConvParams& p, hipStream_t st)
{
    if (p.ntaps != 9 || p.si != 1 || p.so != 1 || p.py || p.px || p.min_dy != -1 || p.min_dx != -1 || p.Cout_pad % 32) return 31;
    if (p.th != W_TH || p.tw != W_TW || p.hp_h != W_HPH || p.hp_w != W_HPW || p.co_blocks != p.Cout_pad / 32) return 32;
    static unsigned seen[4] = {0, 0, 0, 0};
    ssie_allow_full_lds((const void*)conv_wino_kernel<false, false>, seen[0]);
    ssie_allow_full_lds((const void*)conv_wino_kernel<true, false>, seen[1]);
    ssie_allow_full_lds((const void*)conv_wino_kernel<false, true>, seen[2]);
    ssie_allow_full_lds((const void*)conv_wino_kernel<true, true>, seen[3]);
    const size_t tiles = (size_t)p.N * p.tiles_y * p.tiles_x * p.co_blocks;
    const dim3 grid((unsigned)(tiles < 256 ? tiles : 256));
    const size_t lds = ssie_wino_lds_bytes();
    bool up = false;
    for (int s = 0; s < p.nsrc; ++s) up = up || p.src[s].sy != 1.f || p.src[s].sx != 1.f;
    if (ssie_wino_waves8) {
        static unsigned seen8[4] = {0, 0, 0, 0};
        ssie_allow_full_lds((const void*)conv_wino8_kernel<false, false>, seen8[0]);
        ssie_allow_full_lds((const void*)conv_wino8_kernel<true, false>, seen8[1]);
        ssie_allow_full_lds((const void*)conv_wino8_kernel<false, true>, seen8[2]);
        ssie_allow_full_lds((const void*)conv_wino8_kernel<true, true>, seen8[3]);
        if (p.nsrc == 1 && !up) hipLaunchKernelGGL((conv_wino8_kernel<true, false>), grid, dim3(512), lds, st, p);
        else if (p.nsrc == 1) hipLaunchKernelGGL((conv_wino8_kernel<true, true>), grid, dim3(512), lds, st, p);
        else if (!up) hipLaunchKernelGGL((conv_wino8_kernel<false, false>), grid, dim3(512), lds, st, p);
        else hipLaunchKernelGGL((conv_wino8_kernel<false, true>), grid, dim3(512), lds, st, p);
        return hipGetLastError() == hipSuccess ? 0 : 34;
    }
    if (p.nsrc == 1 && !up) hipLaunchKernelGGL((conv_wino_kernel<true, false>), grid, dim3(256), lds, st, p);
    else if (p.nsrc == 1) hipLaunchKernelGGL((conv_wino_kernel<true, true>), grid, dim3(256), lds, st, p);
    else if (!up) hipLaunchKernelGGL((conv_wino_kernel<false, false>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv_wino_kernel<false, true>), grid, dim3(256), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 33;
}
