// Implicit-GEMM convolution kernels for gfx950 on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the hot path, every torch.nn.Conv2d / ConvTranspose2d / Linear call of the
// reference (model.py:17-23, 39-43, 93-97, 140-141) and their autograd backward (model.py:315):
//   * conv_fprop_kernel  : forward conv AND every data-gradient (a dgrad is a conv with a
//                          transposed/flipped packed weight and, for stride-2 layers, four
//                          output-parity classes = transposed convolution without zero-insertion)
//   * conv_wgrad_kernel  : weight gradient, split over pixel slices, deterministic slab reduce
//   * pack / reduce / colsum helpers
//
// GEMM view (fprop): M = output positions (8x16 spatial tile = 4 MFMA M-tiles of 2 rows x 16 cols),
// N = output channels, K = taps x input channels.  The input halo tile is staged ONCE per
// 16-channel chunk in LDS and re-read by all taps (LDS-staged 3x3 / 9x9 tiles); weights are
// staged per tap group.  K is permuted inside each chunk so that one ds_read_b128 feeds four
// consecutive MFMAs: lane (i, h) holds channels 8*kq + 4*h + {0..3}.
#include "conv_device.h"
#include <string.h>

// Diagnostic build only (-DSSIE_STAMP, tools/stamp_conv.py): per-workgroup s_memtime stamps of the fprop phases,
// written to a buffer of their own.  The shipped library never executes a stamp.
#ifdef SSIE_STAMP
__device__ unsigned long long* ssie_stamp_buf = nullptr;
extern "C" int ssie_debug_set_stamp_buffer(void* buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(ssie_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#define STAMP(k) do { if (ssie_stamp_buf && threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    ssie_stamp_buf[(size_t)blockIdx.x * 8 + (k)] = t_; } } while (0)
#define STAMP_DECL unsigned long long st_loop_ = 0, st_epi_ = 0, st_t_ = 0, st_n_ = 0;
#define STAMP_T0 do { st_t_ = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP_ACC(var) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); var += t_ - st_t_; st_t_ = t_; } while (0)
#define STAMP_FLUSH do { if (ssie_stamp_buf && threadIdx.x == 0) { ssie_stamp_buf[(size_t)blockIdx.x * 8 + 2] = st_loop_; \
    ssie_stamp_buf[(size_t)blockIdx.x * 8 + 5] = st_epi_; ssie_stamp_buf[(size_t)blockIdx.x * 8 + 6] = st_n_; } } while (0)
#else
#define STAMP(k)
#define STAMP_DECL
#define STAMP_T0
#define STAMP_ACC(var)
#define STAMP_FLUSH
#endif

// ---------------------------------------------------------------------------------------------
// fprop / dgrad
// ---------------------------------------------------------------------------------------------
// NT: output-channel tile = 32*NT.  NA: max float4 A-loads per thread per chunk (halo tile size / 256).
// Staging is software-pipelined through registers: the global loads of step s+1 (next tap group / next
// 16-channel chunk) are issued right after the barrier that publishes step s and land while the MFMAs of
// step s run; only the short register -> LDS commit sits between two barriers.
// TH: output tile rows (8 or 16; the tile is TH x 16 positions = TH/2 MFMA M-tiles).  The 16-row tile halves the
// weight staging, the barriers and the tile boundaries per MFMA and is used for the stride-1 3x3 / 1x1 layers.
template <int NT, int NA, int TH, int EPI = 0, bool RAG = true>      // EPI / RAG: epilogue shape and "some tile sticks out" (ssie_epi_shape)
__global__ __launch_bounds__(256, ((NA <= 3 && TH == 8) ? 3 : 2)) void conv_fprop_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int BN = 32 * NT;
    constexpr int MT = (NT == 1) ? TH / 8 : TH / 4;
    constexpr int NB = (SSIE_TG * 4 * BN + 255) / 256;
    const int HP = p.hp_h * p.hp_w;
    f32x4* As = (f32x4*)smem_f;                // [HP][4] float4 (16-B slot XOR-swizzled)
    f32x4* Bs = As + HP * 4;                   // [TG*4][BN] float4
    int* tapoff = (int*)(Bs + SSIE_TG * 4 * BN);
    int* s_next = tapoff + SSIE_MAX_TAPS;      // dynamic tile scheduler hand-off slot

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, li = lane & 31;
    const int wn = (NT == 1) ? 0 : (wave & 1);
    const int wm = (NT == 1) ? wave : (wave >> 1);

    STAMP(0);
#ifdef SSIE_STAMP
    if (ssie_stamp_buf && threadIdx.x == 0) ssie_stamp_buf[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#endif
    for (int t = tid; t < p.ntaps; t += 256)
        tapoff[t] = ((int)p.tap_dy[t] - p.min_dy) * p.hp_w + ((int)p.tap_dx[t] - p.min_dx);

    int pixbase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        int mt = wm * MT + m;
        pixbase[m] = ((2 * mt + (li >> 4)) * p.si) * p.hp_w + (li & 15) * p.si;
    }
    const int ngroups = (p.ntaps + SSIE_TG - 1) / SSIE_TG;
    const int nsteps = p.nchunks * ngroups;
    const int HP4 = HP * 4;
    const int total_tiles = p.N * p.tiles_y * p.tiles_x * p.co_blocks;

    // tile index -> (image n, tile row/col, output-channel block); co block fastest so the workgroups that share
    // an input halo tile run next to each other
#define SSIE_DECODE(T, N_, A0_, B0_, CO0_)                                                \
    {                                                                                     \
        int q_ = (T);                                                                     \
        CO0_ = (q_ % p.co_blocks) * BN; q_ /= p.co_blocks;                                \
        B0_ = (q_ % p.tiles_x) * SSIE_TW; q_ /= p.tiles_x;                                \
        A0_ = (q_ % p.tiles_y) * TH; N_ = q_ / p.tiles_y;                            \
    }

    // halo pixel of each of this thread's A loads (tile-invariant: hoisted out of the prefetch)
    int ahy[NA], ahx[NA], aj[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int id = min(tid + i * 256, HP4 - 1);
        const int pix = id >> 2;
        ahy[i] = pix / p.hp_w; ahx[i] = pix - ahy[i] * p.hp_w; aj[i] = id & 3;
    }
    f32x4 pa[NA], pb[NB];
#define SSIE_PREFETCH(CHUNK, G, N_, A0_, B0_, CO0_)                                                          \
    {                                                                                                        \
        if ((G) == 0) {                                                                                      \
            const SrcSel s_ = ssie_pick_src(p, (CHUNK) * SSIE_CK);                                           \
            const int vy0_ = (A0_) * p.si + p.min_dy, vx0_ = (B0_) * p.si + p.min_dx;                        \
            _Pragma("unroll") for (int i_ = 0; i_ < NA; ++i_)                                                \
                pa[i_] = ssie_load_virtual(s_, (N_), vy0_ + ahy[i_], vx0_ + ahx[i_], p.Hv, p.Wv,             \
                                           (CHUNK) * SSIE_CK + 4 * aj[i_] - s_.cbeg);                        \
        }                                                                                                    \
        /* weights: always TG tap-rows (the packed buffer is padded by one tap group, so the over-read past a  \
           short group stays inside the allocation; only the valid rows are committed to LDS) */             \
        const f32x4* wsrc_ = (const f32x4*)p.wpacked + ((size_t)((CHUNK) * p.ntaps + (G) * SSIE_TG) * 4) * p.Cout_pad  \
                             + (CO0_) + (size_t)(tid / BN) * p.Cout_pad + (tid % BN);                        \
        _Pragma("unroll") for (int i_ = 0; i_ < NB; ++i_) pb[i_] = wsrc_[(size_t)i_ * (256 / BN) * p.Cout_pad]; \
    }

    // Persistent workgroups with a dynamic tile queue: the first tile is blockIdx.x, every further one is drawn
    // from a device counter (zeroed by the host before the launch), so workgroups that get more of the shared
    // MFMA pipe (older waves win arbitration) simply take more tiles and all finish together.  The counter value
    // is fetched by one lane a whole step ahead of its use and handed to the workgroup through LDS between the two
    // barriers every step has anyway.  The first loads of the NEXT tile are issued before the epilogue of the current
    // one, so a tile boundary costs one register->LDS commit instead of a cold global-memory round trip.
    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
    int n, a0, b0, co0;
    SSIE_DECODE(tile, n, a0, b0, co0)
    SSIE_PREFETCH(0, 0, n, a0, b0, co0)
    bool first = true;
    const int pub_step = nsteps > 1 ? 1 : 0;
    int fetched = 0x7fffffff;
    STAMP_DECL
    while (tile < total_tiles) {
        STAMP_T0;
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        int ntile = 0x7fffffff;
        int nn = n, na0 = a0, nb0 = b0, nco0 = co0;

        int chunk = 0, g = 0;
        for (int step = 0; step < nsteps; ++step) {
            const int t0 = g * SSIE_TG;
            const int tg = min(SSIE_TG, p.ntaps - t0);
#ifndef ABL_NOBARRIER
            __syncthreads();     // everyone finished reading the previous B (and A when g == 0)
#endif
#ifndef ABL_NOCOMMIT
            if (g == 0) {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int id = tid + i * 256;
                    if (id < HP4) { const int pix = id >> 2, j = id & 3; As[pix * 4 + (j ^ ssie_swz(pix))] = pa[i]; }
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int id = tid + i * 256;
                if (id < tg * 4 * BN) Bs[id] = pb[i];
            }
#endif
            if (tid == 0) {
                if (step == 0 && (nsteps == 1 || !p.tile_counter))
                    fetched = p.tile_counter ? (int)gridDim.x + atomicAdd(p.tile_counter, 1) : tile + (int)gridDim.x;
                if (step == pub_step) *s_next = fetched;
            }
#ifndef ABL_NOBARRIER
            __syncthreads();
#endif
            if (first && step == 0) { STAMP(1); first = false; }
            if (tid == 0 && step == 0 && nsteps > 1 && p.tile_counter)
                fetched = (int)gridDim.x + atomicAdd(p.tile_counter, 1);      // consumed one step later
            if (step == pub_step) {
                ntile = *s_next;
                if (ntile < total_tiles) SSIE_DECODE(ntile, nn, na0, nb0, nco0)
            }
            int nchunk = chunk, ng = g + 1;
            if (ng == ngroups) { ng = 0; ++nchunk; }
#ifndef ABL_NOPREFETCH
            if (step + 1 < nsteps) SSIE_PREFETCH(nchunk, ng, n, a0, b0, co0)
            else if (ntile < total_tiles) SSIE_PREFETCH(0, 0, nn, na0, nb0, nco0)
#endif
            // Software-pipelined fragment reads: two register sets (X for kq = 0, Y for kq = 1).  The ds_reads of the
            // next half-tap are issued BEFORE the MFMAs of the current one, so one wave alone keeps its SIMD's MFMA
            // pipe busy across the ~100-cycle LDS latency (counted lgkmcnt waits instead of a drain per 4 MFMAs).
#define SSIE_LDFRAG(BF, AF, TL, KQ, OFF)                                                              \
            {                                                                                             \
                BF = Bs[((TL) * 4 + (KQ) * 2 + h) * BN + wn * 32 + li];                                   \
                _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_) {                                       \
                    const int hp_ = pixbase[m_] + (OFF);                                                  \
                    AF[m_] = As[hp_ * 4 + (((KQ) * 2 + h) ^ ssie_swz(hp_))];                              \
                }                                                                                         \
            }
#define SSIE_MFMA4(BF, AF)                                                                            \
            _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_) {                                           \
                acc[m_] = MFMA32(AF[m_].x, BF.x, acc[m_]); acc[m_] = MFMA32(AF[m_].y, BF.y, acc[m_]);     \
                acc[m_] = MFMA32(AF[m_].z, BF.z, acc[m_]); acc[m_] = MFMA32(AF[m_].w, BF.w, acc[m_]);     \
            }
            {
                f32x4 bX, bY, aX[MT], aY[MT];
                int off = tapoff[t0];
                SSIE_LDFRAG(bX, aX, 0, 0, off)
                for (int tl = 0; tl < tg; ++tl) {
                    const int off_next = tapoff[t0 + min(tl + 1, tg - 1)];
                    SSIE_LDFRAG(bY, aY, tl, 1, off)
                    SSIE_MFMA4(bX, aX)
                    if (tl + 1 < tg) SSIE_LDFRAG(bX, aX, tl + 1, 0, off_next)
                    SSIE_MFMA4(bY, aY)
                    off = off_next;
                }
            }
#undef SSIE_LDFRAG
#undef SSIE_MFMA4
            chunk = nchunk; g = ng;
        }

        STAMP_ACC(st_loop_);
        // epilogue: C/D layout col = lane&31 (channel), row i = (r&3) + 8*(r>>2) + 4*(lane>>5) (position inside the
        // 2x16 M-tile: tile row i>>4 = r>>3, tile column i&15 = (r&3) + 8*((r>>2)&1) + 4h).  All per-element address
        // arithmetic is therefore a compile-time multiple of two run-time strides; the common case (interior tile,
        // plain bias+activation store) is a straight run of 16 stores per M-tile.
        const int co = co0 + wn * 32 + li;
        if (co < p.Cout) {
            const float bv = p.bias ? p.bias[co] : 0.f;
            const long rowstride = (long)p.so * p.Wout * p.out_cstride;
            const long pixstride = (long)p.so * p.out_cstride;
            const bool full = !RAG || (a0 + TH <= p.Ho && b0 + SSIE_TW <= p.Wo &&
                                       (a0 + TH - 1) * p.so + p.py < p.Hout && (b0 + SSIE_TW - 1) * p.so + p.px < p.Wout);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int mt = wm * MT + m;
                const int arow = a0 + 2 * mt, bcol = b0 + 4 * h;
                const size_t o0 = ((size_t)(n * p.Hout + arow * p.so + p.py) * p.Wout + bcol * p.so + p.px) * p.out_cstride + p.out_coff + co;
                if (full) ssie_epilogue_full<EPI>(p, acc[m], o0, rowstride, pixstride, bv);
                else ssie_epilogue_ragged(p, acc[m], o0, rowstride, pixstride, bv, arow, bcol);
            }
        }
        n = nn; a0 = na0; b0 = nb0; co0 = nco0; tile = ntile;
        STAMP_ACC(st_epi_);
#ifdef SSIE_STAMP
        st_n_ += 1;
#endif
    }
    STAMP_FLUSH;
#ifdef SSIE_STAMP
    if (ssie_stamp_buf && threadIdx.x == 0) ssie_stamp_buf[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime() - ssie_stamp_buf[(size_t)blockIdx.x * 8 + 7];
#endif
#undef SSIE_PREFETCH
#undef SSIE_DECODE
    STAMP(3);
}

#define INST_FPROP(NT, NA, TH) template __global__ void conv_fprop_kernel<NT, NA, TH>(const ConvParams);
INST_FPROP(1, 3, 8) INST_FPROP(1, 6, 8) INST_FPROP(1, 9, 8) INST_FPROP(2, 3, 8) INST_FPROP(2, 6, 8) INST_FPROP(2, 9, 8)
INST_FPROP(1, 6, 16) INST_FPROP(2, 6, 16)
// whole-tile / plain-epilogue forms of the 8 x 16-tile kernels (the small launches: batch 1-2, the pyramid's low levels)
#define INST_FPROP_S(NT, NA) template __global__ void conv_fprop_kernel<NT, NA, 8, 0, false>(const ConvParams); \
                             template __global__ void conv_fprop_kernel<NT, NA, 8, 1, false>(const ConvParams); \
                             template __global__ void conv_fprop_kernel<NT, NA, 8, 2, false>(const ConvParams);
INST_FPROP_S(1, 3) INST_FPROP_S(1, 6) INST_FPROP_S(1, 9) INST_FPROP_S(2, 3) INST_FPROP_S(2, 6) INST_FPROP_S(2, 9)

__device__ f32x4 ssie_zero_page_w[4];     // zero-initialised: source of the padding slots of the wgrad DMA staging
#define GLDS16W(gptr, lptr)                                                                            \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// ---------------------------------------------------------------------------------------------
// wgrad: dW[tap][ci][co] = sum_positions X[pos*si + tap][ci] * G[pos][co]
// GEMM view: M = ci (A operand), N = co (B operand), K = positions (2 per MFMA).
// Each workgroup owns a (ci block, co block, tap group) and a slice of the position tiles; it keeps
// up to 9 taps x 32x32 accumulators per wave in registers and writes ONE partial slab at the end.
// ---------------------------------------------------------------------------------------------
// SW selects the K loop: 0 = generic (any tap list / stride; K pairs = horizontal neighbours, one LDS read per MFMA);
// 1 / 2 = SLIDING WINDOW for stride-1 groups of 9 taps that form a 3 x 3 block (SW = 1) or one row of a 9 x 9 kernel
// (SW = 2): the K pair is a VERTICAL pair of positions (lane half h takes row 2*rp + h) and the loop walks along the row,
// so the x value of halo column c serves tap dx at position c - dx: a circular register window takes ONE new LDS read
// per tap row and step instead of one per tap - 4 (3 x 3) or 2 (1 x 9) reads per 9 MFMAs instead of 10, with
// compile-time LDS offsets (no address arithmetic in the loop).  The fp32 MFMA kernels are limited by energy per FLOP
// (DESIGN.md 3.1), and LDS reads were this kernel's largest non-MFMA term.
template <int CIB, int COB, int NU, int SW>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int MI = CIB / 32, NI = COB / 32, NPAIR = MI * NI, WSPLIT = 4 / NPAIR;
    const int PT = p.th * SSIE_TW;
    float* Xs = smem_f;                           // [HP][CIB]   (HP = rows needed by THIS tap group x hp_w)
    float* Gs = Xs + p.hp_h * p.hp_w * CIB;       // [PT][COB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, li = lane & 31;
    // every wave owns one (ci-tile, co-tile) pair and ALL taps of the group; waves that share a pair split the
    // positions (K) between them and write separate partial slabs - no tap imbalance, no idle wave
    const int pair = wave % NPAIR, wsub = wave / NPAIR;
    const int mi = pair / NI, ni = pair % NI;
    const int slice = blockIdx.x;
    const int cib = blockIdx.y / p.co_blocks, cob = blockIdx.y % p.co_blocks;
    // SW = 4 (9 x 9, two wave pairs per workgroup): the workgroup stages ONE gradient tile for TWO kernel rows (tap groups
    // 2z and 2z + 1) and wave pair wsub takes row 2z + wsub over all positions, instead of both pairs splitting the positions
    // of one row - the nine rows then re-read the gradient tile 5 times instead of 9 (1.5 GB -> 0.85 GB of L2-miss traffic per
    // launch against 0.4 GB algorithmic) and the in-LDS reduction of the two K halves disappears.
    const int ngroups_all = (p.ntaps + SSIE_TG - 1) / SSIE_TG;
    const int grp0 = SW == 4 ? 2 * (int)blockIdx.z : (int)blockIdx.z;
    const int ngrp_here = SW == 4 ? min(2, ngroups_all - grp0) : 1;
    const int mygrp = grp0 + (SW == 4 ? wsub : 0);
    const bool grp_valid = mygrp < ngroups_all;
    const int t0 = (grp_valid ? mygrp : grp0) * SSIE_TG;
    const int tg = min(SSIE_TG, p.ntaps - t0);
    const int ci0 = cib * CIB, co0 = cob * COB;

    // halo rows actually touched by this workgroup's tap group(s) (one kernel row of the 9x9 => 8 rows instead of 16)
    int gmin_dy = 127, gmax_dy = -127;
    for (int tl = grp0 * SSIE_TG; tl < min((grp0 + ngrp_here) * SSIE_TG, p.ntaps); ++tl) { const int dy = p.tap_dy[tl]; gmin_dy = min(gmin_dy, dy); gmax_dy = max(gmax_dy, dy); }
    const int rows = (p.th - 1) * p.si + (gmax_dy - gmin_dy) + 1;
    const int HP = rows * p.hp_w;

    int toff[NU];
    bool tval[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        tval[u] = u < tg && grp_valid;
        const int t = t0 + (u < tg ? u : 0);
        toff[u] = (((int)p.tap_dy[t] - gmin_dy) * p.hp_w + ((int)p.tap_dx[t] - p.min_dx)) * CIB + mi * 32 + li;
    }
    f32x16 acc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

    SrcSel s;
    s.ptr = p.src.ptr; s.C = p.src.C; s.cstride = p.src.cstride; s.coff = p.src.coff;
    s.Hs = p.src.Hs; s.Ws = p.src.Ws; s.sy = p.src.sy; s.sx = p.src.sx; s.cbeg = 0;

    const int tps = (p.tiles_total + p.nslices - 1) / p.nslices;
    const int tile_beg = slice * tps, tile_end = min(tile_beg + tps, p.tiles_total);
    constexpr int CI4 = CIB / 4, CO4 = COB / 4;
    // fused bias gradient (column sums of G): done once per co-block by the (ci block 0, tap group 0) workgroups
    const bool do_bias = p.bias_slabs != nullptr && cib == 0 && blockIdx.z == 0;      // (SW = 4: z = 0 covers rows 0 and 1; still one workgroup per co-block)
    constexpr int BROWS = 256 / COB;
    const int bcol = tid % COB, brow = tid / COB;
    float bsum = 0.f;

    // tile-invariant part of the halo staging: (row, column, channel quad) of this thread's LDS slots
    // ceil(max halo pixels of a sliding-window group * CI4 / 256): 10 x 18 or 8 x 24 at stride 1, 9 x 33 at stride 2
    constexpr int MAXX = (SW == 3 ? 19 : SW == 4 ? 14 : 12) * CIB / 64 + (SW == 3 && CIB == 32 ? 1 : 0);
    constexpr int PT_MAX = 8 * SSIE_TW;
    const int nxs = HP * CI4;
    const bool up = s.sy != 1.f || s.sx != 1.f;
    int xd[MAXX];
    if constexpr (SW != 0) {
#pragma unroll
        for (int it = 0; it < MAXX; ++it) {
            const int id = min(it * 256 + tid, nxs - 1);
            const int pix = id / CI4, hy = pix / p.hp_w;
            xd[it] = (hy << 12) | ((pix - hy * p.hp_w) << 4) | (id % CI4);
        }
    }

#ifdef SSIE_STAMP
    unsigned long long ws_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ws_t_ = __builtin_amdgcn_s_memtime(); ws_[0] = ws_t_; ws_[4] = __builtin_amdgcn_s_memrealtime();
#define WST(k) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); ws_[k] += t_ - ws_t_; ws_t_ = t_; } while (0)
#else
#define WST(k)
#endif
    for (int tile = tile_beg; tile < tile_end; ++tile) {
        int tt = tile;
        const int tx = tt % p.tiles_x; tt /= p.tiles_x;
        const int ty = tt % p.tiles_y;
        const int n = tt / p.tiles_y;
        const int a0 = ty * p.th, b0 = tx * SSIE_TW;
        const int vy0 = a0 * p.si + gmin_dy, vx0 = b0 * p.si + p.min_dx;
        __syncthreads();
        WST(2);
        if constexpr (SW != 0) {
        // staging by global->LDS DMA (16 bytes per lane, LDS written linearly: slot id = it*256 + tid): no staging VGPRs,
        // no ds_write pass and ~10x fewer instructions than the register path - the staging wave shares its SIMD with the
        // other workgroup's MFMA stream and was starved of issue slots (43 % of a wave's time, tools/stamp_wgrad.py).
        // Out-of-image / out-of-channel slots read a zero page; padding channels inside a valid quad are whatever the
        // tensor holds there - they only reach output columns >= Cout, which the slab reduction never reads.
#pragma unroll
        for (int it = 0; it < MAXX; ++it) {
            if (it * 256 >= nxs) break;                                   // wave-uniform
            const int id = it * 256 + tid;
            const int hy = xd[it] >> 12, hx = (xd[it] >> 4) & 255, j = xd[it] & 15;
            const int vy = vy0 + hy, vx = vx0 + hx, c = ci0 + 4 * j;
            const bool ok = (unsigned)vy < (unsigned)p.Hv && (unsigned)vx < (unsigned)p.Wv && c < s.C;
            int y = vy, x = vx;
            if (up) {
                const int cy = min(max(vy, 0), p.Hv - 1), cx = min(max(vx, 0), p.Wv - 1);
                y = min((int)floorf((float)cy * s.sy), s.Hs - 1);
                x = min((int)floorf((float)cx * s.sx), s.Ws - 1);
            }
            const unsigned off = (unsigned)((n * s.Hs + y) * s.Ws + x) * (unsigned)s.cstride + (unsigned)(s.coff + c);
            const f32x4* gp = ok ? (const f32x4*)(s.ptr + off) : (const f32x4*)ssie_zero_page_w;
            if (id < nxs) GLDS16W(gp, (f32x4*)Xs + it * 256 + wave * 64);
        }
#pragma unroll
        for (int it = 0; it < PT_MAX * CO4 / 256; ++it) {
            if (it * 256 >= PT * CO4) break;
            const int id = it * 256 + tid;
            const int pix = id / CO4, j = id % CO4;
            const int a = a0 + pix / SSIE_TW, b = b0 + pix % SSIE_TW, c = co0 + 4 * j;
            const bool ok = a < p.Ho && b < p.Wo && c < ((p.Cout + 3) & ~3);
            const unsigned off = (unsigned)((n * p.Ho + a) * p.Wo + b) * (unsigned)p.g_cstride + (unsigned)(p.g_coff + c);
            const f32x4* gp = ok ? (const f32x4*)(p.g + off) : (const f32x4*)ssie_zero_page_w;
            GLDS16W(gp, (f32x4*)Gs + it * 256 + wave * 64);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
        // generic layers (stride 2, 1 x 1, parity classes): register staging (measured faster there than the DMA path)
        // staging in batches of UB independent 16-byte loads per thread so the global latency is paid once per
        // batch, not once per element
        constexpr int UB = 6;
        for (int base = 0; base < HP * CI4; base += 256 * UB) {
            f32x4 r[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int id = min(base + u * 256 + tid, HP * CI4 - 1);
                const int pix = id / CI4, j = id % CI4;
                const int hy = pix / p.hp_w, hx = pix - hy * p.hp_w;
                r[u] = ssie_load_virtual(s, n, vy0 + hy, vx0 + hx, p.Hv, p.Wv, ci0 + 4 * j);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int id = base + u * 256 + tid;
                if (id < HP * CI4) *(f32x4*)(Xs + (id / CI4) * CIB + 4 * (id % CI4)) = r[u];
            }
        }
        for (int base = 0; base < PT * CO4; base += 256 * UB) {
            f32x4 r[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                // branch-free: clamp the address into the tensor (g buffers are padded to a multiple of 4 channels,
                // padding is zero), zero what lies outside the tile / channel range afterwards
                const int id = min(base + u * 256 + tid, PT * CO4 - 1);
                const int pix = id / CO4, j = id % CO4;
                const int a = a0 + pix / SSIE_TW, b = b0 + pix % SSIE_TW;
                const int c = co0 + 4 * j;
                const bool ok = a < p.Ho && b < p.Wo && c < p.Cout;
                const int ca = min(a, p.Ho - 1), cb = min(b, p.Wo - 1), cc = min(c, ((p.Cout + 3) & ~3) - 4);
                f32x4 v = *(const f32x4*)(p.g + ((size_t)(n * p.Ho + ca) * p.Wo + cb) * p.g_cstride + p.g_coff + cc);
                if (c + 1 >= p.Cout) v.y = 0.f;
                if (c + 2 >= p.Cout) v.z = 0.f;
                if (c + 3 >= p.Cout) v.w = 0.f;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                r[u] = ok ? v : z;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int id = base + u * 256 + tid;
                if (id < PT * CO4) *(f32x4*)(Gs + (id / CO4) * COB + 4 * (id % CO4)) = r[u];
            }
        }
        }
        WST(1);
        __syncthreads();
        WST(2);
        if (do_bias)
            for (int px = brow; px < PT; px += BROWS) bsum += Gs[px * COB + bcol];
        // MFMA K loop over position pairs (this wave's share: kp = wsub, wsub + WSPLIT, ...)
        // (a register-ping-pong software pipeline of this loop cuts its cycles by 30 % in isolation but the clock and the
        //  co-resident workgroup's staging give all of it back: slower by 2-4 % in wall time, tools/stamp_wgrad.py)
        if constexpr (SW != 0) {
            // SW = 1: 3 x 3 stride 1, 2: one row of a 9 x 9 (stride 1), 3: 3 x 3 stride 2 (two new halo columns per step)
            constexpr int NDY = (SW == 2 || SW == 4) ? 1 : 3, NDX = (SW == 2 || SW == 4) ? 9 : 3, SI = SW == 3 ? 2 : 1;
            static_assert(NU == 9, "sliding-window loop: groups of 9 taps");
            constexpr int RSTEP = SW == 4 ? 1 : WSPLIT;                      // SW = 4: the wave pair owns its kernel row's whole tile
            const int krow = SW == 4 ? mygrp - grp0 : 0;                     // this wave's kernel row inside the staged halo rows
            for (int rp = (SW == 4 ? 0 : wsub); rp < (SW == 4 && !grp_valid ? 0 : p.th / 2); rp += RSTEP) {
                const int row = 2 * rp + h;
                const float* xr = Xs + (row * SI + krow) * p.hp_w * CIB + mi * 32 + li; // halo (row*SI + d, column c) = xr[(d*hp_w + c)*CIB]
                const float* gr = Gs + row * SSIE_TW * COB + ni * 32 + li;
                const int rstr = p.hp_w * CIB;
                float win[NDY][NDX];                                            // win[d][c % NDX] = halo column c of tap row d
#pragma unroll
                for (int d = 0; d < NDY; ++d)
#pragma unroll
                    for (int c = 0; c < NDX - SI; ++c) win[d][c] = xr[d * rstr + c * CIB];
#pragma unroll
                for (int x = 0; x < SSIE_TW; ++x) {
#pragma unroll
                    for (int d = 0; d < NDY; ++d)
#pragma unroll
                        for (int e = 0; e < SI; ++e) {
                            const int c = x * SI + NDX - SI + e;                // the SI new halo columns of this step
                            win[d][c % NDX] = xr[d * rstr + c * CIB];
                        }
                    const float b = gr[x * COB];
#pragma unroll
                    for (int d = 0; d < NDY; ++d)
#pragma unroll
                        for (int c = 0; c < NDX; ++c) acc[d * NDX + c] = MFMA32(win[d][(x * SI + c) % NDX], b, acc[d * NDX + c]);
                }
            }
        } else {
        const int npair = PT / 2;
        for (int kp = wsub; kp < npair; kp += WSPLIT) {
            const int pix = 2 * kp + h;
            const int xbase = ((pix / SSIE_TW) * p.si * p.hp_w + (pix % SSIE_TW) * p.si) * CIB;
            const float b = Gs[pix * COB + ni * 32 + li];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const float a = Xs[xbase + toff[u]];
                acc[u] = MFMA32(a, b, acc[u]);
            }
        }
        }
        WST(7);
#ifdef SSIE_STAMP
        ws_[6] += 1;
#endif
    }

    if (do_bias) {
        __syncthreads();
        Gs[brow * COB + bcol] = bsum;
        __syncthreads();
        if (tid < COB) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < BROWS; ++r) t += Gs[r * COB + tid];
            p.bias_slabs[(size_t)slice * p.co_pad + co0 + tid] = t;
        }
    }
    // waves that shared a tile pair add their partial accumulators through LDS (fixed order => deterministic), so the
    // workgroup writes ONE partial slab [slice][tap][ci_pad][co_pad]; row (M) = ci, col (N) = co
    if (WSPLIT > 1 && SW != 4) {
        float* red = smem_f;                      // NPAIR x NU x 16 x 64 floats <= 36.9 KB, inside the staging area
        for (int w = 1; w < WSPLIT; ++w) {
            __syncthreads();
            if (wsub == w) {
#pragma unroll
                for (int u = 0; u < NU; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((pair * NU + u) * 16 + r) * 64 + lane] = acc[u][r];
            }
            __syncthreads();
            if (wsub == 0) {
#pragma unroll
                for (int u = 0; u < NU; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[u][r] += red[((pair * NU + u) * 16 + r) * 64 + lane];
            }
        }
        if (wsub != 0) return;
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        if (!tval[u]) continue;
        const int t = t0 + u;
        float* dst = p.slabs + (((size_t)slice * p.ntaps + t) * p.ci_pad + ci0 + mi * 32) * p.co_pad + co0 + ni * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            dst[(size_t)i * p.co_pad] = acc[u][r];
        }
    }
#ifdef SSIE_STAMP
    WST(5);
    ws_[3] = __builtin_amdgcn_s_memtime(); ws_[4] = __builtin_amdgcn_s_memrealtime() - ws_[4];
    if (ssie_stamp_buf && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        for (int k_ = 0; k_ < 8; ++k_) ssie_stamp_buf[(size_t)blockIdx.x * 8 + k_] = ws_[k_];
#endif
#undef WST
}

#define INST_WGRAD(CI, CO, NU) template __global__ void conv_wgrad_kernel<CI, CO, NU, 0>(const WgradParams);
INST_WGRAD(64, 64, 9) INST_WGRAD(64, 64, 1)
INST_WGRAD(32, 64, 9) INST_WGRAD(32, 64, 1)
INST_WGRAD(64, 32, 9) INST_WGRAD(64, 32, 1)
INST_WGRAD(32, 32, 9) INST_WGRAD(32, 32, 1)
#define INST_WGRAD_SW(CI, CO) template __global__ void conv_wgrad_kernel<CI, CO, 9, 1>(const WgradParams); \
                              template __global__ void conv_wgrad_kernel<CI, CO, 9, 2>(const WgradParams);
INST_WGRAD_SW(64, 64) INST_WGRAD_SW(32, 64) INST_WGRAD_SW(64, 32) INST_WGRAD_SW(32, 32)
template __global__ void conv_wgrad_kernel<64, 64, 9, 3>(const WgradParams);
template __global__ void conv_wgrad_kernel<32, 64, 9, 4>(const WgradParams);

// dst[co*s_co + ci*s_ci + t*s_t] (+)= sum_slices slab[slice][t][ci][co]; the trailing rows are the fused bias gradient
// db[co] (+)= sum_slices bias_slab[slice][co].  Each thread owns 4 consecutive co (one 16-byte load per slice) of a slice GROUP:
// a block = OQ output quads x SG slice groups (SG = 256 / OQ), fixed summation order (deterministic, identical on every rank).
// OQ = 16 / SG = 16 for the 256-slice launches: with 64 / 4 a 64 x 64 x 9 layer was 144 workgroups whose threads each walked 64
// slabs that are 147 KB apart (eight batches of eight loads in flight: latency-bound, 1.9 TB/s); 576 workgroups of two batches
// per thread run the same bytes in half the time.  (The slice reduction is NOT hidden by the side stream any more: SSIE_OVERLAP=0
// and 1 time the same since the persistent convolution kernels leave it no CU to overlap on - it is paid in full.)
template <int OQ>
__device__ __forceinline__ void wgrad_reduce_body(const ReduceDesc& d, const long blk, f32x4* red)
{
    // co_group > 0: the output channels are co_group-sized blocks of DIFFERENT parameter tensors (q | k | v): block j's weights sit
    // j * w_extra floats further than co * s_co says, its bias j * b_extra further than db + co
    constexpr int SG = 256 / OQ;
    const int lo = threadIdx.x % OQ, sg = threadIdx.x / OQ;
    const int nslices = d.nslices, Cin = d.Cin, Cout = d.Cout, co_pad = d.co_pad;
    const int cq = (Cout + 3) / 4;                                 // channel quads per (tap, ci) row
    const long idx = blk * OQ + lo;
    const long total = (long)d.ntaps * Cin * cq;
    const long total_b = total + (d.bias_slabs ? cq : 0);
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    int co = 0, ci = 0, t = 0;
    bool is_w = false, is_b = false;
    if (idx < total) {
        is_w = true;
        co = (int)(idx % cq) * 4; ci = (int)((idx / cq) % Cin); t = (int)(idx / ((long)cq * Cin));
        const size_t slab_sz = (size_t)d.ntaps * d.ci_pad * co_pad;
        const float* sp = d.slabs + ((size_t)t * d.ci_pad + ci) * co_pad + co;
        // 8 independent loads in flight per thread (the slabs are 100+ KB apart: latency-, not bandwidth-bound otherwise)
        int s = sg;
        for (; s + 7 * SG < nslices; s += 8 * SG) {
            f32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *(const f32x4*)(sp + (size_t)(s + SG * q) * slab_sz);
            sum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; s < nslices; s += SG) sum += *(const f32x4*)(sp + (size_t)s * slab_sz);
    } else if (idx < total_b) {
        is_b = true;
        co = (int)(idx - total) * 4;
        for (int s = sg; s < nslices; s += SG) sum += *(const f32x4*)(d.bias_slabs + (size_t)s * co_pad + co);
    }
    red[sg * OQ + lo] = sum;
    __syncthreads();
    if (sg == 0 && (is_w || is_b)) {
        f32x4 tot = red[lo];
#pragma unroll
        for (int g = 1; g < SG; ++g) tot += red[g * OQ + lo];       // fixed order
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (co + k >= Cout) break;
            const long blkc = d.co_group ? (co + k) / d.co_group : 0;
            float* o = is_w ? d.dst + (co + k) * d.s_co + ci * d.s_ci + t * d.s_t + blkc * d.w_extra : d.db + co + k + blkc * d.b_extra;
            *o = (is_w ? d.accumulate : d.accumulate_bias) ? (*o + tot[k]) : tot[k];
        }
    }
}

template <int OQ>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const ReduceDesc d)
{
    __shared__ f32x4 red[256];
    wgrad_reduce_body<OQ>(d, blockIdx.x, red);
}
template __global__ void wgrad_reduce_kernel<64>(const ReduceDesc);
template __global__ void wgrad_reduce_kernel<16>(const ReduceDesc);

// every layer's reduction of one backward pass in ONE launch: the table sits in the kernel-argument segment, a workgroup finds its
// layer with scalar compares over begin[] (no memory chain), then runs the same body - same per-element summation order as the
// per-layer launches, hence bit-identical gradients (tests/test_plan_gpu.py).  23 launches of 5 - 22 us (ramp-dominated, ~1.5 TB/s)
// become one that streams every slab of the step.
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const ReduceBatch b)
{
    __shared__ f32x4 red[256];
    int j = 0;                                                     // begin[] past the last layer holds INT_MAX: 32 scalar compares on two wide loads
#pragma unroll
    for (int i = 1; i <= SSIE_REDUCE_BATCH; ++i) j += (int)blockIdx.x >= b.begin[i] ? 1 : 0;
    const long blk = (long)blockIdx.x - b.begin[j];
    if (b.d[j].wide) wgrad_reduce_body<16>(b.d[j], blk, red);
    else wgrad_reduce_body<64>(b.d[j], blk, red);
}

// per-channel sums of G over all pixels (bias gradient of the transposed conv), two-stage & deterministic.
// Stage 1: a thread owns one channel quad (float4) and every (256 / quads)-th pixel of the block's range, with four
// independent loads in flight (the plain one-load-per-iteration loop was latency-bound at ~1 TB/s); fixed-order LDS
// tree over the pixel lanes; stage 2: one block per channel reduces the per-block partials with 256 threads.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ g, long npix, int cstride, int coff, int C,
                                      float* __restrict__ partial /*[gridDim.x][C]*/)
{
    __shared__ f32x4 red[256];
    const int cq = (C + 3) / 4;                      // channel quads (<= 64); coff and cstride are multiples of 4
    const int lanes = 256 / cq;                      // pixel lanes
    const int q = threadIdx.x % cq, pl = threadIdx.x / cq;
    const long per = (npix + gridDim.x - 1) / gridDim.x;
    const long beg = (long)blockIdx.x * per, end = min(beg + per, npix);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (pl < lanes) {
        const float* gp = g + coff + 4 * q;
        long px = beg + pl;
        for (; px + 3L * lanes < end; px += 4L * lanes) {
            const f32x4 v0 = *(const f32x4*)(gp + px * cstride), v1 = *(const f32x4*)(gp + (px + lanes) * cstride);
            const f32x4 v2 = *(const f32x4*)(gp + (px + 2L * lanes) * cstride), v3 = *(const f32x4*)(gp + (px + 3L * lanes) * cstride);
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; px < end; px += lanes) s0 += *(const f32x4*)(gp + px * cstride);
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (pl == 0) {
        f32x4 t = red[q];
        for (int l = 1; l < lanes; ++l) t += red[l * cq + q];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * q + e < C) partial[(size_t)blockIdx.x * C + 4 * q + e] = t[e];
    }
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ dst, int accumulate)
{
    __shared__ float red[256];
    const int c = blockIdx.x, t = threadIdx.x;
    float sum = 0.f;
    for (int b = t; b < nblk; b += 256) sum += partial[(size_t)b * C + c];
    red[t] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    if (t == 0) dst[c] = accumulate ? dst[c] + red[0] : red[0];
}

// ---------------------------------------------------------------------------------------------
// weight packing: dst[chunk][t][q][n][s] = W[base + n*s_n + k*s_k + tapsel[t]*s_t], k = chunk*16 + (q>>1)*8 + (q&1)*4 + s
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ void ssie_pack_one(const PackDesc& d, long idx4_in)
{
    if (d.copy) {                                   // bias vector into its slot of a contiguous one
        if (idx4_in < d.N) d.dst[d.n_off + idx4_in] = d.w[idx4_in];
        return;
    }
    const int ncols = d.ncnt ? d.ncnt : d.Npad;     // sub-block packs iterate over their own columns only
    long total4 = (long)d.nchunks * d.T * 4 * ncols;
    if (idx4_in >= total4) return;
    int n = (int)(idx4_in % ncols); long r = idx4_in / ncols;
    int q = (int)(r & 3); r >>= 2;
    int t = (int)(r % d.T); int chunk = (int)(r / d.T);
    // destination slot: row pitch Npad, column n_off + n, chunk shifted by k_off / 16 (fp32 layout; bf16 / Winograd packs are whole-tensor)
    const long idx4 = (((long)(chunk + (d.k_off >> 4)) * d.T + t) * 4 + q) * d.Npad + d.n_off + n;
    const int ts = (int)d.tapsel[t] * d.s_t;
    if (d.bf16) {          // 8 bf16 (round to nearest even) per 16-byte slot
        const int kb8 = chunk * 32 + q * 8;
        unsigned w[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            float f2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = kb8 + 2 * s2 + e;
                // (unconditional load from a clamped index, value zeroed afterwards: predicated loads are waited for one by one)
                const float wv = d.w[(long)min(n, d.N - 1) * d.s_n + (long)min(k, d.K - 1) * d.s_k + ts];
                f2[e] = (k < d.K && n < d.N) ? wv : 0.f;
            }
            w[s2] = ssie_pack2bf(f2[0], f2[1]);
        }
        f32x4 o = {__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
        ((f32x4*)d.dst)[idx4] = o;
        return;
    }
    f32x4 v;
    const int kb = chunk * 16 + (q >> 1) * 8 + (q & 1) * 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        int k = kb + s;
        const float wv = d.w[(long)min(n, d.N - 1) * d.s_n + (long)min(k, d.K - 1) * d.s_k + ts];
        v[s] = (k < d.K && n < d.N) ? wv : 0.f;
    }
    ((f32x4*)d.dst)[idx4] = v;
}

// Winograd F(2x2, 3x3) weights: one thread = one (chunk, q, n) column of 4 input channels -> the 16 transform positions
// U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; g[r][s] = the tap with (dy, dx) = (r-1, s-1) (PackDesc.tapsel[r*3+s])
__device__ __forceinline__ void ssie_pack_wino_one(const PackDesc& d, long idx)
{
    const long total = (long)d.nchunks * 4 * d.Npad;
    if (idx >= total) return;
    const int n = (int)(idx % d.Npad); long r = idx / d.Npad;
    const int q = (int)(r & 3); const int chunk = (int)(r >> 2);
    const int kb = chunk * 16 + (q >> 1) * 8 + (q & 1) * 4;
    f32x4 g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ts = (int)d.tapsel[t] * d.s_t;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = kb + s;
            const float wv = d.w[(long)min(n, d.N - 1) * d.s_n + (long)min(k, d.K - 1) * d.s_k + ts];      // 36 loads in flight together
            g[t][s] = (k < d.K && n < d.N) ? wv : 0.f;
        }
    }
    f32x4 m[12];      // G g: 4 x 3
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        m[0 * 3 + s] = g[0 * 3 + s];
        m[1 * 3 + s] = 0.5f * (g[0 * 3 + s] + g[1 * 3 + s] + g[2 * 3 + s]);
        m[2 * 3 + s] = 0.5f * (g[0 * 3 + s] - g[1 * 3 + s] + g[2 * 3 + s]);
        m[3 * 3 + s] = g[2 * 3 + s];
    }
    f32x4* dst = (f32x4*)d.dst;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 u[4];
        u[0] = m[i * 3 + 0];
        u[1] = 0.5f * (m[i * 3 + 0] + m[i * 3 + 1] + m[i * 3 + 2]);
        u[2] = 0.5f * (m[i * 3 + 0] - m[i * 3 + 1] + m[i * 3 + 2]);
        u[3] = m[i * 3 + 2];
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[((size_t)(chunk * 16 + i * 4 + j) * 4 + q) * d.Npad + n] = u[j];
    }
}

// Winograd F(4x4, 3x3) weights (conv_wino4.hip): one thread = one (8-channel step, channel pair g, n) -> the 36 transform positions
// U = G g G^T of both channels, written in the LDS image of a K step: dst[step][n / 32][xi][(n % 32) / 16][g][n % 16][2]
__device__ __forceinline__ void ssie_pack_wino4_one(const PackDesc& d, long idx)
{
    const long total = (long)d.nchunks * 4 * d.Npad;
    if (idx >= total) return;
    const int n = (int)(idx % d.Npad); long r = idx / d.Npad;
    const int g = (int)(r & 3); const int step = (int)(r >> 2);
    const int kb = step * 8 + 2 * g;
    float w9[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ts = (int)d.tapsel[t] * d.s_t;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k = kb + s;
            const float wv = d.w[(long)min(n, d.N - 1) * d.s_n + (long)min(k, d.K - 1) * d.s_k + ts];
            w9[t][s] = (k < d.K && n < d.N) ? wv : 0.f;
        }
    }
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    float2* dst = (float2*)d.dst + ((size_t)(step * (d.Npad / 32) + n / 32) * 36) * 128 + (((n % 32) / 16) * 4 + g) * 16 + (n % 16);
    float m[6][3][2];      // G g
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int e = 0; e < 2; ++e) m[i][s][e] = G[i][0] * w9[0 * 3 + s][e] + G[i][1] * w9[1 * 3 + s][e] + G[i][2] * w9[2 * 3 + s][e];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float2 u;
            u.x = m[i][0][0] * G[j][0] + m[i][1][0] * G[j][1] + m[i][2][0] * G[j][2];
            u.y = m[i][0][1] * G[j][0] + m[i][1][1] * G[j][1] + m[i][2][1] * G[j][2];
            dst[(size_t)(i * 6 + j) * 128] = u;
        }
}

__global__ void pack_weights_kernel(const PackDesc d)
{
    if (d.wino == 2) ssie_pack_wino4_one(d, (long)blockIdx.x * blockDim.x + threadIdx.x);
    else if (d.wino) ssie_pack_wino_one(d, (long)blockIdx.x * blockDim.x + threadIdx.x);
    else ssie_pack_one(d, (long)blockIdx.x * blockDim.x + threadIdx.x);
}

// batched: descriptors resident in device memory, blockIdx.y selects the descriptor
__global__ void pack_weights_batched_kernel(const PackDesc* __restrict__ descs)
{
    const PackDesc& d = descs[blockIdx.y];
    if (d.wino) {
        const long total = (long)d.nchunks * 4 * d.Npad;
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
            if (d.wino == 2) ssie_pack_wino4_one(d, idx); else ssie_pack_wino_one(d, idx);
        }
        return;
    }
    long total4 = d.copy ? d.N : (long)d.nchunks * d.T * 4 * (d.ncnt ? d.ncnt : d.Npad);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (long)gridDim.x * blockDim.x)
        ssie_pack_one(d, idx);
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
extern "C" size_t ssie_fprop_lds_bytes(const ConvParams* p, int nt)
{
    return (size_t)p->hp_h * p->hp_w * 64 + (size_t)SSIE_TG * 4 * 32 * nt * 16 + (size_t)SSIE_MAX_TAPS * 4 + 16;
}

// resident persistent workgroups per CU (tuning knob, see tools/bench_conv.py)
int ssie_fprop_wgs_per_cu = 3;
extern "C" void ssie_debug_set_fprop_wgs_per_cu(int v) { ssie_fprop_wgs_per_cu = v < 1 ? 1 : v; }

template <int NT, int NA, int TH, int EPI, bool RAG>
static int launch_fprop_e(const ConvParams& p, size_t lds, hipStream_t st)
{
    static unsigned seen = 0;
    ssie_allow_full_lds((const void*)conv_fprop_kernel<NT, NA, TH, EPI, RAG>, seen);
    const size_t tiles = (size_t)p.N * p.tiles_y * p.tiles_x * p.co_blocks;
    const int per_cu = (int)((160 * 1024) / lds);
    size_t wgs = (size_t)256 * (per_cu < 1 ? 1 : (per_cu > ssie_fprop_wgs_per_cu ? ssie_fprop_wgs_per_cu : per_cu));
    if (wgs > tiles) wgs = tiles;
    hipLaunchKernelGGL((conv_fprop_kernel<NT, NA, TH, EPI, RAG>), dim3((unsigned)wgs), dim3(256), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 14;
}

template <int NT, int NA, int TH>
static int launch_fprop_t(const ConvParams& p, size_t lds, hipStream_t st)
{
    // 8-row tiles: the whole-tile instantiations by epilogue shape where every tile lies inside the output (ssie_epi_shape)
    if constexpr (TH == 8) if (p.Ho % 8 == 0 && p.Wo % SSIE_TW == 0 && (p.Ho - 1) * p.so + p.py < p.Hout && (p.Wo - 1) * p.so + p.px < p.Wout) {
        const int epi = ssie_epi_shape(p);
        return epi == 1 ? launch_fprop_e<NT, NA, 8, 1, false>(p, lds, st) : epi == 2 ? launch_fprop_e<NT, NA, 8, 2, false>(p, lds, st)
                                                                                   : launch_fprop_e<NT, NA, 8, 0, false>(p, lds, st);
    }
    return launch_fprop_e<NT, NA, TH, 0, true>(p, lds, st);
}

int ssie_fprop_use_v2 = 1;
extern "C" void ssie_debug_set_fprop_v2(int v) { ssie_fprop_use_v2 = v; }

static int ssie_launch_fprop_nt1(const ConvParams& p, hipStream_t st)
{
    if (p.Cout_pad % 32 || p.ntaps < 1 || p.ntaps > SSIE_MAX_TAPS) return 11;
    size_t lds = ssie_fprop_lds_bytes(&p, 1);
    const int na = (p.hp_h * p.hp_w * 4 + 255) / 256;
    if (lds > 160 * 1024 || na > 9) return 13;
    if (p.th == 16) return na > 6 ? 16 : launch_fprop_t<1, 6, 16>(p, lds, st);
    if (p.th != 8) return 17;
    return na <= 3 ? launch_fprop_t<1, 3, 8>(p, lds, st) : (na <= 6 ? launch_fprop_t<1, 6, 8>(p, lds, st) : launch_fprop_t<1, 9, 8>(p, lds, st));
}

int ssie_launch_fprop(const ConvParams& p, hipStream_t st)
{
    if (p.wino == 2) return ssie_launch_fprop_wino4(p, st);
    if (p.wino) return ssie_launch_fprop_wino(p, st);
    if (p.tconv) return ssie_launch_tconv(p, st);
    if (ssie_fprop_use_v2 && ssie_fprop_v2_ok(p)) return ssie_launch_fprop_v2(p, st);
    int nt = (p.Cout_pad % 64 == 0) ? 2 : 1;
    if (nt == 2 && (long)p.N * p.tiles_y * p.tiles_x * p.co_blocks < 256) {
        // small launch (batch 1-2): split the 64-channel tiles into 32-channel ones to double the workgroups
        ConvParams q = p;
        q.co_blocks = p.Cout_pad / 32;
        return ssie_launch_fprop_nt1(q, st);
    }
    if (p.Cout_pad % 32) return 11;
    if (p.ntaps < 1 || p.ntaps > SSIE_MAX_TAPS) return 12;
    size_t lds = ssie_fprop_lds_bytes(&p, nt);
    if (lds > 160 * 1024) return 13;
    const int na = (p.hp_h * p.hp_w * 4 + 255) / 256;
    if (na > 9) return 15;
    if (p.th == 16) {
        if (na > 6) return 16;
        return nt == 2 ? launch_fprop_t<2, 6, 16>(p, lds, st) : launch_fprop_t<1, 6, 16>(p, lds, st);
    }
    if (p.th != 8) return 17;
    if (nt == 2) return na <= 3 ? launch_fprop_t<2, 3, 8>(p, lds, st) : (na <= 6 ? launch_fprop_t<2, 6, 8>(p, lds, st) : launch_fprop_t<2, 9, 8>(p, lds, st));
    return na <= 3 ? launch_fprop_t<1, 3, 8>(p, lds, st) : (na <= 6 ? launch_fprop_t<1, 6, 8>(p, lds, st) : launch_fprop_t<1, 9, 8>(p, lds, st));
}

template <int CI, int CO, int NU, int SW = 0>
static int launch_wgrad_t(const WgradParams& p, hipStream_t st)
{
    size_t lds = ((size_t)p.hp_h * p.hp_w * CI + (size_t)p.th * SSIE_TW * CO) * 4;
    constexpr int NPAIR = (CI / 32) * (CO / 32);
    if (NPAIR < 4) {                                   // scratch of the in-workgroup K-split reduction
        const size_t red = (size_t)NPAIR * NU * 16 * 64 * 4;
        if (lds < red) lds = red;
    }
    if (lds > 160 * 1024) return 23;
    static unsigned seen = 0;
    ssie_allow_full_lds((const void*)conv_wgrad_kernel<CI, CO, NU, SW>, seen);
    dim3 grid(p.nslices, p.ci_blocks * p.co_blocks, p.tap_groups);      // SW = 4: tap_groups = pairs of kernel rows
    hipLaunchKernelGGL((conv_wgrad_kernel<CI, CO, NU, SW>), grid, dim3(256), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 24;
}

int ssie_wgrad_rows2 = 1;        // A/B switch: 1 = two kernel rows per workgroup for the 9 x 9 weight gradient (SW = 4)
extern "C" void ssie_debug_set_wgrad_rows2(int v) { ssie_wgrad_rows2 = v; }
int ssie_wgrad_sliding = 1;      // A/B switch: 1 = sliding-window K loop for stride-1 3x3 / 9x9 layers
extern "C" void ssie_debug_set_wgrad_sliding(int v) { ssie_wgrad_sliding = v; }

// 1: the tap list is a 3 x 3 block, 2: rows of a 9 x 9 kernel (each group of 9 = one row, dx ascending), 0: anything else
static int wgrad_window_kind(const WgradParams& p)
{
    if (!ssie_wgrad_sliding || (p.ntaps != 9 && p.ntaps != 81)) return 0;
    if (!((p.si == 1 && p.th == 8) || (p.si == 2 && p.th == 4 && p.ntaps == 9))) return 0;
    const int ndx = p.ntaps == 9 ? 3 : 9;
    for (int t = 0; t < p.ntaps; ++t) {
        const int g = t / 9, u = t % 9;
        const int dy0 = p.tap_dy[g * 9], dx0 = p.min_dx;
        if (p.tap_dy[t] != dy0 + u / ndx || p.tap_dx[t] != dx0 + u % ndx) return 0;
    }
    return p.ntaps == 9 ? (p.si == 2 ? 3 : 1) : 2;
}

// cib/cob chosen by the caller through ci_pad/ci_blocks (ci_pad = ci_blocks*CIB)
int ssie_launch_wgrad(const WgradParams& p, hipStream_t st)
{
    if (p.wino) return ssie_launch_wgrad_wino(p, st);
    const int cib = p.ci_pad / p.ci_blocks, cob = p.co_pad / p.co_blocks;
    const bool one = p.ntaps == 1;
    const int sw = wgrad_window_kind(p);
    if (sw == 1) {
        if (cib == 64 && cob == 64) return launch_wgrad_t<64, 64, 9, 1>(p, st);
        if (cib == 32 && cob == 64) return launch_wgrad_t<32, 64, 9, 1>(p, st);
        if (cib == 64 && cob == 32) return launch_wgrad_t<64, 32, 9, 1>(p, st);
        if (cib == 32 && cob == 32) return launch_wgrad_t<32, 32, 9, 1>(p, st);
    }
    if (sw == 3 && cib == 64 && cob == 64) return launch_wgrad_t<64, 64, 9, 3>(p, st);
    if (sw == 2 && p.rows2 && cib == 32 && cob == 64) return launch_wgrad_t<32, 64, 9, 4>(p, st);
    if (p.rows2) return 22;                               // geometry built for the two-row kernel but the tap list is not 9 x 9 rows
    if (sw == 2) {
        if (cib == 64 && cob == 64) return launch_wgrad_t<64, 64, 9, 2>(p, st);
        if (cib == 32 && cob == 64) return launch_wgrad_t<32, 64, 9, 2>(p, st);
        if (cib == 64 && cob == 32) return launch_wgrad_t<64, 32, 9, 2>(p, st);
        if (cib == 32 && cob == 32) return launch_wgrad_t<32, 32, 9, 2>(p, st);
    }
    if (cib == 64 && cob == 64) return one ? launch_wgrad_t<64, 64, 1>(p, st) : launch_wgrad_t<64, 64, 9>(p, st);
    if (cib == 32 && cob == 64) return one ? launch_wgrad_t<32, 64, 1>(p, st) : launch_wgrad_t<32, 64, 9>(p, st);
    if (cib == 64 && cob == 32) return one ? launch_wgrad_t<64, 32, 1>(p, st) : launch_wgrad_t<64, 32, 9>(p, st);
    if (cib == 32 && cob == 32) return one ? launch_wgrad_t<32, 32, 1>(p, st) : launch_wgrad_t<32, 32, 9>(p, st);
    return 21;
}

int ssie_wgrad_reduce_wide_min = 64;      // launches with at least this many slices: 16 output quads x 16 slice groups per block
extern "C" void ssie_debug_set_wgrad_reduce_wide_min(int v) { ssie_wgrad_reduce_wide_min = v; }
ReduceDesc ssie_make_reduce(const float* slabs, int nslices, int ntaps, int ci_pad, int co_pad, int Cin, int Cout, float* dst, long s_co, long s_ci,
                            long s_t, const float* bias_slabs, float* db, int accumulate, int accumulate_bias, int co_group, long w_extra, long b_extra)
{
    ReduceDesc d; memset(&d, 0, sizeof(d));
    d.slabs = slabs; d.dst = dst; d.bias_slabs = bias_slabs; d.db = db;
    d.s_co = s_co; d.s_ci = s_ci; d.s_t = s_t; d.w_extra = w_extra; d.b_extra = b_extra;
    d.nslices = nslices; d.ntaps = ntaps; d.ci_pad = ci_pad; d.co_pad = co_pad; d.Cin = Cin; d.Cout = Cout;
    d.accumulate = accumulate; d.accumulate_bias = accumulate_bias < 0 ? accumulate : accumulate_bias; d.co_group = co_group;
    d.wide = nslices >= ssie_wgrad_reduce_wide_min;
    return d;
}
static long reduce_blocks(const ReduceDesc& d)
{
    const long cq = (d.Cout + 3) / 4;
    const long total = (long)d.ntaps * d.Cin * cq + (d.bias_slabs ? cq : 0);
    return d.wide ? (total + 15) / 16 : (total + 63) / 64;
}
int ssie_launch_wgrad_reduce(const float* slabs, int nslices, int ntaps, int ci_pad, int co_pad, int Cin, int Cout,
                             float* dst, long s_co, long s_ci, long s_t, const float* bias_slabs, float* db,
                             int accumulate, hipStream_t st, int accumulate_bias, int co_group, long w_extra, long b_extra)
{
    const ReduceDesc d = ssie_make_reduce(slabs, nslices, ntaps, ci_pad, co_pad, Cin, Cout, dst, s_co, s_ci, s_t, bias_slabs, db, accumulate,
                                          accumulate_bias, co_group, w_extra, b_extra);
    if (d.wide) hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)reduce_blocks(d)), dim3(256), 0, st, d);
    else hipLaunchKernelGGL(wgrad_reduce_kernel<64>, dim3((unsigned)reduce_blocks(d)), dim3(256), 0, st, d);
    return hipGetLastError() == hipSuccess ? 0 : 25;
}

int ssie_launch_wgrad_reduce_batched(const ReduceDesc* d, int n, hipStream_t st)
{
    if (n < 1 || n > SSIE_REDUCE_BATCH) return 25;
    ReduceBatch b; memset(&b, 0, sizeof(b));
    b.n = n;
    long at = 0;
    for (int j = 0; j < n; ++j) { b.begin[j] = (int)at; b.d[j] = d[j]; at += reduce_blocks(d[j]); }
    if (at >= 0x7fffffffL) return 25;
    for (int j = n; j <= SSIE_REDUCE_BATCH; ++j) b.begin[j] = 0x7fffffff;
    hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3((unsigned)at), dim3(256), 0, st, b);
    return hipGetLastError() == hipSuccess ? 0 : 25;
}

int ssie_launch_colsum(const float* g, long npix, int cstride, int coff, int C, float* partial, int nblk,
                       float* dst, int accumulate, hipStream_t st)
{
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, st, g, npix, cstride, coff, C, partial);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(256), 0, st, partial, nblk, C, dst, accumulate);
    return hipGetLastError() == hipSuccess ? 0 : 26;
}

int ssie_launch_pack(const PackDesc& d, hipStream_t st)
{
    long total4 = d.copy ? d.N : (long)d.nchunks * d.T * 4 * d.Npad;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, d);
    return hipGetLastError() == hipSuccess ? 0 : 27;
}

int ssie_launch_pack_batched(const PackDesc* descs_dev, int ndesc, hipStream_t st)
{
    hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(64, ndesc), dim3(256), 0, st, descs_dev);
    return hipGetLastError() == hipSuccess ? 0 : 28;
}
