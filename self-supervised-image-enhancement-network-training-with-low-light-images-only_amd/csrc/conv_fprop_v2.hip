// conv_fprop_v2_kernel / conv_fprop_v2w_kernel: the implicit-GEMM convolution (forward + data gradient) of every launch
// large enough to give each CU a tile, restructured for ONE 512-thread workgroup per CU:
//   * 16 x 16 output positions (8 x 16 at stride 2) x 64 (or 32) channels per tile; 8 waves = 4 (M) x 2 (N), two waves per
//     SIMD; the wide variant (big 64-channel 3x3 layers) takes 16 x 32 positions with a 64 x 64 register tile per wave
//   * LDS is DOUBLE-buffered and filled by direct global->LDS DMA (global_load_lds_dwordx4): no staging VGPRs, no
//     ds_write pass, and exactly ONE barrier per step - the loads of step s+1 are issued right after the barrier of
//     step s and have the whole MFMA phase of step s to land
//   * the halo tile's 16-byte slot swizzle moves to the per-lane SOURCE address (the DMA writes LDS linearly:
//     wave-uniform base + lane*16); out-of-image / out-of-channel slots read a zero page in global memory
//   * persistent workgroups + dynamic tile queue + cross-tile prefetch exactly as in the v1 kernel
// Same math, same packed-weight layout, same epilogue as conv_fprop_kernel (conv_kernels.hip).
#include "conv_device.h"

__device__ f32x4 ssie_zero_page[4];   // zero-initialised: source of padding slots

// Diagnostic build only (-DSSIE_STAMP, tools/stamp_v2.py): wave 0's s_memtime spent per phase, summed per workgroup:
// [0] start [1] barrier wait at a tile's first step [2] barrier wait at other steps [3] end [4] DMA issue [5] epilogue
// [6] tiles [7] MFMA tap loops.  The shipped library never executes a stamp.
#ifdef SSIE_STAMP
__device__ unsigned long long* ssie_stamp_buf_v2 = nullptr;
extern "C" int ssie_debug_set_stamp_buffer_v2(void* buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(ssie_stamp_buf_v2), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#define ST_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t_ = __builtin_amdgcn_s_memtime(); st_[0] = st_t_;
#define ST_ACC(k) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[k] += t_ - st_t_; st_t_ = t_; } while (0)
#define ST_FLUSH do { st_[3] = __builtin_amdgcn_s_memtime(); if (ssie_stamp_buf_v2 && threadIdx.x == 0) \
    for (int k_ = 0; k_ < 8; ++k_) ssie_stamp_buf_v2[(size_t)blockIdx.x * 8 + k_] = st_[k_]; } while (0)
#else
#define ST_DECL
#define ST_ACC(k)
#define ST_FLUSH
#endif

#define GLDS16(gptr, lptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// source address of 4 consecutive channels of virtual pixel (n, vy, vx), or the zero page.  `up` (wave-uniform)
// selects the nearest-up-sampling path; plain sources take a handful of 32-bit integer ops.
__device__ __forceinline__ const f32x4* ssie_virtual_addr(const SrcSel& s, bool up, int n, int vy, int vx, int Hv, int Wv, int c)
{
    const bool ok = (unsigned)vy < (unsigned)Hv && (unsigned)vx < (unsigned)Wv && c < s.C;
    int y = vy, x = vx;
    if (up) {
        const int cy = min(max(vy, 0), Hv - 1), cx = min(max(vx, 0), Wv - 1);
        y = min((int)floorf((float)cy * s.sy), s.Hs - 1);
        x = min((int)floorf((float)cx * s.sx), s.Ws - 1);
    }
    const unsigned off = (unsigned)((n * s.Hs + y) * s.Ws + x) * (unsigned)s.cstride + (unsigned)(s.coff + c);
    return ok ? (const f32x4*)(s.ptr + off) : (const f32x4*)ssie_zero_page;
}

// NW = waves per workgroup: 8 (one 512-thread workgroup per CU, 64- or 32-channel tiles) or 4 (TWO independent
// 256-thread workgroups per CU, 32-channel tiles, ~78 KiB LDS each): the two workgroups drift out of phase, so the DMA
// issue / epilogue / barrier phases of one run underneath the MFMA phase of the other (tools/stamp_v2.py: those phases
// are ~20 % of a lock-stepped 8-wave workgroup's time)
// TH = 16: stride-1 layers; TH = 8: stride-2 layers (8 x 16 output positions read a 17 x 33 halo)
// EPI / RAG: epilogue shape (ssie_epi_shape) and "some tile sticks out of the output"; the plain / whole-tile instantiations carry a
// fraction of the epilogue code (instantiated for the stride-2 geometry only, the one that runs at bench sizes)
// TGM = taps a weight buffer holds: SSIE_TG, or 1 for the 1 x 1 layers' own instantiations - their LDS footprint is then 40 KB instead
// of 106, and TWO workgroups fit a CU.  A 1 x 1 step is 16 MFMAs per wave against a DMA round trip of microseconds with one step of
// prefetch: one workgroup per CU ran feature_fusion at 2 TB/s of a pure streaming pass.
template <int NT, int NA2, int NW, int TH, int EPI = 0, bool RAG = true, int TGM = SSIE_TG>
__global__ __launch_bounds__(64 * NW, (TGM == 1 ? 4 : 2)) void conv_fprop_v2_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int BN = 32 * NT;
    constexpr int NTHR = 64 * NW;
    constexpr int WM = (NT == 2) ? NW / 2 : NW;     // waves along M; a TH x 16 tile has TH/2 M-tiles of 2 x 16 positions
    constexpr int MT = (TH / 2) / WM;
    static_assert(MT >= 1, "tile too small for the wave grid");
    constexpr int BSZ = TGM * 4 * BN;               // float4 per B buffer
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

    // this lane's halo slots: LDS slot id = i*NTHR + tid (linear), it holds channel quad j = (id&3) ^ swz(pixel)
    int ahy[NA2], ahx[NA2], aj[NA2];
#pragma unroll
    for (int i = 0; i < NA2; ++i) {
        const int id = min(tid + i * NTHR, HP4 - 1);
        const int pix = id >> 2;
        ahy[i] = pix / p.hp_w; ahx[i] = pix - ahy[i] * p.hp_w; aj[i] = (id & 3) ^ ssie_swz(pix);
    }

#define V2_DECODE(T, N_, A0_, B0_, CO0_)                                                  \
    {                                                                                     \
        int q_ = (T);                                                                     \
        CO0_ = (q_ % p.co_blocks) * BN; q_ /= p.co_blocks;                                \
        B0_ = (q_ % p.tiles_x) * SSIE_TW; q_ /= p.tiles_x;                                \
        A0_ = (q_ % p.tiles_y) * TH; N_ = q_ / p.tiles_y;                                 \
    }
    // DMA the operands of step (CHUNK, G) of tile (N_, A0_, B0_, CO0_): weights into B buffer BUF, and (first tap group
    // of a chunk only) the halo tile into A buffer ABUF
#define V2_PREFETCH(CHUNK, G, N_, A0_, B0_, CO0_, BUF, ABUF)                                                        \
    {                                                                                                         \
        if ((G) == 0) {                                                                                       \
            const SrcSel s_ = ssie_pick_src(p, (CHUNK) * SSIE_CK);                                            \
            const bool up_ = s_.sy != 1.f || s_.sx != 1.f;                                                    \
            const int vy0_ = (A0_) * p.si + p.min_dy, vx0_ = (B0_) * p.si + p.min_dx;                                       \
            f32x4* abuf_ = As0 + (ABUF) * HP4;                                                                \
            _Pragma("unroll") for (int i_ = 0; i_ < NA2; ++i_) {                                              \
                if (tid + i_ * NTHR < HP4) {                                                                  \
                    const f32x4* g_ = ssie_virtual_addr(s_, up_, (N_), vy0_ + ahy[i_], vx0_ + ahx[i_], p.Hv, p.Wv, \
                                                        (CHUNK) * SSIE_CK + 4 * aj[i_] - s_.cbeg);            \
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
    V2_DECODE(tile, n, a0, b0, co0)
    int gstep = 0;
    int a_cur = 0;          // A buffer holding the halo tile of the step about to be computed
    V2_PREFETCH(0, 0, n, a0, b0, co0, 0, 0)
    int fetched = 0x7fffffff;
    ST_DECL
    // The two waves of a SIMD (w and w+4) run the same program; issuing the next step's DMA (address VALU work) at
    // the same moment would leave the SIMD's MFMA pipe idle.  Waves 4-7 therefore issue it in the middle of their
    // tap loop while waves 0-3 issue it up front.
    const bool late_prefetch = false;   // measured: issuing the DMA inside the tap loop (waves 4-7) was 5-25 % SLOWER

    while (tile < total_tiles) {
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        // the bias is fetched HERE, ahead of the step-0 barrier that drains vmcnt anyway: vmcnt retires in issue order, so the
        // same load issued in the epilogue would wait behind the next tile's whole halo + weight prefetch
        const float bv_tile = (p.bias && co0 + wn * 32 + li < p.Cout) ? p.bias[co0 + wn * 32 + li] : 0.f;
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
            ST_ACC(5);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            ST_ACC(step == 0 ? 1 : 2);
            if (step == (nsteps > 1 ? 1 : 0)) {
                ntile = *s_next;
                if (ntile < total_tiles) V2_DECODE(ntile, nn, na0, nb0, nco0)
            }
            int nchunk = chunk, ng = g + 1;
            if (ng == ngroups) { ng = 0; ++nchunk; }
            const bool more = step + 1 < nsteps;
            const int a_nxt = ((more ? ng : 0) == 0) ? (a_cur ^ 1) : a_cur;     // a new halo tile goes to the other A buffer
#define V2_ISSUE_NEXT                                                                                     \
            {                                                                                             \
                if (more) V2_PREFETCH(nchunk, ng, n, a0, b0, co0, buf ^ 1, a_nxt)                         \
                else if (ntile < total_tiles) V2_PREFETCH(0, 0, nn, na0, nb0, nco0, buf ^ 1, a_nxt)       \
            }
            if (!late_prefetch) V2_ISSUE_NEXT
            ST_ACC(4);

            const f32x4* As = As0 + a_cur * HP4;
            const f32x4* Bs = Bs0 + buf * BSZ;
#define V2_LDFRAG(BF, AF, TL, KQ, OFF)                                                                \
            {                                                                                             \
                BF = Bs[((TL) * 4 + (KQ) * 2 + h) * BN + wn * 32 + li];                                   \
                _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_) {                                       \
                    const int hp_ = pixbase[m_] + (OFF);                                                  \
                    AF[m_] = As[hp_ * 4 + (((KQ) * 2 + h) ^ ssie_swz(hp_))];                              \
                }                                                                                         \
            }
#define V2_MFMA4(BF, AF)                                                                              \
            _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_) {                                           \
                acc[m_] = MFMA32(AF[m_].x, BF.x, acc[m_]); acc[m_] = MFMA32(AF[m_].y, BF.y, acc[m_]);     \
                acc[m_] = MFMA32(AF[m_].z, BF.z, acc[m_]); acc[m_] = MFMA32(AF[m_].w, BF.w, acc[m_]);     \
            }
            {
                f32x4 bX, bY, aX[MT], aY[MT];
                int off = tapoff[t0];
                V2_LDFRAG(bX, aX, 0, 0, off)
                const int pf_at = late_prefetch ? (tg >> 1) : -1;
                for (int tl = 0; tl < tg; ++tl) {
                    const int off_next = tapoff[t0 + min(tl + 1, tg - 1)];
                    V2_LDFRAG(bY, aY, tl, 1, off)
                    V2_MFMA4(bX, aX)
                    if (tl == pf_at) V2_ISSUE_NEXT
                    if (tl + 1 < tg) V2_LDFRAG(bX, aX, tl + 1, 0, off_next)
                    V2_MFMA4(bY, aY)
                    off = off_next;
                }
            }
#undef V2_LDFRAG
#undef V2_MFMA4
#undef V2_ISSUE_NEXT
            // draw the tile after next from the queue; its value is only needed at the next step's hand-off
            if (tid == 0 && step == 0 && nsteps > 1 && p.tile_counter)
                fetched = (int)gridDim.x + atomicAdd(p.tile_counter, 1);
            chunk = nchunk; g = ng; a_cur = a_nxt;
            ST_ACC(7);
        }

        // epilogue (identical mapping to the v1 kernel; TH = 16 so M-tile mt covers tile rows 2*mt, 2*mt+1)
        const int co = co0 + wn * 32 + li;
        if (co < p.Cout) {
            const float bv = bv_tile;
            const long rowstride = (long)p.so * p.Wout * p.out_cstride;
            const long pixstride = (long)p.so * p.out_cstride;
            const bool full = !RAG || (a0 + TH <= p.Ho && b0 + SSIE_TW <= p.Wo &&
                                       (a0 + TH - 1) * p.so + p.py < p.Hout && (b0 + SSIE_TW - 1) * p.so + p.px < p.Wout);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int mt = wm * MT + m;
                const int arow = a0 + 2 * mt, bcol = b0 + 4 * h;
                const size_t o0 = ((size_t)(n * p.Hout + arow * p.so + p.py) * p.Wout + bcol * p.so + p.px) * p.out_cstride + p.out_coff + co;
                if (full) { ssie_epilogue_full<EPI>(p, acc[m], o0, rowstride, pixstride, bv); continue; }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tr = r >> 3, tc = (r & 3) + 8 * ((r >> 2) & 1);
                    const int a = arow + tr, b = bcol + tc;
                    if (a >= p.Ho || b >= p.Wo) continue;
                    if (a * p.so + p.py >= p.Hout || b * p.so + p.px >= p.Wout) continue;
                    const size_t o = o0 + tr * rowstride + tc * pixstride;
                    ssie_epilogue_elem(p, o, acc[m][r] + bv);
                }
            }
        }
        n = nn; a0 = na0; b0 = nb0; co0 = nco0; tile = ntile;
#ifdef SSIE_STAMP
        st_[6] += 1;
#endif
    }
    ST_ACC(5);
    ST_FLUSH;
#undef V2_PREFETCH
#undef V2_DECODE
}

// Wide variant for the big 64-channel stride-1 3x3 layers: tile = 16 x 32 positions x 64 channels; every wave owns
// 2 M-tiles x BOTH N-tiles (a 64-position x 64-channel register tile, 4 accumulators), so one K step of a (tap, kq) pair
// costs 4 ds_read_b128 per 16 MFMAs instead of 3 per 8, and a step moves 76 KB of DMA for twice the FLOPs of the
// 16 x 16 tile's 57.6 KB: -33 % LDS bytes, -34 % DMA bytes, half the barriers per FLOP (the fp32 MFMA kernels are
// clock-limited by energy per FLOP).
// SINGLE = the virtual input is one tensor (no concat): the source descriptor is then a compile-time choice and the two
// unused descriptors never occupy scalar registers (the kernel spills ~130 SGPRs to VGPR lanes otherwise, and every
// restore is a v_readlane + wait states inside the per-step DMA-issue phase).
template <int NA2, bool SINGLE, int EPI = 0, bool RAG = true>
__global__ __launch_bounds__(512, 2) void conv_fprop_v2w_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int NT = 2, NW = 8, BN = 64, TH = 16, TWW = 32;
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

#define V2_DECODE(T, N_, A0_, B0_, CO0_)                                                  \
    {                                                                                     \
        int q_ = (T);                                                                     \
        CO0_ = (q_ % p.co_blocks) * BN; q_ /= p.co_blocks;                                \
        B0_ = (q_ % p.tiles_x) * TWW; q_ /= p.tiles_x;                                  \
        A0_ = (q_ % p.tiles_y) * TH; N_ = q_ / p.tiles_y;                                 \
    }
    // DMA the operands of step (CHUNK, G) of tile (N_, A0_, B0_, CO0_): weights into B buffer BUF, and (first tap group
    // of a chunk only) the halo tile into A buffer ABUF
#define V2_PREFETCH(CHUNK, G, N_, A0_, B0_, CO0_, BUF, ABUF)                                                        \
    {                                                                                                         \
        if ((G) == 0) {                                                                                       \
            const SrcSel s_ = SINGLE ? ssie_only_src(p) : ssie_pick_src(p, (CHUNK) * SSIE_CK);                \
            const bool up_ = s_.sy != 1.f || s_.sx != 1.f;                                                    \
            const int vy0_ = (A0_) + p.min_dy, vx0_ = (B0_) + p.min_dx;                                       \
            f32x4* abuf_ = As0 + (ABUF) * HP4;                                                                \
            _Pragma("unroll") for (int i_ = 0; i_ < NA2; ++i_) {                                              \
                if (tid + i_ * NTHR < HP4) {                                                                  \
                    const f32x4* g_ = ssie_virtual_addr(s_, up_, (N_), vy0_ + ahy[i_], vx0_ + ahx[i_], p.Hv, p.Wv, \
                                                        (CHUNK) * SSIE_CK + 4 * aj[i_] - s_.cbeg);            \
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
    V2_DECODE(tile, n, a0, b0, co0)
    int gstep = 0;
    int a_cur = 0;          // A buffer holding the halo tile of the step about to be computed
    V2_PREFETCH(0, 0, n, a0, b0, co0, 0, 0)
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
        // bias fetched ahead of the step-0 barrier (see conv_fprop_v2_kernel)
        float bv_tile[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) bv_tile[c] = (p.bias && co0 + c * 32 + li < p.Cout) ? p.bias[co0 + c * 32 + li] : 0.f;
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
                if (ntile < total_tiles) V2_DECODE(ntile, nn, na0, nb0, nco0)
            }
            int nchunk = chunk, ng = g + 1;
            if (ng == ngroups) { ng = 0; ++nchunk; }
            const bool more = step + 1 < nsteps;
            const int a_nxt = ((more ? ng : 0) == 0) ? (a_cur ^ 1) : a_cur;     // a new halo tile goes to the other A buffer
#define V2_ISSUE_NEXT                                                                                     \
            {                                                                                             \
                if (more) V2_PREFETCH(nchunk, ng, n, a0, b0, co0, buf ^ 1, a_nxt)                         \
                else if (ntile < total_tiles) V2_PREFETCH(0, 0, nn, na0, nb0, nco0, buf ^ 1, a_nxt)       \
            }
            V2_ISSUE_NEXT

            const char* Ab = (const char*)(As0 + a_cur * HP4);
            const f32x4* Bl = Bs0 + buf * BSZ + h * BN + li;          // this lane's column of the weight group
#define W_LD(BF, AF, TL, KQ)                                                                              \
            {                                                                                             \
                _Pragma("unroll") for (int c_ = 0; c_ < NT; ++c_) BF[c_] = Bl[((TL) * 4 + (KQ) * 2) * BN + c_ * 32]; \
                _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_)                                         \
                    AF[m_] = *(const f32x4*)(Ab + ((KQ) ? (aaddr[TL][m_] ^ 32) : aaddr[TL][m_]));         \
            }
#define W_MFMA(BF, AF)                                                                                    \
            _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_)                                             \
            _Pragma("unroll") for (int c_ = 0; c_ < NT; ++c_) {                                           \
                acc[m_][c_] = MFMA32(AF[m_].x, BF[c_].x, acc[m_][c_]); acc[m_][c_] = MFMA32(AF[m_].y, BF[c_].y, acc[m_][c_]); \
                acc[m_][c_] = MFMA32(AF[m_].z, BF[c_].z, acc[m_][c_]); acc[m_][c_] = MFMA32(AF[m_].w, BF[c_].w, acc[m_][c_]); \
            }
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
#undef V2_ISSUE_NEXT
            // draw the tile after next from the queue; its value is only needed at the next step's hand-off
            if (tid == 0 && step == 0 && nsteps > 1 && p.tile_counter)
                fetched = (int)gridDim.x + atomicAdd(p.tile_counter, 1);
            chunk = nchunk; g = ng; a_cur = a_nxt;
        }

        // epilogue: wave w holds tile rows 2w, 2w+1; M-tile m = columns 16m .. 16m+15, N-tile c = channels 32c .. 32c+31
        {
            const long rowstride = (long)p.so * p.Wout * p.out_cstride;
            const long pixstride = (long)p.so * p.out_cstride;
            const bool full = !RAG || (a0 + TH <= p.Ho && b0 + TWW <= p.Wo &&
                                       (a0 + TH - 1) * p.so + p.py < p.Hout && (b0 + TWW - 1) * p.so + p.px < p.Wout);
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const int co = co0 + c * 32 + li;
                if (co >= p.Cout) continue;
                const float bv = bv_tile[c];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int arow = a0 + 2 * wave, bcol = b0 + 16 * m + 4 * h;
                    const size_t o0 = ((size_t)(n * p.Hout + arow * p.so + p.py) * p.Wout + bcol * p.so + p.px) * p.out_cstride + p.out_coff + co;
                    if (full) ssie_epilogue_full<EPI>(p, acc[m][c], o0, rowstride, pixstride, bv);
                    else ssie_epilogue_ragged(p, acc[m][c], o0, rowstride, pixstride, bv, arow, bcol);
                }
            }
        }
        n = nn; a0 = na0; b0 = nb0; co0 = nco0; tile = ntile;
    }
#undef V2_PREFETCH
#undef V2_DECODE
}


#define INST_V2(NT, NA2, NW, TH) template __global__ void conv_fprop_v2_kernel<NT, NA2, NW, TH>(const ConvParams);
INST_V2(1, 3, 8, 16) INST_V2(1, 5, 8, 16) INST_V2(2, 3, 8, 16) INST_V2(2, 5, 8, 16) INST_V2(1, 6, 4, 16) INST_V2(2, 5, 8, 8)
#define INST_V2S(E) template __global__ void conv_fprop_v2_kernel<2, 5, 8, 8, E, false>(const ConvParams);
INST_V2S(0) INST_V2S(1) INST_V2S(2)
template __global__ void conv_fprop_v2_kernel<2, 3, 8, 16, 1, false, 1>(const ConvParams);       // 1 x 1, plain forward / plain data gradient, whole tiles
template __global__ void conv_fprop_v2_kernel<2, 3, 8, 16, 2, false, 1>(const ConvParams);
template __global__ void conv_fprop_v2_kernel<2, 3, 8, 16, 0, false, 1>(const ConvParams);       // (general epilogue: 128 registers + 4 spilled dwords under the four-waves-per-SIMD bound)
#define INST_V2W(S) template __global__ void conv_fprop_v2w_kernel<5, S, 0, true>(const ConvParams); \
                    template __global__ void conv_fprop_v2w_kernel<5, S, 0, false>(const ConvParams); \
                    template __global__ void conv_fprop_v2w_kernel<5, S, 1, false>(const ConvParams); \
                    template __global__ void conv_fprop_v2w_kernel<5, S, 2, false>(const ConvParams);
INST_V2W(false) INST_V2W(true)

size_t ssie_fprop_v2_lds_bytes(const ConvParams& p, int nt)
{
    return 2 * ((size_t)p.hp_h * p.hp_w * 64 + (size_t)SSIE_TG * 4 * 32 * nt * 16) + (size_t)SSIE_MAX_TAPS * 4 + 16;
}

int ssie_fprop_v2_stride2 = 1;     // A/B switch: 1 = stride-2 64-channel layers on the DMA kernel (8 x 16 tiles)
extern "C" void ssie_debug_set_fprop_v2_stride2(int v) { ssie_fprop_v2_stride2 = v; }

// eligible: stride-1 geometry built with th == 16 (or stride-2, th == 8, 64-channel tiles, enough tiles to fill the chip)
// whose double-buffered tiles fit the 160 KiB LDS
bool ssie_fprop_v2_ok(const ConvParams& p)
{
    const int nt = (p.Cout_pad % 64 == 0) ? 2 : 1;
    if (!((p.th == 16 && p.si == 1) || (p.th == 8 && p.si == 2 && nt == 2 && ssie_fprop_v2_stride2 &&
           (long)p.N * p.tiles_y * p.tiles_x * p.co_blocks >= ssie_fprop_min_tiles16))) return false;
    const int na2 = (p.hp_h * p.hp_w * 4 + 511) / 512;
    return na2 <= 5 && ssie_fprop_v2_lds_bytes(p, nt) <= 160 * 1024;
}

static int g_v2_split = 1;      // ssie_debug_set_fprop_v2_split: 0 = always one 8-wave workgroup per CU
static int g_v2_split_min_tiles = 1024;   // ... for launches with at least this many 32-channel tiles (tests set 1)
extern "C" void ssie_debug_set_fprop_v2_split(int on) { g_v2_split = on; }
extern "C" void ssie_debug_set_fprop_v2_split_min_tiles(int v) { g_v2_split_min_tiles = v; }

template <int NT, int NA2, int NW, int TH = 16, int EPI = 0, bool RAG = true, int TGM = SSIE_TG>
static int launch_v2_t(const ConvParams& p, size_t lds, hipStream_t st)
{
    static unsigned seen = 0;
    ssie_allow_full_lds((const void*)conv_fprop_v2_kernel<NT, NA2, NW, TH, EPI, RAG, TGM>, seen);
    const size_t tiles = (size_t)p.N * p.tiles_y * p.tiles_x * p.co_blocks;
    const size_t cap = (NW == 8 && TGM == SSIE_TG) ? 256 : 512;          // (the 1-tap form: two 8-wave workgroups per CU)
    const size_t wgs = tiles < cap ? tiles : cap;
    hipLaunchKernelGGL((conv_fprop_v2_kernel<NT, NA2, NW, TH, EPI, RAG, TGM>), dim3((unsigned)wgs), dim3(64 * NW), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 18;
}

static int g_v2_onetap = 1;     // ssie_debug_set_fprop_v2_onetap: 0 = 1 x 1 layers on the general instantiation, one workgroup per CU
extern "C" void ssie_debug_set_fprop_v2_onetap(int on) { g_v2_onetap = on; }

// every tile of the launch lies inside the output (so the element-wise edge epilogue is never needed)
static bool whole_tiles(const ConvParams& p, int th, int tw)
{
    return p.Ho % th == 0 && p.Wo % tw == 0 && (p.Ho - 1) * p.so + p.py < p.Hout && (p.Wo - 1) * p.so + p.px < p.Wout;
}

int ssie_launch_fprop_v2(const ConvParams& p, hipStream_t st)
{
    const int nt = (p.Cout_pad % 64 == 0) ? 2 : 1;
    if (p.tw == 32) {      // geometry built for the wide kernel (ssie_make_conv)
        const size_t lds = ssie_fprop_v2_lds_bytes(p, 2);
        const size_t tiles = (size_t)p.N * p.tiles_y * p.tiles_x * p.co_blocks;
        const dim3 grid((unsigned)(tiles < 256 ? tiles : 256));
        const int epi = ssie_epi_shape(p);
        const bool rag = !whole_tiles(p, 16, 32);
        static unsigned seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define V2W_GO(S, E, R, SLOT) { ssie_allow_full_lds((const void*)conv_fprop_v2w_kernel<5, S, E, R>, seen[SLOT]); \
                                hipLaunchKernelGGL((conv_fprop_v2w_kernel<5, S, E, R>), grid, dim3(512), lds, st, p); }
#define V2W_PICK(S, B) { if (rag) V2W_GO(S, 0, true, B) else if (epi == 1) V2W_GO(S, 1, false, B + 1) else if (epi == 2) V2W_GO(S, 2, false, B + 2) else V2W_GO(S, 0, false, B + 3) }
        if (p.nsrc == 1) V2W_PICK(true, 0) else V2W_PICK(false, 4)
#undef V2W_PICK
#undef V2W_GO
        return hipGetLastError() == hipSuccess ? 0 : 19;
    }
    // 32-channel layers: two 4-wave workgroups per CU (both must fit the LDS, and there must be enough tiles to fill
    // them).  Measured: +7 % over one 8-wave workgroup there; splitting a 64-channel tile into two 32-channel
    // workgroups instead costs 6 % (the halo tile is fetched twice).
    {
        const size_t lds1 = ssie_fprop_v2_lds_bytes(p, 1);
        const int na4 = (p.hp_h * p.hp_w * 4 + 255) / 256;
        const size_t tiles32 = (size_t)p.N * p.tiles_y * p.tiles_x * (p.Cout_pad / 32);
        if (g_v2_split && nt == 1 && 2 * (lds1 + 256) <= 160 * 1024 && na4 <= 6 && tiles32 >= (size_t)g_v2_split_min_tiles) {
            ConvParams q = p;
            q.co_blocks = p.Cout_pad / 32;
            return launch_v2_t<1, 6, 4>(q, lds1, st);
        }
    }
    const int na2 = (p.hp_h * p.hp_w * 4 + 511) / 512;
    const size_t lds = ssie_fprop_v2_lds_bytes(p, nt);
    if (p.th == 8) {
        if (!whole_tiles(p, 8, SSIE_TW)) return launch_v2_t<2, 5, 8, 8>(p, lds, st);
        const int epi = ssie_epi_shape(p);
        return epi == 1 ? launch_v2_t<2, 5, 8, 8, 1, false>(p, lds, st) : epi == 2 ? launch_v2_t<2, 5, 8, 8, 2, false>(p, lds, st)
                                                                                  : launch_v2_t<2, 5, 8, 8, 0, false>(p, lds, st);
    }
    if (nt == 2 && na2 <= 3 && p.ntaps == 1 && g_v2_onetap && whole_tiles(p, 16, SSIE_TW) &&
        (size_t)p.N * p.tiles_y * p.tiles_x * p.co_blocks >= (g_v2_onetap == 2 ? 1u : 512u)) {      // (2: tests - any size)
        const int epi = ssie_epi_shape(p);
        const size_t lds1 = 2 * ((size_t)p.hp_h * p.hp_w * 64 + (size_t)4 * 32 * nt * 16) + (size_t)SSIE_MAX_TAPS * 4 + 16;
        if (epi == 1) return launch_v2_t<2, 3, 8, 16, 1, false, 1>(p, lds1, st);
        if (epi == 2) return launch_v2_t<2, 3, 8, 16, 2, false, 1>(p, lds1, st);
        return launch_v2_t<2, 3, 8, 16, 0, false, 1>(p, lds1, st);
    }
    if (nt == 2) return na2 <= 3 ? launch_v2_t<2, 3, 8>(p, lds, st) : launch_v2_t<2, 5, 8>(p, lds, st);
    return na2 <= 3 ? launch_v2_t<1, 3, 8>(p, lds, st) : launch_v2_t<1, 5, 8>(p, lds, st);
}
