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
#include "ssie_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int ssie_swz(int hp) { return (hp >> 2) & 3; }

struct SrcSel {
    const float* ptr; int C, cstride, coff, Hs, Ws; float sy, sx; int cbeg;
};

// select the source holding virtual channel c_first (uniform); explicit selects keep the
// by-value kernarg struct out of scratch memory.
__device__ __forceinline__ SrcSel ssie_pick_src(const SrcDesc* src, int nsrc, int c_first)
{
    SrcSel r;
    int s = 0, cbeg = 0;
    if (nsrc > 1 && c_first >= src[0].C) { s = 1; cbeg = src[0].C; }
    if (nsrc > 2 && c_first >= src[0].C + src[1].C) { s = 2; cbeg = src[0].C + src[1].C; }
    r.ptr = s == 0 ? src[0].ptr : (s == 1 ? src[1].ptr : src[2].ptr);
    r.C = s == 0 ? src[0].C : (s == 1 ? src[1].C : src[2].C);
    r.cstride = s == 0 ? src[0].cstride : (s == 1 ? src[1].cstride : src[2].cstride);
    r.coff = s == 0 ? src[0].coff : (s == 1 ? src[1].coff : src[2].coff);
    r.Hs = s == 0 ? src[0].Hs : (s == 1 ? src[1].Hs : src[2].Hs);
    r.Ws = s == 0 ? src[0].Ws : (s == 1 ? src[1].Ws : src[2].Ws);
    r.sy = s == 0 ? src[0].sy : (s == 1 ? src[1].sy : src[2].sy);
    r.sx = s == 0 ? src[0].sx : (s == 1 ? src[1].sx : src[2].sx);
    r.cbeg = cbeg;
    return r;
}

// load 4 consecutive channels of virtual pixel (n, vy, vx); zero outside the image / channel range
__device__ __forceinline__ f32x4 ssie_load_virtual(const SrcSel& s, int n, int vy, int vx, int Hv, int Wv, int c)
{
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (vy >= 0 && vy < Hv && vx >= 0 && vx < Wv && c < s.C) {
        int y = min((int)floorf((float)vy * s.sy), s.Hs - 1);
        int x = min((int)floorf((float)vx * s.sx), s.Ws - 1);
        v = *(const f32x4*)(s.ptr + ((size_t)(n * s.Hs + y) * s.Ws + x) * s.cstride + s.coff + c);
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// fprop / dgrad
// ---------------------------------------------------------------------------------------------
template <int NT>   // output-channel tile = 32*NT
__global__ __launch_bounds__(256) void conv_fprop_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int BN = 32 * NT;
    constexpr int MT = (NT == 1) ? 1 : 2;
    const int HP = p.hp_h * p.hp_w;
    f32x4* As = (f32x4*)smem_f;                // [HP][4] float4 (16-B slot XOR-swizzled)
    f32x4* Bs = As + HP * 4;                   // [TG*4][BN] float4
    int* tapoff = (int*)(Bs + SSIE_TG * 4 * BN);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, li = lane & 31;
    int bid = blockIdx.x;
    const int cob = bid % p.co_blocks; bid /= p.co_blocks;
    const int tx = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int n = bid / p.tiles_y;
    const int a0 = ty * SSIE_TH, b0 = tx * SSIE_TW;
    const int co0 = cob * BN;
    const int wn = (NT == 1) ? 0 : (wave & 1);
    const int wm = (NT == 1) ? wave : (wave >> 1);

    for (int t = tid; t < p.ntaps; t += 256)
        tapoff[t] = ((int)p.tap_dy[t] - p.min_dy) * p.hp_w + ((int)p.tap_dx[t] - p.min_dx);

    int pixbase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        int mt = wm * MT + m;
        pixbase[m] = ((2 * mt + (li >> 4)) * p.si) * p.hp_w + (li & 15) * p.si;
    }
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    const int ngroups = (p.ntaps + SSIE_TG - 1) / SSIE_TG;
    const int vy0 = a0 * p.si + p.min_dy, vx0 = b0 * p.si + p.min_dx;

    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const SrcSel s = ssie_pick_src(p.src, p.nsrc, chunk * SSIE_CK);
        for (int g = 0; g < ngroups; ++g) {
            __syncthreads();     // everyone finished reading the previous B (and A when g == 0)
            if (g == 0) {
                for (int id = tid; id < HP * 4; id += 256) {
                    int pix = id >> 2, j = id & 3;
                    int hy = pix / p.hp_w, hx = pix - hy * p.hp_w;
                    f32x4 v = ssie_load_virtual(s, n, vy0 + hy, vx0 + hx, p.Hv, p.Wv,
                                                chunk * SSIE_CK + 4 * j - s.cbeg);
                    As[pix * 4 + (j ^ ssie_swz(pix))] = v;
                }
            }
            const int t0 = g * SSIE_TG;
            const int tg = min(SSIE_TG, p.ntaps - t0);
            {
                const f32x4* wsrc = (const f32x4*)p.wpacked
                    + ((size_t)(chunk * p.ntaps + t0) * 4) * p.Cout_pad + co0;
                for (int id = tid; id < tg * 4 * BN; id += 256) {
                    int row = id / BN, col = id % BN;
                    Bs[id] = wsrc[(size_t)row * p.Cout_pad + col];
                }
            }
            __syncthreads();
            for (int tl = 0; tl < tg; ++tl) {
                const int off = tapoff[t0 + tl];
#pragma unroll
                for (int kq = 0; kq < 2; ++kq) {
                    const f32x4 b = Bs[(tl * 4 + kq * 2 + h) * BN + wn * 32 + li];
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int hp = pixbase[m] + off;
                        const f32x4 a = As[hp * 4 + ((kq * 2 + h) ^ ssie_swz(hp))];
                        acc[m] = MFMA32(a.x, b.x, acc[m]);
                        acc[m] = MFMA32(a.y, b.y, acc[m]);
                        acc[m] = MFMA32(a.z, b.z, acc[m]);
                        acc[m] = MFMA32(a.w, b.w, acc[m]);
                    }
                }
            }
        }
    }

    // epilogue: C/D layout col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (position)
    const int co = co0 + wn * 32 + li;
    if (co >= p.Cout) return;
    const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mt = wm * MT + m;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int a = a0 + 2 * mt + (i >> 4), b = b0 + (i & 15);
            if (a >= p.Ho || b >= p.Wo) continue;
            const int oy = a * p.so + p.py, ox = b * p.so + p.px;
            if (oy >= p.Hout || ox >= p.Wout) continue;
            const size_t o = ((size_t)(n * p.Hout + oy) * p.Wout + ox) * p.out_cstride + p.out_coff + co;
            float v = acc[m][r] + bv;
            if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
            if (p.mask_mode == MASK_RELU) v = p.mask_y[o] > 0.f ? v : 0.f;
            else if (p.mask_mode == MASK_SIGMOID) { float y = p.mask_y[o]; v *= y * (1.f - y); }
            if (p.out2) p.out2[o] = v;
            if (p.addsrc) v += p.addsrc[o];
            if (p.accumulate) v += p.out[o];
            p.out[o] = v;
        }
    }
}

template __global__ void conv_fprop_kernel<1>(const ConvParams);
template __global__ void conv_fprop_kernel<2>(const ConvParams);

// ---------------------------------------------------------------------------------------------
// wgrad: dW[tap][ci][co] = sum_positions X[pos*si + tap][ci] * G[pos][co]
// GEMM view: M = ci (A operand), N = co (B operand), K = positions (2 per MFMA).
// Each workgroup owns a (ci block, co block, tap group) and a slice of the position tiles; it keeps
// up to 9 taps x 32x32 accumulators per wave in registers and writes ONE partial slab at the end.
// ---------------------------------------------------------------------------------------------
template <int CIB, int COB, int NU>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int MI = CIB / 32, NI = COB / 32, NPAIR = MI * NI, WSPLIT = 4 / NPAIR;
    const int HP = p.hp_h * p.hp_w;
    const int PT = p.th * SSIE_TW;
    float* Xs = smem_f;                 // [HP][CIB]
    float* Gs = Xs + HP * CIB;          // [PT][COB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, li = lane & 31;
    const int pair = wave % NPAIR, wsub = wave / NPAIR;
    const int mi = pair / NI, ni = pair % NI;
    const int slice = blockIdx.x;
    const int cib = blockIdx.y / p.co_blocks, cob = blockIdx.y % p.co_blocks;
    const int t0 = blockIdx.z * SSIE_TG;
    const int tg = min(SSIE_TG, p.ntaps - t0);
    const int ci0 = cib * CIB, co0 = cob * COB;

    int toff[NU];
    bool tval[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        int tl = wsub * NU + u;
        tval[u] = tl < tg;
        int t = t0 + (tval[u] ? tl : 0);
        toff[u] = (((int)p.tap_dy[t] - p.min_dy) * p.hp_w + ((int)p.tap_dx[t] - p.min_dx)) * CIB + mi * 32 + li;
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
    const bool do_bias = p.bias_slabs != nullptr && cib == 0 && blockIdx.z == 0;
    constexpr int BROWS = 256 / COB;
    const int bcol = tid % COB, brow = tid / COB;
    float bsum = 0.f;

    for (int tile = tile_beg; tile < tile_end; ++tile) {
        int tt = tile;
        const int tx = tt % p.tiles_x; tt /= p.tiles_x;
        const int ty = tt % p.tiles_y;
        const int n = tt / p.tiles_y;
        const int a0 = ty * p.th, b0 = tx * SSIE_TW;
        const int vy0 = a0 * p.si + p.min_dy, vx0 = b0 * p.si + p.min_dx;
        __syncthreads();
        for (int id = tid; id < HP * CI4; id += 256) {
            int pix = id / CI4, j = id % CI4;
            int hy = pix / p.hp_w, hx = pix - hy * p.hp_w;
            f32x4 v = ssie_load_virtual(s, n, vy0 + hy, vx0 + hx, p.Hv, p.Wv, ci0 + 4 * j);
            *(f32x4*)(Xs + pix * CIB + 4 * j) = v;
        }
        for (int id = tid; id < PT * CO4; id += 256) {
            int pix = id / CO4, j = id % CO4;
            int a = a0 + pix / SSIE_TW, b = b0 + pix % SSIE_TW;
            int c = co0 + 4 * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (a < p.Ho && b < p.Wo && c < p.Cout) {
                const float* gp = p.g + ((size_t)(n * p.Ho + a) * p.Wo + b) * p.g_cstride + p.g_coff + c;
                if (c + 3 < p.Cout) v = *(const f32x4*)gp;
                else { v.x = gp[0]; if (c + 1 < p.Cout) v.y = gp[1]; if (c + 2 < p.Cout) v.z = gp[2]; }
            }
            *(f32x4*)(Gs + pix * COB + 4 * j) = v;
        }
        __syncthreads();
        if (do_bias)
            for (int px = brow; px < PT; px += BROWS) bsum += Gs[px * COB + bcol];
        const int npair = PT / 2;
        for (int kp = 0; kp < npair; ++kp) {
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
    // partial slab [slice][tap][ci_pad][co_pad]; row (M) = ci, col (N) = co
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        if (!tval[u]) continue;
        const int t = t0 + wsub * NU + u;
        float* dst = p.slabs + (((size_t)slice * p.ntaps + t) * p.ci_pad + ci0 + mi * 32) * p.co_pad + co0 + ni * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            dst[(size_t)i * p.co_pad] = acc[u][r];
        }
    }
}

#define INST_WGRAD(CI, CO, NU) template __global__ void conv_wgrad_kernel<CI, CO, NU>(const WgradParams);
INST_WGRAD(64, 64, 9) INST_WGRAD(64, 64, 1)
INST_WGRAD(32, 64, 5) INST_WGRAD(32, 64, 1)
INST_WGRAD(64, 32, 5) INST_WGRAD(64, 32, 1)
INST_WGRAD(32, 32, 3) INST_WGRAD(32, 32, 1)

// dst[co*s_co + ci*s_ci + t*s_t] (+)= sum_slices slab[slice][t][ci][co]; the trailing Cout outputs are the fused
// bias gradient db[co] (+)= sum_slices bias_slab[slice][co].  64 outputs x 4 slice groups per block, fixed order
// (deterministic, bit-reproducible across runs and ranks).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int nslices, int ntaps, int ci_pad, int co_pad,
                                    int Cin, int Cout, float* __restrict__ dst, long s_co, long s_ci, long s_t,
                                    const float* __restrict__ bias_slabs, float* __restrict__ db, int accumulate)
{
    __shared__ float red[4][64];
    const int lo = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const long idx = (long)blockIdx.x * 64 + lo;
    const long total = (long)ntaps * Cin * Cout;
    const long total_b = total + (bias_slabs ? Cout : 0);
    float sum = 0.f;
    float* d = nullptr;
    if (idx < total) {
        const int co = (int)(idx % Cout);
        const int ci = (int)((idx / Cout) % Cin);
        const int t = (int)(idx / ((long)Cout * Cin));
        const size_t slab_sz = (size_t)ntaps * ci_pad * co_pad;
        const float* sp = slabs + ((size_t)t * ci_pad + ci) * co_pad + co;
        for (int s = sg; s < nslices; s += 4) sum += sp[(size_t)s * slab_sz];
        d = dst + co * s_co + ci * s_ci + t * s_t;
    } else if (idx < total_b) {
        const int co = (int)(idx - total);
        for (int s = sg; s < nslices; s += 4) sum += bias_slabs[(size_t)s * co_pad + co];
        d = db + co;
    }
    red[sg][lo] = sum;
    __syncthreads();
    if (sg == 0 && d) {
        const float tot = (red[0][lo] + red[1][lo]) + (red[2][lo] + red[3][lo]);
        *d = accumulate ? (*d + tot) : tot;
    }
}

// per-channel sums of G over all pixels (bias gradient), two-stage & deterministic
__global__ void colsum_partial_kernel(const float* __restrict__ g, long npix, int cstride, int coff, int C,
                                      float* __restrict__ partial /*[gridDim.x][C]*/)
{
    __shared__ float red[256];
    const int tid = threadIdx.x;
    int cw = 1; while (cw < C) cw <<= 1; if (cw > 256) cw = 256;
    const int lanes = 256 / cw;
    const int pl = tid / cw, cl = tid % cw;
    const long per = (npix + gridDim.x - 1) / gridDim.x;
    const long beg = (long)blockIdx.x * per, end = min(beg + per, npix);
    for (int c0 = 0; c0 < C; c0 += cw) {
        const int c = c0 + cl;
        float sum = 0.f;
        if (c < C)
            for (long px = beg + pl; px < end; px += lanes) sum += g[px * cstride + coff + c];
        red[tid] = sum;
        __syncthreads();
        if (pl == 0 && c < C) {
            float tot = 0.f;
            for (int q = 0; q < lanes; ++q) tot += red[q * cw + cl];
            partial[(size_t)blockIdx.x * C + c] = tot;
        }
        __syncthreads();
    }
}

__global__ void colsum_final_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ dst, int accumulate)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sum = 0.f;
    for (int b = 0; b < nblk; ++b) sum += partial[(size_t)b * C + c];
    dst[c] = accumulate ? dst[c] + sum : sum;
}

// ---------------------------------------------------------------------------------------------
// weight packing: dst[chunk][t][q][n][s] = W[base + n*s_n + k*s_k + tapsel[t]*s_t], k = chunk*16 + (q>>1)*8 + (q&1)*4 + s
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ void ssie_pack_one(const PackDesc& d, long idx4)
{
    long total4 = (long)d.nchunks * d.T * 4 * d.Npad;
    if (idx4 >= total4) return;
    int n = (int)(idx4 % d.Npad); long r = idx4 / d.Npad;
    int q = (int)(r & 3); r >>= 2;
    int t = (int)(r % d.T); int chunk = (int)(r / d.T);
    f32x4 v;
    const int kb = chunk * 16 + (q >> 1) * 8 + (q & 1) * 4;
    const int ts = (int)d.tapsel[t] * d.s_t;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        int k = kb + s;
        v[s] = (k < d.K && n < d.N) ? d.w[(long)n * d.s_n + (long)k * d.s_k + ts] : 0.f;
    }
    ((f32x4*)d.dst)[idx4] = v;
}

__global__ void pack_weights_kernel(const PackDesc d)
{
    ssie_pack_one(d, (long)blockIdx.x * blockDim.x + threadIdx.x);
}

// batched: descriptors resident in device memory, blockIdx.y selects the descriptor
__global__ void pack_weights_batched_kernel(const PackDesc* __restrict__ descs)
{
    const PackDesc& d = descs[blockIdx.y];
    long total4 = (long)d.nchunks * d.T * 4 * d.Npad;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (long)gridDim.x * blockDim.x)
        ssie_pack_one(d, idx);
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
extern "C" size_t ssie_fprop_lds_bytes(const ConvParams* p, int nt)
{
    return (size_t)p->hp_h * p->hp_w * 64 + (size_t)SSIE_TG * 4 * 32 * nt * 16 + (size_t)p->ntaps * 4;
}

int ssie_launch_fprop(const ConvParams& p, hipStream_t st)
{
    const int nt = (p.Cout_pad % 64 == 0) ? 2 : 1;
    if (p.Cout_pad % 32) return 11;
    if (p.ntaps < 1 || p.ntaps > SSIE_MAX_TAPS) return 12;
    size_t lds = ssie_fprop_lds_bytes(&p, nt);
    if (lds > 160 * 1024) return 13;
    dim3 grid((unsigned)((size_t)p.N * p.tiles_y * p.tiles_x * p.co_blocks));
    if (nt == 2) {
        static bool set2 = false;
        if (!set2) { hipFuncSetAttribute((const void*)conv_fprop_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set2 = true; }
        hipLaunchKernelGGL(conv_fprop_kernel<2>, grid, dim3(256), lds, st, p);
    } else {
        static bool set1 = false;
        if (!set1) { hipFuncSetAttribute((const void*)conv_fprop_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set1 = true; }
        hipLaunchKernelGGL(conv_fprop_kernel<1>, grid, dim3(256), lds, st, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : 14;
}

template <int CI, int CO, int NU>
static int launch_wgrad_t(const WgradParams& p, hipStream_t st)
{
    size_t lds = ((size_t)p.hp_h * p.hp_w * CI + (size_t)p.th * SSIE_TW * CO) * 4;
    if (lds > 160 * 1024) return 23;
    static bool set = false;
    if (!set) { hipFuncSetAttribute((const void*)conv_wgrad_kernel<CI, CO, NU>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    dim3 grid(p.nslices, p.ci_blocks * p.co_blocks, p.tap_groups);
    hipLaunchKernelGGL((conv_wgrad_kernel<CI, CO, NU>), grid, dim3(256), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 24;
}

// cib/cob chosen by the caller through ci_pad/ci_blocks (ci_pad = ci_blocks*CIB)
int ssie_launch_wgrad(const WgradParams& p, hipStream_t st)
{
    const int cib = p.ci_pad / p.ci_blocks, cob = p.co_pad / p.co_blocks;
    const bool one = p.ntaps == 1;
    if (cib == 64 && cob == 64) return one ? launch_wgrad_t<64, 64, 1>(p, st) : launch_wgrad_t<64, 64, 9>(p, st);
    if (cib == 32 && cob == 64) return one ? launch_wgrad_t<32, 64, 1>(p, st) : launch_wgrad_t<32, 64, 5>(p, st);
    if (cib == 64 && cob == 32) return one ? launch_wgrad_t<64, 32, 1>(p, st) : launch_wgrad_t<64, 32, 5>(p, st);
    if (cib == 32 && cob == 32) return one ? launch_wgrad_t<32, 32, 1>(p, st) : launch_wgrad_t<32, 32, 3>(p, st);
    return 21;
}

int ssie_launch_wgrad_reduce(const float* slabs, int nslices, int ntaps, int ci_pad, int co_pad, int Cin, int Cout,
                             float* dst, long s_co, long s_ci, long s_t, const float* bias_slabs, float* db,
                             int accumulate, hipStream_t st)
{
    long total = (long)ntaps * Cin * Cout + (bias_slabs ? Cout : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st,
                       slabs, nslices, ntaps, ci_pad, co_pad, Cin, Cout, dst, s_co, s_ci, s_t, bias_slabs, db, accumulate);
    return hipGetLastError() == hipSuccess ? 0 : 25;
}

int ssie_launch_colsum(const float* g, long npix, int cstride, int coff, int C, float* partial, int nblk,
                       float* dst, int accumulate, hipStream_t st)
{
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, st, g, npix, cstride, coff, C, partial);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(64), 0, st, partial, nblk, C, dst, accumulate);
    return hipGetLastError() == hipSuccess ? 0 : 26;
}

int ssie_launch_pack(const PackDesc& d, hipStream_t st)
{
    long total4 = (long)d.nchunks * d.T * 4 * d.Npad;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, d);
    return hipGetLastError() == hipSuccess ? 0 : 27;
}

int ssie_launch_pack_batched(const PackDesc* descs_dev, int ndesc, hipStream_t st)
{
    hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(64, ndesc), dim3(256), 0, st, descs_dev);
    return hipGetLastError() == hipSuccess ? 0 : 28;
}
