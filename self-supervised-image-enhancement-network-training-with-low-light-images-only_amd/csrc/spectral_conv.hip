// The 9 x 9 convolution (`shallow_conv`, /root/reference/model.py:35,52: 34 % of the loss-forward MACs at 31 bands, 70 % at 256)
// in the FREQUENCY domain: forward, data gradient and weight gradient.
//
// A direct 9 x 9 convolution spends 81 MACs per (pixel, input channel, output channel).  With 32 x 32 overlap-save tiles
// (24 x 24 valid outputs each) the same result costs one complex MAC per (frequency, tile, ci, co) = 4 real MACs per
// 1.06 valid pixels (544 half-spectrum bins per 576 valid pixels): ~20x fewer FLOPs, and fewer roundings per output - in fp32
// the spectral result is CLOSER to the fp64 reference than the direct fp32 sum of 2 511 products (2.7e-7 vs 2.4e-6 of the
// tensor's maximum, measured on the parity fixtures).  The work becomes HBM-bound streaming of spectral tensors:
//
//   spec_fft_tiles      x (N,H,W,C) -> X^[m][f][c]      32 x 32 real FFT of every tile window, half spectrum f = ky*17 + kx
//                       (every spectral tensor is TILE-major: a tile's 544 bins are one contiguous block, so the transform kernels -
//                       which own a tile - stream it, and the per-frequency GEMMs read / write whole 256- or 512-byte channel rows
//                       with f as the fastest grid index, i.e. neighbouring workgroups touch neighbouring rows)
//   spec_weights        w (Co,Ci,9,9) -> B[f][ci][co] = conj(FFT(w padded)), B'[f][co][ci] = conj(FFT(flipped w))
//   spec_gemm           Y^[f][m][n] = sum_k A[f][m][k] * B[f][k][n]   (complex, per frequency; as a real GEMM on the fp32 MFMA)
//   spec_ifft_out       Y^ -> y (N,H,W,Co): inverse transform, the 24 x 24 valid block of every tile (+ bias | accumulate)
//   spec_wgrad_reduce   dW^[s][f][ci][co] = sum_{m in slice s} conj(G^[f][m][co]) * X^[f][m][ci]
//   spec_wgrad_out      dw[co][ci][dy][dx] += (1/1024) sum_s sum_f Re(dW^ e^{+i theta}) over the 9 x 9 taps only
//
// nn.Conv2d is a cross-correlation y[p] = sum_t w[t] x[p + t - 4]; with the tile window starting 4 pixels before the tile,
// output j of a tile is sum_t w[t] xwin[j + t] = circular correlation (no wrap for j < 24) <-> X^ * conj(W^).  The data
// gradient is the same correlation with the flipped kernel on windows of the output gradient; the weight gradient is the
// correlation of the ZERO-PADDED 24 x 24 gradient tile with the input window, of which only lags 0..8 are kept (j + t <= 31:
// no wrap either).
#include "spectral_conv.h"
#include <math.h>

namespace {

constexpr int T = SSIE_SPEC_T, V = SSIE_SPEC_V, KX = SSIE_SPEC_KX, NF = SSIE_SPEC_NF;

// exp(-2 pi i j / 32), j = 0 .. 15 (forward twiddles; the inverse conjugates)
__device__ __forceinline__ float2 tw32(int j)
{
    constexpr float C[16] = {1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960218f,
                             0.38268343236508977f, 0.19509032201612825f, 0.f, -0.19509032201612825f, -0.38268343236508977f, -0.55557023301960218f,
                             -0.70710678118654752f, -0.83146961230254524f, -0.92387953251128674f, -0.98078528040323043f};
    constexpr float S[16] = {0.f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f, 0.70710678118654752f, 0.83146961230254524f,
                             0.92387953251128674f, 0.98078528040323043f, 1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
                             0.70710678118654752f, 0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f};
    return make_float2(C[j], -S[j]);
}

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// in-register radix-2 DIT FFT of 32 complex values, fully unrolled (every index and twiddle is a compile-time constant)
template <bool INVERSE>
__device__ __forceinline__ void fft32(float2 (&v)[32])
{
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int j = ((i & 1) << 4) | ((i & 2) << 2) | (i & 4) | ((i & 8) >> 2) | ((i & 16) >> 4);
        if (i < j) { const float2 t = v[i]; v[i] = v[j]; v[j] = t; }
    }
#pragma unroll
    for (int s = 1; s <= 5; ++s) {
        const int half = 1 << (s - 1), step = 32 >> s;
#pragma unroll
        for (int g = 0; g < 32; g += 2 * half)
#pragma unroll
            for (int k = 0; k < half; ++k) {
                float2 w = tw32(k * step);
                if (INVERSE) w.y = -w.y;
                const float2 t = cmulf(v[g + k + half], w);
                const float2 a = v[g + k];
                v[g + k] = make_float2(a.x + t.x, a.y + t.y);
                v[g + k + half] = make_float2(a.x - t.x, a.y - t.y);
            }
    }
}

constexpr int CG = 8;                              // channels per workgroup of the tile transforms
constexpr int RS = T + 1;                          // padded row stride of the real tile in LDS
// per-channel plane strides.  The spectral <-> global copies run lanes over the 8 channels first (a 64-byte segment of
// X^[f][m][c]): with the natural strides (544 complex = 1088 dwords, 1056 floats) all channels of a bin fall on ONE bank - PMC:
// SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE in both transform kernels.  +4 complex puts the 8 channels 8 banks apart
// (with 4 bins x 2 banks in between: conflict-free); +4 floats puts the two channel quads of the real-tile copies 16 banks apart.
constexpr int PS = T * KX + 4;                     // complex elements per channel plane of Cx
constexpr int RP = T * RS + 4;                     // floats per channel plane of R

// ---- x -> X^ ------------------------------------------------------------------------------------------------------------
// grid (Cp / 8, tiles of this tensor); window origin = (V*a + org, V*b + org); `valid` = 32 for halo windows (org = -4),
// 24 for the zero-padded gradient tiles of the weight gradient (org = 0)
__global__ __launch_bounds__(256) void spec_fft_tiles_kernel(const float* __restrict__ in, int cs, int H, int W, int tiles_y, int tiles_x,
                                                             int org, int valid, float2* __restrict__ out, int m0, int Mtot, int Cp, int ntiles, int ncg)
{
    // the real tile R and the half-complex tile Cx share one buffer (static LDS is limited to 64 KB): a row is pulled into
    // registers, and only after a barrier written back as its spectrum
    __shared__ float2 buf[CG * PS];
    float* R = (float*)buf;                        // [CG][RP] floats: rows of RS  (33.9 KB of the 35.1 KB)
    float2* Cx = buf;                              // [CG][PS] complex: rows of KX
    static_assert(CG * RP * 4 <= CG * PS * 8, "R must fit inside Cx");
    // workgroup id -> (tile, channel group), XCD-aware: consecutive ids go round-robin over the 8 XCDs (each with its own L2), so
    // the channel groups of ONE tile - which share its 128-byte cache lines - take ids 8 apart: same XCD, back to back in time
    const int tid = threadIdx.x;
    const int mloc = (blockIdx.x / (8 * ncg)) * 8 + (blockIdx.x & 7), c0 = ((blockIdx.x >> 3) % ncg) * CG;
    if (mloc >= ntiles) return;
    const int b = mloc % tiles_x, a = (mloc / tiles_x) % tiles_y, n = mloc / (tiles_x * tiles_y);
    const int oy = V * a + org, ox = V * b + org;
    // ALL eight loads of a thread first, branch-free (an address inside the tensor for the slots that are padding, the value zeroed
    // afterwards): written as `if (inside) v = load` in a rolled loop, hipcc issued one load, waited for it (vmcnt(0)) and wrote it to
    // LDS before the next - eight (here) / seventeen (inverse transform) HBM latencies in series per workgroup, which is what these
    // kernels spent their time on (round 3: 128 -> 75 us and 198 -> 115 us per launch)
    {
        constexpr int NLD = T * T * (CG / 4) / 256;
        static_assert(NLD * 256 == T * T * (CG / 4), "whole load rounds");
        f32x4 v[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 256 * i, q = idx & 1, px = idx >> 1, y = px >> 5, x = px & 31;
            const int gy = oy + y, gx = ox + x;
            const bool ok = y < valid && x < valid && gy >= 0 && gy < H && gx >= 0 && gx < W && c0 + 4 * q < cs;
            const size_t off = ok ? (((size_t)n * H + gy) * W + gx) * cs + c0 + 4 * q : 0;
            v[i] = *(const f32x4*)(in + off);
            if (!ok) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 256 * i, q = idx & 1, px = idx >> 1, y = px >> 5, x = px & 31;
#pragma unroll
            for (int j = 0; j < 4; ++j) R[(4 * q + j) * RP + y * RS + x] = v[i][j];
        }
    }
    __syncthreads();
    {   // rows: one real row of 32 per thread
        const int c = tid >> 5, y = tid & 31;
        float2 v[32];
#pragma unroll
        for (int x = 0; x < 32; ++x) v[x] = make_float2(R[c * RP + y * RS + x], 0.f);
        fft32<false>(v);
        __syncthreads();                           // every row of R is in registers: the buffer may now hold spectra
#pragma unroll
        for (int kx = 0; kx < KX; ++kx) Cx[c * PS + y * KX + kx] = v[kx];
    }
    __syncthreads();
    if (tid < CG * KX) {   // columns: one complex column of 32 per thread, in place
        const int c = tid / KX, kx = tid % KX;
        float2 v[32];
#pragma unroll
        for (int y = 0; y < 32; ++y) v[y] = Cx[c * PS + y * KX + kx];
        fft32<false>(v);
#pragma unroll
        for (int ky = 0; ky < 32; ++ky) Cx[c * PS + ky * KX + kx] = v[ky];
    }
    __syncthreads();
    const size_t m = (size_t)m0 + mloc;
    for (int idx = tid; idx < NF * CG; idx += 256) {
        const int c = idx & (CG - 1), f = idx >> 3;
        out[(m * NF + f) * Cp + c0 + c] = Cx[c * PS + f];               // f = ky * KX + kx is the in-plane offset
    }
}

// ---- Y^ -> y ------------------------------------------------------------------------------------------------------------
// grid (ceil(cs / 8), tiles); writes the 24 x 24 valid block of tile m: out = (accumulate ? out : 0) + y + bias
__global__ __launch_bounds__(256) void spec_ifft_out_kernel(const float2* __restrict__ Yf, int m0, int Mtot, int Np, int H, int W, int tiles_y, int tiles_x,
                                                            float* __restrict__ out, int cs, int Cout, const float* __restrict__ bias, int accumulate, int ntiles, int ncg)
{
    __shared__ float2 buf[CG * PS];                // Cx, then (after a barrier) the real tile R: see spec_fft_tiles_kernel
    float* R = (float*)buf;
    float2* Cx = buf;
    const int tid = threadIdx.x;           // (tile, channel group) from the workgroup id as in spec_fft_tiles_kernel
    const int mloc = (blockIdx.x / (8 * ncg)) * 8 + (blockIdx.x & 7), c0 = ((blockIdx.x >> 3) % ncg) * CG;
    if (mloc >= ntiles) return;
    const int b = mloc % tiles_x, a = (mloc / tiles_x) % tiles_y, n = mloc / (tiles_x * tiles_y);
    const size_t m = (size_t)m0 + mloc;
    {   // all 17 loads of a thread first (see spec_fft_tiles_kernel)
        constexpr int NLD = NF * CG / 256;
        static_assert(NLD * 256 == NF * CG, "whole load rounds");
        float2 v[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 256 * i, c = idx & (CG - 1), f = idx >> 3;
            v[i] = Yf[(m * NF + f) * Np + min(c0 + c, Np - 1)];
            if (c0 + c >= Np) v[i] = make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 256 * i, c = idx & (CG - 1), f = idx >> 3;
            Cx[c * PS + f] = v[i];
        }
    }
    __syncthreads();
    if (tid < CG * KX) {
        const int c = tid / KX, kx = tid % KX;
        float2 v[32];
#pragma unroll
        for (int ky = 0; ky < 32; ++ky) v[ky] = Cx[c * PS + ky * KX + kx];
        fft32<true>(v);
#pragma unroll
        for (int y = 0; y < 32; ++y) Cx[c * PS + y * KX + kx] = v[y];
    }
    __syncthreads();
    {   // rows: Hermitian half spectrum -> 32 reals (only the 24 valid rows / columns are kept)
        const int c = tid >> 5, y = tid & 31;
        float2 v[32];
#pragma unroll
        for (int kx = 0; kx < KX; ++kx) v[kx] = Cx[c * PS + y * KX + kx];
        __syncthreads();                           // all spectra are in registers: the buffer may now hold the real tile
        if (y < V) {
#pragma unroll
            for (int kx = KX; kx < 32; ++kx) v[kx] = make_float2(v[32 - kx].x, -v[32 - kx].y);
            fft32<true>(v);
#pragma unroll
            for (int x = 0; x < V; ++x) R[c * RP + y * RS + x] = v[x].x * (1.f / (T * T));
        }
    }
    __syncthreads();
    {
        constexpr int NST = (V * V * (CG / 4) + 255) / 256;            // 5 rounds (4.5 used)
        f32x4 old[NST];
        if (accumulate) {                           // the read-modify-write of the data gradient: all reads in flight before the first add
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int idx = tid + 256 * i, q = idx & 1, px = idx >> 1, y = px / V, x = px - y * V;
                const int gy = V * a + y, gx = V * b + x, c = c0 + 4 * q;
                const bool ok = idx < V * V * (CG / 4) && gy < H && gx < W && c < cs;
                old[i] = *(const f32x4*)(out + (ok ? (((size_t)n * H + gy) * W + gx) * cs + c : 0));
            }
        }
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int idx = tid + 256 * i, q = idx & 1, px = idx >> 1, y = px / V, x = px - y * V;
            const int gy = V * a + y, gx = V * b + x, c = c0 + 4 * q;
            if (idx >= V * V * (CG / 4) || gy >= H || gx >= W || c >= cs) continue;
            float* o = out + (((size_t)n * H + gy) * W + gx) * cs + c;
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (c + j < Cout) ? R[(4 * q + j) * RP + y * RS + x] + (bias ? bias[c + j] : 0.f) : 0.f;
            if (accumulate) v += old[i];
            *(f32x4*)o = v;
        }
    }
}

// ---- weights -> B (forward) and B' (data gradient) ------------------------------------------------------------------------
// Bf[f][ci][co] = conj(FFT(w padded)) = sum_t w[co][ci][t] e^{+2 pi i (ky ty + kx tx)/32};  Bd[f][co][ci] = phase(f) * conj(Bf),
// phase = e^{+2 pi i 8 (ky + kx)/32} (the flipped kernel).  Channels >= Cin / Cout are zero.
// Separable form: one thread per (ci, co, kx) reads its 81 taps ONCE, transforms the 9 rows along x for its kx (t[ty] = sum_tx w[ty][tx]
// e^{+2 pi i kx tx / 32}) and then emits the 32 ky bins (sum_ty t[ty] e^{+2 pi i ky ty / 32}): 81 loads and ~1.4 k multiply-adds
// per thread instead of 81 loads and 162 multiply-adds per BIN (the per-bin kernel read every tap 544 times: 0.48 ms at 256 bands).
// BD = false: Bf[f][ci][co], lanes along co;  BD = true: Bd[f][co][ci] = phase(f) conj(Bf), lanes along ci - both write coalesced rows.
// grid (ceil(Kp * Np / 256), KX)
template <bool BD>
__global__ __launch_bounds__(256) void spec_weights_kernel(const float* __restrict__ w, int Cout, int Cin, int Kp, int Np, float2* __restrict__ out)
{
    __shared__ float2 tw[32];                      // e^{+2 pi i j / 32}
    if (threadIdx.x < 32) { float sn, cs; sincospif((float)threadIdx.x * (1.f / 16.f), &sn, &cs); tw[threadIdx.x] = make_float2(cs, sn); }
    __syncthreads();
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, kx = blockIdx.y;
    if (idx >= Kp * Np) return;
    const int co = BD ? idx / Kp : idx % Np, ci = BD ? idx % Kp : idx / Np;
    float2 t[9];
    if (co < Cout && ci < Cin) {
        const float* wp = w + ((size_t)co * Cin + ci) * 81;
#pragma unroll
        for (int ty = 0; ty < 9; ++ty) {
            float2 a = make_float2(0.f, 0.f);
#pragma unroll
            for (int tx = 0; tx < 9; ++tx) {
                const float2 e = tw[(kx * tx) & 31];
                const float v = wp[ty * 9 + tx];
                a.x += v * e.x; a.y += v * e.y;
            }
            t[ty] = a;
        }
    } else {
#pragma unroll
        for (int ty = 0; ty < 9; ++ty) t[ty] = make_float2(0.f, 0.f);
    }
    for (int ky = 0; ky < T; ++ky) {
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int ty = 0; ty < 9; ++ty) {
            const float2 e = tw[(ky * ty) & 31];
            acc.x += t[ty].x * e.x - t[ty].y * e.y; acc.y += t[ty].x * e.y + t[ty].y * e.x;
        }
        const int f = ky * KX + kx;
        if (BD) out[((size_t)f * Np + co) * Kp + ci] = cmulf(tw[(8 * (ky + kx)) & 31], make_float2(acc.x, -acc.y));
        else out[((size_t)f * Kp + ci) * Np + co] = acc;
    }
}

// ---- per-frequency complex GEMM: C[f][m][n] = sum_k A[f][m][k] * B[f][k][n] ----------------------------------------------
// As a REAL GEMM on the fp32 MFMA, contraction index (k, part r/i): Re = Ar.Br - Ai.Bi, Im = Ar.Bi + Ai.Br.  One
// v_mfma_f32_32x32x2_f32 contracts one k: lane half h = 0 feeds the real parts, h = 1 the imaginary parts:
//   A (rows = 32 tiles m):  h ? Ai[m][k] : Ar[m][k]
//   B (cols = 32 n):        Re tile: h ? -Bi[k][n] : Br[k][n];   Im tile: h ? Br[k][n] : Bi[k][n]
// so a k step costs one 8-byte LDS read per operand and two MFMAs per wave.  Workgroup = 4 waves = (4 / NT) m-tiles x NT n-tiles
// of 32; K in chunks of 32 staged in LDS (A rows padded to 33 complex: conflict-free column reads).
// grid (NF, ceil(M / (128 / NT)), Np / (32 NT)): f is the fastest index (see the layout note at the top)
template <int NT>
__global__ __launch_bounds__(256) void spec_gemm_kernel(const float2* __restrict__ A, const float2* __restrict__ B, float2* __restrict__ C,
                                                        int M, int Kp, int Np, int Ma, int ma0, int Mc, int mc0)
{
    // A holds Ma rows per frequency and this GEMM uses rows [ma0, ma0 + M); C likewise (Mc, mc0)
    constexpr int WMT = 4 / NT, MB = 32 * WMT, NB = 32 * NT;
    __shared__ float2 As[MB * 33];                 // [m][k]
    __shared__ float2 Bs[32 * NB];                 // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, li = lane & 31;
    const int wm = wave / NT, wn = wave % NT;
    const int f = blockIdx.x, mb = blockIdx.y * MB, nb = blockIdx.z * NB;
    const float2* Af = A + ((size_t)ma0 * NF + f) * Kp;          // row m of this frequency: Af + m * NF * Kp
    const float2* Bfp = B + (size_t)f * Kp * Np + nb;
    const size_t arow = (size_t)NF * Kp, crow = (size_t)NF * Np;
    f32x16 cre, cim;
#pragma unroll
    for (int r = 0; r < 16; ++r) { cre[r] = 0.f; cim[r] = 0.f; }
    // K in chunks of 32; the NEXT chunk's operands are fetched into registers before this chunk's MFMAs and written to LDS after them
    // (one chunk at 31 bands; eight at 256, where the exposed load latency of the single-buffered loop was a third of the kernel)
    constexpr int NA = MB * 32 / 256, NBR = 32 * NB / 256;
    float2 ar[NA], br[NBR];
#define SG_FETCH(K0)                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < NA; ++i_) {                                                           \
        const int idx = i_ * 256 + tid, mm = idx >> 5, kk = idx & 31;                                             \
        ar[i_] = (mb + mm < M && (K0) + kk < Kp) ? Af[(size_t)(mb + mm) * arow + (K0) + kk] : make_float2(0.f, 0.f); \
    }                                                                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < NBR; ++i_) {                                                          \
        const int idx = i_ * 256 + tid, kk = idx / NB, nn = idx - kk * NB;                                        \
        br[i_] = ((K0) + kk < Kp) ? Bfp[(size_t)((K0) + kk) * Np + nn] : make_float2(0.f, 0.f);                   \
    }
#define SG_STORE                                                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < NA; ++i_) { const int idx = i_ * 256 + tid; As[(idx >> 5) * 33 + (idx & 31)] = ar[i_]; } \
    _Pragma("unroll") for (int i_ = 0; i_ < NBR; ++i_) Bs[i_ * 256 + tid] = br[i_];
    SG_FETCH(0)
    SG_STORE
    __syncthreads();
    for (int k0 = 0; k0 < Kp; k0 += 32) {
        const bool more = k0 + 32 < Kp;
        if (more) { SG_FETCH(k0 + 32) }
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            const float2 a = As[(wm * 32 + li) * 33 + kk], bq = Bs[kk * NB + wn * 32 + li];
            const float av = h ? a.y : a.x;
            cre = __builtin_amdgcn_mfma_f32_32x32x2f32(av, h ? -bq.y : bq.x, cre, 0, 0, 0);
            cim = __builtin_amdgcn_mfma_f32_32x32x2f32(av, h ? bq.x : bq.y, cim, 0, 0, 0);
        }
        if (more) {
            __syncthreads();                       // every wave is done reading this chunk
            SG_STORE
            __syncthreads();
        }
    }
#undef SG_FETCH
#undef SG_STORE
    // accumulator: lane (col = li = n, h), register r = row m_local = (r & 3) + 8 (r >> 2) + 4 h
    float2* Cf = C + ((size_t)mc0 * NF + f) * Np + nb + wn * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mb + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < M) Cf[(size_t)m * crow] = make_float2(cre[r], cim[r]);
    }
}

// ---- weight gradient, reduction over tiles: dW^[s][f][k][n] = sum_{m in slice s} conj(G^[f][m][n]) * X^[f][m][k] ----------
// As a REAL GEMM on the fp32 MFMA: Re = Gr.Xr + Gi.Xi, Im = Gr.Xi - Gi.Xr, i.e. the contraction index is (tile m, part r/i).
// One v_mfma_f32_32x32x2_f32 contracts exactly one tile: lane half h = 0 feeds the real parts, h = 1 the imaginary parts:
//   A (rows = 32 output channels n):   h ? Gi[m][n] : Gr[m][n]
//   B (cols = 32 input channels k):    Re tile: h ? Xi[m][k] : Xr[m][k];   Im tile: h ? -Xr[m][k] : Xi[m][k]
// so a tile costs three 8-byte loads per lane (two G halves of 32 channels, one X) and four MFMAs (2 n-tiles x {Re, Im}),
// straight from global memory (each (tile, frequency) row of G^ / X^ is one coalesced 512 / 256 byte segment).  grid (NF, slices, Kp / 32);
// the four waves of a workgroup take every fourth tile and are summed through LDS in fixed order (deterministic).
__global__ __launch_bounds__(256) void spec_wgrad_reduce_kernel(const float2* __restrict__ Xf, const float2* __restrict__ Gf, float2* __restrict__ dW,
                                                                int M, int Kp, int nslices)
{
    constexpr int Np = 64;
    __shared__ float red[3][4][16][64];            // waves 1..3 x 4 accumulators
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, li = lane & 31;
    const int f = blockIdx.x, s = blockIdx.y, k0 = blockIdx.z * 32;
    const int per = (M + nslices - 1) / nslices, mbeg = s * per, mend = min(M, mbeg + per);
    f32x16 acc[2][2];                              // [n-tile][Re / Im]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float2* Xp = Xf + (size_t)f * Kp + k0 + li;            // tile m: + m * NF * Kp
    const float2* Gp = Gf + (size_t)f * Np + li;
    const size_t xrow = (size_t)NF * Kp, grow = (size_t)NF * Np;
    constexpr int U = 8;                           // tiles per batch; the NEXT batch's 24 loads are in flight under this batch's 32 MFMAs
    float2 x[2][U], g0[2][U], g1[2][U];
#define WG_LOAD(BUF, M0)                                                                                  \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                       \
        const int mm_ = min((M0) + u, mend - 1);                                                          \
        x[BUF][u] = Xp[(size_t)mm_ * xrow]; g0[BUF][u] = Gp[(size_t)mm_ * grow]; g1[BUF][u] = Gp[(size_t)mm_ * grow + 32]; \
    }
#define WG_MFMA(BUF, M0)                                                                                  \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                       \
        if ((M0) + u < mend) {                                                                            \
            const float a0 = h ? g0[BUF][u].y : g0[BUF][u].x, a1 = h ? g1[BUF][u].y : g1[BUF][u].x;       \
            const float bre = h ? x[BUF][u].y : x[BUF][u].x, bim = h ? -x[BUF][u].x : x[BUF][u].y;        \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bre, acc[0][0], 0, 0, 0);                \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bim, acc[0][1], 0, 0, 0);                \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bre, acc[1][0], 0, 0, 0);                \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bim, acc[1][1], 0, 0, 0);                \
        }                                                                                                 \
    }
    int m = mbeg + wave * U;
    if (m < mend) WG_LOAD(0, m)
    for (; m < mend; m += 8 * U) {                 // two batches per trip so that the buffer index is a compile-time constant
        const int m1 = m + 4 * U;
        if (m1 < mend) WG_LOAD(1, m1)
        WG_MFMA(0, m)
        if (m1 < mend) {
            if (m1 + 4 * U < mend) WG_LOAD(0, m1 + 4 * U)
            WG_MFMA(1, m1)
        }
    }
#undef WG_LOAD
#undef WG_MFMA
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wave - 1][i * 2 + j][r][lane] = acc[i][j][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += red[w][i * 2 + j][r][lane];
        // accumulator layout: lane (col = li = input channel k, h), register r = row n_local = (r & 3) + 8 (r >> 2) + 4 h
        float2* o = dW + ((size_t)s * NF + f) * Kp * Np;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                o[(size_t)(k0 + li) * Np + n] = make_float2(acc[i][0][r], acc[i][1][r]);
            }
    }
}

// The same reduction for wide inputs (Kp >= 128, e.g. 256 bands): a workgroup owns ALL tiles of its (frequency, slice) and its four
// waves split the INPUT channels (KT blocks of 32 each) instead of the tiles, so G^ - which every k block needs whole - is read
// once per 128 KT input channels instead of once per 32 (at 256 bands the kernel above reads G^ eight times: 7.8 GB of counter
// traffic per launch against 3.2 GB of operands), and no cross-wave sum is needed.  grid (NF, slices, Kp / (128 KT)).
template <int KT>
__global__ __launch_bounds__(256) void spec_wgrad_reduce_wide_kernel(const float2* __restrict__ Xf, const float2* __restrict__ Gf, float2* __restrict__ dW,
                                                                     int M, int Kp, int nslices)
{
    constexpr int Np = 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, li = lane & 31;
    const int f = blockIdx.x, s = blockIdx.y, k0 = (blockIdx.z * 4 + wave) * KT * 32;
    const int per = (M + nslices - 1) / nslices, mbeg = s * per, mend = min(M, mbeg + per);
    f32x16 acc[KT][2][2];                          // [k block][n-tile][Re / Im]
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[kt][i][j][r] = 0.f;
    const float2* Xp = Xf + (size_t)f * Kp + k0 + li;            // tile m: + m * NF * Kp
    const float2* Gp = Gf + (size_t)f * Np + li;
    const size_t xrow = (size_t)NF * Kp, grow = (size_t)NF * Np;
    // every lane loads exactly the component it feeds: lane half h takes the real (h = 0) or imaginary (h = 1) part of G^ as the A
    // operand and of X^ as the Re-tile B operand, and the OTHER part of X^ (negated for h = 1) as the Im-tile B operand - 4-byte
    // loads, no per-lane select (written as h ? v.y : v.x on float2 arrays hipcc spilled the arrays to scratch to index them)
    const float* Xq = (const float*)Xp + h;
    const float* Xo = (const float*)Xp + (1 - h);
    const float* Gq = (const float*)Gp + h;
    const float sgn = h ? -1.f : 1.f;
    constexpr int U = 8;                           // tiles per batch; the next batch's loads are in flight under this batch's MFMAs
    float xr[2][U][KT], xo[2][U][KT], g0[2][U], g1[2][U];
#define WW_LOAD(BUF, M0)                                                                                  \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                       \
        const int mm_ = min((M0) + u, mend - 1);                                                          \
        g0[BUF][u] = Gq[(size_t)mm_ * grow * 2]; g1[BUF][u] = Gq[(size_t)mm_ * grow * 2 + 64];            \
        _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) {                                               \
            xr[BUF][u][kt] = Xq[(size_t)mm_ * xrow * 2 + 64 * kt]; xo[BUF][u][kt] = Xo[(size_t)mm_ * xrow * 2 + 64 * kt]; \
        }                                                                                                 \
    }
#define WW_MFMA(BUF, M0)                                                                                  \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                       \
        if ((M0) + u < mend) {                                                                            \
            const float a0 = g0[BUF][u], a1 = g1[BUF][u];                                                 \
            _Pragma("unroll") for (int kt = 0; kt < KT; ++kt) {                                           \
                const float bre = xr[BUF][u][kt], bim = sgn * xo[BUF][u][kt];                             \
                acc[kt][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bre, acc[kt][0][0], 0, 0, 0);    \
                acc[kt][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bim, acc[kt][0][1], 0, 0, 0);    \
                acc[kt][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bre, acc[kt][1][0], 0, 0, 0);    \
                acc[kt][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bim, acc[kt][1][1], 0, 0, 0);    \
            }                                                                                             \
        }                                                                                                 \
    }
    int m = mbeg;
    if (m < mend) WW_LOAD(0, m)
    for (; m < mend; m += 2 * U) {                 // two batches per trip so that the buffer index is a compile-time constant
        const int m1 = m + U;
        if (m1 < mend) WW_LOAD(1, m1)
        WW_MFMA(0, m)
        if (m1 < mend) {
            if (m1 + U < mend) WW_LOAD(0, m1 + U)
            WW_MFMA(1, m1)
        }
    }
#undef WW_LOAD
#undef WW_MFMA
    // accumulator layout: lane (col = li = input channel k, h), register r = row n_local = (r & 3) + 8 (r >> 2) + 4 h
    float2* o = dW + ((size_t)s * NF + f) * Kp * Np;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                o[(size_t)(k0 + 32 * kt + li) * Np + n] = make_float2(acc[kt][i][0][r], acc[kt][i][1][r]);
            }
}

// ---- weight gradient, inverse transform restricted to the 9 x 9 taps, in two separable stages ----------------------------
// stage 1: E[ty][kx][k][n] = sum_s sum_ky dW^[s][ky][kx][k][n] e^{+2 pi i ky ty / 32}           (9 x 17 x Kp x 64 threads)
// stage 2: dw[n][k][ty][tx] += (1/1024) sum_kx wgt(kx) Re(E[ty][kx][k][n] e^{+2 pi i kx tx / 32}), wgt = 1 for kx in {0, 16} else 2
//          (the dropped half of the spectrum is the conjugate)                                    (81 x Kp x 64 threads)
__global__ void spec_wgrad_out1_kernel(const float2* __restrict__ dW, int nslices, int Kp, float2* __restrict__ E)
{
    constexpr int Np = 64;
    __shared__ float2 tw[32];                      // e^{+2 pi i j / 32}
    if (threadIdx.x < 32) { float sn, cs; sincospif((float)threadIdx.x * (1.f / 16.f), &sn, &cs); tw[threadIdx.x] = make_float2(cs, sn); }
    __syncthreads();
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)9 * KX * Kp * Np;
    if (idx >= total) return;
    const int kn = (int)(idx % (Kp * Np)), kx = (int)((idx / (Kp * Np)) % KX), ty = (int)(idx / ((long)Kp * Np * KX));
    float2 acc = make_float2(0.f, 0.f);
    for (int ky = 0; ky < T; ++ky) {
        float2 d = make_float2(0.f, 0.f);
        for (int s = 0; s < nslices; ++s) { const float2 t = dW[((size_t)s * NF + ky * KX + kx) * Kp * Np + kn]; d.x += t.x; d.y += t.y; }
        const float2 e = cmulf(d, tw[(ky * ty) & 31]);
        acc.x += e.x; acc.y += e.y;
    }
    E[idx] = acc;
}

__global__ void spec_wgrad_out2_kernel(const float2* __restrict__ E, int Kp, int Cout, int Cin, float* __restrict__ dw)
{
    constexpr int Np = 64;
    __shared__ float2 tw[32];
    if (threadIdx.x < 32) { float sn, cs; sincospif((float)threadIdx.x * (1.f / 16.f), &sn, &cs); tw[threadIdx.x] = make_float2(cs, sn); }
    __syncthreads();
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = idx & 63, k = (idx >> 6) % Kp, tap = idx / (64 * Kp);
    if (tap >= 81 || k >= Cin || n >= Cout) return;
    const int ty = tap / 9, tx = tap % 9;
    float acc = 0.f;
    for (int kx = 0; kx < KX; ++kx) {
        const float2 e = E[((size_t)(ty * KX + kx) * Kp + k) * Np + n];
        const float2 t = tw[(kx * tx) & 31];
        acc += ((kx == 0 || kx == 16) ? 1.f : 2.f) * (e.x * t.x - e.y * t.y);
    }
    dw[((size_t)n * Cin + k) * 81 + tap] += acc * (1.f / (T * T));
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------------------
int ssie_spec_tiles(int H, int W, int* tiles_y, int* tiles_x)
{
    *tiles_y = (H + V - 1) / V; *tiles_x = (W + V - 1) / V;
    return *tiles_y * *tiles_x;
}

int ssie_launch_spec_fft(const float* in, int cs, int Cp, int N, int H, int W, int halo, float2* out, int m0, int Mtot, hipStream_t st)
{
    int ty, tx; const int per = ssie_spec_tiles(H, W, &ty, &tx);
    if (Cp % CG || cs % 4) return 91;
    const int ntiles = N * per, ncg = Cp / CG;
    hipLaunchKernelGGL(spec_fft_tiles_kernel, dim3((unsigned)((ntiles + 7) / 8 * 8 * ncg)), dim3(256), 0, st, in, cs, H, W, ty, tx, halo ? -4 : 0, halo ? T : V, out, m0, Mtot, Cp, ntiles, ncg);
    return hipGetLastError() == hipSuccess ? 0 : 92;
}

int ssie_launch_spec_ifft(const float2* Yf, int m0, int Mtot, int Np, int N, int H, int W, float* out, int cs, int Cout, const float* bias,
                          int accumulate, hipStream_t st)
{
    int ty, tx; const int per = ssie_spec_tiles(H, W, &ty, &tx);
    if (cs % 4) return 93;
    const int ntiles = N * per, ncg = (cs + CG - 1) / CG;
    hipLaunchKernelGGL(spec_ifft_out_kernel, dim3((unsigned)((ntiles + 7) / 8 * 8 * ncg)), dim3(256), 0, st, Yf, m0, Mtot, Np, H, W, ty, tx, out, cs, Cout, bias, accumulate, ntiles, ncg);
    return hipGetLastError() == hipSuccess ? 0 : 94;
}

int ssie_launch_spec_weights(const float* w, int Cout, int Cin, int Kp, int Np, float2* Bf, float2* Bd, hipStream_t st)
{
    const dim3 grid((unsigned)((Kp * Np + 255) / 256), KX);
    hipLaunchKernelGGL(spec_weights_kernel<false>, grid, dim3(256), 0, st, w, Cout, Cin, Kp, Np, Bf);
    if (Bd) hipLaunchKernelGGL(spec_weights_kernel<true>, grid, dim3(256), 0, st, w, Cout, Cin, Kp, Np, Bd);
    return hipGetLastError() == hipSuccess ? 0 : 95;
}

// Bias gradient of the 9 x 9 layer from the spectra the weight gradient already holds: the zero-padded 24 x 24 tiles of the output
// gradient partition the image, so sum_pixels g[co] = sum_tiles Re G^[tile][f = 0][co] (the DC bin of an un-normalised transform is
// the tile's sum).  Stage 1: a block sums 64 tiles (4 parts x 16 tiles per thread, fixed order); stage 2: colsum_final_kernel over
// the blocks.  Replaces a colsum pass over the whole 2N x H x W x 64 gradient tensor (268 MB, 53 us at N = 32).
__global__ __launch_bounds__(256) void spec_bias_partial_kernel(const float2* __restrict__ Gn, int M, float* __restrict__ partial)
{
    __shared__ float red[4][64];
    const int co = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 64;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = m0 + part + 4 * i;
        const float v = Gn[(size_t)min(m, M - 1) * NF * 64 + co].x;        // (unconditional: predicated loads are waited for one by one)
        s += m < M ? v : 0.f;
    }
    red[part][co] = s;
    __syncthreads();
    if (part == 0) partial[(size_t)blockIdx.x * 64 + co] = (red[0][co] + red[1][co]) + (red[2][co] + red[3][co]);
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ dst, int accumulate);   // conv_kernels.hip
int ssie_launch_spec_bias(const float2* Gn, int M, float* partial, float* db, int accumulate, hipStream_t st)
{
    const int nblk = (M + 63) / 64;
    hipLaunchKernelGGL(spec_bias_partial_kernel, dim3(nblk), dim3(256), 0, st, Gn, M, partial);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(64), dim3(256), 0, st, (const float*)partial, nblk, 64, db, accumulate);
    return hipGetLastError() == hipSuccess ? 0 : 101;
}

int ssie_launch_spec_gemm(const float2* A, int Ma, int ma0, const float2* B, float2* C, int Mc, int mc0, int M, int Kp, int Np, hipStream_t st)
{
    if (Kp % 8 || Np % 32) return 96;
    if ((M + 63) / 64 > 65535) return 97;
    if (Np % 64 == 0) hipLaunchKernelGGL(spec_gemm_kernel<2>, dim3(NF, (M + 63) / 64, Np / 64), dim3(256), 0, st, A, B, C, M, Kp, Np, Ma, ma0, Mc, mc0);
    else hipLaunchKernelGGL(spec_gemm_kernel<1>, dim3(NF, (M + 127) / 128, Np / 32), dim3(256), 0, st, A, B, C, M, Kp, Np, Ma, ma0, Mc, mc0);
    return hipGetLastError() == hipSuccess ? 0 : 98;
}

int ssie_launch_spec_wgrad(const float2* Xf, const float2* Gf, float2* dWs, int M, int Kp, int nslices, int Cout, int Cin, float* dw, hipStream_t st)
{
    if (Kp % 32) return 99;
    if (Kp % 256 == 0) hipLaunchKernelGGL(spec_wgrad_reduce_wide_kernel<2>, dim3(NF, nslices, Kp / 256), dim3(256), 0, st, Xf, Gf, dWs, M, Kp, nslices);
    else if (Kp % 128 == 0) hipLaunchKernelGGL(spec_wgrad_reduce_wide_kernel<1>, dim3(NF, nslices, Kp / 128), dim3(256), 0, st, Xf, Gf, dWs, M, Kp, nslices);
    else hipLaunchKernelGGL(spec_wgrad_reduce_kernel, dim3(NF, nslices, Kp / 32), dim3(256), 0, st, Xf, Gf, dWs, M, Kp, nslices);
    float2* E = dWs + (size_t)nslices * NF * Kp * 64;          // [9][17][Kp][64] behind the slices
    const long t1 = (long)9 * KX * Kp * 64;
    hipLaunchKernelGGL(spec_wgrad_out1_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, st, (const float2*)dWs, nslices, Kp, E);
    const int t2 = 81 * Kp * 64;
    hipLaunchKernelGGL(spec_wgrad_out2_kernel, dim3((t2 + 255) / 256), dim3(256), 0, st, (const float2*)E, Kp, Cout, Cin, dw);
    return hipGetLastError() == hipSuccess ? 0 : 100;
}
