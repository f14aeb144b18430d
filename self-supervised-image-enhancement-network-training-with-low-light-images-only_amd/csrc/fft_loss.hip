// Fourier-magnitude loss, forward + cotangent, one H x W band plane per workgroup, entirely in LDS.
//
// Replaces fourier_spectrum_loss (/root/reference/model.py:456-473) and the autograd adjoint of its
// torch.fft.fft2:  L_f = mean | |M*F(x)| - |M*F(S)| |,  dL/dS = c_f * Re(H*W*ifft2(M * g_Z)).
//   * x and S are real, so ONE complex FFT of z = x + i*S yields both spectra:
//       F(x)[k] = (Z[k] + conj(Z[-k]))/2,   F(S)[k] = (Z[k] - conj(Z[-k]))/(2i)
//   * forward = radix-2 decimation-in-frequency (natural in, bit-reversed out); the inverse is a
//     decimation-in-time pass that consumes the bit-reversed layout directly, so no reorder pass exists
//   * the radial mask is the reference's UNSHIFTED, non-Hermitian mask: bins k and
//     -k are masked independently.
// Plane rows are padded by one complex element so the column passes are bank-conflict free.
#include "loss_kernels.h"
#include <math.h>

#define FFT_THREADS 1024

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// R consecutive radix-2 stages (st0 .. st0+R-1) of `lines` independent length-n transforms, done in registers: a thread
// gathers the 2^R elements that only interact with each other during those stages, runs the R butterfly levels on them
// and writes them back - one LDS round trip and one barrier per R stages instead of per stage.  The butterflies, their
// twiddles and their order per element are exactly those of the plain radix-2 schedule (forward: decimation in frequency,
// natural in / bit-reversed out; inverse: decimation in time on the bit-reversed layout), so results are unchanged.
// Lanes run over LINES first: row passes then touch addresses (W+1)*8 B apart and column passes 8 B apart, both
// bank-conflict free, and all lanes of a wave share each twiddle (LDS broadcast).
template <bool INVERSE, int R, int NT = FFT_THREADS>
__device__ __forceinline__ void fft_block(float2* z, const float2* tw, int logM, int n, int logn, int lines, int loglines,
                                          int es, int ls, int tid, int st0)
{
    constexpr int RR = 1 << R;
    const int lo = INVERSE ? st0 : logn - st0 - R;          // lowest of the R index bits handled here
    const int items = lines * (n >> R);
    for (int id = tid; id < items; id += NT) {
        const int line = id & (lines - 1), g = id >> loglines;
        const int base = ((g >> lo) << (lo + R)) | (g & ((1 << lo) - 1));
        float2* zl = z + line * ls;
        float2 v[RR];
#pragma unroll
        for (int r = 0; r < RR; ++r) v[r] = zl[(base + (r << lo)) * es];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const int bit = INVERSE ? t : R - 1 - t;
            const int s = 1 << (lo + bit);
            const int tshift = logM - (lo + bit + 1);       // twiddle index step = M / (2 s)
#pragma unroll
            for (int r0 = 0; r0 < RR; ++r0) {
                if (r0 & (1 << bit)) continue;
                const int r1 = r0 | (1 << bit);
                const int j = (base + (r0 << lo)) & (s - 1);
                float2 w = tw[j << tshift];
                const float2 a = v[r0], b = v[r1];
                if (INVERSE) {
                    w.y = -w.y;
                    const float2 bw = cmul(b, w);
                    v[r0] = make_float2(a.x + bw.x, a.y + bw.y);
                    v[r1] = make_float2(a.x - bw.x, a.y - bw.y);
                } else {
                    v[r0] = make_float2(a.x + b.x, a.y + b.y);
                    v[r1] = cmul(make_float2(a.x - b.x, a.y - b.y), w);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RR; ++r) zl[(base + (r << lo)) * es] = v[r];
    }
    __syncthreads();
}

// all log2(n) stages of `lines` transforms (element stride es, line stride ls), four stages per LDS round trip
template <bool INVERSE, int NT = FFT_THREADS>
__device__ __forceinline__ void fft_pass(float2* z, const float2* tw, int logM, int n, int logn, int lines, int loglines,
                                         int es, int ls, int tid)
{
    int st = 0;
    for (; logn - st >= 4; st += 4) fft_block<INVERSE, 4, NT>(z, tw, logM, n, logn, lines, loglines, es, ls, tid, st);
    switch (logn - st) {
    case 3: fft_block<INVERSE, 3, NT>(z, tw, logM, n, logn, lines, loglines, es, ls, tid, st); break;
    case 2: fft_block<INVERSE, 2, NT>(z, tw, logM, n, logn, lines, loglines, es, ls, tid, st); break;
    case 1: fft_block<INVERSE, 1, NT>(z, tw, logM, n, logn, lines, loglines, es, ls, tid, st); break;
    default: break;
    }
}

// Sizes that are not powers of two: plain O(n^2) DFT of every line, in place.  One wave owns a whole line: lane j (+64q)
// accumulates output bin j over all inputs (every lane reads the same input element - an LDS broadcast), and the wave writes
// the line back only after it has consumed it, so no second buffer is needed.  tw[t] = exp(-2 pi i t / n), n entries; the
// twiddle index (j*k) mod n is advanced incrementally.  ~25x the work of the radix-2 path, still far from dominating a step.
template <bool INVERSE>
__device__ __forceinline__ void dft_pass(float2* z, const float2* tw, int n, int lines, int es, int ls, int tid)
{
    constexpr int Q = 3;                              // outputs per lane: n <= 192
    const int lane = tid & 63, wave = tid >> 6;
    for (int line = wave; line < lines; line += FFT_THREADS / 64) {
        float2* zl = z + line * ls;
        float2 acc[Q];
        int idx[Q], jj[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { acc[q] = make_float2(0.f, 0.f); idx[q] = 0; jj[q] = (lane + 64 * q) % n; }
        for (int k = 0; k < n; ++k) {
            const float2 v = zl[k * es];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                float2 w = tw[idx[q]];
                if (INVERSE) w.y = -w.y;
                const float2 t = cmul(v, w);
                acc[q].x += t.x; acc[q].y += t.y;
                idx[q] += jj[q]; if (idx[q] >= n) idx[q] -= n;
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (lane + 64 * q < n) zl[(lane + 64 * q) * es] = acc[q];
    }
    __syncthreads();
}

__global__ __launch_bounds__(FFT_THREADS) void fft_loss_kernel(const FftParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    // XCD-aware plane mapping: blocks b and b+8 share an XCD (round-robin dispatch), so give all B band
    // planes of one patch to one XCD: their strided 4-byte reads then share the same L2 lines.
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int n = xcd + 8 * (j / p.B), c = j % p.B;
    if (n >= p.N) { if (threadIdx.x == 0) p.partials[b] = 0.f; return; }
    const int H = p.H, W = p.W, LS = W + 1, tid = threadIdx.x;
    const int M = H > W ? H : W;
    const bool pow2 = p.logH >= 0;               // radix-2 path; otherwise the direct DFT with per-axis twiddle tables
    float2* z = (float2*)smem_f;                 // [H][W+1]
    float2* tw = z + H * LS;                     // pow2: [M/2] exp(-2 pi i t / M);  generic: [W] for rows, then [H] for columns
    float2* twc = tw + W;
    float* red = (float*)(tw + (pow2 ? (M >> 1) : (H + W)));        // [FFT_THREADS/64]

    if (pow2) {
        for (int t = tid; t < (M >> 1); t += FFT_THREADS) {
            float sn, cs; sincospif(-2.0f * (float)t / (float)M, &sn, &cs);
            tw[t] = make_float2(cs, sn);
        }
    } else {
        for (int t = tid; t < W; t += FFT_THREADS) { float sn, cs; sincospif(-2.0f * (float)t / (float)W, &sn, &cs); tw[t] = make_float2(cs, sn); }
        for (int t = tid; t < H; t += FFT_THREADS) { float sn, cs; sincospif(-2.0f * (float)t / (float)H, &sn, &cs); twc[t] = make_float2(cs, sn); }
    }
    const size_t base = (size_t)n * H * W;
    for (int id = tid; id < H * W; id += FFT_THREADS) {
        const int h = id / W, w = id - h * W;
        z[h * LS + w] = make_float2(p.x[(base + id) * p.x_cs + c], p.S[(base + id) * p.s_cs + c]);
    }
    __syncthreads();
    const int logM = p.logH > p.logW ? p.logH : p.logW;
    if (pow2) {
        fft_pass<false>(z, tw, logM, W, p.logW, H, p.logH, 1, LS, tid);      // rows
        fft_pass<false>(z, tw, logM, H, p.logH, W, p.logW, LS, 1, tid);      // columns
    } else {
        dft_pass<false>(z, tw, W, H, 1, LS, tid);
        dft_pass<false>(z, twc, H, W, LS, 1, tid);
    }

    // pointwise: loss and g_Z per conjugate pair {k, -k}
    float lsum = 0.f;
    for (int id = tid; id < H * W; id += FFT_THREADS) {
        const int ky = id / W, kx = id - ky * W;
        const int qy = ky ? H - ky : 0, qx = kx ? W - kx : 0;
        const int idm = qy * W + qx;
        if (id > idm) continue;
        // position of bin (ky, kx) in the plane: bit-reversed after the radix-2 DIF passes, natural after the direct DFT
        const int pk = pow2 ? (int)(__brev((unsigned)ky) >> (32 - p.logH)) * LS + (int)(__brev((unsigned)kx) >> (32 - p.logW)) : ky * LS + kx;
        const int pm = pow2 ? (int)(__brev((unsigned)qy) >> (32 - p.logH)) * LS + (int)(__brev((unsigned)qx) >> (32 - p.logW)) : qy * LS + qx;
        const float2 Zk = z[pk], Zm = z[pm];
        const float2 Fx = make_float2(0.5f * (Zk.x + Zm.x), 0.5f * (Zk.y - Zm.y));
        const float2 Fs = make_float2(0.5f * (Zk.y + Zm.y), -0.5f * (Zk.x - Zm.x));
        const float ax = sqrtf(Fx.x * Fx.x + Fx.y * Fx.y), as = sqrtf(Fs.x * Fs.x + Fs.y * Fs.y);
        const float diff = ax - as;
        const float coef = as > 0.f ? -((float)(diff > 0.f) - (float)(diff < 0.f)) * p.scale_g / as : 0.f;
        const bool mk = p.mask[id] != 0, mm = p.mask[idm] != 0;
        float2 Gk = make_float2(0.f, 0.f), Gm = make_float2(0.f, 0.f);
        if (mk) { lsum += fabsf(diff); Gk = make_float2(coef * Fs.x, coef * Fs.y); }
        if (id != idm && mm) { lsum += fabsf(diff); Gm = make_float2(coef * Fs.x, -coef * Fs.y); }
        z[pk] = Gk;
        if (id != idm) z[pm] = Gm;
    }
    __syncthreads();
    if (pow2) {
        fft_pass<true>(z, tw, logM, H, p.logH, W, p.logW, LS, 1, tid);       // columns (bit-reversed in, natural out)
        fft_pass<true>(z, tw, logM, W, p.logW, H, p.logH, 1, LS, tid);       // rows
    } else {
        dft_pass<true>(z, twc, H, W, LS, 1, tid);
        dft_pass<true>(z, tw, W, H, 1, LS, tid);
    }
    for (int id = tid; id < H * W; id += FFT_THREADS) {
        const int h = id / W, w = id - h * W;
        p.gS[(base + id) * p.s_cs + c] += z[h * LS + w].x;
    }
    // block reduction of the loss
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
    if ((tid & 63) == 0) red[tid >> 6] = lsum;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int q = 0; q < FFT_THREADS / 64; ++q) s += red[q];
        p.partials[b] = s * p.inv_n0;
    }
}

// ---------------------------------------------------------------------------------------------
// Three passes over a half-spectrum workspace, every pass a batch of independent 1-D transforms in LDS.  Two users:
//   * planes that do not fit the LDS (training patches above 128 x 128, model.py:456-473 accepts any patch_size)
//   * BAND-GROUPED rows (fft_rows_*_grouped_kernel, power-of-two W, planes of 64 x 64 and more): the tensors are NHWC, so a
//     workgroup that owns ONE band plane reads 4 bytes per 128-byte line - measured, that gather (3.7 cycles per element and CU)
//     is what the whole-plane kernel above spends its time on, not the transform.  Passes A and C therefore take BG = 16
//     neighbouring bands of R image rows at once: 64 contiguous bytes per pixel in, 64 contiguous bytes per pixel out.
//   A  rows:    Z = FFT_W(x + i S) per row, split into the row spectra X^[h][kx], S^[h][kx] of the two REAL inputs,
//               kx = 0..W/2 only (the other half is the conjugate), stored row-major  ws[plane][a][h][kx]  (pass A writes runs of
//               W/2+1 bins, pass B reads CB = 16 neighbouring columns of a row as one 128-byte run)
//   B  columns: FFT_H of both -> F(x)[ky][kx], F(S)[ky][kx]; loss + g_Z per bin, weighted by M[k] + M[-k] on the interior
//               columns (the mirror bin (-ky, W-kx) lives in the dropped half and has the same magnitudes; Re() of the
//               adjoint makes its contribution the conjugate's - SURVEY §2.1 "cannot simply double"); inverse FFT_H
//               in place -> G^[kx][h]
//   C  rows:    gS[h][w] += Re sum_{kx <= W/2} G^[kx][h] e^{+2 pi i kx w / W}  (complex inverse with the upper half zero)
// Power-of-two lengths use the radix-2 passes above, other lengths the direct DFT (out of place, any length).
// ---------------------------------------------------------------------------------------------
template <bool INVERSE, int NT = FFT_THREADS>
__device__ __forceinline__ void dft_lines(const float2* src, float2* dst, const float2* tw, int n, int lines, int es, int ls, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    for (int line = wave; line < lines; line += NT / 64) {
        const float2* sl = src + line * ls; float2* dl = dst + line * ls;
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane, jj = j % n;
            float2 acc = make_float2(0.f, 0.f);
            int idx = 0;
            for (int k = 0; k < n; ++k) {
                const float2 v = sl[k * es];
                float2 w = tw[idx];
                if (INVERSE) w.y = -w.y;
                const float2 t = cmul(v, w);
                acc.x += t.x; acc.y += t.y;
                idx += jj; if (idx >= n) idx -= n;
            }
            if (j < n) dl[j * es] = acc;
        }
    }
    __syncthreads();
}

template <int NT = FFT_THREADS>
__device__ __forceinline__ void fill_twiddles(float2* tw, int n, int count, int tid)
{
    for (int t = tid; t < count; t += NT) { float sn, cs; sincospif(-2.0f * (float)t / (float)n, &sn, &cs); tw[t] = make_float2(cs, sn); }
}

__device__ __forceinline__ int brev_n(int v, int logn) { return (int)(__brev((unsigned)v) >> (32 - logn)); }

struct FftBigGeom { int R, logR, CB, logCB, WH, plane0, nplanes, row_tiles, col_groups;
                    int BG, logBG, groups, patch0, npatches; };        // band-grouped passes A / C: BG bands per workgroup, whole patches per chunk

// pass A.  grid = nplanes * row_tiles
__global__ __launch_bounds__(FFT_THREADS) void fft_rows_fwd_kernel(const FftParams p, const FftBigGeom g)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int tid = threadIdx.x, H = p.H, W = p.W, LS = W + 1, R = g.R;
    const int pl = blockIdx.x / g.row_tiles, rt = blockIdx.x % g.row_tiles;
    const int plane = g.plane0 + pl, n = plane / p.B, c = plane % p.B, h0 = rt * R;
    const bool pow2 = p.logW >= 0;
    float2* z = (float2*)smem_f;                       // [R][W+1]
    float2* z2 = z + R * LS;                           // direct DFT only: output buffer
    float2* tw = pow2 ? z2 : z2 + R * LS;
    fill_twiddles(tw, W, pow2 ? (W >> 1) : W, tid);
    const size_t base = (size_t)n * H * W;
    for (int id = tid; id < R * W; id += FFT_THREADS) {
        const int r = id / W, w = id - r * W, h = h0 + r;
        float2 v = make_float2(0.f, 0.f);
        if (h < H) { const size_t px = base + (size_t)h * W + w; v = make_float2(p.x[px * p.x_cs + c], p.S[px * p.s_cs + c]); }
        z[r * LS + w] = v;
    }
    __syncthreads();
    const float2* zo = z;
    if (pow2) fft_pass<false>(z, tw, p.logW, W, p.logW, R, g.logR, 1, LS, tid);
    else { dft_lines<false>(z, z2, tw, W, R, 1, LS, tid); zo = z2; }
    float2* wsx = (float2*)p.ws + (size_t)pl * 2 * g.WH * H;
    float2* wss = wsx + (size_t)g.WH * H;
    for (int id = tid; id < R * g.WH; id += FFT_THREADS) {
        const int r = id / g.WH, kx = id - r * g.WH, h = h0 + r;           // lanes along kx: contiguous stores
        if (h >= H) continue;
        const int qx = kx ? W - kx : 0;
        const float2 Zk = zo[r * LS + (pow2 ? brev_n(kx, p.logW) : kx)], Zm = zo[r * LS + (pow2 ? brev_n(qx, p.logW) : qx)];
        wsx[(size_t)h * g.WH + kx] = make_float2(0.5f * (Zk.x + Zm.x), 0.5f * (Zk.y - Zm.y));
        wss[(size_t)h * g.WH + kx] = make_float2(0.5f * (Zk.y + Zm.y), -0.5f * (Zk.x - Zm.x));
    }
}

// pass A, band-grouped (power-of-two W).  grid = npatches * groups * row_tiles; LDS lines = BG bands x R rows.  512 threads: 64
// lines of 128 points are 512 four-stage items; the 66 KB footprint lets two workgroups share a CU
#define FFT_ROWS_THREADS 512
template <int BG>
__global__ __launch_bounds__(FFT_ROWS_THREADS) void fft_rows_fwd_grouped_kernel(const FftParams p, const FftBigGeom g)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int Q = BG / 4, NT = FFT_ROWS_THREADS;                          // float4s per pixel of the band group
    const int tid = threadIdx.x, H = p.H, W = p.W, LS = W + 1, R = g.R;
    const int per = g.groups * g.row_tiles;
    const int pn = blockIdx.x / per, rem = blockIdx.x - pn * per, cg = rem / g.row_tiles, rt = rem - cg * g.row_tiles;
    const int n = g.patch0 + pn, c0 = cg * BG, h0 = rt * R;
    float2* z = (float2*)smem_f;                       // [BG * R][W+1], line = band * R + row
    float2* tw = z + BG * R * LS;
    fill_twiddles<NT>(tw, W, W >> 1, tid);
    const size_t base = (size_t)n * H * W;
    for (int id = tid; id < R * W * Q; id += NT) {
        const int q = id % Q, px = id / Q, r = px >> p.logW, w = px & (W - 1), h = h0 + r, c = c0 + 4 * q;
        f32x4 xv = {0.f, 0.f, 0.f, 0.f}, sv = xv;
        if (h < H && c < p.x_cs) {                     // x_cs is a multiple of 4: the whole float4 is inside the pixel's band vector
            const size_t pix = base + (size_t)h * W + w;
            xv = *(const f32x4*)(p.x + pix * p.x_cs + c); sv = *(const f32x4*)(p.S + pix * p.s_cs + c);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            z[((4 * q + j) * R + r) * LS + w] = (c + j < p.B) ? make_float2(xv[j], sv[j]) : make_float2(0.f, 0.f);
    }
    __syncthreads();
    fft_pass<false, NT>(z, tw, p.logW, W, p.logW, BG * R, g.logBG + g.logR, 1, LS, tid);
    const size_t pf = (size_t)2 * g.WH * H;            // complex elements per plane in the workspace
    for (int id = tid; id < BG * R * g.WH; id += NT) {
        const int line = id / g.WH, kx = id - line * g.WH, b = line >> g.logR, r = line & (R - 1), h = h0 + r;
        if (h >= H || c0 + b >= p.B) continue;
        float2* wsx = (float2*)p.ws + ((size_t)pn * p.B + c0 + b) * pf;
        float2* wss = wsx + (size_t)g.WH * H;
        const int qx = kx ? W - kx : 0;
        const float2 Zk = z[line * LS + brev_n(kx, p.logW)], Zm = z[line * LS + brev_n(qx, p.logW)];
        wsx[(size_t)h * g.WH + kx] = make_float2(0.5f * (Zk.x + Zm.x), 0.5f * (Zk.y - Zm.y));
        wss[(size_t)h * g.WH + kx] = make_float2(0.5f * (Zk.y + Zm.y), -0.5f * (Zk.x - Zm.x));
    }
}

// pass B.  grid = nplanes * col_groups.  256 threads: a workgroup transforms 2 * CB = 32 columns of 128 points at a time, i.e. 256
// four-stage items - with 1024 threads seven of eight sat out every stage block (102 -> see DESIGN.md for the measured effect);
// the LDS footprint (33 KB) lets four such workgroups share a CU
#define FFT_COLS_THREADS 256
__global__ __launch_bounds__(FFT_COLS_THREADS) void fft_cols_kernel(const FftParams p, const FftBigGeom g)
{
    constexpr int NT = FFT_COLS_THREADS;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int tid = threadIdx.x, H = p.H, W = p.W, LS = H + 1, CB = g.CB;
    const int pl = blockIdx.x / g.col_groups, cg = blockIdx.x % g.col_groups, kx0 = cg * CB;
    const bool pow2 = p.logH >= 0;
    float2* a = (float2*)smem_f;                       // [CB][H+1]  X^ -> F(x)
    float2* b = a + CB * LS;                           // [CB][H+1]  S^ -> F(S) -> g_Z -> G^
    float2* a2 = b + CB * LS; float2* b2 = a2 + CB * LS;      // direct DFT only
    float2* tw = pow2 ? a2 : b2 + CB * LS;
    float* red = (float*)(tw + (pow2 ? (H >> 1) : H));
    fill_twiddles<NT>(tw, H, pow2 ? (H >> 1) : H, tid);
    float2* wsx = (float2*)p.ws + (size_t)pl * 2 * g.WH * H;
    float2* wss = wsx + (size_t)g.WH * H;
    // (branch-free bodies in the three global-memory loops of this kernel: with `if (inside) v = load` hipcc waits for every
    // predicated load before it issues the next one - vmcnt(0) per iteration - and a workgroup pays its HBM / L2 latencies in series)
#pragma unroll 4
    for (int id = tid; id < CB * H; id += NT) {
        const int h = id >> g.logCB, col = id & (CB - 1), kx = kx0 + col;      // lanes along the CB columns: one 128-byte run per row
        const size_t o = (size_t)h * g.WH + min(kx, g.WH - 1);
        float2 va = wsx[o], vb = wss[o];
        if (kx >= g.WH) { va = make_float2(0.f, 0.f); vb = va; }
        a[col * LS + h] = va; b[col * LS + h] = vb;
    }
    __syncthreads();
    float2* fa = a; float2* fb = b;
    if (pow2) fft_pass<false, NT>(a, tw, p.logH, H, p.logH, 2 * CB, g.logCB + 1, 1, LS, tid);      // a and b are adjacent: 2 CB lines in one go
    else { dft_lines<false, NT>(a, a2, tw, H, CB, 1, LS, tid); dft_lines<false, NT>(b, b2, tw, H, CB, 1, LS, tid); fa = a2; fb = b2; }
    float lsum = 0.f;
#pragma unroll 4
    for (int id = tid; id < CB * H; id += NT) {
        const int col = id / H, ky = id - col * H, kx = kx0 + col, kxc = min(kx, g.WH - 1);
        const int pos = col * LS + (pow2 ? brev_n(ky, p.logH) : ky);
        const bool mir = kxc != 0 && 2 * kxc != W;                              // the mirror bin lives in the dropped half
        const unsigned char m1 = p.mask[ky * W + kxc], m2 = p.mask[mir ? (ky ? H - ky : 0) * W + (W - kxc) : ky * W + kxc];
        const float2 Fx = fa[pos], Fs = fb[pos];
        const float ax = sqrtf(Fx.x * Fx.x + Fx.y * Fx.y), as = sqrtf(Fs.x * Fs.x + Fs.y * Fs.y);
        const float diff = ax - as;
        const float wgt = kx < g.WH ? (m1 ? 1.f : 0.f) + (mir && m2 ? 1.f : 0.f) : 0.f;
        lsum += wgt * fabsf(diff);
        const float coef = as > 0.f ? -((float)(diff > 0.f) - (float)(diff < 0.f)) * p.scale_g * wgt / as : 0.f;
        fb[pos] = make_float2(coef * Fs.x, coef * Fs.y);
    }
    __syncthreads();
    const float2* go = fb;
    if (pow2) fft_pass<true, NT>(fb, tw, p.logH, H, p.logH, CB, g.logCB, 1, LS, tid);
    else { dft_lines<true, NT>(fb, b, tw, H, CB, 1, LS, tid); go = b; }
    for (int id = tid; id < CB * H; id += NT) {
        const int h = id >> g.logCB, col = id & (CB - 1), kx = kx0 + col;
        if (kx < g.WH) wss[(size_t)h * g.WH + kx] = go[col * LS + h];
    }
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
    if ((tid & 63) == 0) red[tid >> 6] = lsum;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int q = 0; q < NT / 64; ++q) s += red[q];
        p.partials[(size_t)(g.plane0 + pl) * g.col_groups + cg] = s * p.inv_n0;
    }
}

// pass C.  grid = nplanes * row_tiles
__global__ __launch_bounds__(FFT_THREADS) void fft_rows_inv_kernel(const FftParams p, const FftBigGeom g)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int tid = threadIdx.x, H = p.H, W = p.W, LS = W + 1, R = g.R;
    const int pl = blockIdx.x / g.row_tiles, rt = blockIdx.x % g.row_tiles;
    const int plane = g.plane0 + pl, n = plane / p.B, c = plane % p.B, h0 = rt * R;
    const bool pow2 = p.logW >= 0;
    float2* z = (float2*)smem_f;
    float2* z2 = z + R * LS;
    float2* tw = pow2 ? z2 : z2 + R * LS;
    fill_twiddles(tw, W, pow2 ? (W >> 1) : W, tid);
    const float2* wss = (const float2*)p.ws + (size_t)pl * 2 * g.WH * H + (size_t)g.WH * H;
    for (int id = tid; id < R * W; id += FFT_THREADS) {
        const int r = id / W, kx = id - r * W, h = h0 + r;
        float2 v = make_float2(0.f, 0.f);
        if (kx < g.WH && h < H) v = wss[(size_t)h * g.WH + kx];
        z[r * LS + (pow2 ? brev_n(kx, p.logW) : kx)] = v;
    }
    __syncthreads();
    const float2* zo = z;
    if (pow2) fft_pass<true>(z, tw, p.logW, W, p.logW, R, g.logR, 1, LS, tid);
    else { dft_lines<true>(z, z2, tw, W, R, 1, LS, tid); zo = z2; }
    const size_t base = (size_t)n * H * W;
    for (int id = tid; id < R * W; id += FFT_THREADS) {
        const int r = id / W, w = id - r * W, h = h0 + r;
        if (h < H) p.gS[(base + (size_t)h * W + w) * p.s_cs + c] += zo[r * LS + w].x;
    }
}

// pass C, band-grouped (power-of-two W).  grid = npatches * groups * row_tiles
template <int BG>
__global__ __launch_bounds__(FFT_ROWS_THREADS) void fft_rows_inv_grouped_kernel(const FftParams p, const FftBigGeom g)
{
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int Q = BG / 4, NT = FFT_ROWS_THREADS;
    const int tid = threadIdx.x, H = p.H, W = p.W, LS = W + 1, R = g.R;
    const int per = g.groups * g.row_tiles;
    const int pn = blockIdx.x / per, rem = blockIdx.x - pn * per, cg = rem / g.row_tiles, rt = rem - cg * g.row_tiles;
    const int n = g.patch0 + pn, c0 = cg * BG, h0 = rt * R;
    float2* z = (float2*)smem_f;
    float2* tw = z + BG * R * LS;
    fill_twiddles<NT>(tw, W, W >> 1, tid);
    const size_t pf = (size_t)2 * g.WH * H;
#pragma unroll 4
    for (int id = tid; id < BG * R * W; id += NT) {
        const int line = id >> p.logW, kx = id & (W - 1), b = line >> g.logR, r = line & (R - 1), h = h0 + r;
        const bool ok = kx < g.WH && h < H && c0 + b < p.B;
        float2 v = ((const float2*)p.ws + ((size_t)pn * p.B + min(c0 + b, p.B - 1)) * pf + (size_t)g.WH * H)[(size_t)min(h, H - 1) * g.WH + min(kx, g.WH - 1)];
        if (!ok) v = make_float2(0.f, 0.f);
        z[line * LS + brev_n(kx, p.logW)] = v;
    }
    __syncthreads();
    fft_pass<true, NT>(z, tw, p.logW, W, p.logW, BG * R, g.logBG + g.logR, 1, LS, tid);
    const size_t base = (size_t)n * H * W;
    // read-modify-write of gS, four elements per thread at a time with all four reads in flight before the first add (a rolled
    // load -> add -> store loop pays one HBM latency per iteration)
    for (int id0 = tid; id0 < R * W * Q; id0 += 4 * NT) {
        f32x4 gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = id0 + u * NT, q = id % Q, px = id / Q, r = px >> p.logW, w = px & (W - 1), h = h0 + r, c = c0 + 4 * q;
            const bool ok = id < R * W * Q && h < H && c < p.s_cs && c < p.B;
            gv[u] = *(const f32x4*)(p.gS + (ok ? (base + (size_t)h * W + w) * p.s_cs + c : 0));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int id = id0 + u * NT, q = id % Q, px = id / Q, r = px >> p.logW, w = px & (W - 1), h = h0 + r, c = c0 + 4 * q;
            if (id >= R * W * Q || h >= H || c >= p.s_cs || c >= p.B) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (c + j < p.B) gv[u][j] += z[((4 * q + j) * R + r) * LS + w].x;
            *(f32x4*)(p.gS + (base + (size_t)h * W + w) * p.s_cs + c) = gv[u];
        }
    }
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

static bool is_pow2(int v) { return v >= 2 && !(v & (v - 1)); }

// 1: radix-2, plane in LDS (power-of-two H and W up to 128 x 128); 2: direct DFT, plane in LDS (any H, W <= 192 that fits);
// 3: three-pass path through an HBM workspace (everything else up to 4096 per side); 0: unsupported
int ssie_fft_supported(int H, int W)
{
    if (H < 2 || W < 2) return 0;
    const bool p2 = is_pow2(H) && is_pow2(W);
    size_t lds = (size_t)H * (W + 1) * 8 + (p2 ? (size_t)(H > W ? H : W) * 4 : (size_t)(H + W) * 8) + 64;
    if (lds <= 160 * 1024 && (p2 || (H <= 192 && W <= 192))) return p2 ? 1 : 2;
    return (H <= 4096 && W <= 4096) ? 3 : 0;
}

int ssie_fft_grouped = 1;             // include/ssie_debug.h: 0 = planes that fit the LDS always run the whole-plane kernel
extern "C" void ssie_debug_set_fft_grouped(int v) { ssie_fft_grouped = v; }
static bool grouped_geom(int N, int B, int H, int W, int* BGo, int* Ro);

// the path a (N, B, H, W) problem takes: ssie_fft_supported's answer, except that planes which would fit the LDS go through the
// three-pass path with band-grouped rows when that applies (power-of-two W, at least 64 x 64 - smaller planes are launch-bound
// either way - and a whole patch per workspace chunk)
int ssie_fft_path(int N, int B, int H, int W)
{
    const int kind = ssie_fft_supported(H, W);
    if ((kind == 1 || kind == 2) && ssie_fft_grouped && (long)H * W >= 4096 && grouped_geom(N, B, H, W, nullptr, nullptr)) return 3;
    return kind;
}

void ssie_fft_set_logs(FftParams& p)
{
    const int kind = ssie_fft_path(p.N, p.B, p.H, p.W);
    p.path = kind;
    if (kind == 1) { p.logH = ilog2(p.H); p.logW = ilog2(p.W); }
    else if (kind == 3) { p.logH = is_pow2(p.H) ? ilog2(p.H) : -1; p.logW = is_pow2(p.W) ? ilog2(p.W) : -1; }   // per axis
    else { p.logH = -1; p.logW = -1; }
}

int ssie_fft_grid(int N, int B) { return 8 * ((N + 7) / 8) * B; }

namespace {
const size_t kBigLdsBudget = 96 * 1024;
size_t kBigChunkBytes = 192u << 20;            // workspace of one plane chunk: stays (mostly) in the 256 MiB Infinity Cache between passes; measured at 256 bands: 96 MB 34.99 ms, 192 MB 34.74, 400 MB 34.85 per step (ssie_debug_set_fft_chunk_mb)
int big_R(int W) { const bool p2 = is_pow2(W); int R = 16; while (R > 1 && (size_t)R * (W + 1) * 8 * (p2 ? 1 : 2) > kBigLdsBudget) R >>= 1; return R; }
int big_CB(int H) { const bool p2 = is_pow2(H); int C = 16; while (C > 1 && (size_t)C * (H + 1) * 8 * (p2 ? 2 : 4) > kBigLdsBudget) C >>= 1; return C; }
size_t big_plane_floats(int H, int W) { return (size_t)2 * (W / 2 + 1) * H * 2; }
int big_chunk_planes(int N, int B, int H, int W)
{
    long c = (long)(kBigChunkBytes / (big_plane_floats(H, W) * 4)); if (c < 1) c = 1;
    if (c > (long)N * B) c = (long)N * B;
    return (int)c;
}
}

// number of loss partial sums the Fourier kernels write (and the finalize kernel reads)
extern "C" void ssie_debug_set_fft_chunk_mb(int mb) { kBigChunkBytes = (size_t)(mb < 1 ? 1 : mb) << 20; }   // plans created afterwards

int ssie_fft_partials(int N, int B, int H, int W)
{
    if (ssie_fft_path(N, B, H, W) == 3) return N * B * ssie_ceil_div(W / 2 + 1, big_CB(H));
    return ssie_fft_grid(N, B);
}

size_t ssie_fft_workspace_floats(int N, int B, int H, int W)
{
    if (ssie_fft_path(N, B, H, W) != 3) return 0;
    return (size_t)big_chunk_planes(N, B, H, W) * big_plane_floats(H, W);
}

// band-grouped passes A / C: BG = 16 bands (8 / 4 for narrow cubes) x R rows per workgroup within a 66 KB LDS budget (two
// workgroups per CU); needs a power-of-two W and a workspace chunk that holds all B planes of at least one patch
static bool grouped_geom(int N, int B, int H, int W, int* BGo, int* Ro)
{
    if (!is_pow2(W) || B < 2) return false;
    if (big_chunk_planes(N, B, H, W) < B) return false;
    const int BG = B > 8 ? 16 : B > 4 ? 8 : 4;
    int R = 16;
    while (R > 1 && ((size_t)BG * R * (W + 1) * 8 > 66 * 1024 || R > H)) R >>= 1;
    if ((size_t)BG * R * (W + 1) * 8 + (size_t)(W / 2) * 8 + 64 > 150 * 1024) return false;
    if (BGo) *BGo = BG; if (Ro) *Ro = R;
    return true;
}

static void allow_big_lds(const void* fn)
{
    static bool done[16 * 12] = {false};          // per (device, kernel): hipFuncSetAttribute is per device
    static const void* fns[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int dev = 0; hipGetDevice(&dev);
    int slot = 0; for (; slot < 12 && fns[slot] && fns[slot] != fn; ++slot) {}
    if (slot == 12 || dev < 0 || dev >= 16) { hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); return; }
    fns[slot] = fn;
    if (!done[dev * 12 + slot]) { hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done[dev * 12 + slot] = true; }
}

static int launch_fft_big(const FftParams& p, hipStream_t st)
{
    if (!p.ws) return 53;
    FftBigGeom g;
    g.R = big_R(p.W); g.logR = ilog2(g.R); g.CB = big_CB(p.H); g.logCB = ilog2(g.CB); g.WH = p.W / 2 + 1;
    g.row_tiles = ssie_ceil_div(p.H, g.R); g.col_groups = ssie_ceil_div(g.WH, g.CB);
    const bool p2w = p.logW >= 0, p2h = p.logH >= 0;
    const size_t ldsA = (size_t)g.R * (p.W + 1) * 8 * (p2w ? 1 : 2) + (size_t)(p2w ? p.W / 2 : p.W) * 8 + 64;
    const size_t ldsB = (size_t)g.CB * (p.H + 1) * 8 * (p2h ? 2 : 4) + (size_t)(p2h ? p.H / 2 : p.H) * 8 + 64 + FFT_THREADS / 64 * 4;
    const dim3 colsT(FFT_COLS_THREADS);
    if (ldsA > 160 * 1024 || ldsB > 160 * 1024) return 54;
    allow_big_lds((const void*)fft_rows_fwd_kernel); allow_big_lds((const void*)fft_cols_kernel); allow_big_lds((const void*)fft_rows_inv_kernel);
    const int total = p.N * p.B, chunk = big_chunk_planes(p.N, p.B, p.H, p.W);
    int BG = 0, Rg = 0;
    // (float4 accesses along the band axis: 16-byte aligned tensors with channel strides that are multiples of 4; anything else
    // takes the one-plane-per-workgroup passes below - same workspace, same partial sums)
    const bool al4 = p.x_cs % 4 == 0 && p.s_cs % 4 == 0 && (((uintptr_t)p.x | (uintptr_t)p.S | (uintptr_t)p.gS) & 15) == 0;
    if (ssie_fft_grouped && p2w && al4 && grouped_geom(p.N, p.B, p.H, p.W, &BG, &Rg)) {
        // band-grouped rows: chunks of whole patches
        g.BG = BG; g.logBG = ilog2(BG); g.groups = ssie_ceil_div(p.B, BG);
        g.R = Rg; g.logR = ilog2(Rg); g.row_tiles = ssie_ceil_div(p.H, Rg);
        const size_t ldsG = (size_t)BG * Rg * (p.W + 1) * 8 + (size_t)(p.W / 2) * 8 + 64;
        const int pchunk = chunk / p.B;
#define FG_LAUNCH(K, BGV) do { allow_big_lds((const void*)K<BGV>);                                                          \
        hipLaunchKernelGGL(K<BGV>, dim3(g.npatches * g.groups * g.row_tiles), dim3(FFT_ROWS_THREADS), ldsG, st, p, g); } while (0)
        for (int n0 = 0; n0 < p.N; n0 += pchunk) {
            g.patch0 = n0; g.npatches = p.N - n0 < pchunk ? p.N - n0 : pchunk;
            g.plane0 = n0 * p.B; g.nplanes = g.npatches * p.B;
            if (BG == 16) FG_LAUNCH(fft_rows_fwd_grouped_kernel, 16); else if (BG == 8) FG_LAUNCH(fft_rows_fwd_grouped_kernel, 8); else FG_LAUNCH(fft_rows_fwd_grouped_kernel, 4);
            hipLaunchKernelGGL(fft_cols_kernel, dim3(g.nplanes * g.col_groups), colsT, ldsB, st, p, g);
            if (BG == 16) FG_LAUNCH(fft_rows_inv_grouped_kernel, 16); else if (BG == 8) FG_LAUNCH(fft_rows_inv_grouped_kernel, 8); else FG_LAUNCH(fft_rows_inv_grouped_kernel, 4);
        }
#undef FG_LAUNCH
        return hipGetLastError() == hipSuccess ? 0 : 55;
    }
    for (int p0 = 0; p0 < total; p0 += chunk) {
        g.plane0 = p0; g.nplanes = total - p0 < chunk ? total - p0 : chunk;
        hipLaunchKernelGGL(fft_rows_fwd_kernel, dim3(g.nplanes * g.row_tiles), dim3(FFT_THREADS), ldsA, st, p, g);
        hipLaunchKernelGGL(fft_cols_kernel, dim3(g.nplanes * g.col_groups), colsT, ldsB, st, p, g);
        hipLaunchKernelGGL(fft_rows_inv_kernel, dim3(g.nplanes * g.row_tiles), dim3(FFT_THREADS), ldsA, st, p, g);
    }
    return hipGetLastError() == hipSuccess ? 0 : 55;
}

int ssie_launch_fft_loss(const FftParams& p, hipStream_t st)
{
    const int kind = p.path;                           // decided by ssie_fft_set_logs (the workspace and partial-sum sizes follow it)
    if (!kind || !ssie_fft_supported(p.H, p.W)) return 51;
    // the geometry below is recomputed from the CURRENT development switches: it must still fit the allocation
    if (ssie_fft_path(p.N, p.B, p.H, p.W) != kind || ssie_fft_workspace_floats(p.N, p.B, p.H, p.W) > p.ws_floats ||
        ssie_fft_partials(p.N, p.B, p.H, p.W) > p.npartials) return 56;
    if (kind == 3) return launch_fft_big(p, st);
    if ((kind == 1) != (p.logH >= 0)) return 51;
    const int M = p.H > p.W ? p.H : p.W;
    size_t lds = (size_t)p.H * (p.W + 1) * 8 + (kind == 1 ? (size_t)(M / 2) * 8 : (size_t)(p.H + p.W) * 8) + 64;
    const int vblocks = ssie_fft_grid(p.N, p.B);
    allow_big_lds((const void*)fft_loss_kernel);
    hipLaunchKernelGGL(fft_loss_kernel, dim3(vblocks), dim3(FFT_THREADS), lds, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 52;
}

// The reference's mask (model.py:460-464) in float32, exactly as torch builds it:
// torch.linspace(-1, 1, n) (float32, symmetric halves), sqrt(X^2 + Y^2) >= float32(cutoff)
static void linspace_f32(int n, float* out)
{
    const float start = -1.f, end = 1.f;
    const float step = (end - start) / (float)(n - 1);
    const int halfway = n / 2;
    for (int i = 0; i < n; ++i)
        out[i] = i < halfway ? start + step * (float)i : end - step * (float)(n - i - 1);
}

void ssie_fourier_mask_host(int H, int W, float cutoff, uint8_t* out)
{
    float* ys = new float[H]; float* xs = new float[W];
    linspace_f32(H, ys); linspace_f32(W, xs);
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            volatile float xx = xs[j] * xs[j]; volatile float yy = ys[i] * ys[i];
            volatile float r2 = xx + yy;
            out[i * W + j] = sqrtf(r2) >= cutoff ? 1 : 0;
        }
    delete[] ys; delete[] xs;
}

extern "C" int ssie_fourier_mask(int H, int W, float cutoff, uint8_t* out_host)
{
    if (!out_host || H < 2 || W < 2) return 1;
    ssie_fourier_mask_host(H, W, cutoff, out_host);
    return 0;
}
