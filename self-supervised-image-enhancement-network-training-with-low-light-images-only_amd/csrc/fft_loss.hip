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
template <bool INVERSE, int R>
__device__ __forceinline__ void fft_block(float2* z, const float2* tw, int logM, int n, int logn, int lines, int loglines,
                                          int es, int ls, int tid, int st0)
{
    constexpr int RR = 1 << R;
    const int lo = INVERSE ? st0 : logn - st0 - R;          // lowest of the R index bits handled here
    const int items = lines * (n >> R);
    for (int id = tid; id < items; id += FFT_THREADS) {
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
template <bool INVERSE>
__device__ __forceinline__ void fft_pass(float2* z, const float2* tw, int logM, int n, int logn, int lines, int loglines,
                                         int es, int ls, int tid)
{
    int st = 0;
    for (; logn - st >= 4; st += 4) fft_block<INVERSE, 4>(z, tw, logM, n, logn, lines, loglines, es, ls, tid, st);
    switch (logn - st) {
    case 3: fft_block<INVERSE, 3>(z, tw, logM, n, logn, lines, loglines, es, ls, tid, st); break;
    case 2: fft_block<INVERSE, 2>(z, tw, logM, n, logn, lines, loglines, es, ls, tid, st); break;
    case 1: fft_block<INVERSE, 1>(z, tw, logM, n, logn, lines, loglines, es, ls, tid, st); break;
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

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

static bool is_pow2(int v) { return v >= 2 && !(v & (v - 1)); }

// 1: radix-2 path (power-of-two H and W), 2: direct-DFT path (any H, W <= 192), 0: the plane does not fit the 160 KiB LDS
int ssie_fft_supported(int H, int W)
{
    if (H < 2 || W < 2) return 0;
    const bool p2 = is_pow2(H) && is_pow2(W);
    if (!p2 && (H > 192 || W > 192)) return 0;
    size_t lds = (size_t)H * (W + 1) * 8 + (p2 ? (size_t)(H > W ? H : W) * 4 : (size_t)(H + W) * 8) + 64;
    return lds <= 160 * 1024 ? (p2 ? 1 : 2) : 0;
}

int ssie_fft_grid(int N, int B) { return 8 * ((N + 7) / 8) * B; }

int ssie_launch_fft_loss(const FftParams& p, hipStream_t st)
{
    const int kind = ssie_fft_supported(p.H, p.W);
    if (!kind || (kind == 1) != (p.logH >= 0)) return 51;
    const int M = p.H > p.W ? p.H : p.W;
    size_t lds = (size_t)p.H * (p.W + 1) * 8 + (kind == 1 ? (size_t)(M / 2) * 8 : (size_t)(p.H + p.W) * 8) + 64;
    static bool set = false;
    if (!set) { hipFuncSetAttribute((const void*)fft_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    hipLaunchKernelGGL(fft_loss_kernel, dim3(ssie_fft_grid(p.N, p.B)), dim3(FFT_THREADS), lds, st, p);
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
