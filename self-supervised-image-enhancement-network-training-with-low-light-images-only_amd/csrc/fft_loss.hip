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

#define FFT_THREADS 512

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// one radix-2 pass over `lines` independent lines of length n (element stride es, line stride ls)
template <bool INVERSE>
__device__ __forceinline__ void fft_pass(float2* z, const float2* tw, int twM, int n, int logn, int lines, int es, int ls, int tid)
{
    const int half = n >> 1;
    for (int st = 0; st < logn; ++st) {
        const int s = INVERSE ? (1 << st) : (half >> st);
        const int tstep = twM / (2 * s);
        for (int id = tid; id < lines * half; id += FFT_THREADS) {
            const int line = id / half, k = id - line * half;
            const int blk = k / s, j = k - blk * s;
            float2* p0 = z + line * ls + (blk * 2 * s + j) * es;
            float2* p1 = p0 + s * es;
            float2 w = tw[j * tstep];
            float2 a = *p0, b = *p1;
            if (INVERSE) {
                w.y = -w.y;
                b = cmul(b, w);
                *p0 = make_float2(a.x + b.x, a.y + b.y);
                *p1 = make_float2(a.x - b.x, a.y - b.y);
            } else {
                *p0 = make_float2(a.x + b.x, a.y + b.y);
                *p1 = cmul(make_float2(a.x - b.x, a.y - b.y), w);
            }
        }
        __syncthreads();
    }
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
    float2* z = (float2*)smem_f;                 // [H][W+1]
    float2* tw = z + H * LS;                     // [M/2]  exp(-2 pi i t / M)
    float* red = (float*)(tw + (M >> 1));        // [FFT_THREADS/64]

    for (int t = tid; t < (M >> 1); t += FFT_THREADS) {
        float sn, cs; sincospif(-2.0f * (float)t / (float)M, &sn, &cs);
        tw[t] = make_float2(cs, sn);
    }
    const size_t base = (size_t)n * H * W;
    for (int id = tid; id < H * W; id += FFT_THREADS) {
        const int h = id / W, w = id - h * W;
        z[h * LS + w] = make_float2(p.x[(base + id) * p.x_cs + c], p.S[(base + id) * p.s_cs + c]);
    }
    __syncthreads();
    fft_pass<false>(z, tw, M, W, p.logW, H, 1, LS, tid);      // rows
    fft_pass<false>(z, tw, M, H, p.logH, W, LS, 1, tid);      // columns

    // pointwise: loss and g_Z per conjugate pair {k, -k}
    float lsum = 0.f;
    for (int id = tid; id < H * W; id += FFT_THREADS) {
        const int ky = id / W, kx = id - ky * W;
        const int qy = (H - ky) & (H - 1), qx = (W - kx) & (W - 1);
        const int idm = qy * W + qx;
        if (id > idm) continue;
        const int pk = (int)(__brev((unsigned)ky) >> (32 - p.logH)) * LS + (int)(__brev((unsigned)kx) >> (32 - p.logW));
        const int pm = (int)(__brev((unsigned)qy) >> (32 - p.logH)) * LS + (int)(__brev((unsigned)qx) >> (32 - p.logW));
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
    fft_pass<true>(z, tw, M, H, p.logH, W, LS, 1, tid);       // columns (bit-reversed in, natural out)
    fft_pass<true>(z, tw, M, W, p.logW, H, 1, LS, tid);       // rows
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

int ssie_fft_supported(int H, int W)
{
    if (H < 2 || W < 2 || (H & (H - 1)) || (W & (W - 1))) return 0;
    size_t lds = (size_t)H * (W + 1) * 8 + (size_t)(H > W ? H : W) * 4 + 64;
    return lds <= 160 * 1024;
}

int ssie_fft_grid(int N, int B) { return 8 * ((N + 7) / 8) * B; }

int ssie_launch_fft_loss(const FftParams& p, hipStream_t st)
{
    if (!ssie_fft_supported(p.H, p.W)) return 51;
    const int M = p.H > p.W ? p.H : p.W;
    size_t lds = (size_t)p.H * (p.W + 1) * 8 + (size_t)(M / 2) * 8 + 64;
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
