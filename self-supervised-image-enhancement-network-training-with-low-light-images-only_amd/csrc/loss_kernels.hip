// Fused self-supervised loss forward + hand-derived cotangents (HBM-bound kernels).
//
// Replaces compute_loss's loss section and its autograd backward, /root/reference/model.py:445-454
// (smooth_loss), :475-481 (spectral_smoothness_loss), :491-542 (structure_aware_loss), :551-564.
// The Fourier term (:456-473) lives in fft_loss.hip.  Derivation: SURVEY.md §2.2, restated and proven
// equal to autograd in oracle/loss_cotangents.py.
//
// Layout: every tensor is NHWC, so the 32 lanes of a half-wave read one pixel's band vector as one
// coalesced 128-byte segment; channel reductions (mean_c |dR|, sum_c exp(..), sum_c s*R) are wavefront
// shuffles inside the half-wave; the spatial 5-point stencil is served by L1/L2.
#include "loss_kernels.h"

__device__ __forceinline__ float sgnf(float v) { return (float)(v > 0.f) - (float)(v < 0.f); }

__device__ __forceinline__ float half_sum(float v)   // sum over the 32 lanes of this half-wave
{
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}

__global__ __launch_bounds__(256) void loss_direct_kernel(const LossParams p)
{
    const int tid = threadIdx.x, lane = tid & 31, grp = tid >> 5;
    const long npix = (long)p.N * p.H * p.W;
    const int B = p.B;
    float acc_rec = 0.f, acc_rf = 0.f, acc_il = 0.f, acc_id = 0.f, acc_sp = 0.f;

    for (long pix = (long)blockIdx.x * 8 + grp; pix < npix; pix += (long)gridDim.x * 8) {
        const int w = (int)(pix % p.W);
        const int h = (int)((pix / p.W) % p.H);
        const bool hasR = w + 1 < p.W, hasL = w > 0, hasD = h + 1 < p.H, hasU = h > 0;
        const long pR = pix + 1, pL = pix - 1, pD = pix + p.W, pU = pix - p.W;
        const float* RL0 = p.RL + pix * p.rl_cs;
        const float I0 = RL0[B];
        const float IR = hasR ? p.RL[pR * p.rl_cs + B] : 0.f, IL = hasL ? p.RL[pL * p.rl_cs + B] : 0.f;
        const float ID = hasD ? p.RL[pD * p.rl_cs + B] : 0.f, IU = hasU ? p.RL[pU * p.rl_cs + B] : 0.f;
        const float D0 = p.D[pix * p.d_cs];
        const float DR = hasR ? p.D[pR * p.d_cs] : 0.f, DL = hasL ? p.D[pL * p.d_cs] : 0.f;
        const float DD = hasD ? p.D[pD * p.d_cs] : 0.f, DU = hasU ? p.D[pU * p.d_cs] : 0.f;

        // ---- phase 1: channel reductions per edge ----
        float aR = 0.f, aL = 0.f, aD = 0.f, aU = 0.f;      // sum_c |dR|
        float eR = 0.f, eL = 0.f, eD = 0.f, eU = 0.f;      // sum_c exp(-a2 |dR|)
        float sRsum = 0.f;                                  // sum_c sg(R I - x) R
        for (int c = lane; c < B; c += 32) {
            const float r0 = RL0[c];
            const float xr = p.x[pix * p.x_cs + c];
            sRsum += sgnf(r0 * I0 - xr) * r0;
            if (hasR) { float d = fabsf(p.RL[pR * p.rl_cs + c] - r0); aR += d; eR += expf(-p.a2 * d); }
            if (hasL) { float d = fabsf(r0 - p.RL[pL * p.rl_cs + c]); aL += d; eL += expf(-p.a2 * d); }
            if (hasD) { float d = fabsf(p.RL[pD * p.rl_cs + c] - r0); aD += d; eD += expf(-p.a2 * d); }
            if (hasU) { float d = fabsf(r0 - p.RL[pU * p.rl_cs + c]); aU += d; eU += expf(-p.a2 * d); }
        }
        aR = half_sum(aR); aL = half_sum(aL); aD = half_sum(aD); aU = half_sum(aU);
        eR = half_sum(eR); eL = half_sum(eL); eD = half_sum(eD); eU = half_sum(eU);
        sRsum = half_sum(sRsum);
        const float invC = 1.f / (float)B;
        const float wR = expf(-p.a1 * aR * invC), wL = expf(-p.a1 * aL * invC);
        const float wD = expf(-p.a1 * aD * invC), wU = expf(-p.a1 * aU * invC);
        const float uR = IR - I0, uL = I0 - IL, uD = ID - I0, uU = I0 - IU;          // dI per edge
        const float vR = DR - D0, vL = D0 - DL, vD = DD - D0, vU = D0 - DU;          // dD per edge

        // ---- per-pixel (channel-free) cotangents: I_low and I_delta ----
        if (lane == 0) {
            float gI = p.c_rec * p.inv_n0 * sRsum;
            float gDv = 0.f;
            if (hasR) { gI -= p.c_il * wR * sgnf(uR) * p.inv_nIx; gDv -= p.c_id * sgnf(vR) * eR * p.inv_nRx; acc_il += wR * fabsf(uR) * p.inv_nIx; }
            if (hasL) { gI += p.c_il * wL * sgnf(uL) * p.inv_nIx; gDv += p.c_id * sgnf(vL) * eL * p.inv_nRx; }
            if (hasD) { gI -= p.c_il * wD * sgnf(uD) * p.inv_nIy; gDv -= p.c_id * sgnf(vD) * eD * p.inv_nRy; acc_il += wD * fabsf(uD) * p.inv_nIy; }
            if (hasU) { gI += p.c_il * wU * sgnf(uU) * p.inv_nIy; gDv += p.c_id * sgnf(vU) * eU * p.inv_nRy; }
            p.gRL[pix * p.rl_cs + B] = gI;
            p.gD[pix * p.d_cs] = gDv;
            p.G8b[pix * p.e_cs + B] = 0.f;            // I_enh is unused by the loss (model.py:546)
        }

        // ---- phase 2: per-channel cotangents ----
        const float kil_x = p.c_il * p.a1 * invC * p.inv_nIx, kil_y = p.c_il * p.a1 * invC * p.inv_nIy;
        const float kid_x = p.c_id * p.a2 * p.inv_nRx, kid_y = p.c_id * p.a2 * p.inv_nRy;
        for (int c = lane; c < B; c += 32) {
            const float r0 = RL0[c];
            const float e0 = p.E[pix * p.e_cs + c];
            const float xr = p.x[pix * p.x_cs + c];
            const float d0 = r0 - e0;
            const float srec = sgnf(r0 * I0 - xr);
            acc_rec += fabsf(r0 * I0 - xr) * p.inv_n0;
            acc_rf += fabsf(d0) * p.inv_n0;
            float gR = p.c_rec * p.inv_n0 * srec * I0;
            float gdel = sgnf(d0) * p.inv_n0;
            if (hasR) {
                const float r1 = p.RL[pR * p.rl_cs + c], e1 = p.E[pR * p.e_cs + c];
                const float dR = r1 - r0, dd = (r1 - e1) - d0, ex = expf(-p.a2 * fabsf(dR));
                gR += sgnf(dR) * (kil_x * wR * fabsf(uR) + kid_x * fabsf(vR) * ex);      // -q, q<0 form
                gdel -= 0.5f * sgnf(dd) * p.inv_nRx;
                acc_rf += 0.5f * fabsf(dd) * p.inv_nRx;
                acc_id += fabsf(vR) * ex * p.inv_nRx;
            }
            if (hasL) {
                const float r1 = p.RL[pL * p.rl_cs + c], e1 = p.E[pL * p.e_cs + c];
                const float dR = r0 - r1, dd = d0 - (r1 - e1), ex = expf(-p.a2 * fabsf(dR));
                gR -= sgnf(dR) * (kil_x * wL * fabsf(uL) + kid_x * fabsf(vL) * ex);
                gdel += 0.5f * sgnf(dd) * p.inv_nRx;
            }
            if (hasD) {
                const float r1 = p.RL[pD * p.rl_cs + c], e1 = p.E[pD * p.e_cs + c];
                const float dR = r1 - r0, dd = (r1 - e1) - d0, ex = expf(-p.a2 * fabsf(dR));
                gR += sgnf(dR) * (kil_y * wD * fabsf(uD) + kid_y * fabsf(vD) * ex);
                gdel -= 0.5f * sgnf(dd) * p.inv_nRy;
                acc_rf += 0.5f * fabsf(dd) * p.inv_nRy;
                acc_id += fabsf(vD) * ex * p.inv_nRy;
            }
            if (hasU) {
                const float r1 = p.RL[pU * p.rl_cs + c], e1 = p.E[pU * p.e_cs + c];
                const float dR = r0 - r1, dd = d0 - (r1 - e1), ex = expf(-p.a2 * fabsf(dR));
                gR -= sgnf(dR) * (kil_y * wU * fabsf(uU) + kid_y * fabsf(vU) * ex);
                gdel += 0.5f * sgnf(dd) * p.inv_nRy;
            }
            gR += p.c_rf * gdel;
            p.gRL[pix * p.rl_cs + c] = gR;
            p.G8b[pix * p.e_cs + c] = p.ge_raw ? -p.c_rf * gdel : -p.c_rf * gdel * e0 * (1.f - e0);     // gE (through pass-2's sigmoid)
            // spectral TV on S (band axis = lane axis)
            const float s0 = p.S[pix * p.s_cs + c];
            float gs = 0.f;
            if (c > 0) gs += sgnf(s0 - p.S[pix * p.s_cs + c - 1]);
            if (c + 1 < B) { const float t = p.S[pix * p.s_cs + c + 1] - s0; gs -= sgnf(t); acc_sp += fabsf(t) * p.inv_nsp; }
            p.gS[pix * p.s_cs + c] = p.c_sp * p.inv_nsp * gs;
        }
    }

    // block reduction of the five loss sums
    __shared__ float red[5][8];
    acc_rec = half_sum(acc_rec); acc_rf = half_sum(acc_rf); acc_il = half_sum(acc_il);
    acc_id = half_sum(acc_id); acc_sp = half_sum(acc_sp);
    if (lane == 0) { red[0][grp] = acc_rec; red[1][grp] = acc_rf; red[2][grp] = acc_il; red[3][grp] = acc_id; red[4][grp] = acc_sp; }
    __syncthreads();
    if (tid < 5) {
        float s = 0.f;
        for (int g = 0; g < 8; ++g) s += red[tid][g];
        p.partials[(size_t)blockIdx.x * 8 + tid] = s;
    }
}

// terms[0..5] = (rec, rf, il, id, fourier, sp); out[0] = total, out[1..6] = terms in LOSS order.
// One 256-thread block; fixed-order tree reduction in double => deterministic.
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ partials, int nblk,
                                     const float* __restrict__ fpartials, int nfblk,
                                     float c_rec, float c_rf, float c_il, float c_id, float c_f, float c_sp,
                                     float* __restrict__ out)
{
    __shared__ double red[6][256];
    const int t = threadIdx.x;
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int b = t; b < nblk; b += 256)
#pragma unroll
        for (int k = 0; k < 5; ++k) s[k] += (double)partials[(size_t)b * 8 + k];
    for (int b = t; b < nfblk; b += 256) s[5] += (double)fpartials[b];
#pragma unroll
    for (int k = 0; k < 6; ++k) red[k][t] = s[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o)
#pragma unroll
            for (int k = 0; k < 6; ++k) red[k][t] += red[k][t + o];
        __syncthreads();
    }
    if (t == 0) {
        const double rec = red[0][0], rf = red[1][0], il = red[2][0], id = red[3][0], sp = red[4][0], f = red[5][0];
        out[0] = (float)(c_rec * rec + c_rf * rf + c_il * il + c_id * id + c_f * f + c_sp * sp);
        out[1] = (float)rec; out[2] = (float)rf; out[3] = (float)il; out[4] = (float)id; out[5] = (float)f; out[6] = (float)sp;
    }
}

// close the product node S = R*(D+I) (model.py:233):  gR += gS*(D+I); q = sum_c gS*R; gD += q; gI += q
__global__ __launch_bounds__(256) void product_node_kernel(const float* __restrict__ gS, int s_cs,
                                                           const float* __restrict__ RL, float* __restrict__ gRL, int rl_cs,
                                                           const float* __restrict__ D, float* __restrict__ gD, int d_cs,
                                                           long npix, int B)
{
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    for (long pix = (long)blockIdx.x * 8 + grp; pix < npix; pix += (long)gridDim.x * 8) {
        const float m = D[pix * d_cs] + RL[pix * rl_cs + B];
        float q = 0.f;
        for (int c = lane; c < B; c += 32) {
            const float g = gS[pix * s_cs + c];
            q += g * RL[pix * rl_cs + c];
            gRL[pix * rl_cs + c] += g * m;
        }
        q = half_sum(q);
        if (lane == 0) { gD[pix * d_cs] += q; gRL[pix * rl_cs + B] += q; }
    }
}

// S = R*I_delta + R*I_low (model.py:233); pad channels of S stay zero
__global__ void compose_kernel(const float* __restrict__ RL, int rl_cs, const float* __restrict__ D, int d_cs,
                               float* __restrict__ S, int s_cs, long npix, int B)
{
    const long total = npix * s_cs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / s_cs; const int c = (int)(i % s_cs);
        float v = 0.f;
        if (c < B) { const float r = RL[pix * rl_cs + c]; v = r * D[pix * d_cs] + r * RL[pix * rl_cs + B]; }
        S[i] = v;
    }
}

// logical (N,C,H,W) tensor with arbitrary element strides -> dense NHWC with zero channel padding
__global__ void ingest_kernel(const float* __restrict__ x, long sn, long sc, long sh, long sw,
                              float* __restrict__ out, int N, int C, int H, int W, int cs)
{
    const long total = (long)N * H * W * cs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cs); long r = i / cs;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); const long n = r / H;
        out[i] = c < C ? x[n * sn + c * sc + h * sh + w * sw] : 0.f;
    }
}

// dst = (accumulate ? dst : 0) + src * act'(y)   on the first C channels of each pixel
__global__ void mask_axpy_kernel(const float* __restrict__ src, int src_cs, const float* __restrict__ y, int y_cs, int mode,
                                 float* __restrict__ dst, int dst_cs, long npix, int C, int accumulate)
{
    const long total = npix * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / C; const int c = (int)(i % C);
        float v = src[pix * src_cs + c];
        if (mode == MASK_RELU) v = y[pix * y_cs + c] > 0.f ? v : 0.f;
        else if (mode == MASK_SIGMOID) { const float yy = y[pix * y_cs + c]; v *= yy * (1.f - yy); }
        float* d = dst + pix * dst_cs + c;
        *d = accumulate ? *d + v : v;
    }
}

// adjoint of nearest up-sampling (F.interpolate backward): dst[lo] (+)= sum of src[hi] with src_index(hi) == lo
__global__ void upsample_adjoint_kernel(const float* __restrict__ src, int Hv, int Wv, int src_cs,
                                        float* __restrict__ dst, int Hs, int Ws, int dst_cs, int N, int C,
                                        float sy, float sx, int accumulate)
{
    const long total = (long)N * Hs * Ws * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); long r = i / C;
        const int x = (int)(r % Ws); r /= Ws;
        const int y = (int)(r % Hs); const long n = r / Hs;
        // candidate window: hi rows whose nearest source is y (scale < 1 => at most ceil(1/s)+1 rows)
        int y0 = (int)floorf((float)y / sy) - 1; if (y0 < 0) y0 = 0;
        int x0 = (int)floorf((float)x / sx) - 1; if (x0 < 0) x0 = 0;
        const int ny = (int)ceilf(1.f / sy) + 3, nx = (int)ceilf(1.f / sx) + 3;
        float s = 0.f;
        for (int yy = y0; yy < y0 + ny && yy < Hv; ++yy) {
            if (min((int)floorf((float)yy * sy), Hs - 1) != y) continue;
            for (int xx = x0; xx < x0 + nx && xx < Wv; ++xx) {
                if (min((int)floorf((float)xx * sx), Ws - 1) != x) continue;
                s += src[((n * Hv + yy) * (long)Wv + xx) * src_cs + c];
            }
        }
        float* d = dst + ((n * Hs + y) * (long)Ws + x) * dst_cs + c;
        *d = accumulate ? *d + s : s;
    }
}

// torch.optim.Adam defaults (model.py:213): p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float gscale, float b1, float b2, float step_size, float inv_sqrt_bc2, float eps)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gr = g[i] * gscale;
        const float mi = m[i] + (gr - m[i]) * (1.f - b1);          // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + (1.f - b2) * gr * gr;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] -= step_size * (mi / denom);
    }
}

// ---------------------------------------------------------------------------------------------
static inline unsigned grid_for(long total, int per_block, int cap = 4096)
{
    long b = (total + per_block - 1) / per_block;
    if (b > cap) b = cap; if (b < 1) b = 1;
    return (unsigned)b;
}

int ssie_launch_loss_direct(const LossParams& p, int nblk, hipStream_t st)
{
    hipLaunchKernelGGL(loss_direct_kernel, dim3(nblk), dim3(256), 0, st, p);
    return hipGetLastError() == hipSuccess ? 0 : 41;
}
int ssie_launch_loss_finalize(const float* partials, int nblk, const float* fpartials, int nfblk,
                              const float* coefs6, float* out, hipStream_t st)
{
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, partials, nblk, fpartials, nfblk,
                       coefs6[0], coefs6[1], coefs6[2], coefs6[3], coefs6[4], coefs6[5], out);
    return hipGetLastError() == hipSuccess ? 0 : 42;
}
int ssie_launch_product_node(const float* gS, int s_cs, const float* RL, float* gRL, int rl_cs, const float* D, float* gD,
                             int d_cs, long npix, int B, hipStream_t st)
{
    hipLaunchKernelGGL(product_node_kernel, dim3(grid_for(npix, 8)), dim3(256), 0, st, gS, s_cs, RL, gRL, rl_cs, D, gD, d_cs, npix, B);
    return hipGetLastError() == hipSuccess ? 0 : 43;
}
int ssie_launch_compose(const float* RL, int rl_cs, const float* D, int d_cs, float* S, int s_cs, long npix, int B, hipStream_t st)
{
    hipLaunchKernelGGL(compose_kernel, dim3(grid_for(npix * s_cs, 256)), dim3(256), 0, st, RL, rl_cs, D, d_cs, S, s_cs, npix, B);
    return hipGetLastError() == hipSuccess ? 0 : 44;
}
int ssie_launch_ingest(const float* x, long sn, long sc, long sh, long sw, float* out, int N, int C, int H, int W, int cs, hipStream_t st)
{
    hipLaunchKernelGGL(ingest_kernel, dim3(grid_for((long)N * H * W * cs, 256)), dim3(256), 0, st, x, sn, sc, sh, sw, out, N, C, H, W, cs);
    return hipGetLastError() == hipSuccess ? 0 : 45;
}
int ssie_launch_mask_axpy(const float* src, int src_cs, const float* y, int y_cs, int mode, float* dst, int dst_cs,
                          long npix, int C, int accumulate, hipStream_t st)
{
    hipLaunchKernelGGL(mask_axpy_kernel, dim3(grid_for(npix * C, 256)), dim3(256), 0, st, src, src_cs, y, y_cs, mode, dst, dst_cs, npix, C, accumulate);
    return hipGetLastError() == hipSuccess ? 0 : 46;
}
int ssie_launch_upsample_adjoint(const float* src, int Hv, int Wv, int src_cs, float* dst, int Hs, int Ws, int dst_cs,
                                 int N, int C, int accumulate, hipStream_t st)
{
    const float sy = (Hs == Hv) ? 1.f : (float)Hs / (float)Hv, sx = (Ws == Wv) ? 1.f : (float)Ws / (float)Wv;
    hipLaunchKernelGGL(upsample_adjoint_kernel, dim3(grid_for((long)N * Hs * Ws * C, 256)), dim3(256), 0, st,
                       src, Hv, Wv, src_cs, dst, Hs, Ws, dst_cs, N, C, sy, sx, accumulate);
    return hipGetLastError() == hipSuccess ? 0 : 47;
}
int ssie_launch_adam(float* p, const float* g, float* m, float* v, long n, float gscale, float lr, int step,
                     float b1, float b2, float eps, hipStream_t st)
{
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, st, p, g, m, v, n, gscale, b1, b2,
                       (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), eps);
    return hipGetLastError() == hipSuccess ? 0 : 48;
}

// ---------------------------------------------------------------------------------------------
// fp32 -> bf16 conversion for the bf16 inference path
// ---------------------------------------------------------------------------------------------
__global__ void to_bf16_kernel(const f32x4* __restrict__ src, uint2* __restrict__ dst, long n4)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = src[i];
        dst[i] = make_uint2(ssie_pack2bf(v[0], v[1]), ssie_pack2bf(v[2], v[3]));
    }
}

int ssie_launch_to_bf16(const float* src, void* dst, long n, hipStream_t st)
{
    if (n % 4) return 71;
    const long n4 = n / 4;
    long blocks = (n4 + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const f32x4*)src, (uint2*)dst, n4);
    return hipGetLastError() == hipSuccess ? 0 : 72;
}
