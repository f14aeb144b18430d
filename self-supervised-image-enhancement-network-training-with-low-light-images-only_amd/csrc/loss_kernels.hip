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

// ---------------------------------------------------------------------------------------------
// Tiled variant (the one the plan runs for up to 252 bands): one workgroup per TH x TW pixel tile.
//   * the tile of R|I (pass 1) and of R_enh (pass 2) is staged ONCE, with a one-pixel halo, into LDS by 16-byte loads;
//     x and S (no spatial neighbours needed) go straight to registers, issued BEFORE the staging loads so that every byte
//     the tile needs is in flight together
//   * LPP lanes share a pixel, each owning four consecutive bands (one float4): all HBM accesses are 16 bytes per lane and
//     contiguous over the pixel's band vector; the band reductions are LPP-lane butterflies
//   * one pass: the per-band edge terms (|dR|, exp(-a2 |dR|), sg(dR)) of the four edges are computed once, kept in registers
//     across the band reduction and reused for the cotangents (the half-wave kernel above evaluates each twice)
// Same arithmetic per element as loss_direct_kernel (single subtractions for every sg() argument), so the results agree to
// the rounding of the reductions' summation order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }

// The tiled kernels are VALU-bound at wide band counts (PMC at 256 bands: ~300 vector instructions per band and pixel, VALU busy
// 45 % of the kernel, HBM 1.3 TB/s), so their two hot primitives are the short forms:
//   sg(v)   = copysign(v != 0, v): compare + select + bit-field insert (exact; the generic kernel's (v > 0) - (v < 0) is five)
//   exp(-t) = v_exp_f32(-t * log2 e): two instructions, relative error ~4e-7 at t = 10 (expf: ~15 instructions, 1e-7); the
//             results only weight magnitudes (never decide a sign) and are held to 1e-5 of the tensor maximum by the tests
__device__ __forceinline__ float sg3(float v) { return __builtin_copysignf(v != 0.f ? 1.f : 0.f, v); }
__device__ __forceinline__ float expneg(float t) { return __builtin_amdgcn_exp2f(t * -1.44269504088896341f); }

template <int LPP>
__device__ __forceinline__ float grp_sum(float v)      // sum over the LPP lanes that share a pixel
{
#pragma unroll
    for (int o = LPP >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// tile geometry of the tiled kernels: four passes of 256 / LPP pixels, 16 (LPP <= 16) or 8 pixels wide
template <int LPP> struct LossTile {
    static constexpr int SLOTS = 256 / LPP, TW = LPP <= 16 ? 16 : 8, TH = 4 * SLOTS / TW, PW = TW + 2, PH = TH + 2, PXH = PH * PW;
    static constexpr int NIT = (PXH * LPP + 255) / 256;           // staging iterations per thread (LPP 16-byte slots per halo pixel)
};

// (tile size as RUN-TIME arguments and a run-time staging loop: with compile-time geometry, or with the staging loads batched as in
// loss_chunk_kernel below, hipcc allocates 256 instead of 227 registers and the kernel runs 0.26 - 0.29 instead of 0.19 ms at 31 bands)
template <int LPP>
__global__ __launch_bounds__(256) void loss_tile_kernel(const LossParams p, int TH, int TW, int tiles_y, int tiles_x)
{
    extern __shared__ __attribute__((aligned(16))) float smem_l[];
    constexpr int SLOTS = 256 / LPP;                   // pixels per pass
    constexpr int MAXPASS = 4;
    const int B = p.B, H = p.H, W = p.W;
    const int nq = p.rl_cs >> 2, nqx = p.x_cs >> 2;    // float4s per pixel of RL / E and of x / S
    const int PW = TW + 2, PH = TH + 2, pxh = PH * PW;
    float* RLt = smem_l;                               // [PH][PW][rl_cs]
    float* Et = RLt + (size_t)pxh * p.rl_cs;           // [PH][PW][rl_cs]
    float* Dt = Et + (size_t)pxh * p.rl_cs;            // [PH][PW]
    const int tid = threadIdx.x, sub = tid % LPP, grp = tid / LPP;
    const int c0 = sub * 4;
    const bool lane_rl = sub < nq, lane_x = sub < nqx;
    float cm[4];                                       // 1 for real bands, 0 for the I / padding channels of this lane's float4
#pragma unroll
    for (int k = 0; k < 4; ++k) cm[k] = (c0 + k < B) ? 1.f : 0.f;
    const float invC = 1.f / (float)B;
    const float kil_x = p.c_il * p.a1 * invC * p.inv_nIx, kil_y = p.c_il * p.a1 * invC * p.inv_nIy;
    const float kid_x = p.c_id * p.a2 * p.inv_nRx, kid_y = p.c_id * p.a2 * p.inv_nRy;
    float acc_rec = 0.f, acc_rf = 0.f, acc_il = 0.f, acc_id = 0.f, acc_sp = 0.f;
    const int per_img = tiles_y * tiles_x, ntiles = p.N * per_img, npass = (TH * TW + SLOTS - 1) / SLOTS;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / per_img, tr = tile - n * per_img, ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int h0 = ty * TH, w0 = tx * TW;
        const long img = (long)n * H * W;
        // ---- x and S of this lane's pixels, all passes, straight to registers ----
        f32x4 xv[MAXPASS], sv[MAXPASS];
#pragma unroll
        for (int ps = 0; ps < MAXPASS; ++ps) {
            const int idx = ps * SLOTS + grp, ph = idx / TW, pw = idx - ph * TW;
            const bool ok = ps < npass && idx < TH * TW && h0 + ph < H && w0 + pw < W && lane_x;
            const long pix = img + (long)(h0 + ph) * W + (w0 + pw);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            xv[ps] = ok ? ld4(p.x + pix * p.x_cs + c0) : z;
            sv[ps] = ok ? ld4(p.S + pix * p.s_cs + c0) : z;
        }
        __syncthreads();                               // the previous tile's LDS reads are done
        // ---- stage R|I, R_enh (halo of one pixel; zeros outside the image) and I_delta ----
        for (int i = tid; i < pxh * nq; i += 256) {
            const int px = i / nq, q = i - px * nq, py = px / PW, pxx = px - py * PW;
            const int hh = h0 - 1 + py, ww = w0 - 1 + pxx;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
            if (hh >= 0 && hh < H && ww >= 0 && ww < W) {
                const long pix = img + (long)hh * W + ww;
                a = ld4(p.RL + pix * p.rl_cs + 4 * q); b = ld4(p.E + pix * p.e_cs + 4 * q);
            }
            *(f32x4*)(RLt + (size_t)px * p.rl_cs + 4 * q) = a;
            *(f32x4*)(Et + (size_t)px * p.rl_cs + 4 * q) = b;
        }
        for (int i = tid; i < pxh; i += 256) {
            const int py = i / PW, pxx = i - py * PW, hh = h0 - 1 + py, ww = w0 - 1 + pxx;
            Dt[i] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? p.D[(img + (long)hh * W + ww) * p.d_cs] : 0.f;
        }
        __syncthreads();

#pragma unroll
        for (int ps = 0; ps < MAXPASS; ++ps) {
            if (ps >= npass) break;
            const int idx = ps * SLOTS + grp, ph = idx / TW, pw = idx - ph * TW;
            const int h = h0 + ph, w = w0 + pw;
            const bool live = idx < TH * TW && h < H && w < W;          // uniform over the LPP lanes of a pixel
            const int li = live ? (ph + 1) * PW + (pw + 1) : PW + 1;    // dead slots read a valid LDS address and store nothing
            const bool hasR = live && w + 1 < W, hasL = live && w > 0, hasD = live && h + 1 < H, hasU = live && h > 0;
            const float fR = hasR ? 1.f : 0.f, fL = hasL ? 1.f : 0.f, fD = hasD ? 1.f : 0.f, fU = hasU ? 1.f : 0.f;
            const int nb[4] = {li + 1, li - 1, li + PW, li - PW};         // R, L, D, U
            const float sgnd[4] = {1.f, -1.f, 1.f, -1.f};               // edge difference = sgnd * (neighbour - own)
            const float hasf[4] = {fR, fL, fD, fU};
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 r0 = lane_rl ? *(const f32x4*)(RLt + (size_t)li * p.rl_cs + c0) : z4;
            const f32x4 e0 = lane_rl ? *(const f32x4*)(Et + (size_t)li * p.rl_cs + c0) : z4;
            const float I0 = RLt[(size_t)li * p.rl_cs + B], D0 = Dt[li];
            f32x4 dRv[4], exv[4], ddv[4];
            float a_sum[4], e_sum[4], u[4], v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 rn = lane_rl ? *(const f32x4*)(RLt + (size_t)nb[k] * p.rl_cs + c0) : z4;
                const f32x4 en = lane_rl ? *(const f32x4*)(Et + (size_t)nb[k] * p.rl_cs + c0) : z4;
                u[k] = sgnd[k] * (RLt[(size_t)nb[k] * p.rl_cs + B] - I0);
                v[k] = sgnd[k] * (Dt[nb[k]] - D0);
                float as = 0.f, es = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = sgnd[k] * (rn[j] - r0[j]);                 // one subtraction, like the reference's R[1:] - R[:-1]
                    const float ex = expneg(p.a2 * fabsf(d));
                    dRv[k][j] = d; exv[k][j] = ex;
                    ddv[k][j] = sgnd[k] * ((rn[j] - en[j]) - (r0[j] - e0[j]));
                    as += cm[j] * fabsf(d); es += cm[j] * ex;
                }
                a_sum[k] = grp_sum<LPP>(as) * hasf[k]; e_sum[k] = grp_sum<LPP>(es) * hasf[k];
            }
            const f32x4 xr = xv[ps];
            float srs = 0.f;
            float srec[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = r0[j] * I0 - xr[j];
                srec[j] = cm[j] * sg3(t);
                srs += srec[j] * r0[j];
                acc_rec += live ? cm[j] * fabsf(t) * p.inv_n0 : 0.f;
            }
            srs = grp_sum<LPP>(srs);
            float wgt[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) wgt[k] = expneg(p.a1 * a_sum[k] * invC);
            // ---- per-pixel cotangents (I_low, I_delta): identical on all LPP lanes; the lane owning channel B stores gI ----
            float gI = p.c_rec * p.inv_n0 * srs, gDv = 0.f;
            {
                const float nIx = p.inv_nIx, nIy = p.inv_nIy, nRx = p.inv_nRx, nRy = p.inv_nRy;
                gI += -fR * p.c_il * wgt[0] * sg3(u[0]) * nIx + fL * p.c_il * wgt[1] * sg3(u[1]) * nIx
                      - fD * p.c_il * wgt[2] * sg3(u[2]) * nIy + fU * p.c_il * wgt[3] * sg3(u[3]) * nIy;
                gDv += -fR * p.c_id * sg3(v[0]) * e_sum[0] * nRx + fL * p.c_id * sg3(v[1]) * e_sum[1] * nRx
                       - fD * p.c_id * sg3(v[2]) * e_sum[2] * nRy + fU * p.c_id * sg3(v[3]) * e_sum[3] * nRy;
                if (sub == 0) acc_il += fR * wgt[0] * fabsf(u[0]) * nIx + fD * wgt[2] * fabsf(u[2]) * nIy;
            }
            // ---- per-band cotangents ----
            const float kil[4] = {kil_x, kil_x, kil_y, kil_y}, kid[4] = {kid_x, kid_x, kid_y, kid_y};
            const float nR[4] = {p.inv_nRx, p.inv_nRx, p.inv_nRy, p.inv_nRy};
            const float esg[4] = {1.f, -1.f, 1.f, -1.f};                    // own-end sign of an edge's flux
            f32x4 gR, g8, gs;
            const f32x4 sr = sv[ps];
            // band neighbours of S across the float4 boundary come from the adjacent lanes of the pixel group
            const float s_prev = __shfl_up(sr[3], 1), s_next = __shfl_down(sr[0], 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j;
                const float d0 = r0[j] - e0[j];
                float g = p.c_rec * p.inv_n0 * srec[j] * I0;
                float gdel = sg3(d0) * p.inv_n0;
                acc_rf += live ? cm[j] * fabsf(d0) * p.inv_n0 : 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = dRv[k][j], ex = exv[k][j], dd = ddv[k][j];
                    g += esg[k] * hasf[k] * sg3(d) * (kil[k] * wgt[k] * fabsf(u[k]) + kid[k] * fabsf(v[k]) * ex);
                    gdel -= esg[k] * hasf[k] * 0.5f * sg3(dd) * nR[k];
                    if (k == 0 || k == 2) {                                  // each edge's loss is counted at its left / upper end
                        acc_rf += cm[j] * hasf[k] * 0.5f * fabsf(dd) * nR[k];
                        acc_id += cm[j] * hasf[k] * fabsf(v[k]) * ex * nR[k];
                    }
                }
                g += p.c_rf * gdel;
                const float ge = p.ge_raw ? -p.c_rf * gdel : -p.c_rf * gdel * e0[j] * (1.f - e0[j]);
                gR[j] = c < B ? g : (c == B ? gI : 0.f);
                g8[j] = c < B ? ge : 0.f;
                // spectral TV along the band axis (model.py:475-481)
                const float s0 = sr[j];
                const float sm = j > 0 ? sr[j - 1] : s_prev, sp = j < 3 ? sr[j + 1] : s_next;
                float gsv = 0.f;
                if (c > 0 && c < B) gsv += sg3(s0 - sm);
                if (c + 1 < B) { const float t = sp - s0; gsv -= sg3(t); acc_sp += live ? fabsf(t) * p.inv_nsp : 0.f; }
                gs[j] = p.c_sp * p.inv_nsp * gsv;
            }
            if (live) {
                const long pix = img + (long)h * W + w;
                if (lane_rl) { *(f32x4*)(p.gRL + pix * p.rl_cs + c0) = gR; *(f32x4*)(p.G8b + pix * p.e_cs + c0) = g8; }
                if (lane_x) *(f32x4*)(p.gS + pix * p.s_cs + c0) = gs;
                if (sub == 0) p.gD[pix * p.d_cs] = gDv;
            }
        }
    }
    // block reduction of the five loss sums (every block writes its slot, also when it had no tile)
    __shared__ float red[5][4];
    float s5[5] = {acc_rec, acc_rf, acc_il, acc_id, acc_sp};
#pragma unroll
    for (int k = 0; k < 5; ++k) { float t = s5[k]; for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o); s5[k] = t; }
    __syncthreads();
    if ((tid & 63) == 0) for (int k = 0; k < 5; ++k) red[k][tid >> 6] = s5[k];
    __syncthreads();
    if (tid < 5) p.partials[(size_t)blockIdx.x * 8 + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
}

// ---------------------------------------------------------------------------------------------
// Band-chunked variant of the tiled kernel, for band counts whose pixel vector does not fit one pass of the kernel above (more
// than 252 bands: BASELINE configs[2] has 256) or whose tile would leave one workgroup per CU.  A pixel's band vector is walked
// in chunks of CB = 4 * LPP bands; a tile stages only the chunk it is working on (R|I and R_enh with the one-pixel halo), so the
// LDS footprint is that of a CB-band problem (2 - 3 workgroups per CU: one workgroup's staging runs under another's arithmetic).
// Two sweeps over the chunks:
//   sweep 1 (R only)  a_k = sum_c |dR_c| per edge -> the edge weights w_k = exp(-a1 mean_c |dR|), the only quantities a per-band
//                     cotangent needs from OTHER bands
//   sweep 2           everything per band (cotangents of R, R_enh, S; loss sums), and the remaining band sums (sum_c exp(-a2
//                     |dR_c|) per edge, sum_c sg(R I - x) R), which only feed the per-PIXEL cotangents of I_low / I_delta that are
//                     stored after the sweep.  R is read twice (the second time from L2).
// Per element the arithmetic is that of loss_tile_kernel (single subtractions for every sg() argument).
// ---------------------------------------------------------------------------------------------
template <int LPP>
__global__ __launch_bounds__(256) void loss_chunk_kernel(const LossParams p, int tiles_y, int tiles_x)
{
    extern __shared__ __attribute__((aligned(16))) float smem_l[];
    constexpr int SLOTS = 256 / LPP;                   // pixels per pass
    constexpr int MAXPASS = 4;
    constexpr int CB = 4 * LPP;                        // bands per chunk
    constexpr int TW = LossTile<LPP>::TW, TH = LossTile<LPP>::TH, PW = TW + 2, PH = TH + 2, pxh = PH * PW, NIT = LossTile<LPP>::NIT;
    const int B = p.B, H = p.H, W = p.W;
    float* RLt = smem_l;                               // [PH][PW][CB]
    float* Et = RLt + (size_t)pxh * CB;                // [PH][PW][CB]
    float* It = Et + (size_t)pxh * CB;                 // [PH][PW]  I_low
    float* Dt = It + pxh;                              // [PH][PW]  I_delta
    const int tid = threadIdx.x, sub = tid % LPP, grp = tid / LPP;
    const int nchunk = (B + CB - 1) / CB;
    const int nq = p.rl_cs >> 2;                       // float4s per pixel of RL / E (x / S: the same count or one less, never read past B)
    const float invC = 1.f / (float)B;
    const float kil_x = p.c_il * p.a1 * invC * p.inv_nIx, kil_y = p.c_il * p.a1 * invC * p.inv_nIy;
    const float kid_x = p.c_id * p.a2 * p.inv_nRx, kid_y = p.c_id * p.a2 * p.inv_nRy;
    float acc_rec = 0.f, acc_rf = 0.f, acc_il = 0.f, acc_id = 0.f, acc_sp = 0.f;
    const int per_img = tiles_y * tiles_x, ntiles = p.N * per_img, npass = (TH * TW + SLOTS - 1) / SLOTS;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const float sgnd[4] = {1.f, -1.f, 1.f, -1.f};      // edge difference = sgnd * (neighbour - own), edges R, L, D, U

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / per_img, tr = tile - n * per_img, ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int h0 = ty * TH, w0 = tx * TW;
        const long img = (long)n * H * W;
        // per-pass pixel of this lane group
        int li[MAXPASS]; bool live[MAXPASS]; float hasf[MAXPASS][4];
#pragma unroll
        for (int ps = 0; ps < MAXPASS; ++ps) {
            const int idx = ps * SLOTS + grp, ph = idx / TW, pw = idx - ph * TW, h = h0 + ph, w = w0 + pw;
            live[ps] = ps < npass && idx < TH * TW && h < H && w < W;
            li[ps] = live[ps] ? (ph + 1) * PW + (pw + 1) : PW + 1;       // dead slots read a valid LDS address and store nothing
            hasf[ps][0] = (live[ps] && w + 1 < W) ? 1.f : 0.f; hasf[ps][1] = (live[ps] && w > 0) ? 1.f : 0.f;
            hasf[ps][2] = (live[ps] && h + 1 < H) ? 1.f : 0.f; hasf[ps][3] = (live[ps] && h > 0) ? 1.f : 0.f;
        }
        __syncthreads();                               // the previous tile's LDS reads are done
        for (int i = tid; i < pxh; i += 256) {
            const int py = i / PW, pxx = i - py * PW, hh = h0 - 1 + py, ww = w0 - 1 + pxx;
            const bool in = hh >= 0 && hh < H && ww >= 0 && ww < W;
            const long pix = img + (long)hh * W + ww;
            It[i] = in ? p.RL[pix * p.rl_cs + B] : 0.f;
            Dt[i] = in ? p.D[pix * p.d_cs] : 0.f;
        }
        // ---- sweep 1: sum_c |dR_c| per edge ----
        float asum[MAXPASS][4];
#pragma unroll
        for (int ps = 0; ps < MAXPASS; ++ps)
#pragma unroll
            for (int k = 0; k < 4; ++k) asum[ps][k] = 0.f;
        for (int ch = 0; ch < nchunk; ++ch) {
            if (ch) __syncthreads();                   // the previous chunk's LDS reads are done
            {   // all of a thread's staging loads in flight together (see loss_tile_kernel)
                f32x4 av[NIT];
#pragma unroll
                for (int u = 0; u < NIT; ++u) {
                    const int i = tid + u * 256, px = i / LPP, q = i % LPP, py = px / PW, pxx = px - py * PW;
                    const int hh = h0 - 1 + py, ww = w0 - 1 + pxx, cq = ch * LPP + q;
                    av[u] = z4;
                    if (px < pxh && hh >= 0 && hh < H && ww >= 0 && ww < W && cq < nq) av[u] = ld4(p.RL + (img + (long)hh * W + ww) * p.rl_cs + 4 * cq);
                }
#pragma unroll
                for (int u = 0; u < NIT; ++u) {
                    const int i = tid + u * 256, px = i / LPP, q = i % LPP;
                    if (px < pxh) *(f32x4*)(RLt + (size_t)px * CB + 4 * q) = av[u];
                }
            }
            __syncthreads();
            const int c0 = ch * CB + sub * 4;
            float cm[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cm[j] = (c0 + j < B) ? 1.f : 0.f;
#pragma unroll
            for (int ps = 0; ps < MAXPASS; ++ps) {
                if (ps >= npass) break;
                const f32x4 r0 = *(const f32x4*)(RLt + (size_t)li[ps] * CB + sub * 4);
                const int nb[4] = {li[ps] + 1, li[ps] - 1, li[ps] + PW, li[ps] - PW};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 rn = *(const f32x4*)(RLt + (size_t)nb[k] * CB + sub * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) asum[ps][k] += cm[j] * fabsf(sgnd[k] * (rn[j] - r0[j]));
                }
            }
        }
        float wgt[MAXPASS][4];
#pragma unroll
        for (int ps = 0; ps < MAXPASS; ++ps)
#pragma unroll
            for (int k = 0; k < 4; ++k) wgt[ps][k] = expneg(p.a1 * (grp_sum<LPP>(asum[ps][k]) * hasf[ps][k]) * invC);

        // ---- sweep 2: per-band cotangents, loss sums, and the band sums of the per-pixel cotangents ----
        float esum[MAXPASS][4], srs[MAXPASS];
#pragma unroll
        for (int ps = 0; ps < MAXPASS; ++ps) { srs[ps] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) esum[ps][k] = 0.f; }
        for (int ch = 0; ch < nchunk; ++ch) {
            const int c0 = ch * CB + sub * 4;
            const bool lane_c = c0 < B;                // this lane's float4 holds at least one real band
            // x and S of this lane's pixels straight to registers, issued before the staging loads
            f32x4 xv[MAXPASS], sv[MAXPASS];
#pragma unroll
            for (int ps = 0; ps < MAXPASS; ++ps) {
                const int idx = ps * SLOTS + grp, ph = idx / TW, pw = idx - ph * TW;
                const long pix = img + (long)(h0 + ph) * W + (w0 + pw);
                const bool ok = live[ps] && lane_c;
                xv[ps] = ok ? ld4(p.x + pix * p.x_cs + c0) : z4;
                sv[ps] = ok ? ld4(p.S + pix * p.s_cs + c0) : z4;
            }
            __syncthreads();                           // sweep 1's / the previous chunk's LDS reads are done
#pragma unroll
            for (int b0 = 0; b0 < NIT; b0 += 4) {      // batches of four slots per thread (eight loads in flight)
                f32x4 av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = tid + (b0 + u) * 256, px = i / LPP, q = i % LPP, py = px / PW, pxx = px - py * PW;
                    const int hh = h0 - 1 + py, ww = w0 - 1 + pxx, cq = ch * LPP + q;
                    av[u] = z4; bv[u] = z4;
                    if (b0 + u < NIT && px < pxh && hh >= 0 && hh < H && ww >= 0 && ww < W && cq < nq) {
                        const long pix = img + (long)hh * W + ww;
                        av[u] = ld4(p.RL + pix * p.rl_cs + 4 * cq); bv[u] = ld4(p.E + pix * p.e_cs + 4 * cq);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = tid + (b0 + u) * 256, px = i / LPP, q = i % LPP;
                    if (b0 + u < NIT && px < pxh) {
                        *(f32x4*)(RLt + (size_t)px * CB + 4 * q) = av[u];
                        *(f32x4*)(Et + (size_t)px * CB + 4 * q) = bv[u];
                    }
                }
            }
            __syncthreads();
            float cm[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cm[j] = (c0 + j < B) ? 1.f : 0.f;
#pragma unroll
            for (int ps = 0; ps < MAXPASS; ++ps) {
                if (ps >= npass) break;
                const int l0 = li[ps];
                const int nb[4] = {l0 + 1, l0 - 1, l0 + PW, l0 - PW};
                const f32x4 r0 = *(const f32x4*)(RLt + (size_t)l0 * CB + sub * 4);
                const f32x4 e0 = *(const f32x4*)(Et + (size_t)l0 * CB + sub * 4);
                const float I0 = It[l0], D0 = Dt[l0];
                float u[4], v[4];
                f32x4 dRv[4], exv[4], ddv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 rn = *(const f32x4*)(RLt + (size_t)nb[k] * CB + sub * 4);
                    const f32x4 en = *(const f32x4*)(Et + (size_t)nb[k] * CB + sub * 4);
                    u[k] = sgnd[k] * (It[nb[k]] - I0);
                    v[k] = sgnd[k] * (Dt[nb[k]] - D0);
                    float es = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = sgnd[k] * (rn[j] - r0[j]);
                        const float ex = expneg(p.a2 * fabsf(d));
                        dRv[k][j] = d; exv[k][j] = ex;
                        ddv[k][j] = sgnd[k] * ((rn[j] - en[j]) - (r0[j] - e0[j]));
                        es += cm[j] * ex;
                    }
                    esum[ps][k] += es;
                }
                const f32x4 xr = xv[ps];
                float srec[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = r0[j] * I0 - xr[j];
                    srec[j] = cm[j] * sg3(t);
                    srs[ps] += srec[j] * r0[j];
                    acc_rec += live[ps] ? cm[j] * fabsf(t) * p.inv_n0 : 0.f;
                }
                const float kil[4] = {kil_x, kil_x, kil_y, kil_y}, kid[4] = {kid_x, kid_x, kid_y, kid_y};
                const float nR[4] = {p.inv_nRx, p.inv_nRx, p.inv_nRy, p.inv_nRy};
                const float esg[4] = {1.f, -1.f, 1.f, -1.f};                    // own-end sign of an edge's flux
                f32x4 gR, g8, gs;
                const f32x4 sr = sv[ps];
                const int idx = ps * SLOTS + grp, ph = idx / TW, pw = idx - ph * TW;
                const long pix = img + (long)(h0 + ph) * W + (w0 + pw);
                // band neighbours of S across the float4 boundary: adjacent lanes of the pixel group; across a CHUNK boundary the
                // neighbouring band is read from memory (one L1-resident float)
                float s_prev = __shfl_up(sr[3], 1), s_next = __shfl_down(sr[0], 1);
                if (live[ps] && lane_c) {
                    if (sub == 0 && c0 > 0) s_prev = p.S[pix * p.s_cs + c0 - 1];
                    if (sub == LPP - 1 && c0 + 4 < B) s_next = p.S[pix * p.s_cs + c0 + 4];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c0 + j;
                    const float d0 = r0[j] - e0[j];
                    float g = p.c_rec * p.inv_n0 * srec[j] * I0;
                    float gdel = sg3(d0) * p.inv_n0;
                    acc_rf += live[ps] ? cm[j] * fabsf(d0) * p.inv_n0 : 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float d = dRv[k][j], ex = exv[k][j], dd = ddv[k][j], hf = hasf[ps][k];
                        g += esg[k] * hf * sg3(d) * (kil[k] * wgt[ps][k] * fabsf(u[k]) + kid[k] * fabsf(v[k]) * ex);
                        gdel -= esg[k] * hf * 0.5f * sg3(dd) * nR[k];
                        if (k == 0 || k == 2) {                                  // each edge's loss is counted at its left / upper end
                            acc_rf += cm[j] * hf * 0.5f * fabsf(dd) * nR[k];
                            acc_id += cm[j] * hf * fabsf(v[k]) * ex * nR[k];
                        }
                    }
                    g += p.c_rf * gdel;
                    const float ge = p.ge_raw ? -p.c_rf * gdel : -p.c_rf * gdel * e0[j] * (1.f - e0[j]);
                    gR[j] = c < B ? g : 0.f;           // channel B (I_low) and the padding channels: stored after the sweep / zero
                    g8[j] = c < B ? ge : 0.f;
                    const float s0 = sr[j];
                    const float sm = j > 0 ? sr[j - 1] : s_prev, sp = j < 3 ? sr[j + 1] : s_next;
                    float gsv = 0.f;
                    if (c > 0 && c < B) gsv += sg3(s0 - sm);
                    if (c + 1 < B) { const float t = sp - s0; gsv -= sg3(t); acc_sp += live[ps] ? fabsf(t) * p.inv_nsp : 0.f; }
                    gs[j] = p.c_sp * p.inv_nsp * gsv;
                }
                if (live[ps] && lane_c) {
                    *(f32x4*)(p.gRL + pix * p.rl_cs + c0) = gR; *(f32x4*)(p.G8b + pix * p.e_cs + c0) = g8;
                    *(f32x4*)(p.gS + pix * p.s_cs + c0) = gs;
                }
            }
        }
        // ---- per-pixel cotangents (I_low, I_delta), after the last chunk ----
#pragma unroll
        for (int ps = 0; ps < MAXPASS; ++ps) {
            if (ps >= npass) break;
            const int l0 = li[ps];
            const int nb[4] = {l0 + 1, l0 - 1, l0 + PW, l0 - PW};
            const float I0 = It[l0], D0 = Dt[l0];
            float u[4], v[4], es[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u[k] = sgnd[k] * (It[nb[k]] - I0); v[k] = sgnd[k] * (Dt[nb[k]] - D0);
                es[k] = grp_sum<LPP>(esum[ps][k]) * hasf[ps][k];
            }
            const float srsum = grp_sum<LPP>(srs[ps]);
            const float fR = hasf[ps][0], fL = hasf[ps][1], fD = hasf[ps][2], fU = hasf[ps][3];
            const float nIx = p.inv_nIx, nIy = p.inv_nIy, nRx = p.inv_nRx, nRy = p.inv_nRy;
            float gI = p.c_rec * p.inv_n0 * srsum, gDv = 0.f;
            gI += -fR * p.c_il * wgt[ps][0] * sg3(u[0]) * nIx + fL * p.c_il * wgt[ps][1] * sg3(u[1]) * nIx
                  - fD * p.c_il * wgt[ps][2] * sg3(u[2]) * nIy + fU * p.c_il * wgt[ps][3] * sg3(u[3]) * nIy;
            gDv += -fR * p.c_id * sg3(v[0]) * es[0] * nRx + fL * p.c_id * sg3(v[1]) * es[1] * nRx
                   - fD * p.c_id * sg3(v[2]) * es[2] * nRy + fU * p.c_id * sg3(v[3]) * es[3] * nRy;
            if (live[ps] && sub == 0) {
                acc_il += fR * wgt[ps][0] * fabsf(u[0]) * nIx + fD * wgt[ps][2] * fabsf(u[2]) * nIy;
                const int idx = ps * SLOTS + grp, ph = idx / TW, pw = idx - ph * TW;
                const long pix = img + (long)(h0 + ph) * W + (w0 + pw);
                p.gD[pix * p.d_cs] = gDv;
                if ((B & 3) == 0) {                    // I_low opens a float4 of its own: [gI, 0, 0, 0] (and zeros in G8b: I_enh is unused, model.py:546)
                    *(f32x4*)(p.gRL + pix * p.rl_cs + B) = f32x4{gI, 0.f, 0.f, 0.f};
                    *(f32x4*)(p.G8b + pix * p.e_cs + B) = z4;
                } else p.gRL[pix * p.rl_cs + B] = gI;  // inside the last band float4, which the sweep stored with a 0 here (same wave, program order)
            }
        }
    }
    // block reduction of the five loss sums (every block writes its slot, also when it had no tile)
    __shared__ float red[5][4];
    float s5[5] = {acc_rec, acc_rf, acc_il, acc_id, acc_sp};
#pragma unroll
    for (int k = 0; k < 5; ++k) { float t = s5[k]; for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o); s5[k] = t; }
    __syncthreads();
    if ((tid & 63) == 0) for (int k = 0; k < 5; ++k) red[k][tid >> 6] = s5[k];
    __syncthreads();
    if (tid < 5) p.partials[(size_t)blockIdx.x * 8 + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
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

// the same node with 16-byte accesses: LPP lanes share a pixel, lane `sub` owning float4 `sub` of the band vector (nq <= LPP;
// aligned tensors with channel strides that are multiples of 4); the band sum is an LPP-lane butterfly, and the lane that owns
// channel B (I_low) adds q to it inside its own float4, so every element is written by exactly one lane
template <int LPP>
__global__ __launch_bounds__(256) void product_node_vec_kernel(const float* __restrict__ gS, int s_cs,
                                                               const float* __restrict__ RL, float* __restrict__ gRL, int rl_cs,
                                                               const float* __restrict__ D, float* __restrict__ gD, int d_cs,
                                                               long npix, int B)
{
    constexpr int PPB = 256 / LPP;
    const int sub = threadIdx.x % LPP, grp = threadIdx.x / LPP;
    const int nq = rl_cs >> 2, nqs = s_cs >> 2;
    const bool own = sub < nq && 4 * sub <= B;             // quads up to the one that holds channel B
    for (long pix0 = (long)blockIdx.x * PPB; pix0 < npix; pix0 += (long)gridDim.x * PPB) {
        const long pix = pix0 + grp;
        const bool live = pix < npix;
        const long pc = live ? pix : npix - 1;
        const float m = D[pc * d_cs] + RL[pc * rl_cs + B];
        f32x4 r = {0.f, 0.f, 0.f, 0.f}, g = r, o = r;
        if (own) { r = *(const f32x4*)(RL + pc * rl_cs + 4 * sub); o = *(const f32x4*)(gRL + pc * rl_cs + 4 * sub); }
        if (own && sub < nqs) g = *(const f32x4*)(gS + pc * s_cs + 4 * sub);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * sub + j < B) { q += g[j] * r[j]; o[j] += g[j] * m; }
        q = grp_sum<LPP>(q);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * sub + j == B) o[j] += q;
        if (live && own) *(f32x4*)(gRL + pc * rl_cs + 4 * sub) = o;
        if (live && sub == 0) gD[pix * d_cs] += q;
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

// the same on float4 (channel counts and strides that are multiples of 4, 16-byte aligned tensors): a quarter of the vector-memory
// instructions, and a thread's two or three loads are independent (the scalar kernel above moves 4 bytes per lane and instruction)
__global__ void mask_axpy_vec_kernel(const float* __restrict__ src, int src_cs, const float* __restrict__ y, int y_cs, int mode,
                                     float* __restrict__ dst, int dst_cs, long npix, int C4, int accumulate)
{
    const long total = npix * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / C4; const int c = (int)(i % C4) * 4;
        f32x4 v = *(const f32x4*)(src + pix * src_cs + c);
        f32x4 yy = {1.f, 1.f, 1.f, 1.f}, d = {0.f, 0.f, 0.f, 0.f};
        if (mode != MASK_NONE) yy = *(const f32x4*)(y + pix * y_cs + c);
        if (accumulate) d = *(const f32x4*)(dst + pix * dst_cs + c);
        if (mode == MASK_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = yy[j] > 0.f ? v[j] : 0.f;
        } else if (mode == MASK_SIGMOID) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= yy[j] * (1.f - yy[j]);
        }
        *(f32x4*)(dst + pix * dst_cs + c) = accumulate ? d + v : v;
    }
}

// adjoint of nearest up-sampling (F.interpolate backward): dst[lo] (+)= sum of src[hi] with src_index(hi) == lo
__global__ void upsample_adjoint_kernel(const float* __restrict__ src, int Hv, int Wv, int src_cs,
                                        float* __restrict__ dst, int Hs, int Ws, int dst_cs, int N, int C,
                                        float sy, float sx, int accumulate,
                                        const float* __restrict__ mask_y, int y_cs, float* __restrict__ dst_masked, int dm_cs)
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
        const long pix = (n * Hs + y) * (long)Ws + x;
        float* d = dst + pix * dst_cs + c;
        s = accumulate ? *d + s : s;
        *d = s;
        // optional second output: the total times relu'(mask_y) (the ReLU-mask launch that used to follow, model.py:157,161)
        if (dst_masked) dst_masked[pix * dm_cs + c] = mask_y[pix * y_cs + c] > 0.f ? s : 0.f;
    }
}

// the same adjoint when the up-sampling factor is an exact integer F (even image sizes: every pyramid level of the 128 x 128
// training patches): src index = floor(dst / F), so dst pixel (y, x) is the sum of the F x F block below it.  One thread per
// (pixel, channel quad), 16-byte accesses; the general kernel above searches a 5 x 5 candidate window per element.
template <int F>
__global__ void upsample_adjoint_exact_kernel(const float* __restrict__ src, int src_cs, float* __restrict__ dst, int Hs, int Ws, int dst_cs,
                                              int N, int C4, int accumulate,
                                              const float* __restrict__ mask_y, int y_cs, float* __restrict__ dst_masked, int dm_cs)
{
    const long total = (long)N * Hs * Ws * C4;
    const int Wv = Ws * F, Hv = Hs * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % C4); long r = i / C4;
        const int x = (int)(r % Ws); r /= Ws;
        const int y = (int)(r % Hs); const long n = r / Hs;
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < F; ++dy)
#pragma unroll
            for (int dx = 0; dx < F; ++dx)
                s4 += *(const f32x4*)(src + ((n * Hv + (long)y * F + dy) * Wv + (long)x * F + dx) * src_cs + 4 * q);
        const long pix = (n * Hs + y) * (long)Ws + x;
        float* d = dst + pix * dst_cs + 4 * q;
        f32x4 yv = {0.f, 0.f, 0.f, 0.f};
        if (dst_masked) yv = *(const f32x4*)(mask_y + pix * y_cs + 4 * q);        // issued with the other loads
        if (accumulate) s4 += *(const f32x4*)d;
        *(f32x4*)d = s4;
        if (dst_masked) {
            f32x4 m;
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = yv[e] > 0.f ? s4[e] : 0.f;
            *(f32x4*)(dst_masked + pix * dm_cs + 4 * q) = m;
        }
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

static void allow_lds(const void* fn, size_t bytes)
{
    if (bytes <= 64 * 1024) return;
    static const void* seen[8][16];                   // hipFuncSetAttribute is per device
    int dev = 0; hipGetDevice(&dev);
    if (dev >= 0 && dev < 16)
        for (int i = 0; i < 8; ++i) {
            if (seen[i][dev] == fn) return;
            if (!seen[i][dev]) { seen[i][dev] = fn; break; }
        }
    hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

int ssie_loss_force_generic = 0;      // include/ssie_debug.h: 1 = always the half-wave-per-pixel kernel (tests run both)

int ssie_loss_chunk_lpp = 8;          // dev switch (tools): lanes per pixel of the band-chunked kernel, 8 (32-band chunks) or 16 (64-band chunks)
extern "C" void ssie_debug_set_loss_chunk_lpp(int v) { ssie_loss_chunk_lpp = v; }
int ssie_loss_chunked = 0;            // include/ssie_debug.h: 1 = the band-chunked tiled kernel also where the one-pass tiled kernel applies

int ssie_launch_loss_direct(const LossParams& p, int nblk, hipStream_t st)
{
    const int nq = p.rl_cs / 4;
    const bool aligned = (((uintptr_t)p.x | (uintptr_t)p.RL | (uintptr_t)p.S | (uintptr_t)p.E | (uintptr_t)p.gRL | (uintptr_t)p.gS |
                           (uintptr_t)p.G8b) & 15) == 0;
    const bool layout = aligned && p.rl_cs == p.e_cs && p.x_cs == p.s_cs && p.rl_cs % 4 == 0 && p.x_cs % 4 == 0 && p.x_cs <= p.rl_cs &&
                        p.B < p.rl_cs;
    const bool tiled = !ssie_loss_force_generic && layout && nq <= 64;
    // more than 252 bands (or on request): chunks of 64 bands, 4 x 16 pixel tiles (54 KB of LDS: two to three workgroups per CU)
    const bool chunked = !ssie_loss_force_generic && layout && p.x_cs >= ssie_round_up(p.B, 4) && (nq > 64 || ssie_loss_chunked);
    if (chunked) {
        // chunks of 32 bands on 8 x 16 pixel tiles (47 KB of LDS: three workgroups per CU, halo 1.4x) or of 64 bands on 4 x 16 tiles
        const int lpp = ssie_loss_chunk_lpp == 16 ? 16 : 8;
        const int TW = 16, TH = lpp == 8 ? LossTile<8>::TH : LossTile<16>::TH;
        const int tiles_y = ssie_ceil_div(p.H, TH), tiles_x = ssie_ceil_div(p.W, TW);
        const size_t lds = (size_t)(TH + 2) * (TW + 2) * (2 * 4 * lpp + 2) * 4;
        if (lpp == 8) { allow_lds((const void*)loss_chunk_kernel<8>, lds); hipLaunchKernelGGL(loss_chunk_kernel<8>, dim3(nblk), dim3(256), lds, st, p, tiles_y, tiles_x); }
        else { allow_lds((const void*)loss_chunk_kernel<16>, lds); hipLaunchKernelGGL(loss_chunk_kernel<16>, dim3(nblk), dim3(256), lds, st, p, tiles_y, tiles_x); }
        return hipGetLastError() == hipSuccess ? 0 : 41;
    }
    if (!tiled) {
        hipLaunchKernelGGL(loss_direct_kernel, dim3(nblk), dim3(256), 0, st, p);
        return hipGetLastError() == hipSuccess ? 0 : 41;
    }
    const int lpp = nq <= 8 ? 8 : nq <= 16 ? 16 : nq <= 32 ? 32 : 64;
    const int TW = lpp <= 16 ? 16 : 8, TH = 4 * (256 / lpp) / TW;          // 4 passes of 256 / LPP pixels (= LossTile<lpp>)
    const int tiles_y = ssie_ceil_div(p.H, TH), tiles_x = ssie_ceil_div(p.W, TW);
    const size_t lds = (size_t)(TH + 2) * (TW + 2) * (2 * p.rl_cs + 1) * 4;
#define LAUNCH_TILE(L) do { allow_lds((const void*)loss_tile_kernel<L>, lds);                                               \
        hipLaunchKernelGGL(loss_tile_kernel<L>, dim3(nblk), dim3(256), lds, st, p, TH, TW, tiles_y, tiles_x); } while (0)
    if (lpp == 8) LAUNCH_TILE(8); else if (lpp == 16) LAUNCH_TILE(16); else if (lpp == 32) LAUNCH_TILE(32); else LAUNCH_TILE(64);
#undef LAUNCH_TILE
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
    const bool al = s_cs % 4 == 0 && rl_cs % 4 == 0 && B < rl_cs && (((uintptr_t)gS | (uintptr_t)RL | (uintptr_t)gRL) & 15) == 0;
    const int nq = rl_cs / 4;
    if (al && nq <= 64) {
#define PN_LAUNCH(L) hipLaunchKernelGGL(product_node_vec_kernel<L>, dim3(grid_for(npix, 256 / L)), dim3(256), 0, st, gS, s_cs, RL, gRL, rl_cs, D, gD, d_cs, npix, B)
        if (nq <= 8) PN_LAUNCH(8); else if (nq <= 16) PN_LAUNCH(16); else if (nq <= 32) PN_LAUNCH(32); else PN_LAUNCH(64);
#undef PN_LAUNCH
        return hipGetLastError() == hipSuccess ? 0 : 43;
    }
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
    if (C % 4 == 0 && src_cs % 4 == 0 && y_cs % 4 == 0 && dst_cs % 4 == 0 && (((uintptr_t)src | (uintptr_t)y | (uintptr_t)dst) & 15) == 0) {
        hipLaunchKernelGGL(mask_axpy_vec_kernel, dim3(grid_for(npix * (C / 4), 256)), dim3(256), 0, st, src, src_cs, y, y_cs, mode, dst, dst_cs, npix, C / 4, accumulate);
        return hipGetLastError() == hipSuccess ? 0 : 46;
    }
    hipLaunchKernelGGL(mask_axpy_kernel, dim3(grid_for(npix * C, 256)), dim3(256), 0, st, src, src_cs, y, y_cs, mode, dst, dst_cs, npix, C, accumulate);
    return hipGetLastError() == hipSuccess ? 0 : 46;
}
int ssie_launch_upsample_adjoint(const float* src, int Hv, int Wv, int src_cs, float* dst, int Hs, int Ws, int dst_cs,
                                 int N, int C, int accumulate, hipStream_t st, const float* mask_y, int y_cs, float* dst_masked, int dm_cs)
{
    if (dst_masked && !mask_y) return 49;
    const bool al = C % 4 == 0 && src_cs % 4 == 0 && dst_cs % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0 &&
                    (!dst_masked || (y_cs % 4 == 0 && dm_cs % 4 == 0 && (((uintptr_t)mask_y | (uintptr_t)dst_masked) & 15) == 0));
    if (al && Hv == 2 * Hs && Wv == 2 * Ws) {
        hipLaunchKernelGGL(upsample_adjoint_exact_kernel<2>, dim3(grid_for((long)N * Hs * Ws * (C / 4), 256)), dim3(256), 0, st, src, src_cs, dst, Hs, Ws, dst_cs, N, C / 4, accumulate, mask_y, y_cs, dst_masked, dm_cs);
        return hipGetLastError() == hipSuccess ? 0 : 47;
    }
    if (al && Hv == 4 * Hs && Wv == 4 * Ws) {
        hipLaunchKernelGGL(upsample_adjoint_exact_kernel<4>, dim3(grid_for((long)N * Hs * Ws * (C / 4), 256)), dim3(256), 0, st, src, src_cs, dst, Hs, Ws, dst_cs, N, C / 4, accumulate, mask_y, y_cs, dst_masked, dm_cs);
        return hipGetLastError() == hipSuccess ? 0 : 47;
    }
    const float sy = (Hs == Hv) ? 1.f : (float)Hs / (float)Hv, sx = (Ws == Wv) ? 1.f : (float)Ws / (float)Wv;
    hipLaunchKernelGGL(upsample_adjoint_kernel, dim3(grid_for((long)N * Hs * Ws * C, 256)), dim3(256), 0, st,
                       src, Hv, Wv, src_cs, dst, Hs, Ws, dst_cs, N, C, sy, sx, accumulate, mask_y, y_cs, dst_masked, dm_cs);
    return hipGetLastError() == hipSuccess ? 0 : 48;
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

// logical (N,C,H,W) fp32 tensor with arbitrary element strides -> dense NHWC bf16 with zero channel padding (cs8 channels per
// pixel, a multiple of 8): the bf16 enhance-only path's ingest + conversion in one pass (the fp32 NHWC copy is not needed there)
__global__ void ingest_bf16_kernel(const float* __restrict__ x, long sn, long sc, long sh, long sw,
                                   uint2* __restrict__ out, int N, int C, int H, int W, int cs8)
{
    const int q4 = cs8 >> 2;                             // 4-channel groups per pixel
    const long total = (long)N * H * W * q4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % q4); long r = i / q4;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); const long n = r / H;
        const float* px = x + n * sn + h * sh + w * sw;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = 4 * q + j; v[j] = c < C ? px[c * sc] : 0.f; }
        out[i] = make_uint2(ssie_pack2bf(v[0], v[1]), ssie_pack2bf(v[2], v[3]));
    }
}

int ssie_launch_ingest_bf16(const float* x, long sn, long sc, long sh, long sw, void* out, int N, int C, int H, int W, int cs8, hipStream_t st)
{
    if (cs8 % 8) return 73;
    const long total = (long)N * H * W * (cs8 / 4);
    long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(ingest_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, sn, sc, sh, sw, (uint2*)out, N, C, H, W, cs8);
    return hipGetLastError() == hipSuccess ? 0 : 74;
}

int ssie_launch_to_bf16(const float* src, void* dst, long n, hipStream_t st)
{
    if (n % 4) return 71;
    const long n4 = n / 4;
    long blocks = (n4 + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const f32x4*)src, (uint2*)dst, n4);
    return hipGetLastError() == hipSuccess ? 0 : 72;
}
