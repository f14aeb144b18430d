// Standalone self-supervised loss operator of the C-ABI (SURVEY §8(b) `selfsup_loss_fwd_bwd`):
// the six loss terms of /root/reference/model.py:551-555 on GIVEN tensors (x, R_low|I_low, I_delta, S, R_enh)
// -> 7 scalars (model.py:566-574) + the 5 direct cotangents dL/d{R_low, I_low, I_delta, S, R_enh}
// (what autograd produces for these leaves in loss.backward(), model.py:315).
// Same kernels as the plan executor (loss_direct_kernel, fft_loss_kernel, loss_finalize_kernel); this entry only exists so that
// the loss arithmetic can be checked elementwise on caller-provided inputs.
#include "loss_kernels.h"
#include "../../include/ssie_hip.h"
#include <string.h>
#include <math.h>

namespace {
const int kLossBlocks = 2048;
size_t up64(size_t v) { return (v + 63) / 64 * 64; }
struct Layout { size_t lpart, fpart, cf, fftws, total; };     // float offsets
Layout layout(int N, int B, int H, int W)
{
    Layout l;
    size_t o = 0;
    l.lpart = o; o = up64(o + (size_t)kLossBlocks * 8);
    l.fpart = o; o = up64(o + (size_t)ssie_fft_partials(N, B, H, W));
    l.cf = o; o = up64(o + 8);
    l.fftws = o; o = up64(o + ssie_fft_workspace_floats(N, B, H, W));
    l.total = o;
    return l;
}
}

extern int ssie_loss_force_generic;
extern "C" void ssie_debug_set_loss_generic(int v) { ssie_loss_force_generic = v; }
extern int ssie_loss_chunked;
extern "C" void ssie_debug_set_loss_chunked(int v) { ssie_loss_chunked = v; }

extern "C" size_t ssie_selfsup_loss_workspace_bytes(int N, int bands, int H, int W)
{
    if (N < 1 || bands < 2 || H < 2 || W < 2) return 0;
    return layout(N, bands, H, W).total * 4;
}

extern "C" int ssie_selfsup_loss_fwd_bwd(const float* x, int x_cs, const float* RL, int rl_cs, const float* D, int d_cs,
                                         const float* S, int s_cs, const float* E, int e_cs,
                                         int N, int bands, int H, int W, const float* coefs8, const uint8_t* fourier_mask_dev,
                                         float* gRL, float* gD, float* gS, float* gE, float* scalars7,
                                         void* ws, size_t ws_bytes, void* stream)
{
    if (!x || !RL || !D || !S || !E || !coefs8 || !fourier_mask_dev || !gRL || !gD || !gS || !gE || !scalars7 || !ws) return SSIE_E_ARG;
    if (N < 1 || bands < 2 || H < 2 || W < 2) return SSIE_E_SHAPE;
    if (x_cs < bands || s_cs < bands || rl_cs < bands + 1 || e_cs < bands + 1 || d_cs < 1) return SSIE_E_SHAPE;
    if (!ssie_fft_supported(H, W)) return SSIE_E_SHAPE;
    const Layout l = layout(N, bands, H, W);
    if (ws_bytes < l.total * 4) return SSIE_E_WORKSPACE;
    float* w = (float*)ws;
    hipStream_t st = (hipStream_t)stream;
    LossParams lp; memset(&lp, 0, sizeof(lp));
    lp.x = x; lp.x_cs = x_cs; lp.RL = RL; lp.rl_cs = rl_cs; lp.D = D; lp.d_cs = d_cs; lp.S = S; lp.s_cs = s_cs; lp.E = E; lp.e_cs = e_cs;
    lp.gRL = gRL; lp.gD = gD; lp.gS = gS; lp.G8b = gE; lp.ge_raw = 1;
    lp.N = N; lp.H = H; lp.W = W; lp.B = bands;
    lp.c_rec = coefs8[0]; lp.c_rf = coefs8[1]; lp.c_il = coefs8[2]; lp.c_id = coefs8[3]; lp.c_sp = coefs8[5];
    lp.a1 = coefs8[6]; lp.a2 = coefs8[7];
    const double n = N, c = bands, h = H, wd = W;
    lp.inv_n0 = (float)(1.0 / (n * c * h * wd)); lp.inv_nIx = (float)(1.0 / (n * h * (wd - 1))); lp.inv_nIy = (float)(1.0 / (n * (h - 1) * wd));
    lp.inv_nRx = (float)(1.0 / (n * c * h * (wd - 1))); lp.inv_nRy = (float)(1.0 / (n * c * (h - 1) * wd));
    lp.inv_nsp = (float)(1.0 / (n * (c - 1) * h * wd));
    lp.partials = w + l.lpart;
    if (ssie_launch_loss_direct(lp, kLossBlocks, st)) return SSIE_E_LAUNCH;
    FftParams fp; memset(&fp, 0, sizeof(fp));
    fp.x = x; fp.x_cs = x_cs; fp.S = S; fp.s_cs = s_cs; fp.gS = gS; fp.mask = fourier_mask_dev;
    fp.N = N; fp.B = bands; fp.H = H; fp.W = W;
    ssie_fft_set_logs(fp);
    fp.scale_g = (float)(coefs8[4] / (n * c * h * wd)); fp.inv_n0 = lp.inv_n0; fp.partials = w + l.fpart; fp.ws = w + l.fftws;
    fp.ws_floats = l.total - l.fftws; fp.npartials = (int)(l.cf - l.fpart);
    if (ssie_launch_fft_loss(fp, st)) return SSIE_E_LAUNCH;
    const float cf[6] = {coefs8[0], coefs8[1], coefs8[2], coefs8[3], coefs8[4], coefs8[5]};
    if (ssie_launch_loss_finalize(w + l.lpart, kLossBlocks, w + l.fpart, ssie_fft_partials(N, bands, H, W), cf, scalars7, st)) return SSIE_E_LAUNCH;
    return 0;
}
