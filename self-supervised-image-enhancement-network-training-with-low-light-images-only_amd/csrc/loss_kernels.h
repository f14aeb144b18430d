// Parameter blocks + host launchers of the HBM-bound loss / elementwise kernels.
#pragma once
#include "ssie_common.h"

struct LossParams {
    const float* x; int x_cs;        // input_low, NHWC padded
    const float* RL; int rl_cs;      // pass-1 sigmoid output: R = ch [0,B), I_low = ch B
    const float* D; int d_cs;        // I_delta at ch 0
    const float* S; int s_cs;        // enhanced cube
    const float* E; int e_cs;        // pass-2 sigmoid output (R_enh = ch [0,B))
    float* gRL;                      // dL/d(R, I_low)           (geometry of RL)
    float* gD;                       // dL/dI_delta              (geometry of D)
    float* gS;                       // dL/dS direct terms       (geometry of S)
    float* G8b;                      // dL/d(pre-sigmoid) of pass 2 (geometry of E)
    int N, H, W, B;
    float c_rec, c_rf, c_il, c_id, c_sp, a1, a2;
    float inv_n0, inv_nIx, inv_nIy, inv_nRx, inv_nRy, inv_nsp;
    float* partials;                 // [nblk][8]
    int ge_raw;                      // 1: G8b receives dL/dR_enh itself (standalone operator); 0: times E(1-E), i.e. w.r.t. pass 2's pre-sigmoid output
};

struct FftParams {
    const float* x; int x_cs;
    const float* S; int s_cs;
    float* gS;                       // += c_f * Re(H*W*ifft2(M * g_Z))
    const uint8_t* mask;             // [H][W], unshifted layout (model.py:460-464)
    int N, B, H, W, logH, logW;
    float scale_g;                   // c_f / (N*B*H*W)
    float inv_n0;
    float* partials;                 // [ssie_fft_partials(N, B, H, W)]
    float* ws;                       // three-pass path only: ssie_fft_workspace_floats(N, B, H, W) floats
    int path;                        // ssie_fft_path(N, B, H, W) at the time ssie_fft_set_logs ran: 1 / 2 whole plane in LDS, 3 three passes
    // capacities of `ws` / `partials` as ALLOCATED: the launch geometry follows process-global development switches (chunk size,
    // grouped rows), so a launch whose geometry no longer fits what was allocated under earlier settings is refused, not run
    size_t ws_floats; int npartials;
};

int ssie_launch_loss_direct(const LossParams& p, int nblk, hipStream_t st);
int ssie_launch_loss_finalize(const float* partials, int nblk, const float* fpartials, int nfblk,
                              const float* coefs6, float* out, hipStream_t st);
int ssie_launch_product_node(const float* gS, int s_cs, const float* RL, float* gRL, int rl_cs, const float* D, float* gD,
                             int d_cs, long npix, int B, hipStream_t st);
int ssie_launch_compose(const float* RL, int rl_cs, const float* D, int d_cs, float* S, int s_cs, long npix, int B, hipStream_t st);
// fp32 -> bf16 (round to nearest even), n elements, n % 4 == 0 (inference path: input cube, attention output)
int ssie_launch_to_bf16(const float* src, void* dst, long n, hipStream_t st);
int ssie_launch_ingest_bf16(const float* x, long sn, long sc, long sh, long sw, void* out, int N, int C, int H, int W, int cs8, hipStream_t st);
int ssie_launch_ingest(const float* x, long sn, long sc, long sh, long sw, float* out, int N, int C, int H, int W, int cs, hipStream_t st);
int ssie_launch_mask_axpy(const float* src, int src_cs, const float* y, int y_cs, int mode, float* dst, int dst_cs,
                          long npix, int C, int accumulate, hipStream_t st);
// optional second output: dst_masked = relu'(mask_y) * (the total written to dst), same geometry as dst (the ReLU-mask launch folded in)
int ssie_launch_upsample_adjoint(const float* src, int Hv, int Wv, int src_cs, float* dst, int Hs, int Ws, int dst_cs,
                                 int N, int C, int accumulate, hipStream_t st,
                                 const float* mask_y = nullptr, int y_cs = 0, float* dst_masked = nullptr, int dm_cs = 0);
int ssie_launch_adam(float* p, const float* g, float* m, float* v, long n, float gscale, float lr, int step,
                     float b1, float b2, float eps, hipStream_t st);
int ssie_fft_supported(int H, int W);
int ssie_fft_path(int N, int B, int H, int W);      // the path taken: as ssie_fft_supported, or 3 where the band-grouped three-pass path applies
int ssie_fft_grid(int N, int B);
int ssie_fft_partials(int N, int B, int H, int W);
size_t ssie_fft_workspace_floats(int N, int B, int H, int W);
void ssie_fft_set_logs(FftParams& p);        // fills logH / logW for the path ssie_fft_supported(H, W) selects
int ssie_launch_fft_loss(const FftParams& p, hipStream_t st);
void ssie_fourier_mask_host(int H, int W, float cutoff, uint8_t* out);

// fused inference tail (tail_kernels.hip): feature_fusion + final_conv + S = R*(I_delta + I_low) in one launch
int ssie_tail_supported(int H, int W, int H2, int W2, int H4, int W4);
size_t ssie_tail_weight_floats();     // floats of the composite-weight buffer (fp32 weights + the bf16 MFMA operand images)
int ssie_launch_tail_weights(const float* wf, const float* bf, const float* wl, const float* bl, float* out, hipStream_t st);
int ssie_launch_tail(const void* d1, const void* d2, const void* d3, int bf16_in, int N, int H, int W, int H2, int W2, int H4, int W4,
                     const float* wc, const float* RL, int rl_cs, float* D, int d_cs, float* S, int s_cs, int B, hipStream_t st);
// final_conv (3x3, 64 -> 1) of the training step as VALU kernels (tail_kernels.hip); f / Gf are (N,H,W,64) with 64 floats per pixel
int ssie_launch_skinny_fwd(const float* f, const float* w, const float* bias, float* D, int d_cs, int N, int H, int W, hipStream_t st);
int ssie_launch_skinny_dgrad(const float* gD, int d_cs, const float* w, float* Gf, int N, int H, int W, hipStream_t st);
size_t ssie_skinny_wgrad_ws_floats();
int ssie_launch_skinny_wgrad(const float* f, const float* gD, int d_cs, float* part, float* dw, float* db, int N, int H, int W, hipStream_t st);
