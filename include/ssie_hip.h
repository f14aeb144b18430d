/* C-ABI of libssie_hip.so — the MI355X (gfx950) hot path of SS-HSLIE.
 *
 * The reference (medemirhan/Self-supervised-Image-Enhancement-Network-Training-With-Low-Light-Images-Only)
 * has no FFI of its own: its hot path is Python calling torch operators.  Each entry point below
 * therefore names the reference call site(s) it replaces (paths relative to /root/reference).
 * A maintainer binds them with ctypes; the package's own binding is
 * `<pkg>/hostlib.py`.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless stated otherwise; tensors are NHWC
 *     ("band-innermost"); `cstride` = floats per pixel of the buffer, `coff` = first channel used;
 *     cstride, coff and channel counts of multi-source inputs are multiples of 4 (16-byte vectors)
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work
 *   - functions never allocate device memory and never synchronise; scratch comes from the caller
 *     (`ws`, sized by the matching *_workspace_bytes query)
 *   - return value: 0 on success, >0 = argument/shape error (nothing was launched), see SSIE_E_*
 *   - thread-compatible: one plan / one workspace must not be used from two threads at once
 */
#ifndef SSIE_HIP_H
#define SSIE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSIE_E_OK 0
#define SSIE_E_ARG 1        /* null pointer / bad enum */
#define SSIE_E_SHAPE 2      /* unsupported shape or alignment */
#define SSIE_E_WORKSPACE 3  /* workspace too small */
#define SSIE_E_LAUNCH 4     /* HIP launch error */

#define SSIE_ACT_NONE 0
#define SSIE_ACT_RELU 1
#define SSIE_ACT_SIGMOID 2

/* one channel-slice of a virtual input: concat-by-pointer + nearest up-sampling on read
 * (replaces torch.cat model.py:59,63,146,172 and F.interpolate model.py:156-169) */
typedef struct {
    const float* ptr;
    int C, cstride, coff;
    int Hs, Ws;               /* physical size; the op's (Hv, Wv) is the virtual (up-sampled) size */
} ssie_src_t;

const char* ssie_version(void);
int ssie_device_ok(void);     /* 1 when the current HIP device is gfx950 */

/* ---- granular operators (used by the parity tests; the plan below uses the same kernels) ---- */
size_t ssie_op_workspace_bytes(int cin, int cout, int k);

/* nn.Conv2d(+bias)(+ReLU|sigmoid)(+skip add): model.py:17-23, :47, :140-141; nn.Linear as k=1: model.py:93-97
 * weight OIHW with cin_w input channels (<= sum of source C; extra source channels are zero padding);
 * padding (k-1)/2; stride 1 or 2 (k=3).  out2 (optional) receives act(v) before the skip add. */
int ssie_conv2d_fwd(const ssie_src_t* srcs, int nsrc, int N, int Hv, int Wv,
                    const float* weight, int cin_w, const float* bias, int cout, int k, int stride, int act,
                    const float* addsrc, float* out2, float* out, int out_cstride, int out_coff,
                    void* ws, size_t ws_bytes, void* stream);

/* nn.ConvTranspose2d(k=3, stride=2, padding=1, output_padding=1)(+ReLU): model.py:39-43; weight (in,out,k,k) */
int ssie_conv_transpose2d_fwd(const ssie_src_t* src, int N, const float* weight, const float* bias, int cout,
                              int act, float* out, int out_cstride, int out_coff,
                              void* ws, size_t ws_bytes, void* stream);

/* autograd backward of the above (model.py:315): data gradient w.r.t. input channels
 * [ci_off, ci_off+cs) of a layer with cin_total inputs.  g = gradient w.r.t. the conv output
 * (N,Ho,Wo,cout).  gx has the input's size (Hin,Win); optional mask multiplies by act'(mask_y)
 * (mask_mode 1 = ReLU: y>0, 2 = sigmoid: y(1-y)); accumulate adds into gx. */
int ssie_conv2d_dgrad(const float* g, int g_cstride, int g_coff, int N, int Ho, int Wo, int cout,
                      const float* weight, int cin_total, int ci_off, int cs, int k, int stride,
                      float* gx, int Hin, int Win, int gx_cstride, int gx_coff,
                      const float* mask_y, int mask_mode, int accumulate,
                      void* ws, size_t ws_bytes, void* stream);
int ssie_conv_transpose2d_dgrad(const float* g, int g_cstride, int g_coff, int N, int Hin, int Win, int cout,
                                const float* weight, int cin,
                                float* gx, int gx_cstride, int gx_coff,
                                const float* mask_y, int mask_mode, int accumulate,
                                void* ws, size_t ws_bytes, void* stream);

/* weight / bias gradient; dw in the parameter's own layout (OIHW, or (in,out,k,k) for the transposed
 * conv); db may be NULL */
int ssie_conv2d_wgrad(const ssie_src_t* src, int N, int Hv, int Wv,
                      const float* g, int g_cstride, int g_coff, int cout, int k, int stride,
                      int cin_total, int ci_off, float* dw, float* db, int accumulate,
                      void* ws, size_t ws_bytes, void* stream);
int ssie_conv_transpose2d_wgrad(const ssie_src_t* x, int N, const float* g, int g_cstride, int g_coff, int cout,
                                float* dw, float* db, int accumulate,
                                void* ws, size_t ws_bytes, void* stream);

/* 4-head x 16-dim softmax self-attention over T tokens (model.py:107-114) and its backward.
 * qkv: (N*T, 192) rows = tokens, q|k|v at channel 0|64|128; out/gout: (N*T, 64); lse, delta_ws: (N,4,T) */
int ssie_attention_fwd(const float* qkv, float* out, float* lse, int N, int T, void* stream);
int ssie_attention_bwd(const float* qkv, const float* out, const float* gout, const float* lse,
                       float* delta_ws, float* gqkv, int N, int T, void* stream);

/* device-side batch assembly = the host loop of model.py:301-310 + utils.data_augmentation (utils.py:7-34):
 * crops_dev = n records {const float* cube (H,W,C fp32 on device); int H, W, x0 (row), y0 (col), mode 0..7};
 * out = (n, P, P, cs) NHWC patches, channels >= C zero-padded */
int ssie_assemble_batch(const void* crops_dev, int n, float* out, int P, int C, int cs, void* stream);
int ssie_aug_source_index(int mode, int P, int i, int j, int* si, int* sj);   /* host: index map of the 8 augmentations */

/* ---- plan executor: the whole hot path as a static launch schedule -----------------------------
 * coefs8 = {c_loss_reconstruction, c_loss_r_fidelity, c_loss_i_smooth_low, c_loss_i_smooth_delta,
 *           c_loss_fourier, c_loss_spectral_cons, alpha_i_smooth_low, alpha_i_smooth_delta}
 * (LowLightEnhance.__init__ keywords, model.py:178-182).  H and W must be even (model.py:59) and >= 8.
 * Returns NULL on an unsupported shape. */
void* ssie_plan_create(int N, int bands, int H, int W, const float* coefs8);
void ssie_plan_destroy(void* plan);
size_t ssie_plan_workspace_bytes(void* plan);
int ssie_plan_set_coefs(void* plan, const float* coefs8);

/* flat parameter buffer layout = the reference's state_dict order (46 tensors, model.py:595-607);
 * offsets in floats, each tensor 16-byte aligned */
size_t ssie_plan_param_floats(void* plan);
int ssie_plan_num_params(void* plan);
int ssie_plan_param_info(void* plan, int idx, char* name, int name_cap, size_t* off_floats, int* ndim, int* shape4);

/* named activation / gradient buffers inside the workspace (NHWC): dims5 = {N, H, W, C, cstride}.
 * "RL_1" = sigmoid(recon): R_low = channels [0,bands), I_low = channel bands; "D" = I_delta; "S";
 * "RL_2" = decomposition of S; "scalars" = {total, L_reconstruction, L_R_fidelity, L_I_smooth_low,
 * L_I_smooth_delta, L_fourier, L_spectral_cons} (model.py:566-574) */
int ssie_plan_buffer(void* plan, const char* name, size_t* off_floats, int* dims5);

/* bind to caller-owned device memory (workspace 256-byte aligned; grads may be NULL for inference).
 * Zeroes the workspace, uploads descriptors + the Fourier mask, synchronises `stream` once. */
int ssie_plan_bind(void* plan, void* workspace, size_t ws_bytes, float* params, float* grads, void* stream);

/* LowLightEnhance.forward (model.py:229-234): x is the logical (N,bands,H,W) fp32 tensor with element
 * strides strides4 = {sN, sC, sH, sW} (channels_last or contiguous alike) */
int ssie_plan_enhance_fwd(void* plan, const float* x, const long* strides4, void* stream);
/* same outputs, computed with bf16 storage + bf16 MFMA (fp32 accumulate): the mixed-precision inference path of
 * BASELINE.json configs[4]; no counterpart in the reference (fp32 only, model.py:229-234) */
int ssie_plan_enhance_fwd_bf16(void* plan, const float* x, const long* strides4, void* stream);

/* compute_loss (+ loss.backward() when with_backward != 0): model.py:544-575, :315.  Writes "scalars";
 * with_backward also zeroes and fills the flat gradient buffer (zero_grad, model.py:313).
 * Any even patch size (model.py:456-473 takes any patch_size): the Fourier term keeps one H x (W+1) complex plane in LDS when
 * it fits (<= 128 x 128; power-of-two sizes radix-2, other sizes up to 192 per side a direct DFT) and otherwise runs as three
 * passes (rows, columns, rows) over a half-spectrum workspace inside the plan workspace. */
int ssie_plan_loss_fwd_bwd(void* plan, const float* x, const long* strides4, int with_backward, void* stream);

/* Replay everything ssie_plan_loss_fwd_bwd (with backward) enqueues behind the input conversion as ONE hipGraph: the op list of a
 * plan is fixed and touches only the plan's own buffers, so it is captured once (second call) and replayed afterwards - same
 * kernels, same order, bit-identical results, a fifth of the host time per step (0.39 -> 0.08 ms at batch 32 of 128x128x31).  The
 * reference has no counterpart (its step is eager autograd, model.py:313-316).  Rebinding or new loss coefficients drop the graph. */
int ssie_plan_set_graph(void* plan, int on);

/* ---- standalone self-supervised loss operator (SURVEY §8(b) `selfsup_loss_fwd_bwd`) ------------------------------
 * The six loss terms of model.py:551-555 on GIVEN tensors and their direct cotangents (what loss.backward(), model.py:315,
 * hands to these five leaves): smooth_loss :450-454, fourier_spectrum_loss :456-473, spectral_smoothness_loss :475-481,
 * structure_aware_loss :491-542, L_reconstruction :551.  All tensors NHWC with their own `*_cs` floats per pixel:
 *   x, S (N,H,W,>=bands); RL = R_low in channels [0,bands) and I_low in channel `bands`; D = I_delta in channel 0;
 *   E = R_enh in channels [0,bands) (buffer has > bands channels like RL).
 * Outputs: gRL (geometry of RL: dL/dR_low | dL/dI_low), gD (geometry of D), gS (geometry of S), gE (geometry of E: dL/dR_enh,
 * channel `bands` = 0), scalars7 = {total, L_reconstruction, L_R_fidelity, L_I_smooth_low, L_I_smooth_delta, L_fourier,
 * L_spectral_cons} (device).  fourier_mask_dev = H*W bytes on the device, from ssie_fourier_mask(). */
size_t ssie_selfsup_loss_workspace_bytes(int N, int bands, int H, int W);
int ssie_selfsup_loss_fwd_bwd(const float* x, int x_cs, const float* RL, int rl_cs, const float* D, int d_cs,
                              const float* S, int s_cs, const float* E, int e_cs,
                              int N, int bands, int H, int W, const float* coefs8, const uint8_t* fourier_mask_dev,
                              float* gRL, float* gD, float* gS, float* gE, float* scalars7,
                              void* ws, size_t ws_bytes, void* stream);

/* one compute_loss+backward with a HIP event after every launch: device milliseconds, algorithmic FLOPs and
 * launch counts per kernel class (synchronises; used by bench.py's roofline leg).  Arrays have 14 entries:
 * {conv_fprop<64>, conv_fprop<32>, conv_wgrad, wgrad_reduce, colsum, pack, loss, fft_loss, attention, elementwise,
 *  spectral 9x9 conv, Winograd F(2x2,3x3) conv, Winograd 3x3 weight gradient, Winograd F(4x4,3x3) conv}.  The last four are counted
 * with the FLOPs of the direct convolution they replace (the matrix pipe executes 16/36, 16/36 and 36/144 of them). */
#define SSIE_NKINDS 14
int ssie_plan_profile_step(void* plan, const float* x, const long* strides4, void* stream,
                           double* ms, double* flops, int* counts);

/* torch.optim.Adam.step with default hyper-parameters (model.py:213, :316) over flat buffers;
 * grads are multiplied by grad_scale first (1/world_size after an all-reduce-sum) */
int ssie_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                   float grad_scale, float lr, int step, float beta1, float beta2, float eps, void* stream);

/* host helper: the reference's radial Fourier mask (model.py:460-464), float32-exact; out_host = H*W bytes */
int ssie_fourier_mask(int H, int W, float cutoff, uint8_t* out_host);

#ifdef __cplusplus
}
#endif
#endif
