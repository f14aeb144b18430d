/* Test / development entry points of libssie_hip.so.  NOT part of the drop-in boundary (include/ssie_hip.h): nothing in the
 * product path calls these.  They exist so that `tests/` can force a specific kernel variant through the same parity cases,
 * run the backward schedule on injected cotangents, and so that `tools/` can time single launches.
 * The setters change process-global launch heuristics; tests restore the defaults (given in brackets) afterwards. */
#ifndef SSIE_DEBUG_H
#define SSIE_DEBUG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* backward schedule only (everything ssie_plan_loss_fwd_bwd runs after the three loss kernels), on cotangents the caller
 * wrote into the plan buffers "gRL", "gD", "gS", "G8_2" after a ssie_plan_loss_fwd_bwd call filled the activations */
int ssie_plan_backward_from_cotangents(void* plan, void* stream);

/* per-launch device ms / algorithmic FLOPs / kind in launch order of one train step; returns the op count (or -error) */
int ssie_plan_profile_ops(void* plan, const float* x, const long* strides4, void* stream,
                          double* ms, double* flops, int* kinds, int cap, char* tags, int tags_cap);
/* per-launch timing of one op list; which = 0 fp32 enhance forward, 1 = bf16 enhance forward */
int ssie_plan_profile_list(void* plan, const float* x, const long* strides4, void* stream, int which,
                           double* ms, double* flops, int* kinds, int cap, char* tags, int tags_cap);
/* algorithmic HBM bytes per launch (each operand read once, each result written once; 0 where none is stated) in the order of the
 * matching profile call: which = 0 / 1 = the lists of ssie_plan_profile_list, 2 = the train step of ssie_plan_profile_ops; returns
 * the op count (or -error).  ssie_plan_class_bytes: the same summed per kernel class of the train step (SSIE_NKINDS entries) */
int ssie_plan_op_bytes(void* plan, int which, double* bytes, int cap);
int ssie_plan_class_bytes(void* plan, double* bytes);
/* launches per op list: {enhance forward, second decomposition pass, loss + backward} */
int ssie_plan_num_ops(void* plan, int* counts3);

/* the bf16 attention of the enhance-only path on its own: qkv fp32 (N, T, 192) -> out bf16 (N, T, 64); scratch (optional, from
 * ..._scratch_bytes) enables the pre-converted key / value path the plan uses from 256 tokens on */
size_t ssie_debug_attention_bf16_scratch_bytes(int N, int T);
int ssie_debug_attention_fwd_bf16(const float* qkv, void* out_bf16, int N, int T, void* scratch, size_t scratch_bytes, void* stream);

/* launch heuristics */
void ssie_debug_set_overlap(int on);                    /* [0] 1 = slab reductions of the weight gradients on a side stream; 0 = launch order on the caller's stream */
void ssie_debug_set_graph(int on);                      /* [0] 1 = ssie_plan_loss_fwd_bwd (with backward) replays one hipGraph per plan behind the ingest */
void ssie_debug_set_fprop_min_tiles16(int v);           /* [256] launches with fewer 16x16 tiles use the 8x16 register-staged kernel */
void ssie_debug_set_fprop_wide(int v);                  /* [1] 0 = no 16x32 tiles */
void ssie_debug_set_fprop_wide_min_tiles(int v);        /* [512] */
void ssie_debug_set_wgrad_reduce_wide_min(int v);       /* [64] slice count from which the slab reduction runs 16 slice groups per block (a huge value = never) */
void ssie_debug_set_fft_chunk_mb(int mb);                /* [192] workspace chunk (MiB) of the three-pass Fourier loss (plans created afterwards) */
void ssie_debug_set_attn_bf16_prepass(int v);           /* [1] 0 = the bf16 attention converts K / V per workgroup instead of once */
void ssie_debug_set_bf16_conv9(int v);                  /* [1] 0 = the bf16 9x9 layer stays on the generic kernel */
void ssie_debug_set_bf16_conv9_min_tiles(int v);        /* [256] */
void ssie_debug_set_bf16_ws_geo(int v);                 /* [1] 0 = bf16 stride-2 / transposed 64-channel layers stay on the eight-wave kernel */
void ssie_debug_set_bf16_ws_geo_min_tiles(int v);       /* [256] */
void ssie_debug_set_bf16_resw(int v);                   /* [1] bf16 single-source 3x3 layers of <= 64 input channels keep their packed weights in LDS (1 = 8 consumer waves, 2 = 4, 0 = off) */
void ssie_debug_set_bf16_ws(int v);                     /* [3] bf16 16x32 layers: 0 = eight-wave kernel, 1 / 3 = wave-specialised (8 / 4 consumer waves + 4 producer waves), 2 = DMA interleaved between taps */
void ssie_debug_set_fprop_tile16(int v);                /* [1] */
void ssie_debug_set_fprop_v2(int v);                    /* [1] 0 = never the 512-thread DMA kernel */
void ssie_debug_set_fprop_v2_stride2(int v);            /* [1] */
void ssie_debug_set_fprop_v2_split(int on);             /* [1] 0 = one 8-wave workgroup per CU for 32-channel layers too */
void ssie_debug_set_fprop_v2_split_min_tiles(int v);    /* [1024] */
void ssie_debug_set_fprop_wgs_per_cu(int v);
void ssie_debug_set_tconv(int v);                       /* [1] 0 = stride-2 transposed 3x3 convolutions always as four output-parity launches (plans created afterwards) */
void ssie_debug_set_tconv_min_tiles(int v);             /* [8] fewest 16x16 input tiles for the one-launch kernel; v < 0 = the default */
void ssie_debug_set_fft_grouped(int v);                 /* [1] 0 = Fourier loss planes that fit the LDS always run the whole-plane kernel (fft_loss_kernel); plans / operator calls made afterwards */
void ssie_debug_set_loss_chunk_lpp(int v);              /* [8] lanes per pixel of loss_chunk_kernel: 8 = 32-band chunks on 8x16 tiles, 16 = 64-band chunks on 4x16 tiles */
void ssie_debug_set_loss_chunked(int v);                /* [0] 1 = the band-chunked tiled loss kernel (loss_chunk_kernel, normally only above 252 bands) for every band count */
void ssie_debug_set_wino(int v);                        /* [1] 0 = stride-1 3x3 launches never run the Winograd F(2x2,3x3) kernel (plans created afterwards) */
void ssie_debug_set_wino4(int v);                       /* [1] 0 = no stride-1 3x3 launch runs the Winograd F(4x4,3x3) kernel (conv_wino4.hip); plans created afterwards */
void ssie_debug_set_wino4_min_tiles(int v);             /* [256] fewest 16x64x32-channel tiles for which it is chosen (tests: 1 = wherever eligible, 1 << 30 = never, < 0 = restore the default) */
void ssie_debug_set_wino_min_tiles(int v);               /* [32] fewest 16x32x32-channel tiles for which it is chosen (< 0 = restore the default) */
void ssie_debug_set_wgrad_wino(int v);                  /* [1] 0 = stride-1 3x3 weight gradients never run the Winograd F(3x3,2x2) kernel (plans created afterwards) */
void ssie_debug_set_wgrad_wino_min_tiles(int v);        /* [64] fewest 8x16 position tiles for which it is chosen; v < 0 = the default */
void ssie_debug_set_fused_tail(int on);                 /* [1] 0 = inference keeps feature_fusion / final_conv / compose as separate launches (plans bound afterwards) */
void ssie_debug_set_spectral9(int on);                  /* [1] 0 = the 9 x 9 convolution (shallow_conv) on the direct MFMA kernels instead of the frequency domain (plans created afterwards) */
void ssie_debug_set_fold_masks(int on);                 /* [1] 0 = the backward's ReLU / sigmoid masks as separate mask_axpy launches instead of second outputs of the producing launches (plans created afterwards) */
void ssie_debug_set_batched_reduce(int on);             /* [1] 0 = one weight-gradient slab reduction per layer right behind its producer instead of ONE batched launch at the end of the backward pass (plans created afterwards; bit-identical either way) */
void ssie_debug_set_tconv_split_below(int v);          /* [256] whole-tile 8-row launches of the one-launch transposed convolution with fewer tiles than this run 32-channel workgroups (0 = never) */
void ssie_debug_set_wino_half_below(int v);            /* [256] whole-tile Winograd F(2x2,3x3) launches with fewer 32-channel tiles than this run 16-channel workgroups (0 = never) */
void ssie_debug_set_fprop_v2_onetap(int on);            /* [1] 0 = 1 x 1 layers on the general 16 x 16-tile instantiation (one workgroup per CU) instead of the 1-tap one (40 KB of LDS, two per CU) */
void ssie_debug_set_tconv_half_tiles_below(int v);     /* [257] the one-launch transposed convolution takes 8-row tiles when its 16-row tiles would number fewer than this (0 = always 16 rows) */
void ssie_debug_set_bf16_two_wgs(int on);               /* [1] 0 = one workgroup per CU for the bf16 32-channel-tile convolution (two fit: 2 x 77 KB LDS) */
void ssie_debug_set_qkv_fused(int on);                  /* [1] 0 = q_linear / k_linear / v_linear as three 64 -> 64 launches each way instead of one 64 -> 192 layer (plans created afterwards) */
void ssie_debug_set_skinny_final(int on);               /* [1] 0 = final_conv (64 -> 1) forward / gradients on the MFMA tile kernels (plans created afterwards) */
void ssie_debug_set_loss_generic(int v);                 /* [0] 1 = the half-wave-per-pixel loss kernel instead of the tiled one */
void ssie_debug_set_wgrad_sliding(int v);               /* [1] 0 = generic wgrad K loop everywhere */
void ssie_debug_set_wgrad_rows2(int v);                 /* [1] 0 = one 9x9 kernel row per workgroup (gradient tile re-read 9x instead of 5x) */
/* (diagnostic builds compiled with -DSSIE_STAMP additionally export two s_memtime stamp-buffer setters, see tools/stamp_*.py;
 * the shipped library has no stamp code) */

#ifdef __cplusplus
}
#endif
#endif
