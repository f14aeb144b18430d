// Shared host/device declarations for the SS-HSLIE MI355X (gfx950) hot path.
//
// Data layout (all device tensors): fp32, NHWC ("band-innermost"), the channel count of
// every buffer padded to a multiple of 4 so that each pixel's band vector is 16-byte
// aligned.  The reference hands the model channels_last tensors (model.py:301,312), so the
// band axis is already the fastest-moving one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SSIE_MAX_SRC 3
#define SSIE_MAX_TAPS 81
#define SSIE_TG 9            // taps staged per weight-group (one 3x3 kernel, or one row of the 9x9)
#define SSIE_CK 16           // input channels per K-chunk
#define SSIE_TH 8            // output tile rows
#define SSIE_TW 16           // output tile cols

// fp32 -> bf16, round to nearest even, as a PLAIN CAST: hipcc emits v_cvt_pk_bf16_f32, which keeps a NaN a NaN (integer
// rounding on the bit pattern turns some NaNs into 0 or infinity - MI355X_MICROARCH.md "Correctness boundaries")
typedef __bf16 ssie_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned short ssie_f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ unsigned ssie_pack2bf(float lo, float hi)
{
    ssie_bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// One source of a "virtual input" tensor: channel-concatenation by pointer, nearest
// up-sampling on read (F.interpolate(mode='nearest'), model.py:156-169) and zero padding
// are all resolved while staging the LDS halo tile, so no cat/interpolate copy ever exists.
struct SrcDesc {
    const float* ptr;
    int C;         // channels taken from this source (multiple of 16 when nsrc > 1)
    int cstride;   // floats per pixel in the buffer
    int coff;      // first channel inside the pixel
    int Hs, Ws;    // physical spatial size
    float sy, sx;  // nearest scale: src = min((int)floorf(v * s), Hs - 1); 1.0f = identity
};

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };
enum { MASK_NONE = 0, MASK_RELU = 1, MASK_SIGMOID = 2 };   // multiply by act'(y) given stored y

struct ConvParams {
    SrcDesc src[SSIE_MAX_SRC];
    int nsrc;
    int N, Hv, Wv;          // virtual input size
    int Cin;                // total virtual channels
    int nchunks;            // ceil(Cin / 16)
    int Ho, Wo;             // output-position grid enumerated by the tiles
    int si;                 // input stride of the position grid (1 or 2)
    int ntaps;
    int min_dy, min_dx;     // halo origin relative to (a*si, b*si)
    int hp_h, hp_w;         // halo tile size in pixels
    int8_t tap_dy[SSIE_MAX_TAPS];
    int8_t tap_dx[SSIE_MAX_TAPS];
    const float* wpacked;   // [chunk][tap][kq*2+h][Cout_pad][4]
    int Cout, Cout_pad;
    float* out;             // output tensor
    int out_cstride, out_coff;
    int Hout, Wout;         // physical output size
    int so, py, px;         // output position = (a*so + py, b*so + px)
    const float* bias;      // may be null
    int act;
    const float* addsrc;    // optional skip tensor (same geometry as out): out = act(v) + addsrc
    float* out2;            // optional second store of act(v) (pre-skip value, needed for ReLU mask)
    const float* mask_y;    // optional: v *= act'(mask_y) (same geometry as out)
    int mask_mode;
    int accumulate;         // out += v instead of out = v
    int tiles_y, tiles_x, co_blocks;
    int th;                 // output tile rows: 8, or 16 for the stride-1 3x3 / 1x1 layers
    int tw;                 // output tile columns: 16, or 32 for the wide v2 kernel (big 64-channel stride-1 3x3 layers)
    int* tile_counter;      // optional dynamic tile queue (device int, zero before the launch); null = static stride
    int out_bf16;           // bf16 inference kernel only: element type of `out` (addsrc / out2 are always bf16 there)
    int wino;               // 1: geometry and weights (U = G g G^T, 16 transform positions) of conv_wino_kernel (conv_wino.hip); 2: of conv_wino4_kernel (36 positions)
    int tconv;              // 1: all four output-parity classes of a stride-2 transposed 3 x 3 convolution in one launch (conv_tconv.hip)
    int out2_mode;          // what `out2` receives (fp32 kernels; conv_device.h ssie_epilogue_*): 0 = act(v) after the mask, before the skip add
                            // (the pre-skip copy of the forward); 1 = v BEFORE the mask (one launch writes a gradient and its masked copy:
                            // out = mask * v, out2 = v); 2 = the mask applied to the ACCUMULATED total (out = old + v unmasked, out2 = mask * out)
    int out2_cstride;       // bf16 inference kernel only: elements per pixel of `out2` when it differs from out_cstride (0 = the same) -
                            // the fp32 R|I output (B + 1 padded to 4) and its bf16 twin (padded to 8) at band counts like 64 or 256
};

// Epilogue shape of a launch (template parameter EPI of the persistent fprop kernels): 1 = plain forward layer (bias + ReLU / nothing),
// 2 = plain data gradient (optional ReLU mask, optional accumulate), 0 = anything else.  The persistent kernels are measurably faster the
// less epilogue code they carry (round 4: the general + edge-tile epilogue made conv_wino_kernel 75 KB of code, the plain forms 28 KB,
// -5 % time), so the two common shapes get their own instantiations.
static inline int ssie_epi_shape(const ConvParams& p)
{
    const bool plain = !p.out2 && !p.addsrc;
    if (plain && !p.mask_y && !p.accumulate && p.act != ACT_SIGMOID) return 1;
    if (plain && !p.bias && p.act == ACT_NONE && p.mask_mode != MASK_SIGMOID) return 2;
    return 0;
}

struct WgradParams {
    SrcDesc src;            // the layer input (single source per launch)
    int N, Hv, Wv;
    int ci0_total;          // first weight input-channel this source maps to (concat offset)
    int Cin;                // channels of this source
    const float* g;         // output gradient (pre-activation), NHWC
    int g_cstride, g_coff, Cout;
    int Ho, Wo, si;
    int ntaps, min_dy, min_dx, hp_h, hp_w;
    int8_t tap_dy[SSIE_MAX_TAPS];
    int8_t tap_dx[SSIE_MAX_TAPS];
    float* slabs;           // [slice][tap][ci_pad][co_pad]
    float* bias_slabs;      // optional [slice][co_pad]: fused bias gradient (column sums of g)
    int ci_pad, co_pad;
    int nslices, tiles_total, tiles_y, tiles_x, th;
    int ci_blocks, co_blocks, tap_groups;
    int wsplit;             // waves sharing one (ci,co) tile pair = partial slabs written per workgroup (4 / tile pairs)
    int rows2;              // 9 x 9 only: tap_groups counts PAIRS of kernel rows, one wave pair per row (conv_wgrad_kernel SW = 4)
    int wino;               // 1: Winograd F(3x3,2x2) kernel (conv_wgrad_wino.hip); its slabs hold the nine taps like the direct kernel's
};

struct PackDesc {
    const float* w; float* dst;
    int K, N, Npad, T, nchunks;
    int s_k, s_n, s_t;
    int8_t tapsel[SSIE_MAX_TAPS];
    // sub-block packs (several weight tensors into ONE packed operand, e.g. q | k | v as a 64 -> 192 layer): this descriptor fills output
    // columns [n_off, n_off + N) of rows Npad wide (ncnt = N: iterate over its own columns only; 0 = over Npad) and input channels
    // [k_off, k_off + K) (k_off a multiple of 16).  copy = 1: dst[n_off + i] = w[i], i < N (a bias vector into a contiguous one)
    int n_off, ncnt, k_off, copy;
    int bf16;               // 1: pack for the bf16 kernel - dst[chunk32][t][slot 0..3][n][8 bf16], k = chunk*32 + 8*slot + s
    int wino;               // 1: Winograd F(2x2,3x3) weights - dst[chunk][xi 0..15][q][n][4] = (G g G^T)[xi], tapsel[r*3+s] = source tap of g[r][s]
                            // 2: Winograd F(4x4,3x3) weights - dst[step of 8][n / 32][xi 0..35][(n % 32) / 16][pair g][n % 16][2] (conv_wino4.hip)
};

// host launchers (conv_kernels.hip); return 0 on success
int ssie_launch_fprop(const ConvParams& p, hipStream_t st);
bool ssie_fprop_v2_ok(const ConvParams& p);            // conv_fprop_v2.hip
int ssie_launch_fprop_v2(const ConvParams& p, hipStream_t st);
int ssie_launch_tconv(const ConvParams& p, hipStream_t st);        // conv_tconv.hip; p over ssie_taps_transposed_all
int ssie_launch_fprop_wino(const ConvParams& p, hipStream_t st);   // conv_wino.hip; p from ssie_conv_to_wino
int ssie_launch_fprop_wino4(const ConvParams& p, hipStream_t st);  // conv_wino4.hip; p from ssie_conv_to_wino (kind 2)
int ssie_launch_fprop_bf16(const ConvParams& p, hipStream_t st);   // conv_fprop_bf16.hip; p from ssie_make_conv_bf16
extern int ssie_fprop_min_tiles16;                     // launches with fewer tiles than this use the 8 x 16 register-staged kernel (layer_ops.hip)
extern int ssie_fprop_use_v2;                          // tuning / A-B switch (1 = use the 512-thread DMA kernel when eligible)
int ssie_launch_wgrad(const WgradParams& p, hipStream_t st);
int ssie_launch_wgrad_reduce(const float* slabs, int nslices, int ntaps, int ci_pad, int co_pad, int Cin, int Cout,
                             float* dst, long s_co, long s_ci, long s_t, const float* bias_slabs, float* db,
                             int accumulate, hipStream_t st, int accumulate_bias = -1, int co_group = 0, long w_extra = 0, long b_extra = 0);
// one layer's slab reduction as data: the batched launch (conv_kernels.hip) runs up to SSIE_REDUCE_BATCH of them, the table passed BY VALUE
// in the kernel-argument segment (scalar loads, no device-side table to keep in step with the op list)
struct ReduceDesc {
    const float* slabs; float* dst; const float* bias_slabs; float* db;
    long s_co, s_ci, s_t, w_extra, b_extra;
    int nslices, ntaps, ci_pad, co_pad, Cin, Cout, accumulate, accumulate_bias, co_group, wide;
};
#define SSIE_REDUCE_BATCH 32
struct ReduceBatch { int n; int begin[SSIE_REDUCE_BATCH + 1]; ReduceDesc d[SSIE_REDUCE_BATCH]; };     // begin[j] = first workgroup of layer j
static_assert(sizeof(ReduceBatch) <= 4096, "the kernel-argument segment holds 4 KB");
ReduceDesc ssie_make_reduce(const float* slabs, int nslices, int ntaps, int ci_pad, int co_pad, int Cin, int Cout, float* dst, long s_co, long s_ci,
                            long s_t, const float* bias_slabs, float* db, int accumulate, int accumulate_bias = -1, int co_group = 0,
                            long w_extra = 0, long b_extra = 0);
int ssie_launch_wgrad_reduce_batched(const ReduceDesc* d, int n, hipStream_t st);      // n <= SSIE_REDUCE_BATCH, disjoint destinations
int ssie_launch_wgrad_wino(const WgradParams& p, hipStream_t st);                      // conv_wgrad_wino.hip
int ssie_launch_colsum(const float* g, long npix, int cstride, int coff, int C, float* partial, int nblk,
                       float* dst, int accumulate, hipStream_t st);
int ssie_launch_pack(const PackDesc& d, hipStream_t st);
int ssie_launch_pack_batched(const PackDesc* descs_dev, int ndesc, hipStream_t st);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: `seen` is a bit mask of the devices a kernel has been
// enabled on (one `static unsigned` per launcher instantiation)
static inline void ssie_allow_full_lds(const void* fn, unsigned& seen)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 31) { hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); return; }
    if (seen & (1u << dev)) return;
    hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    seen |= 1u << dev;
}

static inline int ssie_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int ssie_round_up(int a, int b) { return ssie_ceil_div(a, b) * b; }
