// Host-side geometry builders shared by the granular C-ABI operators and the plan executor.
#pragma once
#include "ssie_common.h"

struct TapList { int n; int8_t dy[SSIE_MAX_TAPS], dx[SSIE_MAX_TAPS], sel[SSIE_MAX_TAPS]; };

struct Epilogue {
    const float* bias; int act; const float* addsrc; float* out2;
    const float* mask_y; int mask_mode; int accumulate;
    int out2_mode;          // see ConvParams.out2_mode
};

TapList ssie_taps_conv(int k);
TapList ssie_taps_dgrad_s1(int k);
TapList ssie_taps_transposed(int k, int pad, int py, int px);
TapList ssie_taps_transposed_all(void);          // the nine taps of the 3 x 3 stride-2 case in output-parity class order (1 + 2 + 2 + 4)
bool ssie_tconv_eligible(const SrcDesc& in, int N, int Hin, int Win, int Nc);
void ssie_conv_to_tconv(ConvParams& p);
bool ssie_fits_i32(long n, long h, long w, long cstride);   // n*h*w*cstride <= 2^31 - 1: the kernels' 32-bit element offsets cannot wrap
SrcDesc ssie_make_src(const float* ptr, int C, int cstride, int coff, int Hs, int Ws, int Hv, int Wv);
size_t ssie_packed_floats(int K, int N, int T);
PackDesc ssie_make_pack(const float* w, float* dst, int K, int N, const TapList& t, int s_k, int s_n, int s_t);
size_t ssie_wino_packed_floats(int K, int N);
int ssie_wino_eligible(const ConvParams& p, const TapList& t);     // 0 = no, 1 = F(2x2,3x3) (conv_wino.hip), 2 = F(4x4,3x3) (conv_wino4.hip)
PackDesc ssie_make_pack_wino(const float* w, float* dst, int K, int N, const TapList& t, int s_k, int s_n, int s_t, int kind);
void ssie_conv_to_wino(ConvParams& p, const float* u, int kind);
PackDesc ssie_make_pack_bf16(const float* w, float* dst, int K, int N, const TapList& t, int s_k, int s_n, int s_t);
int ssie_make_conv(ConvParams& p, const SrcDesc* srcs, int nsrc, int N, int Hv, int Wv, const TapList& t, int si,
                   int Ho, int Wo, const float* wpacked, int Cout,
                   float* out, int Hout, int Wout, int out_cstride, int out_coff, int so, int py, int px,
                   const Epilogue& e);
int ssie_make_conv_bf16(ConvParams& p, const SrcDesc* srcs, int nsrc, int N, int Hv, int Wv, const TapList& t, int si,
                        int Ho, int Wo, const float* wpacked, int Cout,
                        float* out, int out_bf16, int Hout, int Wout, int out_cstride, int out_coff, int so, int py, int px,
                        const Epilogue& e);
int ssie_make_wgrad(WgradParams& p, const SrcDesc& src, int N, int Hv, int Wv, int ci0_weight,
                    const float* g, int g_cstride, int g_coff, int Cout, int Ho, int Wo, int si,
                    const TapList& t, float* slabs, int target_wgs);
size_t ssie_wgrad_slab_floats(const WgradParams& p);
int ssie_run_wgrad(const SrcDesc& x, int x_creal, int N, int Hv, int Wv, const float* g, int g_cstride, int g_coff, int gC,
                   int Ho, int Wo, int si, const TapList& t, float* dw, long s_co, long s_ci, long s_t, float* db,
                   int accumulate, float* slabs, size_t slab_cap_floats, hipStream_t st);
