// Host launchers of the frequency-domain 9 x 9 convolution (spectral_conv.hip).
#pragma once
#include "ssie_common.h"

#define SSIE_SPEC_T 32                 // tile edge (FFT length)
#define SSIE_SPEC_V 24                 // valid outputs per tile edge (T - 9 + 1)
#define SSIE_SPEC_KX 17                // half spectrum: kx = 0 .. 16
#define SSIE_SPEC_NF (32 * 17)         // frequency bins per tile

// tiles per image (and the tile grid)
int ssie_spec_tiles(int H, int W, int* tiles_y, int* tiles_x);
// in (N,H,W,cs) -> out[m0 + tile][f][Cp] (complex, tile-major), Mtot tiles in the tensor; halo = 1: windows start 4 pixels before the tile
// (forward / data gradient), 0: the 24 x 24 tile itself, zero-padded (weight gradient)
int ssie_launch_spec_fft(const float* in, int cs, int Cp, int N, int H, int W, int halo, float2* out, int m0, int Mtot, hipStream_t st);
// Yf[m0 + tile][f][Np] -> out (N,H,W,cs): valid 24 x 24 block of every tile, channels < Cout (+ bias | += )
int ssie_launch_spec_ifft(const float2* Yf, int m0, int Mtot, int Np, int N, int H, int W, float* out, int cs, int Cout, const float* bias,
                          int accumulate, hipStream_t st);
// w (Cout,Cin,9,9) -> Bf[f][Kp][Np] (forward) and, when Bd != null, Bd[f][Np][Kp] (data gradient)
int ssie_launch_spec_weights(const float* w, int Cout, int Cin, int Kp, int Np, float2* Bf, float2* Bd, hipStream_t st);
// C[mc0 + m][f][n] = sum_k A[ma0 + m][f][k] B[f][k][n], m < M (A / C tile-major; Ma / Mc = their tile counts)
int ssie_launch_spec_gemm(const float2* A, int Ma, int ma0, const float2* B, float2* C, int Mc, int mc0, int M, int Kp, int Np, hipStream_t st);
// dw (Cout = 64, Cin, 9, 9) += correlation of the gradient tiles with the input windows; dWs = (nslices * NF + 9 * 17) * Kp * 64 complex scratch
int ssie_launch_spec_wgrad(const float2* Xf, const float2* Gf, float2* dWs, int M, int Kp, int nslices, int Cout, int Cin, float* dw, hipStream_t st);
// db[co] (+)= sum over all pixels of the gradient whose zero-padded tile spectra are Gn[M tiles][f][64] (the DC bins); partial: (M + 63) / 64 * 64 floats
int ssie_launch_spec_bias(const float2* Gn, int M, float* partial, float* db, int accumulate, hipStream_t st);
