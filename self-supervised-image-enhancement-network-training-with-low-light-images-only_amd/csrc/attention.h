#pragma once
#include "ssie_common.h"
// qkv: (N*T, qs) rows = tokens, q at [0,64), k at [64,128), v at [128,192); o/go: (N*T, os) 64 channels;
// lse/delta: (N, 4, T).  scale = 1/sqrt(16) (model.py:110-111).
int ssie_launch_attn_fwd(const float* qkv, int qs, float* o, int os, float* lse, int N, int T, hipStream_t st);
int ssie_launch_attn_bwd(const float* qkv, int qs, const float* o, const float* go, int os, const float* lse,
                         float* delta, float* gqkv, int N, int T, hipStream_t st);
// bf16-MFMA forward of the mixed-precision enhance-only path: fp32 qkv in, bf16 output (N*T, os), no lse
// scratch (may be null): room for the pre-converted bf16 keys / values, ssie_attn_bf16_scratch_bytes(N, T)
size_t ssie_attn_bf16_scratch_bytes(int N, int T);
int ssie_launch_attn_fwd_bf16(const float* qkv, int qs, void* o, int os, int N, int T, hipStream_t st, void* scratch = nullptr, size_t scratch_bytes = 0);
