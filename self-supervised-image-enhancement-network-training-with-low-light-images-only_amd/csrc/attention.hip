// Spatial self-attention core of the illumination U-Net bottleneck (4 heads x 16 dims, softmax over
// H/8*W/8 tokens, no LayerNorm).  Replaces TransformerBlock.forward's attention lines
// (/root/reference/model.py:107-114) and their autograd backward; the q/k/v and feed-forward Linear
// layers (:104-106, :115-117) run on the MFMA 1x1-conv path.
//
// Tokens are the NHWC pixels of the bottleneck feature map, so the (N*T, 192) qkv buffer is read
// directly; one thread owns one query row (forward, dQ) or one key row (dK, dV) and streams the other
// side through LDS in tiles of 256 rows with an online softmax, so any token count works (256 tokens
// for a 128x128 patch, 16 384 for a 1024x1024 image) without materialising the T x T logits.
#include "attention.h"

#define AT_D 16
#define AT_TILE 256

__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, int qs, float* __restrict__ o, int os,
                                                       float* __restrict__ lse, int T, float scale)
{
    __shared__ float Ks[AT_TILE][AT_D];
    __shared__ float Vs[AT_TILE][AT_D];
    const int tid = threadIdx.x;
    const int head = blockIdx.y, n = blockIdx.z;
    const int qi = blockIdx.x * 256 + tid;
    const float* base = qkv + (size_t)n * T * qs;
    float q[AT_D], acc[AT_D];
#pragma unroll
    for (int d = 0; d < AT_D; ++d) { q[d] = qi < T ? base[(size_t)qi * qs + head * AT_D + d] * scale : 0.f; acc[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < T; k0 += AT_TILE) {
        __syncthreads();
        for (int id = tid; id < AT_TILE * 4; id += 256) {
            const int r = id >> 2, c4 = (id & 3) * 4;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (k0 + r < T) {
                kv = *(const f32x4*)(base + (size_t)(k0 + r) * qs + 64 + head * AT_D + c4);
                vv = *(const f32x4*)(base + (size_t)(k0 + r) * qs + 128 + head * AT_D + c4);
            }
            *(f32x4*)&Ks[r][c4] = kv; *(f32x4*)&Vs[r][c4] = vv;
        }
        __syncthreads();
        const int kn = min(AT_TILE, T - k0);
        for (int j = 0; j < kn; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < AT_D; ++d) s += q[d] * Ks[j][d];
            if (s > m) {
                const float f = expf(m - s);
                l *= f;
#pragma unroll
                for (int d = 0; d < AT_D; ++d) acc[d] *= f;
                m = s;
            }
            const float pj = expf(s - m);
            l += pj;
#pragma unroll
            for (int d = 0; d < AT_D; ++d) acc[d] += pj * Vs[j][d];
        }
    }
    if (qi < T) {
        const float inv = 1.f / l;
#pragma unroll
        for (int d = 0; d < AT_D; ++d) o[((size_t)n * T + qi) * os + head * AT_D + d] = acc[d] * inv;
        lse[((size_t)n * 4 + head) * T + qi] = m + logf(l);
    }
}

// dQ: one thread per query row.  p_ij = exp(s_ij - lse_i), dS = p (dP - delta), dQ_i = scale * sum_j dS_ij K_j.
// delta_i = sum_j p_ij dP_ij / sum_j p_ij is accumulated from the SAME p and dP that form dS (first sweep), so
// sum_j dS_ij cancels to rounding like torch's softmax backward does; the cheaper dO.O form leaves an error
// proportional to mean(K) that swamps the (often tiny) query/key gradients of this block.
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ qkv, int qs,
                                                          const float* __restrict__ go, int os, const float* __restrict__ lse,
                                                          float* __restrict__ gqkv, float* __restrict__ delta, int T, float scale)
{
    __shared__ float Ks[AT_TILE][AT_D];
    __shared__ float Vs[AT_TILE][AT_D];
    const int tid = threadIdx.x, head = blockIdx.y, n = blockIdx.z;
    const int qi = blockIdx.x * 256 + tid;
    const float* base = qkv + (size_t)n * T * qs;
    float q[AT_D], dO[AT_D], dq[AT_D];
    float ls = 0.f;
#pragma unroll
    for (int d = 0; d < AT_D; ++d) {
        q[d] = qi < T ? base[(size_t)qi * qs + head * AT_D + d] * scale : 0.f;
        dO[d] = qi < T ? go[((size_t)n * T + qi) * os + head * AT_D + d] : 0.f;
        dq[d] = 0.f;
    }
    if (qi < T) ls = lse[((size_t)n * 4 + head) * T + qi];
    float sp = 0.f, spd = 0.f, dl = 0.f;
    for (int sweep = 0; sweep < 2; ++sweep) {
        for (int k0 = 0; k0 < T; k0 += AT_TILE) {
            __syncthreads();
            for (int id = tid; id < AT_TILE * 4; id += 256) {
                const int r = id >> 2, c4 = (id & 3) * 4;
                f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
                if (k0 + r < T) {
                    kv = *(const f32x4*)(base + (size_t)(k0 + r) * qs + 64 + head * AT_D + c4);
                    vv = *(const f32x4*)(base + (size_t)(k0 + r) * qs + 128 + head * AT_D + c4);
                }
                *(f32x4*)&Ks[r][c4] = kv; *(f32x4*)&Vs[r][c4] = vv;
            }
            __syncthreads();
            const int kn = min(AT_TILE, T - k0);
            for (int j = 0; j < kn; ++j) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < AT_D; ++d) { s += q[d] * Ks[j][d]; dp += dO[d] * Vs[j][d]; }
                const float pj = expf(s - ls);
                if (sweep == 0) { sp += pj; spd += pj * dp; }
                else {
                    const float ds = pj * (dp - dl);
#pragma unroll
                    for (int d = 0; d < AT_D; ++d) dq[d] += ds * Ks[j][d];
                }
            }
        }
        if (sweep == 0) {
            dl = spd / sp;
            if (qi < T) delta[((size_t)n * 4 + head) * T + qi] = dl;
        }
    }
    if (qi < T)
#pragma unroll
        for (int d = 0; d < AT_D; ++d) gqkv[((size_t)n * T + qi) * qs + head * AT_D + d] = dq[d] * scale;
}

// dK, dV: one thread per key row, queries streamed through LDS
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, int qs, const float* __restrict__ go, int os,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ gqkv, int T, float scale)
{
    __shared__ float Qs[AT_TILE][AT_D];
    __shared__ float Gs[AT_TILE][AT_D];
    __shared__ float Ls[AT_TILE];
    __shared__ float Ds[AT_TILE];
    const int tid = threadIdx.x, head = blockIdx.y, n = blockIdx.z;
    const int kj = blockIdx.x * 256 + tid;
    const float* base = qkv + (size_t)n * T * qs;
    float k[AT_D], v[AT_D], dk[AT_D], dv[AT_D];
#pragma unroll
    for (int d = 0; d < AT_D; ++d) {
        k[d] = kj < T ? base[(size_t)kj * qs + 64 + head * AT_D + d] : 0.f;
        v[d] = kj < T ? base[(size_t)kj * qs + 128 + head * AT_D + d] : 0.f;
        dk[d] = 0.f; dv[d] = 0.f;
    }
    for (int q0 = 0; q0 < T; q0 += AT_TILE) {
        __syncthreads();
        for (int id = tid; id < AT_TILE * 4; id += 256) {
            const int r = id >> 2, c4 = (id & 3) * 4;
            f32x4 qv = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
            if (q0 + r < T) {
                qv = *(const f32x4*)(base + (size_t)(q0 + r) * qs + head * AT_D + c4);
                gv = *(const f32x4*)(go + ((size_t)n * T + q0 + r) * os + head * AT_D + c4);
            }
            *(f32x4*)&Qs[r][c4] = qv; *(f32x4*)&Gs[r][c4] = gv;
        }
        if (q0 + tid < T) { Ls[tid] = lse[((size_t)n * 4 + head) * T + q0 + tid]; Ds[tid] = delta[((size_t)n * 4 + head) * T + q0 + tid]; }
        __syncthreads();
        const int qn = min(AT_TILE, T - q0);
        for (int i = 0; i < qn; ++i) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < AT_D; ++d) { s += Qs[i][d] * k[d]; dp += Gs[i][d] * v[d]; }
            const float pij = expf(s * scale - Ls[i]);
            const float ds = pij * (dp - Ds[i]);
#pragma unroll
            for (int d = 0; d < AT_D; ++d) { dv[d] += pij * Gs[i][d]; dk[d] += ds * Qs[i][d]; }
        }
    }
    if (kj < T)
#pragma unroll
        for (int d = 0; d < AT_D; ++d) {
            gqkv[((size_t)n * T + kj) * qs + 64 + head * AT_D + d] = dk[d] * scale;
            gqkv[((size_t)n * T + kj) * qs + 128 + head * AT_D + d] = dv[d];
        }
}

int ssie_launch_attn_fwd(const float* qkv, int qs, float* o, int os, float* lse, int N, int T, hipStream_t st)
{
    dim3 grid((T + 255) / 256, 4, N);
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, st, qkv, qs, o, os, lse, T, 0.25f);
    return hipGetLastError() == hipSuccess ? 0 : 61;
}

int ssie_launch_attn_bwd(const float* qkv, int qs, const float* o, const float* go, int os, const float* lse,
                         float* delta, float* gqkv, int N, int T, hipStream_t st)
{
    dim3 grid((T + 255) / 256, 4, N);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, st, qkv, qs, go, os, lse, gqkv, delta, T, 0.25f);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), 0, st, qkv, qs, go, os, lse, delta, gqkv, T, 0.25f);
    return hipGetLastError() == hipSuccess ? 0 : 62;
}

// ---- granular C-ABI (parity tests) ----
extern "C" int ssie_attention_fwd(const float* qkv, float* out, float* lse, int N, int T, void* stream)
{
    if (!qkv || !out || !lse || N < 1 || T < 1) return 1;
    return ssie_launch_attn_fwd(qkv, 192, out, 64, lse, N, T, (hipStream_t)stream) ? 4 : 0;
}

extern "C" int ssie_attention_bwd(const float* qkv, const float* out, const float* gout, const float* lse,
                                  float* delta_ws, float* gqkv, int N, int T, void* stream)
{
    if (!qkv || !out || !gout || !lse || !delta_ws || !gqkv || N < 1 || T < 1) return 1;
    return ssie_launch_attn_bwd(qkv, 192, out, gout, 64, lse, delta_ws, gqkv, N, T, (hipStream_t)stream) ? 4 : 0;
}
