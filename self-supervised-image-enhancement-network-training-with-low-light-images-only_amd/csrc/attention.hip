// Spatial self-attention core of the illumination U-Net bottleneck (4 heads x 16 dims, softmax over
// H/8*W/8 tokens, no LayerNorm).  Replaces TransformerBlock.forward's attention lines
// (/root/reference/model.py:107-114) and their autograd backward; the q/k/v and feed-forward Linear
// layers (:104-106, :115-117) run on the MFMA 1x1-conv path.
//
// Tokens are the NHWC pixels of the bottleneck feature map, so the (N*T, 192) qkv buffer is read
// directly; the forward is an fp32-MFMA flash attention (below); in the backward eight threads share one query row (dQ) or
// one key row (dK, dV) and stream the other side through LDS in tiles of 256 rows, so any token count works (256 tokens
// for a 128x128 patch, 16 384 for a 1024x1024 image) without materialising the T x T logits.
#include "attention.h"

#define AT_D 16
#define AT_TILE 256

// Forward: fp32-MFMA flash attention.  One wave owns 32 queries of one head; per block of 32 keys it computes
//   S^T = K Q^T      (32 keys x 32 queries, 8 x v_mfma_f32_32x32x2_f32 over the 16 head dims), then
//   O^T += V^T P^T   (16 dims (padded to 32 rows) x 32 queries, 16 MFMAs over the 32 keys)
// Working on the TRANSPOSED logits puts the query on the lane axis of the accumulator layout, so (a) the online-softmax
// state (running max / sum / rescale) is one scalar per lane, and (b) the probabilities P^T already sit in registers in
// exactly the B-operand layout of the second product: accumulator register s of lane (q, h) is key row
// (s&3) + 8*(s>>2) + 4h, which we simply DEFINE as the h-th key of contraction step s (the order of a sum is free), and
// V^T is fetched from LDS to match.  No shuffle, no LDS round trip for P.  The two lane halves of a query share the
// running max (one cross-half exchange per block) and keep separate partial sums, joined once at the end.
#define AK_W 64          // keys per wave per iteration (two 32-key blocks)
#define AK_ST 256        // keys staged in LDS per iteration: the four waves of a workgroup take 64 each
#define AK_LD 20         // padded LDS row (floats): 16-byte aligned rows, conflict-free 128-bit reads

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// A workgroup = 32 queries of one head; its FOUR waves split the keys (wave w takes keys 64w .. 64w+63 of every staged
// 256-key slab) and merge their (max, sum, O) states through LDS at the end.  With one wave per 32 queries a 16 384-token
// image gave only two waves per SIMD - too few to overlap one wave's softmax (VALU, cross-half exchange) with another's
// MFMAs; the key split quadruples the waves in flight.  Logits are kept in base-2 units (log2(e) folded into the query
// scale), so the softmax exponentials are bare v_exp_f32.
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, int qs, float* __restrict__ o, int os,
                                                       float* __restrict__ lse, int T, float scale)
{
    __shared__ __attribute__((aligned(16))) float Ks[AK_ST][AK_LD];
    __shared__ __attribute__((aligned(16))) float Vs[AK_ST][AK_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, li = lane & 31;
    const int head = blockIdx.y, n = blockIdx.z;
    const int qi = blockIdx.x * 32 + li;
    const float* base = qkv + (size_t)n * T * qs;
    const float LOG2E = 1.4426950408889634f;

    // B operand of S^T: lane (q, h) supplies Q[q][8h + s] * scale * log2(e) at contraction step s
    float qb[8];
    {
        f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;
        if (qi < T) {
            q0 = *(const f32x4*)(base + (size_t)qi * qs + head * AT_D + 8 * h);
            q1 = *(const f32x4*)(base + (size_t)qi * qs + head * AT_D + 8 * h + 4);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) { qb[s] = q0[s] * (scale * LOG2E); qb[4 + s] = q1[s] * (scale * LOG2E); }
    }
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    float m = -INFINITY, lsum = 0.f;       // m in base-2 units

    for (int k0 = 0; k0 < T; k0 += AK_ST) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < AK_ST * 4 / 256; ++it) {
            const int id = it * 256 + tid;
            const int r = id >> 2, c4 = (id & 3) * 4;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
            if (k0 + r < T) {
                kv = *(const f32x4*)(base + (size_t)(k0 + r) * qs + 64 + head * AT_D + c4);
                vv = *(const f32x4*)(base + (size_t)(k0 + r) * qs + 128 + head * AT_D + c4);
            }
            *(f32x4*)&Ks[r][c4] = kv; *(f32x4*)&Vs[r][c4] = vv;
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < AK_W / 32; ++sub) {
            const int lr = wave * AK_W + sub * 32;          // first LDS row of this wave's block
            const int kb = k0 + lr;
            if (kb >= T) break;
            // S^T = K Q^T: A operand lane (key, h) supplies K[key][8h + s]
            const f32x4 ka0 = *(const f32x4*)&Ks[lr + li][8 * h], ka1 = *(const f32x4*)&Ks[lr + li][8 * h + 4];
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) st = MFMA32(ka0[s], qb[s], st);
#pragma unroll
            for (int s = 0; s < 4; ++s) st = MFMA32(ka1[s], qb[4 + s], st);
            if (kb + 32 > T) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb + (r & 3) + 8 * (r >> 2) + 4 * h >= T) st[r] = -INFINITY;
            }
            float mb = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mb = fmaxf(mb, st[r]);
            mb = fmaxf(mb, __shfl_xor(mb, 32));
            const float m_new = fmaxf(m, mb);                 // finite: key kb < T is never masked
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = __builtin_amdgcn_exp2f(st[r] - m_new); ps += st[r]; }
            lsum = lsum * alpha + ps;
#pragma unroll
            for (int r = 0; r < 8; ++r) oacc[r] *= alpha;     // rows d < 16 of O^T live in registers 0..7
            m = m_new;
            // O^T += V^T P^T: step s contracts key rows (s&3) + 8*(s>>2) + 4h
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float a = Vs[lr + (s & 3) + 8 * (s >> 2) + 4 * h][li & 15];
                oacc = MFMA32(a, st[s], oacc);
            }
        }
    }
    // merge the four waves' partial states (a wave that saw no key has m = -inf, l = 0, O = 0)
    lsum += __shfl_xor(lsum, 32);
    __syncthreads();
    float* mg = &Ks[0][0];                   // [3 waves][10][64 lanes] floats = 7.7 KB, inside the staging area
    if (wave > 0) {
        float* d = mg + (size_t)(wave - 1) * 10 * 64 + lane;
        d[0] = m; d[64] = lsum;
#pragma unroll
        for (int r = 0; r < 8; ++r) d[(2 + r) * 64] = oacc[r];
    }
    __syncthreads();
    if (wave == 0) {
        float mw[3], lw[3];
        float mt = m;
#pragma unroll
        for (int w = 0; w < 3; ++w) { mw[w] = mg[(size_t)w * 10 * 64 + lane]; lw[w] = mg[(size_t)w * 10 * 64 + 64 + lane]; mt = fmaxf(mt, mw[w]); }
        const float f0 = __builtin_amdgcn_exp2f(m - mt);
        float ltot = lsum * f0;
#pragma unroll
        for (int r = 0; r < 8; ++r) oacc[r] *= f0;
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const float f = __builtin_amdgcn_exp2f(mw[w] - mt);
            ltot += lw[w] * f;
#pragma unroll
            for (int r = 0; r < 8; ++r) oacc[r] += mg[(size_t)w * 10 * 64 + (2 + r) * 64 + lane] * f;
        }
        if (qi < T) {
            const float inv = 1.f / ltot;
            float* op = o + ((size_t)n * T + qi) * os + head * AT_D + 4 * h;
            const f32x4 v0 = {oacc[0] * inv, oacc[1] * inv, oacc[2] * inv, oacc[3] * inv};
            const f32x4 v1 = {oacc[4] * inv, oacc[5] * inv, oacc[6] * inv, oacc[7] * inv};
            *(f32x4*)op = v0; *(f32x4*)(op + 8) = v1;
            if (h == 0) lse[((size_t)n * 4 + head) * T + qi] = mt * 0.6931471805599453f + logf(ltot);
        }
    }
}

// bf16-MFMA variant for the mixed-precision enhance-only path (no lse, bf16 output): same transposed-logits scheme with
// v_mfma_f32_32x32x16_bf16 - S^T is ONE MFMA per 32 x 32 block (K = the 16 head dims) and O^T += V^T P^T two (K = 32 keys).
// The contraction order over keys is again free: MFMA j, lane half h contracts the keys held by accumulator registers
// 8j .. 8j+7 of that half, and V^T is staged in LDS already permuted to that order (Vt[block][d][16j + 8h + (r&7)]), so both
// operands of the second product are single 128-bit reads / register packs.  fp32 softmax and accumulation.
typedef __bf16 at_bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(at_bf16x8, (a)), __builtin_bit_cast(at_bf16x8, (b)), (c), 0, 0, 0)
__device__ __forceinline__ unsigned at_f2bf(float f) { return ssie_f2bf(f); }
__device__ __forceinline__ unsigned at_pack2(float a, float b) { return ssie_pack2bf(a, b); }
#define AV_LD 40         // Vt row: 32 bf16 keys padded to 40 (80 B): conflict-free 128-bit reads across the 16 dims

__global__ __launch_bounds__(256) void attn_fwd_bf16_kernel(const float* __restrict__ qkv, int qs, unsigned short* __restrict__ o, int os, int T, float scale)
{
    __shared__ __attribute__((aligned(16))) unsigned short Kh[AK_ST][16];             // [key][dim]
    __shared__ __attribute__((aligned(16))) unsigned short Vt[AK_ST / 32][16][AV_LD];  // [32-key block][dim][permuted key]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, li = lane & 31;
    const int head = blockIdx.y, n = blockIdx.z;
    // 128 queries per workgroup: every wave owns 32 queries and walks ALL keys of a staged slab.  (With the four waves sharing 32
    // queries and splitting the keys, a workgroup converted and staged the whole K / V of its head for 32 queries: 2 048
    // workgroups x 2 MB at 16 384 tokens - the staging, not the MFMAs or the softmax, was most of the kernel.)
    const int qi = blockIdx.x * 128 + wave * 32 + li;
    const float* base = qkv + (size_t)n * T * qs;
    const float LOG2E = 1.4426950408889634f;

    uint4 qb;                                    // B operand of S^T: Q[q][8h .. 8h+7] * scale * log2(e), bf16
    {
        f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;
        if (qi < T) {
            q0 = *(const f32x4*)(base + (size_t)qi * qs + head * AT_D + 8 * h);
            q1 = *(const f32x4*)(base + (size_t)qi * qs + head * AT_D + 8 * h + 4);
        }
        const float c = scale * LOG2E;
        qb = make_uint4(at_pack2(q0[0] * c, q0[1] * c), at_pack2(q0[2] * c, q0[3] * c), at_pack2(q1[0] * c, q1[1] * c), at_pack2(q1[2] * c, q1[3] * c));
    }
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    float m = -INFINITY, lsum = 0.f;

    for (int k0 = 0; k0 < T; k0 += AK_ST) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < AK_ST * 4 / 256; ++it) {
            const int id = it * 256 + tid;
            const int r = id >> 2, c4 = (id & 3) * 4;           // key row r of the slab, dims c4 .. c4+3
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
            if (k0 + r < T) {
                kv = *(const f32x4*)(base + (size_t)(k0 + r) * qs + 64 + head * AT_D + c4);
                vv = *(const f32x4*)(base + (size_t)(k0 + r) * qs + 128 + head * AT_D + c4);
            }
            *(uint2*)&Kh[r][c4] = make_uint2(at_pack2(kv[0], kv[1]), at_pack2(kv[2], kv[3]));
            const int key = r & 31, blk = r >> 5;
            const int rr = (key & 3) + 4 * (key >> 3), hh = (key >> 2) & 1;
            const int pos = (rr >> 3) * 16 + hh * 8 + (rr & 7);
#pragma unroll
            for (int e = 0; e < 4; ++e) Vt[blk][c4 + e][pos] = (unsigned short)at_f2bf(vv[e]);
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < AK_ST / 32; ++sub) {
            const int lr = sub * 32;
            const int kb = k0 + lr;
            if (kb >= T) break;
            const uint4 ka = *(const uint4*)&Kh[lr + li][8 * h];
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            st = MFMA_BF16(ka, qb, st);
            if (kb + 32 > T) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb + (r & 3) + 8 * (r >> 2) + 4 * h >= T) st[r] = -INFINITY;
            }
            float mb = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mb = fmaxf(mb, st[r]);
            mb = fmaxf(mb, __shfl_xor(mb, 32));
            const float m_new = fmaxf(m, mb);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = __builtin_amdgcn_exp2f(st[r] - m_new); ps += st[r]; }
            lsum = lsum * alpha + ps;
#pragma unroll
            for (int r = 0; r < 8; ++r) oacc[r] *= alpha;
            m = m_new;
            const int blk = lr >> 5;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint4 va = *(const uint4*)&Vt[blk][li & 15][j * 16 + h * 8];
                const uint4 pb = make_uint4(at_pack2(st[8 * j + 0], st[8 * j + 1]), at_pack2(st[8 * j + 2], st[8 * j + 3]),
                                            at_pack2(st[8 * j + 4], st[8 * j + 5]), at_pack2(st[8 * j + 6], st[8 * j + 7]));
                oacc = MFMA_BF16(va, pb, oacc);
            }
        }
    }
    lsum += __shfl_xor(lsum, 32);
    if (qi < T) {
        const float inv = 1.f / lsum;
        unsigned short* op = o + ((size_t)n * T + qi) * os + head * AT_D + 4 * h;
        *(uint2*)op = make_uint2(at_pack2(oacc[0] * inv, oacc[1] * inv), at_pack2(oacc[2] * inv, oacc[3] * inv));
        *(uint2*)(op + 8) = make_uint2(at_pack2(oacc[4] * inv, oacc[5] * inv), at_pack2(oacc[6] * inv, oacc[7] * inv));
    }
}

// ---- bf16 forward with PRE-CONVERTED keys / values (the 16 384-token enhance-only image) ----
// attn_fwd_bf16_kernel converts and permutes the K / V slab it stages for every 128 queries: at 16 384 tokens that is 512
// workgroups x 2 MB of fp32 reads, 16-byte loads at a 768-byte stride, 2-byte LDS scatter writes and no prefetch - the staging,
// not the softmax, was half the kernel.  attn_kv_bf16_kernel does the conversion ONCE into a scratch buffer in exactly the LDS
// image of a slab (Kh[key][16] and the permuted, row-padded Vt[block][dim][40]), and attn_fwd_bf16p_kernel stages slab s + 1 by
// LDS-DMA (18 pieces of 1 KB) under the MFMAs and softmax of slab s.
__global__ __launch_bounds__(256) void attn_kv_bf16_kernel(const float* __restrict__ qkv, int qs, unsigned short* __restrict__ kg,
                                                           unsigned short* __restrict__ vg, int T, int Tpad)
{
    const int key = blockIdx.x * 64 + (threadIdx.x >> 2), head = threadIdx.x & 3, n = blockIdx.y;
    if (key >= Tpad) return;
    f32x4 kv[4], vv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        kv[c] = f32x4{0.f, 0.f, 0.f, 0.f}; vv[c] = kv[c];
        if (key < T) {
            kv[c] = *(const f32x4*)(qkv + ((size_t)n * T + key) * qs + 64 + head * AT_D + 4 * c);
            vv[c] = *(const f32x4*)(qkv + ((size_t)n * T + key) * qs + 128 + head * AT_D + 4 * c);
        }
    }
    unsigned short* kd = kg + (((size_t)n * 4 + head) * Tpad + key) * 16;
    *(uint4*)kd = make_uint4(at_pack2(kv[0][0], kv[0][1]), at_pack2(kv[0][2], kv[0][3]), at_pack2(kv[1][0], kv[1][1]), at_pack2(kv[1][2], kv[1][3]));
    *(uint4*)(kd + 8) = make_uint4(at_pack2(kv[2][0], kv[2][1]), at_pack2(kv[2][2], kv[2][3]), at_pack2(kv[3][0], kv[3][1]), at_pack2(kv[3][2], kv[3][3]));
    const int k32 = key & 31, blk = key >> 5;
    const int rr = (k32 & 3) + 4 * (k32 >> 3), hh = (k32 >> 2) & 1;
    const int pos = (rr >> 3) * 16 + hh * 8 + (rr & 7);
    unsigned short* vd = vg + ((((size_t)n * 4 + head) * (Tpad / 32) + blk) * 16) * AV_LD + pos;
#pragma unroll
    for (int d = 0; d < 16; ++d) vd[d * AV_LD] = (unsigned short)at_f2bf(vv[d >> 2][d & 3]);
}

#define ATP_GLDS16(gptr, lptr)                                                                         \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
// A workgroup = 128 queries of one head x TWO key halves: wave w = (query group w & 3, key half w >> 2) takes the 32-key blocks
// 4 (w >> 2) .. + 3 of every staged 256-key slab and the two halves merge their (reference, sum, O) states through LDS at the end.
// With one wave per 32 queries the image gave two waves per SIMD: too few to cover one wave's serial MFMA -> max -> vote -> exp ->
// pack -> MFMA chain with another's; the key split makes it four.
// The row sums come out of the second product: V^T has 16 real rows of the MFMA's 32, so the lanes of rows 16 - 31 feed ONES
// (a constant LDS region addressed with the same immediates) and accumulator register 8 is sum_k P[k][query] - of the bf16-rounded
// weights the numerator uses, over both lane halves; the 16 v_add_f32 per block and the cross-half exchange at the end go away.
#define ATP_HB (AK_ST / 64)          // 32-key blocks per key half of a slab
__global__ __launch_bounds__(512) void attn_fwd_bf16p_kernel(const float* __restrict__ qkv, int qs, const unsigned short* __restrict__ kg,
                                                             const unsigned short* __restrict__ vg, unsigned short* __restrict__ o, int os,
                                                             int T, int Tpad, float scale)
{
    constexpr int KSL = AK_ST * 16, VSL = (AK_ST / 32) * 16 * AV_LD;         // ushorts per K / V slab: 8 KB and 10 KB
    __shared__ __attribute__((aligned(16))) unsigned short Kh[2][KSL];
    __shared__ __attribute__((aligned(16))) unsigned short Vt[2][VSL];
    __shared__ __attribute__((aligned(16))) unsigned short Ones[ATP_HB * 16 * AV_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3, kh = wave >> 2;
    const int h = lane >> 5, li = lane & 31;
    const int head = blockIdx.y, n = blockIdx.z;
    const int qi = blockIdx.x * 128 + wq * 32 + li;
    const float* base = qkv + (size_t)n * T * qs;
    const float LOG2E = 1.4426950408889634f;
    const unsigned short* kh_g = kg + ((size_t)n * 4 + head) * Tpad * 16;
    const unsigned short* vt_g = vg + ((size_t)n * 4 + head) * (Tpad / 32) * 16 * AV_LD;

    for (int i = tid; i < ATP_HB * 16 * AV_LD / 2; i += 512) ((unsigned*)Ones)[i] = 0x3f803f80u;      // bf16 1.0 pairs (visible after the first barrier)
    uint4 qb;
    {
        f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;
        if (qi < T) {
            q0 = *(const f32x4*)(base + (size_t)qi * qs + head * AT_D + 8 * h);
            q1 = *(const f32x4*)(base + (size_t)qi * qs + head * AT_D + 8 * h + 4);
        }
        const float c = scale * LOG2E;
        qb = make_uint4(at_pack2(q0[0] * c, q0[1] * c), at_pack2(q0[2] * c, q0[3] * c), at_pack2(q1[0] * c, q1[1] * c), at_pack2(q1[2] * c, q1[3] * c));
    }
    f32x16 oacc;                 // [0..7]: O^T rows of this lane half; [8]: the row sum (rows 16 - 31 all hold it)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    // minus the reference of the logits (see the loop); this wave's first block sets it
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;

    // one slab = 8 K pieces + 10 V pieces of 1 KB (64 lanes x 16 B): wave w takes K piece w and V pieces w (, w + 8)
#define ATP_STAGE(SLAB, BUF)                                                                                          \
    {                                                                                                                 \
        const unsigned short* ks_ = kh_g + (size_t)(SLAB) * KSL;                                                      \
        const unsigned short* vs_ = vt_g + (size_t)(SLAB) * VSL;                                                      \
        ATP_GLDS16(ks_ + wave * 512 + lane * 8, &Kh[BUF][wave * 512]);                                                \
        ATP_GLDS16(vs_ + wave * 512 + lane * 8, &Vt[BUF][wave * 512]);                                                \
        if (wave + 8 < VSL / 512) ATP_GLDS16(vs_ + (wave + 8) * 512 + lane * 8, &Vt[BUF][(wave + 8) * 512]);          \
    }
    const int nslab = Tpad / AK_ST;
    ATP_STAGE(0, 0)
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1, kbase = s * AK_ST + kh * (AK_ST / 2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s + 1 < nslab) ATP_STAGE(s + 1, buf ^ 1)
        if (kbase >= T) continue;                                   // wave-uniform: this half of the last slab is padding
        const unsigned short* Kb = Kh[buf] + kh * (AK_ST / 2) * 16;
        // rows 16 - 31 of V^T: ones; the offsets added below are the same compile-time immediates for every lane
        const unsigned short* Va = (li & 16) ? Ones + h * 8 : Vt[buf] + (kh * ATP_HB * 16 + (li & 15)) * AV_LD + h * 8;
        // the logits of block i + 1 are issued before the softmax of block i: the MFMA -> max -> exp -> pack -> MFMA chain of
        // one block is serial
        f32x16 stn = MFMA_BF16(*(const uint4*)(Kb + li * 16 + 8 * h), qb, negm);
#pragma unroll
        for (int i = 0; i < ATP_HB; ++i) {
            const int lr = i * 32;
            const int kb = kbase + lr;
            if (kb >= T) break;
            // LAZY running maximum: the accumulator starts at -m (negm: 16 registers that only change on the slow path), so the
            // MFMA returns s - m directly, and the state is re-based only when a logit of this block exceeds the reference by
            // more than 2^8 (softmax is invariant to the reference; 2^8 is far inside fp32 / bf16 range).  After the first
            // blocks that is rare, and the common path per block is: one MFMA, a max tree + one wave vote, 16 v_exp_f32, 8 packed
            // converts, two MFMAs - the vector instructions are what bounds this kernel (head dimension 16: 3 MFMAs per 1 024 logits).
            f32x16 st = stn;
            if (i + 1 < ATP_HB) stn = MFMA_BF16(*(const uint4*)(Kb + (lr + 32 + li) * 16 + 8 * h), qb, negm);
            if (kb + 32 > T) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb + (r & 3) + 8 * (r >> 2) + 4 * h >= T) st[r] = -INFINITY;
            }
            float mb = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mb = fmaxf(fmaxf(mb, st[r]), st[r + 1]);
            mb = fmaxf(mb, st[15]);
            const bool first = s == 0 && i == 0;                    // this wave's very first block sets the reference to its own maximum
            if (first || __builtin_amdgcn_ballot_w64(mb > 8.f)) {
                // slow path (whole wave): both lane halves of a query share the reference
                mb = fmaxf(mb, __shfl_xor(mb, 32));
                const float d = first ? mb : fmaxf(mb, 0.f);        // afterwards the reference only ever rises
                // the first block only moves the reference (the sums are still 0): no rescale - exp2(-d) is +inf when every logit
                // of the first 32 keys sits below -128, and 0 * inf would leave that query NaN for good
                const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-d);
#pragma unroll
                for (int r = 0; r < 9; ++r) oacc[r] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { st[r] -= d; stn[r] -= d; negm[r] -= d; }          // (the prefetched block was formed against the old reference)
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = __builtin_amdgcn_exp2f(st[r]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint4 va = *(const uint4*)(Va + (i * 16) * AV_LD + j * 16);
                const uint4 pb = make_uint4(at_pack2(st[8 * j + 0], st[8 * j + 1]), at_pack2(st[8 * j + 2], st[8 * j + 3]),
                                            at_pack2(st[8 * j + 4], st[8 * j + 5]), at_pack2(st[8 * j + 6], st[8 * j + 7]));
                oacc = MFMA_BF16(va, pb, oacc);
            }
        }
    }
#undef ATP_STAGE
    // merge the two key halves (fixed order: half 0 + half 1).  A half without keys (a token count inside the first half slab)
    // has sum 0 and takes the other's reference.
    __syncthreads();
    float* red = (float*)&Kh[0][0];                                  // 10 x 256 floats of the 16 KB the key slabs held
    static_assert(10 * 256 * sizeof(float) <= sizeof(unsigned short) * 2 * KSL, "merge scratch does not fit the key slabs");
    const int slot = wq * 64 + lane;
    if (kh == 1) {
        red[slot] = negm[0];
#pragma unroll
        for (int r = 0; r < 9; ++r) red[(1 + r) * 256 + slot] = oacc[r];
    }
    __syncthreads();
    if (kh == 0 && qi < T) {
        const float l2 = red[9 * 256 + slot];
        const float nm1 = negm[0], nm2 = l2 == 0.f ? nm1 : red[slot];
        const float nmn = fminf(nm1, nm2);                          // minus the larger reference
        const float a1 = __builtin_amdgcn_exp2f(nmn - nm1), a2 = __builtin_amdgcn_exp2f(nmn - nm2);
        const float inv = 1.f / (oacc[8] * a1 + l2 * a2);
        float ov[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) ov[r] = (oacc[r] * a1 + red[(1 + r) * 256 + slot] * a2) * inv;
        unsigned short* op = o + ((size_t)n * T + qi) * os + head * AT_D + 4 * h;
        *(uint2*)op = make_uint2(at_pack2(ov[0], ov[1]), at_pack2(ov[2], ov[3]));
        *(uint2*)(op + 8) = make_uint2(at_pack2(ov[4], ov[5]), at_pack2(ov[6], ov[7]));
    }
}

// Backward.  A workgroup = 32 rows x 8 PARTS: thread (part, row) owns its row's share of every staged 256-row tile of the other
// side (rows 32 part .. 32 part + 31 of the tile), and the eight partial results of a row are added through LDS in part order
// (fixed order: bit-reproducible).  With one thread per row and 256 rows per workgroup a 128 x 128 patch batch (256 tokens x 4
// heads x 32 images) was 128 workgroups of 256-iteration scalar loops: half the chip idle, 112 + 146 us; the split gives 1 024
// workgroups of 32-iteration loops.  The two lane halves of a wave hold two parts, so a tile row read is an LDS broadcast.
//
// dQ: p_ij = exp(s_ij - lse_i), dS = p (dP - delta), dQ_i = scale * sum_j dS_ij K_j.
// delta_i = sum_j p_ij dP_ij / sum_j p_ij is accumulated from the SAME p and dP that form dS (first sweep), so
// sum_j dS_ij cancels to rounding like torch's softmax backward does; the cheaper dO.O form leaves an error
// proportional to mean(K) that swamps the (often tiny) query/key gradients of this block.
#define AB_ROWS 32
#define AB_PARTS 8
#define AB_SPAN (AT_TILE / AB_PARTS)
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ qkv, int qs,
                                                          const float* __restrict__ go, int os, const float* __restrict__ lse,
                                                          float* __restrict__ gqkv, float* __restrict__ delta, int T, float scale)
{
    __shared__ __attribute__((aligned(16))) float Ks[AT_TILE][AT_D];
    __shared__ __attribute__((aligned(16))) float Vs[AT_TILE][AT_D];
    __shared__ float red[AB_PARTS][AT_D + 1][AB_ROWS];
    const int tid = threadIdx.x, head = blockIdx.y, n = blockIdx.z;
    const int rl = tid & (AB_ROWS - 1), part = tid >> 5;
    const int qi = blockIdx.x * AB_ROWS + rl;
    const float* base = qkv + (size_t)n * T * qs;
    float q[AT_D], dO[AT_D], dq[AT_D];
    float ls = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < AT_D; d4 += 4) {
        f32x4 qv = {0.f, 0.f, 0.f, 0.f}, gv = qv;
        if (qi < T) {
            qv = *(const f32x4*)(base + (size_t)qi * qs + head * AT_D + d4);
            gv = *(const f32x4*)(go + ((size_t)n * T + qi) * os + head * AT_D + d4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { q[d4 + e] = qv[e] * scale; dO[d4 + e] = gv[e]; dq[d4 + e] = 0.f; }
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
            const int j0 = part * AB_SPAN, j1 = min(j0 + AB_SPAN, T - k0);
            for (int j = j0; j < j1; ++j) {
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
            red[part][0][rl] = sp; red[part][1][rl] = spd;
            __syncthreads();
            float a = 0.f, c = 0.f;
#pragma unroll
            for (int p = 0; p < AB_PARTS; ++p) { a += red[p][0][rl]; c += red[p][1][rl]; }
            dl = c / a;
            if (part == 0 && qi < T) delta[((size_t)n * 4 + head) * T + qi] = dl;
        }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < AT_D; ++d) red[part][d][rl] = dq[d];
    __syncthreads();
    // thread (part, row) adds dims 2 part, 2 part + 1 of its row over the eight parts
    if (qi < T) {
        float o2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float a = 0.f;
#pragma unroll
            for (int p = 0; p < AB_PARTS; ++p) a += red[p][2 * part + e][rl];
            o2[e] = a * scale;
        }
        *(float2*)(gqkv + ((size_t)n * T + qi) * qs + head * AT_D + 2 * part) = make_float2(o2[0], o2[1]);
    }
}

// dK, dV: 32 key rows x 8 parts per workgroup, queries (with their dO, lse, delta) streamed through LDS
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, int qs, const float* __restrict__ go, int os,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ gqkv, int T, float scale)
{
    __shared__ __attribute__((aligned(16))) float QG[2][AT_TILE][AT_D];          // one array: the reduction below reuses both halves
    float (*Qs)[AT_D] = QG[0];
    float (*Gs)[AT_D] = QG[1];
    __shared__ float Ls[AT_TILE];
    __shared__ float Ds[AT_TILE];
    const int tid = threadIdx.x, head = blockIdx.y, n = blockIdx.z;
    const int rl = tid & (AB_ROWS - 1), part = tid >> 5;
    const int kj = blockIdx.x * AB_ROWS + rl;
    const float* base = qkv + (size_t)n * T * qs;
    float k[AT_D], v[AT_D], dk[AT_D], dv[AT_D];
#pragma unroll
    for (int d4 = 0; d4 < AT_D; d4 += 4) {
        f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
        if (kj < T) {
            kv = *(const f32x4*)(base + (size_t)kj * qs + 64 + head * AT_D + d4);
            vv = *(const f32x4*)(base + (size_t)kj * qs + 128 + head * AT_D + d4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { k[d4 + e] = kv[e]; v[d4 + e] = vv[e]; dk[d4 + e] = 0.f; dv[d4 + e] = 0.f; }
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
        const int i0 = part * AB_SPAN, i1 = min(i0 + AB_SPAN, T - q0);
        for (int i = i0; i < i1; ++i) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < AT_D; ++d) { s += Qs[i][d] * k[d]; dp += Gs[i][d] * v[d]; }
            const float pij = expf(s * scale - Ls[i]);
            const float ds = pij * (dp - Ds[i]);
#pragma unroll
            for (int d = 0; d < AT_D; ++d) { dv[d] += pij * Gs[i][d]; dk[d] += ds * Qs[i][d]; }
        }
    }
    // partial sums of the eight parts, added in part order; the tile buffers are free now: red[part][0..31][row]
    __syncthreads();
    float (*red)[2 * AT_D][AB_ROWS] = (float (*)[2 * AT_D][AB_ROWS])&QG[0][0][0];      // 8 x 32 x 32 floats = Qs + Gs
    static_assert(sizeof(float) * AB_PARTS * 2 * AT_D * AB_ROWS <= 2 * sizeof(float) * AT_TILE * AT_D, "reduction scratch does not fit the tile buffers");
#pragma unroll
    for (int d = 0; d < AT_D; ++d) { red[part][d][rl] = dk[d]; red[part][AT_D + d][rl] = dv[d]; }
    __syncthreads();
    if (kj < T) {
        // thread (part, row): values 4 part .. 4 part + 3 of the row's 32 (dK | dV)
        float o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = 0.f;
#pragma unroll
            for (int p = 0; p < AB_PARTS; ++p) a += red[p][4 * part + e][rl];
            o4[e] = part < 4 ? a * scale : a;
        }
        float* dst = gqkv + ((size_t)n * T + kj) * qs + head * AT_D + (part < 4 ? 64 + 4 * part : 128 + 4 * (part - 4));
        *(f32x4*)dst = f32x4{o4[0], o4[1], o4[2], o4[3]};
    }
}

int ssie_launch_attn_fwd(const float* qkv, int qs, float* o, int os, float* lse, int N, int T, hipStream_t st)
{
    dim3 grid((T + 31) / 32, 4, N);
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, st, qkv, qs, o, os, lse, T, 0.25f);
    return hipGetLastError() == hipSuccess ? 0 : 61;
}

// qkv fp32 (N*T, qs); o bf16 (N*T, os)
int ssie_attn_bf16_prepass = 1;       // ssie_debug_set_attn_bf16_prepass: 0 = every workgroup converts the K / V slabs it stages
extern "C" void ssie_debug_set_attn_bf16_prepass(int v) { ssie_attn_bf16_prepass = v; }
size_t ssie_attn_bf16_scratch_bytes(int N, int T)
{
    const size_t Tpad = (size_t)(T + AK_ST - 1) / AK_ST * AK_ST;
    return (size_t)N * 4 * Tpad * 16 * 2 + (size_t)N * 4 * (Tpad / 32) * 16 * AV_LD * 2;
}
// scratch (optional): ssie_attn_bf16_scratch_bytes(N, T) bytes for the pre-converted keys / values
int ssie_launch_attn_fwd_bf16(const float* qkv, int qs, void* o, int os, int N, int T, hipStream_t st, void* scratch, size_t scratch_bytes)
{
    if (ssie_attn_bf16_prepass && scratch && T >= AK_ST && scratch_bytes >= ssie_attn_bf16_scratch_bytes(N, T)) {
        const int Tpad = (T + AK_ST - 1) / AK_ST * AK_ST;
        unsigned short* kg = (unsigned short*)scratch;
        unsigned short* vg = kg + (size_t)N * 4 * Tpad * 16;
        hipLaunchKernelGGL(attn_kv_bf16_kernel, dim3(Tpad / 64, N), dim3(256), 0, st, qkv, qs, kg, vg, T, Tpad);
        hipLaunchKernelGGL(attn_fwd_bf16p_kernel, dim3((T + 127) / 128, 4, N), dim3(512), 0, st, qkv, qs, (const unsigned short*)kg,
                           (const unsigned short*)vg, (unsigned short*)o, os, T, Tpad, 0.25f);
        return hipGetLastError() == hipSuccess ? 0 : 64;
    }
    dim3 grid((T + 127) / 128, 4, N);
    hipLaunchKernelGGL(attn_fwd_bf16_kernel, grid, dim3(256), 0, st, qkv, qs, (unsigned short*)o, os, T, 0.25f);
    return hipGetLastError() == hipSuccess ? 0 : 63;
}

int ssie_launch_attn_bwd(const float* qkv, int qs, const float* o, const float* go, int os, const float* lse,
                         float* delta, float* gqkv, int N, int T, hipStream_t st)
{
    dim3 grid((T + AB_ROWS - 1) / AB_ROWS, 4, N);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, st, qkv, qs, go, os, lse, gqkv, delta, T, 0.25f);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), 0, st, qkv, qs, go, os, lse, delta, gqkv, T, 0.25f);
    return hipGetLastError() == hipSuccess ? 0 : 62;
}

// ---- granular C-ABI (parity tests) ----
// TEST ENTRY (include/ssie_debug.h): the bf16 attention of the enhance-only path on its own; out = bf16 (N, T, 64)
extern "C" size_t ssie_debug_attention_bf16_scratch_bytes(int N, int T) { return ssie_attn_bf16_scratch_bytes(N, T); }
extern "C" int ssie_debug_attention_fwd_bf16(const float* qkv, void* out_bf16, int N, int T, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!qkv || !out_bf16 || N < 1 || T < 1) return 1;
    return ssie_launch_attn_fwd_bf16(qkv, 192, out_bf16, 64, N, T, (hipStream_t)stream, scratch, scratch_bytes) ? 4 : 0;
}

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
