// Micro-benchmark (dev tool): can a wave's VALU instructions execute underneath its own MFMAs?  One wave per SIMD (or two with
// wgs/cu = 2), a loop of independent v_mfma_f32_32x32x2_f32 with K independent v_add_f32 after each one.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int K>
__global__ __launch_bounds__(256) void k(float* out, int iters, float c)
{
    extern __shared__ float smem[];
    const int tid = threadIdx.x;
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = (float)(tid + i);
    float a0 = (float)tid * 0.001f, b0 = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m] = MFMA32(a0, b0, acc[m]);
#pragma unroll
            for (int v = 0; v < K; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(m * K + v) & 15]) : "v"(c));
        }
    }
    float s = 0.f;
    for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int K>
void run(int wgs_per_cu, int iters)
{
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipFuncSetAttribute((const void*)k<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    size_t lds = wgs_per_cu == 1 ? 100 * 1024 : 70 * 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((k<K>), dim3(grid), dim3(256), lds, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<K>), dim3(grid), dim3(256), lds, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double mf = (double)grid * 4 * iters * 8.0;
    printf("K=%2d valu/mfma  wgs/cu %d  %8.3f ms  %7.1f TF   %6.1f ns per MFMA per wave\n", K, wgs_per_cu, ms,
           mf * 2 * 32 * 32 * 2 / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * 8.0) / 1.0);
    hipFree(out);
}

int main()
{
    const int it = 20000;
    for (int w = 1; w <= 2; ++w) {
        run<0>(w, it); run<2>(w, it); run<4>(w, it); run<8>(w, it); run<12>(w, it); run<16>(w, it); run<24>(w, it);
    }
    return 0;
}
