// Micro-benchmark (dev tool): what fp32-MFMA rate does the fprop inner-loop instruction mix reach?
//  mode 0: MFMAs only, operands in registers, NACC independent accumulators
//  mode 1: + the kernel's ds_read_b128 fragment reads (A swizzled halo pattern, B linear), double-buffered
//  mode 2: mode 1 + one __syncthreads() every 144 MFMAs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* As = (f32x4*)smem;            // 18*10*4 float4
    f32x4* Bs = As + 18 * 10 * 4;        // 36*64 float4
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, li = lane & 31;
    for (int i = tid; i < 18 * 10 * 4 + 36 * 64; i += 256) ((f32x4*)smem)[i] = f32x4{(float)(i & 7) * 0.01f, 0.5f, 0.25f, 0.125f};
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int wn = wave & 1, wm = wave >> 1;
    int pixbase[2];
    for (int m = 0; m < 2; ++m) pixbase[m] = (2 * (wm * 2 + m) + (li >> 4)) * 18 + (li & 15);
    f32x4 a0 = As[lane], b0 = Bs[lane];
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 18; ++t)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q % NACC] = MFMA32(a0[q & 3], b0[q & 3], acc[q % NACC]);
        } else {
            f32x4 bX, bY, aX[2], aY[2];
            int off = 0;
            bX = Bs[(0 * 4 + 0 * 2 + h) * 64 + wn * 32 + li];
            for (int m = 0; m < 2; ++m) { int hp = pixbase[m] + off; aX[m] = As[hp * 4 + ((0 * 2 + h) ^ ((hp >> 2) & 3))]; }
            for (int tl = 0; tl < 9; ++tl) {
                const int offn = ((tl + 1) % 9 / 3) * 18 + (tl + 1) % 3;
                bY = Bs[(tl * 4 + 2 + h) * 64 + wn * 32 + li];
                for (int m = 0; m < 2; ++m) { int hp = pixbase[m] + off; aY[m] = As[hp * 4 + ((2 + h) ^ ((hp >> 2) & 3))]; }
                for (int m = 0; m < 2; ++m) { acc[m % NACC] = MFMA32(aX[m].x, bX.x, acc[m % NACC]); acc[m % NACC] = MFMA32(aX[m].y, bX.y, acc[m % NACC]); acc[m % NACC] = MFMA32(aX[m].z, bX.z, acc[m % NACC]); acc[m % NACC] = MFMA32(aX[m].w, bX.w, acc[m % NACC]); }
                if (tl + 1 < 9) {
                    bX = Bs[((tl + 1) * 4 + h) * 64 + wn * 32 + li];
                    for (int m = 0; m < 2; ++m) { int hp = pixbase[m] + offn; aX[m] = As[hp * 4 + (h ^ ((hp >> 2) & 3))]; }
                }
                for (int m = 0; m < 2; ++m) { acc[m % NACC] = MFMA32(aY[m].x, bY.x, acc[m % NACC]); acc[m % NACC] = MFMA32(aY[m].y, bY.y, acc[m % NACC]); acc[m % NACC] = MFMA32(aY[m].z, bY.z, acc[m % NACC]); acc[m % NACC] = MFMA32(aY[m].w, bY.w, acc[m % NACC]); }
                off = offn;
            }
            if (MODE == 2) __syncthreads();
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int NACC>
void run(const char* name, int wgs_per_cu, int iters)
{
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    size_t lds = (18 * 10 * 4 + 36 * 64) * 16;
    hipFuncSetAttribute((const void*)k<MODE, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    size_t pad = wgs_per_cu == 1 ? 100 * 1024 : (wgs_per_cu == 2 ? 70 * 1024 : 0);   // force the residency
    if (lds < pad) lds = pad;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(grid), dim3(256), lds, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, NACC>), dim3(grid), dim3(256), lds, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)grid * 4 * iters * 144.0 * 2 * 32 * 32 * 2;
    printf("%-34s wgs/cu %d  %8.3f ms  %7.1f TF\n", name, wgs_per_cu, ms, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main()
{
    const int it = 400;
    for (int w = 1; w <= 3; ++w) {
        run<0, 1>("mfma only, 1 acc chain", w, it);
        run<0, 2>("mfma only, 2 acc", w, it);
        run<0, 4>("mfma only, 4 acc", w, it);
        run<1, 2>("+ LDS fragment reads (pipelined)", w, it);
        run<2, 2>("+ LDS reads + barrier/144 MFMA", w, it);
    }
    return 0;
}
