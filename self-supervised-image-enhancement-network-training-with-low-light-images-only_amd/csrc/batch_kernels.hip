// Device-side batch assembly: random crop + one of the 8 flip/rotate augmentations, written straight into the
// NHWC batch buffer the plan ingests.  Replaces the host loop of /root/reference/model.py:301-310 and
// utils.data_augmentation (utils.py:7-34); cubes stay resident in HBM as (H, W, C) fp32 arrays, exactly the
// layout load_hsi returns (utils.py:36-57), so no host round trip and no H2D copy per step.
//
// np.rot90 / np.flipud on a square P x P crop m, dst[i][j] =
//   0: m[i][j]            1: m[P-1-i][j]        2: m[j][P-1-i]        3: m[j][i]
//   4: m[P-1-i][P-1-j]    5: m[i][P-1-j]        6: m[P-1-j][i]        7: m[P-1-j][P-1-i]
#include "ssie_common.h"
#include "../../include/ssie_hip.h"

__host__ __device__ inline void ssie_aug_src(int mode, int P, int i, int j, int& si, int& sj)
{
    switch (mode) {
        case 0: si = i; sj = j; break;
        case 1: si = P - 1 - i; sj = j; break;
        case 2: si = j; sj = P - 1 - i; break;
        case 3: si = j; sj = i; break;
        case 4: si = P - 1 - i; sj = P - 1 - j; break;
        case 5: si = i; sj = P - 1 - j; break;
        case 6: si = P - 1 - j; sj = i; break;
        default: si = P - 1 - j; sj = P - 1 - i; break;
    }
}

struct CropDesc { const float* cube; int H, W, x0, y0, mode; };   // x0 = row offset, y0 = column offset (model.py:306-309)

__global__ void assemble_batch_kernel(const CropDesc* __restrict__ crops, float* __restrict__ out, int P, int C, int cs)
{
    const CropDesc d = crops[blockIdx.y];
    float* dst = out + (size_t)blockIdx.y * P * P * cs;
    const long total = (long)P * P * cs;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cs); const long px = idx / cs;
        const int j = (int)(px % P), i = (int)(px / P);
        float v = 0.f;
        if (c < C) {
            int si, sj; ssie_aug_src(d.mode, P, i, j, si, sj);
            v = d.cube[((size_t)(d.x0 + si) * d.W + (d.y0 + sj)) * C + c];
        }
        dst[idx] = v;
    }
}

// crops_dev: n descriptors in device memory {cube ptr, H, W, x0, y0, mode}; out: (n, P, P, cs) NHWC, cs >= C padded with zeros
extern "C" int ssie_assemble_batch(const void* crops_dev, int n, float* out, int P, int C, int cs, void* stream)
{
    if (!crops_dev || !out || n < 1 || P < 1 || C < 1 || cs < C) return SSIE_E_ARG;
    long total = (long)P * P * cs;
    int gx = (int)((total + 255) / 256); if (gx > 256) gx = 256;
    hipLaunchKernelGGL(assemble_batch_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, (const CropDesc*)crops_dev, out, P, C, cs);
    return hipGetLastError() == hipSuccess ? 0 : SSIE_E_LAUNCH;
}

// host helper for tests: the index map above
extern "C" int ssie_aug_source_index(int mode, int P, int i, int j, int* si, int* sj)
{
    if (mode < 0 || mode > 7 || !si || !sj) return SSIE_E_ARG;
    ssie_aug_src(mode, P, i, j, *si, *sj);
    return 0;
}
