#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__global__ void k(float* o) {
    f32x2 a = {1.f + threadIdx.x, 10.f}, b = {0.25f, 3.f};
    f32x2 s = pk_add(a, b), d = pk_sub(a, b);
    if (threadIdx.x == 0) { o[0] = s.x; o[1] = s.y; o[2] = d.x; o[3] = d.y; }
}
int main() { float* o; (void)hipMalloc(&o, 16); k<<<1, 64>>>(o); float h[4]; (void)hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
    printf("add %g %g (want 1.25 13)  sub %g %g (want 0.75 7)\n", h[0], h[1], h[2], h[3]); return 0; }
