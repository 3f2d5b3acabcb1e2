// Dependent-chain rate of v_mfma_f32_32x32x16_f16 on one wave, and the shader clock a lone small kernel actually runs at
// (clock64() = shader cycles, wall_clock64() = 100 MHz constant).  Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_chain tools/mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(64) void chain(float* out, long long* t, int n)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.01f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = w1 - w0; }
}
int main()
{
    float* o; long long* t; hipMalloc(&o, 4 * 64 * 1024); hipMalloc(&t, 16 * 1024);
    long long h[2];
    for (int blocks : {1, 32, 1024}) {
        for (int rep = 0; rep < 3; ++rep) {
            const int n = 2000;
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, o, t, n);
            hipDeviceSynchronize();
            hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            printf("blocks %4d: %lld shader cycles, %lld wall ticks (100 MHz) for %d dependent MFMAs: %.1f cycles/MFMA, clock %.0f MHz, %.1f ns/MFMA\n", blocks,
                   h[0], h[1], 4 * n, (double)h[0] / (4 * n), (double)h[0] / h[1] * 100.0, (double)h[1] * 10.0 / (4 * n));
        }
    }
    return 0;
}
