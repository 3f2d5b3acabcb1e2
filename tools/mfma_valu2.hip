// Asymmetric form of tools/mfma_valu.hip: waves 0-3 of a 512-thread workgroup (one per SIMD) issue ONLY MFMAs, waves 4-7 (their SIMD neighbours) ONLY v_fma_f32.
// If the SIMD can run a VALU instruction of one wave under the MFMA of another, each kind takes as long as it does alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void split(float* out, long long* t, int n_mfma, int n_valu, int mode)
{
    // mode 0: both kinds; 1: MFMA waves only work; 2: VALU waves only work
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.01f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 1.f; }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    const float c1 = 0.999f, c2 = 0.001f;
    __syncthreads();
    const long long t0 = clock64();
    if (wave < 4) {
        if (mode != 2)
            for (int it = 0; it < n_mfma; ++it) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            }
    } else if (mode != 1) {
        for (int it = 0; it < n_valu; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(c1), "v"(c2));
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + v[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 8 + wave] = t1 - t0;
}

int main()
{
    float* o; long long* t;
    hipMalloc(&o, 4 * 512 * 256); hipMalloc(&t, 8 * 8 * 256);
    long long h[8 * 256];
    const int n_mfma = 4000;
    for (int n_valu : {1000, 2000, 4000, 8000}) {
        for (int mode : {1, 2, 0}) {
            for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(split, dim3(256), dim3(512), 0, 0, o, t, n_mfma, n_valu, mode); (void)hipDeviceSynchronize(); }
            (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
            double m = 0, v = 0;
            for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w] / (256 * 4);
            printf("%s: %d MFMAs per MFMA wave, %d v_fma per VALU wave: MFMA waves %.0f cycles (%.1f per MFMA), VALU waves %.0f cycles (%.2f per v_fma)\n",
                   mode == 0 ? "both      " : mode == 1 ? "MFMA alone" : "VALU alone", 2 * n_mfma, 16 * n_valu, m, m / (2 * n_mfma), v, v / (16.0 * n_valu));
        }
    }
    return 0;
}
