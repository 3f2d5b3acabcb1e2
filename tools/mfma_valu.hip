// Can vector ALU instructions of the SAME SIMD issue under a running v_mfma_f32_32x32x16_f16?  One wave (or two) per SIMD alternates one MFMA with K independent
// v_fma_f32 (two accumulators, sixteen independent VALU registers: no dependency stalls); shader cycles per MFMA against K tell whether the K x 4 VALU cycles
// hide under the MFMA's passes or add to them.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_valu tools/mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int K>
__global__ __launch_bounds__(512) void mix(float* out, long long* t, int n)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.01f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 1.f; }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    const float c1 = 0.999f, c2 = 0.001f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(h * K + k) & 15]) : "v"(c1), "v"(c2));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + v[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <int K>
void run(float* o, long long* t, int threads)
{
    const int n = 4000, blocks = 256;
    long long h[256];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(mix<K>, dim3(blocks), dim3(threads), 0, 0, o, t, n);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks; ++i) s += (double)h[i];
    printf("waves/SIMD %d  K = %2d VALU per MFMA: %.1f shader cycles per MFMA of one wave (%.1f per SIMD-MFMA)\n", threads / 256, K, s / blocks / (2.0 * n), s / blocks / (2.0 * n) / (threads / 256));
}

int main()
{
    float* o; long long* t;
    hipMalloc(&o, 4 * 512 * 256); hipMalloc(&t, 8 * 256);
    for (int threads : {256, 512}) {
        run<0>(o, t, threads); run<1>(o, t, threads); run<2>(o, t, threads); run<4>(o, t, threads); run<6>(o, t, threads);
        run<8>(o, t, threads); run<12>(o, t, threads); run<16>(o, t, threads); run<24>(o, t, threads);
    }
    return 0;
}
