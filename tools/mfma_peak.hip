// fp32 MFMA ceiling probes (development aid): pure MFMA, MFMA + LDS reads like the igemm inner loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float smem[256 * 36];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 256 * 36; i += 256) smem[i] = seed * (float)((i * 37) % 101 - 50) * 0.01f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 a[2], b[2];
    a[0] = a[1] = b[0] = b[1] = f32x4{seed, seed * 2, seed * 3, seed * 4};
    const float* As = smem + (lane & 31) * 36 + (lane >> 5) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 32; kk += 8) {
            if (MODE >= 1) {
                a[0] = *reinterpret_cast<const f32x4*>(As + kk);
                a[1] = *reinterpret_cast<const f32x4*>(As + 32 * 36 + kk);
                b[0] = *reinterpret_cast<const f32x4*>(As + 128 * 36 + kk);
                b[1] = *reinterpret_cast<const f32x4*>(As + 160 * 36 + kk);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, float seed)
{
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 2000, blocks = 256 * 3;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, seed);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, seed);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flops = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("%-34s seed %.2f: %.3f ms %.1f TF\n", name, seed, ms, flops / ms / 1e9);
    hipFree(out);
}
int main()
{
    for (float seed : {0.f, 0.37f}) {
        run<0>("pure MFMA (regs only)", seed);
        run<1>("MFMA + 4 ds_read_b128 / 16 MFMA", seed);
        run<2>("  ... + barrier per 64 MFMA", seed);
    }
    return 0;
}
