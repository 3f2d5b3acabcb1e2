// Does straight-line code larger than the instruction cache slow a lone wave down?  A loop whose body is N dependent-free fp64 FMA pairs
// (unrolled at compile time), run for the same total number of instructions with N small (fits the 64 KB I-cache) and N large.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/icache tools/icache.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__global__ __launch_bounds__(64) void body(double* out, long long* t, int reps)
{
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001 + i;
    const long long c0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int k = 0; k < N; ++k) a[k & 7] = __builtin_fma(a[k & 7], 1.0000001, 0.5 + k);      // 8 independent chains
    }
    const long long c1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) t[0] = c1 - c0;
}
template <int N> void run(double* o, long long* t, long long total)
{
    const int reps = (int)(total / N);
    long long h;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(body<N>, dim3(1), dim3(64), 0, 0, o, t, reps); hipDeviceSynchronize(); }
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("body of %6d FMAs (~%4d KB of code): %.2f cycles per instruction\n", N, N * 16 / 1024, (double)h / ((double)reps * N));
}
int main()
{
    double* o; long long* t; hipMalloc(&o, 512); hipMalloc(&t, 8);
    const long long total = 1 << 22;
    run<512>(o, t, total); run<2048>(o, t, total); run<4096>(o, t, total); run<8192>(o, t, total); run<16384>(o, t, total); run<32768>(o, t, total);
    return 0;
}
