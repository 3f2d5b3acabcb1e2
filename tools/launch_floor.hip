// Floor of a dependent kernel chain on one stream: empty kernels with small / large kernarg blocks, and a kernel that reads its
// kernarg block through dynamic indices (what igemm_stream_kernel's prologue does).  Build: hipcc --offload-arch=gfx950 -O3 -o tools/launch_floor tools/launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct Big { int v[600]; };
__global__ void k_small(int* out, int x) { if (x == 12345) out[0] = x; }
__global__ void k_big(int* out, Big b) { if (b.v[0] == 12345) out[0] = b.v[1]; }
__global__ void k_big_dyn(int* out, Big b)
{
    int s = 0, i = threadIdx.x & 7;
    for (int k = 0; k < 25; ++k) { i = b.v[i] & 511; s += i; }     // 25 dependent reads of the kernarg block
    if (s == 12345) out[0] = s;
}
int main()
{
    int* d; hipMalloc(&d, 4);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    Big big; for (int i = 0; i < 600; ++i) big.v[i] = (i * 7 + 3) % 600;
    const int R = 200;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a, st);
            for (int i = 0; i < R; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k_small, dim3(64), dim3(64), 0, st, d, i);
                else if (mode == 1) hipLaunchKernelGGL(k_big, dim3(64), dim3(64), 0, st, d, big);
                else hipLaunchKernelGGL(k_big_dyn, dim3(64), dim3(64), 0, st, d, big);
            }
            hipEventRecord(b, st);
            hipStreamSynchronize(st);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("mode %d (%s): %.2f us per launch  (HIP_FORCE_DEV_KERNARG=%s)\n", mode, mode == 0 ? "small kernarg" : mode == 1 ? "2.4 KB kernarg" : "2.4 KB kernarg, 25 dependent reads",
                            ms * 1e3 / R, getenv("HIP_FORCE_DEV_KERNARG") ? getenv("HIP_FORCE_DEV_KERNARG") : "unset");
        }
    }
    return 0;
}
