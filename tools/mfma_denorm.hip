// Does v_mfma_f32_32x32x16_f16 honour f16 SUBNORMAL inputs on gfx950, or flush them to zero?
// (The split-f16 arithmetic stores lo = x - f16(x); for |x| < 0.125 that lo is an f16 subnormal.)
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_denorm.hip -o tools/mfma_denorm && tools/mfma_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(float* out, unsigned short a_bits, unsigned short b_bits)
{
    f16x8 a, b;
    const _Float16 av = __builtin_bit_cast(_Float16, a_bits), bv = __builtin_bit_cast(_Float16, b_bits);
    for (int i = 0; i < 8; ++i) { a[i] = av; b[i] = bv; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
    // VALU conversion of an fp32 value in the f16 subnormal range (what the loader does for lo)
    if (threadIdx.x == 0) { const float x = 3.0e-6f; const _Float16 h = (_Float16)x; out[1] = (float)h; }
}

int main()
{
    float* d; float h[2];
    hipMalloc(&d, 8);
    struct { const char* what; unsigned short a, b; double expect; } cases[] = {
        {"A = 2^-20 (subnormal), B = 1024", 0x0010, 0x6400, 16 * 9.5367431640625e-07 * 1024},
        {"A = 1024, B = 2^-20 (subnormal)", 0x6400, 0x0010, 16 * 9.5367431640625e-07 * 1024},
        {"A = 2^-14 (smallest normal), B = 1", 0x0400, 0x3C00, 16 * 6.103515625e-05},
        {"A = 2^-24 (smallest subnormal), B = 1", 0x0001, 0x3C00, 16 * 5.9604644775390625e-08},
    };
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c.a, c.b);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("%-42s mfma sum over k=16: %.9g (exact %.9g) -> %s\n", c.what, h[0], c.expect, h[0] == (float)c.expect ? "honoured" : (h[0] == 0.f ? "FLUSHED" : "other"));
    }
    printf("VALU cvt fp32 3.0e-6 -> f16 -> fp32: %.9g (%s)\n", h[1], h[1] == 0.f ? "FLUSHED" : "subnormal kept");
    return 0;
}
