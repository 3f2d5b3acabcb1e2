// How fast can ONE wave stream a weight panel with the two lane layouts?  (a) MFMA operand layout: lane (l & 31) = row, (l >> 5) = 16-byte
// half: four b128 loads per 128-byte record, every instruction touches 32 rows; (b) row-major: 8 lanes cover one 128-byte record, every
// instruction touches 8 rows.  One wave per block, `blocks` blocks (each on its own panel rows), D K-steps in flight, 400 K-steps.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/load_pattern tools/load_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 12800, STEPS = K / 32, D = 6;
template <int MODE>
__global__ __launch_bounds__(64) void stream(const float* w, unsigned w_bytes, unsigned* out, long long* t)
{
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, w_bytes, 0x00020000);
    const unsigned row0 = blockIdx.x * 32;
    unsigned base[4];
    if (MODE == 0) { for (int j = 0; j < 4; ++j) base[j] = ((row0 + (lane & 31)) * K) * 4u + (lane >> 5) * 16u + (j == 1 ? 64 : j == 2 ? 32 : j == 3 ? 96 : 0); }
    else { for (int j = 0; j < 4; ++j) base[j] = ((row0 + 8 * j + (lane >> 3)) * K) * 4u + (lane & 7) * 16u; }
    u32x4 r[D][4];
    unsigned acc = 0;
    const long long w0 = wall_clock64();
#pragma unroll
    for (int s = 0; s < D; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) r[s][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, base[j] + s * 128, 0, 0));
    for (int ks = 0; ks + D <= STEPS; ks += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc ^= r[s][j][0] ^ r[s][j][1] ^ r[s][j][2] ^ r[s][j][3];
            const unsigned ko = (unsigned)(ks + s + D) * 128u;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[s][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, base[j] + ko, 0, 0));
        }
    }
    const long long w1 = wall_clock64();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) t[blockIdx.x] = w1 - w0;
}
int main()
{
    const size_t rows = 512, bytes = rows * K * 4;
    float* w; unsigned* o; long long* t;
    hipMalloc(&w, bytes); hipMemset(w, 1, bytes); hipMalloc(&o, 4 * 64 * 64); hipMalloc(&t, 8 * 64);
    long long h[16];
    for (int blocks : {1, 16}) {
        for (int mode = 0; mode < 2; ++mode)
            for (int rep = 0; rep < 3; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(stream<0>, dim3(blocks), dim3(64), 0, 0, w, (unsigned)bytes, o, t);
                else hipLaunchKernelGGL(stream<1>, dim3(blocks), dim3(64), 0, 0, w, (unsigned)bytes, o, t);
                hipDeviceSynchronize();
                hipMemcpy(h, t, 8 * blocks, hipMemcpyDeviceToHost);
                if (rep == 2) printf("blocks %2d mode %d (%s): %.1f ns per K-step of 4 KB (block 0), %.1f GB/s per wave\n", blocks, mode,
                                     mode == 0 ? "MFMA operand layout, 32 rows / instr" : "row-major, 8 rows / instr", h[0] * 10.0 / STEPS, 4096.0 / (h[0] * 10.0 / STEPS));
            }
    }
    return 0;
}
