// What bounds a tile's weight stream at ~235 ns per K-step?  32 (or more) waves, each streaming the 128-byte K-step records of its own 32
// panel rows (row-major form: 8 lanes per record), D steps in flight, in (a) sequential K order or (b) the conv4 schedule order
// (consecutive steps 2 KB apart inside a row), cold (first touch after a 512 MB flush) or warm (second pass over the same rows).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/load_pattern2 tools/load_pattern2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 12800, STEPS = K / 32, D = 8;
__global__ __launch_bounds__(64) void stream(const float* w, unsigned w_bytes, const int* order, unsigned* out, long long* t, int rows_per_block, int contig)
{
    const int lane = threadIdx.x;
    __shared__ int s_ord[STEPS + 64];
    for (int i = lane; i < STEPS + 64; i += 64) s_ord[i] = i < STEPS ? order[i] : 0;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, w_bytes, 0x00020000);
    const unsigned row0 = blockIdx.x * rows_per_block;
    unsigned base[4];
    for (int j = 0; j < 4; ++j) base[j] = ((row0 + 8 * j + (lane >> 3)) * K) * 4u + (lane & 7) * 16u;
    // contig: the tile's K-step records stored as ONE 4 KB block per step (fragment-major panel): 1 KB per instruction, 4 KB per step, sequential
    if (contig) for (int j = 0; j < 4; ++j) base[j] = blockIdx.x * (unsigned)(STEPS * 4096) + j * 1024u + lane * 16u;
    const unsigned step_bytes = contig ? 4096u : 128u;
    u32x4 r[D][4];
    unsigned acc = 0;
    const long long w0 = wall_clock64();
#pragma unroll
    for (int s = 0; s < D; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) r[s][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, base[j] + s_ord[s] * step_bytes, 0, 0));
    for (int ks = 0; ks + D <= STEPS; ks += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc ^= r[s][j][0] ^ r[s][j][1] ^ r[s][j][2] ^ r[s][j][3];
            const int nx = ks + s + D;
            const unsigned ko = nx < STEPS ? (unsigned)__builtin_amdgcn_readfirstlane(s_ord[nx]) * step_bytes : 0x80000000u;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[s][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, base[j] + ko, 0, 0));
        }
    }
    const long long w1 = wall_clock64();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) t[blockIdx.x] = w1 - w0;
}
__global__ void flush(float* p, size_t n) { for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] += 1.f; }
int main()
{
    const size_t rows = 1024 * 8, bytes = rows * K * 4;
    float* w; unsigned* o; long long* t; int* ord; float* big;
    hipMalloc(&w, bytes); hipMemset(w, 1, bytes); hipMalloc(&o, 4 * 64 * 1024); hipMalloc(&t, 8 * 1024); hipMalloc(&ord, 4 * STEPS);
    hipMalloc(&big, 1ull << 30); hipMemset(big, 0, 1ull << 30);
    std::vector<int> seq(STEPS), sch(STEPS);
    for (int i = 0; i < STEPS; ++i) seq[i] = i;
    for (int i = 0, c = 0; c < 16; ++c) for (int tap = 0; tap < 25; ++tap) sch[i++] = tap * 16 + c;      // (slice, tap): panel K-step = tap * 16 + slice
    std::vector<long long> h(1024);
    for (int blocks : {1, 32, 256}) {
        for (int mode = 0; mode < 3; ++mode) {
            hipMemcpy(ord, mode == 1 ? sch.data() : seq.data(), 4 * STEPS, hipMemcpyHostToDevice);
            for (int warm = 0; warm < 2; ++warm) {
                if (!warm) { hipLaunchKernelGGL(flush, dim3(4096), dim3(256), 0, 0, big, (size_t)(1ull << 28)); hipDeviceSynchronize(); }
                hipLaunchKernelGGL(stream, dim3(blocks), dim3(64), 0, 0, w, (unsigned)bytes, ord, o, t, blocks <= 32 ? 32 : 4, mode == 2 ? 1 : 0);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), t, 8 * blocks, hipMemcpyDeviceToHost);
                long long mx = 0; for (int b = 0; b < blocks; ++b) mx = h[b] > mx ? h[b] : mx;
                printf("blocks %3d  %-10s %-5s: %.1f ns per K-step (slowest wave), %.1f GB/s per wave, %.0f GB/s in total\n", blocks, mode == 1 ? "schedule" : mode == 2 ? "contiguous" : "sequential", warm ? "warm" : "cold",
                       mx * 10.0 / STEPS, 4096.0 / (mx * 10.0 / STEPS), blocks * 4096.0 / (mx * 10.0 / STEPS));
            }
        }
    }
    return 0;
}
