// Where is the power wall of the f16 matrix pipe for THIS kernel's operand statistics?  (development aid, round 4)
// Pure v_mfma_f32_32x32x16_f16 chains from registers -- no LDS, no loads, 12 waves per CU like the halo kernel -- with operands drawn like
// the split-f16 arithmetic's: hi = f16(x), lo = f16(x - hi) of x ~ N(0, 1) activations / weights scaled into [2^13, 2^14).
// One MFMA issues every 32 cycles (8 passes x 4): 100 % pipe occupancy by construction, so TF / 2500 * 2.4 GHz = the clock the chip
// sustains under that load.  MODE 0: all-zero operands, 1: hi x hi only, 2: the kernel's mix (al*bh, ah*bl, ah*bh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const f16x8* __restrict__ src, float* out, int iters)
{
    const int tid = threadIdx.x;
    __shared__ f16x8 lds[MODE >= 3 ? 3072 : 1];          // MODE 3 / 4: the operands come out of LDS, 16 ds_read_b128 per 24 MFMAs like igemm_halo_kernel<2,2>
    if (MODE >= 3) {
        for (int i = tid; i < 3072; i += 256) lds[i] = src[(i & 4095) + 4096 * (i % 4)];
        __syncthreads();
    }
    // per lane: 2 A fragments (hi, lo) and 2 B fragments (hi, lo) for 2 x 2 blocks
    f16x8 ah[2], al[2], bh[2], bl[2];
    for (int i = 0; i < 2; ++i) {
        ah[i] = src[(tid * 8 + i * 4 + 0) & 4095]; al[i] = src[4096 + ((tid * 8 + i * 4 + 1) & 4095)];
        bh[i] = src[8192 + ((tid * 8 + i * 4 + 2) & 4095)]; bl[i] = src[12288 + ((tid * 8 + i * 4 + 3) & 4095)];
    }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 3) {
            const int o = (it * 37 + tid) & 1023;            // conflict-free: consecutive lanes, consecutive 16-byte slots
#pragma unroll
            for (int q = 0; q < 2; ++q) {                      // two 16-deep blocks per K-step: 8 fragments each
                ah[0] = lds[o + q * 64]; ah[1] = lds[o + 256 + q * 64]; al[0] = lds[o + 512 + q * 64]; al[1] = lds[o + 768 + q * 64];
                bh[0] = lds[o + 1024 + q * 64]; bh[1] = lds[o + 1280 + q * 64]; bl[0] = lds[o + 1536 + q * 64]; bl[1] = lds[o + 1792 + q * 64];
                if (q == 0) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                        }
                }
            }
            if (MODE == 4) { __syncthreads(); __syncthreads(); }      // the kernel's two barriers per K-step
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (MODE >= 2) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                } else {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            }
        // rotate the operands so that consecutive MFMAs see different data (as a K loop does)
        f16x8 t = ah[0]; ah[0] = ah[1]; ah[1] = t; t = bh[0]; bh[0] = bh[1]; bh[1] = t;
        t = al[0]; al[0] = al[1]; al[1] = t; t = bl[0]; bl[0] = bl[1]; bl[1] = t;
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

static float gauss() { float u = (rand() + 1.f) / (RAND_MAX + 2.f), v = (rand() + 1.f) / (RAND_MAX + 2.f); return sqrtf(-2 * logf(u)) * cosf(6.2831853f * v); }

template <int MODE>
void run(const char* name, const f16x8* src, int wgs_per_cu)
{
    float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
    const int iters = 4000, blocks = 256 * wgs_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, 10);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    const double flops = (double)blocks * 4 * iters * (MODE >= 3 ? 24 : 12) * 32768.0;
    const double tf = flops / best / 1e9;
    printf("%-44s %d WG/CU: %8.3f ms %7.1f TF  = %.2f GHz at 100%% pipe occupancy\n", name, wgs_per_cu, best, tf, tf / 2516.6 * 2.4);
    hipFree(out);
}
int main()
{
    std::vector<_Float16> h(4 * 4096 * 8);
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) {
        const float x = gauss(), w = gauss() * 12000.f;
        const _Float16 xh = (_Float16)x, wh = (_Float16)w;
        h[i] = xh; h[4096 * 8 + i] = (_Float16)(x - (float)xh);
        h[2 * 4096 * 8 + i] = wh; h[3 * 4096 * 8 + i] = (_Float16)(w - (float)wh);
    }
    f16x8* src; hipMalloc(&src, h.size() * 2);
    std::vector<_Float16> z(h.size(), (_Float16)0.f);
    hipMemcpy(src, z.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<1>("all-zero operands", src, 3);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int w : {1, 2, 3}) {
        run<1>("hi x hi (gaussian activations x weights)", src, w);
        run<2>("split-f16 mix (al*bh, ah*bl, ah*bh)", src, w);
    }
    run<3>("  mix + 16 ds_read_b128 per 24 MFMAs", src, 3);
    run<4>("  ... + two barriers per 24 MFMAs", src, 3);
    return 0;
}
