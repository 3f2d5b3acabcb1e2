// Experimental variants of the implicit-GEMM kernel on one big-layer shape (development aid).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipix2pose_amd/csrc tools/igemm_exp.hip -o tools/igemm_exp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kernels.h"
using namespace p2p;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// BK: k-step; NBUF: LDS buffers; MID: store next tile to LDS in the middle of the MFMA block; PRIO: setprio
__device__ unsigned long long g_clk[4];
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int BK, int NBUF, int MID, int PRIO, int EPI = 1, int ABL = 0>
__global__ __launch_bounds__(256) void kexp(const IgemmParams p)
{
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    constexpr int WGM = 2, WGN = 2, TM = 2, TN = 2;
    constexpr int BM = 128, BN = 128;
    constexpr int LD = BK + 4;
    constexpr int CPR = BK / 4;              // float4 columns per row
    constexpr int RPP = 256 / CPR;           // rows per pass
    constexpr int A_PASSES = BM / RPP, B_PASSES = BN / RPP;
    constexpr int STG = NBUF * (BM + BN) * LD;
    constexpr int CT = EPI == 0 ? 0 : (EPI == 1 ? BM * (BN + 4) : 64 * (BN + 4));
    __shared__ __attribute__((aligned(16))) float smem[STG > CT ? STG : CT];
    __shared__ int row_base[BM], row_yx[BM], row_out[BM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN, li = lane & 31, lk = lane >> 5;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int nblk = gridDim.x;
    int t;
    { const int b = blockIdx.x; const int q = nblk >> 3, r = nblk & 7; const int xcd = b & 7, idx = b >> 3;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
    const int tile_n = t % tiles_n, tile_m = t / tiles_n, m0 = tile_m * BM, n0 = tile_n * BN;
    const int HgWg = p.Hg * p.Wg;
    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r; int base = -1, yx = 0, op = -1;
        if (m < p.M) { const int n = m / HgWg; const int rem = m - n * HgWg; const int gy = rem / p.Wg; const int gx = rem - gy * p.Wg;
            base = n * p.Hin * p.Win; yx = ((gy * p.in_stride) << 16) | (gx * p.in_stride); op = (n * p.Hout + gy * p.os + p.oy) * p.Wout + gx * p.os + p.ox; }
        row_base[r] = base; row_yx[r] = yx; row_out[r] = op;
    }
    __syncthreads();
    const int lrow = tid / CPR, lcol = (tid % CPR) * 4;
    int a_base[A_PASSES], a_yx[A_PASSES];
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j) { a_base[j] = row_base[lrow + RPP * j]; a_yx[j] = row_yx[lrow + RPP * j]; }
    const float* wrow = p.w + (size_t)(n0 + lrow) * p.K + lcol;
    const int ksteps = p.K / BK, cpt = (p.seg[0].C) / BK;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[A_PASSES], rb[B_PASSES];
    auto gload = [&](int ks) {
        const int tap = ks / cpt; const int chunk = ks - tap * cpt;
        const IgemmSeg sg = p.seg[0];
        const int c = chunk * BK + lcol;
        const int dy = p.dy[tap], dx = p.dx[tap];
#pragma unroll
        for (int j = 0; j < A_PASSES; ++j) {
            const int iy = (a_yx[j] >> 16) + dy, ix = (a_yx[j] & 0xffff) + dx;
            const bool ok = a_base[j] >= 0 && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            size_t off = (size_t)(a_base[j] + iy * p.Win + ix) * sg.cstride + c;
            if (ABL & 4) off &= 1023;
            if (ok) v = *reinterpret_cast<const f32x4*>(sg.ptr + off);
            ra[j] = v;
        }
        const float* wp = wrow + (size_t)ks * BK;
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) { size_t off = (size_t)(RPP * j) * p.K; if (ABL & 4) off &= 1023; rb[j] = *reinterpret_cast<const f32x4*>((ABL & 4 ? p.w : wp) + off); }
    };
    auto lstore = [&](int buf) {
        float* As = smem + buf * (BM + BN) * LD; float* Bs = As + BM * LD;
        if (ABL & 16) {
            // row image: [hi16 x 32 | lo16 x 32] (128 B); this thread owns k = lcol..lcol+3
#pragma unroll
            for (int j = 0; j < A_PASSES; ++j) {
                typedef __fp16 h2 __attribute__((ext_vector_type(2)));
                const h2 h01 = __builtin_amdgcn_cvt_pkrtz(ra[j][0], ra[j][1]), h23 = __builtin_amdgcn_cvt_pkrtz(ra[j][2], ra[j][3]);
                const h2 l01 = __builtin_amdgcn_cvt_pkrtz(ra[j][0] - (float)h01[0], ra[j][1] - (float)h01[1]);
                const h2 l23 = __builtin_amdgcn_cvt_pkrtz(ra[j][2] - (float)h23[0], ra[j][3] - (float)h23[1]);
                char* row = reinterpret_cast<char*>(As + (lrow + RPP * j) * LD);
                *reinterpret_cast<uint2*>(row + lcol * 2) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
                *reinterpret_cast<uint2*>(row + 64 + lcol * 2) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
            }
        } else
#pragma unroll
        for (int j = 0; j < A_PASSES; ++j) *reinterpret_cast<f32x4*>(As + (lrow + RPP * j) * LD + lcol) = ra[j];
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<f32x4*>(Bs + (lrow + RPP * j) * LD + lcol) = rb[j];
    };
    auto mma = [&](const float* As, const float* Bs, int kk) {
        if (ABL & 8) {
            // timing probe for a split-f16 (3 MFMA) path: one K-step of 32 = 2 k16 blocks; each block reads
            // hi and lo fragments (16 B each) of both operands and issues 3 MFMAs per tile pair
            if (kk >= 16) return;      // kk = 0, 8 stand for the two k16 blocks
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) { ah[i] = *reinterpret_cast<const f16x8*>(As + i * 32 * LD + kk); al[i] = *reinterpret_cast<const f16x8*>(As + i * 32 * LD + kk + 16); }
#pragma unroll
            for (int j = 0; j < TN; ++j) { bh[j] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * LD + kk); bl[j] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * LD + kk + 16); }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                }
            return;
        }
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LD + kk);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LD + kk);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    };
    gload(0); lstore(0);
    __syncthreads();
    int cur = 0;
    for (int ks = 0; ks < ksteps; ++ks) {
        const bool more = ks + 1 < ksteps;
        if (more && !(ABL & 1)) gload(ks + 1);
        const float* As = smem + cur * (BM + BN) * LD + (wm * TM * 32 + li) * LD + lk * 4;
        const float* Bs = smem + cur * (BM + BN) * LD + BM * LD + (wn * TN * 32 + li) * LD + lk * 4;
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        if (NBUF == 2 && MID) {
#pragma unroll
            for (int kk = 0; kk < BK / 2; kk += 8) mma(As, Bs, kk);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            if (more) lstore(cur ^ 1);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = BK / 2; kk < BK; kk += 8) mma(As, Bs, kk);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            cur ^= 1;
        } else if (NBUF == 2) {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 8) mma(As, Bs, kk);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            if (more) lstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        } else {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 8) mma(As, Bs, kk);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            if (more && !(ABL & 2)) lstore(0);
            __syncthreads();
        }
    }
    if (EPI == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) { const int col = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const int op = row_out[(wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk]; if (op >= 0) p.out[(size_t)op * p.out_cstride + col] = acc[i][j][r]; } }
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = clock64() - c0; g_clk[1] = wall_clock64() - w0; }
    if (EPI == 2) {
        constexpr int CLD2 = BN + 4;
        float* Cs2 = smem;
        const int c4b = (tid % 32) * 4, colb = n0 + c4b;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (wm == h) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) Cs2[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CLD2 + (wn * TN + j) * 32 + li] = acc[i][j][r];
            }
            __syncthreads();
            for (int r = tid / 32; r < 64; r += 8) {
                const int op = row_out[h * 64 + r];
                if (op < 0) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(Cs2 + r * CLD2 + c4b);
                if (p.residual) { const f32x4 rs = *reinterpret_cast<const f32x4*>(p.residual + (size_t)op * p.res_cstride + colb); v += rs; }
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                *reinterpret_cast<f32x4*>(p.out + (size_t)op * p.out_cstride + colb) = v;
            }
            __syncthreads();
        }
        return;
    }
    // simple epilogue (transposed float4 stores)
    constexpr int CLD = BN + 4;
    float* Cs = smem;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) Cs[((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CLD + (wn * TN + j) * 32 + li] = acc[i][j][r];
    __syncthreads();
    const int c4 = (tid % 32) * 4, col = n0 + c4;
    for (int r = tid / 32; r < BM; r += 8) {
        const int op = row_out[r];
        if (op < 0) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(Cs + r * CLD + c4);
        if (p.residual) { const f32x4 rs = *reinterpret_cast<const f32x4*>(p.residual + (size_t)op * p.res_cstride + col); v += rs; }
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        *reinterpret_cast<f32x4*>(p.out + (size_t)op * p.out_cstride + col) = v;
    }
}


// distance-2 register prefetch: loads for step k+2 are issued while step k computes (two register sets)
__global__ __launch_bounds__(256) void kpf2(const IgemmParams p)
{
    constexpr int BK = 32, WGN = 2, TM = 2, TN = 2, BM = 128, BN = 128, LD = 36, RPP = 32, A_PASSES = 4, B_PASSES = 4;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LD];
    __shared__ int row_base[BM], row_yx[BM], row_out[BM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN, li = lane & 31, lk = lane >> 5;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int nblk = gridDim.x;
    int t;
    { const int b = blockIdx.x; const int q = nblk >> 3, r = nblk & 7; const int xcd = b & 7, idx = b >> 3;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
    const int tile_n = t % tiles_n, tile_m = t / tiles_n, m0 = tile_m * BM, n0 = tile_n * BN;
    const int HgWg = p.Hg * p.Wg;
    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r; int base = -1, yx = 0, op = -1;
        if (m < p.M) { const int n = m / HgWg; const int rem = m - n * HgWg; const int gy = rem / p.Wg; const int gx = rem - gy * p.Wg;
            base = n * p.Hin * p.Win; yx = ((gy * p.in_stride) << 16) | (gx * p.in_stride); op = (n * p.Hout + gy * p.os + p.oy) * p.Wout + gx * p.os + p.ox; }
        row_base[r] = base; row_yx[r] = yx; row_out[r] = op;
    }
    __syncthreads();
    const int lrow = tid / 8, lcol = (tid % 8) * 4;
    int a_base[A_PASSES], a_yx[A_PASSES];
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j) { a_base[j] = row_base[lrow + RPP * j]; a_yx[j] = row_yx[lrow + RPP * j]; }
    const float* wrow = p.w + (size_t)(n0 + lrow) * p.K + lcol;
    const int ksteps = p.K / BK, cpt = (p.seg[0].C) / BK;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra0[A_PASSES], rb0[B_PASSES], ra1[A_PASSES], rb1[B_PASSES];
    auto gload = [&](int ks, f32x4* ra, f32x4* rb) {
        const int tap = ks / cpt; const int chunk = ks - tap * cpt;
        const IgemmSeg sg = p.seg[0];
        const int c = chunk * BK + lcol;
        const int dy = p.dy[tap], dx = p.dx[tap];
#pragma unroll
        for (int j = 0; j < A_PASSES; ++j) {
            const int iy = (a_yx[j] >> 16) + dy, ix = (a_yx[j] & 0xffff) + dx;
            const bool ok = a_base[j] >= 0 && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(sg.ptr + (size_t)(a_base[j] + iy * p.Win + ix) * sg.cstride + c);
            ra[j] = v;
        }
        const float* wp = wrow + (size_t)ks * BK;
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) rb[j] = *reinterpret_cast<const f32x4*>(wp + (size_t)(RPP * j) * p.K);
    };
    auto lstore = [&](const f32x4* ra, const f32x4* rb) {
        float* As = smem; float* Bs = As + BM * LD;
#pragma unroll
        for (int j = 0; j < A_PASSES; ++j) *reinterpret_cast<f32x4*>(As + (lrow + RPP * j) * LD + lcol) = ra[j];
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<f32x4*>(Bs + (lrow + RPP * j) * LD + lcol) = rb[j];
    };
    const float* As = smem + (wm * TM * 32 + li) * LD + lk * 4;
    const float* Bs = smem + BM * LD + (wn * TN * 32 + li) * LD + lk * 4;
    auto compute = [&]() {
#pragma unroll
        for (int kk = 0; kk < BK; kk += 8) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LD + kk);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LD + kk);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
    };
    gload(0, ra0, rb0); lstore(ra0, rb0);
    if (ksteps > 1) gload(1, ra0, rb0);
    __syncthreads();
    for (int ks = 0; ks < ksteps; ks += 2) {       // ksteps even in the probed shapes
        if (ks + 2 < ksteps) gload(ks + 2, ra1, rb1);
        compute();
        __syncthreads();
        if (ks + 1 < ksteps) { lstore(ra0, rb0); }
        __syncthreads();
        if (ks + 3 < ksteps) gload(ks + 3, ra0, rb0);
        if (ks + 1 < ksteps) compute();
        __syncthreads();
        if (ks + 2 < ksteps) { lstore(ra1, rb1); }
        __syncthreads();
    }
    float* o = p.out;
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int col = n0 + (wn * TN + j) * 32 + li;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const int op = row_out[(wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk]; if (op >= 0) o[(size_t)op * p.out_cstride + col] = acc[i][j][r]; } }
}

template <int BK, int NBUF, int MID, int PRIO, int EPI = 1, int ABL = 0>
static float run(const IgemmParams& p, int iters)
{
    const int tiles = ((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((kexp<BK, NBUF, MID, PRIO, EPI, ABL>), dim3(tiles), dim3(256), 0, 0, p);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((kexp<BK, NBUF, MID, PRIO, EPI, ABL>), dim3(tiles), dim3(256), 0, 0, p);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    unsigned long long h[4]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof(h));
    printf("   [block0: %llu shader clk / %llu wall ticks -> %.3f GHz if wall=100MHz]\n", h[0], h[1], (double)h[0] / (double)h[1] * 0.1);
    return ms / iters;
}

int main(int argc, char** argv)
{
    // deconv2-like: N x 32 x 32 x 256 -> 256, 5x5
    const int N = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 32, C = argc > 3 ? atoi(argv[3]) : 256, Co = argc > 4 ? atoi(argv[4]) : 256, KS = argc > 5 ? atoi(argv[5]) : 5;
    const size_t nin = (size_t)N * H * H * C, nout = (size_t)N * H * H * Co, nw = (size_t)((Co + 127) / 128 * 128) * KS * KS * C;
    float *x, *w, *y;
    hipMalloc(&x, nin * 4); hipMalloc(&w, nw * 4); hipMalloc(&y, nout * 4);
    std::vector<float> h(nin); for (size_t i = 0; i < nin; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(x, h.data(), nin * 4, hipMemcpyHostToDevice);
    std::vector<float> hw(nw); for (size_t i = 0; i < nw; ++i) hw[i] = (float)((i * 40503u) >> 4 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
    IgemmParams p; memset(&p, 0, sizeof(p));
    p.seg[0] = {x, C, C, 0}; p.seg0_chunks = C / 32; p.chunks_per_tap = C / 32;
    p.N = N; p.Hin = p.Win = H; p.Hg = p.Wg = H; p.M = N * H * H; p.in_stride = 1; p.ntaps = KS * KS;
    for (int a = 0; a < KS; ++a) for (int b = 0; b < KS; ++b) { p.dy[a * KS + b] = a - KS / 2; p.dx[a * KS + b] = b - KS / 2; }
    p.w = w; p.K = KS * KS * C; p.Cout = Co; p.ksteps = p.K / 32; p.ksplit = 1;
    p.out = y; p.Hout = p.Wout = H; p.os = 1; p.out_cstride = Co;
    float* res = nullptr;
    if (argc > 6 && atoi(argv[6])) { hipMalloc(&res, nout * 4); hipMemset(res, 0, nout * 4); p.residual = res; p.res_cstride = Co; }
    const double gf = 2.0 * p.M * Co * p.K / 1e9;
    const int it = 5;
    float ms;
    {
        const int tiles = ((p.M + 127) / 128) * ((p.Cout + 127) / 128);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(kpf2, dim3(tiles), dim3(256), 0, 0, p); hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kpf2, dim3(tiles), dim3(256), 0, 0, p);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); ms /= it;
        printf("prefetch distance 2    %8.3f ms %7.1f TF\n", ms, gf / ms);
    }
    ms = run<32, 2, 0, 0>(p, it); printf("BK32 NBUF2            %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<32, 1, 0, 0, 2>(p, it); printf("BK32 NBUF1 EPI2 (38KB) %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<32, 1, 0, 0, 2, 8>(p, it); printf("  probe: f16x3 MFMA block, same loader %8.3f ms %7.1f TF-equivalent\n", ms, gf / ms);
    ms = run<32, 1, 0, 0, 2, 24>(p, it); printf("  probe: f16x3 + in-loader f32->f16 split of A %8.3f ms %7.1f TF-equivalent\n", ms, gf / ms);
    ms = run<32, 2, 0, 0, 2, 24>(p, it); printf("  probe: f16x3 split, NBUF2 (1 barrier/step)    %8.3f ms %7.1f TF-equivalent\n", ms, gf / ms);
    ms = run<32, 2, 1, 0, 2, 24>(p, it); printf("  probe: f16x3 split, NBUF2 MID                 %8.3f ms %7.1f TF-equivalent\n", ms, gf / ms);
    ms = run<32, 2, 0, 1, 2, 24>(p, it); printf("  probe: f16x3 split, NBUF2 PRIO                %8.3f ms %7.1f TF-equivalent\n", ms, gf / ms);
    ms = run<32, 1, 0, 1, 2, 24>(p, it); printf("  probe: f16x3 split, NBUF1 PRIO                %8.3f ms %7.1f TF-equivalent\n", ms, gf / ms);
    ms = run<32, 1, 0, 0, 2, 9>(p, it); printf("  probe: f16x3, no gload             %8.3f ms %7.1f TF-equivalent\n", ms, gf / ms);
    ms = run<32, 1, 0, 0, 2, 12>(p, it); printf("  probe: f16x3, hot-line loads       %8.3f ms %7.1f TF-equivalent\n", ms, gf / ms);
    ms = run<32, 1, 0, 0, 2, 4>(p, it); printf("  ablate: hot-line loads %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<32, 1, 0, 0, 2, 1>(p, it); printf("  ablate: no gload      %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<32, 1, 0, 0, 2, 2>(p, it); printf("  ablate: no lstore     %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<32, 1, 0, 0, 2, 3>(p, it); printf("  ablate: neither       %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<32, 2, 0, 0, 2>(p, it); printf("BK32 NBUF2 EPI2 (74KB) %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<16, 2, 0, 0, 2>(p, it); printf("BK16 NBUF2 EPI2 (41KB) %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<16, 1, 0, 0, 2>(p, it); printf("BK16 NBUF1 EPI2 (34KB) %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<32, 2, 1, 0>(p, it); printf("BK32 NBUF2 MID        %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<32, 2, 0, 1>(p, it); printf("BK32 NBUF2 PRIO       %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<32, 2, 1, 1>(p, it); printf("BK32 NBUF2 MID PRIO   %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<32, 1, 0, 0>(p, it); printf("BK32 NBUF1            %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<16, 2, 0, 0>(p, it); printf("BK16 NBUF2            %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<16, 2, 1, 0>(p, it); printf("BK16 NBUF2 MID        %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<16, 1, 0, 0>(p, it); printf("BK16 NBUF1            %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<32, 1, 0, 0, 0>(p, it); printf("BK32 NBUF1 EPI0 (37KB) %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<16, 2, 0, 0, 0>(p, it); printf("BK16 NBUF2 EPI0 (41KB) %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<16, 1, 0, 0, 0>(p, it); printf("BK16 NBUF1 EPI0 (20KB) %8.3f ms %7.1f TF\n", ms, gf / ms);
    ms = run<32, 2, 0, 0, 0>(p, it); printf("BK32 NBUF2 EPI0 (74KB) %8.3f ms %7.1f TF\n", ms, gf / ms);
    return 0;
}
