// Halo-tiled implicit-GEMM convolution for stride-1 multi-tap layers (gfx950, PREC_F16X3).
//
// igemm.hip gathers the A operand once per (tap, channel slice): for a 5x5 convolution every input
// pixel travels global -> registers -> f16 split -> LDS 25 times, and with the split-f16 arithmetic
// (three 8-pass MFMAs per 16-deep block) that loader traffic — L1 at 64 B/clk, LDS writes at ~75 B/clk —
// is as long as the matrix work.  Here the M tile is a spatial patch (8 rows x 16
// columns of one sample), and the (8+dy) x (16+dx) halo of a 32-channel slice is staged in LDS ONCE, in
// the same [hi f16 x32 | lo f16 x32] row image the generic kernel uses; every tap of the slice then reads
// its A fragments from that image at a constant byte shift.  Per K-step only the weight panel moves
// (global -> LDS as before).  K order is (slice, tap) instead of (tap, slice).
//
// Serves Conv2D 3x3 / 5x5 stride 1 'SAME', the transposed-conv phases (2x2 .. 3x3 taps, os = 2) and the
// two-segment skip concatenations (reference ae_model.py:193-231, resnet50_mod.py:40-118) whenever the
// grid is a multiple of the 8x16 patch; everything else stays on igemm.hip.
#include "kernels.h"
#include <algorithm>

namespace p2p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int REC = 144;                   // bytes per staged row / pixel: [hi x32 | lo x32 | 16 pad] (36 dwords: conflict-free b128)
constexpr int TY = 8, TX = 16;             // spatial patch = 128 GEMM rows
constexpr int MAX_HALO = (TY + 4) * (TX + 4);
constexpr int HALO_PASSES = (MAX_HALO * 8 + 255) / 256;    // float4 loads per thread per slice (8)

template <int TN>
__global__ __launch_bounds__(256, 3) void igemm_halo_kernel(const IgemmParams p)
{
    constexpr int BM = TY * TX;
    constexpr int BN = 2 * TN * 32;
    constexpr int B_PASSES = BN / 32;
    constexpr int HALO_BYTES = MAX_HALO * REC;
    constexpr int STAGE_BYTES = HALO_BYTES + BN * REC;
    constexpr int CTILE_BYTES = 64 * (BN + 4) * 4;
    __shared__ __attribute__((aligned(16))) char smem[STAGE_BYTES > CTILE_BYTES ? STAGE_BYTES : CTILE_BYTES];
    __shared__ int s_tap[IGEMM_MAX_TAPS + 3];     // byte shift of a tap inside the halo image

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lk = lane >> 5;

    // halo extents from the tap table (wave-uniform)
    int dy0 = 0, dy1 = 0, dx0 = 0, dx1 = 0;
    for (int t = 0; t < p.ntaps; ++t) {
        dy0 = min(dy0, (int)p.dy[t]); dy1 = max(dy1, (int)p.dy[t]);
        dx0 = min(dx0, (int)p.dx[t]); dx1 = max(dx1, (int)p.dx[t]);
    }
    const int HPX = TX + dx1 - dx0, HPY = TY + dy1 - dy0;
    const int halo_px = HPX * HPY;

    // XCD-aware tile order (block b runs on XCD b % 8): contiguous runs of tiles per XCD, n-tile fastest
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int tiles_x = p.Wg / TX, tiles_y = p.Hg / TY;
    int t;
    {
        const int nblk = gridDim.x, b = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = t % tiles_n;
    int tm = t / tiles_n;
    const int tx0 = (tm % tiles_x) * TX; tm /= tiles_x;
    const int ty0 = (tm % tiles_y) * TY;
    const int n = tm / tiles_y;
    const int n0 = tile_n * BN;

    const float* gw = p.w;
    const float* gscale = p.scale;
    const float* gshift = p.shift;
    if (p.n_groups > 1) {                       // groups are runs of samples
        const int row = n * p.Hg * p.Wg;
        int g = 0;
        while (g + 1 < p.n_groups && p.grp[g + 1].row0 <= row) ++g;
        gw = p.grp[g].w; gscale = p.grp[g].scale; gshift = p.grp[g].shift;
    }

    if (tid < p.ntaps) s_tap[tid] = (((int)p.dy[tid] - dy0) * HPX + ((int)p.dx[tid] - dx0)) * REC;

    // ---- halo loader: float4 idx = tid + 256 j -> halo pixel idx / 8, quad idx % 8
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned h_pix[HALO_PASSES];               // pixel index into the input tensor, OOB outside the image / the halo
    int h_dst[HALO_PASSES];
#pragma unroll
    for (int j = 0; j < HALO_PASSES; ++j) {
        const int idx = tid + 256 * j;
        const int hp = idx >> 3, q = idx & 7;
        const int hy = hp / HPX, hx = hp - hy * HPX;
        const int iy = ty0 + dy0 + hy, ix = tx0 + dx0 + hx;
        const bool ok = hp < halo_px && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
        h_pix[j] = ok ? (unsigned)((n * p.Hin + iy) * p.Win + ix) : OOB;
        h_dst[j] = hp < halo_px ? hp * REC + q * 8 : -1;
    }
    const int hq4 = (tid & 7) * 4;             // channel offset of this thread's quad inside the slice
    const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.seg[0].ptr, 0, p.seg_bytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.seg[1].ptr ? p.seg[1].ptr : p.seg[0].ptr), 0, p.seg_bytes[1], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)gw, 0, p.w_bytes, 0x00020000);

    f32x4 rh[HALO_PASSES];
    auto hload = [&](int chunk) {
        const bool s1 = chunk >= p.seg0_chunks;                      // wave-uniform
        const unsigned cs = (unsigned)(s1 ? p.seg[1].cstride : p.seg[0].cstride);
        const unsigned co = (unsigned)((s1 ? p.seg[1].coff + (chunk - p.seg0_chunks) * IGEMM_BK : p.seg[0].coff + chunk * IGEMM_BK) + hq4);
#pragma unroll
        for (int j = 0; j < HALO_PASSES; ++j) {
            const unsigned off = h_pix[j] != OOB ? (h_pix[j] * cs + co) * 4u : OOB;
            rh[j] = __builtin_bit_cast(f32x4, s1 ? __builtin_amdgcn_raw_buffer_load_b128(rs_a1, off, 0, 0)
                                                 : __builtin_amdgcn_raw_buffer_load_b128(rs_a0, off, 0, 0));
        }
    };
    auto hstore = [&]() {
#pragma unroll
        for (int j = 0; j < HALO_PASSES; ++j) {
            if (h_dst[j] < 0) continue;
            const f32x4 v = rh[j];
            const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
            fp16x2 l01, l23;          // residuals are exact in fp32; round them to nearest
            l01[0] = (__fp16)(v[0] - (float)h01[0]); l01[1] = (__fp16)(v[1] - (float)h01[1]);
            l23[0] = (__fp16)(v[2] - (float)h23[0]); l23[1] = (__fp16)(v[3] - (float)h23[1]);
            *reinterpret_cast<uint2*>(smem + h_dst[j]) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
            *reinterpret_cast<uint2*>(smem + h_dst[j] + 64) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
        }
    };

    // ---- weight loader: rows (tid >> 3) + 32 j of the n-tile, 16-byte segment (tid & 7)
    const int lrow = tid >> 3;
    const int lcol = (tid & 7) * 4;
    unsigned b_off[B_PASSES];
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) b_off[j] = ((unsigned)(n0 + lrow + 32 * j) * (unsigned)p.K + (unsigned)lcol) * 4u;
    f32x4 rb[B_PASSES];
    char* Bst = smem + HALO_BYTES;
    auto bload = [&](int tap, int chunk) {
        const int koff = (tap * p.chunks_per_tap + chunk) * (IGEMM_BK * 4);     // the panel's K order is (tap, slice)
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off[j], koff, 0));
    };
    auto bstore = [&]() {
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<f32x4*>(Bst + (lrow + 32 * j) * REC + lcol * 4) = rb[j];
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A fragment of m-tile i: rows 32 i .. of this wave's 64 = patch rows wm*4 + 2i + (li >> 4), column li & 15
    const char* As = smem + ((wm * 4 + (li >> 4)) * HPX + (li & 15)) * REC + lk * 16;
    const int a_tile = 2 * HPX * REC;
    const char* Bs = Bst + (wn * TN * 32 + li) * REC + lk * 16;

    const int n_chunks = p.chunks_per_tap;
    hload(0);
    bload(0, 0);
    hstore();
    bstore();
    __syncthreads();
    if (n_chunks > 1) hload(1);

    int tap = 0, chunk = 0;
    const int total = n_chunks * p.ntaps;
    for (int ks = 0; ks < total; ++ks) {
        const int shift = __builtin_amdgcn_readfirstlane(s_tap[tap]);
        int ntap = tap + 1, nchunk = chunk;
        if (ntap == p.ntaps) { ntap = 0; ++nchunk; }
        const bool more = ks + 1 < total;
        if (more) bload(ntap, nchunk);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f16x8 ah[2], al[2], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(As + shift + i * a_tile + kb * 32);
                al[i] = *reinterpret_cast<const f16x8*>(As + shift + i * a_tile + kb * 32 + 64);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * REC + kb * 32);
                bl[j] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * REC + kb * 32 + 64);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();                  // everyone is done reading the weight tile (and, at a slice end, the halo)
        if (more) {
            bstore();
            if (nchunk != chunk) hstore();                       // next slice's halo (prefetched at the start of this one)
            __syncthreads();
            if (nchunk != chunk && nchunk + 1 < n_chunks) hload(nchunk + 1);
        }
        tap = ntap; chunk = nchunk;
    }

    // ---- epilogue (as igemm.hip): accumulators transposed through LDS, 64 GEMM rows per pass, so that a
    //      thread owns 4 consecutive channels of one pixel.  C/D layout of the 32x32 MFMA:
    //      col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    constexpr int CLD = BN + 4;
    constexpr int TPR = BN / 4;
    constexpr int RPP = 256 / TPR;
    float* Cs = reinterpret_cast<float*>(smem);
    const int c4 = (tid % TPR) * 4;
    const int col = n0 + c4;
    const int r0 = tid / TPR;
    const bool cok = col < p.Cout;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (cok) {
        if (gscale) sc = *reinterpret_cast<const f32x4*>(gscale + col);
        if (gshift) sh = *reinterpret_cast<const f32x4*>(gshift + col);
    }
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        if (wm == h) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Cs[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CLD + (wn * TN + j) * 32 + li] = acc[i][j][r];
        }
        __syncthreads();
        if (cok) {
            constexpr int NIT = 64 / RPP;
            int ops[NIT];
            f32x4 rs[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = h * 64 + r0 + it * RPP;
                const int gy = ty0 + (row >> 4), gx = tx0 + (row & 15);
                ops[it] = (n * p.Hout + gy * p.os + p.oy) * p.Wout + gx * p.os + p.ox;
                rs[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (p.residual) {          // all residual loads before the first store
#pragma unroll
                for (int it = 0; it < NIT; ++it) rs[it] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)ops[it] * p.res_cstride + col);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                f32x4 v = *reinterpret_cast<const f32x4*>(Cs + (r0 + it * RPP) * CLD + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]) + rs[it][e];
                if (p.act == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (p.act == ACT_LEAKY) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.alpha;
                }
                *reinterpret_cast<f32x4*>(p.out + (size_t)ops[it] * p.out_cstride + p.out_coff + col) = v;
            }
        }
        if (h == 0) __syncthreads();
    }
}

}  // namespace

bool igemm_halo_supported(const IgemmParams& p)
{
    if (p.prec != PREC_F16X3 || p.mode != EPI_NORMAL || p.ksplit > 1 || p.in_stride != 1 || p.ntaps < 4) return false;
    if (p.Hin != p.Hg || p.Win != p.Wg || p.Hg % TY || p.Wg % TX || p.Cout % 64) return false;
    int dy0 = 0, dy1 = 0, dx0 = 0, dx1 = 0;
    for (int t = 0; t < p.ntaps; ++t) {
        dy0 = std::min(dy0, (int)p.dy[t]); dy1 = std::max(dy1, (int)p.dy[t]);
        dx0 = std::min(dx0, (int)p.dx[t]); dx1 = std::max(dx1, (int)p.dx[t]);
    }
    return dy1 - dy0 <= 4 && dx1 - dx0 <= 4;
}

hipError_t launch_igemm_halo(const IgemmParams& p, hipStream_t s)
{
    const int m_tiles = p.N * (p.Hg / TY) * (p.Wg / TX);
    if (p.Cout % 128 == 0) {
        hipLaunchKernelGGL((igemm_halo_kernel<2>), dim3(m_tiles * (p.Cout / 128)), dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL((igemm_halo_kernel<1>), dim3(m_tiles * (p.Cout / 64)), dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace p2p
