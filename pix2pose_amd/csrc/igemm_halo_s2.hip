// Halo-tiled implicit GEMM for Conv2D 5x5 stride 2 'SAME' on output grids of 16x16 and more (gfx950, PREC_F16X3): the encoder of
// the "paper" backbone -- conv2_1 || conv2_2 (64x64x128 -> 32x32x256) and conv3_1 || conv3_2 (32x32x256 -> 16x16x256), reference
// pix2pose_model/ae_model.py:79-98.  These ran on the generic kernel (igemm.hip), which gathers, splits and stages the A operand
// once per (tap, slice) -- 25 times per input pixel: 300 algorithmic TFLOP/s against 400 - 440 for the halo-tiled 5x5 layers, and
// a fifth of a step of that backbone.
//
// Same two ideas as its neighbours: the 8x16 output patch of igemm_halo.hip (128 GEMM rows x 128 output channels, 4 waves as
// 2 x 2 of 64 x 64, halo records [hi x32 | lo x32 | pad] with a row pitch that keeps two-row fragments conflict-free, swizzled
// weight rows), and the PARITY PLANES of igemm_halo8.hip: output (y, x), tap (ky, kx) reads input (2y + ky - 1, 2x + kx - 1);
// with ky - 1 = 2a + p that is plane (p, q) of the input, x_pq[i][j] = x[2i + p][2j + q], at (y + a, x + b) -- four stride-1
// sub-convolutions (2x2, 2x3, 3x2, 3x3 taps) accumulating into ONE output.  The K loop is (plane, slice, tap of the plane); per
// (plane, slice) the (8 + 2) x (16 + 2) halo of the plane is staged once, addressed in the NHWC tensor directly; the weight panel
// keeps its (tap, cin) order.
#include "kernels.h"
#include <algorithm>

namespace p2p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int REC = 144;
constexpr int WREC = 128;
constexpr int TY = 8, TX = 16;
constexpr int HPY = TY + 2, HPX = TX + 2;                  // taps within +-1 in plane coordinates
constexpr int PITCH = HPX * REC + 14 * 16;                 // 2816: a multiple of 16 slots (igemm_halo.hip)
constexpr int HALO_BYTES = HPY * PITCH;                    // 28160
constexpr int HALO_PASSES = (HPY * HPX * 8 + 255) / 256;   // 6
constexpr int BN = 128, TN = 2, B_PASSES = BN / 32;
constexpr int STAGE_BYTES = HALO_BYTES + BN * WREC;        // 44544
constexpr int CTILE_BYTES = 64 * (BN + 4) * 4;             // 33792

__global__ __launch_bounds__(256, 3) void igemm_halo_s2_kernel(const IgemmParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[STAGE_BYTES > CTILE_BYTES ? STAGE_BYTES : CTILE_BYTES];
    __shared__ int s_tap[4][9];        // per plane: panel tap index
    __shared__ int s_shift[4][9];      // per plane: byte shift of the tap inside the halo image
    __shared__ int s_ntaps[4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lk = lane >> 5;

    if (tid < 4 * 9) {
        // plane (py, px): taps ky in {1, 3} (py = 0) or {0, 2, 4} (py = 1), likewise kx; a = (ky - 1 - py) / 2 in {-1, 0, 1}
        const int g = tid / 9, k = tid - g * 9;
        const int py = g >> 1, px = g & 1, ny = py ? 3 : 2, nx = px ? 3 : 2;
        int tap = 0, shift = 0;
        if (k < ny * nx) {
            const int iy = k / nx, ix = k - iy * nx;
            const int ky = py ? 2 * iy : 2 * iy + 1, kx = px ? 2 * ix : 2 * ix + 1;
            const int a = (ky - 1 - py) / 2, b = (kx - 1 - px) / 2;          // exact: numerators are even (and -2 / 2 = -1)
            tap = ky * 5 + kx;
            shift = (a + 1) * PITCH + (b + 1) * REC;
        }
        if (k == 0) s_ntaps[g] = ny * nx;
        s_tap[g][k] = tap;
        s_shift[g][k] = shift;
    }

    // XCD-aware tile order (block b runs on XCD b % 8): contiguous runs of tiles per XCD, n-tile fastest
    const int tiles_n = p.Cout / BN;
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

    // ---- halo loader: float4 idx = tid + 256 j -> quad idx % 8 of halo pixel perm(idx / 8) (igemm_halo.hip); the pixel index is
    //      that of plane (0, 0), the other planes add (py * Win + px); validity per plane in one bit each
    constexpr unsigned OOB = 0xFFFFFFF0u;
    int h_pix[HALO_PASSES];
    unsigned h_ok = 0;                          // bit 4 j + plane
    unsigned h_dst2[(HALO_PASSES + 1) / 2];
#pragma unroll
    for (int j = 0; j < (HALO_PASSES + 1) / 2; ++j) h_dst2[j] = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < HALO_PASSES; ++j) {
        const int idx = tid + 256 * j;
        const int t8 = idx >> 3, q = idx & 7;
        const int hp = (t8 & ~7) | ((t8 & 1) << 2) | ((t8 >> 1) & 3);
        const int hy = hp / HPX, hx = hp - hy * HPX;
        const bool in_halo = hp < HPX * HPY;
        const int iy = 2 * (ty0 - 1 + hy), ix = 2 * (tx0 - 1 + hx);
        unsigned ok = 0;
        if (in_halo)
            for (int g = 0; g < 4; ++g) {
                const int y = iy + (g >> 1), x = ix + (g & 1);
                if ((unsigned)y < (unsigned)p.Hin && (unsigned)x < (unsigned)p.Win) ok |= 1u << g;
            }
        h_pix[j] = (n * p.Hin + iy) * p.Win + ix;
        h_ok |= ok << (4 * j);
        const unsigned dst = in_halo ? (unsigned)(hy * PITCH + hx * REC + q * 8) : 0xFFFFu;
        h_dst2[j >> 1] = (j & 1) ? ((h_dst2[j >> 1] & 0x0000FFFFu) | (dst << 16)) : ((h_dst2[j >> 1] & 0xFFFF0000u) | dst);
    }
    const int hq4 = (tid & 7) * 4;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.seg[0].ptr, 0, p.seg_bytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)gw, 0, p.w_bytes, 0x00020000);

    f32x4 rh[HALO_PASSES];
    auto hload = [&](int plane, int chunk) {
        const unsigned cs = (unsigned)p.seg[0].cstride;
        const unsigned co = (unsigned)(p.seg[0].coff + chunk * IGEMM_BK + hq4);
        const int poff = (plane >> 1) * p.Win + (plane & 1);
#pragma unroll
        for (int j = 0; j < HALO_PASSES; ++j) {
            const unsigned off = (h_ok >> (4 * j + plane)) & 1u ? ((unsigned)(h_pix[j] + poff) * cs + co) * 4u : OOB;
            rh[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, off, 0, 0));
        }
    };
    auto hstore = [&]() {
#pragma unroll
        for (int j = 0; j < HALO_PASSES; ++j) {
            const unsigned dst = (j & 1) ? (h_dst2[j >> 1] >> 16) : (h_dst2[j >> 1] & 0xFFFFu);
            if (dst == 0xFFFFu) continue;
            const f32x4 v = rh[j];
            const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
            fp16x2 l01, l23;          // residuals are exact in fp32; round them to nearest
            l01[0] = (__fp16)(v[0] - (float)h01[0]); l01[1] = (__fp16)(v[1] - (float)h01[1]);
            l23[0] = (__fp16)(v[2] - (float)h23[0]); l23[1] = (__fp16)(v[3] - (float)h23[1]);
            *reinterpret_cast<uint2*>(smem + dst) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
            *reinterpret_cast<uint2*>(smem + dst + 64) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
        }
    };

    // ---- weight loader: rows (tid >> 3) + 32 j of the n-tile, 16-byte segment (tid & 7); swizzled 128-byte rows
    const int lrow = tid >> 3;
    const int lcol = (tid & 7) * 4;
    unsigned b_off[B_PASSES];
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) b_off[j] = ((unsigned)(n0 + lrow + 32 * j) * (unsigned)p.K + (unsigned)lcol) * 4u;
    f32x4 rb[B_PASSES];
    char* Bst = smem + HALO_BYTES;
    auto bload = [&](int ptap, int chunk) {
        const int koff = (ptap * p.chunks_per_tap + chunk) * (IGEMM_BK * 4);     // the panel's K order is (tap, slice)
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off[j], koff, 0));
    };
    const int b_dst = lrow * WREC + (((tid & 7) ^ ((lrow >> 1) & 7)) << 4);
    auto bstore = [&]() {
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<f32x4*>(Bst + b_dst + 32 * j * WREC) = rb[j];
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A fragment of m-tile i: rows 32 i .. of this wave's 64 = patch rows wm*4 + 2i + (li >> 4), column li & 15
    const char* As = smem + (wm * 4 + (li >> 4)) * PITCH + (li & 15) * REC + lk * 16;
    constexpr int a_tile = 2 * PITCH;
    const char* Bs = Bst + (wn * TN * 32 + li) * WREC;
    int b_sw[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) b_sw[kb][hf] = ((kb * 2 + lk + 4 * hf) ^ ((li >> 1) & 7)) << 4;

    __syncthreads();                               // tap tables
    const int n_chunks = p.chunks_per_tap;
    const int n_slices = 4 * n_chunks;             // (plane, chunk) pairs, plane-major
    hload(0, 0);
    bload(s_tap[0][0], 0);
    hstore();
    bstore();
    __syncthreads();
    if (n_slices > 1) hload(n_chunks > 1 ? 0 : 1, n_chunks > 1 ? 1 : 0);

    int plane = 0, chunk = 0, tap = 0;
    for (;;) {
        const int shift = __builtin_amdgcn_readfirstlane(s_shift[plane][tap]);
        // next K-step: next tap of this slice, else first tap of the next (plane, chunk)
        int ntap = tap + 1, nchunk = chunk, nplane = plane;
        if (ntap == s_ntaps[plane]) { ntap = 0; if (++nchunk == n_chunks) { nchunk = 0; ++nplane; } }
        const bool more = nplane < 4;
        const bool new_slice = nchunk != chunk || nplane != plane;
        if (more) bload(__builtin_amdgcn_readfirstlane(s_tap[nplane][ntap]), nchunk);
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
                bh[j] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * WREC + b_sw[kb][0]);
                bl[j] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * WREC + b_sw[kb][1]);
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
        if (!more) break;
        bstore();
        if (new_slice) hstore();                             // next slice's halo (prefetched at the start of this one)
        __syncthreads();
        if (new_slice) {                                     // prefetch the slice after the next
            int c2 = nchunk + 1, p2 = nplane;
            if (c2 == n_chunks) { c2 = 0; ++p2; }
            if (p2 < 4) hload(p2, c2);
        }
        tap = ntap; chunk = nchunk; plane = nplane;
    }

    // ---- epilogue (as igemm_halo.hip): accumulators transposed through LDS, 64 GEMM rows per pass
    constexpr int CLD = BN + 4;
    constexpr int TPR = BN / 4;
    constexpr int RPP = 256 / TPR;
    float* Cs = reinterpret_cast<float*>(smem);
    const int c4 = (tid % TPR) * 4;
    const int col = n0 + c4;
    const int r0 = tid / TPR;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (gscale) sc = *reinterpret_cast<const f32x4*>(gscale + col);
    if (gshift) sh = *reinterpret_cast<const f32x4*>(gshift + col);
    float amax = 0.f;      // operand-range guard (kernels.h)
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
        constexpr int NIT = 64 / RPP;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = h * 64 + r0 + it * RPP;
            const int gy = ty0 + (row >> 4), gx = tx0 + (row & 15);
            const size_t op = ((size_t)n * p.Hout + gy) * p.Wout + gx;
            f32x4 v = *reinterpret_cast<const f32x4*>(Cs + (r0 + it * RPP) * CLD + c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
            if (p.act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = relu_nan(v[e]);
            } else if (p.act == ACT_LEAKY) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.alpha;
            }
            amax = range_note4(amax, v);
            *reinterpret_cast<f32x4*>(p.out + op * p.out_cstride + p.out_coff + col) = v;
        }
        if (h == 0) __syncthreads();
    }    range_commit(p.range_acc, amax);
}

}  // namespace

bool igemm_halo_s2_supported(const IgemmParams& p)
{
    static const bool on = dev_env("P2P_NO_HALO_S2") == nullptr;
    if (!on || p.prec != PREC_F16X3 || p.mode != EPI_NORMAL || p.ksplit > 1 || p.in_stride != 2 || p.ntaps != 25) return false;
    if (p.Hin != 2 * p.Hg || p.Win != 2 * p.Wg || p.Hg % TY || p.Wg % TX || p.Cout % BN) return false;
    if (p.seg[1].C != 0 || p.residual || p.seg1_stride || p.os != 1 || p.oy || p.ox || p.Hout != p.Hg || p.Wout != p.Wg) return false;
    for (int t = 0; t < 25; ++t)
        if (p.dy[t] != t / 5 - 1 || p.dx[t] != t % 5 - 1) return false;       // 5x5, TF 'SAME' at stride 2: one row / column before
    return true;
}

hipError_t launch_igemm_halo_s2(const IgemmParams& p, hipStream_t s)
{
    const int tiles = p.N * (p.Hg / TY) * (p.Wg / TX) * (p.Cout / BN);
    hipLaunchKernelGGL(igemm_halo_s2_kernel, dim3(tiles), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace p2p
