// Halo-tiled implicit GEMM for the layers that live on an 8x8 output grid (gfx950, PREC_F16X3):
//   * conv4_1 || conv4_2  -- Conv2D 5x5 stride 2 'SAME', 16x16x512 -> 8x8x512 (reference ae_model.py:190-195), and
//   * the four sub-pixel phases of the first transposed convolution, 8x8x256 -> 16x16x256 (ae_model.py:202-205).
// Both ran on the generic kernel (igemm.hip), which gathers, splits and stages the A operand once per (tap, slice): 25
// times per input pixel for conv4 (328-344 algorithmic TFLOP/s against 430-444 for the halo-tiled 5x5 layers).
//
// M tile = the 8x8 grids of TWO samples (128 GEMM rows; a wave pair owns one sample), N tile = 128 output channels.
// Per 32-channel slice the (8+2) x (8+2) halo of both samples is staged in LDS once, as the same [hi f16 x32 | lo f16 x32]
// 144-byte records igemm_halo.hip uses, and every tap reads its A fragments from it at a constant byte shift; only the
// weight tile moves per K-step.
//
// Stride 2 goes through PARITY PLANES: output (y, x), tap (ky, kx) reads input (2y + ky - 1, 2x + kx - 1); with
// ky - 1 = 2a + p that is plane (p, q) of the input, x_pq[i][j] = x[2i + p][2j + q], at (y + a, x + b) -- four stride-1
// sub-convolutions (2x2, 2x3, 3x2, 3x3 taps: the transposed-conv phase decomposition run backwards) accumulating into one
// output.  The K loop is (plane, slice, tap of the plane); the loader addresses the plane's pixels in the NHWC tensor
// directly (no space-to-depth copy), the weight panel keeps its (tap, cin) order.
//
// LDS banking: an MFMA fragment of 32 rows is 4 grid rows x 8 columns; ds_read_b128 serves 16-lane groups that hold
// 4 columns of each of 4 rows ({0-3,12-15,20-27}: rows 0,1,2,3 / columns 0-3,4-7,4-7,0-3), which is conflict-free
// exactly when the row pitch is 8 slots (128 B) modulo 256 B next to the 9-slot record stride; rows are padded to that.
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
constexpr int G8 = 8;                                      // grid side
constexpr int MAX_HP = G8 + 2;                             // halo side (taps within +-1 in plane coordinates)
constexpr int MAX_PITCH = MAX_HP * REC + 14 * 16;          // 1664 B: 10 records + 14 slots = 104 slots = 8 (mod 16)
constexpr int SAMPLE_BYTES = MAX_HP * MAX_PITCH;           // 16640
constexpr int HALO_BYTES = 2 * SAMPLE_BYTES;
constexpr int BN = 128;
constexpr int HALO_PASSES = (2 * MAX_HP * MAX_HP * 8 + 255) / 256;     // 7

template <int STRIDE2>
__global__ __launch_bounds__(256, 3) void igemm_halo8_kernel(const IgemmParams p)
{
    constexpr int TN = 2, B_PASSES = BN / 32;
    constexpr int STAGE_BYTES = HALO_BYTES + BN * WREC;
    constexpr int CTILE_BYTES = 64 * (BN + 4) * 4;
    __shared__ __attribute__((aligned(16))) char smem[STAGE_BYTES > CTILE_BYTES ? STAGE_BYTES : CTILE_BYTES];
    __shared__ int s_tap[4][9];        // per plane: panel tap index
    __shared__ int s_shift[4][9];      // per plane: byte shift of the tap inside the halo image
    __shared__ int s_ntaps[4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;        // wm: sample of the pair
    const int li = lane & 31, lk = lane >> 5;

    // halo extents in plane coordinates (wave-uniform)
    int dy0 = -1, dx0 = -1, HPX = MAX_HP, HPY = MAX_HP;
    if (!STRIDE2) {
        int dy1 = 0, dx1 = 0;
        dy0 = dx0 = 0;
        for (int t = 0; t < p.ntaps; ++t) {
            dy0 = min(dy0, (int)p.dy[t]); dy1 = max(dy1, (int)p.dy[t]);
            dx0 = min(dx0, (int)p.dx[t]); dx1 = max(dx1, (int)p.dx[t]);
        }
        HPX = G8 + dx1 - dx0; HPY = G8 + dy1 - dy0;
    }
    const int PITCH = HPX * REC + (((8 - 9 * HPX) % 16 + 16) % 16) * 16;     // = 8 slots (mod 16)
    const int n_planes = STRIDE2 ? 4 : 1;

    if (tid < 4 * 9) {
        const int g = tid / 9, k = tid - g * 9;
        int tap = 0, shift = 0;
        if (STRIDE2) {
            // plane (py, px): taps ky in {1, 3} (py = 0) or {0, 2, 4} (py = 1), likewise kx; a = (ky - 1 - py) / 2
            const int py = g >> 1, px = g & 1, ny = py ? 3 : 2, nx = px ? 3 : 2;
            if (k < ny * nx) {
                const int iy = k / nx, ix = k - iy * nx;
                const int ky = py ? 2 * iy : 2 * iy + 1, kx = px ? 2 * ix : 2 * ix + 1;
                const int a = (ky - 1 - py) / 2, b = (kx - 1 - px) / 2;          // exact: numerators are even (and -2 / 2 = -1)
                tap = ky * 5 + kx;
                shift = (a - dy0) * PITCH + (b - dx0) * REC;
            }
            if (k == 0) s_ntaps[g] = ny * nx;
        } else {
            if (g == 0 && k < p.ntaps) { tap = k; shift = ((int)p.dy[k] - dy0) * PITCH + ((int)p.dx[k] - dx0) * REC; }
            if (k == 0) s_ntaps[g] = g == 0 ? p.ntaps : 0;
        }
        s_tap[g][k] = tap;
        s_shift[g][k] = shift;
    }

    // XCD-aware tile order (block b runs on XCD b % 8): contiguous runs of tiles per XCD, n-tile fastest
    const int tiles_n = p.Cout / BN;
    int t;
    {
        const int nblk = gridDim.x, b = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = t % tiles_n;
    const int tile_m = t / tiles_n;
    const int n0 = tile_n * BN;
    // rows of this tile (grouped launches: every object starts on a tile boundary, as in igemm.hip)
    int m0 = tile_m * 128, m_end = p.M;
    const float* gw = p.w;
    const float* gscale = p.scale;
    const float* gshift = p.shift;
    if (p.n_groups > 1) {
        int g = 0;
        while (g + 1 < p.n_groups && tile_m >= p.grp[g + 1].tile0) ++g;
        m0 = p.grp[g].row0 + (tile_m - p.grp[g].tile0) * 128;
        m_end = p.grp[g + 1].row0;
        gw = p.grp[g].w; gscale = p.grp[g].scale; gshift = p.grp[g].shift;
    }
    const int n_first = m0 >> 6;                    // 64 rows per sample
    const int n_valid = (m_end - m0 + 63) >> 6;     // samples of this tile that exist (1 or 2)

    // ---- halo loader: float4 idx = tid + 256 j -> quad idx % 8 of halo pixel perm(idx / 8) (pairs of octets take pixels 4 records
    //      apart: conflict-free ds_write_b64, see igemm_halo.hip)
    constexpr unsigned OOB = 0xFFFFFFF0u;
    int h_pix[HALO_PASSES];                     // pixel index of the halo pixel in plane (0, 0) / in the stride-1 tensor
    unsigned h_ok = 0;                          // validity: bit 4 j + plane (7 passes x 4 planes in one register)
    unsigned h_dst2[(HALO_PASSES + 1) / 2];
#pragma unroll
    for (int j = 0; j < (HALO_PASSES + 1) / 2; ++j) h_dst2[j] = 0xFFFFFFFFu;
    const int per_sample = HPX * HPY;
#pragma unroll
    for (int j = 0; j < HALO_PASSES; ++j) {
        const int idx = tid + 256 * j;
        const int t8 = idx >> 3, q = idx & 7;
        const int hp = (t8 & ~7) | ((t8 & 1) << 2) | ((t8 >> 1) & 3);
        const int s = hp / per_sample, rem = hp - s * per_sample;
        const int hy = rem / HPX, hx = rem - hy * HPX;
        const bool in_halo = s < 2;
        const int n = n_first + s;
        unsigned ok = 0;
        int pix = 0;
        if (in_halo && s < n_valid) {
            if (STRIDE2) {
                const int iy = 2 * (hy + dy0), ix = 2 * (hx + dx0);
                pix = (n * p.Hin + iy) * p.Win + ix;
                for (int g = 0; g < 4; ++g) {
                    const int y = iy + (g >> 1), x = ix + (g & 1);
                    if ((unsigned)y < (unsigned)p.Hin && (unsigned)x < (unsigned)p.Win) ok |= 1u << g;
                }
            } else {
                const int iy = hy + dy0, ix = hx + dx0;
                pix = (n * p.Hin + iy) * p.Win + ix;
                if ((unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) ok = 1u;
            }
        }
        h_pix[j] = pix;
        h_ok |= ok << (4 * j);
        const unsigned dst = in_halo ? (unsigned)(s * SAMPLE_BYTES + hy * PITCH + hx * REC + q * 8) : 0xFFFFu;
        h_dst2[j >> 1] = (j & 1) ? ((h_dst2[j >> 1] & 0x0000FFFFu) | (dst << 16)) : ((h_dst2[j >> 1] & 0xFFFF0000u) | dst);
    }
    const int hq4 = (tid & 7) * 4;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.seg[0].ptr, 0, p.seg_bytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)gw, 0, p.w_bytes, 0x00020000);

    f32x4 rh[HALO_PASSES];
    auto hload = [&](int plane, int chunk) {
        const unsigned cs = (unsigned)p.seg[0].cstride;
        const unsigned co = (unsigned)(p.seg[0].coff + chunk * IGEMM_BK + hq4);
        const int poff = STRIDE2 ? (plane >> 1) * p.Win + (plane & 1) : 0;
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

    // A fragment of m-tile i of sample wm: grid rows 4 i + (li >> 3), column li & 7
    const char* As = smem + wm * SAMPLE_BYTES + (li >> 3) * PITCH + (li & 7) * REC + lk * 16;
    const int a_tile = 4 * PITCH;
    const char* Bs = Bst + (wn * TN * 32 + li) * WREC;
    int b_sw[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) b_sw[kb][hf] = ((kb * 2 + lk + 4 * hf) ^ ((li >> 1) & 7)) << 4;

    __syncthreads();                               // tap tables
    const int n_chunks = p.chunks_per_tap;
    const int n_slices = n_planes * n_chunks;      // (plane, chunk) pairs, plane-major
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
        const bool more = nplane < n_planes;
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
            if (p2 < n_planes) hload(p2, c2);
        }
        tap = ntap; chunk = nchunk; plane = nplane;
    }

    // ---- epilogue (as igemm_halo.hip): accumulators transposed through LDS, one sample (64 GEMM rows) per pass
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
        if (h < n_valid) {
            const int n = n_first + h;
            constexpr int NIT = 64 / RPP;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = r0 + it * RPP;
                const int gy = row >> 3, gx = row & 7;
                const size_t op = ((size_t)n * p.Hout + gy * p.os + p.oy) * p.Wout + gx * p.os + p.ox;
                f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * CLD + c4);
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
        }
        if (h == 0) __syncthreads();
    }    range_commit(p.range_acc, amax);
}

}  // namespace

// 1: stride-1 layer on an 8x8 grid; 2: the stride-2 5x5 'SAME' convolution from a 16x16 input; 0: not for this kernel
int igemm_halo8_mode(const IgemmParams& p)
{
    static const bool on = dev_env("P2P_NO_HALO8") == nullptr;
    if (!on || p.prec != PREC_F16X3 || p.mode != EPI_NORMAL || p.ksplit > 1 || p.Hg != G8 || p.Wg != G8 || p.Cout % BN) return 0;
    if (p.seg[1].C != 0 || p.residual || p.seg1_stride) return 0;
    if (p.in_stride == 1 && p.Hin == G8 && p.Win == G8 && p.ntaps >= 4 && p.ntaps <= 9) {
        for (int t = 0; t < p.ntaps; ++t)
            if (p.dy[t] < -1 || p.dy[t] > 1 || p.dx[t] < -1 || p.dx[t] > 1) return 0;
        return 1;
    }
    if (p.in_stride == 2 && p.Hin == 2 * G8 && p.Win == 2 * G8 && p.ntaps == 25) {
        for (int t = 0; t < 25; ++t)
            if (p.dy[t] != t / 5 - 1 || p.dx[t] != t % 5 - 1) return 0;       // 5x5, TF 'SAME' at stride 2: one row/column before
        return 2;
    }
    return 0;
}

hipError_t launch_igemm_halo8(const IgemmParams& p, hipStream_t s)
{
    const int m_tiles = p.n_groups > 1 ? p.grp[p.n_groups].tile0 : (p.M + 127) / 128;
    const int tiles = m_tiles * (p.Cout / BN);
    if (igemm_halo8_mode(p) == 2) hipLaunchKernelGGL((igemm_halo8_kernel<1>), dim3(tiles), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((igemm_halo8_kernel<0>), dim3(tiles), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace p2p
