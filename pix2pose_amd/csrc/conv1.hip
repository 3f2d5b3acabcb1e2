// First layer of the ResNet-50 front (ZeroPadding2D(3) + Conv2D 7x7/2, 3 -> 64, BatchNorm, ReLU;
// reference pix2pose_model/resnet50_mod.py:200-203) on the f16 matrix cores, PREC_F16X3 models only.
//
// Cin = 3 makes the layer useless for the generic implicit-GEMM kernel (K = 147, 12-byte pixels), and the
// VALU kernel (misc_kernels.hip, still used by PREC_F32 models and the paper backbone) spends its time
// re-reading weights.  Here the contraction is laid out per kernel ROW: for one kh the 7 taps x 3
// channels of an output pixel are 7 consecutive input pixels, staged in LDS as 4-channel f16 pixels
// (r, g, b, 0) so the run is 28 (+4 zero-weighted) contiguous halves = one K = 32 MFMA block that a lane
// reads with a single aligned ds_read_b128.  K = 7 blocks of 32 (34 % padding — irrelevant, the layer is
// bound by its 268 MB output).  A workgroup keeps the whole split weight panel (56 KB) in LDS and walks
// 16 output rows of one image, two rows (nine input rows) at a time.
//
// GEMM orientation: rows = 16 output channels, columns = 16 consecutive output pixels, so a lane ends
// up with 4 consecutive channels of one pixel (float4 store).
#include "kernels.h"
#include <cstdlib>

namespace p2p {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

namespace {

// The same kernel serves the first layer of the "paper" encoder (conv1_1 || conv1_2: Conv2D 5x5/2 'SAME', 3 -> 64 + 64, BatchNorm,
// LeakyReLU; reference ae_model.py:74-78): KH = 5 kernel rows of 5 taps (20 + 12 zero-weighted halves per K block), TF 'SAME' at
// stride 2 pads ONE pixel before, and the 128 output channels are two workgroup columns of 64 (blockIdx.y), each with its own
// half of the weight panel in LDS.
constexpr int C1_COUT = 64;                                      // output channels per workgroup
constexpr int C1_HIN = 128, C1_HOUT = 64;
constexpr int C1_ROWS_OUT = 2;                                   // output rows per tile
constexpr int C1_ROW_PX = C1_HIN + 6;                            // 134 staged pixels per row (pixel -PAD first; the last run ends at 133)
constexpr int C1_ROW_BYTES = C1_ROW_PX * 8;                      // 4 halves per pixel; 1072 B, 16-B aligned
constexpr int C1_TILES_PER_WG = 8;                               // 16 output rows per workgroup (large batches; small ones: one tile, 32 workgroups per image)

// POOL (resnet50 front, large batches): the 3x3 / 2 'same' max-pool that follows the layer (resnet50_mod.py:204) is taken from the
// accumulators and only the 32 channels the decoder's skip connection reads (ae_model.py:186) are stored at full resolution -- the
// layer wrote 268 MB per 256 inputs for the pooling kernel to fetch again (402 MB with its halo re-reads) and 67 MB to come out.
// A wave then owns BOTH rows of a tile for 16 pixels: the vertical maximum is per lane, the row shared with the next tile
// is carried in registers (the same lane owns the same pixel and channels there), the horizontal one is two DPP row shifts
// (a pixel tile is one DPP row) plus one pixel per wave exchanged through LDS.  A workgroup computes one more tile than it stores
// (the row below its last pooled row).  Values are the same ones the two kernels produce: max is exact.
template <int CTRL>
__device__ __forceinline__ float dpp_row(const float v)       // lane i of a 16-lane row <- lane i + n (row_shl:n); out of the row: own value
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

template <int C1_KH, int C1_PAD, int NCO, bool POOL>
__global__ __launch_bounds__(256, 2) void conv1_f16x3_kernel(const float* __restrict__ x, int N, const Conv1Groups G,
                                                             int act, float alpha, float* __restrict__ out, int tiles_per_wg,
                                                             float* __restrict__ pool_out, unsigned* __restrict__ range_acc)
{
    float amax = 0.f;      // operand-range guard (kernels.h)
    constexpr int C1_ROWS_IN = (C1_ROWS_OUT - 1) * 2 + C1_KH;        // 9 / 7 input rows
    constexpr int C1_PLANE = C1_ROWS_IN * C1_ROW_BYTES;              // hi plane, then lo plane
    constexpr int C1_W_BYTES = C1_KH * 2 * 4 * C1_COUT * 16;         // [kh][hi,lo][k-group][cout][8 halves] = 57344 / 40960
    constexpr int C1_PASSES = (C1_ROWS_IN * C1_HIN + 255) / 256;     // pixel loads per thread per tile (5 / 4)
    const int co0 = blockIdx.y * C1_COUT;                            // this workgroup's output channels
    __shared__ __attribute__((aligned(16))) char smem[C1_W_BYTES + 2 * C1_PLANE];
    __shared__ float s_ex[POOL ? 4 * 64 : 1];                        // POOL: first pixel of every wave's strip, 64 channels
    char* ws = smem;
    char* xs = smem + C1_W_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wgs_per_img = C1_HOUT / (C1_ROWS_OUT * tiles_per_wg);
    const int n = blockIdx.x / wgs_per_img;
    const int oy_base = (blockIdx.x - n * wgs_per_img) * C1_ROWS_OUT * tiles_per_wg;
    int g = 0;                                  // object of this sample (mixed batches: groups are runs of samples)
    while (g + 1 < G.n_groups && G.start[g + 1] <= n) ++g;
    const float* __restrict__ w_alt = G.w[g] + (size_t)blockIdx.y * (C1_W_BYTES / 4);
    const float* __restrict__ scale = G.scale[g] + co0;
    const float* __restrict__ shift = G.shift[g] + co0;

    // weights -> LDS (linear copy), zero the staging planes once (the padding pixels stay zero)
    for (int i = tid; i < C1_W_BYTES / 16; i += 256)
        reinterpret_cast<uint4*>(ws)[i] = reinterpret_cast<const uint4*>(w_alt)[i];
    for (int i = tid; i < 2 * C1_PLANE / 16; i += 256) reinterpret_cast<uint4*>(xs)[i] = make_uint4(0, 0, 0, 0);

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)((size_t)N * C1_HIN * C1_HIN * 12), 0x00020000);

    float rx[C1_PASSES][3];
    auto gload = [&](int oy0) {
        const int iy0 = 2 * oy0 - C1_PAD;
#pragma unroll
        for (int j = 0; j < C1_PASSES; ++j) {
            const int idx = tid + 256 * j;
            const int r = idx >> 7, px = idx & 127;
            const int iy = iy0 + r;
            const bool ok = r < C1_ROWS_IN && iy >= 0 && iy < C1_HIN;
            const unsigned off = ok ? (unsigned)((((size_t)n * C1_HIN + iy) * C1_HIN + px) * 12) : 0xFFFFFFF0u;
            // three dword loads: hipcc (ROCm 7.2) lowers __builtin_amdgcn_raw_buffer_load_b96 to ONE dword
#pragma unroll
            for (int e = 0; e < 3; ++e) rx[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, off, e * 4, 0));
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < C1_PASSES; ++j) {
            const int idx = tid + 256 * j;
            const int r = idx >> 7, px = idx & 127;
            if (r >= C1_ROWS_IN) continue;
            const float* v = rx[j];
            const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), h2 = __builtin_amdgcn_cvt_pkrtz(v[2], 0.f);
            fp16x2 l01, l2;
            l01[0] = (__fp16)(v[0] - (float)h01[0]); l01[1] = (__fp16)(v[1] - (float)h01[1]);
            l2[0] = (__fp16)(v[2] - (float)h2[0]); l2[1] = (__fp16)0.f;
            char* d = xs + r * C1_ROW_BYTES + (px + C1_PAD) * 8;
            *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h2));
            *reinterpret_cast<uint2*>(d + C1_PLANE) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l2));
        }
    };

    // wave -> output row (wave >> 1) of the tile, pixels 32 (wave & 1) .. +31 (two 16-pixel column tiles)
    const int li = lane & 15, lg = lane >> 4;
    // POOL: wave -> both rows of the tile (m = row), pixels 16 wave .. +15
    const int orow = POOL ? 0 : wave >> 1, ox0 = POOL ? wave * 16 : (wave & 1) * 32;
    constexpr int M_STEP = POOL ? 2 * C1_ROW_BYTES : 256;            // second pixel tile: the next output row / the next 16 pixels
    // pixel operand: the run of output pixel ox starts at staged pixel 2 ox (= input pixel 2 ox - 3): byte 16 ox
    const char* xb = xs + (2 * orow) * C1_ROW_BYTES + (ox0 + li) * 16 + lg * 16;
    const char* wb = ws + lg * (C1_COUT * 16) + li * 16;

    // POOL: pooled row r = max over output rows 2r .. 2r + 2: one tile past the stored ones, unless that is the padding row
    const int tiles_run = tiles_per_wg + ((POOL && oy_base + C1_ROWS_OUT * tiles_per_wg < C1_HOUT) ? 1 : 0);
    f32x4 carry[4];                          // POOL: max of the previous tile's two rows at this lane's pixel
    auto emit = [&](const f32x4 (&wv)[4], int prow) {      // horizontal 3-max at the even pixels of row-maximum wv -> pooled row prow
        if (li == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(&s_ex[wave * 64 + q * 16 + lg * 4]) = wv[q];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = fmaxf(wv[q][e], fmaxf(dpp_row<0x101>(wv[q][e]), dpp_row<0x102>(wv[q][e])));
            if (li == 14 && wave < 3) {      // pixel x + 2 is the next wave's first one (past the last wave: the padding column)
                const f32x4 o = *reinterpret_cast<const f32x4*>(&s_ex[(wave + 1) * 64 + q * 16 + lg * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = fmaxf(h[e], o[e]);
            }
            if (!(li & 1))
                *reinterpret_cast<f32x4*>(pool_out + (((size_t)n * (C1_HOUT / 2) + prow) * (C1_HOUT / 2) + ((ox0 + li) >> 1)) * C1_COUT + q * 16 + lg * 4) = h;
        }
    };

    gload(oy_base);
    for (int t = 0; t < tiles_run; ++t) {
        const int oy0 = oy_base + t * C1_ROWS_OUT;
        __syncthreads();                     // previous tile's reads (and the initial fills) are done
        lstore();
        __syncthreads();
        if (t + 1 < tiles_run) gload(oy0 + C1_ROWS_OUT);

        f32x4 acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < C1_KH; ++kh) {
            f16x8 xh[2], xl[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                xh[m] = *reinterpret_cast<const f16x8*>(xb + kh * C1_ROW_BYTES + m * M_STEP);
                xl[m] = *reinterpret_cast<const f16x8*>(xb + kh * C1_ROW_BYTES + m * M_STEP + C1_PLANE);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(wb + (kh * 2 + 0) * (4 * C1_COUT * 16) + q * 256);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(wb + (kh * 2 + 1) * (4 * C1_COUT * 16) + q * 256);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[m], acc[m][q], 0, 0, 0);
                    acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[m], acc[m][q], 0, 0, 0);
                    acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[m], acc[m][q], 0, 0, 0);
                }
            }
        }
        if (POOL) {
            // lane: channels 16 q + 4 lg .. +3 of output pixels (oy0 + m, ox0 + li)
            f32x4 v[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + q * 16 + lg * 4);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + q * 16 + lg * 4);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float u = fmaf(acc[m][q][e], sc[e], sh[e]);
                        if (act == ACT_RELU) u = relu_nan(u);
                        else if (act == ACT_LEAKY) u = u > 0.f ? u : u * alpha;
                        v[m][q][e] = u;
                    }
#pragma unroll
                for (int m = 0; m < 2; ++m) amax = range_note4(amax, v[m][q]);      // the pooled values are maxima of these
            }
            if (t < tiles_per_wg) {          // the skip connection's channels 0 .. 31 of both rows
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float* op = out + (((size_t)n * C1_HOUT + oy0 + m) * C1_HOUT + ox0 + li) * NCO + co0 + lg * 4;
                    *reinterpret_cast<f32x4*>(op) = v[m][0];
                    *reinterpret_cast<f32x4*>(op + 16) = v[m][1];
                }
            }
            if (t > 0) {                     // the row this tile shares with the pooled row of the tile before
                f32x4 wv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) wv[q][e] = fmaxf(carry[q][e], v[0][q][e]);
                emit(wv, (oy0 >> 1) - 1);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) carry[q][e] = fmaxf(v[0][q][e], v[1][q][e]);
            continue;
        }
        // lane: channels 16 q + 4 lg .. +3 of output pixel (oy0 + orow, ox0 + 16 m + li)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float* op = out + (((size_t)n * C1_HOUT + oy0 + orow) * C1_HOUT + ox0 + 16 * m + li) * NCO + co0 + lg * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + q * 16 + lg * 4);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + q * 16 + lg * 4);
                f32x4 v = acc[m][q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float u = fmaf(v[e], sc[e], sh[e]);
                    if (act == ACT_RELU) u = relu_nan(u);
                    else if (act == ACT_LEAKY) u = u > 0.f ? u : u * alpha;
                    v[e] = u;
                }
                amax = range_note4(amax, v);
                *reinterpret_cast<f32x4*>(op + q * 16) = v;
            }
        }
    }
    if (POOL && tiles_run == tiles_per_wg) {      // last pooled row of the image: the row below is padding
        __syncthreads();                           // the last emit's exchange reads are done
        emit(carry, (C1_HOUT >> 1) - 1);
    }
    range_commit(range_acc, amax);
}

}  // namespace

size_t conv1_f16x3_panel_floats(int KH, int Cout) { return (size_t)(Cout / C1_COUT) * KH * 2 * 4 * C1_COUT * 16 / 4; }

// k within a kernel row: kk = kw * 4 + c (c < 3), group = kk / 8; layout [cout / 64][kh][hi,lo][group][cout % 64][8 halves]
size_t conv1_f16x3_panel_index(int KH, int kh, int plane, int kw, int c, int cout)
{
    const int kk = kw * 4 + c;
    return (((((size_t)(cout / C1_COUT) * KH + kh) * 2 + plane) * 4 + (kk >> 3)) * C1_COUT + cout % C1_COUT) * 8 + (kk & 7);
}

bool conv1_f16x3_supported(int KH, int Cout) { return (KH == 7 && Cout == 64) || (KH == 5 && Cout == 128); }

// KH = 7: ZeroPadding2D(3) + 7x7/2 'valid', 3 -> 64 (resnet50 front); KH = 5: 5x5/2 'SAME' (one pixel before), 3 -> 128 (paper encoder)
// pool_out (KH = 7 only): also produce MaxPooling2D(3, 2, 'same') of the layer, [N][32][32][64]; `out` is then only guaranteed to hold
// channels 0 .. 31 (the decoder's skip connection)
hipError_t launch_conv1_f16x3(const float* x, int N, int KH, int Cout, const Conv1Groups& G, int act, float alpha, float* out, float* pool_out,
                              unsigned* range_acc, hipStream_t s)
{
    if (N <= 0) return hipSuccess;
    if (!conv1_f16x3_supported(KH, Cout) || G.n_groups < 1 || G.n_groups > IGEMM_MAX_GROUPS) return hipErrorInvalidValue;
    if ((size_t)N * C1_HIN * C1_HIN * 12 >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    // a handful of images (one detection at a time): one 2-row tile per workgroup so that the layer covers the chip
    // (an output pixel is computed the same way whichever workgroup owns its row)
    // 16 .. 127 images: 16 rows per workgroup would leave the layer on 64 .. 508 workgroups of four waves (a quarter to a whole wave of
    // one workgroup per CU): 4- and 8-row workgroups instead (the pooled form computes one tile more than it stores: 3/2 and 5/4 of the
    // products, on a layer bound by its stores)
    static const int tpw_mid = dev_env("P2P_CONV1_TPW") ? atoi(dev_env("P2P_CONV1_TPW")) : 0;      // development builds: tiles per workgroup for 16 .. 127 images
    const int tpw = N >= 128 ? C1_TILES_PER_WG : N >= 16 ? (tpw_mid ? tpw_mid : N >= 64 ? 4 : 2) : (N >= 4 ? 2 : 1);
    const int wgs = N * (C1_HOUT / (C1_ROWS_OUT * tpw));
    static const bool no_fuse = dev_env("P2P_NO_POOL_FUSE") != nullptr;      // development switch (A/B)
    if (KH == 7 && pool_out && N >= 16 && !no_fuse) {
        hipLaunchKernelGGL((conv1_f16x3_kernel<7, 3, 64, true>), dim3(wgs), dim3(256), 0, s, x, N, G, act, alpha, out, tpw, pool_out, range_acc);
        return hipGetLastError();
    }
    if (KH == 7) hipLaunchKernelGGL((conv1_f16x3_kernel<7, 3, 64, false>), dim3(wgs), dim3(256), 0, s, x, N, G, act, alpha, out, tpw, nullptr, range_acc);
    else hipLaunchKernelGGL((conv1_f16x3_kernel<5, 1, 128, false>), dim3(wgs, 2), dim3(256), 0, s, x, N, G, act, alpha, out, tpw, nullptr, range_acc);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && KH == 7 && pool_out) e = launch_maxpool3s2(out, N, C1_HOUT, C1_HOUT, Cout, pool_out, s);
    return e;
}

}  // namespace p2p
