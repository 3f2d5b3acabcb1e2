// The two output heads of the generator (Conv2DTranspose 5x5/2 128->3 tanh, 128->1 sigmoid;
// reference pix2pose_model/ae_model.py:233-236) as one halo-tiled kernel for gfx950.
//
// The heads are a 3x3-tap convolution over the 64x64 grid with 16 outputs (4 sub-pixel phases x
// (x, y, z, prob)), K = 9 x 128.  As an implicit GEMM with only 16 output columns the layer is pure
// operand traffic: every input pixel is gathered nine times (once per tap) and the generic kernel
// (igemm.hip) moved 3.6 GB per launch for a 0.54 GB tensor.  Here a workgroup walks full-width row tiles
// (TH rows each) of one sample, brings the (TH+2) x 66 halo of a 32-channel slice into LDS ONCE (split into f16
// hi / lo halves on the way, like the igemm loader), and all nine taps read their operands from that
// image.  HBM traffic = the tensor once (+ halo rows out of L2) + the output.
//
// Arithmetic: PREC_F16X3 only (v_mfma_f32_16x16x32_f16, three products per block, fp32 accumulate);
// the GEMM is taken transposed (rows = the 16 outputs, columns = 16 consecutive pixels) so that a lane
// ends up with (x, y, z, prob) of one output pixel and stores a float4.  PREC_F32 models keep the
// generic kernel.
#include "kernels.h"

namespace p2p {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int HEADS_TH = 4;                 // grid rows per tile (one per wave)
constexpr int HEADS_TILES = 4;              // row tiles a workgroup walks
constexpr int HEADS_W = 64;                 // grid width (full rows: the x halo is the zero padding)
constexpr int HEADS_WP = HEADS_W + 2;
constexpr int HEADS_HP = HEADS_TH + 2;
constexpr int HEADS_REC = 144;              // bytes per halo pixel: [hi f16 x32 | lo f16 x32 | pad]; 36 dwords => conflict-free b128
constexpr int HEADS_CIN = 128;
constexpr int HEADS_CHUNKS = HEADS_CIN / 32;
constexpr int HEADS_LOADS = HEADS_HP * HEADS_W * 8 / 256;   // float4 loads per thread per chunk (12)

__global__ __launch_bounds__(256, 2) void heads_halo_kernel(const IgemmParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[HEADS_HP * HEADS_WP * HEADS_REC];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // a workgroup walks HEADS_TILES consecutive row tiles of one sample (the load pipeline keeps running
    // across tiles).  XCD-aware order: block b runs on XCD b % 8; every XCD gets a contiguous run of
    // workgroups so the halo rows two neighbours share come out of the same L2
    const int wgs_per_sample = p.Hg / (HEADS_TH * HEADS_TILES);
    const int n_wgs = p.N * wgs_per_sample;
    const int per_xcd = (n_wgs + 7) / 8;
    const int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (wg >= n_wgs) return;
    const int n = wg / wgs_per_sample;
    const int gy_first = (wg - n * wgs_per_sample) * HEADS_TH * HEADS_TILES;

    // per-object panels of a grouped launch (groups are runs of samples)
    const float* w = p.w;
    const float* scale = p.scale;
    const float* shift = p.shift;
    if (p.n_groups > 1) {
        const int row = n * p.Hg * p.Wg;
        int g = 0;
        while (g + 1 < p.n_groups && p.grp[g + 1].row0 <= row) ++g;
        w = p.grp[g].w; scale = p.grp[g].scale; shift = p.grp[g].shift;
    }

    // zero the two padding columns once (no chunk ever writes them)
    for (int i = tid; i < HEADS_HP * 2 * (HEADS_REC / 16); i += 256) {
        const int r = i / (2 * (HEADS_REC / 16)), rem = i - r * (2 * (HEADS_REC / 16));
        const int side = rem / (HEADS_REC / 16), q = rem - side * (HEADS_REC / 16);
        *reinterpret_cast<uint4*>(smem + (r * HEADS_WP + (side ? HEADS_WP - 1 : 0)) * HEADS_REC + q * 16) = make_uint4(0, 0, 0, 0);
    }

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.seg[0].ptr), 0, (int)p.seg_bytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, (int)p.w_bytes, 0x00020000);

    // halo gather: float4 idx = tid + 256 j -> pixel idx/8 (row-major over HP x 64), quad idx%8
    unsigned x_off[HEADS_LOADS];      // byte offset of the element in row gy_first - 1 + r (may be "negative": only used when valid)
    int s_off[HEADS_LOADS];
#pragma unroll
    for (int j = 0; j < HEADS_LOADS; ++j) {
        const int idx = tid + 256 * j;
        const int pix = idx >> 3, q = idx & 7;
        const int r = pix / HEADS_W, x = pix - r * HEADS_W;
        x_off[j] = (unsigned)(((((long long)n * p.Hg + gy_first - 1 + r) * HEADS_W + x) * HEADS_CIN + q * 4) * 4);
        s_off[j] = (r * HEADS_WP + x + 1) * HEADS_REC + q * 8;
    }
    f32x4 rx[HEADS_LOADS];
    auto gload = [&](int it) {                    // it = tile * HEADS_CHUNKS + chunk
        const int tile = it / HEADS_CHUNKS, chunk = it - tile * HEADS_CHUNKS;
        const int gy_top = gy_first + tile * HEADS_TH - 1;
        const unsigned tile_off = (unsigned)(tile * HEADS_TH * HEADS_W * HEADS_CIN * 4);
#pragma unroll
        for (int j = 0; j < HEADS_LOADS; ++j) {
            const int r = (tid + 256 * j) / (8 * HEADS_W);
            const int gy = gy_top + r;
            const unsigned off = (gy >= 0 && gy < p.Hg) ? x_off[j] + tile_off : 0xFFFFFFF0u;
            rx[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, chunk * 128, 0));
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < HEADS_LOADS; ++j) {
            const f32x4 v = rx[j];
            const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
            fp16x2 l01, l23;
            l01[0] = (__fp16)(v[0] - (float)h01[0]); l01[1] = (__fp16)(v[1] - (float)h01[1]);
            l23[0] = (__fp16)(v[2] - (float)h23[0]); l23[1] = (__fp16)(v[3] - (float)h23[1]);
            *reinterpret_cast<uint2*>(smem + s_off[j]) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
            *reinterpret_cast<uint2*>(smem + s_off[j] + 64) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
        }
    };

    // MFMA operands: A = weights (row = output l%16), B = pixels (column = pixel l%16); k = 8 (l/16) + i
    const int li = lane & 15, lg = lane >> 4;
    const unsigned w_lane = (unsigned)(li * p.K * 4 + lg * 16);           // bytes into the split panel
    const char* xs = smem + ((wave + 1) * HEADS_WP + li + 1) * HEADS_REC + lg * 16;

    f32x4 acc[4];
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + lg * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + lg * 4);

    gload(0);
    for (int it = 0; it < HEADS_TILES * HEADS_CHUNKS; ++it) {
        const int tile = it / HEADS_CHUNKS, chunk = it - tile * HEADS_CHUNKS;
        if (chunk == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // weight fragments of this chunk: 9 taps x (hi, lo); K order is (tap, cin), 128 B per 32-deep block
        f16x8 wh[9], wl[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kb = t * HEADS_CHUNKS + chunk;
            wh[t] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, kb * 128, 0));
            wl[t] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, kb * 128 + 64, 0));
        }
        if (it) __syncthreads();             // every wave is done reading the previous slice
        lstore();
        __syncthreads();
        if (it + 1 < HEADS_TILES * HEADS_CHUNKS) gload(it + 1);     // flies under this slice's MFMAs (and the epilogue)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const char* xt = xs + (dy * HEADS_WP + dx) * HEADS_REC;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f16x8 xh = *reinterpret_cast<const f16x8*>(xt + m * 16 * HEADS_REC);
                const f16x8 xl = *reinterpret_cast<const f16x8*>(xt + m * 16 * HEADS_REC + 64);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], xh, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], xl, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], xh, acc[m], 0, 0, 0);
            }
        }
        if (chunk == HEADS_CHUNKS - 1) {
            // lane: outputs 4 lg .. 4 lg + 3 = (x, y, z, prob) of phase lg for grid pixel (gy0 + wave, 16 m + li)
            const int oy = 2 * (gy_first + tile * HEADS_TH + wave) + (lg >> 1);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int ox = 2 * (16 * m + li) + (lg & 1);
                f32x4 v = acc[m], o;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
                o[0] = tanhf(v[0]); o[1] = tanhf(v[1]); o[2] = tanhf(v[2]);
                o[3] = 1.f / (1.f + __expf(-v[3]));
                *reinterpret_cast<f32x4*>(p.out + (((size_t)n * p.Hout + oy) * p.Wout + ox) * 4) = o;
            }
        }
    }
}

}  // namespace

bool heads_halo_supported(const IgemmParams& p)
{
    return p.prec == PREC_F16X3 && p.mode == EPI_HEAD && p.ntaps == 9 && p.Cout == 16 && p.Wg == HEADS_W && p.Hg % (HEADS_TH * HEADS_TILES) == 0 &&
           p.Hin == p.Hg && p.Win == p.Wg && p.in_stride == 1 && p.os == 2 && p.Hout == 2 * p.Hg && p.Wout == 2 * p.Wg &&
           p.seg[0].C == HEADS_CIN && p.seg[0].cstride == HEADS_CIN && p.seg[0].coff == 0 && p.seg[1].C == 0 && p.ksplit <= 1 &&
           p.dy[0] == -1 && p.dx[0] == -1 && p.dy[8] == 1 && p.dx[8] == 1;
}

hipError_t launch_heads_halo(const IgemmParams& p, hipStream_t s)
{
    const int n_wgs = p.N * (p.Hg / (HEADS_TH * HEADS_TILES));
    const int per_xcd = (n_wgs + 7) / 8;
    hipLaunchKernelGGL(heads_halo_kernel, dim3(per_xcd * 8), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace p2p
