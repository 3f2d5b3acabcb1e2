// EXPERIMENT (round 4), NOT part of the library.  Built, routed for the identity blocks' 2a / 2c layers, verified BIT-IDENTICAL to the generic
// and the small-launch routes (tests/test_stream_gpu.py green: swapping the MFMA operand roles does not change a bit) -- and 1.6 - 2.5x SLOWER
// than igemm.hip (res2b_2c 128 -> 225 us, res3b_2a 45 -> 107 us per 256 inputs): a lane-per-pixel operand gather and 32-byte store pieces
// touch 32 cache lines per instruction, the texture-address unit becomes the bound and the partial-line writes halve the store efficiency.
// What the generic kernel spends on LDS staging is exactly what makes its global accesses full-line.  Kept for the record.
//
// The 1x1 convolutions of the ResNet bottleneck blocks (reference pix2pose_model/resnet50_mod.py:40-118: the `2a` and `2c` layers of the
// identity blocks) at large batch, gfx950, PREC_F16X3: a streaming kernel without LDS and without barriers.
//
// These layers are plain GEMMs over pixels with K = 64 .. 512 and they are bound by the bytes they move (the block input is read by `2a` and
// again as the residual of `2c`, which also writes the block output: DESIGN.md section 5).  The generic kernel (igemm.hip) walks a workgroup
// through four serial phases -- stage operands, K loop, two epilogue passes through an LDS transpose, each exposing one memory latency -- and
// sustains 4.3 - 4.8 TB/s of algorithmic traffic where a plain elementwise stream reaches 5.9 (tools/bw_probe.py).  Here every WAVE is its
// own pipeline: it owns 32 pixels x 64 output channels, issues its operand loads and the residual loads up front, and stores straight from
// the accumulators.
//
// Orientation: the MFMA is taken TRANSPOSED -- A operand = the weight panel's rows (output channels), B operand = the pixels -- so that the
// C/D layout (column = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) leaves a lane with FOUR CONSECUTIVE CHANNELS of one pixel per
// accumulator quad: scale / shift / residual / output move as float4 with no transpose.  The products are the ones igemm.hip and
// igemm_stream.hip form, in their order -- (w hi x lo), (w lo x hi), (w hi x hi) per 16-deep block, blocks and K-steps ascending -- and the
// matrix instruction does not care which operand carries the rows: the outputs are bit-identical to the generic route
// (tests/test_stream_gpu.py compares the routes).
#include "kernels.h"
#include <cstdlib>

namespace p2p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int NB = 2;                    // 32-channel blocks per wave tile (64 output channels)
constexpr int MAX_KSTEPS = 16;           // K <= 512

// hi/lo split of 8 consecutive fp32 values, as the loaders of the batched kernels do it (igemm.hip: lstore)
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, f16x8& hi, f16x8& lo)
{
    const fp16x2 h0 = __builtin_amdgcn_cvt_pkrtz(a[0], a[1]), h1 = __builtin_amdgcn_cvt_pkrtz(a[2], a[3]);
    const fp16x2 h2 = __builtin_amdgcn_cvt_pkrtz(b[0], b[1]), h3 = __builtin_amdgcn_cvt_pkrtz(b[2], b[3]);
    fp16x2 l0, l1, l2, l3;          // residuals are exact in fp32; round them to nearest
    l0[0] = (__fp16)(a[0] - (float)h0[0]); l0[1] = (__fp16)(a[1] - (float)h0[1]);
    l1[0] = (__fp16)(a[2] - (float)h1[0]); l1[1] = (__fp16)(a[3] - (float)h1[1]);
    l2[0] = (__fp16)(b[0] - (float)h2[0]); l2[1] = (__fp16)(b[1] - (float)h2[1]);
    l3[0] = (__fp16)(b[2] - (float)h3[0]); l3[1] = (__fp16)(b[3] - (float)h3[1]);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 hv = {__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1), __builtin_bit_cast(unsigned, h2), __builtin_bit_cast(unsigned, h3)};
    const u32x4 lv = {__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1), __builtin_bit_cast(unsigned, l2), __builtin_bit_cast(unsigned, l3)};
    hi = __builtin_bit_cast(f16x8, hv);
    lo = __builtin_bit_cast(f16x8, lv);
}

template <int D>          // K-steps of operands in flight per wave
__global__ __launch_bounds__(256, 2) void igemm_1x1_kernel(const IgemmParams p)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    // XCD-aware order (block b runs on XCD b % 8): every XCD gets a contiguous run of wave tiles, channel tile fastest -- the waves that
    // share a pixel block (and read the same activations) sit in one workgroup
    const int ct = p.Cout / (32 * NB);
    const int n_tiles = (p.M / 32) * ct;
    int t;
    {
        const int nblk = gridDim.x, b = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = b & 7, idx = b >> 3;
        t = ((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx) * 4 + wave;
    }
    if (t >= n_tiles) return;
    const int m = (t / ct) * 32 + li;                  // this lane's pixel (B operand column, C/D column)
    const int n0 = (t % ct) * (32 * NB);

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.seg[0].ptr, 0, p.seg_bytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const unsigned x_off = ((unsigned)m * (unsigned)p.seg[0].cstride + (unsigned)(p.seg[0].coff + lk * 8)) * 4u;
    unsigned w_off[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) w_off[c] = ((unsigned)(n0 + c * 32 + li) * (unsigned)p.K + (unsigned)(lk * 4)) * 4u;

    // ring of D K-steps of raw operands: x [kb0 k..k+3 | kb0 k+4..k+7 | kb1 .. | kb1 ..] fp32, w [hi kb0 | lo kb0 | hi kb1 | lo kb1] f16x8
    f32x4 rx[D][4], rw[D][NB][4];
    const int ksteps = p.ksteps;
    auto issue = [&](int s, int step) {
        const unsigned ko = step < ksteps ? (unsigned)step * 128u : 0x80000000u;        // past the end: out of range, the hardware returns zeros
        rx[s][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, x_off + ko, 0, 0));
        rx[s][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, x_off + ko + 16, 0, 0));
        rx[s][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, x_off + ko + 64, 0, 0));
        rx[s][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, x_off + ko + 80, 0, 0));
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            rw[s][c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off[c] + ko, 0, 0));          // hi, k block 0
            rw[s][c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off[c] + ko + 64, 0, 0));     // lo, k block 0
            rw[s][c][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off[c] + ko + 32, 0, 0));     // hi, k block 1
            rw[s][c][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off[c] + ko + 96, 0, 0));     // lo, k block 1
        }
    };
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s, s);

    // the residual (the block input: the bigger stream of a `2c` layer) is requested before the K loop starts
    const size_t o_pix = (size_t)m * p.out_cstride + p.out_coff + n0 + 4 * lk;
    const size_t r_pix = (size_t)m * p.res_cstride + n0 + 4 * lk;
    f32x4 rs[NB][4];
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            rs[c][g] = p.residual ? *reinterpret_cast<const f32x4*>(p.residual + r_pix + c * 32 + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};

    f32x16 acc[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    auto consume = [&](int s) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f16x8 xh, xl;
            split8(rx[s][2 * kb], rx[s][2 * kb + 1], xh, xl);
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const f16x8 wh = __builtin_bit_cast(f16x8, rw[s][c][2 * kb]);
                const f16x8 wl = __builtin_bit_cast(f16x8, rw[s][c][2 * kb + 1]);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc[c], 0, 0, 0);       // (activation lo) x (weight hi)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc[c], 0, 0, 0);       // (activation hi) x (weight lo)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc[c], 0, 0, 0);       // (activation hi) x (weight hi)
            }
        }
    };
    int ks = 0;
    for (; ks + D <= ksteps; ks += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            consume(s);
            issue(s, ks + s + D);
        }
    }
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (ks + s < ksteps) consume(s);

    // epilogue: the generic kernels' expression -- fmaf(acc, scale, shift) + residual, activation -- on the lane's channel quads
    float amax = 0.f;      // operand-range guard (kernels.h)
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = n0 + c * 32 + 8 * g + 4 * lk;
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + ch);
            if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + ch);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = fmaf(acc[c][4 * g + e], sc[e], sh[e]);
                v[e] += rs[c][g][e];
            }
            if (p.act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (p.act == ACT_LEAKY) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.alpha;
            }
            amax = range_note4(amax, v);
            *reinterpret_cast<f32x4*>(p.out + o_pix + c * 32 + 8 * g) = v;
        }
    range_commit(p.range_acc, amax);
}

}  // namespace

bool igemm_1x1_supported(const IgemmParams& p)
{
    static const bool on = getenv("P2P_NO_1X1") == nullptr;      // development switch (A/B; same bits)
    if (!on || p.prec != PREC_F16X3 || p.mode != EPI_NORMAL || p.ksplit > 1 || p.n_groups > 1) return false;
    if (p.ntaps != 1 || p.dy[0] != 0 || p.dx[0] != 0 || p.in_stride != 1 || p.seg[1].C != 0 || p.seg1_stride) return false;
    if (p.Hin != p.Hg || p.Win != p.Wg || p.os != 1 || p.oy || p.ox || p.Hout != p.Hg || p.Wout != p.Wg) return false;
    if (p.M % 32 || p.Cout % (32 * NB) || p.ksteps < 1 || p.ksteps > MAX_KSTEPS) return false;
    if (p.seg_bytes[0] >= 0x7FFFFF00u || p.w_bytes >= 0x7FFFFF00u) return false;
    return true;
}

hipError_t launch_igemm_1x1(const IgemmParams& p, hipStream_t s)
{
    const int tiles = (p.M / 32) * (p.Cout / (32 * NB));
    const int blocks = (tiles + 3) / 4;
    if (p.ksteps >= 4) hipLaunchKernelGGL(igemm_1x1_kernel<2>, dim3(blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(igemm_1x1_kernel<2>, dim3(blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace p2p
