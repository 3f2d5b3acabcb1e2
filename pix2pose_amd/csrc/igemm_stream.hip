// Small-batch implicit GEMM (gfx950, PREC_F16X3): one WAVE per 32x32 (or 64x32) output tile, operands streamed
// global -> registers with a deep software pipeline, no LDS staging and no barriers.
//
// Why: the reference's real caller hands over ONE detection at a time (tools/5_evaluation_bop_basic.py:289-304,
// ros_kinetic/ros_pix2pose.py:332-333: est_pose per roi = a generator pass over 1 input, then one over K = 3).  At that
// size the batched kernels (igemm.hip, igemm_halo*.hip) put a layer on 2 - 32 workgroups whose K loops -- 300 K-steps for
// deconv1, 400 for conv4 -- advance one global-load latency per step (one staging buffer, prefetch distance 1): conv4 took
// 0.34 ms of a 2 ms pass while streaming 26 MB of weights at 78 GB/s.
//
// What keeps a detection's bits independent of the batch it travels in: an output element is ONE chain of
// v_mfma_f32_32x32x16_f16 over the layer's K-steps, in a fixed K order, and its value depends on nothing else -- not on the
// tile shape, not on which wave owns it.  Splitting K across workgroups would shorten the chain but change the fp32
// summation order with the batch size; here the chain is kept (same instruction, same K-step order as the batched kernel
// that serves the layer, same (al bh, ah bl, ah bh) order inside a step, same epilogue expression) and the parallelism comes
// from giving every 32x32 tile its own wave: conv4 at one input = 2 x 16 waves, deconv2 = 32 x 8.  What made the batched
// K loop slow is latency, so a wave keeps D K-steps of loads in flight (D x 8 KB): the 400-step chain of conv4 then runs at
// its MFMA dependency rate (6 MFMAs per step) instead of one memory round trip per step.
//
// Operands: A = fp32 activations, gathered per lane straight from the NHWC tensor (row = lane & 31, 8 consecutive channels per
// 16-deep block: two b128 loads) and split hi/lo in registers with the loaders' exact conversion (cvt_pkrtz, residual rounded
// to nearest); B = the layer's pre-split weight panel, whose 128-byte K-step record [hi x32 | lo x32] of row (n0 + lane & 31)
// is exactly the four b128 fragments a lane needs.  K-step order comes from a table (StreamOrder) describing the batched
// kernel's loop nest: groups of taps, and inside a group (channel slice, tap).
#include "kernels.h"
#include <cstdlib>

namespace p2p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr unsigned OOB = 0x80000000u;          // every tensor on this path is < 2 GB (igemm_stream_supported)
constexpr int MAX_STEPS = 512;                 // K-steps of one launch (conv4: 400)

// hi/lo split of 8 consecutive fp32 values, as hstore() / lstore() of the batched kernels do it
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

template <int TM, int TN, int D>
__global__ __launch_bounds__(64) void igemm_stream_kernel(const IgemmParams p, const StreamMulti mp)
{
    __shared__ int2 s_step[MAX_STEPS];         // x: byte shift of the A gather (tap shift + channel slice), y: weight K offset (bytes) | tap << 24 | segment << 31

    const int lane = threadIdx.x;
    const int li = lane & 31, lk = lane >> 5;
    // blockIdx.y: the sub-problem of a merged launch (the four phases of a transposed convolution: same input, same grid, own
    // weight panel / taps / output phase), or the K range of a split-K launch
    const StreamPhase& P = mp.ph[mp.n > 1 ? blockIdx.y : 0];
    const StreamOrder& o = P.o;
    const int tiles_n = p.Cout / (TN * 32);
    const int tile_n = blockIdx.x % tiles_n;
    const int tile_m = blockIdx.x / tiles_n;
    const int n0 = tile_n * TN * 32;
    const int m0 = tile_m * TM * 32;
    const int cpt = p.chunks_per_tap;
    const int ntaps = P.ntaps;
    // split-K (dense_enc): this wave walks K-steps [ks0, ks0 + total) of the layer and stores raw partial sums
    int ks0 = 0, total = ntaps * cpt;
    if (p.ksplit > 1) {
        const int ks_per = (p.ksteps + p.ksplit - 1) / p.ksplit;
        ks0 = blockIdx.y * ks_per;
        total = max(0, min(p.ksteps, ks0 + ks_per) - ks0);
    }

    // ---- K-step table: step -> (group, slice, tap of the group), the batched kernel's loop nest
    for (int i = lane; i < total; i += 64) {
        const int idx = ks0 + i;
        int g = 0;
        while (g + 1 < o.n_groups && (int)o.gstart[g + 1] * cpt <= idx) ++g;
        const int ng = (int)o.gstart[g + 1] - (int)o.gstart[g];
        const int r = idx - (int)o.gstart[g] * cpt;
        const int chunk = r / ng, k = r - chunk * ng;
        const int tap = o.tap[o.gstart[g] + k];
        const bool s1 = chunk >= p.seg0_chunks;
        const int shift_px = p.seg1_stride && s1 ? 0 : (int)P.dy[tap] * p.Win + (int)P.dx[tap];
        const int a_toff = (shift_px * (s1 ? p.seg[1].cstride : p.seg[0].cstride) + (s1 ? chunk - p.seg0_chunks : chunk) * IGEMM_BK) * 4;
        const int koff = (tap * cpt + chunk) * (IGEMM_BK * 4);
        s_step[i] = make_int2(a_toff, koff | (tap << 24) | (s1 ? (int)0x80000000 : 0));
    }

    // ---- rows of this wave: lane & 31 of each 32-row sub-tile
    unsigned a_off0[TM], a_off1[TM], a_mask[TM];
    int opx[TM];
    const int HgWg = p.Hg * p.Wg;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + i * 32 + li;
        a_off0[i] = a_off1[i] = 0;
        a_mask[i] = 0;
        opx[i] = -1;
        if (m < p.M) {
            const int n = m / HgWg;
            const int rem = m - n * HgWg;
            const int gy = rem / p.Wg;
            const int gx = rem - gy * p.Wg;
            const int iy0 = gy * p.in_stride, ix0 = gx * p.in_stride;
            const unsigned pix = (unsigned)((n * p.Hin + iy0) * p.Win + ix0);
            a_off0[i] = (pix * (unsigned)p.seg[0].cstride + (unsigned)(p.seg[0].coff + lk * 8)) * 4u;
            const unsigned pix1 = p.seg1_stride ? (unsigned)((n * p.seg1_Hin + gy * p.seg1_stride) * p.seg1_Win + gx * p.seg1_stride) : pix;
            a_off1[i] = (pix1 * (unsigned)p.seg[1].cstride + (unsigned)(p.seg[1].coff + lk * 8)) * 4u;
            unsigned mk = 0;
            for (int t = 0; t < ntaps; ++t) {
                const int iy = iy0 + (int)P.dy[t], ix = ix0 + (int)P.dx[t];
                if ((unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) mk |= 1u << t;
            }
            a_mask[i] = mk;
            opx[i] = p.ksplit > 1 ? m : (n * p.Hout + gy * p.os + P.oy) * p.Wout + gx * p.os + P.ox;
        }
    }
    unsigned b_off[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) b_off[j] = ((unsigned)(n0 + j * 32 + li) * (unsigned)P.K + (unsigned)(lk * 4)) * 4u;
    const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.seg[0].ptr, 0, p.seg_bytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.seg[1].ptr ? p.seg[1].ptr : p.seg[0].ptr), 0, p.seg_bytes[1], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, P.w_bytes, 0x00020000);

    __syncthreads();               // one wave: orders the table writes before the reads

    // ring of D K-steps of raw operands: A [kb0 k..k+3 | kb0 k+4..k+7 | kb1 .. | kb1 ..] fp32, B [hi kb0 | lo kb0 | hi kb1 | lo kb1] f16x8
    f32x4 ra[D][TM][4], rb[D][TN][4];
    auto issue = [&](int s, int step) {
        const bool live = step < total;                                    // wave-uniform
        const int2 e = s_step[live ? step : 0];
        const int a_toff = __builtin_amdgcn_readfirstlane(e.x);
        const int w1 = __builtin_amdgcn_readfirstlane(e.y);
        const bool s1 = w1 < 0;
        const unsigned bit = 1u << ((w1 >> 24) & 31);
        const int koff = w1 & 0x00FFFFFF;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned off = (live && (a_mask[i] & bit)) ? (s1 ? a_off1[i] : a_off0[i]) + (unsigned)a_toff : OOB;
            if (s1) {
                ra[s][i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a1, off, 0, 0));
                ra[s][i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a1, off + 16, 0, 0));
                ra[s][i][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a1, off + 64, 0, 0));
                ra[s][i][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a1, off + 80, 0, 0));
            } else {
                ra[s][i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a0, off, 0, 0));
                ra[s][i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a0, off + 16, 0, 0));
                ra[s][i][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a0, off + 64, 0, 0));
                ra[s][i][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a0, off + 80, 0, 0));
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const unsigned off = live ? b_off[j] + (unsigned)koff : OOB;
            rb[s][j][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));          // hi, k block 0
            rb[s][j][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off + 64, 0, 0));     // lo, k block 0
            rb[s][j][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off + 32, 0, 0));     // hi, k block 1
            rb[s][j][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off + 96, 0, 0));     // lo, k block 1
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto consume = [&](int s) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) split8(ra[s][i][2 * kb], ra[s][i][2 * kb + 1], ah[i], al[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = __builtin_bit_cast(f16x8, rb[s][j][2 * kb]);
                bl[j] = __builtin_bit_cast(f16x8, rb[s][j][2 * kb + 1]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };

#pragma unroll
    for (int s = 0; s < D; ++s) issue(s, s);
    int ks = 0;
    for (; ks + D <= total; ks += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            consume(s);
            issue(s, ks + s + D);          // steps past the end load nothing (out-of-range offsets)
        }
    }
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (ks + s < total) consume(s);

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    //      Same expression as the batched kernels: fmaf(acc, scale, shift) + residual, activation.
    if (p.ksplit > 1) {            // raw partial sums [split][m][Cout]; scale / shift / activation belong to the reduction
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = __shfl(opx[i], (r & 3) + 8 * (r >> 2) + 4 * lk, 64);
                    if (m >= 0) p.partial[((size_t)blockIdx.y * p.M + m) * p.Cout + n0 + j * 32 + li] = acc[i][j][r];
                }
        return;
    }
    float amax = 0.f;      // operand-range guard (kernels.h)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + j * 32 + li;
        const float sc = P.scale ? P.scale[col] : 1.f;
        const float sh = P.shift ? P.shift[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int ops[16];
            float rs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ops[r] = __shfl(opx[i], (r & 3) + 8 * (r >> 2) + 4 * lk, 64);
                rs[r] = 0.f;
            }
            if (p.residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (ops[r] >= 0) rs[r] = p.residual[(size_t)ops[r] * p.res_cstride + col];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (ops[r] < 0) continue;
                float v = fmaf(acc[i][j][r], sc, sh) + rs[r];
                if (p.act == ACT_RELU) v = relu_nan(v);
                else if (p.act == ACT_LEAKY) v = v > 0.f ? v : v * p.alpha;
                amax = range_note1(amax, v);
                p.out[(size_t)ops[r] * p.out_cstride + p.out_coff + col] = v;
            }
        }
    }
    range_commit(p.range_acc, amax);
}

}  // namespace

bool igemm_stream_supported(const IgemmParams& p)
{
    if (p.prec != PREC_F16X3 || p.mode != EPI_NORMAL || p.n_groups > 1) return false;
    const int steps = p.ksplit > 1 ? (p.ksteps + p.ksplit - 1) / p.ksplit : p.ntaps * p.chunks_per_tap;
    if (p.Cout % 32 || steps > MAX_STEPS || p.ntaps > IGEMM_MAX_TAPS) return false;
    if (p.seg_bytes[0] >= 0x7FFFFF00u || p.seg_bytes[1] >= 0x7FFFFF00u || p.w_bytes >= 0x7FFFFF00u || (long long)p.K * 4 >= (1 << 24)) return false;
    return true;
}

int igemm_stream_waves(const IgemmParams& p, int tm)
{
    return ((p.M + 32 * tm - 1) / (32 * tm)) * (p.Cout / 32);
}

void stream_phase_of(const IgemmParams& p, const StreamOrder& o, StreamPhase* ph)
{
    ph->w = p.w; ph->scale = p.scale; ph->shift = p.shift;
    ph->w_bytes = p.w_bytes; ph->K = p.K; ph->ntaps = p.ntaps; ph->oy = p.oy; ph->ox = p.ox;
    for (int t = 0; t < IGEMM_MAX_TAPS + 3; ++t) { ph->dy[t] = p.dy[t]; ph->dx[t] = p.dx[t]; }
    ph->o = o;
}

// mp.n == 1: the launch p describes (mp.ph[0] = stream_phase_of(p)); mp.n > 1: mp.n sub-problems that share p's input, grid, Cout and
// epilogue and differ in weight panel, taps and output phase -- the phases of a transposed convolution in one launch
hipError_t launch_igemm_stream(const IgemmParams& p, const StreamMulti& mp, hipStream_t s)
{
    const int ny = mp.n > 1 ? mp.n : (p.ksplit > 1 ? p.ksplit : 1);
    // 64 x 32 tiles once the 32 x 32 ones would put more than ~3 waves on every CU (the weight fragments are then shared by two row blocks;
    // 64 x 64 tiles per wave were measured slower at every size: 3 K-steps in flight and twice the instruction stream per wave)
    if (igemm_stream_waves(p, 1) * ny > 768) {
        hipLaunchKernelGGL((igemm_stream_kernel<2, 1, 4>), dim3(igemm_stream_waves(p, 2), ny), dim3(64), 0, s, p, mp);
    } else {
        hipLaunchKernelGGL((igemm_stream_kernel<1, 1, 6>), dim3(igemm_stream_waves(p, 1), ny), dim3(64), 0, s, p, mp);
    }
    return hipGetLastError();
}

}  // namespace p2p
