// Implicit-GEMM NHWC convolution for gfx950 (CDNA4) on the fp32 matrix cores.
//
//   v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 cycles/instr/SIMD,
//   157 TFLOP/s chip peak.  The Pix2Pose parity bar (XYZ within 1e-3 abs through ~30 layers)
//   is why the arithmetic is fp32 and not bf16 (DESIGN.md "Precision").
//
// One kernel serves every dense contraction of the generator graphs
// (reference pix2pose_model/ae_model.py:70-150,175-240; resnet50_mod.py:40-118):
//   * Conv2D 1x1 / 3x3 / 5x5, stride 1 or 2, TF 'SAME' or 'valid'  (tap table dy/dx)
//   * Conv2DTranspose 5x5/2 as four sub-pixel phase convolutions    (os = 2, oy/ox = phase)
//   * Dense layers as 1-tap convolutions over a 1x1 grid            (optionally split-K)
//   * skip concatenation as a two-segment channel gather (no copy), channel slices by stride
//   * folded BatchNorm scale/shift, residual add, ReLU / LeakyReLU, or the tanh/sigmoid heads
//     in the epilogue.
//
// Tiling: 256 threads = 4 wave64; workgroup tile BM x BN (128x128 / 128x64 / 128x32), K-step 32.
// Both operands are staged K-contiguous in LDS (row stride 36 floats => conflict-free
// ds_read_b128: 16 rows x 4 banks cover all 64 banks).  A lane's b128 holds 4 consecutive k;
// lanes 0-31 take k..k+3 and lanes 32-63 take k+4..k+7, so the j-th element of every lane
// forms one 32x32x2 MFMA step (the k-permutation is the same for A and B, which is all a
// contraction needs).  Global->LDS is register-staged (the padded LDS image rules out
// global_load_lds); the global loads of step k+1 are issued before the 64 (128x128 tile) MFMAs
// of step k and written to the single LDS buffer between two barriers.
//
// Precision modes (template PREC):
//   PREC_F32   v_mfma_f32_32x32x2_f32, bitwise an fp32 fmaf chain (157 TFLOP/s peak).
//   PREC_F16X3 fp32 emulated on the f16 matrix pipe: every fp32 value x is split as hi = f16(x),
//              lo = f16(x - hi) (22 significant bits) and a*b ~= ah*bh + ah*bl + al*bh with fp32
//              accumulation: three v_mfma_f32_32x32x16_f16 per 16-deep block, 16x the fp32 MFMA rate
//              per instruction => ~5.3x per algorithmic MAC.  Activations stay fp32 in HBM and are
//              split by the loader when it writes the LDS tile (measured free: the VALU work hides
//              under the MFMAs); weights are split offline into the same [hi x32 | lo x32] row image,
//              pre-scaled by 2^10 so their lo parts stay in the f16 normal range (the epilogue scale
//              carries 2^-10).  Error vs exact fp32 products ~2^-22 relative: the network output moves
//              by ~1e-6, the same size as fp32 summation-order noise (tests hold both modes to 1e-4).
#include "kernels.h"

namespace p2p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

constexpr int LDS_LD = IGEMM_BK + 4;  // padded row stride (floats)

template <int WGM, int WGN, int TM, int TN, int PREC>
__global__ __launch_bounds__(256, 3) void igemm_kernel(const IgemmParams p)   // >= 3 waves per SIMD: <= 168 VGPR+AGPR
{
    constexpr int BM = WGM * TM * 32;
    constexpr int BN = WGN * TN * 32;
    constexpr int A_PASSES = BM / 32;
    constexpr int B_PASSES = BN / 32;
    static_assert(WGM * WGN == 4, "4 waves per workgroup");

    // one staging buffer (two barriers per K-step) + a (TM*32)-row C tile for the epilogue: 38 KB for
    // the 128x128 tile => 3 workgroups per CU (VGPR-limited) instead of 2; measured equal or better
    // than double buffering on every layer shape (tools/igemm_exp.hip).
    constexpr int STAGE_FLOATS = (BM + BN) * LDS_LD;
    constexpr int CTILE_FLOATS = TM * 32 * (BN + 4);
    __shared__ __attribute__((aligned(16))) float smem[STAGE_FLOATS > CTILE_FLOATS ? STAGE_FLOATS : CTILE_FLOATS];
    __shared__ int row_base[BM];   // n*Hin*Win, or -1 for rows past M
    __shared__ int row_yx[BM];     // (iy0 << 16) | ix0
    __shared__ int row_pix1[BM];   // pixel index into segment 1 when it lives on its own grid (seg1_stride != 0)
    __shared__ int row_out[BM];    // output pixel index, or -1
    __shared__ int s_tap[IGEMM_MAX_TAPS + 3];   // dy * Win + dx (pixels): read from LDS in the K loop --
                                                // indexing the kernarg arrays there costs a dependent
                                                // global load + wait at the top of every K-step

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, lk = lane >> 5;

    // ---- XCD-aware tile mapping: workgroup b runs on XCD b%8; give each XCD a contiguous
    //      run of tiles (n-tile fastest) so A rows and the weight panel are shared in its L2.
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const int nblk = gridDim.x;
    int t;
    {
        const int b = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = t % tiles_n;
    const int tile_m = t / tiles_n;
    const int n0 = tile_n * BN;
    // grouped launch: this tile's object decides the weight panel and the row range
    int m0 = tile_m * BM, m_end = p.M;
    const float* gw = p.w;
    const float* gscale = p.scale;
    const float* gshift = p.shift;
    if (p.n_groups > 1) {
        int g = 0;
        while (g + 1 < p.n_groups && tile_m >= p.grp[g + 1].tile0) ++g;
        m0 = p.grp[g].row0 + (tile_m - p.grp[g].tile0) * BM;
        m_end = p.grp[g + 1].row0;
        gw = p.grp[g].w; gscale = p.grp[g].scale; gshift = p.grp[g].shift;
    }

    if (tid < p.ntaps) s_tap[tid] = (int)p.dy[tid] * p.Win + (int)p.dx[tid];
    const int HgWg = p.Hg * p.Wg;
    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r;
        int base = -1, yx = 0, op = -1;
        if (m < m_end) {
            const int n = m / HgWg;
            const int rem = m - n * HgWg;
            const int gy = rem / p.Wg;
            const int gx = rem - gy * p.Wg;
            base = n * p.Hin * p.Win;
            yx = ((gy * p.in_stride) << 16) | (gx * p.in_stride);
            op = (n * p.Hout + gy * p.os + p.oy) * p.Wout + gx * p.os + p.ox;
        }
        row_base[r] = base;
        row_yx[r] = yx;
        if (p.seg1_stride) {
            int px1 = 0;
            if (m < m_end) {
                const int n = m / HgWg;
                const int rem = m - n * HgWg;
                const int gy = rem / p.Wg;
                px1 = (n * p.seg1_Hin + gy * p.seg1_stride) * p.seg1_Win + (rem - gy * p.Wg) * p.seg1_stride;
            }
            row_pix1[r] = px1;
        }
        row_out[r] = op;
    }
    __syncthreads();

    // ---- loader state: thread handles rows (tid>>3)+32*j, 16-byte column segment (tid&7).
    //      The gather uses raw buffer loads (32-bit per-lane byte offsets against a wave-uniform
    //      descriptor): per K-step and row it costs one add, one bit test and one select -- a tap
    //      that falls outside the image gets an out-of-range offset and the hardware returns zeros,
    //      so there is no branch and no pointer arithmetic in the loop.  Which taps are inside the
    //      image is a per-row bit mask computed once per tile.
    const int lrow = tid >> 3;
    const int lcol = (tid & 7) * 4;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned a_off0[A_PASSES], a_off1[A_PASSES], a_mask[A_PASSES];
#pragma unroll
    for (int j = 0; j < A_PASSES; ++j) {
        const int base = row_base[lrow + 32 * j];
        const int yx = row_yx[lrow + 32 * j];
        const int iy0 = yx >> 16, ix0 = yx & 0xffff;
        const unsigned pix = (unsigned)(base + iy0 * p.Win + ix0);
        a_off0[j] = (pix * (unsigned)p.seg[0].cstride + (unsigned)(p.seg[0].coff + lcol)) * 4u;
        const unsigned pix1 = p.seg1_stride ? (unsigned)row_pix1[lrow + 32 * j] : pix;
        a_off1[j] = (pix1 * (unsigned)p.seg[1].cstride + (unsigned)(p.seg[1].coff + lcol)) * 4u;
        unsigned m = 0;
        if (base >= 0)
            for (int t = 0; t < p.ntaps; ++t) {
                const int iy = iy0 + (int)p.dy[t], ix = ix0 + (int)p.dx[t];
                if ((unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) m |= 1u << t;
            }
        a_mask[j] = m;
    }
    unsigned b_off[B_PASSES];
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) b_off[j] = ((unsigned)(n0 + lrow + 32 * j) * (unsigned)p.K + (unsigned)lcol) * 4u;
    const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.seg[0].ptr, 0, p.seg_bytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.seg[1].ptr ? p.seg[1].ptr : p.seg[0].ptr), 0, p.seg_bytes[1], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)gw, 0, p.w_bytes, 0x00020000);

    // ---- split-K range
    const int ks_per = (p.ksteps + p.ksplit - 1) / p.ksplit;
    const int ks0 = blockIdx.y * ks_per;
    const int ks1 = min(p.ksteps, ks0 + ks_per);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[A_PASSES], rb[B_PASSES];

    // (tap, chunk) of a K-step are carried incrementally (no division in the loop).
    auto gload = [&](int ks, int tap, int chunk) {
        const bool s1 = chunk >= p.seg0_chunks;                      // wave-uniform
        const int toff = __builtin_amdgcn_readfirstlane(
            (s_tap[tap] * (s1 ? p.seg[1].cstride : p.seg[0].cstride) + (s1 ? chunk - p.seg0_chunks : chunk) * IGEMM_BK) * 4);
        const unsigned bit = 1u << tap;
#pragma unroll
        for (int j = 0; j < A_PASSES; ++j) {
            const unsigned off = (a_mask[j] & bit) ? (s1 ? a_off1[j] : a_off0[j]) + (unsigned)toff : OOB;
            ra[j] = __builtin_bit_cast(f32x4, s1 ? __builtin_amdgcn_raw_buffer_load_b128(rs_a1, off, 0, 0)
                                                 : __builtin_amdgcn_raw_buffer_load_b128(rs_a0, off, 0, 0));
        }
        const int koff = ks * (IGEMM_BK * 4);                         // bytes along K, wave-uniform
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j) rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_off[j], koff, 0));
    };
    auto lstore = [&]() {
        float* As = smem;
        float* Bs = As + BM * LDS_LD;
        if (PREC == PREC_F16X3) {
            // row image [hi f16 x32 | lo f16 x32] (128 B): this thread owns k = lcol .. lcol+3
#pragma unroll
            for (int j = 0; j < A_PASSES; ++j) {
                const f32x4 v = ra[j];
                const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
                fp16x2 l01, l23;          // residuals are exact in fp32; round them to nearest
                l01[0] = (__fp16)(v[0] - (float)h01[0]); l01[1] = (__fp16)(v[1] - (float)h01[1]);
                l23[0] = (__fp16)(v[2] - (float)h23[0]); l23[1] = (__fp16)(v[3] - (float)h23[1]);
                char* row = reinterpret_cast<char*>(As + (lrow + 32 * j) * LDS_LD);
                *reinterpret_cast<uint2*>(row + lcol * 2) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
                *reinterpret_cast<uint2*>(row + 64 + lcol * 2) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
            }
        } else {
#pragma unroll
            for (int j = 0; j < A_PASSES; ++j)
                *reinterpret_cast<f32x4*>(As + (lrow + 32 * j) * LDS_LD + lcol) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < B_PASSES; ++j)      // weights: fp32 panel, or the pre-split panel with the same row image
            *reinterpret_cast<f32x4*>(Bs + (lrow + 32 * j) * LDS_LD + lcol) = rb[j];
    };

    int tap = ks0 / p.chunks_per_tap, chunk = ks0 - tap * p.chunks_per_tap;
    if (ks0 < ks1) {
        gload(ks0, tap, chunk);
        lstore();
    }
    __syncthreads();

    const float* As = smem + (wm * TM * 32 + li) * LDS_LD + lk * 4;
    const float* Bs = smem + BM * LDS_LD + (wn * TN * 32 + li) * LDS_LD + lk * 4;
    for (int ks = ks0; ks < ks1; ++ks) {
        const bool more = ks + 1 < ks1;
        if (++chunk == p.chunks_per_tap) { chunk = 0; ++tap; }
        if (more) gload(ks + 1, tap, chunk);   // global loads of the next K-step fly under this step's MFMAs
        if (PREC == PREC_F16X3) {
            // two 16-deep blocks; a lane's fragment = 8 consecutive k (16 B) of the hi or lo half-row:
            // byte offset 32*kb + 16*(lane>>5) (+64 for lo); As/Bs already carry the 16*(lane>>5) part
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(As + i * 32 * LDS_LD + kb * 8);
                    al[i] = *reinterpret_cast<const f16x8*>(As + i * 32 * LDS_LD + kb * 8 + 16);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[j] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * LDS_LD + kb * 8);
                    bl[j] = *reinterpret_cast<const f16x8*>(Bs + j * 32 * LDS_LD + kb * 8 + 16);
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
        } else
#pragma unroll
        for (int kk = 0; kk < IGEMM_BK; kk += 8) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDS_LD + kk);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDS_LD + kk);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                  // everyone is done reading the staging buffer
        if (more) {
            lstore();
            __syncthreads();
        }
    }

    // ---- epilogue.  The accumulators are transposed through LDS (the staging buffer is free after
    //      the last barrier), TM*32 pixel rows at a time (the rows of the waves with wm == h), so
    //      that every thread then owns 4 consecutive channels of one pixel: scale/shift, the
    //      residual and the output move as float4 and a 128-channel pixel row is one contiguous
    //      512-byte store.  (Lane-per-channel dword stores left the 1x1 ResNet-front layers
    //      epilogue-bound at ~1.2 TB/s.)
    //      C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    constexpr int CLD = BN + 4;
    constexpr int TPR = BN / 4;            // threads per pixel row
    constexpr int RPP = 256 / TPR;         // rows per sweep of the workgroup
    constexpr int PROWS = TM * 32;         // rows per pass
    float* Cs = smem;
    const int c4 = (tid % TPR) * 4;
    const int col = n0 + c4;
    const int r0 = tid / TPR;
    const bool cok = col < p.Cout;         // Cout is a multiple of 4 on this path
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (cok && p.ksplit <= 1) {
        if (gscale) sc = *reinterpret_cast<const f32x4*>(gscale + col);
        if (gshift) sh = *reinterpret_cast<const f32x4*>(gshift + col);
    }
    float amax = 0.f;      // operand-range guard (kernels.h)
#pragma unroll 1
    for (int h = 0; h < WGM; ++h) {
#ifndef P2P_ABL_CS      // (ablation builds: what the transpose through LDS costs the HBM-bound 1x1 layers)
        if (wm == h) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Cs[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CLD + (wn * TN + j) * 32 + li] = acc[i][j][r];
        }
        __syncthreads();
#endif
        if (cok) {
            constexpr int NIT = PROWS / RPP;
            if (p.ksplit > 1) {
#pragma unroll 2
                for (int it = 0; it < NIT; ++it) {
                    const int r = r0 + it * RPP;
                    const int m = m0 + h * PROWS + r;
                    if (m < m_end)
                        *reinterpret_cast<f32x4*>(p.partial + ((size_t)blockIdx.y * p.M + m) * p.Cout + col) = *reinterpret_cast<const f32x4*>(Cs + r * CLD + c4);
                }
            } else {
                // all residual loads of the pass are issued before the first store: the 1x1 residual layers are
                // HBM-bound and a load -> add -> store chain per row left only two loads in flight per thread
                int ops[NIT];
                f32x4 rs[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    ops[it] = row_out[h * PROWS + r0 + it * RPP];
                    rs[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#ifndef P2P_ABL_RES
                if (p.residual)
#else
                if (false)
#endif
                {
#pragma unroll
                    for (int it = 0; it < NIT; ++it)
#ifdef P2P_NT_LOAD
                        if (ops[it] >= 0) rs[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.residual + (size_t)ops[it] * p.res_cstride + col));
#else
                        if (ops[it] >= 0) rs[it] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)ops[it] * p.res_cstride + col);
#endif
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int op = ops[it];
                    if (op < 0) continue;
                    f32x4 v = *reinterpret_cast<const f32x4*>(Cs + (r0 + it * RPP) * CLD + c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
                    if (p.mode == EPI_HEAD) {
                        // col = phase*4 + ch: this thread holds (x, y, z, prob) of output pixel (2gy+py, 2gx+px)
                        const int ph = col >> 2;
                        f32x4 o;
                        o[0] = tanhf(v[0]); o[1] = tanhf(v[1]); o[2] = tanhf(v[2]);
                        o[3] = 1.f / (1.f + __expf(-v[3]));
                        *reinterpret_cast<f32x4*>(p.out + (size_t)(op + (ph >> 1) * p.Wout + (ph & 1)) * 4) = o;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rs[it][e];
                        if (p.act == ACT_RELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = relu_nan(v[e]);
                        } else if (p.act == ACT_LEAKY) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.alpha;
                        }
                        amax = range_note4(amax, v);
#ifdef P2P_NT_STORE
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p.out + (size_t)op * p.out_cstride + p.out_coff + col));
#else
                        *reinterpret_cast<f32x4*>(p.out + (size_t)op * p.out_cstride + p.out_coff + col) = v;
#endif
                    }
                }
            }
        }
#ifndef P2P_ABL_CS
        if (h + 1 < WGM) __syncthreads();
#endif
    }
    range_commit(p.range_acc, amax);
}

template <int WGM, int WGN, int TM, int TN, int PREC>
static hipError_t launch_cfg(const IgemmParams& p, hipStream_t s)
{
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    const int m_tiles = p.n_groups > 1 ? p.grp[p.n_groups].tile0 : (p.M + BM - 1) / BM;
    const int tiles = m_tiles * ((p.Cout + BN - 1) / BN);
    dim3 grid(tiles, p.ksplit > 1 ? p.ksplit : 1);
    hipLaunchKernelGGL((igemm_kernel<WGM, WGN, TM, TN, PREC>), grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

int igemm_tile_m(int) { return 128; }

hipError_t launch_igemm(const IgemmParams& p, int cfg, hipStream_t s)
{
    switch (cfg) {
    case 0: return p.prec == PREC_F16X3 ? launch_cfg<2, 2, 2, 2, PREC_F16X3>(p, s) : launch_cfg<2, 2, 2, 2, PREC_F32>(p, s);   // 128 x 128
    case 1: return p.prec == PREC_F16X3 ? launch_cfg<2, 2, 2, 1, PREC_F16X3>(p, s) : launch_cfg<2, 2, 2, 1, PREC_F32>(p, s);   // 128 x 64
    case 2: return p.prec == PREC_F16X3 ? launch_cfg<4, 1, 1, 1, PREC_F16X3>(p, s) : launch_cfg<4, 1, 1, 1, PREC_F32>(p, s);   // 128 x 32
    default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int ksplit, int M, int Cout,
                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                     int act, float alpha, float* __restrict__ out, unsigned* __restrict__ range_acc)
{
    const size_t total = (size_t)M * Cout;
    float amax = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < ksplit; ++z) s += partial[(size_t)z * total + i];
        const int c = (int)(i % Cout);
        float v = fmaf(s, scale ? scale[c] : 1.f, shift ? shift[c] : 0.f);
        if (act == ACT_RELU) v = relu_nan(v);
        else if (act == ACT_LEAKY) v = v > 0.f ? v : v * alpha;
        amax = range_note1(amax, v);
        out[i] = v;
    }
    range_commit(range_acc, amax);
}

hipError_t launch_splitk_reduce(const float* partial, int ksplit, int M, int Cout, const float* scale,
                                const float* shift, int act, float alpha, float* out, unsigned* range_acc, hipStream_t s)
{
    const size_t total = (size_t)M * Cout;
    const int blocks = (int)min((size_t)2048, (total + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, partial, ksplit, M, Cout, scale,
                       shift, act, alpha, out, range_acc);
    return hipGetLastError();
}

// Mixed-object batches: row m belongs to the object g with G.start[g] <= m < G.start[g + 1] and takes that object's scale / shift (the
// per-object bias of dense_enc).  ONE launch for the batch (a launch per object cost 13 us each: 0.4 ms per 30-object pass); the sum
// order over the K slabs and the epilogue expression are those of splitk_reduce_kernel, so the bits are the same.
__global__ __launch_bounds__(256) void splitk_reduce_groups_kernel(const float* __restrict__ partial, int ksplit, int M, int Cout, const Conv1Groups G,
                                                                    int act, float alpha, float* __restrict__ out, unsigned* __restrict__ range_acc)
{
    const int m = blockIdx.x;
    int g = 0;
    while (g + 1 < G.n_groups && G.start[g + 1] <= m) ++g;
    const float* __restrict__ scale = G.scale[g];
    const float* __restrict__ shift = G.shift[g];
    const size_t slab = (size_t)M * Cout;
    float amax = 0.f;
    for (int c = threadIdx.x; c < Cout; c += 256) {
        const size_t o = (size_t)m * Cout + c;
        float s = 0.f;
        for (int z = 0; z < ksplit; ++z) s += partial[(size_t)z * slab + o];
        float v = fmaf(s, scale ? scale[c] : 1.f, shift ? shift[c] : 0.f);
        if (act == ACT_RELU) v = relu_nan(v);
        else if (act == ACT_LEAKY) v = v > 0.f ? v : v * alpha;
        amax = range_note1(amax, v);
        out[o] = v;
    }
    range_commit(range_acc, amax);
}

hipError_t launch_splitk_reduce_groups(const float* partial, int ksplit, int M, int Cout, const Conv1Groups& G, int act, float alpha, float* out,
                                       unsigned* range_acc, hipStream_t s)
{
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(splitk_reduce_groups_kernel, dim3(M), dim3(256), 0, s, partial, ksplit, M, Cout, G, act, alpha, out, range_acc);
    return hipGetLastError();
}

}  // namespace p2p
