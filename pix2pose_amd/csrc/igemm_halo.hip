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

// LDS images.
//   halo: one 144-byte record per pixel of the (TY + dy) x (16 + dx) patch: [hi x32 | lo x32 | 16 pad].  A wave's 32-row MFMA
//   fragment = 2 patch rows x 16 columns, and ds_read_b128 is served in 16-lane groups that MIX the two rows
//   ({0-3,12-15,20-27}, {4-11,16-19,28-31}: MI355X_MICROARCH.md, LDS): within a row the 9-slot record stride (a slot = 16 B,
//   16 slots per bank row) is conflict-free, and the two rows interleave without conflicts exactly when the ROW PITCH is a
//   multiple of 16 slots -- so every row is padded by `skew` slots, (9 * HPX + skew) % 16 == 0.  (A pitch of HPX records
//   made 4 of the 16 slots of every group 2-way for the 5x5 layers: SQ_LDS_BANK_CONFLICT = 25-34 % of the LDS cycles.)
//   A tap is still ONE constant byte shift of the whole fragment: dy * pitch + dx * 144.
//   weights: 128-byte rows [hi x32 | lo x32] without padding; the 16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 7),
//   which is conflict-free for the fragment reads (rows li, one chunk) and for the row-contiguous stores.
constexpr int REC = 144;
constexpr int TX = 16;                     // patch width; patch height TY = 4 * WGM rows (a wave owns 4 patch rows = 64 GEMM rows)
constexpr int MAX_PITCH = 3072;                // largest pitch over HPX = 16..20: 20 * 144 + 12 * 16 (HPX = 20)
constexpr int WREC = 128;

template <int WGM, int TN>
__global__ __launch_bounds__(256, WGM == 2 ? 3 : 2) void igemm_halo_kernel(const IgemmParams p)
{
    constexpr int WGN = 4 / WGM;
    constexpr int TY = 4 * WGM;
    constexpr int BN = WGN * TN * 32;
    constexpr int B_PASSES = BN / 32;
    constexpr int MAX_HALO = (TY + 4) * (TX + 4);
    constexpr int HALO_PASSES = (MAX_HALO * 8 + 255) / 256;    // float4 loads per thread per slice
    constexpr int HALO_BYTES = (TY + 4) * MAX_PITCH;
    constexpr int STAGE_BYTES = HALO_BYTES + BN * WREC;
    constexpr int CTILE_BYTES = 64 * (BN + 4) * 4;
    __shared__ __attribute__((aligned(16))) char smem[STAGE_BYTES > CTILE_BYTES ? STAGE_BYTES : CTILE_BYTES];
    __shared__ int s_tap[IGEMM_MAX_TAPS + 3];     // byte shift of a tap inside the halo image

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 31, lk = lane >> 5;

    // halo extents from the tap table (wave-uniform)
    int dy0 = 0, dy1 = 0, dx0 = 0, dx1 = 0;
    for (int t = 0; t < p.ntaps; ++t) {
        dy0 = min(dy0, (int)p.dy[t]); dy1 = max(dy1, (int)p.dy[t]);
        dx0 = min(dx0, (int)p.dx[t]); dx1 = max(dx1, (int)p.dx[t]);
    }
    const int HPX = TX + dx1 - dx0, HPY = TY + dy1 - dy0;
    const int halo_px = HPX * HPY;
    const int PITCH = HPX * REC + ((16 - (9 * HPX) % 16) % 16) * 16;      // row pitch: a multiple of 16 slots (256 B)

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

    if (tid < p.ntaps) s_tap[tid] = ((int)p.dy[tid] - dy0) * PITCH + ((int)p.dx[tid] - dx0) * REC;

    // ---- halo loader: float4 idx = tid + 256 j -> quad idx % 8 of halo pixel perm(idx / 8).  The ds_write_b64 stores are
    //      served in contiguous 16-lane groups over 32 banks: two pixels per group, which collide unless they are 4 records
    //      apart (4 * 36 dwords = 16 mod 32) -- so consecutive octets of lanes take pixels hp and hp + 4.
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned h_pix[HALO_PASSES];               // pixel index into the input tensor, OOB outside the image / the halo
    unsigned h_dst2[(HALO_PASSES + 1) / 2];    // LDS byte offsets, two 16-bit values per register (0xFFFF = not part of the halo)
#pragma unroll
    for (int j = 0; j < (HALO_PASSES + 1) / 2; ++j) h_dst2[j] = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < HALO_PASSES; ++j) {
        const int idx = tid + 256 * j;
        const int t8 = idx >> 3, q = idx & 7;
        const int hp = (t8 & ~7) | ((t8 & 1) << 2) | ((t8 >> 1) & 3);
        const int hy = hp / HPX, hx = hp - hy * HPX;
        const int iy = ty0 + dy0 + hy, ix = tx0 + dx0 + hx;
        const bool ok = hp < halo_px && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
        h_pix[j] = ok ? (unsigned)((n * p.Hin + iy) * p.Win + ix) : OOB;
        const unsigned dst = hp < halo_px ? (unsigned)(hy * PITCH + hx * REC + q * 8) : 0xFFFFu;
        h_dst2[j >> 1] = (j & 1) ? ((h_dst2[j >> 1] & 0x0000FFFFu) | (dst << 16)) : ((h_dst2[j >> 1] & 0xFFFF0000u) | dst);
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
    const int b_dst = lrow * WREC + (((tid & 7) ^ ((lrow >> 1) & 7)) << 4);          // (row + 32 j keeps (row >> 1) & 7)
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
    const int a_tile = 2 * PITCH;
    // B fragment (kb, half): chunk kb*2 + lk (+4 for lo) of row li, swizzled by (li >> 1) & 7
    const char* Bs = Bst + (wn * TN * 32 + li) * WREC;
    int b_sw[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) b_sw[kb][hf] = ((kb * 2 + lk + 4 * hf) ^ ((li >> 1) & 7)) << 4;

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
#ifndef P2P_ABL_B
        if (more) bload(ntap, nchunk);
#endif
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
#ifndef P2P_ABL_BAR
        __syncthreads();                  // everyone is done reading the weight tile (and, at a slice end, the halo)
#endif
        if (more) {
#ifndef P2P_ABL_B
            bstore();
#endif
#ifndef P2P_ABL_HSTORE
            if (nchunk != chunk) hstore();                       // next slice's halo (prefetched at the start of this one)
#endif
#ifndef P2P_ABL_BAR
            __syncthreads();
#endif
#ifndef P2P_ABL_HLOAD
            if (nchunk != chunk && nchunk + 1 < n_chunks) hload(nchunk + 1);
#endif
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
    float amax = 0.f;      // operand-range guard (kernels.h)
#pragma unroll 1
    for (int h = 0; h < WGM; ++h) {
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
                    for (int e = 0; e < 4; ++e) v[e] = relu_nan(v[e]);
                } else if (p.act == ACT_LEAKY) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.alpha;
                }
                amax = range_note4(amax, v);
                *reinterpret_cast<f32x4*>(p.out + (size_t)ops[it] * p.out_cstride + p.out_coff + col) = v;
            }
        }
        if (h + 1 < WGM) __syncthreads();
    }
    range_commit(p.range_acc, amax);
}

}  // namespace

// Cout % 128 == 0: 8x16 patch, 128x128 tile (4 waves as 2x2, 64x64 each).  Cout == 64 mod 128: the same 64x64 wave tile needs
// all four waves along M -- a 16x16 patch, 256x64 tile (with the 8x16 patch a wave had 64x32 and the kernel was bound by
// its LDS reads: 12 fragment loads per 12 MFMAs instead of 16 per 24) -- or the 8x16 patch with 64x32 wave tiles when the
// grid is not a multiple of 16 rows.
static bool tall_patch() { static const bool on = dev_env("P2P_HALO_TALL") == nullptr || atoi(dev_env("P2P_HALO_TALL")) != 0; return on; }

bool igemm_halo_supported(const IgemmParams& p)
{
    constexpr int TY = 8;
    if (p.prec != PREC_F16X3 || p.mode != EPI_NORMAL || p.ksplit > 1 || p.in_stride != 1 || p.ntaps < 4) return false;
    if (p.Hin != p.Hg || p.Win != p.Wg || p.Hg % TY || p.Wg % TX || p.Cout % 64) return false;
    int dy0 = 0, dy1 = 0, dx0 = 0, dx1 = 0;
    for (int t = 0; t < p.ntaps; ++t) {
        dy0 = std::min(dy0, (int)p.dy[t]); dy1 = std::max(dy1, (int)p.dy[t]);
        dx0 = std::min(dx0, (int)p.dx[t]); dx1 = std::max(dx1, (int)p.dx[t]);
    }
    return dy1 - dy0 <= 4 && dx1 - dx0 <= 4;
}

int igemm_halo_family(const IgemmParams& p)      // profile slot (p2p_mi355.h): 3 = 128x128 tiles, 4 = the Cout % 128 == 64 variants
{
    return p.Cout % 128 == 0 ? 3 : 4;
}

hipError_t launch_igemm_halo(const IgemmParams& p, hipStream_t s)
{
    const int m_tiles = p.N * (p.Hg / 8) * (p.Wg / TX);
    if (p.Cout % 128 == 0) {
        hipLaunchKernelGGL((igemm_halo_kernel<2, 2>), dim3(m_tiles * (p.Cout / 128)), dim3(256), 0, s, p);
    } else if (tall_patch() && p.Hg % 16 == 0) {
        hipLaunchKernelGGL((igemm_halo_kernel<4, 2>), dim3(m_tiles / 2 * (p.Cout / 64)), dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL((igemm_halo_kernel<2, 1>), dim3(m_tiles * (p.Cout / 64)), dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace p2p
