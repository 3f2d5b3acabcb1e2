// Winograd / Cook-Toom minimal filtering F(4, 5) ALONG THE ROW AXIS for the three 5x5 stride-1 'SAME' decoder layers (gfx950, PREC_F16X3).
//
// deconv1 / deconv2 / deconv3 (reference pix2pose_model/ae_model.py:207-211,217-220,227-230) are 67 % of the generator's MACs, and the
// direct kernel (igemm_halo.hip) already runs them at 0.82 of the power wall of their MFMA stream: the only lever left is fewer products.
// Four outputs of a row need 8 products per (vertical tap, channel) instead of 20:
//
//     y[4t + i] = sum_j AT[i][j] * ( sum_{ky, c} V_j[y + ky - 2][t][c] * U_j[ky][c][co] ),   V_j = sum_p BT[j][p] x[4t - 2 + p],
//                                                                                              U_j = sum_kx G[j][kx] w[ky][kx]
//
// with the points {0, +-1, +-2, +-1/2, inf}.  The vertical axis stays direct (5 taps): the eight positions j are eight independent
// GEMMs [(y, t) x (ky, c)] x [(ky, c) x co] -- 2.5x fewer MFMA products than the direct form.  Arithmetic: the transforms in fp32, the
// 22-bit hi/lo f16 split AFTER the transform (both operands), three products per block, fp32 accumulation, fp32 inverse transform.
// profiles/r06_wino_error_study.json is the error study behind the choice (network output 3.2e-5 from the fp64 graph; the direct form 7.5e-6; the 2-D
// forms F(2x2,5x5) / F(4x4,5x5) 5.6e-5 / 9.9e-5 -- and their 36 / 64 accumulator sets per tile cannot be fused).
//
// Two kernels:
//   wino_input_kernel   x (two channel segments: the skip concatenation is never materialised) -> V in HBM, already split, in the
//                       plane layout the GEMM kernel's LDS image has: [n][patch column][16-channel slice][plane = (j, hi/lo, k half)]
//                       [row][tile 0..3][8 halves].  A fragment read of the GEMM kernel is then 32 consecutive 16-byte slots of a plane:
//                       conflict-free without padding, and a vertical tap is a constant byte shift (64 B per row).
//   wino_gemm_kernel    512 threads = 8 waves, wave j owns position j: 128 (y, t) pairs (32 rows x 16 columns of one sample, or two
//                       16 x 16 samples) x 64 output channels = 128 accumulator registers.  Per K-step (one vertical tap of a
//                       16-channel slice): 8 V fragments from LDS, 4 U fragments STRAIGHT FROM GLOBAL in fragment order (no wave shares
//                       another wave's weights, so an LDS round trip would buy nothing; the panel of one 64-channel tile is 1.3-2.6 MB
//                       and stays in the L2 of the XCDs that work on that tile), 24 MFMAs.  The V image of the next slice is copied
//                       global -> registers -> LDS two pieces per K-step into the second buffer: one barrier per slice.  Epilogue:
//                       the eight positions meet in LDS (four passes of 32 pairs), inverse transform, BN + LeakyReLU, 256-byte stores.
#include "kernels.h"

namespace p2p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr unsigned OOB = 0xFFFFFFF0u;

// Workgroup barrier for LDS traffic only: __syncthreads() carries a release fence, which on gfx950 is s_waitcnt vmcnt(0) -- every global
// STORE of the epilogue would have to reach L2 before the next exchange pass (and, in the persistent loop, before the next tile) could start.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------------------------------------------------------------------------
// input transform.  Thread = (row of an 8-row block, tile t of the 16-column patch, channel quad of a 32-channel group); quads fastest, so
// a load instruction reads whole 128-byte pixel records and a store instruction writes 128-byte runs (two rows x four tiles) of 4 planes.
// ------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoParams p)
{
    const int tid = threadIdx.x;
    const int quad = tid & 7, t = (tid >> 3) & 3, r = tid >> 5;
    int b = blockIdx.x;
    const int rblocks = p.H >> 3;
    const int rb = b % rblocks; b /= rblocks;
    const int cgroups = p.Cin >> 5;
    const int cg = b % cgroups; b /= cgroups;
    const int PC = p.W >> 4;
    const int pc = b % PC;
    const int n = b / PC;
    const int y = rb * 8 + r;

    const bool s1 = cg >= p.seg0_groups;               // block-uniform
    const IgemmSeg sg = s1 ? p.seg[1] : p.seg[0];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sg.ptr, 0, s1 ? p.seg_bytes[1] : p.seg_bytes[0], 0x00020000);
    const unsigned c0 = (unsigned)(sg.coff + (s1 ? cg - p.seg0_groups : cg) * 32 + quad * 4);
    const int x0 = pc * 16 + t * 4 - 2;
    const unsigned rowpix = (unsigned)((n * p.H + y) * p.W);
    f32x4 d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int x = x0 + k;
        const unsigned off = (unsigned)x < (unsigned)p.W ? ((rowpix + (unsigned)x) * (unsigned)sg.cstride + c0) * 4u : OOB;
        d[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
    f32x4 v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float d0 = d[0][e], d1 = d[1][e], d2 = d[2][e], d3 = d[3][e], d4 = d[4][e], d5 = d[5][e], d6 = d[6][e], d7 = d[7][e];
        // BT of F(4,5) at {0, 1, -1, 2, -2, 1/2, -1/2, inf}; every coefficient is exact in binary, the order of operations is fixed
        const float a12 = __builtin_fmaf(-4.25f, d4, d2 + d6), b12 = __builtin_fmaf(-4.25f, d3, d1 + d5);
        const float a34 = __builtin_fmaf(-1.25f, d4, __builtin_fmaf(0.25f, d2, d6)), b34 = __builtin_fmaf(-2.5f, d3, __builtin_fmaf(0.5f, d1, 2.f * d5));
        const float a56 = __builtin_fmaf(-5.f, d4, __builtin_fmaf(4.f, d2, d6)), b56 = __builtin_fmaf(-2.5f, d3, __builtin_fmaf(2.f, d1, 0.5f * d5));
        v[0][e] = __builtin_fmaf(5.25f, d2 - d4, d6 - d0);
        v[1][e] = a12 + b12; v[2][e] = a12 - b12;
        v[3][e] = a34 + b34; v[4][e] = a34 - b34;
        v[5][e] = a56 + b56; v[6][e] = a56 - b56;
        v[7][e] = __builtin_fmaf(5.25f, d3 - d5, d7 - d1);
    }
    // plane (j, hl, lk) of slice cg * 2 + (quad >> 2); this thread holds halves [4 (quad & 1), +4) of the 16-byte piece (row y, tile t)
    const int S = p.Cin >> 4;
    const int slice = cg * 2 + (quad >> 2), lk = (quad >> 1) & 1;
    const size_t plane_bytes = (size_t)p.H * 64;
    char* dst = reinterpret_cast<char*>(p.V) + ((((size_t)n * PC + pc) * S + slice) * 32 + lk) * plane_bytes + (size_t)y * 64 + t * 16 + (quad & 1) * 8;
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f32x4 w = v[j];
        amax = range_note4(amax, w);
        const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(w[0], w[1]), h23 = __builtin_amdgcn_cvt_pkrtz(w[2], w[3]);
        fp16x2 l01, l23;              // residuals are exact in fp32; round them to nearest
        l01[0] = (__fp16)(w[0] - (float)h01[0]); l01[1] = (__fp16)(w[1] - (float)h01[1]);
        l23[0] = (__fp16)(w[2] - (float)h23[0]); l23[1] = (__fp16)(w[3] - (float)h23[1]);
        *reinterpret_cast<uint2*>(dst + (size_t)(j * 4) * plane_bytes) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
        *reinterpret_cast<uint2*>(dst + (size_t)(j * 4 + 2) * plane_bytes) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
    }
    range_commit(p.range_acc, amax);       // the transformed operand is what the split sees: up to 15x the activation
}

// ------------------------------------------------------------------------------------------------------------------------------------
// the eight position GEMMs + inverse transform + epilogue
// ------------------------------------------------------------------------------------------------------------------------------------
// LDS image of one 16-channel slice: 32 planes [R rows][4 tiles][16 B].  Single: R = 36 = rows y0 - 2 .. y0 + 33 of one sample.
// DUAL (16x16 grids: two samples per workgroup): R = 38 = [2 zero rows | sample A's 16 | 2 zero rows | sample B's 16 | 2 zero rows].
// m-tile i = 8 output rows x 4 tiles; its tap ky reads image rows base_i + ky + (0..7): base = 8 i, or {0, 8, 18, 26}.
// Wave j reads only the four planes of ITS position -- so it also stages them itself (9-10 pieces of 1 KB per slice, two per K-step,
// each written to LDS two K-steps after its load was issued: V comes from HBM): the K loop has no workgroup barrier at all, the eight
// waves drift freely.  The kernel is persistent (one workgroup per CU walks the tiles): the output stores of a tile drain behind the next
// tile's K loop instead of the whole chip storing in lockstep.
template <bool DUAL>
__global__ __launch_bounds__(512, 2) void wino_gemm_kernel(const WinoParams p)
{
    constexpr int R = DUAL ? 38 : 36;
    constexpr int PLANE = R * 64;
    constexpr int BUF = 32 * PLANE;
    constexpr int XLD = 68;                         // exchange image: [position 8][pair 32][64 channels + 4] floats
    constexpr int XBUF = 8 * 32 * XLD * 4;
    static_assert(2 * XBUF <= 2 * BUF, "exchange image fits the two slice buffers");
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int j = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = position (kept in an SGPR: it selects the weight stream's descriptor)
    const int li = lane & 31, lk = lane >> 5;

    const int S = p.Cin >> 4;
    const int PC = p.W >> 4;
    const int RB = DUAL ? 1 : p.H >> 5;
    const int NT = p.Cout >> 6;
    const int units = DUAL ? (p.n_groups > 1 ? p.grp[p.n_groups].unit0 : (p.N + 1) >> 1) : p.N;
    const int KS = p.ksplit > 1 ? p.ksplit : 1;       // split K: tile = (unit, patch, channel tile, range of S / KS channel slices)
    const int Sl = S / KS;                            // (even: launch_wino_gemm)
    const int ntiles = units * PC * RB * NT * KS;
    const unsigned slice_bytes = 32u * (unsigned)p.H * 64u;
    const size_t unit_block = (size_t)S * slice_bytes;           // bytes of one (sample, patch column)

    // XCD-aware order (block b runs on XCD b % 8): every sweep of gridDim.x tiles is cut into contiguous runs per XCD, channel tile fastest:
    // the workgroups that share a V patch (one per channel tile) run on one XCD at the same time
    int tl0;
    {
        const int nblk = gridDim.x, b = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = b & 7, idx = b >> 3;
        tl0 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int pair = tid >> 4, cq = tid & 15;       // epilogue role
    float amax = 0.f;

    for (int tl = tl0; tl < ntiles; tl += gridDim.x) {
    const int ks = tl % KS;
    const int s_begin = ks * Sl;
    const int ntile = (tl / KS) % NT;
    int rest = tl / (KS * NT);
    const int rb = rest % RB; rest /= RB;
    const int pc = rest % PC;
    const int unit = rest / PC;
    int n0 = DUAL ? unit * 2 : unit;
    int n_end = p.N;                                 // first sample that is not this object's
    const int y0 = rb * 32;

    const float* gu = p.U;
    const float* gscale = p.scale;
    const float* gshift = p.shift;
    if (p.n_groups > 1) {                            // groups are runs of samples; DUAL: every group is paired up on its own (unit0)
        int g = 0;
        if (DUAL) {
            while (g + 1 < p.n_groups && p.grp[g + 1].unit0 <= unit) ++g;
            n0 = p.grp[g].sample0 + 2 * (unit - p.grp[g].unit0);
            n_end = p.grp[g + 1].sample0;
        } else {
            while (g + 1 < p.n_groups && p.grp[g + 1].sample0 <= n0) ++g;
        }
        gu = p.grp[g].U; gscale = p.grp[g].scale; gshift = p.grp[g].shift;
    }
    const bool has_b = DUAL && n0 + 1 < n_end;

    // ---- V: global -> registers -> LDS, this wave's four planes only, in 1 KB pieces (16 rows x 4 tiles of one plane per wave-instruction).
    //      Single: piece q < 8 = rows [16 (q & 1), +16) of plane q >> 1; piece 8 = the last four rows of all four planes (16 lanes each).
    //      DUAL:   piece q < 8 = sample q & 1 of plane q >> 1 (the zero rows between and around the samples are written once per tile).
    //      Per lane only the validity of border rows differs: three address registers (one for DUAL), everything else is scalar.
    const char* vbase = reinterpret_cast<const char*>(p.V) + ((size_t)n0 * PC + pc) * unit_block;
    const unsigned vrange = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(has_b ? 2 * unit_block : unit_block));     // (kept scalar: a vector select here makes every load a waterfall loop)
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, vrange, 0x00020000);
    const unsigned plane_bytes = (unsigned)p.H * 64u;
    unsigned vo_a, vo_b, vo_c;
    if (DUAL) {
        vo_a = vo_b = vo_c = (unsigned)lane * 16u;
    } else {
        const int row = lane >> 2, t = lane & 3;
        vo_b = (unsigned)((y0 + 14 + row) * 64 + t * 16);                                   // rows y0 + 14 .. y0 + 29: always inside
        vo_a = y0 - 2 + row >= 0 ? vo_b - 1024u : OOB;                                      // rows y0 - 2 .. y0 + 13
        const int yc = y0 + 30 + ((lane & 15) >> 2);                                        // rows y0 + 30 .. y0 + 33 of plane lane >> 4
        vo_c = yc < p.H ? (unsigned)(lane >> 4) * plane_bytes + (unsigned)(yc * 64 + t * 16) : OOB;
    }
    char* wreg = smem + j * 4 * PLANE + lane * 16;                                          // rows 0..15 of the wave's first plane, buffer 0
    char* wreg_c = smem + (j * 4 + (lane >> 4)) * PLANE + 32 * 64 + (lane & 15) * 16;       // single: the quarter piece
    f32x4 rv[2][2];                                                                         // two groups of two pieces in flight
    auto vload1 = [&](f32x4& dst, int slice, int q, bool on) {     // `on` is wave-uniform: off = out of range = no traffic
#ifdef P2P_ABL_WV
        on = false;
#endif
        unsigned so = (unsigned)(s_begin + slice) * slice_bytes + (unsigned)(j * 4) * plane_bytes;
        unsigned vo;
        if (DUAL) {
            if (q >= 8) return;
            so += (unsigned)(q >> 1) * plane_bytes + ((q & 1) ? (unsigned)unit_block : 0u);
            vo = vo_a;
            if ((q & 1) && !has_b) on = false;
        } else if (q < 8) {
            so += (unsigned)(q >> 1) * plane_bytes;
            vo = (q & 1) ? vo_b : vo_a;
        } else if (q == 8) {
            vo = vo_c;
        } else return;
        dst = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo | (on ? 0u : OOB), so, 0));
    };
    auto vstore1 = [&](const f32x4& src, int buf, int q) {
        if (DUAL) {
            if (q < 8) *reinterpret_cast<f32x4*>(wreg + buf * BUF + (q >> 1) * PLANE + ((q & 1) ? 20 : 2) * 64) = src;
        } else if (q < 8) {
            *reinterpret_cast<f32x4*>(wreg + buf * BUF + (q >> 1) * PLANE + (q & 1) * 1024) = src;
        } else if (q == 8) {
            *reinterpret_cast<f32x4*>(wreg_c + buf * BUF) = src;
        }
    };
    auto vload = [&](int set, int slice, int g, bool on) { vload1(rv[set][0], slice, 2 * g, on); vload1(rv[set][1], slice, 2 * g + 1, on); };
    auto vstore = [&](int set, int buf, int g) { vstore1(rv[set][0], buf, 2 * g); vstore1(rv[set][1], buf, 2 * g + 1); };

    // ---- U: this wave's stream (channel tile, position j): K-step kb = 4 fragments of 1 KB, contiguous.  The panel carries one K-step
    //      of padding behind its last stream, so the load one K-step ahead needs no condition.
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(gu) + ((size_t)(ntile * 8 + j) * S * 5 + (size_t)s_begin * 5) * 4096), 0,
                                                                          (unsigned)((Sl * 5 + 1) * 4096), 0x00020000);
    const unsigned uoff = (unsigned)lane * 16u;
    f16x8 u[2][4];                                   // (tile 0 hi, tile 0 lo, tile 1 hi, tile 1 lo) of the even / odd K-steps
    auto uload = [&](int set, int kb) {
#pragma unroll
        for (int f = 0; f < 4; ++f) u[set][f] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_u, uoff + f * 1024, kb * 4096, 0));
    };


    // V fragment of m-tile i, tap ky, half hl: plane (j, hl, lk), image row base_i + ky + (li >> 2), tile li & 3
    const char* img0 = smem + (j * 4 + lk) * PLANE + (li >> 2) * 64 + (li & 3) * 16;

    lds_barrier();                                   // (persistent loop) the previous tile's exchange image has been read
    if (DUAL) {
        // the six zero rows of this wave's planes in both buffers (the exchange image overwrites them every tile): 2 x 4 x 6 rows x 4 slots
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int sl = lane + 64 * k;                  // 0..191: (buffer, plane, zero row 0..5, slot)
            const int bf = sl / 96, r2 = sl - bf * 96;
            const int pl = r2 / 24, r3 = r2 - pl * 24;
            const int zr = r3 >> 2, t = r3 & 3;
            const int row = zr < 2 ? zr : (zr < 4 ? 16 + zr : 32 + zr);     // 0, 1, 18, 19, 36, 37
            *reinterpret_cast<f32x4*>(smem + bf * BUF + (j * 4 + pl) * PLANE + row * 64 + t * 16) = z;
        }
    }
    // prologue: slice 0 -> buffer 0; the first two groups of slice 1 on their way.  All ten pieces of slice 0 are loaded in ONE round trip
    // (the accumulators are not live yet: the registers are there) -- through the two staging sets it took three, with every wave of the CU
    // waiting in lockstep behind the tile's top barrier.
#ifndef P2P_WINO_SERIAL_PROLOGUE
    {
        f32x4 pv[5][2];
#pragma unroll
        for (int g = 0; g < 5; ++g) { vload1(pv[g][0], 0, 2 * g, true); vload1(pv[g][1], 0, 2 * g + 1, true); }
        uload(0, 0);
#pragma unroll
        for (int g = 0; g < 5; ++g) { vstore1(pv[g][0], 0, 2 * g); vstore1(pv[g][1], 0, 2 * g + 1); }
    }
#else
#pragma unroll
    for (int g = 0; g < 5; ++g) { vload(g & 1, 0, g, true); vstore(g & 1, 0, g); }
    uload(0, 0);
#endif
    vload(0, 1, 0, true);                            // S is even (Cin % 32 == 0)
    vload(1, 1, 1, true);

    f32x16 acc[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][i][r] = 0.f;

    // Two slices (10 K-steps) per iteration: buffer, weight register set and staging register set of every K-step are compile-time.
    // K-step (s, ky): write group ky of slice s + 1 (loaded two K-steps ago), issue the load of the group two K-steps ahead.
    for (int s2 = 0; s2 < Sl; s2 += 2) {
#pragma unroll
        for (int kk = 0; kk < 10; ++kk) {
            const int half = kk / 5, ky = kk % 5;
            const int s = s2 + half;
#ifndef P2P_ABL_WU
            uload((kk + 1) & 1, s * 5 + ky + 1);
#else
            if (s2 == 0) uload((kk + 1) & 1, s * 5 + ky + 1);
#endif
            vstore(kk & 1, half ^ 1, ky);                        // (a slice past the last one: zeros nobody reads)
            if (ky < 3) vload(kk & 1, s + 1, ky + 2, half == 0 || s + 1 < Sl);
            else vload(kk & 1, s + 2, ky - 3, s + 2 < Sl);
            __builtin_amdgcn_sched_barrier(0);                   // the loads above stay AHEAD of this K-step's matrix work (the scheduler would sink them to their uses)
            const char* img = img0 + half * BUF;
            f16x8 vh[4], vl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (DUAL ? (i < 2 ? 8 * i : 8 * i + 2) : 8 * i) + ky;
                vh[i] = *reinterpret_cast<const f16x8*>(img + row * 64);
                vl[i] = *reinterpret_cast<const f16x8*>(img + row * 64 + 2 * PLANE);
            }
            const f16x8* uc = u[kk & 1];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uc[2 * c + 1], vh[i], acc[c][i], 0, 0, 0);
                    acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uc[2 * c], vl[i], acc[c][i], 0, 0, 0);
                    acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uc[2 * c], vh[i], acc[c][i], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    lds_barrier();                                   // every wave is done with its planes: the exchange image may overwrite them

    // ---- epilogue.  C/D layout of the 32x32 MFMA with U as the A operand: row = channel (r & 3) + 8 (r >> 2) + 4 lk of the 32-tile,
    //      column li = pair.  Pass i: the eight waves put m-tile i into the exchange image (a lane writes 4 consecutive channels),
    //      then thread (pair = tid >> 4, channel quad = tid & 15) combines the eight positions into four output pixels.
    const int col = ntile * 64 + cq * 4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (gscale) sc = *reinterpret_cast<const f32x4*>(gscale + col);
    if (gshift) sh = *reinterpret_cast<const f32x4*>(gshift + col);
#ifdef P2P_ABL_WEPI
    {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[c][i][r];
        if (t == 123.456f) p.out[tid] = t;
        continue;
    }
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* X = reinterpret_cast<float*>(smem + (i & 1) * XBUF);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc[c][i][4 * q], acc[c][i][4 * q + 1], acc[c][i][4 * q + 2], acc[c][i][4 * q + 3]};
                *reinterpret_cast<f32x4*>(X + (j * 32 + li) * XLD + c * 32 + 8 * q + 4 * lk) = v;
            }
        lds_barrier();
        f32x4 m[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) m[jj] = *reinterpret_cast<const f32x4*>(X + (jj * 32 + pair) * XLD + cq * 4);
        int n, y;
        if (DUAL) { n = n0 + (i >> 1); y = (i & 1) * 8 + (pair >> 2); }
        else { n = n0; y = y0 + i * 8 + (pair >> 2); }
        if (!DUAL || n == n0 || has_b) {
            const size_t pix = ((size_t)n * p.H + y) * p.W + pc * 16 + (pair & 3) * 4;
            float* o = p.out + pix * p.out_cstride + p.out_coff + col;
            f32x4 yv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // AT of F(4,5): rows (1 1 1 1 1 1 1 0), (0 1 -1 2 -2 1/2 -1/2 0), (0 1 1 4 4 1/4 1/4 0), (0 1 -1 8 -8 1/8 -1/8 1)
                const float s12 = m[1][e] + m[2][e], d12 = m[1][e] - m[2][e];
                const float s34 = m[3][e] + m[4][e], d34 = m[3][e] - m[4][e];
                const float s56 = m[5][e] + m[6][e], d56 = m[5][e] - m[6][e];
                yv[0][e] = (m[0][e] + s12) + (s34 + s56);
                yv[1][e] = __builtin_fmaf(0.5f, d56, __builtin_fmaf(2.f, d34, d12));
                yv[2][e] = __builtin_fmaf(0.25f, s56, __builtin_fmaf(4.f, s34, s12));
                yv[3][e] = __builtin_fmaf(0.125f, d56, __builtin_fmaf(8.f, d34, d12)) + m[7][e];
            }
            if (KS > 1) {           // raw sums of this K range; scale / shift / activation belong to the reduction
                float* po = p.partial + ((size_t)ks * p.N * p.H * p.W + pix) * p.Cout + col;
#pragma unroll
                for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(po + (size_t)k * p.Cout) = yv[k];
                continue;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 v = yv[k];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(v[e], sc[e], sh[e]);
                if (p.act == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = relu_nan(v[e]);
                } else if (p.act == ACT_LEAKY) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.alpha;
                }
                amax = range_note4(amax, v);
#ifdef P2P_ABL_WST
                if (v[0] == 123.456f)
#endif
                *reinterpret_cast<f32x4*>(o + (size_t)k * p.out_cstride) = v;
            }
        }
    }
    }   // tiles
    range_commit(p.range_acc, amax);
}

}  // namespace

bool wino_supported(int H, int W, int Cin0, int Cin1, int Cout)
{
    if (Cin0 % 32 || Cin1 % 32 || Cout % 64 || W % 16) return false;
    return (H == 16 && W == 16) || H % 32 == 0;
}

// bytes of V for N samples
size_t wino_v_bytes(int N, int H, int W, int Cin) { return (size_t)N * H * W * Cin * 8; }

int wino_gemm_grid(const WinoParams& p)
{
    const bool dual = p.H == 16;
    const int units = dual ? (p.n_groups > 1 ? p.grp[p.n_groups].unit0 : (p.N + 1) / 2) : p.N * (p.H / 32);
    return units * (p.W / 16) * (p.Cout / 64) * (p.ksplit > 1 ? p.ksplit : 1);
}

hipError_t launch_wino_input(const WinoParams& p, hipStream_t s)
{
    const int grid = p.N * (p.W / 16) * (p.Cin / 32) * (p.H / 8);
    hipLaunchKernelGGL(wino_input_kernel, dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_wino_gemm(const WinoParams& p, hipStream_t s)
{
    // persistent: one workgroup per CU (the kernel's LDS image admits no second one) walks the tiles
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipGetLastError();
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (p.ksplit > 1 && ((p.Cin >> 4) % (2 * p.ksplit) || !p.partial || p.n_groups > 1)) return hipErrorInvalidValue;      // even slice ranges; one object
    const int tiles = wino_gemm_grid(p);
    const int grid = tiles < n_cu ? tiles : n_cu;
    if (p.H == 16) hipLaunchKernelGGL((wino_gemm_kernel<true>), dim3(grid), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((wino_gemm_kernel<false>), dim3(grid), dim3(512), 0, s, p);
    return hipGetLastError();
}

}  // namespace p2p
