// Winograd / Cook-Toom minimal filtering F(4, 3) ALONG THE ROW AXIS for the stride-2 transposed convolutions up2 / up3 (gfx950, PREC_F16X3).
//
// Conv2DTranspose 5x5 / 2 'SAME' (reference pix2pose_model/ae_model.py:212-215,222-225) is four sub-pixel phases: phase
// (py, px) is a correlation ON THE INPUT GRID with (2 + py) x (2 + px) taps at offsets dy, dx in {-1, 0, +1}:
//
//     out[2y + py][2x + px] = sum_{dy, dx, c} x[y + dy][x + dx][c] * w[py + 1 - 2 dy][px + 1 - 2 dx][c][co]
//
// Along a row both column phases are 3-tap filters (the 2-tap one zero-extended), so ONE input transform serves all four phases:
//
//     out[2y + py][2 (4t + i) + px] = sum_j AT[i][j] * ( sum_{dy, c} V_j[y + dy][t][c] * U^{py,px}_j[dy][c][co] ),
//     V_j = sum_p BT[j][p] x[4t - 1 + p],   U_j = sum_dx G[j][dx + 1] w[..][px + 1 - 2 dx]          points {0, +-1, +-2, inf}
//
// 6 products per 4 outputs and vertical tap instead of 12 (8): 15 position-products per input pixel for the four phases instead of 25.
// Arithmetic as in wino.hip: transforms in fp32, the 22-bit hi/lo f16 split AFTER the transform, three MFMA products per block, fp32
// accumulation, fp32 inverse transform.  Error study: profiles/r06_wino_up_error_study.json (layer 8e-7 relative rms against 3.4e-7 of the
// direct phases; the network output moves from 3.2e-5 to 4.0e-5 of the fp64 graph in the worst weight family).
//
// Two kernels:
//   wino3_input_kernel  x -> V in HBM, split, plane layout [n][patch column][16-channel slice][plane = (j, hi/lo, k half)][row][tile 0..3][8 halves]
//                       (6 bytes per input element).
//   wino3_gemm_kernel   768 threads = 12 waves: six positions on four SIMDs need a multiple of four waves, and three waves per SIMD leave 168
//                       registers each.  Wave (j, mh) owns position j of one of the tile's two 16-row units and all 64 output channels
//                       (2 x 2 x 16 accumulators), stages its unit's rows of its position's planes itself (no workgroup barrier in the K
//                       loop) and reads, per K-step (one vertical tap of a 16-channel slice), 4 V fragments from LDS and 4 U fragments
//                       straight from global for 12 MFMAs.  A tile = (patch, phase, 64-channel tile); a phase with py = 0 walks 2 K-steps
//                       per slice, py = 1 three -- the phase a workgroup serves rotates with the sweep so that every CU gets the same mix.
//                       (First form, measured equal: wave = (position, 32-channel half) sharing the planes of a position through a
//                       workgroup barrier per slice; timing ablations of both: tools/experiments/README.md.)
#include "kernels.h"

#include <type_traits>

namespace p2p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

#ifdef P2P_W3_STAMPS      // A/B builds: 100 MHz time stamps of the phases of a workgroup's tiles (tools/w3_stamps.py)
__device__ unsigned long long g_w3_stamps[8 * 16 * 16];
// slots 0..11: end of the K loop of wave 0..11; 12: tile start (after the top barrier), 13: K loop start, 14: after the end-of-K barrier (wave 0)
#define W3_STAMP(k) do { if (lane == 0 && (blockIdx.x & 31) == 0 && (blockIdx.x >> 5) < 8 && n_tile_seen < 16) g_w3_stamps[((blockIdx.x >> 5) * 16 + n_tile_seen) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define W3_STAMP(k) do {} while (0)
#endif

namespace {

constexpr unsigned OOB = 0xFFFFFFF0u;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------------------------------------------------------------------------
// input transform.  Thread = (row of an 8-row block, tile t of the 16-column patch, channel quad of a 32-channel group), quads fastest.
// ------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino3_input_kernel(const Wino3Params p)
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

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const unsigned c0 = (unsigned)(cg * 32 + quad * 4);
    const int x0 = pc * 16 + t * 4 - 1;
    const unsigned rowpix = (unsigned)((n * p.H + y) * p.W);
    f32x4 d[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int x = x0 + k;
        const unsigned off = (unsigned)x < (unsigned)p.W ? ((rowpix + (unsigned)x) * (unsigned)p.Cin + c0) * 4u : OOB;
        d[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
    f32x4 v[6];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float d0 = d[0][e], d1 = d[1][e], d2 = d[2][e], d3 = d[3][e], d4 = d[4][e], d5 = d[5][e];
        // BT of F(4,3) at {0, 1, -1, 2, -2, inf}: integer coefficients, the order of operations is fixed
        const float a12 = __builtin_fmaf(-4.f, d2, d4), b12 = __builtin_fmaf(-4.f, d1, d3);
        const float a34 = d4 - d2, b34 = 2.f * (d3 - d1);
        v[0][e] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
        v[1][e] = a12 + b12; v[2][e] = a12 - b12;
        v[3][e] = a34 + b34; v[4][e] = a34 - b34;
        v[5][e] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
    }
    const int S = p.Cin >> 4;
    const int slice = cg * 2 + (quad >> 2), lk = (quad >> 1) & 1;
    const size_t plane_bytes = (size_t)p.H * 64;
    char* dst = reinterpret_cast<char*>(p.V) + ((((size_t)n * PC + pc) * S + slice) * 24 + lk) * plane_bytes + (size_t)y * 64 + t * 16 + (quad & 1) * 8;
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const f32x4 w = v[j];
        amax = range_note4(amax, w);
        const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(w[0], w[1]), h23 = __builtin_amdgcn_cvt_pkrtz(w[2], w[3]);
        fp16x2 l01, l23;              // residuals are exact in fp32; round them to nearest
        l01[0] = (__fp16)(w[0] - (float)h01[0]); l01[1] = (__fp16)(w[1] - (float)h01[1]);
        l23[0] = (__fp16)(w[2] - (float)h23[0]); l23[1] = (__fp16)(w[3] - (float)h23[1]);
        *reinterpret_cast<uint2*>(dst + (size_t)(j * 4) * plane_bytes) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
        *reinterpret_cast<uint2*>(dst + (size_t)(j * 4 + 2) * plane_bytes) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
    }
    range_commit(p.range_acc, amax);       // the transformed operand is what the split sees: up to 10x the activation
}

// ------------------------------------------------------------------------------------------------------------------------------------
// the six position GEMMs + inverse transform + epilogue
// ------------------------------------------------------------------------------------------------------------------------------------
// Wave (j, mh) owns position j of one UNIT = 16 output rows x 16 columns of one sample (two m-tiles of 8 rows x 4 tiles) and all 64
// channels of the tile: 2 x 2 x 16 accumulator registers.  A workgroup tile = two units: the two 16-row halves of a 32-row block (H >= 32)
// or two samples (H = 16).  The wave stages ITS unit's rows of ITS position's four planes itself -- 18 rows (one halo row above and below:
// out-of-image rows load as zeros) x 4 tiles x 16 B per plane, double-buffered -- so the K loop has no workgroup barrier; per K-step
// (one vertical tap of a 16-channel slice) it reads 4 V fragments from LDS and 4 U fragments straight from global (4 KB per 12 MFMAs: the
// L2 traffic per MFMA of wino.hip) for 12 MFMAs.
// NU = units per workgroup: 2 (shipped) = one 12-wave workgroup per CU; 1 (A/B builds, -DP2P_W3_NU1) = 6-wave workgroups, two per CU
// (55 KB of LDS each), meant to interleave one workgroup's prologue / exchange / store phases with the other's K loop -- with 32-48
// K-steps per tile those phases are a quarter of a tile's time (MFMA busy 50 % for up2 / up3 against 67 % for conv4's 320-K-step tiles).
// Measured SLOWER (up2 + up3 + conv4 + up1: 1118 vs 1022 us per 256 inputs): six waves do not spread evenly over four SIMDs.
template <int NU>
__global__ __launch_bounds__(NU * 384, 2 / NU) void wino3_gemm_kernel(const Wino3Params p)
{
    constexpr int NTHR = NU * 384;
    constexpr int WPLANE = 18 * 64;                 // one plane of a wave's image
    constexpr int WIMG = 4 * WPLANE;                // the four planes (hl, lk) of its position
    constexpr int BUF = 6 * NU * WIMG;
    constexpr int XLD = 68;                         // exchange image: [position 6][pair 32][64 channels + 4] floats
    constexpr int XBUF = 6 * 32 * XLD * 4;
    static_assert(NU * XBUF <= 2 * BUF, "the exchange images fit the two slice buffers");
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = NU == 2 ? wv >> 1 : wv, mh = NU == 2 ? wv & 1 : 0;      // position, unit of the workgroup tile
    const int li = lane & 31, lk = lane >> 5;

    const int S = p.Cin >> 4;
    const int PC = p.W >> 4;
    const int UPS = p.H >> 4;                        // units per sample
    const int NT = p.Cout >> 6;
    const int G4 = 4 * NT;                           // tiles of one patch: (py, px, channel tile)
    const int wunits = NU == 1 ? p.N * UPS : UPS == 1 ? (p.n_groups > 1 ? p.grp[p.n_groups].unit0 : (p.N + 1) >> 1) : p.N * (UPS >> 1);
    const int ntiles = wunits * PC * G4;
    const unsigned plane_bytes = (unsigned)p.H * 64u;
    const unsigned slice_bytes = 24u * plane_bytes;
    const size_t unit_block = (size_t)S * slice_bytes;           // bytes of one (sample, patch column)

    // XCD-aware order (block b runs on XCD b % 8): every sweep of gridDim.x tiles is cut into contiguous runs per XCD, the tiles of one
    // patch next to each other: the workgroups that share a V patch run on one XCD at the same time
    int tl0;
    {
        const int nblk = gridDim.x, b = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = b & 7, idx = b >> 3;
        tl0 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int cq = tid & 15;                         // epilogue role: (pair = role >> 4, channel quad), roles 0..511
    float amax = 0.f;
    char* const wimg = smem + wv * WIMG;             // this wave's image in buffer 0

#ifdef P2P_W3_STAMPS
    int n_tile_seen = -1;
#endif
    for (int tl = tl0; tl < ntiles; tl += gridDim.x) {
#ifdef P2P_W3_STAMPS
    ++n_tile_seen;
#endif
    int rest = tl / G4;
    // the phase rotates with the sweep: a workgroup's tiles would otherwise all be of one phase (gridDim.x is a multiple of G4), and the
    // phases with py = 1 are 1.5x the work of the others
    const int ph = (tl % G4 + (rest * G4) / (int)gridDim.x) % G4;
    const int ntile = ph % NT, pyx = ph / NT;
    const int py = pyx >> 1, px = pyx & 1;
    const int pc = rest % PC;
    const int wunit = rest / PC;
    // the two units of the tile: unit u = (sample tn0 + u, rows 0..15) on 16-row grids, (sample tn0, rows ty0 + 16 u ..) otherwise
    int tn0, tn_end = p.N, ty0 = 0;
    const float* gu = p.U;
    const float* gscale = p.scale;
    const float* gshift = p.shift;
    if (NU == 1) {
        tn0 = wunit / UPS; ty0 = (wunit % UPS) * 16;
        if (p.n_groups > 1) {
            int g = 0;
            while (g + 1 < p.n_groups && p.grp[g + 1].sample0 <= tn0) ++g;
            gu = p.grp[g].U; gscale = p.grp[g].scale; gshift = p.grp[g].shift;
        }
    } else if (UPS == 1) {
        tn0 = wunit * 2;
        if (p.n_groups > 1) {                        // groups are runs of samples; every group is paired up on its own (unit0)
            int g = 0;
            while (g + 1 < p.n_groups && p.grp[g + 1].unit0 <= wunit) ++g;
            tn0 = p.grp[g].sample0 + 2 * (wunit - p.grp[g].unit0);
            tn_end = p.grp[g + 1].sample0;
            gu = p.grp[g].U; gscale = p.grp[g].scale; gshift = p.grp[g].shift;
        }
    } else {
        const int hb = UPS >> 1;                     // 32-row blocks per sample
        tn0 = wunit / hb; ty0 = (wunit % hb) * 32;
        if (p.n_groups > 1) {
            int g = 0;
            while (g + 1 < p.n_groups && p.grp[g + 1].sample0 <= tn0) ++g;
            gu = p.grp[g].U; gscale = p.grp[g].scale; gshift = p.grp[g].shift;
        }
    }
    const int n_me = (NU == 2 && UPS == 1) ? tn0 + mh : tn0;     // this wave's unit
    const int y0 = NU == 1 ? ty0 : UPS == 1 ? 0 : ty0 + mh * 16;
    const bool have = n_me < tn_end;

    // ---- V: global -> registers -> LDS, this wave's image: planes (j, hl, lk), rows y0 - 1 .. y0 + 16.
    //      Piece q < 4 = image rows 0..15 of plane q (1 KB); piece 4 = image rows 16, 17 of the four planes (32 lanes).
    const char* vbase = reinterpret_cast<const char*>(p.V) + ((size_t)(have ? n_me : 0) * PC + pc) * unit_block;
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (unsigned)unit_block, 0x00020000);
    unsigned vo_a, vo_c;
    {
        const int row = lane >> 2, t = lane & 3;
        const int ya = y0 - 1 + row;
        vo_a = (have && ya >= 0) ? (unsigned)(ya * 64 + t * 16) : OOB;                       // (ya <= y0 + 14 < H)
        const int yc = y0 + 15 + ((lane >> 2) & 1);
        vo_c = (have && lane < 32 && yc < p.H) ? (unsigned)(lane >> 3) * plane_bytes + (unsigned)(yc * 64 + t * 16) : OOB;
    }
    char* wreg = wimg + lane * 16;
    char* wreg_c = wimg + (lane >> 3) * WPLANE + 1024 + (lane & 7) * 16;
    f32x4 rv[5];
    const unsigned so_w = (unsigned)(j * 4) * plane_bytes;
    auto vload_all = [&](int slice, bool on) {          // `on` is wave-uniform: off = out of range = no traffic
#ifdef P2P_ABL3_V
        on = false;
#endif
#ifdef P2P_ABL3_VHOT
        slice = 0;                                   // timing experiment: every slice reads slice 0 again (L2-resident)
#endif
        const unsigned so = (unsigned)slice * slice_bytes + so_w;
#pragma unroll
        for (int q = 0; q < 5; ++q)
            rv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, (q < 4 ? vo_a : vo_c) | (on ? 0u : OOB), q < 4 ? so + (unsigned)q * plane_bytes : so, 0));
    };
    auto vstore_all = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(wreg + buf * BUF + q * WPLANE) = rv[q];
        if (lane < 32) *reinterpret_cast<f32x4*>(wreg_c + buf * BUF) = rv[4];
    };

    f32x16 acc[2][2];                                // [m-tile of the unit][32-channel half]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

    // V fragment of m-tile i, tap ky, half hl: plane (hl, lk) of the wave's image, row 8 i + ky + (li >> 2), tile li & 3
    const char* img0 = wimg + lk * WPLANE + (li >> 2) * 64 + (li & 3) * 16;

    lds_barrier();                                   // (persistent loop) the previous tile's exchange images have been read
    if (wv == 0) W3_STAMP(12);

    auto body = [&](auto nky_c) {
        constexpr int NKY = decltype(nky_c)::value;
        // U: the stream (py, px, channel tile, position j): K-step kb = 4 fragments of 1 KB (channel half 0 hi, lo, half 1 hi, lo).
        // Panel: [py][px][channel tile][position][slice][ky][4 KB]; one K-step of padding at its end.
        const size_t py_base = py ? (size_t)2 * NT * 6 * S * 2 * 4096 : 0;
        const size_t stream = (size_t)((px * NT + ntile) * 6 + j) * S * NKY * 4096;
        const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(gu) + py_base + stream), 0,
                                                                              (unsigned)((S * NKY + 1) * 4096), 0x00020000);
        const unsigned uoff = (unsigned)lane * 16u;
        f16x8 u[2][4];                               // (half 0 hi, lo, half 1 hi, lo) of the even / odd K-steps
        f16x8 vhp[2];                                // the next K-step's hi fragments
        auto uload = [&](int set, int kb) {
#pragma unroll
            for (int f = 0; f < 4; ++f) u[set][f] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_u, uoff + f * 1024, kb * 4096, 0));
        };
        // prologue: slice 0 -> buffer 0
        vload_all(0, true);
        uload(0, 0);
        vstore_all(0);
        if (wv == 0) W3_STAMP(13);
        // two slices per iteration: buffer and weight register set of every K-step are compile-time
        for (int s2 = 0; s2 < S; s2 += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int s = s2 + half;
#pragma unroll
                for (int ky = 0; ky < NKY; ++ky) {
                    const int kk = half * NKY + ky;
#ifdef P2P_ABL3_U
                    if (s2 == 0)
#endif
                    uload((kk + 1) & 1, s * NKY + ky + 1);               // (the panel's padding covers the K-step past the stream's end)
                    // the next slice's V pieces AFTER the next K-step's U fragments: vmcnt counts in issue order, so the wait for those
                    // fragments (one K-step from now) would otherwise wait for the V pieces too -- they have until the K-step after that
                    if (ky == 0) vload_all(s + 1, s + 1 < S);
                    __builtin_amdgcn_sched_barrier(0);
                    const char* img = img0 + half * BUF + ky * 64;
                    // hi fragments one K-step ahead inside a slice: a K-step starts with the four (U lo x V hi) products while its lo
                    // fragments are still on their way from LDS (12 MFMAs per K-step do not hide an LDS round trip in front of them)
                    f16x8 vh[2], vl[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        vh[i] = ky == 0 ? *reinterpret_cast<const f16x8*>(img + i * 512) : vhp[i];
                        vl[i] = *reinterpret_cast<const f16x8*>(img + i * 512 + 2 * WPLANE);
                    }
                    if (ky + 1 < NKY) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) vhp[i] = *reinterpret_cast<const f16x8*>(img + 64 + i * 512);
                    }
                    const f16x8* uc = u[kk & 1];
                    // (every accumulator still takes its three products in the order lo x hi, hi x lo, hi x hi: same bits)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uc[2 * c + 1], vh[i], acc[i][c], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uc[2 * c], vl[i], acc[i][c], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uc[2 * c], vh[i], acc[i][c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                vstore_all(half ^ 1);                                    // (a slice past the last one: zeros nobody reads)
            }
        }
    };
    if (py) body(std::integral_constant<int, 3>{});
    else body(std::integral_constant<int, 2>{});
    W3_STAMP(wv);
    lds_barrier();                                   // every wave is done with its image: the exchange images may overwrite them
    if (wv == 0) W3_STAMP(14);

    // ---- epilogue.  C/D layout of the 32x32 MFMA with U as the A operand: row = channel (r & 3) + 8 (r >> 2) + 4 lk of the 32-channel
    //      half, column li = pair.  Pass i: every wave puts m-tile i of its unit into exchange image mh, then thread (pair = tid >> 4,
    //      channel quad = tid & 15) of the first eight waves combines the six positions of both images into four output pixels each.
    const int col = ntile * 64 + cq * 4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (tid < 512) {                                // (6-wave workgroups: a thread's second role has the same channel quad, 384 % 16 == 0)
        if (gscale) sc = *reinterpret_cast<const f32x4*>(gscale + pyx * p.Cout + col);
        if (gshift) sh = *reinterpret_cast<const f32x4*>(gshift + col);
    }
#ifdef P2P_ABL3_EPI
    {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][c][r];
        if (t == 123.456f) p.out[tid] = t;
        continue;
    }
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i) lds_barrier();                        // pass 0's images have been read
        float* Xw = reinterpret_cast<float*>(smem + mh * XBUF);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc[i][c][4 * q], acc[i][c][4 * q + 1], acc[i][c][4 * q + 2], acc[i][c][4 * q + 3]};
                *reinterpret_cast<f32x4*>(Xw + (j * 32 + li) * XLD + c * 32 + 8 * q + 4 * lk) = v;
            }
        lds_barrier();
        for (int role = tid; role < 512; role += NTHR) {
            const int pair = role >> 4;
#pragma unroll
            for (int im = 0; im < NU; ++im) {
                // unit `im` of the tile
                const int n = (NU == 2 && UPS == 1) ? tn0 + im : tn0, yb = NU == 1 ? ty0 : UPS == 1 ? 0 : ty0 + im * 16;
                const bool hv = n < tn_end;
                const float* X = reinterpret_cast<const float*>(smem + im * XBUF);
                f32x4 m[6];
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) m[jj] = *reinterpret_cast<const f32x4*>(X + (jj * 32 + pair) * XLD + cq * 4);
                const int y = yb + i * 8 + (pair >> 2);
                if (hv) {
                    const size_t pix = ((size_t)n * (2 * p.H) + (2 * y + py)) * (2 * p.W) + 2 * (pc * 16 + (pair & 3) * 4) + px;
                    float* o = p.out + pix * p.out_cstride + p.out_coff + col;
                    f32x4 yv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // AT of F(4,3): rows (1 1 1 1 1 0), (0 1 -1 2 -2 0), (0 1 1 4 4 0), (0 1 -1 8 -8 1)
                        const float s12 = m[1][e] + m[2][e], d12 = m[1][e] - m[2][e];
                        const float s34 = m[3][e] + m[4][e], d34 = m[3][e] - m[4][e];
                        yv[0][e] = (m[0][e] + s12) + s34;
                        yv[1][e] = __builtin_fmaf(2.f, d34, d12);
                        yv[2][e] = __builtin_fmaf(4.f, s34, s12);
                        yv[3][e] = __builtin_fmaf(8.f, d34, d12) + m[5][e];
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
                        *reinterpret_cast<f32x4*>(o + (size_t)(2 * k) * p.out_cstride) = v;
                    }
                }
            }
        }
    }
    }   // tiles
    range_commit(p.range_acc, amax);
}

}  // namespace

#ifdef P2P_W3_STAMPS
extern "C" __attribute__((visibility("default"))) int p2p_dbg_w3_stamps(unsigned long long* host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w3_stamps), sizeof(g_w3_stamps));
}
#endif

bool wino3_supported(int H, int W, int Cin, int Cout)
{
    if (Cin % 32 || Cout % 64 || W % 16) return false;
    return (H == 16 && W == 16) || H % 32 == 0;
}

// bytes of V for N samples
size_t wino3_v_bytes(int N, int H, int W, int Cin) { return (size_t)N * H * W * Cin * 6; }

hipError_t launch_wino3_input(const Wino3Params& p, hipStream_t s)
{
    const int grid = p.N * (p.W / 16) * (p.Cin / 32) * (p.H / 8);
    hipLaunchKernelGGL(wino3_input_kernel, dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_wino3_gemm(const Wino3Params& p, hipStream_t s)
{
    // persistent: the workgroups of a full chip walk the tiles; the grid is kept a multiple of the tiles of one patch (the phase rotation counts on it)
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipGetLastError();
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
#ifdef P2P_W3_NU1
    constexpr int NU = 1;                            // A/B builds: 6-wave workgroups, two per CU -- measured 9 % slower (tools/experiments/README.md)
#else
    constexpr int NU = 2;
#endif
    const int g4 = 4 * (p.Cout / 64);
    const int ups = p.H / 16;
    const int wunits = NU == 1 ? p.N * ups : ups == 1 ? (p.n_groups > 1 ? p.grp[p.n_groups].unit0 : (p.N + 1) / 2) : p.N * (ups / 2);
    const int tiles = wunits * (p.W / 16) * g4;
    const int slots = n_cu * (2 / NU);
    int grid = tiles < slots ? tiles : slots / g4 * g4;
    if (grid < 1) grid = tiles < g4 ? tiles : g4;
    hipLaunchKernelGGL((wino3_gemm_kernel<NU>), dim3(grid), dim3(NU * 384), 0, s, p);
    return hipGetLastError();
}

}  // namespace p2p
