// Winograd F(4, 3) along the row axis for the two stride-2 layers on an 8x8 grid (gfx950, PREC_F16X3): the transposed convolution up1
// (8x8 -> 16x16; reference pix2pose_model/ae_model.py:201-204) and the 5x5 / 2 'SAME' convolution conv4 (16x16 -> 8x8; ae_model.py:190-195,
// resnet50 and paper encoders alike).  Same arithmetic and kernel structure as wino3.hip (which serves the 16x16 and 32x32 grids): six position
// GEMMs on a 12-wave workgroup, wave (j, mh) = position j of one half of the tile's rows x all 64 output channels, staging its own planes;
// what differs is the geometry -- an 8x8 grid is two 4-column tiles wide, so a 32-pair m-tile is TWO samples, a wave's unit FOUR samples and a
// workgroup tile EIGHT samples -- and, for conv4, where K comes from:
//
//   MODE 0 (up1)    the four sub-pixel phases as (2 + py) x 3-tap correlations on the input grid, as in wino3.hip.
//   MODE 1 (conv4)  y[i][k] = sum x[2i + kh - 1][2k + kw - 1] w[kh][kw]: on the four parity planes P(a, b)[r][s] = x[2r + a][2s + b] this is a
//                   sum of four stride-1 correlations with (2 + a) x (2 + b) taps -- odd planes: offsets {-1, 0, +1} (kh = 0, 2, 4), even planes:
//                   {0, +1} (kh = 1, 3).  Along a plane row both column parities are 3-tap filters (the even one zero-extended), so every
//                   plane gets the F(4,3) input transform and the position GEMM walks K = (plane, channel slice, vertical tap): 15 position-
//                   products per output pixel and input channel instead of 25.  Small launches split K over the four planes (raw partial
//                   sums after the inverse transform -- it is linear -- then splitk_reduce_kernel applies BatchNorm and the activation).
//
// V layout (input transform -> GEMM): [unit of 8 samples][K slice][plane = (j, hi/lo, k half)][sample 8][row 8][tile 2][8 halves]; K slice =
// 16-channel slice (MODE 0) or (parity plane, 16-channel slice) (MODE 1).  LDS image of a K slice: 24 planes of [sample 8][10 rows: zero, the
// 8 rows, zero][tile 2][16 B]; m-tile i = samples 2i, 2i + 1; a vertical tap is a constant byte shift (32 B per row).
#include "kernels.h"

#include <type_traits>

namespace p2p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr unsigned VPLANE = 2048;                   // bytes of one V plane in HBM: 8 samples x 8 rows x 2 tiles x 16 B
constexpr unsigned VSLICE = 24 * VPLANE;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// unit (group of eight samples of one object) and slot of sample n
__device__ __forceinline__ void unit_of(const Wino3oParams& p, int n, int& unit, int& slot)
{
    if (p.n_groups > 1) {
        int g = 0;
        while (g + 1 < p.n_groups && p.grp[g + 1].sample0 <= n) ++g;
        const int k = n - p.grp[g].sample0;
        unit = p.grp[g].unit0 + (k >> 3); slot = k & 7;
    } else { unit = n >> 3; slot = n & 7; }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// input transform.  Thread = (channel quad of a 32-channel group, tile t, row, sub-block); a block = 32 channels of two samples (MODE 0)
// or of the two column-parity planes of one (sample, row parity) (MODE 1).
// ------------------------------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void wino3o_input_kernel(const Wino3oParams p)
{
    const int tid = threadIdx.x;
    const int quad = tid & 7, t = (tid >> 3) & 1, r = (tid >> 4) & 7, sub = tid >> 7;
    const int cgroups = p.Cin >> 5;
    int b = blockIdx.x;
    const int cg = b % cgroups; b /= cgroups;
    const int S = p.Cin >> 4;
    int n, sq;                                       // sample, K slice of the first of this thread's two 16-channel slices
    f32x4 d[6];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const unsigned c0 = (unsigned)(cg * 32 + quad * 4);
    if (MODE == 0) {
        n = b * 2 + sub;
        sq = cg * 2;
        const unsigned rowpix = (unsigned)((n * 8 + r) * 8);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int x = t * 4 - 1 + k;
            const unsigned off = (n < p.N && (unsigned)x < 8u) ? ((rowpix + (unsigned)x) * (unsigned)p.Cin + c0) * 4u : OOB;
            d[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
    } else {
        const int a = b & 1;
        n = b >> 1;
        const int bb = sub;                          // column parity
        sq = (a * 2 + bb) * S + cg * 2;
        const unsigned rowpix = (unsigned)((n * 16 + 2 * r + a) * 16);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int x = 2 * (t * 4 - 1 + k) + bb;
            const unsigned off = (unsigned)x < 16u ? ((rowpix + (unsigned)x) * (unsigned)p.Cin + c0) * 4u : OOB;
            d[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
    }
    if (n >= p.N) return;
    f32x4 v[6];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float d0 = d[0][e], d1 = d[1][e], d2 = d[2][e], d3 = d[3][e], d4 = d[4][e], d5 = d[5][e];
        // BT of F(4,3) at {0, 1, -1, 2, -2, inf}, the same expressions as wino3_input_kernel
        const float a12 = __builtin_fmaf(-4.f, d2, d4), b12 = __builtin_fmaf(-4.f, d1, d3);
        const float a34 = d4 - d2, b34 = 2.f * (d3 - d1);
        v[0][e] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
        v[1][e] = a12 + b12; v[2][e] = a12 - b12;
        v[3][e] = a34 + b34; v[4][e] = a34 - b34;
        v[5][e] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
    }
    int unit, slot;
    unit_of(p, n, unit, slot);
    const int lk = (quad >> 1) & 1;
    const size_t nsl = MODE == 0 ? (size_t)S : (size_t)4 * S;
    char* dst = reinterpret_cast<char*>(p.V) + ((size_t)unit * nsl + sq + (quad >> 2)) * VSLICE + (size_t)lk * VPLANE + slot * 256 + r * 32 + t * 16 + (quad & 1) * 8;
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const f32x4 w = v[j];
        amax = range_note4(amax, w);
        const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(w[0], w[1]), h23 = __builtin_amdgcn_cvt_pkrtz(w[2], w[3]);
        fp16x2 l01, l23;
        l01[0] = (__fp16)(w[0] - (float)h01[0]); l01[1] = (__fp16)(w[1] - (float)h01[1]);
        l23[0] = (__fp16)(w[2] - (float)h23[0]); l23[1] = (__fp16)(w[3] - (float)h23[1]);
        *reinterpret_cast<uint2*>(dst + (size_t)(j * 4) * VPLANE) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
        *reinterpret_cast<uint2*>(dst + (size_t)(j * 4 + 2) * VPLANE) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
    }
    range_commit(p.range_acc, amax);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// the six position GEMMs + inverse transform + epilogue
// ------------------------------------------------------------------------------------------------------------------------------------
// Wave (j, mh) owns position j of FOUR samples (two m-tiles) and all 64 channels of the tile, stages its own image -- its position's four
// planes of those samples -- and double-buffers it privately: no workgroup barrier in the K loop (see wino3_gemm_kernel).
template <int MODE>
__global__ __launch_bounds__(768) void wino3o_gemm_kernel(const Wino3oParams p)
{
    constexpr int SROW = 320;                       // LDS bytes of one sample of a plane: 10 rows x 2 tiles x 16 B
    constexpr int WPLANE = 4 * SROW + 128;          // one plane of a wave's image (four samples); + 128 B: the two k halves of a fragment read
                                                    // (lanes 0..31 / 32..63, planes lk = 0 / 1) start 32 banks apart (1280 B apart they collided:
                                                    // bank-conflict cycles 35 % of the LDS-active cycles, profiles/r06_sq_counters.txt)
    constexpr int WIMG = 4 * WPLANE;
    constexpr int BUF = WIMG;                       // a wave's two buffers sit next to each other (every LDS offset stays a 16-bit immediate)
    constexpr int LDS_BYTES = 24 * WIMG;
    constexpr int XLD = 68;                         // exchange image: [position 6][pair 32][64 channels + 4] floats
    constexpr int XBUF = 6 * 32 * XLD * 4;
    static_assert(2 * XBUF <= LDS_BYTES, "two exchange images fit the slice buffers");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = wv >> 1, mh = wv & 1;
    const int li = lane & 31, lk = lane >> 5;

    const int S = p.Cin >> 4;
    const int NT = p.Cout >> 6;
    const int KS = MODE == 1 ? p.ksplit : 1;
    const int G4 = MODE == 0 ? 4 * NT : NT * KS;    // tiles of one unit: (py, px, channel tile), or (K split, channel tile)
    const int units = p.n_groups > 1 ? p.grp[p.n_groups].unit0 : (p.N + 7) >> 3;
    const int ntiles = units * G4;
    const size_t unit_block = (size_t)(MODE == 0 ? S : 4 * S) * VSLICE;

    int tl0;
    {
        const int nblk = gridDim.x, b = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = b & 7, idx = b >> 3;
        tl0 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int pair = tid >> 4, cq = tid & 15;       // epilogue role (threads 0..511)
    float amax = 0.f;
    char* const wimg = smem + wv * 2 * WIMG;         // this wave's image in buffer 0

    for (int tl = tl0; tl < ntiles; tl += gridDim.x) {
    const int unit = tl / G4;
    int ph = tl % G4;
    if (MODE == 0) ph = (ph + (unit * G4) / (int)gridDim.x) % G4;        // the phase rotates with the sweep (see wino3.hip)
    const int ntile = ph % NT, pyx = ph / NT;       // MODE 0: pyx = phase; MODE 1: pyx = K split
    const int py = pyx >> 1, px = pyx & 1;
    int n0 = unit * 8, n_end = p.N;
    const float* gu = p.U;
    const float* gscale = p.scale;
    const float* gshift = p.shift;
    if (p.n_groups > 1) {
        int g = 0;
        while (g + 1 < p.n_groups && p.grp[g + 1].unit0 <= unit) ++g;
        n0 = p.grp[g].sample0 + 8 * (unit - p.grp[g].unit0);
        n_end = p.grp[g + 1].sample0;
        gu = p.grp[g].U; gscale = p.grp[g].scale; gshift = p.grp[g].shift;
    }
    const int nvalid = n_end - n0 < 8 ? n_end - n0 : 8;

    // ---- V: global -> registers -> LDS: piece q = plane (hl, lk) = q of this wave's position, its four samples (1 KB).
    const char* vbase = reinterpret_cast<const char*>(p.V) + (size_t)unit * unit_block;
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (unsigned)unit_block, 0x00020000);
    const unsigned so_w = (unsigned)(j * 4) * VPLANE + (unsigned)mh * 1024u;
    const unsigned vo = 4 * mh + (lane >> 4) < nvalid ? (unsigned)lane * 16u : OOB;      // a missing sample (the last unit of an object) reads zeros
    // LDS: sample lane >> 4, row ((lane >> 1) & 7) + 1, tile lane & 1
    char* wreg = wimg + (lane >> 4) * SROW + (((lane >> 1) & 7) + 1) * 32 + (lane & 1) * 16;
    f32x4 rv[4];
    auto vload_all = [&](int sq, bool on) {
        const unsigned so = (unsigned)sq * VSLICE + so_w;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            rv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo | (on ? 0u : OOB), so + (unsigned)q * VPLANE, 0));
    };
    auto vstore_all = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(wreg + buf * BUF + q * WPLANE) = rv[q];
    };

    f32x16 acc[2][2];                                // [m-tile of the wave's four samples][32-channel half]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;

    // V fragment of m-tile i, image row offset ro: plane (hl, lk), sample 2i + (li >> 4) of the wave's four, image row ro + ((li >> 1) & 7), tile li & 1
    const char* img0 = wimg + lk * WPLANE + (li >> 4) * SROW + ((li >> 1) & 7) * 32 + (li & 1) * 16;

    lds_barrier();                                   // (persistent loop) the previous tile's exchange images have been read
    {
        // the zero rows (image rows 0 and 9 of every sample) of this wave's image in both buffers: 2 x 4 planes x 4 samples x 2 rows x 2 slots = 128
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int sl = lane + 64 * k;
            const int bf = sl >> 6, pl = (sl >> 4) & 3, smp = (sl >> 2) & 3, zr = (sl >> 1) & 1, t = sl & 1;
            *reinterpret_cast<f32x4*>(wimg + bf * BUF + pl * WPLANE + smp * SROW + zr * 9 * 32 + t * 16) = z;
        }
    }

    // one run of K slices [sq0, sq1) with NKY vertical taps each, image row offset RO + ky, U K-steps from kb0 on
    auto body = [&](auto nky_c, auto ro_c, int sq0, int sq1, int kb0, const __amdgpu_buffer_rsrc_t rs_u) {
        constexpr int NKY = decltype(nky_c)::value;
        constexpr int RO = decltype(ro_c)::value;
        const unsigned uoff = (unsigned)lane * 16u;
        f16x8 u[2][4];
        auto uload = [&](int set, int kb) {
#pragma unroll
            for (int f = 0; f < 4; ++f) u[set][f] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_u, uoff + f * 1024, kb * 4096, 0));
        };
        vload_all(sq0, true);
        uload(0, kb0);
        vstore_all(0);
        for (int s2 = sq0; s2 < sq1; s2 += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int s = s2 + half;
#pragma unroll
                for (int ky = 0; ky < NKY; ++ky) {
                    const int kk = half * NKY + ky;
                    uload((kk + 1) & 1, kb0 + (s - sq0) * NKY + ky + 1);
                    if (ky == 0) vload_all(s + 1, s + 1 < sq1);          // after the U fragments: vmcnt counts in issue order (wino3.hip)
                    __builtin_amdgcn_sched_barrier(0);
                    const char* img = img0 + half * BUF + (RO + ky) * 32;
                    f16x8 vh[2], vl[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        vh[i] = *reinterpret_cast<const f16x8*>(img + i * 2 * SROW);
                        vl[i] = *reinterpret_cast<const f16x8*>(img + i * 2 * SROW + 2 * WPLANE);
                    }
                    const f16x8* uc = u[kk & 1];
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uc[2 * c + 1], vh[i], acc[i][c], 0, 0, 0);
                            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uc[2 * c], vl[i], acc[i][c], 0, 0, 0);
                            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uc[2 * c], vh[i], acc[i][c], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                vstore_all(half ^ 1);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    if (MODE == 0) {
        // panel [py][px][channel tile][position][slice][ky][4 KB] (+ one K-step of padding), as in wino3.hip
        const size_t py_base = py ? (size_t)2 * NT * 6 * S * 2 * 4096 : 0;
        const int nky = 2 + py;
        const size_t stream = (size_t)((px * NT + ntile) * 6 + j) * S * nky * 4096;
        const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(gu) + py_base + stream), 0,
                                                                              (unsigned)((S * nky + 1) * 4096), 0x00020000);
        if (py) body(I3{}, I0{}, 0, S, 0, rs_u);
        else body(I2{}, I0{}, 0, S, 0, rs_u);
    } else {
        // panel [channel tile][position][K-steps: (a = 0: b, slice, ky < 2) then (a = 1: b, slice, ky < 3)][4 KB] (+ one K-step of padding)
        const size_t stream = (size_t)(ntile * 6 + j) * (10 * S) * 4096;
        const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(gu) + stream), 0,
                                                                              (unsigned)((10 * S + 1) * 4096), 0x00020000);
        if (KS == 1) {
            body(I2{}, I1{}, 0, 2 * S, 0, rs_u);                 // even rows: taps dy = 0, +1 -> image rows y + 1 + ky
            body(I3{}, I0{}, 2 * S, 4 * S, 4 * S, rs_u);         // odd rows: taps dy = -1, 0, +1 -> image rows y + ky
        } else if (pyx < 2) body(I2{}, I1{}, pyx * S, (pyx + 1) * S, pyx * 2 * S, rs_u);
        else body(I3{}, I0{}, pyx * S, (pyx + 1) * S, 4 * S + (pyx - 2) * 3 * S, rs_u);
    }
    lds_barrier();                                   // every wave is done with its image: the exchange images may overwrite them

    // ---- epilogue (see wino3.hip).  Pass i: every wave puts m-tile i of its four samples into exchange image mh; pair = sample
    //      4 im + 2 i + (pair >> 4), row (pair >> 1) & 7, tile pair & 1 of image im.
    const int col = ntile * 64 + cq * 4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    const bool raw = MODE == 1 && KS > 1;           // partial sums: BatchNorm and the activation happen in the reduction
    if (tid < 512 && !raw) {
        if (gscale) sc = *reinterpret_cast<const f32x4*>(gscale + (MODE == 0 ? pyx * p.Cout : 0) + col);
        if (gshift) sh = *reinterpret_cast<const f32x4*>(gshift + col);
    }
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
        if (tid < 512) {
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                const int smp = 4 * im + 2 * i + (pair >> 4);
                if (smp >= nvalid) continue;
                const float* X = reinterpret_cast<const float*>(smem + im * XBUF);
                f32x4 m[6];
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) m[jj] = *reinterpret_cast<const f32x4*>(X + (jj * 32 + pair) * XLD + cq * 4);
                const int n = n0 + smp, y = (pair >> 1) & 7, x0 = (pair & 1) * 4;
                float* o;
                size_t ostep;
                if (MODE == 0) {
                    const size_t pix = ((size_t)n * 16 + (2 * y + py)) * 16 + 2 * x0 + px;
                    o = p.out + pix * p.out_cstride + p.out_coff + col;
                    ostep = (size_t)2 * p.out_cstride;
                } else if (raw) {
                    o = p.partial + ((size_t)pyx * p.N * 64 + ((size_t)n * 8 + y) * 8 + x0) * p.Cout + col;
                    ostep = (size_t)p.Cout;
                } else {
                    o = p.out + (((size_t)n * 8 + y) * 8 + x0) * p.out_cstride + p.out_coff + col;
                    ostep = (size_t)p.out_cstride;
                }
                f32x4 yv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
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
                    if (!raw) {
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
                    }
                    *reinterpret_cast<f32x4*>(o + (size_t)k * ostep) = v;
                }
            }
        }
    }
    }   // tiles
    range_commit(p.range_acc, amax);
}

}  // namespace

bool wino3o_supported(int mode, int Cin, int Cout) { return (mode == 0 || mode == 1) && Cin % 32 == 0 && Cout % 64 == 0; }

// bytes of V for `units` groups of eight samples
size_t wino3o_v_bytes(int mode, int units, int Cin) { return (size_t)units * (mode == 0 ? 1 : 4) * (Cin / 16) * VSLICE; }

int wino3o_units(const Wino3oParams& p) { return p.n_groups > 1 ? p.grp[p.n_groups].unit0 : (p.N + 7) / 8; }

hipError_t launch_wino3o_input(const Wino3oParams& p, int mode, hipStream_t s)
{
    const int cg = p.Cin / 32;
    if (mode == 0) hipLaunchKernelGGL((wino3o_input_kernel<0>), dim3(((p.N + 1) / 2) * cg), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wino3o_input_kernel<1>), dim3(p.N * 2 * cg), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_wino3o_gemm(const Wino3oParams& p, int mode, hipStream_t s)
{
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipGetLastError();
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int g4 = mode == 0 ? 4 * (p.Cout / 64) : (p.Cout / 64) * p.ksplit;
    const int tiles = wino3o_units(p) * g4;
    int grid = tiles < n_cu ? tiles : n_cu / g4 * g4;
    if (grid < 1) grid = tiles < g4 ? tiles : g4;
    if (mode == 0) hipLaunchKernelGGL((wino3o_gemm_kernel<0>), dim3(grid), dim3(768), 0, s, p);
    else hipLaunchKernelGGL((wino3o_gemm_kernel<1>), dim3(grid), dim3(768), 0, s, p);
    return hipGetLastError();
}

}  // namespace p2p
