// The ResNet identity bottleneck block of resblock.hip (resnet50_mod.py:40-73) on 512-THREAD workgroups: eight waves of at most 128 registers, two
// workgroups per CU = FOUR waves per SIMD instead of two.
//
// Why: tools/mfma_valu.hip -- the vector instructions a wave issues between its MFMAs do not hide under them (every one adds ~1/7 of an MFMA period), while
// the MFMAs of ANOTHER wave of the SIMD hide them completely.  The fused block carries 2200 vector instructions beside 456 MFMAs per wave (hi/lo splits of
// the loader and of two epilogues, BatchNorm, the range guard, addressing) and marches through three barrier-separated phases: with two 256-register waves
// per SIMD there is rarely a second wave with MFMAs ready.  Same patch, same LDS images, same K-step order and MFMA chain per output element as
// resblock_kernel -- the SAME bits (tests/test_resblock_gpu.py) -- with the tiles of every phase dealt to eight waves:
//
//   phase A  12 (m-tile, n-tile) pairs of the halo GEMM: wave = (n-tile, m-group), one or two m-tiles (6 m-tiles on 4 groups / 3 on 2)
//   phase B  8 pairs: one 32 x 32 tile per wave; weight fragments straight from global in fragment order, X fragments one K-step ahead
//   phase C  per 128-channel chunk 8 / 16 pairs: wave = (half of the patch, n-tile), one or two m-tiles
//
// Identity blocks only (res2b/c, res3b/c/d); the projection blocks keep resblock_kernel<..., PROJ>.
#include "kernels.h"

namespace p2p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int REC = 144;        // activation record: [hi f16 x32 | lo f16 x32 | 16 B pad] (igemm_halo.hip)
constexpr int WREC = 128;       // weight row of a K-step: [hi x32 | lo x32], 16-byte chunk c of row r at c ^ ((r >> 1) & 7)
constexpr unsigned OOB = 0xFFFFFFF0u;

constexpr int cmax8(int a, int b) { return a > b ? a : b; }

// hi = f16(v) toward zero (cvt_pkrtz), lo = f16(v - hi) to nearest: the loaders' split (igemm.hip lstore, igemm_halo.hip hstore)
__device__ __forceinline__ void split4(const f32x4 v, uint2& hi, uint2& lo)
{
    const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
    fp16x2 l01, l23;
    l01[0] = (__fp16)(v[0] - (float)h01[0]); l01[1] = (__fp16)(v[1] - (float)h01[1]);
    l23[0] = (__fp16)(v[2] - (float)h23[0]); l23[1] = (__fp16)(v[3] - (float)h23[1]);
    hi = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    lo = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}

template <int F1, int PY, int HPX>
__global__ __launch_bounds__(512, 4) void resblock8_kernel(const ResBlockParams p)
{
    constexpr int NT = 512;
    constexpr int C = 4 * F1, CIN = C;
    constexpr int SA = CIN / 32;                  // K-steps of 2a
    constexpr int SB = F1 / 32;                   // channel slices of t_a / t_b
    constexpr int NCH = C / 128;                  // 128-channel output chunks of 2c
    constexpr int HX0 = HPX == 18 ? 1 : 0;
    constexpr int HPY = PY + 2, NPA = HPY * HPX, MA = (NPA + 31) / 32;
    constexpr int PITCH = (HPX * REC + 255) / 256 * 256;
    constexpr int TSLICE = HPY * PITCH, T_BYTES = SB * TSLICE;
    constexpr int NPIX = PY * 16, MT = PY / 2;
    constexpr int T2SLICE = NPIX * REC, T2_BYTES = SB * T2SLICE;
    constexpr int XS_BYTES = MA * 32 * REC;
    constexpr int WA_BYTES = F1 * WREC;
    constexpr int WC_BYTES = 2 * 128 * WREC;      // two K-steps of a 128-row chunk of the 2c panel
    constexpr int NST = SB / 2;
    constexpr int TMC = MT / 2;                   // phase C: m-tiles per wave (waves = 2 halves of the patch x 4 n-tiles of the chunk)
    constexpr int CLD = 128 + 4;
    constexpr int CS_BYTES = TMC * 32 * CLD * 4;
    constexpr int ZERO_OFF = cmax8(cmax8(2 * (XS_BYTES + WA_BYTES), T_BYTES), T2_BYTES + cmax8(WC_BYTES, CS_BYTES));
    constexpr int SS_OFF = ZERO_OFF + 128, SS_FLOATS = 4 * F1 + 2 * C;
    constexpr int SMEM = SS_OFF + SS_FLOATS * 4;
    static_assert(SMEM <= 80 * 1024, "two workgroups per CU");
    static_assert((F1 == 64 || F1 == 128) && (PY == 8 || PY == 4) && SA % 2 == 0 && SS_FLOATS % 4 == 0, "shapes");
    __shared__ __attribute__((aligned(256))) char smem[SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;

    // XCD-aware tile order (block b runs on XCD b % 8): contiguous runs of patches per XCD
    const int tiles_x = p.W / 16, tiles_y = p.H / PY;
    int t;
    {
        const int nblk = gridDim.x, b = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int x0 = (t % tiles_x) * 16; t /= tiles_x;
    const int y0 = (t % tiles_y) * PY;
    const int n = t / tiles_y;

    int g = 0;                                   // mixed-object batches: groups are runs of samples
    while (g + 1 < p.n_groups && p.grp[g + 1].sample0 <= n) ++g;
    const float* ss = reinterpret_cast<const float*>(smem + SS_OFF);      // [s2a F1 | h2a F1 | s2b F1 | h2b F1 | s2c C | h2c C]
    for (int i = tid * 4; i < SS_FLOATS; i += NT * 4)
        *reinterpret_cast<f32x4*>(smem + SS_OFF + i * 4) = *reinterpret_cast<const f32x4*>(p.grp[g].ss + i);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wa = __builtin_amdgcn_make_buffer_rsrc((void*)p.grp[g].w2a, 0, p.wa_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wc = __builtin_amdgcn_make_buffer_rsrc((void*)p.grp[g].w2c, 0, p.wc_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wbf = __builtin_amdgcn_make_buffer_rsrc((void*)p.grp[g].w2b_frag, 0, (unsigned)(F1 * 9 * F1 * 4), 0x00020000);

    if (tid < 8) *reinterpret_cast<f32x4*>(smem + ZERO_OFF + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    // weight loader: rows (tid >> 3) + 64 j, 16-byte segment tid & 7, swizzled chunk
    const int lrow = tid >> 3, lseg = tid & 7;
    const int w_dst = lrow * WREC + ((lseg ^ ((lrow >> 1) & 7)) << 4);
    float amax = 0.f;                             // operand-range guard (kernels.h): t_a, t_b and the block output

    // =========================================================================================== phase A: t_a = relu(bn(W2a x)) on the halo
    constexpr int WN = F1 / 32 >= 4 ? 4 : F1 / 32;        // waves along channels: 2 | 4
    constexpr int NMG = 8 / WN;                            // m-groups: 4 | 2
    const int ntA = wave % WN, mgA = wave / WN;
    const bool twoA = mgA + NMG < MA;                      // this wave owns m-tiles mgA and mgA + NMG (wave-uniform)
    static_assert(MA <= 2 * NMG, "at most two m-tiles per wave");
    f32x16 accA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[i][r] = 0.f;
    {
        char* Xs = smem;                           // [2][MA * 32 records]
        char* Was = smem + 2 * XS_BYTES;           // [2][F1 rows]
        constexpr int XL = (MA * 256 + NT - 1) / NT;      // float4 per thread and K-step: 3 | 2 (the last one of F1 = 128 half used)
        unsigned x_off[XL];
        int x_dst[XL];
        bool x_st[XL];
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            // float4 idx = tid + 512 j -> quad idx & 7 of halo pixel perm(idx >> 3) (resblock.hip: the store pattern of the halo loaders)
            const int idx = tid + NT * j;
            const int t8 = idx >> 3, q = idx & 7;
            const int hp = (t8 & ~7) | ((t8 & 1) << 2) | ((t8 >> 1) & 3);
            const int hy = hp / HPX, hx = hp - hy * HPX;
            const int iy = y0 - 1 + hy, ix = x0 - HX0 + hx;
            x_st[j] = idx < MA * 256;
            const bool ok = x_st[j] && hp < NPA && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            x_off[j] = ok ? ((((unsigned)(n * p.H + iy) * (unsigned)p.W + (unsigned)ix) * (unsigned)CIN + (unsigned)(q * 4)) * 4u) : OOB;
            x_dst[j] = hp * REC + q * 8;
        }
        constexpr int WL = F1 / 64;
        unsigned wa_off[WL];
#pragma unroll
        for (int j = 0; j < WL; ++j) wa_off[j] = ((unsigned)(lrow + 64 * j) * (unsigned)CIN + (unsigned)lseg * 4u) * 4u;
        f32x4 rx[2][XL], rw[2][WL];               // two stages of global loads in flight
        auto gload = [&](int s, f32x4 (&qx)[XL], f32x4 (&qw)[WL]) {
#pragma unroll
            for (int j = 0; j < XL; ++j) qx[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, x_off[j], s * 128, 0));
#pragma unroll
            for (int j = 0; j < WL; ++j) qw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wa, wa_off[j], s * 128, 0));
        };
        auto lstore = [&](const f32x4 (&qx)[XL], const f32x4 (&qw)[WL], int b) {
#pragma unroll
            for (int j = 0; j < XL; ++j) {
                if (!x_st[j]) continue;
                uint2 hi, lo;
                split4(qx[j], hi, lo);
                *reinterpret_cast<uint2*>(Xs + b * XS_BYTES + x_dst[j]) = hi;
                *reinterpret_cast<uint2*>(Xs + b * XS_BYTES + x_dst[j] + 64) = lo;
            }
#pragma unroll
            for (int j = 0; j < WL; ++j) *reinterpret_cast<f32x4*>(Was + b * WA_BYTES + w_dst + 64 * j * WREC) = qw[j];
        };
        gload(0, rx[0], rw[0]);
        gload(1, rx[1], rw[1]);
        lstore(rx[0], rw[0], 0);
        __syncthreads();
        const char* Xf = Xs + (mgA * 32 + li) * REC + lk * 16;
        const char* Wf = Was + (ntA * 32 + li) * WREC;
        int w_sw[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) w_sw[kb][hf] = ((kb * 2 + lk + 4 * hf) ^ ((li >> 1) & 7)) << 4;
        for (int g0 = 0; g0 < SA; g0 += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int stg = g0 + d;            // staged in buffer d; its ring slot d is free again
                if (stg + 2 < SA) gload(stg + 2, rx[d], rw[d]);
                if (stg + 1 < SA) lstore(rx[(d + 1) & 1], rw[(d + 1) & 1], (d + 1) & 1);      // (its buffer was last read before the previous barrier)
                // (no explicit fragment prefetch here: at four waves per SIMD the other waves cover an LDS round trip, and the registers are not there)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(Wf + d * WA_BYTES + w_sw[kb][0]);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(Wf + d * WA_BYTES + w_sw[kb][1]);
                    {
                        const f16x8 xh = *reinterpret_cast<const f16x8*>(Xf + d * XS_BYTES + kb * 32);
                        const f16x8 xl = *reinterpret_cast<const f16x8*>(Xf + d * XS_BYTES + kb * 32 + 64);
                        accA[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, accA[0], 0, 0, 0);      // (al bh, ah bl, ah bh) with a = activation, b = weight,
                        accA[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, accA[0], 0, 0, 0);      // operand roles swapped: D[channel][pixel]
                        accA[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, accA[0], 0, 0, 0);
                    }
                    if (twoA) {
                        const f16x8 xh = *reinterpret_cast<const f16x8*>(Xf + d * XS_BYTES + NMG * 32 * REC + kb * 32);
                        const f16x8 xl = *reinterpret_cast<const f16x8*>(Xf + d * XS_BYTES + NMG * 32 * REC + kb * 32 + 64);
                        accA[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, accA[1], 0, 0, 0);
                        accA[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, accA[1], 0, 0, 0);
                        accA[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, accA[1], 0, 0, 0);
                    }
                }
                __syncthreads();
            }
        }
    }

    // 2b weights: this wave's n-tile straight from global in fragment order (ResBlockGroup::w2b_frag), three K-steps ahead; K order (slice, tap)
    const int ntB = wave % WN, mtB = wave / WN;
    static_assert(MT * WN == 8, "one 32 x 32 tile per wave in phase B");
    constexpr int TOTAL = SB * 9;
    f16x8 rwf[3][4];                              // (k half 0 hi, lo, k half 1 hi, lo) of the K-steps in flight
    auto wfload = [&](int ks, f16x8 (&q)[4]) {
        const int chunk = ks / 9, tap = ks - chunk * 9;                       // the panel's order is (tap, slice)
        const unsigned base = (unsigned)((ntB * TOTAL + tap * SB + chunk) * 4096);
#pragma unroll
        for (int f = 0; f < 4; ++f) q[f] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wbf, (unsigned)lane * 16u + f * 1024u, base, 0));
    };
#pragma unroll
    for (int d = 0; d < 3; ++d) wfload(d, rwf[d]);

    // ---- epilogue A: lane = halo pixel mt * 32 + li, channels ntA * 32 + 8 g + 4 lk + {0..3}; the image replaces the staging buffers
    {
        char* T = smem + ntA * TSLICE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 1 && !twoA) break;
            const int hp = (mgA + i * NMG) * 32 + li;
            const int hy = hp / HPX, hx = hp - hy * HPX;
            const int iy = y0 - 1 + hy, ix = x0 - HX0 + hx;
            const bool inside = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            char* dst = T + hy * PITCH + hx * REC + lk * 8;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int c = ntA * 32 + 8 * gq + 4 * lk;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(ss + c);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(ss + F1 + c);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = relu_nan(fmaf(accA[i][4 * gq + e], sc[e], sh[e]));
                amax = range_note4(amax, v);
                if (!inside) v = f32x4{0.f, 0.f, 0.f, 0.f};                  // the 3x3 convolution zero-pads t_a
                uint2 hi, lo;
                split4(v, hi, lo);
                if (hp < NPA) {
                    *reinterpret_cast<uint2*>(dst + gq * 16) = hi;
                    *reinterpret_cast<uint2*>(dst + gq * 16 + 64) = lo;
                }
            }
        }
    }
    __syncthreads();                              // the t_a image is complete

    // =========================================================================================== phase B: t_b = relu(bn(W2b * t_a)), 3x3
    f32x16 accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) accB[r] = 0.f;
    {
        // X fragment of the m-tile: patch rows 2 mtB + (li >> 4), column li & 15; tap (dy, dx) = constant shift (dy + 1) PITCH + (dx + HX0) REC.
        // HPX == 16: the image is one patch wide -- a lane whose tap column falls outside reads the zero record
        const int xbase = (li >> 4) * PITCH + (li & 15) * REC + lk * 16 + mtB * 2 * PITCH;
        f16x8 xc[2][2];                           // [k half][hi | lo]
        auto xread = [&](int ks, f16x8 (&q)[2][2]) {
            const int chunk = ks / 9, tap = ks - chunk * 9;
            const int ky = tap / 3, kx = tap - ky * 3;                        // tap t = kh * 3 + kw at (kh - 1, kw - 1)   (pack_conv)
            int xo = xbase + chunk * TSLICE + ky * PITCH + (kx - 1 + HX0) * REC;
            if (HPX == 16) {
                const int col = (li & 15) + kx - 1;
                if ((unsigned)col > 15u) xo = ZERO_OFF + lk * 16;
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                q[kb][0] = *reinterpret_cast<const f16x8*>(smem + xo + kb * 32);
                q[kb][1] = *reinterpret_cast<const f16x8*>(smem + xo + kb * 32 + 64);
            }
        };
#pragma unroll 1
        for (int k0 = 0; k0 < TOTAL; k0 += 3) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int ks = k0 + d;
                f16x8 wq[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) wq[f] = rwf[d][f];
                if (ks + 3 < TOTAL) wfload(ks + 3, rwf[d]);
                xread(ks, xc);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[2 * kb], xc[kb][1], accB, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[2 * kb + 1], xc[kb][0], accB, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[2 * kb], xc[kb][0], accB, 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();                              // every wave is done with the t_a image (the t_b image and the 2c stage replace it)

    // weight loader of the last convolution: a stage = K-steps (2 st, 2 st + 1) of the 128 rows of chunk q: 4 float4 per thread
    unsigned wc_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) wc_off[j] = ((unsigned)(lrow + 64 * j) * (unsigned)F1 + (unsigned)lseg * 4u) * 4u;
    f32x4 rwc[2][2];
    auto wcload = [&](int q, int st) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                rwc[k2][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wc, wc_off[j] + (unsigned)(q * 128 * F1 * 4), (2 * st + k2) * 128, 0));
    };
    char* Wcs = smem + T2_BYTES;
    auto wcstore = [&]() {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(Wcs + k2 * 128 * WREC + w_dst + 64 * j * WREC) = rwc[k2][j];
    };
    wcload(0, 0);

    // ---- epilogue B: the t_b image replaces the t_a image (everyone is past the barrier behind the K loop)
    {
        char* dst = smem + ntB * T2SLICE + (mtB * 32 + li) * REC + lk * 8;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int c = ntB * 32 + 8 * gq + 4 * lk;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(ss + 2 * F1 + c);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(ss + 3 * F1 + c);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = relu_nan(fmaf(accB[4 * gq + e], sc[e], sh[e]));
            amax = range_note4(amax, v);
            uint2 hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<uint2*>(dst + gq * 16) = hi;
            *reinterpret_cast<uint2*>(dst + gq * 16 + 64) = lo;
        }
    }
    wcstore();
    __syncthreads();

    // =========================================================================================== phase C: out = relu(bn(W2c t_b) + x)
    {
        const int wm = wave >> 2, wn = wave & 3;  // half of the patch (TMC m-tiles), n-tile of the chunk
        const char* Af = smem + (wm * TMC * 32 + li) * REC + lk * 16;
        const char* Bf = Wcs + (wn * 32 + li) * WREC;
        int w_sw[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) w_sw[kb][hf] = ((kb * 2 + lk + 4 * hf) ^ ((li >> 1) & 7)) << 4;
        // epilogue ownership: thread = 4 consecutive channels c4 of rows r0 + 16 it of a pass's C tile (TMC * 32 rows = 2 TMC patch rows of 16 pixels):
        // row r0 + 16 it of pass h is patch pixel (2 h TMC + it, r0)
        float* Cs = reinterpret_cast<float*>(Wcs);
        const int c4 = (tid & 31) * 4, r0 = tid >> 5;
        constexpr int NIT = TMC * 2;
        for (int q = 0; q < NCH; ++q) {
            f32x4 rs[NIT];
            const unsigned obase = (unsigned)(((n * p.H + y0) * p.W + x0 + r0) * C + q * 128 + c4);      // elements: tensors are < 2^30 floats
            const unsigned orow = (unsigned)(p.W * C);
            auto rsload = [&](int h) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) rs[it] = *reinterpret_cast<const f32x4*>(p.x + (obase + (unsigned)(2 * h * TMC + it) * orow));
            };
            rsload(0);
            f32x16 acc[TMC];
#pragma unroll
            for (int i = 0; i < TMC; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            for (int st = 0; st < NST; ++st) {
                const bool more = st + 1 < NST || q + 1 < NCH;
                if (more) wcload(st + 1 < NST ? q : q + 1, st + 1 < NST ? st + 1 : 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k2 = u >> 1, kb = u & 1;
                    const f16x8 bh = *reinterpret_cast<const f16x8*>(Bf + k2 * 128 * WREC + w_sw[kb][0]);
                    const f16x8 bl = *reinterpret_cast<const f16x8*>(Bf + k2 * 128 * WREC + w_sw[kb][1]);
#pragma unroll
                    for (int i = 0; i < TMC; ++i) {
                        const f16x8 ah = *reinterpret_cast<const f16x8*>(Af + (2 * st + k2) * T2SLICE + i * 32 * REC + kb * 32);
                        const f16x8 al = *reinterpret_cast<const f16x8*>(Af + (2 * st + k2) * T2SLICE + i * 32 * REC + kb * 32 + 64);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i], 0, 0, 0);
                    }
                }
                __syncthreads();                  // everyone is done reading the weight stage
                if (st + 1 < NST) {
                    wcstore();
                    __syncthreads();
                }
            }
            // epilogue of the chunk through the (now free) weight stage: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
            const f32x4 sc = *reinterpret_cast<const f32x4*>(ss + 4 * F1 + q * 128 + c4);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(ss + 4 * F1 + C + q * 128 + c4);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (wm == h) {
#pragma unroll
                    for (int i = 0; i < TMC; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) Cs[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CLD + wn * 32 + li] = acc[i][r];
                }
                __syncthreads();
                f32x4 o[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(Cs + (r0 + 16 * it) * CLD + c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rs[it][e];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = relu_nan(v[e]);
                    amax = range_note4(amax, v);
                    o[it] = v;
                }
                if (h + 1 < 2) rsload(h + 1);             // the next pass's residual rows fly under this pass's stores
#pragma unroll
                for (int it = 0; it < NIT; ++it) *reinterpret_cast<f32x4*>(p.out + (obase + (unsigned)(2 * h * TMC + it) * orow)) = o[it];
                __syncthreads();                  // the C tile is free again (next pass / next chunk's weight stage)
            }
            if (q + 1 < NCH) {
                wcstore();
                __syncthreads();
            }
        }
    }
    range_commit(p.range_acc, amax);
}

}  // namespace

hipError_t launch_resblock8(const ResBlockParams& p, int F1, hipStream_t s)
{
    if (!resblock_supported(F1, p.H, p.W)) return hipErrorInvalidValue;
    const int grid = resblock_grid(F1, p.N, p.H, p.W);
    if (F1 == 64) hipLaunchKernelGGL((resblock8_kernel<64, 8, 18>), dim3(grid), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((resblock8_kernel<128, 4, 16>), dim3(grid), dim3(512), 0, s, p);
    return hipGetLastError();
}

}  // namespace p2p
