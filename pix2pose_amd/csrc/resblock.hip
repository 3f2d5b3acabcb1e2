// One ResNet identity bottleneck block as ONE kernel (gfx950, PREC_F16X3):
//
//     t_a = relu(bn(conv1x1(x)))      "2a"   C  -> F1      resnet50_mod.py:57-61
//     t_b = relu(bn(conv3x3(t_a)))    "2b"   F1 -> F1      resnet50_mod.py:63-66
//     out = relu(bn(conv1x1(t_b)) + x) "2c"  F1 -> C = 4 F1, residual add, ReLU      resnet50_mod.py:68-73
//
// As three launches (igemm.hip / igemm_halo.hip / igemm.hip) every block sends its activations across HBM three times: the 1x1 layers
// are bandwidth-bound (0.47 of the HBM peak, 12.8 % of a pass's kernel time) and only removing bytes helps them.  Here a workgroup owns a
// PY x 16 patch of one sample and keeps both intermediates in LDS:
//
//   phase A  2a on the (PY + 2) x HPX halo of the patch: the block input streams through LDS in 32-channel slices (split hi/lo by the
//            loader exactly as igemm.hip does), accumulators in registers; epilogue: BN + ReLU, ZERO for halo pixels outside the image
//            (the 3x3 convolution pads t_a, not x), split hi/lo, written as the [hi x32 | lo x32] record image igemm_halo.hip stages
//   phase B  2b from that image: (slice, tap) K order, only the weight tile moves per K-step; epilogue: BN + ReLU, split, second image
//   phase C  2c from the second image in 128-channel chunks; epilogue through an LDS C tile as in igemm.hip: BN, + x (re-read: it left
//            L2 long ago at this footprint, the memory-side cache serves it), ReLU, float4 stores of whole 512-byte pixel rows
//
// Bits: every output element is the same chain of v_mfma_f32_32x32x16_f16 over the same K-step order as in the three-launch route -- 2a:
// slices in order; 2b: (slice, tap); 2c: slices in order; (al bh, ah bl, ah bh) inside a step; the same epilogue expressions; the same
// cvt_pkrtz / round-to-nearest split of the fp32 intermediate -- so the routes agree bit for bit (tests/test_resblock_gpu.py), and a
// detection's bits do not depend on the batch it travels in (small launches keep the streaming route).  Phases A and B run the MFMA in
// transposed orientation (weights as the row operand): a lane then owns 4 consecutive channels of one pixel and the split image is
// written with 8-byte stores; swapping the operand roles changes no bit.
//
// Shapes: F1 = 64  (res2: 32x32x256): 8 x 16 patch, 10 x 18 halo (192 GEMM rows, 12 idle), LDS 70.7 KB
//         F1 = 128 (res3: 16x16x512): 4 x 16 patch, 6 x 16 halo -- the image is one patch wide, so the columns left and right of it are
//                  padding: lanes whose tap falls there read a zero record instead of a halo column --, LDS 71.8 KB
// Two workgroups per CU either way (their phases interleave: one's memory-bound phase A under the other's MFMA-bound phase B).
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

constexpr int cmax(int a, int b) { return a > b ? a : b; }

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

// DA / DB: K-steps of global loads in flight in phases A / B (register rings).  A K-step of these phases is 12 - 18 MFMAs per wave -- a few
// hundred cycles, a fraction of a loaded memory round trip -- so with the usual one-step-ahead prefetch every step waited for its operands.
// PROJ: the projection blocks res2a / res3a (resnet50_mod.py:76-118): the block input has CIN channels on a grid STRIDE times finer than the output
// (the 1x1 layers of a stride-2 block sample its even pixels), and the block's last 1x1 convolution carries the projection shortcut as a second K
// segment [t_b (F1) | x (CIN)] (model.hip: pack_merged_shortcut) instead of a residual add.
// WDIR: phase B takes its weight operand straight from global (the fragment-ordered copy of the 2b panel, ResBlockGroup::w2b_frag) into
// registers, DB K-steps ahead: no weight tile in LDS, no barrier in the K loop of phase B (the t_a image is read-only).  Same fragments, same
// MFMA order: same bits.
#ifdef P2P_RB_NO_WDIR
constexpr bool RB_WDIR = false;
#else
constexpr bool RB_WDIR = true;
#endif
// DBUF: phase A stages K-step s + 1 into the OTHER of two staging buffers before the MFMAs of K-step s: the split + LDS stores of the loader run
// under the matrix pipe and a K-step ends in ONE barrier (single-buffered: MFMAs | barrier | stores | barrier).  Same K order: same bits.
#ifndef P2P_RB_DA64
#define P2P_RB_DA64 2         // stages of global loads in flight in phase A of the F1 = 64 identity blocks (4 with the explicit schedule: 112 bytes of scratch per lane)
#endif
#ifndef P2P_RB_DA3A
#define P2P_RB_DA3A 2         // the same for the projection block res3a (4: 40 bytes of scratch per lane)
#endif
#ifdef P2P_RB_NO_PIPE
constexpr bool RB_PIPE = false;
#else
constexpr bool RB_PIPE = true;                  // phases B / C: the next K-step's fragment reads in front of this K-step's MFMAs (explicit schedule)
#endif
#ifdef P2P_RB_NO_DBUF
constexpr bool RB_DBUF = false;
#else
constexpr bool RB_DBUF = true;
#endif
template <int F1, int PY, int HPX, int DA, int DB, int KA, bool PRIV, int ABL = 0, int CIN = 4 * F1, int STRIDE = 1, bool PROJ = false, bool WDIR = RB_WDIR, bool DBUF = RB_DBUF, bool PIPE = RB_PIPE>
__global__ __launch_bounds__(256, 2) void resblock_kernel(const ResBlockParams p)
{
    constexpr int C = 4 * F1;                     // output channels
    constexpr int SA = CIN / 32;                  // K-steps of 2a
    constexpr int SX = CIN / 32;                  // PROJ: K-steps of the shortcut segment of the last convolution
    static_assert(PROJ || (CIN == C && STRIDE == 1), "identity blocks keep their shape");
    constexpr int SB = F1 / 32;                   // channel slices of t_a / t_b
    constexpr int NCH = C / 128;                  // 128-channel output chunks of 2c
    constexpr int HX0 = HPX == 18 ? 1 : 0;        // halo columns left of the patch (0: the image is one patch wide)
    constexpr int HPY = PY + 2, NPA = HPY * HPX, MA = (NPA + 31) / 32;
    constexpr int PITCH = (HPX * REC + 255) / 256 * 256;      // halo row pitch: a multiple of 16 slots (igemm_halo.hip: LDS layout)
    constexpr int TSLICE = HPY * PITCH, T_BYTES = SB * TSLICE;
    constexpr int NPIX = PY * 16, MT = PY / 2;    // patch pixels; their 32-row m-tiles (two patch rows each)
    constexpr int T2SLICE = NPIX * REC, T2_BYTES = SB * T2SLICE;
    constexpr int XS_BYTES = MA * 32 * REC;
    constexpr int WA_BYTES = F1 * WREC;
    // KA: K-steps per staged stage of phase A.  PRIV: phase B with wave-private weight tiles (every wave stages its OWN copy of its n-tile's 32
    // rows: no workgroup barrier in the K loop) instead of one shared tile per K-step between two barriers.
    constexpr int WB_BYTES = PRIV ? 4 * 32 * WREC : F1 * WREC;
    constexpr int WC_BYTES = 2 * 128 * WREC;      // two K-steps of a 128-row chunk of the 2c panel
    constexpr int NST = SB / 2;                   // such stages per chunk
    constexpr int TMC = 2, TNC = MT == 4 ? 2 : 1, WGMC = MT / TMC, WGNC = 4 / WGMC;      // phase-C wave tiling (igemm.hip's)
    constexpr int CLD = 128 + 4;
    constexpr int CS_BYTES = TMC * 32 * CLD * 4;
    constexpr int XC_BYTES = NPIX * REC;          // PROJ: a 32-channel slice of the block input at the patch's own pixels, staged per K-step
    constexpr int NBUF = DBUF ? 2 : KA;            // staging buffers of phase A
    static_assert(!DBUF || (KA == 1 && DA % 2 == 0), "double buffering: one K-step per stage, buffer parity = ring slot parity");
    constexpr int ZERO_OFF = cmax(cmax(NBUF * (XS_BYTES + WA_BYTES), T_BYTES + WB_BYTES),
                                  T2_BYTES + cmax(PROJ ? 128 * WREC + XC_BYTES : WC_BYTES, CS_BYTES));
    constexpr int SS_OFF = ZERO_OFF + 128, SS_FLOATS = 4 * F1 + 2 * C;      // the folded BatchNorm vectors: read by every epilogue, kept in LDS
    constexpr int SMEM = SS_OFF + SS_FLOATS * 4;
    static_assert(SMEM <= 80 * 1024, "two workgroups per CU");
    static_assert(MA % (F1 == 64 ? 2 : 1) == 0 && (F1 == 64 || F1 == 128) && (PY == 8 || PY == 4), "shapes");
    __shared__ __attribute__((aligned(256))) char smem[SMEM];
    static_assert(SS_FLOATS % 4 == 0, "float4 copies");

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (kept in an SGPR: it selects tiles and the fragment stream's offset -- as a vector value every load of phase B became a waterfall loop)
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
    // [s2a F1 | h2a F1 | s2b F1 | h2b F1 | s2c C | h2c C]: fetched once (a global load per use put a memory round trip in front of every
    // group of four channels in the epilogues)
    const float* ss = reinterpret_cast<const float*>(smem + SS_OFF);
    for (int i = tid * 4; i < SS_FLOATS; i += 1024)
        *reinterpret_cast<f32x4*>(smem + SS_OFF + i * 4) = *reinterpret_cast<const f32x4*>(p.grp[g].ss + i);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wa = __builtin_amdgcn_make_buffer_rsrc((void*)p.grp[g].w2a, 0, p.wa_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wb = __builtin_amdgcn_make_buffer_rsrc((void*)p.grp[g].w2b, 0, p.wb_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wc = __builtin_amdgcn_make_buffer_rsrc((void*)p.grp[g].w2c, 0, p.wc_bytes, 0x00020000);

    if (tid < 8) *reinterpret_cast<f32x4*>(smem + ZERO_OFF + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    // weight loader (all phases): rows (tid >> 3) + 32 j, 16-byte segment tid & 7, swizzled chunk
    const int lrow = tid >> 3, lseg = tid & 7;
    const int w_dst = lrow * WREC + ((lseg ^ ((lrow >> 1) & 7)) << 4);
    float amax = 0.f;                             // operand-range guard (kernels.h): t_a, t_b and the block output

    // =========================================================================================== phase A: t_a = relu(bn(W2a x)) on the halo
    f32x16 accA[3];                               // one n-tile (32 channels) x three m-tiles (96 halo pixels) per wave
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[i][r] = 0.f;
    constexpr int WNA = F1 / 32 >= 4 ? 4 : F1 / 32, WMA = 4 / WNA;       // waves along channels / along halo pixels
    static_assert(MA == 3 * WMA, "three m-tiles per wave");
    const int ntA = wave % WNA, mt0A = (wave / WNA) * 3;
    {
        char* Xs = smem;                           // [KA][MA * 32 records]
        char* Was = smem + NBUF * XS_BYTES;        // [KA | 2][F1 rows]
        unsigned x_off[MA];
        int x_dst[MA];
#pragma unroll
        for (int j = 0; j < MA; ++j) {
            // float4 idx = tid + 256 j -> quad idx & 7 of halo pixel perm(idx >> 3): consecutive octets of lanes take pixels hp and hp + 4
            // (two pixels per 16-lane store group collide unless they are 4 records apart, igemm_halo.hip)
            const int idx = tid + 256 * j;
            const int t8 = idx >> 3, q = idx & 7;
            const int hp = (t8 & ~7) | ((t8 & 1) << 2) | ((t8 >> 1) & 3);
            const int hy = hp / HPX, hx = hp - hy * HPX;
            const int iy = y0 - 1 + hy, ix = x0 - HX0 + hx;
            const bool ok = hp < NPA && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            x_off[j] = ok ? ((((unsigned)(n * p.H * STRIDE + iy * STRIDE) * (unsigned)(p.W * STRIDE) + (unsigned)(ix * STRIDE)) * (unsigned)CIN + (unsigned)(q * 4)) * 4u) : OOB;      // unsigned from the pixel index on: tensors up to 4 GB
            x_dst[j] = hp * REC + q * 8;
        }
        unsigned wa_off[F1 / 32];
#pragma unroll
        for (int j = 0; j < F1 / 32; ++j) wa_off[j] = ((unsigned)(lrow + 32 * j) * (unsigned)CIN + (unsigned)lseg * 4u) * 4u;
        constexpr int NSTG = SA / KA;              // stages of KA K-steps
        static_assert(SA % KA == 0 && NSTG % DA == 0, "ring depth divides the stages");
        f32x4 rx[DA][KA][MA], rw[DA][KA][F1 / 32];      // ring slot d holds stages congruent to d (mod DA); all indices are static after unrolling
        auto gload = [&](int stg, auto& qx, auto& qw) {
#pragma unroll
            for (int k2 = 0; k2 < KA; ++k2) {
                const int s = stg * KA + k2;
#pragma unroll
                for (int j = 0; j < MA; ++j)
                    qx[k2][j] = (ABL & 4) ? f32x4{1.f, 2.f, 3.f, 4.f}
                                          : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, x_off[j], s * 128, 0));      // (the range check takes the vector offset alone: OOB stays out of range, the slice travels in the scalar offset)
#pragma unroll
                for (int j = 0; j < F1 / 32; ++j) qw[k2][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wa, wa_off[j], s * 128, 0));
            }
        };
        auto lstore = [&](const auto& qx, const auto& qw, int buf = 0) {
#pragma unroll
            for (int k2 = 0; k2 < KA; ++k2) {
                const int b = DBUF ? buf : k2;
#pragma unroll
                for (int j = 0; j < MA; ++j) {
                    uint2 hi, lo;
                    split4(qx[k2][j], hi, lo);
                    *reinterpret_cast<uint2*>(Xs + b * XS_BYTES + x_dst[j]) = hi;
                    *reinterpret_cast<uint2*>(Xs + b * XS_BYTES + x_dst[j] + 64) = lo;
                }
#pragma unroll
                for (int j = 0; j < F1 / 32; ++j) *reinterpret_cast<f32x4*>(Was + b * WA_BYTES + w_dst + 32 * j * WREC) = qw[k2][j];
            }
        };
#pragma unroll
        for (int d = 0; d < DA; ++d) gload(d, rx[d], rw[d]);
        lstore(rx[0], rw[0]);
        __syncthreads();
        const char* Xf = Xs + (mt0A * 32 + li) * REC + lk * 16;
        const char* Wf = Was + (ntA * 32 + li) * WREC;
        int w_sw[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) w_sw[kb][hf] = ((kb * 2 + lk + 4 * hf) ^ ((li >> 1) & 7)) << 4;
        for (int g0 = 0; g0 < NSTG; g0 += DA) {
#pragma unroll
            for (int d = 0; d < DA; ++d) {
                const int stg = g0 + d;                                       // the staged stage; its ring slot d is free again
                if (stg + DA < NSTG) gload(stg + DA, rx[d], rw[d]);
                if constexpr (PIPE && KA == 1 && ABL == 0) {
                    // explicit schedule: all sixteen fragment reads of the K-step in front of its eighteen MFMAs (the scheduler sinks each read to its
                    // first use: read, wait, MFMA ...); the loader's split + stores of the next stage are left to interleave with the MFMAs
                    const int b = DBUF ? (d & 1) : 0;
                    f16x8 wf[2][2], xf[2][3][2];
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        wf[kb][0] = *reinterpret_cast<const f16x8*>(Wf + b * WA_BYTES + w_sw[kb][0]);
                        wf[kb][1] = *reinterpret_cast<const f16x8*>(Wf + b * WA_BYTES + w_sw[kb][1]);
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            xf[kb][i][0] = *reinterpret_cast<const f16x8*>(Xf + b * XS_BYTES + i * 32 * REC + kb * 32);
                            xf[kb][i][1] = *reinterpret_cast<const f16x8*>(Xf + b * XS_BYTES + i * 32 * REC + kb * 32 + 64);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (DBUF && stg + 1 < NSTG) lstore(rx[(d + 1) % DA], rw[(d + 1) % DA], (d + 1) & 1);      // (its buffer was last read before the previous barrier)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            accA[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kb][0], xf[kb][i][1], accA[i], 0, 0, 0);
                            accA[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kb][1], xf[kb][i][0], accA[i], 0, 0, 0);
                            accA[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kb][0], xf[kb][i][0], accA[i], 0, 0, 0);
                        }
                } else {
                if (DBUF && stg + 1 < NSTG) lstore(rx[(d + 1) % DA], rw[(d + 1) % DA], (d + 1) & 1);
#pragma unroll
                for (int k2 = 0; k2 < KA; ++k2)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        const int b = DBUF ? (d & 1) : k2;
                        const f16x8 wh = *reinterpret_cast<const f16x8*>(Wf + b * WA_BYTES + w_sw[kb][0]);
                        const f16x8 wl = *reinterpret_cast<const f16x8*>(Wf + b * WA_BYTES + w_sw[kb][1]);
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            const f16x8 xh = *reinterpret_cast<const f16x8*>(Xf + b * XS_BYTES + i * 32 * REC + kb * 32);
                            const f16x8 xl = *reinterpret_cast<const f16x8*>(Xf + b * XS_BYTES + i * 32 * REC + kb * 32 + 64);
                            if (ABL & 32) { accA[i][0] += (float)wh[0] + (float)wl[0] + (float)xh[0] + (float)xl[0]; continue; }
                            accA[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, accA[i], 0, 0, 0);      // (al bh, ah bl, ah bh) with a = activation, b = weight,
                            accA[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, accA[i], 0, 0, 0);      // operand roles swapped: D[channel][pixel]
                            accA[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, accA[i], 0, 0, 0);
                        }
                    }
                }
                __syncthreads();
                if (!DBUF && stg + 1 < NSTG) {
                    lstore(rx[(d + 1) % DA], rw[(d + 1) % DA]);
                    __syncthreads();
                }
            }
        }
    }

    // 2b weights.  PRIV: every wave streams its OWN n-tile's 32 rows (two lanes per 128-byte row) through a private 4 KB staging tile, so the
    // K loop of phase B has no workgroup barrier (LDS operations of one wave execute in order, the t_a image is read-only; waves that share an
    // n-tile -- F1 = 64: two m-halves -- fetch it twice).  Otherwise: one shared F1-row tile per K-step, two barriers per step.
    constexpr int WNB = F1 / 32 >= 4 ? 4 : F1 / 32, WMB = 4 / WNB;
    const int ntB = wave % WNB, mt0B = (wave / WNB) * 2;
    constexpr int TOTAL = SB * 9;
    static_assert(TOTAL % DB == 0, "ring depth divides the K-steps");
    constexpr int NWB = PRIV ? 4 : F1 / 32;       // float4 per thread and K-step
    unsigned wb_off[NWB];
    int wb_dst[NWB];
#pragma unroll
    for (int j = 0; j < NWB; ++j) {
        if (PRIV) {
            wb_off[j] = ((unsigned)(ntB * 32 + (lane >> 1)) * (unsigned)(9 * F1) + (unsigned)(lane & 1) * 16u) * 4u + 16u * j;
            wb_dst[j] = (lane >> 1) * WREC + ((((lane & 1) * 4 + j) ^ ((lane >> 2) & 7)) << 4);
        } else {
            wb_off[j] = ((unsigned)(lrow + 32 * j) * (unsigned)(9 * F1) + (unsigned)lseg * 4u) * 4u;
            wb_dst[j] = w_dst + 32 * j * WREC;
        }
    }
    f32x4 rwb[DB][NWB];
    auto wbload = [&](int ks, auto& qw) {
        const int chunk = ks / 9, tap = ks - chunk * 9;                       // K order (slice, tap); the panel's is (tap, slice)
        const int koff = (tap * SB + chunk) * 128;
#pragma unroll
        for (int j = 0; j < NWB; ++j)
            qw[j] = (ABL & 8) ? f32x4{1.f, 2.f, 3.f, 4.f} : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wb, wb_off[j], koff, 0));
    };
    char* Wbs = smem + T_BYTES + (PRIV ? wave * (32 * WREC) : 0);
    auto wbstore = [&](const auto& qw) {
#pragma unroll
        for (int j = 0; j < NWB; ++j) *reinterpret_cast<f32x4*>(Wbs + wb_dst[j]) = qw[j];
    };
    const __amdgpu_buffer_rsrc_t rs_wbf = __builtin_amdgcn_make_buffer_rsrc((void*)p.grp[g].w2b_frag, 0, (unsigned)(F1 * 9 * F1 * 4), 0x00020000);
    f16x8 rwf[DB][4];                             // WDIR: (k half 0 hi, lo, k half 1 hi, lo) of the K-steps in flight
    auto wfload = [&](int ks, f16x8 (&q)[4]) {
        const int chunk = ks / 9, tap = ks - chunk * 9;                       // K order (slice, tap); the panel's is (tap, slice)
        const unsigned base = (unsigned)((ntB * TOTAL + tap * SB + chunk) * 4096);
#pragma unroll
        for (int f = 0; f < 4; ++f) q[f] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_wbf, (unsigned)lane * 16u + f * 1024u, base, 0));
    };
#pragma unroll
    for (int d = 0; d < DB; ++d) {
        if (WDIR) wfload(d, rwf[d]);
        else wbload(d, rwb[d]);
    }

    // ---- epilogue A: lane = halo pixel mt * 32 + li, channels ntA * 32 + 8 g + 4 lk + {0..3}; the image replaces the staging buffers
    //      (everyone left the K loop through its last barrier)
    {
        char* T = smem + ntA * TSLICE;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int hp = (mt0A + i) * 32 + li;
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
    if (!WDIR) wbstore(rwb[0]);
    __syncthreads();                              // the t_a image is complete

    // =========================================================================================== phase B: t_b = relu(bn(W2b * t_a)), 3x3
    f32x16 accB[2];                               // one n-tile x two m-tiles (four patch rows) per wave
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[i][r] = 0.f;
    static_assert(MT == 2 * WMB, "two m-tiles per wave");
    {
        // X fragment of m-tile i: patch rows 2 i + (li >> 4), column li & 15; tap (dy, dx) = constant shift (dy + 1) PITCH + (dx + HX0) REC.
        // HPX == 16: the image is one patch wide -- a lane whose tap column falls outside reads the zero record
        const int xbase = (li >> 4) * PITCH + (li & 15) * REC + lk * 16;
        const char* Wf = Wbs + ((PRIV ? 0 : ntB * 32) + li) * WREC;
        int w_sw[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) w_sw[kb][hf] = ((kb * 2 + lk + 4 * hf) ^ ((li >> 1) & 7)) << 4;
        // X fragment offsets of K-step ks (both m-tiles)
        auto xoff = [&](int ks, int (&xo)[2]) {
            const int chunk = ks / 9, tap = ks - chunk * 9;
            const int ky = tap / 3, kx = tap - ky * 3;                        // tap t = kh * 3 + kw at (kh - 1, kw - 1)   (pack_conv)
            const int shift = chunk * TSLICE + ky * PITCH + (kx - 1 + HX0) * REC;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                xo[i] = xbase + shift + (mt0B + i) * 2 * PITCH;
                if (HPX == 16) {
                    const int col = (li & 15) + kx - 1;
                    if ((unsigned)col > 15u) xo[i] = ZERO_OFF + lk * 16;
                }
            }
        };
        if constexpr (WDIR && PIPE) {
            // software pipeline: the eight X fragments of K-step ks + 1 and the weight fragments of K-step ks + DB are requested BEFORE the twelve
            // MFMAs of K-step ks (left to itself the scheduler sinks every fragment read to its first use: read, wait, one to three MFMAs, read ...)
            f16x8 xc[2][2][2], xn[2][2][2];               // [k half][m-tile][hi | lo]
            auto xread = [&](int ks, f16x8 (&q)[2][2][2]) {
                int xo[2];
                xoff(ks, xo);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        q[kb][i][0] = *reinterpret_cast<const f16x8*>(smem + xo[i] + kb * 32);
                        q[kb][i][1] = *reinterpret_cast<const f16x8*>(smem + xo[i] + kb * 32 + 64);
                    }
            };
            xread(0, xc);
            for (int k0 = 0; k0 < TOTAL; k0 += DB) {
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const int ks = k0 + d;
                    f16x8 wq[4];
#pragma unroll
                    for (int f = 0; f < 4; ++f) wq[f] = rwf[d][f];
                    if (ks + DB < TOTAL) wfload(ks + DB, rwf[d]);
                    if (ks + 1 < TOTAL) xread(ks + 1, xn);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            accB[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[2 * kb], xc[kb][i][1], accB[i], 0, 0, 0);
                            accB[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[2 * kb + 1], xc[kb][i][0], accB[i], 0, 0, 0);
                            accB[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[2 * kb], xc[kb][i][0], accB[i], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int h = 0; h < 2; ++h) xc[kb][i][h] = xn[kb][i][h];
                }
            }
        } else
        for (int k0 = 0; k0 < TOTAL; k0 += DB) {
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                const int ks = k0 + d;
                const int chunk = ks / 9, tap = ks - chunk * 9;
                f16x8 wq[4];
                if (WDIR) {
#pragma unroll
                    for (int f = 0; f < 4; ++f) wq[f] = rwf[d][f];
                    if (ks + DB < TOTAL) wfload(ks + DB, rwf[d]);
                } else if (ks + DB < TOTAL) wbload(ks + DB, rwb[d]);
                (void)chunk; (void)tap;
                int xo[2];
                xoff(ks, xo);
                f16x8 wh[2], wl[2], xh[2][2], xl[2][2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    if (WDIR) { wh[kb] = wq[2 * kb]; wl[kb] = wq[2 * kb + 1]; }
                    else {
                        wh[kb] = *reinterpret_cast<const f16x8*>(Wf + w_sw[kb][0]);
                        wl[kb] = *reinterpret_cast<const f16x8*>(Wf + w_sw[kb][1]);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        xh[kb][i] = *reinterpret_cast<const f16x8*>(smem + xo[i] + kb * 32);
                        xl[kb][i] = *reinterpret_cast<const f16x8*>(smem + xo[i] + kb * 32 + 64);
                    }
                }
                // PRIV: the private tile's fragments are in flight (LDS serves a wave in order): the next K-step's tile may follow them
                if (PRIV && !WDIR && ks + 1 < TOTAL) wbstore(rwb[(d + 1) % DB]);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (ABL & 16) { accB[i][0] += (float)wh[kb][0] + (float)wl[kb][0] + (float)xh[kb][i][0] + (float)xl[kb][i][0]; continue; }
                        accB[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[kb], xl[kb][i], accB[i], 0, 0, 0);
                        accB[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[kb], xh[kb][i], accB[i], 0, 0, 0);
                        accB[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[kb], xh[kb][i], accB[i], 0, 0, 0);
                    }
                if (!PRIV && !WDIR) {
                    __syncthreads();              // everyone is done reading the shared weight tile
                    if (ks + 1 < TOTAL) {
                        wbstore(rwb[(d + 1) % DB]);
                        __syncthreads();
                    }
                }
            }
        }
    }
    if (PRIV || WDIR) __syncthreads();            // every wave is done with the t_a image (the t_b image and the 2c stage replace it)

    // weight loader of the last convolution.  Identity blocks: a stage = K-steps (2 st, 2 st + 1) of the 128 rows of chunk q: 8 float4 per thread.
    // Projection blocks: a stage = ONE K-step of chunk q (its K runs over [t_b | x]: row stride F1 + CIN), and the x slice of the K-step beside it.
    constexpr int K2C = PROJ ? F1 + CIN : F1;
    unsigned wc_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wc_off[j] = ((unsigned)(lrow + 32 * j) * (unsigned)K2C + (unsigned)lseg * 4u) * 4u;
    f32x4 rwc[2][4];
    auto wcload = [&](int q, int st) {
#pragma unroll
        for (int k2 = 0; k2 < (PROJ ? 1 : 2); ++k2)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                rwc[k2][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wc, wc_off[j] + (unsigned)(q * 128 * K2C * 4), ((PROJ ? st : 2 * st) + k2) * 128, 0));
    };
    char* Wcs = smem + T2_BYTES;
    auto wcstore = [&]() {
#pragma unroll
        for (int k2 = 0; k2 < (PROJ ? 1 : 2); ++k2)
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(Wcs + k2 * 128 * WREC + w_dst + 32 * j * WREC) = rwc[k2][j];
    };
    wcload(0, 0);

    // ---- epilogue B: the t_b image replaces the t_a image (everyone is past the last barrier of the K loop)
    {
        char* T2 = smem + ntB * T2SLICE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            char* dst = T2 + ((mt0B + i) * 32 + li) * REC + lk * 8;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int c = ntB * 32 + 8 * gq + 4 * lk;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(ss + 2 * F1 + c);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(ss + 3 * F1 + c);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = relu_nan(fmaf(accB[i][4 * gq + e], sc[e], sh[e]));
                amax = range_note4(amax, v);
                uint2 hi, lo;
                split4(v, hi, lo);
                *reinterpret_cast<uint2*>(dst + gq * 16) = hi;
                *reinterpret_cast<uint2*>(dst + gq * 16 + 64) = lo;
            }
        }
    }
    wcstore();
    __syncthreads();

    if constexpr (PROJ) {
    // =========================================================================================== phase C (projection): out = relu(bn([W2c | W1] [t_b ; x]))
    // K-outer, chunk-inner: every 128-channel output chunk keeps its accumulators, so a K-step's activation fragments (from the t_b image, or from
    // the x slice staged for it) are read once for all chunks; per (K-step, chunk) only the 16 KB weight tile moves.  K order per output element:
    // t_b slices, then x slices -- igemm.hip's order over the two segments.
        constexpr int KC = SB + SX, NSTAGE = KC * NCH;
        const int wm = wave / WGNC, wn = wave % WGNC;
        char* Xcs = Wcs + 128 * WREC;
        const char* Bf = Wcs + (wn * TNC * 32 + li) * WREC;
        int w_sw[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) w_sw[kb][hf] = ((kb * 2 + lk + 4 * hf) ^ ((li >> 1) & 7)) << 4;
        // x loader: float4 idx = tid + 256 j -> quad idx & 7 of patch pixel perm(idx >> 3) (the store pattern of the halo loaders)
        constexpr int XP = NPIX * 8 / 256;
        unsigned xc_off[XP];
        int xc_dst[XP];
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int idx = tid + 256 * j;
            const int t8 = idx >> 3, q = idx & 7;
            const int px = (t8 & ~7) | ((t8 & 1) << 2) | ((t8 >> 1) & 3);
            xc_off[j] = (((unsigned)(n * p.H * STRIDE + (y0 + (px >> 4)) * STRIDE) * (unsigned)(p.W * STRIDE) + (unsigned)((x0 + (px & 15)) * STRIDE)) * (unsigned)CIN + (unsigned)(q * 4)) * 4u;
            xc_dst[j] = px * REC + q * 8;
        }
        f32x4 rxc[XP];
        auto xcload = [&](int sx) {
#pragma unroll
            for (int j = 0; j < XP; ++j) rxc[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xc_off[j], sx * 128, 0));
        };
        auto xcstore = [&]() {
#pragma unroll
            for (int j = 0; j < XP; ++j) {
                uint2 hi, lo;
                split4(rxc[j], hi, lo);
                *reinterpret_cast<uint2*>(Xcs + xc_dst[j]) = hi;
                *reinterpret_cast<uint2*>(Xcs + xc_dst[j] + 64) = lo;
            }
        };
        f32x16 acc[NCH][TMC][TNC];
#pragma unroll
        for (int q = 0; q < NCH; ++q)
#pragma unroll
            for (int i = 0; i < TMC; ++i)
#pragma unroll
                for (int j = 0; j < TNC; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;
        // (stage 0's weight tile was stored with the t_b image; its activations come from that image)
#pragma unroll 1
        for (int ks = 0; ks < KC; ++ks) {
            const char* Af = (ks < SB ? smem + ks * T2SLICE : Xcs) + (wm * TMC * 32 + li) * REC + lk * 16;
            f16x8 ah[2][TMC], al[2][TMC];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < TMC; ++i) {
                    ah[kb][i] = *reinterpret_cast<const f16x8*>(Af + i * 32 * REC + kb * 32);
                    al[kb][i] = *reinterpret_cast<const f16x8*>(Af + i * 32 * REC + kb * 32 + 64);
                }
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int stage = ks * NCH + q;
                const bool more = stage + 1 < NSTAGE;
                const bool next_ks = q + 1 == NCH;                       // the next stage starts K-step ks + 1
                if (more) {
                    wcload(next_ks ? 0 : q + 1, next_ks ? ks + 1 : ks);
                    if (next_ks && ks + 1 >= SB) xcload(ks + 1 - SB);
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    f16x8 bh[TNC], bl[TNC];
#pragma unroll
                    for (int j = 0; j < TNC; ++j) {
                        bh[j] = *reinterpret_cast<const f16x8*>(Bf + j * 32 * WREC + w_sw[kb][0]);
                        bl[j] = *reinterpret_cast<const f16x8*>(Bf + j * 32 * WREC + w_sw[kb][1]);
                    }
#pragma unroll
                    for (int i = 0; i < TMC; ++i)
#pragma unroll
                        for (int j = 0; j < TNC; ++j) {
                            acc[q][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kb][i], bh[j], acc[q][i][j], 0, 0, 0);
                            acc[q][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb][i], bl[j], acc[q][i][j], 0, 0, 0);
                            acc[q][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb][i], bh[j], acc[q][i][j], 0, 0, 0);
                        }
                }
                __syncthreads();                  // everyone is done with the weight tile (and, at the end of a K-step, with its x slice)
                if (more) {
                    wcstore();
                    if (next_ks && ks + 1 >= SB) xcstore();
                    __syncthreads();
                }
            }
        }
        // epilogue: chunk after chunk through the (now free) stage region
        float* Cs = reinterpret_cast<float*>(Wcs);
        const int c4 = (tid & 31) * 4, r0 = tid >> 5;
        constexpr int NIT = TMC * 32 / 8;
        const unsigned orow = (unsigned)(p.W * C);
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const unsigned obase = (unsigned)(((n * p.H + y0) * p.W + x0 + r0) * C + q * 128 + c4);
            const f32x4 sc = *reinterpret_cast<const f32x4*>(ss + 4 * F1 + q * 128 + c4);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(ss + 4 * F1 + C + q * 128 + c4);
#pragma unroll
            for (int h = 0; h < WGMC; ++h) {
                if (wm == h) {
#pragma unroll
                    for (int i = 0; i < TMC; ++i)
#pragma unroll
                        for (int j = 0; j < TNC; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                Cs[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CLD + (wn * TNC + j) * 32 + li] = acc[q][i][j][r];
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(Cs + (r0 + 8 * it) * CLD + c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = relu_nan(fmaf(v[e], sc[e], sh[e]) + 0.f);      // igemm.hip's epilogue with no residual: + 0
                    amax = range_note4(amax, v);
                    *reinterpret_cast<f32x4*>(p.out + (obase + (unsigned)(4 * h + (it >> 1)) * orow + (unsigned)(8 * (it & 1) * C))) = v;
                }
                __syncthreads();
            }
        }
    } else {
    // =========================================================================================== phase C: out = relu(bn(W2c t_b) + x)
    {
        const int wm = wave / WGNC, wn = wave % WGNC;
        const char* Af = smem + (wm * TMC * 32 + li) * REC + lk * 16;
        const char* Bf = Wcs + (wn * TNC * 32 + li) * WREC;
        int w_sw[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) w_sw[kb][hf] = ((kb * 2 + lk + 4 * hf) ^ ((li >> 1) & 7)) << 4;
        // epilogue ownership (igemm.hip): thread = 4 consecutive channels c4 of rows r0 + 8 it of the pass's 64-row C tile
        float* Cs = reinterpret_cast<float*>(Wcs);
        const int c4 = (tid & 31) * 4, r0 = tid >> 5;
        constexpr int NIT = TMC * 32 / 8;
        for (int q = 0; q < NCH; ++q) {
            // residual rows of this chunk: requested before the MFMAs.  Row r0 + 8 it of pass h is patch pixel (4 h + (it >> 1), r0 + 8 (it & 1))
            // (pass 0's rows now; a second pass's rows while pass 0 is stored)
            f32x4 rs[NIT];
            const unsigned obase = (unsigned)(((n * p.H + y0) * p.W + x0 + r0) * C + q * 128 + c4);      // elements: tensors are < 2^30 floats
            const unsigned orow = (unsigned)(p.W * C);
            auto rsload = [&](int h) {
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    rs[it] = (ABL & 1) ? f32x4{0.f, 0.f, 0.f, 0.f}
                                       : *reinterpret_cast<const f32x4*>(p.x + (obase + (unsigned)(4 * h + (it >> 1)) * orow + (unsigned)(8 * (it & 1) * C)));
            };
            rsload(0);
            f32x16 acc[TMC][TNC];
#pragma unroll
            for (int i = 0; i < TMC; ++i)
#pragma unroll
                for (int j = 0; j < TNC; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            for (int st = 0; st < NST; ++st) {
                const bool more = st + 1 < NST || q + 1 < NCH;
                if (more) wcload(st + 1 < NST ? q : q + 1, st + 1 < NST ? st + 1 : 0);
                if constexpr (PIPE && ABL == 0) {
                    // the stage's four (K-step, k half) sub-steps as a software pipeline: the fragments of sub-step u + 1 are requested before the MFMAs of u
                    f16x8 fa[2][TMC][2], fb[2][TNC][2];       // [buffer][tile][hi | lo]
                    auto fread = [&](int u, f16x8 (&qa)[TMC][2], f16x8 (&qb)[TNC][2]) {
                        const int k2 = u >> 1, kb = u & 1;
#pragma unroll
                        for (int i = 0; i < TMC; ++i) {
                            qa[i][0] = *reinterpret_cast<const f16x8*>(Af + (2 * st + k2) * T2SLICE + i * 32 * REC + kb * 32);
                            qa[i][1] = *reinterpret_cast<const f16x8*>(Af + (2 * st + k2) * T2SLICE + i * 32 * REC + kb * 32 + 64);
                        }
#pragma unroll
                        for (int j = 0; j < TNC; ++j) {
                            qb[j][0] = *reinterpret_cast<const f16x8*>(Bf + k2 * 128 * WREC + j * 32 * WREC + w_sw[kb][0]);
                            qb[j][1] = *reinterpret_cast<const f16x8*>(Bf + k2 * 128 * WREC + j * 32 * WREC + w_sw[kb][1]);
                        }
                    };
                    fread(0, fa[0], fb[0]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (u + 1 < 4) fread(u + 1, fa[(u + 1) & 1], fb[(u + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < TMC; ++i)
#pragma unroll
                            for (int j = 0; j < TNC; ++j) {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u & 1][i][1], fb[u & 1][j][0], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u & 1][i][0], fb[u & 1][j][1], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u & 1][i][0], fb[u & 1][j][0], acc[i][j], 0, 0, 0);
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        f16x8 ah[TMC], al[TMC], bh[TNC], bl[TNC];
#pragma unroll
                        for (int i = 0; i < TMC; ++i) {
                            ah[i] = *reinterpret_cast<const f16x8*>(Af + (2 * st + k2) * T2SLICE + i * 32 * REC + kb * 32);
                            al[i] = *reinterpret_cast<const f16x8*>(Af + (2 * st + k2) * T2SLICE + i * 32 * REC + kb * 32 + 64);
                        }
#pragma unroll
                        for (int j = 0; j < TNC; ++j) {
                            bh[j] = *reinterpret_cast<const f16x8*>(Bf + k2 * 128 * WREC + j * 32 * WREC + w_sw[kb][0]);
                            bl[j] = *reinterpret_cast<const f16x8*>(Bf + k2 * 128 * WREC + j * 32 * WREC + w_sw[kb][1]);
                        }
#pragma unroll
                        for (int i = 0; i < TMC; ++i)
#pragma unroll
                            for (int j = 0; j < TNC; ++j) {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
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
            for (int h = 0; h < WGMC; ++h) {
                if (wm == h) {
#pragma unroll
                    for (int i = 0; i < TMC; ++i)
#pragma unroll
                        for (int j = 0; j < TNC; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                Cs[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * CLD + (wn * TNC + j) * 32 + li] = acc[i][j][r];
                }
                __syncthreads();
                f32x4 o[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(Cs + (r0 + 8 * it) * CLD + c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rs[it][e];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = relu_nan(v[e]);
                    amax = range_note4(amax, v);
                    o[it] = v;
                }
                if (h + 1 < WGMC) rsload(h + 1);          // the next pass's residual rows fly under this pass's stores
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const f32x4 v = o[it];
                    if (!(ABL & 2) || v[0] == 12345.678f)
                        *reinterpret_cast<f32x4*>(p.out + (obase + (unsigned)(4 * h + (it >> 1)) * orow + (unsigned)(8 * (it & 1) * C))) = v;
                }
                __syncthreads();                  // the C tile is free again (next pass / next chunk's weight stage)
            }
            if (q + 1 < NCH) {
                wcstore();
                __syncthreads();
            }
        }
    }
    }
    range_commit(p.range_acc, amax);
}

}  // namespace

// (F1, H, W) of the ResNet-50 front's identity blocks: res2b/c (64, 32, 32), res3b/c/d (128, 16, 16)
bool resblock_supported(int F1, int H, int W)
{
    return (F1 == 64 && H % 8 == 0 && W % 16 == 0 && W > 16) || (F1 == 128 && H % 4 == 0 && W == 16);
}

int resblock_grid(int F1, int N, int H, int W) { return N * (H / (F1 == 64 ? 8 : 4)) * (W / 16); }

// Projection blocks of the ResNet-50 front: res2a (F1 = 64, 64 input channels, stride 1, 32x32) and res3a (F1 = 128, 256 input channels on the
// 32x32 grid, stride 2, 16x16 output).  p.H / p.W are the OUTPUT grid; w2c = the merged [2c | shortcut] panel.
hipError_t launch_resproj(const ResBlockParams& p, int F1, hipStream_t s)
{
    if (!resblock_supported(F1, p.H, p.W)) return hipErrorInvalidValue;
    const int grid = resblock_grid(F1, p.N, p.H, p.W);
    if (F1 == 64) hipLaunchKernelGGL((resblock_kernel<64, 8, 18, 2, 3, 1, false, 0, 64, 1, true>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((resblock_kernel<128, 4, 16, P2P_RB_DA3A, 4, 1, false, 0, 256, 2, true>), dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_resblock(const ResBlockParams& p, int F1, hipStream_t s)
{
    if (!resblock_supported(F1, p.H, p.W)) return hipErrorInvalidValue;
    const int grid = resblock_grid(F1, p.N, p.H, p.W);
    // Development builds: P2P_RB_VARIANT = KA (1 | 2) + 4 * PRIV picks among bit-identical forms of phases A / B, P2P_RB_ABL a TIMING ablation of the
    // default form (results are garbage): 1 no residual loads, 2 no output stores, 4 no input loads, 8 no 2b weight loads, 16 / 32 no MFMAs in phase
    // B / A.  The shipped library holds the default form only.
#define P2P_RB_LAUNCH(KA_, DA_, PRIV_, ABL_)                                                                                              \
    do {                                                                                                                                  \
        if (F1 == 64) hipLaunchKernelGGL((resblock_kernel<64, 8, 18, (DA_ == 4 ? P2P_RB_DA64 : DA_), 3, KA_, PRIV_, ABL_>), dim3(grid), dim3(256), 0, s, p);      \
        else hipLaunchKernelGGL((resblock_kernel<128, 4, 16, DA_, 4, KA_, PRIV_, ABL_>), dim3(grid), dim3(256), 0, s, p);              \
    } while (0)
#ifdef P2P_DEV_SWITCHES
    static const int variant = dev_env("P2P_RB_VARIANT") ? atoi(dev_env("P2P_RB_VARIANT")) : 1;
    static const int abl = dev_env("P2P_RB_ABL") ? atoi(dev_env("P2P_RB_ABL")) : 0;
    switch (abl) {
    case 1: P2P_RB_LAUNCH(1, 4, false, 1); return hipGetLastError();
    case 2: P2P_RB_LAUNCH(1, 4, false, 2); return hipGetLastError();
    case 3: P2P_RB_LAUNCH(1, 4, false, 3); return hipGetLastError();
    case 4: P2P_RB_LAUNCH(1, 4, false, 4); return hipGetLastError();
    case 7: P2P_RB_LAUNCH(1, 4, false, 7); return hipGetLastError();
    case 8: P2P_RB_LAUNCH(1, 4, false, 8); return hipGetLastError();
    case 16: P2P_RB_LAUNCH(1, 4, false, 16); return hipGetLastError();
    case 32: P2P_RB_LAUNCH(1, 4, false, 32); return hipGetLastError();
    case 48: P2P_RB_LAUNCH(1, 4, false, 48); return hipGetLastError();
    default: break;
    }
    switch (variant) {
    case 5: P2P_RB_LAUNCH(1, 4, true, 0); return hipGetLastError();
    default: break;
    }
#endif
    P2P_RB_LAUNCH(1, 4, false, 0);
#undef P2P_RB_LAUNCH
    return hipGetLastError();
}

}  // namespace p2p
