// The two output heads of the generator (Conv2DTranspose 5x5/2 128->3 tanh, 128->1 sigmoid;
// reference pix2pose_model/ae_model.py:233-236) as one halo-tiled kernel for gfx950.
//
// The heads are a 3x3-tap convolution over the 64x64 grid with 16 outputs (4 sub-pixel phases x
// (x, y, z, prob)), K = 9 x 128.  As an implicit GEMM with only 16 output columns the layer is pure
// operand traffic: every input pixel is gathered nine times (once per tap) and the generic kernel
// (igemm.hip) moved 3.6 GB per launch for a 0.54 GB tensor.  Here a workgroup owns 16 full-width grid rows of one
// sample and, per 32-channel slice, sweeps them with a rolling window of 6 rows x 66 columns in LDS (split into f16
// hi / lo halves on the way, like the igemm loader); all nine taps read their operands from that image.
// HBM traffic = the tensor 1.125 times (18 rows fetched per 16 rows of output) + the output.
//
// Arithmetic: PREC_F16X3 only (v_mfma_f32_16x16x32_f16, three products per block, fp32 accumulate);
// the GEMM is taken transposed (rows = the 16 outputs, columns = 16 consecutive pixels) so that a lane
// ends up with (x, y, z, prob) of one output pixel and stores a float4.  PREC_F32 models keep the
// generic kernel.
#include "kernels.h"
#include <cstdlib>

namespace p2p {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int HEADS_TH = 4;                 // grid rows per step (one per wave)
constexpr int HEADS_TILES_BIG = 4;          // steps a workgroup walks: 16 consecutive grid rows of one sample (template HEADS_TILES; 1 for small launches)
constexpr int HEADS_W = 64;                 // grid width (full rows: the x halo is the zero padding)
constexpr int HEADS_WP = HEADS_W + 2;
constexpr int HEADS_RING = HEADS_TH + 2;    // LDS row ring: the 6 input rows a step reads; 4 of them are replaced per step
constexpr int HEADS_CIN = 128;
constexpr int HEADS_CHUNKS = HEADS_CIN / 32;
constexpr int HEADS_LOADS = HEADS_TH * HEADS_W * 8 / 256;   // float4 loads per thread per step (8)

// LDS image of a 32-channel slice: EIGHT PLANES -- the four 16-byte k-chunks of the hi halves, then of the lo halves --
// each holding one 16-byte slot per pixel of the 6 x 66 ring.  The B operand of v_mfma_f32_16x16x32_f16 is read by lane
// (pixel li, k-chunk lg); ds_read_b128 serves lanes in 16-lane groups that mix k-chunks 0 and 1 (or 2 and 3) of DIFFERENT
// pixels ({0-3,12-15 | 20-27}: MI355X_MICROARCH.md, LDS), so a per-pixel record [hi x32 | lo x32] puts chunk 1 of pixel
// li + 4.. on the slots chunk 0 of pixel li uses (2-way on 7 of 8 slots: SQ_LDS_BANK_CONFLICT was half of the LDS cycles
// and the kernel was bound by its LDS reads).  With one plane per chunk a group reads 8 + 8 consecutive slots of two planes
// whose bases differ by a multiple of 256 B: conflict-free.  Planes 2,3 start 32 B later than a multiple of 256 B so that
// the ds_write_b64 stores of a pixel's quads (two per plane) spread over the 32 store banks (2-way instead of 4-way).
constexpr int HEADS_PLANE = HEADS_RING * HEADS_WP * 16;     // 6336 B
constexpr int HEADS_P1 = 6400, HEADS_P2 = 12832, HEADS_P3 = HEADS_P2 + 6400, HEADS_LO = HEADS_P3 + 6400;   // 25632
constexpr int HEADS_XBYTES = 2 * HEADS_LO;
// the weight fragments of a slice (9 taps x 16 outputs x [hi x32 | lo x32]) live in LDS too, in the same plane form (one
// 16-byte slot per (tap, output row) in each of 8 planes of 9 * 256 B): 72 registers per lane otherwise, which with the
// four steps' accumulators no longer fit two waves per SIMD
constexpr int HEADS_WPLANE = 9 * 16 * 16;    // 2304 B = 9 * 256
constexpr int HEADS_SMEM = HEADS_XBYTES + 8 * HEADS_WPLANE;
static_assert(HEADS_P1 >= HEADS_PLANE && HEADS_P1 % 256 == 0 && (HEADS_P3 - HEADS_P2) % 256 == 0 && HEADS_LO % 16 == 0, "plane layout");

__device__ inline int heads_plane(int c) { return c == 0 ? 0 : c == 1 ? HEADS_P1 : c == 2 ? HEADS_P2 : HEADS_P3; }

template <int HEADS_TILES>
__global__ __launch_bounds__(256, 2) void heads_halo_kernel(const IgemmParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[HEADS_SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // a workgroup owns 16 consecutive grid rows of one sample.  XCD-aware order: block b runs on XCD b % 8; every XCD
    // gets a contiguous run of workgroups so the halo rows two neighbours share come out of the same L2
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

    // zero the two padding columns of every plane once (no slice ever writes them)
    for (int i = tid; i < 8 * HEADS_RING * 2; i += 256) {
        const int pl = i / (HEADS_RING * 2), rem = i - pl * (HEADS_RING * 2);
        const int r = rem >> 1, side = rem & 1;
        *reinterpret_cast<uint4*>(smem + (pl >= 4 ? HEADS_LO : 0) + heads_plane(pl & 3) + (r * HEADS_WP + (side ? HEADS_WP - 1 : 0)) * 16) = make_uint4(0, 0, 0, 0);
    }

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.seg[0].ptr), 0, (int)p.seg_bytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, (int)p.w_bytes, 0x00020000);

    // loader: float4 idx = tid + 256 j -> pixel idx/8 (row-major over 4 rows x 64), quad idx%8 (4 channels of the 32-wide slice)
    // A step brings in 4 new rows; the first step of a slice 6 (two extra passes over rows -1, 0 handled as a half step).
    const int lq = tid & 7;                                  // quad: k-chunk lq >> 1, 8-byte half lq & 1
    const int l_dst = heads_plane(lq >> 1) + (lq & 1) * 8;   // + (ring_row * 66 + x + 1) * 16
    constexpr int SETS = HEADS_TILES == 2 ? 2 : 1;          // HEADS_TILES == 2: fetches run TWO stages ahead (two register sets)
    f32x4 rx[SETS][HEADS_LOADS];
    // rows row0 .. row0 + nrows - 1 of the sample (nrows <= 4) into registers
    auto gload = [&](int chunk, int row0, int nrows, int set = 0) {
#pragma unroll
        for (int j = 0; j < HEADS_LOADS; ++j) {
            const int pix = (tid + 256 * j) >> 3;
            const int r = pix >> 6, x = pix & 63;
            const int gy = row0 + r;
            const unsigned off = (r < nrows && gy >= 0 && gy < p.Hg)
                ? (unsigned)(((((long long)n * p.Hg + gy) * HEADS_W + x) * HEADS_CIN + lq * 4) * 4) : 0xFFFFFFF0u;
            rx[set][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, chunk * 128, 0));
        }
    };
    auto lstore = [&](int row0, int nrows, int set = 0) {
#pragma unroll
        for (int j = 0; j < HEADS_LOADS; ++j) {
            const int pix = (tid + 256 * j) >> 3;
            const int r = pix >> 6, x = pix & 63;
            if (r >= nrows) continue;
            const int ring = (row0 + r + HEADS_RING) % HEADS_RING;          // row -1 -> slot 5
            char* dst = smem + l_dst + (ring * HEADS_WP + x + 1) * 16;
            const f32x4 v = rx[set][j];
            const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
            fp16x2 l01, l23;
            l01[0] = (__fp16)(v[0] - (float)h01[0]); l01[1] = (__fp16)(v[1] - (float)h01[1]);
            l23[0] = (__fp16)(v[2] - (float)h23[0]); l23[1] = (__fp16)(v[3] - (float)h23[1]);
            *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
            *reinterpret_cast<uint2*>(dst + HEADS_LO) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
        }
    };

    // MFMA operands: A = weights (row = output l%16), B = pixels (column = pixel l%16); k = 8 (l/16) + i
    const int li = lane & 15, lg = lane >> 4;
    const char* wsm = smem + HEADS_XBYTES + lg * HEADS_WPLANE + li * 16;   // + tap * 256 (+ 4 planes for lo)
    const char* xs = smem + heads_plane(lg) + (li + 1) * 16;             // + (ring_row * 66 + 16 m + dx) * 16

    f32x4 acc[HEADS_TILES][4];
#pragma unroll
    for (int t = 0; t < HEADS_TILES; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Slice-major walk: for each 32-channel slice the workgroup sweeps its 16 rows in 4 steps with a rolling 6-row window
    // (18 rows fetched per 16 rows of output instead of 24); the four steps' accumulators stay in registers across the
    // slices.  Per slice: a 2-row preamble (rows gy_first - 1, gy_first), then four 4-row fetches (rows gy_first + 1 + 4 s ..);
    // the next fetch is in flight while a step computes.
    auto load_weights = [&](int chunk, f32x4* wq) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int i = tid + 256 * j;                      // 16-byte piece: tap i / 128, row (i % 128) / 8, piece i % 8
            const int t = i >> 7, row = (i >> 3) & 15, c8 = i & 7;
            const unsigned off = i < 9 * 128 ? (unsigned)(row * p.K * 4 + (t * HEADS_CHUNKS + chunk) * 128 + c8 * 16) : 0xFFFFFFF0u;
            wq[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0));
        }
    };
    auto store_weights = [&](const f32x4* wq) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int i = tid + 256 * j;
            const int t = i >> 7, row = (i >> 3) & 15, c8 = i & 7;
            if (i < 9 * 128) *reinterpret_cast<f32x4*>(smem + HEADS_XBYTES + c8 * HEADS_WPLANE + (t * 16 + row) * 16) = wq[j];
        }
    };
    auto compute = [&](int step) {
            // This WAVE owns the 16-pixel column block `wave` of the step's four grid rows (not one full row): an input fragment
        // (ring row, dx) then serves up to three output rows (dy = -1, 0, 1) out of one LDS read.  Loop nest (dx; input row; output
        // row): per dx the three taps' weight fragments (6 reads) stay in registers while the six ring rows go by (12 reads):
        // 54 ds_read_b128 per 108 MFMAs instead of 90 -- the kernel was bound by its LDS reads.  An output pixel's chain of MFMAs
        // is now ordered (slice; dx; dy) with (wl xh, wh xl, wh xh) inside -- the same for every launch shape of this kernel.
        const int gy0 = gy_first + 4 * step;                               // first grid row of the step
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            f16x8 wh[3], wl[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                wh[dy] = *reinterpret_cast<const f16x8*>(wsm + (dy * 3 + dx + 1) * 256);
                wl[dy] = *reinterpret_cast<const f16x8*>(wsm + (dy * 3 + dx + 1) * 256 + 4 * HEADS_WPLANE);
            }
#pragma unroll
            for (int rr = 0; rr < HEADS_RING; ++rr) {                      // input rows gy0 - 1 .. gy0 + 4
                const int slot = (gy0 - 1 + rr + HEADS_RING) % HEADS_RING;
                const char* xrow = xs + slot * (HEADS_WP * 16) + wave * 256 + dx * 16;
                const f16x8 xh = *reinterpret_cast<const f16x8*>(xrow);
                const f16x8 xl = *reinterpret_cast<const f16x8*>(xrow + HEADS_LO);
#pragma unroll
                for (int o = 0; o < HEADS_TH; ++o) {                       // output row gy0 + o reads input row gy0 + o + dy
                    const int dy = rr - 1 - o;
                    if (dy < -1 || dy > 1) continue;
                    acc[step][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[dy + 1], xh, acc[step][o], 0, 0, 0);
                    acc[step][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[dy + 1], xl, acc[step][o], 0, 0, 0);
                    acc[step][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[dy + 1], xh, acc[step][o], 0, 0, 0);
                }
            }
        }
    };

    if constexpr (HEADS_TILES == 2) {
        // TWO-AHEAD schedule (large launches): the kernel is latency-bound on its input stream -- a step's compute (~1 us) is shorter than
        // the fetch it hides --, so the fetch of stage k + 2 is issued when stage k's registers are free: twice the bytes in flight per
        // workgroup.  Stages per slice: preamble (rows gy_first - 1, gy_first), step 0 (4 rows), step 1 (4 rows); 4 slices = 12 stages,
        // register set k & 1.  The workgroup owns 8 rows (10 fetched per 8).
        auto rows_of = [&](int k, int* row0, int* nrows) {
            const int j = k % 3;
            *row0 = j == 0 ? gy_first - 1 : gy_first + 1 + 4 * (j - 1);
            *nrows = j == 0 ? 2 : 4;
        };
        auto fetch = [&](int k) {
            if (k >= 3 * HEADS_CHUNKS) return;
            int row0, nrows;
            rows_of(k, &row0, &nrows);
            gload(k / 3, row0, nrows, k & 1);
        };
        fetch(0);
        fetch(1);
#pragma unroll
        for (int k = 0; k < 3 * HEADS_CHUNKS; ++k) {
            const int j = k % 3;
            int row0, nrows;
            rows_of(k, &row0, &nrows);
            if (j == 0) {
                f32x4 wq[5];
                load_weights(k / 3, wq);
                __syncthreads();                      // every wave is done with the previous slice's last step
                store_weights(wq);
                lstore(row0, nrows, k & 1);
                fetch(k + 2);
            } else {
                if (j == 2) __syncthreads();          // every wave is done reading the rows this store replaces
                lstore(row0, nrows, k & 1);
                fetch(k + 2);
                __syncthreads();
                compute(j - 1);
            }
        }
    } else {
    gload(0, gy_first - 1, 2);
#pragma unroll 1
    for (int chunk = 0; chunk < HEADS_CHUNKS; ++chunk) {
        // weight fragments of this slice: 9 taps x 16 rows x 128 B ([hi x32 | lo x32]; the panel's K order is (tap, cin))
        f32x4 wq[5];
        load_weights(chunk, wq);
        __syncthreads();                          // every wave is done with the previous slice's last step
        store_weights(wq);
        lstore(gy_first - 1, 2);
        gload(chunk, gy_first + 1, 4);
#pragma unroll
        for (int step = 0; step < HEADS_TILES; ++step) {
            if (step) __syncthreads();            // every wave is done reading the rows this store replaces
            lstore(gy_first + 1 + 4 * step, 4);
            if (step + 1 < HEADS_TILES) gload(chunk, gy_first + 1 + 4 * (step + 1), 4);      // flies under this step's MFMAs
            else if (chunk + 1 < HEADS_CHUNKS) gload(chunk + 1, gy_first - 1, 2);
            __syncthreads();
            compute(step);
        }
    }
    }

    // epilogue.  lane: outputs 4 lg .. 4 lg + 3 = (x, y, z, prob) of phase lg for grid pixel (gy_first + 4 step + o, 16 wave + li)
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + lg * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + lg * 4);
#pragma unroll
    for (int step = 0; step < HEADS_TILES; ++step) {
#pragma unroll
        for (int o = 0; o < HEADS_TH; ++o) {
            const int oy = 2 * (gy_first + step * HEADS_TH + o) + (lg >> 1);
            const int ox = 2 * (16 * wave + li) + (lg & 1);
            f32x4 v = acc[step][o], q;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
            q[0] = tanhf(v[0]); q[1] = tanhf(v[1]); q[2] = tanhf(v[2]);
            q[3] = 1.f / (1.f + __expf(-v[3]));
            *reinterpret_cast<f32x4*>(p.out + (((size_t)n * p.Hout + oy) * p.Wout + ox) * 4) = q;
        }
    }
}

}  // namespace

bool heads_halo_supported(const IgemmParams& p)
{
    return p.prec == PREC_F16X3 && p.mode == EPI_HEAD && p.ntaps == 9 && p.Cout == 16 && p.Wg == HEADS_W && p.Hg % (HEADS_TH * HEADS_TILES_BIG) == 0 &&
           p.Hin == p.Hg && p.Win == p.Wg && p.in_stride == 1 && p.os == 2 && p.Hout == 2 * p.Hg && p.Wout == 2 * p.Wg &&
           p.seg[0].C == HEADS_CIN && p.seg[0].cstride == HEADS_CIN && p.seg[0].coff == 0 && p.seg[1].C == 0 && p.ksplit <= 1 &&
           p.dy[0] == -1 && p.dx[0] == -1 && p.dy[8] == 1 && p.dx[8] == 1;
}

hipError_t launch_heads_halo(const IgemmParams& p, hipStream_t s)
{
    // few samples: one 4-row step per workgroup (16 workgroups per sample instead of 4); an output pixel's chain of MFMAs over
    // (slice, tap) is the same either way.  Measured per pass (tools/time_small.py): 16 inputs -20 us, 32 inputs -18 us, 64 inputs +-0 or worse
    static const int small_below = dev_env("P2P_HEADS_SMALL_BELOW") ? atoi(dev_env("P2P_HEADS_SMALL_BELOW")) : 130;      // development builds: workgroups of the 16-row form below which the 4-row form runs
    const bool small = p.N * (p.Hg / (HEADS_TH * HEADS_TILES_BIG)) < small_below;
    const int n_wgs = p.N * (p.Hg / (HEADS_TH * (small ? 1 : HEADS_TILES_BIG)));
    const int per_xcd = (n_wgs + 7) / 8;
    static const bool two_ahead = dev_env("P2P_HEADS_TWO_AHEAD") != nullptr && atoi(dev_env("P2P_HEADS_TWO_AHEAD")) != 0;      // development switch (A/B; same bits)
    if (small) hipLaunchKernelGGL(heads_halo_kernel<1>, dim3(per_xcd * 8), dim3(256), 0, s, p);
    else if (two_ahead && p.Hg % (HEADS_TH * 2) == 0) {
        const int n2 = p.N * (p.Hg / (HEADS_TH * 2));
        hipLaunchKernelGGL(heads_halo_kernel<2>, dim3((n2 + 7) / 8 * 8), dim3(256), 0, s, p);
    } else hipLaunchKernelGGL(heads_halo_kernel<HEADS_TILES_BIG>, dim3(per_xcd * 8), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace p2p
