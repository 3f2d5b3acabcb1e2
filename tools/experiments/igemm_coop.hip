// EXPERIMENT, not part of the library (measured in round 3, same bits and the same 237 ns per K-step as pix2pose_amd/csrc/igemm_stream.hip:
// both are bound by what one CU can stream, DESIGN.md section 3 "Small launches").  To try it again: copy next to igemm_stream.hip, add it to
// pix2pose_amd/build.py SOURCES, declare launch_igemm_coop in kernels.h and call it instead of launch_igemm_stream in model.hip:launch_prepared.
//
// Small-launch implicit GEMM, cooperative form (gfx950, PREC_F16X3): one workgroup per 32x32 (or 64x32) output tile; four LOADER waves
// stream the operands global -> registers (8 K-steps in flight) -> LDS, one or two MFMA waves walk the tile's chain out of LDS.
//
// Same contract as igemm_stream.hip -- an output element is the same chain of v_mfma_f32_32x32x16_f16 over the batched kernel's K-step
// order (StreamOrder), so a crop's bits do not depend on the batch it travels in -- and the same launch interface.  What changed is who
// issues what.  A lone wave doing everything took 235 ns per K-step where its MFMA chain needs 80 (tools/mfma_chain.hip: a dependent
// 32x32x16 MFMA issues every 32.1 cycles at 2.4 GHz): a wave issues in order, so the ~100 other instructions of a K-step are ADDED
// to the chain, and a buffer_load_dwordx4 in the MFMA operand layout (lane = row: 32 rows per instruction) occupies the texture
// path for 37 ns against 19 ns for the row-major form (8 lanes = one 128-byte record, tools/load_pattern.hip) -- and that was with
// the operand split (58 VALU instructions per step) still to come.  Depth of the pipeline, the split and where the weights are cached
// made no difference to the lone wave (DESIGN.md section 5 lists the measurements).  Here:
//   * waves MW .. MW+3 (loaders): per K-step each thread fetches 16 bytes of MW + 1 operand rows in the row-major form, 8 steps ahead,
//     splits the activations hi/lo with the loaders' exact conversion, and writes both tiles as [hi x32 | lo x32] 128-byte rows into
//     one of three LDS stages (same swizzle as the batched kernels' weight tile);
//   * waves 0 .. MW-1 (MFMA): read the fragments of step s + 1 while the six MFMAs of step s run; nothing else is in their stream;
//   * one s_barrier per K-step with an LDS-only wait (a __syncthreads() would also wait for the loaders' global loads in flight).
// Stage protocol (buffer = step mod 3): at barrier B_s steps <= s + 1 are in LDS and the MFMA waves have finished reading step s;
// after it the loaders overwrite buffer (s + 2) mod 3, whose step s - 1 was last read before B_{s-1}.
#include "../../pix2pose_amd/csrc/kernels.h"
#include <cstdlib>

namespace p2p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr unsigned OOB = 0x80000000u;          // every tensor on this path is < 2 GB (igemm_stream_supported)
constexpr int MAX_STEPS = 512;                 // K-steps of one launch (conv4: 400)

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int MW>
__global__ __launch_bounds__(64 * (MW + 4)) void igemm_coop_kernel(const IgemmParams p, const StreamMulti mp)
{
    constexpr int BM = 32 * MW;
    constexpr int NT = 64 * (MW + 4);                      // MW MFMA waves + 4 loader waves
    constexpr int D = 8;                                   // K-steps of global loads in flight per loader thread
    constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + 32 * 128;
    constexpr int AP = MW;                                 // A rows per loader thread (256 loader threads: 32 rows x 8 pieces per pass)
    __shared__ __attribute__((aligned(16))) char smem[3 * STAGE];
    __shared__ int2 s_step[MAX_STEPS];         // x: byte shift of the A gather (tap shift + channel slice), y: weight K offset (bytes) | tap << 24 | segment << 31
    __shared__ int row_out[BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const StreamPhase& P = mp.ph[mp.n > 1 ? blockIdx.y : 0];
    const StreamOrder& o = P.o;
    const int tiles_n = p.Cout / 32;
    const int tile_n = blockIdx.x % tiles_n;
    const int tile_m = blockIdx.x / tiles_n;
    const int n0 = tile_n * 32;
    const int m0 = tile_m * BM;
    const int cpt = p.chunks_per_tap;
    const int ntaps = P.ntaps;
    int ks0 = 0, total = ntaps * cpt;
    if (p.ksplit > 1) {                        // split-K (dense_enc): K-steps [ks0, ks0 + total), raw partial sums
        const int ks_per = (p.ksteps + p.ksplit - 1) / p.ksplit;
        ks0 = blockIdx.y * ks_per;
        total = max(0, min(p.ksteps, ks0 + ks_per) - ks0);
    }

    // ---- K-step table: step -> (group, slice, tap of the group), the batched kernel's loop nest
    for (int i = tid; i < total; i += NT) {
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
    const int HgWg = p.Hg * p.Wg;
    if (tid < BM) {
        const int m = m0 + tid;
        int op = -1;
        if (m < p.M) {
            const int n = m / HgWg, rem = m - n * HgWg, gy = rem / p.Wg, gx = rem - gy * p.Wg;
            op = p.ksplit > 1 ? m : (n * p.Hout + gy * p.os + P.oy) * p.Wout + gx * p.os + P.ox;
        }
        row_out[tid] = op;
    }
    lds_barrier();                             // tables
    const int total_pad = (total + D - 1) / D * D;

    if (wave >= MW) {
        // =========================================================== loader waves
        const int lt = tid - 64 * MW, lrow = lt >> 3, lq = lt & 7;
        unsigned a_off0[AP], a_off1[AP], a_mask[AP];
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            const int m = m0 + lrow + 32 * j;
            a_off0[j] = a_off1[j] = 0;
            a_mask[j] = 0;
            if (m < p.M) {
                const int n = m / HgWg, rem = m - n * HgWg, gy = rem / p.Wg, gx = rem - gy * p.Wg;
                const int iy0 = gy * p.in_stride, ix0 = gx * p.in_stride;
                const unsigned pix = (unsigned)((n * p.Hin + iy0) * p.Win + ix0);
                a_off0[j] = (pix * (unsigned)p.seg[0].cstride + (unsigned)(p.seg[0].coff + lq * 4)) * 4u;
                const unsigned pix1 = p.seg1_stride ? (unsigned)((n * p.seg1_Hin + gy * p.seg1_stride) * p.seg1_Win + gx * p.seg1_stride) : pix;
                a_off1[j] = (pix1 * (unsigned)p.seg[1].cstride + (unsigned)(p.seg[1].coff + lq * 4)) * 4u;
                unsigned mk = 0;
                for (int t = 0; t < ntaps; ++t) {
                    const int iy = iy0 + (int)P.dy[t], ix = ix0 + (int)P.dx[t];
                    if ((unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) mk |= 1u << t;
                }
                a_mask[j] = mk;
            }
        }
        const unsigned b_off = ((unsigned)(n0 + lrow) * (unsigned)P.K + (unsigned)(lq * 4)) * 4u;
        const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.seg[0].ptr, 0, p.seg_bytes[0], 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.seg[1].ptr ? p.seg[1].ptr : p.seg[0].ptr), 0, p.seg_bytes[1], 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, P.w_bytes, 0x00020000);
        // LDS destinations: 128-byte rows [hi x32 | lo x32], the 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7) (igemm_halo.hip)
        int a_dst[AP];
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            const int r = lrow + 32 * j;
            a_dst[j] = r * 128 + ((((lq >> 1) ^ ((r >> 1) & 7))) << 4) + (lq & 1) * 8;          // hi quad; the lo quad sits at ^ 64
        }
        const int b_dst = A_BYTES + lrow * 128 + ((lq ^ ((lrow >> 1) & 7)) << 4);

        f32x4 ra[D][AP], rb[D];
        int2 e_nxt = s_step[0];                 // the table entry of a step is read one step early: no LDS round trip in front of the loads
        auto issue = [&](int s, int step) {
            const bool live = step < total;                                    // wave-uniform
            const int2 e = e_nxt;
            e_nxt = s_step[min(step + 1, MAX_STEPS - 1)];
            const int a_toff = __builtin_amdgcn_readfirstlane(e.x);
            const int w1 = __builtin_amdgcn_readfirstlane(e.y);
            const bool s1 = w1 < 0;
            const unsigned bit = live ? 1u << ((w1 >> 24) & 31) : 0u;
            const int koff = w1 & 0x00FFFFFF;
#pragma unroll
            for (int j = 0; j < AP; ++j) {
                const unsigned off = (a_mask[j] & bit) ? (s1 ? a_off1[j] : a_off0[j]) + (unsigned)a_toff : OOB;
                ra[s][j] = __builtin_bit_cast(f32x4, s1 ? __builtin_amdgcn_raw_buffer_load_b128(rs_a1, off, 0, 0)
                                                        : __builtin_amdgcn_raw_buffer_load_b128(rs_a0, off, 0, 0));
            }
            rb[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, live ? b_off + (unsigned)koff : OOB, 0, 0));
        };
        auto store = [&](int s, int step) {
            char* buf = smem + (step % 3) * STAGE;
#pragma unroll
            for (int j = 0; j < AP; ++j) {
                const f32x4 v = ra[s][j];
                const fp16x2 h01 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), h23 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
                fp16x2 l01, l23;          // residuals are exact in fp32; round them to nearest
                l01[0] = (__fp16)(v[0] - (float)h01[0]); l01[1] = (__fp16)(v[1] - (float)h01[1]);
                l23[0] = (__fp16)(v[2] - (float)h23[0]); l23[1] = (__fp16)(v[3] - (float)h23[1]);
                *reinterpret_cast<uint2*>(buf + a_dst[j]) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
                *reinterpret_cast<uint2*>(buf + (a_dst[j] ^ 64)) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
            }
            *reinterpret_cast<f32x4*>(buf + b_dst) = rb[s];
        };

#pragma unroll
        for (int s = 0; s < D; ++s) issue(s, s);
        store(0, 0);
        issue(0, D);
        store(1, 1);
        issue(1, D + 1);
        lds_barrier();                                                         // B_0
        // No branches inside the loop (every wave walks total_pad steps; steps past the end load nothing and store zeros into stages
        // nobody reads any more): at a control-flow join the compiler's wait-count bookkeeping assumes the worst and drains the pipeline.
        for (int s0 = 0; s0 < total_pad; s0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int s = s0 + u;
                store((u + 2) % D, s + 2);
                issue((u + 2) % D, s + 2 + D);
                lds_barrier();                                                 // B_{s+1}
            }
        }
        return;
    }

    // =============================================================== MFMA waves: rows 32 wave .. of the tile
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int arow = wave * 32 + li;
    int a_sw[2][2], b_sw[2][2];                 // [k block][hi / lo] byte offsets of this lane's fragments inside a stage
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            a_sw[kb][hf] = arow * 128 + (((kb * 2 + lk + 4 * hf) ^ ((arow >> 1) & 7)) << 4);
            b_sw[kb][hf] = A_BYTES + li * 128 + (((kb * 2 + lk + 4 * hf) ^ ((li >> 1) & 7)) << 4);
        }
    f16x8 fa[2][2][2], fb[2][2][2];             // [set][k block][hi / lo]
    auto fread = [&](int set, int step) {
        const char* buf = smem + (step % 3) * STAGE;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                fa[set][kb][hf] = *reinterpret_cast<const f16x8*>(buf + a_sw[kb][hf]);
                fb[set][kb][hf] = *reinterpret_cast<const f16x8*>(buf + b_sw[kb][hf]);
            }
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][kb][1], fb[set][kb][0], acc, 0, 0, 0);      // al bh
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][kb][0], fb[set][kb][1], acc, 0, 0, 0);      // ah bl
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][kb][0], fb[set][kb][0], acc, 0, 0, 0);      // ah bh
        }
    };
#ifdef P2P_COOP_TIMING
    const long long t_all = clock64();
#endif
    lds_barrier();                                                             // B_0: steps 0, 1 are in LDS
    fread(0, 0);
    int s = 0;
    for (; s + 2 <= total; s += 2) {           // no branch inside (see the loaders' loop)
        fread(1, s + 1);                       // lands while this step's MFMAs run: keep the reads IN FRONT of the chain (sched_barrier:
        __builtin_amdgcn_sched_barrier(0);     // left alone the scheduler sinks them behind the MFMAs and the wave then waits for LDS)
        mma(0);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();                         // B_{s+1} (waits for the fragment reads: the loaders may then overwrite their stage)
        fread(0, s + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(1);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
    }
    if (s < total) {
        mma(0);
        lds_barrier();
        ++s;
    }
    for (; s < total_pad; ++s) lds_barrier();  // the loaders' padded steps
#ifdef P2P_COOP_TIMING
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && total >= 100)
        printf("coop: %d steps, MFMA wave %lld cycles\n", total, (long long)(clock64() - t_all));
#endif

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    //      Same expression as the batched kernels: fmaf(acc, scale, shift) + residual, activation.
    const int col = n0 + li;
    if (p.ksplit > 1) {            // raw partial sums [split][m][Cout]; scale / shift / activation belong to the reduction
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = row_out[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk];
            if (m >= 0) p.partial[((size_t)blockIdx.y * p.M + m) * p.Cout + col] = acc[r];
        }
        return;
    }
    const float sc = P.scale ? P.scale[col] : 1.f;
    const float sh = P.shift ? P.shift[col] : 0.f;
    int ops[16];
    float rs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        ops[r] = row_out[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk];
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
        float v = fmaf(acc[r], sc, sh) + rs[r];
        if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.act == ACT_LEAKY) v = v > 0.f ? v : v * p.alpha;
        p.out[(size_t)ops[r] * p.out_cstride + p.out_coff + col] = v;
    }
}

}  // namespace

// same launch interface as launch_igemm_stream (igemm_stream.hip)
hipError_t launch_igemm_coop(const IgemmParams& p, const StreamMulti& mp, hipStream_t s)
{
    const int ny = mp.n > 1 ? mp.n : (p.ksplit > 1 ? p.ksplit : 1);
    const int tiles32 = ((p.M + 31) / 32) * (p.Cout / 32);
    // 64 x 32 tiles (two MFMA waves sharing the weight rows) once the 32 x 32 ones would put more than ~2 workgroups on every CU
    if (tiles32 * ny > 512) hipLaunchKernelGGL((igemm_coop_kernel<2>), dim3(((p.M + 63) / 64) * (p.Cout / 32), ny), dim3(384), 0, s, p, mp);
    else hipLaunchKernelGGL((igemm_coop_kernel<1>), dim3(tiles32, ny), dim3(320), 0, s, p, mp);
    return hipGetLastError();
}

}  // namespace p2p
